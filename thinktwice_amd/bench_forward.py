"""bench.py workload: the full `forward_inference` at the thinktwice.py configuration
(BASELINE.json configs[2]: encoder + 5-stage decoder, batch 8 per GPU, synthetic inputs resident in
HBM, random-init weights of the reference architecture)."""
import os
import time

import torch

from . import model as tm
from . import ops, params, synth

MFMA_PEAK_TF = {"f32": 157.3, "bf16": 2500.0}


class ForwardWorkload:
    def __init__(self, batch, device, dtype=None):
        dtype = dtype or os.environ.get("TT_BENCH_DTYPE", "bf16")
        self.dtype = dtype
        tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
        self.B = batch
        self.name = (f"forward_inference thinktwice.py cfg: {batch} frames x (2 sweeps x 4 cams x 448x896 + "
                     f"65536-pt LiDAR), ResNet50+PAFPN+DepthNet+UNet+LSS splat, LidarNet, fusion, 5-stage decoder")
        self.precision_note = ("bf16 storage / f32 accumulate in the camera + LiDAR trunks and the value projections; "
                               "BEV fusion, depth softmax / lift-splat and the decoder heads in f32"
                               if dtype == "bf16" else "f32 everywhere (exact-f32 MFMA), parity mode")
        self.model, self.cfg = tm.build_thinktwice(dtype=tdt, device=str(device))
        sd = params.init_params(self.cfg, seed=0)
        self.model.load_state_dict(sd)
        del sd
        rank = int(os.environ.get("RANK", "0"))
        # weak scaling: rank r owns global frames [r*batch, (r+1)*batch)
        self.batch = tm.batch_to_device(synth.make_batch(batch, seed=1234 + rank * batch), device)
        self.last = None
        # launch mode: eager by default -- at batch 8 the host runs ahead of the GPU through the camera trunk and
        # the measured step is the same or slightly better than the graph replay (62.96 vs 63.98 ms, r01 v6);
        # TT_BENCH_GRAPH=1 replays one captured HIP graph per forward.  The graph pays at batch 1: see tick_latency.
        self.graph = None
        self.launch_note = "eager launches"
        if os.environ.get("TT_BENCH_GRAPH", "0") != "0":
            try:
                from .encoder_decoder import InferenceGraph
                self.graph = InferenceGraph(self.model, self.batch, channel_last_out=True)
                self.launch_note = "hipGraph replay of the whole forward (LiDAR side stream captured as fork/join)"
            except Exception as e:   # report and fall back: the eager path is the same code
                import sys
                print(f"[bench] HIP graph capture failed ({type(e).__name__}: {e}); using eager launches",
                      file=sys.stderr, flush=True)
                self.graph = None
                torch.cuda.synchronize()

    def step(self):
        if self.graph is not None:
            self.last = self.graph.replay()
        else:
            self.last = self.model.forward_inference(self.batch, channel_last_out=True)
        return self.last

    def frames_per_step(self):
        return self.B

    def roofline(self):
        """Dominant kernel = conv_igemm (MFMA-bound): algorithmic FLOPs of every conv/linear launch of one
        forward divided by the summed launch durations (HIP events on the launch stream)."""
        torch.cuda.synchronize()
        ops.CONV_PROFILE = []
        self.model.use_side_stream = False     # per-launch timing needs the launches serialised on one stream
        self.model.forward_inference(self.batch, channel_last_out=True)   # eager: the graph replay bypasses the hook
        torch.cuda.synchronize()
        self.model.use_side_stream = True
        rec = []
        for r in ops.CONV_PROFILE:
            if len(r) > 4:      # sparse launch: FLOPs of the LIVE rows only
                rows_live = min(int(r[5]), int(r[4].item())) if r[4] is not None else int(r[5])
                r = (r[0] * rows_live, r[1], r[2], r[3].replace("M<=", f"M={rows_live} of <="))
            rec.append(r)
        ops.CONV_PROFILE = None
        flops = sum(r[0] for r in rec)
        ms = sum(r[1].elapsed_time(r[2]) for r in rec)
        ach = flops / (ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TF[self.dtype]
        top = sorted(rec, key=lambda r: -r[1].elapsed_time(r[2]))[:5]
        dump = os.environ.get("TT_BENCH_DUMP")
        if dump and self.dtype != "bf16":      # the f32 parity leg of the default run must not overwrite the bf16 table
            root, ext = os.path.splitext(dump)
            dump = f"{root}.{self.dtype}{ext}"
        if dump:
            import json
            agg = {}
            for r in rec:
                a = agg.setdefault(r[3], [0, 0.0, 0.0])
                a[0] += 1
                a[1] += r[1].elapsed_time(r[2])
                a[2] += r[0]
            rows_ = sorted(({"shape": k, "calls": v[0], "ms": round(v[1], 3), "gflop": round(v[2] / 1e9, 2),
                             "tf": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in agg.items()),
                           key=lambda d: -d["ms"])
            with open(dump, "w") as f:
                json.dump(rows_, f, indent=0)
        dec = [r for r in rec if " N=256 K=256 k1x1s1" in r[3] and not r[3].startswith("sparse")]
        self._decoder_gemm = None
        if dec:
            mmax = max(int(r[3].split()[0][2:]) for r in dec)
            big = [r for r in dec if int(r[3].split()[0][2:]) == mmax]
            gf = sum(r[0] for r in big)
            gms = sum(r[1].elapsed_time(r[2]) for r in big)
            tf = gf / (gms * 1e-3) / 1e12
            # HBM roofline of this GEMM: reads M*K + writes M*N elements once
            esz = 2 if self.dtype == "bf16" else 4
            bytes_ = sum(int(r[3].split()[0][2:]) * (256 + 256) * esz for r in big)
            self._decoder_gemm = {"shape": big[0][3], "launches": len(big), "tflops": round(tf, 1),
                                  "mfma_frac": round(tf / peak, 4),
                                  "hbm_gbs": round(bytes_ / (gms * 1e-3) / 1e9, 1),
                                  "note": "value_proj / fpn_linear GEMM (K=N=256): 128 (bf16) / 64 (f32) FLOP per HBM "
                                          "byte, i.e. HBM-bound below ~1 PF (bf16)"}
        traffic, traffic_note = self._pmc_traffic()
        return {"kernel": "conv_igemm_kernel (all conv/linear launches of one forward)", "bound": "mfma",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                "traffic": traffic, "traffic_note": traffic_note, "launches": len(rec),
                "conv_ms_per_step": round(ms, 3),
                "algorithmic_gflop_per_step": round(flops / 1e9, 1),
                "slowest_launches": [{"gflop": round(r[0] / 1e9, 2), "ms": round(r[1].elapsed_time(r[2]), 3),
                                      "tf": round(r[0] / (r[1].elapsed_time(r[2]) * 1e-3) / 1e12, 1),
                                      "shape": r[3]} for r in top]}

    def _pmc_traffic(self):
        """HBM bytes per step of the same kernel set (every conv/linear launch of one forward), from the committed
        rocprofv3 `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this command (counters cannot be read from inside
        the process): FETCH_SIZE x 2 (gfx950 16 B/lane correction, MI355X_MICROARCH.md) + WRITE_SIZE, in KB."""
        import json
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles",
                            "r01_forward_bf16_v11_pmc.json")
        if self.dtype != "bf16" or self.B != 8 or not os.path.exists(path):
            return None, "no PMC summary for this dtype / batch"
        forwards = 4     # the profiled command: bench.py --steps 2 --warmup 1 (+1 roofline pass)
        rd = wr = 0.0
        for name, c in json.load(open(path)).items():
            if "conv_" in name or "igemm" in name or "splitk" in name:
                rd += c.get("FETCH_SIZE", {}).get("sum", 0.0)
                wr += c.get("WRITE_SIZE", {}).get("sum", 0.0)
        return int((2.0 * rd + wr) * 1024 / forwards), ("bytes per step over the same launches, from profiles/"
                                                         "r01_forward_bf16_v11_pmc.json (separate --pmc passes, "
                                                         "FETCH_SIZE x2 corrected)")

    def extra(self):
        out = {}
        if getattr(self, "_decoder_gemm", None):
            out["decoder_gemm"] = self._decoder_gemm
        if os.environ.get("TT_BENCH_TICK", "1") != "0" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
            out["tick_latency"] = self.tick_latency()
        if self.dtype == "bf16" and os.environ.get("TT_BENCH_F32", "1") != "0" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
            # the same workload in f32 parity mode (<= 1.3e-5 vs the reference goldens), 3 timed steps
            import time
            del self.last
            torch.cuda.empty_cache()
            w = ForwardWorkload(self.B, self.batch["img"].device, dtype="f32")
            w.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                w.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            r = w.roofline()
            out["f32_parity_mode"] = {"value": round(self.B / dt, 3), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3),
                                      "roofline": {k: r[k] for k in ("achieved", "peak", "unit", "frac")},
                                      "decoder_gemm": w._decoder_gemm}
        return out

    def tick_latency(self, ticks=10):
        """BASELINE configs[4] (closed loop): per-tick latency at batch 1, host-synchronised every tick, for the eager
        launches, the captured HIP graph, and the previous-sweep BEV cache under both launch modes."""
        from .encoder_decoder import InferenceGraph
        dev = self.batch["img"].device
        b1 = tm.batch_to_device(synth.make_batch(1, seed=4321), dev)

        def timed(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(ticks):
                fn()
                torch.cuda.synchronize()
            return round((time.perf_counter() - t0) / ticks * 1e3, 3)

        out = {"batch": 1, "ticks": ticks}
        out["eager_ms"] = timed(lambda: self.model.forward_inference(b1, channel_last_out=True))
        key = self.model.forward_inference(b1, channel_last_out=True)["_key_bev_cl"].contiguous().clone()
        out["eager_prev_sweep_cache_ms"] = timed(
            lambda: self.model.forward_inference(b1, channel_last_out=True, prev_bev=key))
        try:
            g = InferenceGraph(self.model, b1, channel_last_out=True)
            out["graph_ms"] = timed(g.replay)
            gc = InferenceGraph(self.model, b1, channel_last_out=True, prev_bev=key)
            out["graph_prev_sweep_cache_ms"] = timed(gc.replay)
        except Exception as e:
            out["graph_error"] = f"{type(e).__name__}: {e}"
        return out

    def cpu_baseline(self):
        """oracle on ONE frame in a bounded subprocess (<= 32 threads, 240 s cap)."""
        import json
        import subprocess
        import sys
        threads = min(os.cpu_count() or 1, 32)
        root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
        try:
            r = subprocess.run([sys.executable, "-m", "thinktwice_amd.cpu_baseline", str(threads)], cwd=root,
                               capture_output=True, text=True, timeout=240)
            return json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:   # timeout / failure: report, never block the bench
            return {"value": None, "unit": "frames/s", "cores": threads, "kind": "port",
                    "sample": f"oracle forward did not finish within 240 s ({type(e).__name__})"}
