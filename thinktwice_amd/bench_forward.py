"""bench.py workload: the full `forward_inference` at the thinktwice.py configuration
(BASELINE.json configs[2]: encoder + 5-stage decoder, batch 8 per GPU, synthetic inputs resident in
HBM, random-init weights of the reference architecture)."""
import os
import time

import torch

from . import model as tm
from . import ops, params, synth

MFMA_PEAK_TF = {"f32": 157.3, "bf16": 2500.0, "f16": 2500.0, "bf16x3": 2500.0, "bf16x3h": 2500.0}
TORCH_DTYPE = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "bf16x3": "f32x3", "bf16x3h": "f32x3h"}
HBM_PEAK_GBS = 8000.0
PROFILE_TAG = "r06"            # profiles/<tag>_forward_<dtype>_{kernel_stats.csv,pmc.json}: the committed rocprofv3 passes of this build


def mfma_per_product(kernel_label, dtype):
    """MFMAs executed per algorithmic product by the kernel a conv launch ran on (label = tt_conv_last_kernel()).  bf16x3
    kernels issue three bf16 MFMAs per product: conv_igemm_glds_kernel<..., GATHER, X3> with X3 = true (the LAST template
    argument -- round 3 tested the GATHER flag and printed 1 for the dominant dense tile), conv_x3_pipe_kernel, the run-staged
    sparse kernel.  The exact-f32 / 16-bit kernels issue one."""
    if dtype not in ("bf16x3", "bf16x3h"):
        return 1
    k = kernel_label.replace(" + tail", "").replace(" pre-split A", "").strip()
    if k.startswith("conv_h2_kernel"):        # half storage x f16 (hi, lo) weights: two f16 MFMAs per product
        return 2
    if k.startswith(("conv_x3_pipe_kernel", "conv_x3_run3_kernel", "sp_conv_runs")):
        return 3
    if k.startswith("conv_igemm_glds_kernel<") and k.endswith(", true>"):
        return 3
    return 1


class ForwardWorkload:
    def __init__(self, batch, device, dtype=None):
        dtype = dtype or os.environ.get("TT_BENCH_DTYPE", "bf16x3h")
        self.dtype = dtype
        tdt = TORCH_DTYPE[dtype]
        self.B = batch
        self.name = (f"forward_inference thinktwice.py cfg: {batch} frames x (2 sweeps x 4 cams x 448x896 + "
                     f"65536-pt LiDAR), ResNet50+PAFPN+DepthNet+UNet+LSS splat, LidarNet, fusion, 5-stage decoder")
        self.precision_note = {
            "f32": "f32 everywhere (exact-f32 MFMA), parity mode",
            "bf16x3h": "f32 storage and bf16x3 products (operands split into bf16 hi+lo pairs, three v_mfma_f32_32x32x16_bf16 per "
                       "product, f32 accumulate) everywhere EXCEPT the PAFPN's ten 3 x 3 convolutions (fpn / downsample / pafpn "
                       "convs, lss.py:284-348), which read an IEEE-half copy of their input and multiply it with an f16 (hi, lo) "
                       "weight pair: two v_mfma_f32_32x32x16_f16 per product (csrc/conv_h2.hip); the neck's sums (laterals, "
                       "top-down, bottom-up) stay f32.  Every output within 1e-3 of the reference, waypoint L2 0.86 mm at B = 8, "
                       "integer work bit-equal (tests/test_forward.py MODES); the all-bf16x3 mode is the `bf16x3_mode` leg",
            "bf16x3": "f32 storage everywhere; the camera + LiDAR trunks and the value projections multiply in bf16x3 "
                      "(operands split into bf16 hi+lo pairs, three v_mfma_f32_32x32x16_bf16 per product, f32 "
                      "accumulate: ~1e-5 relative error); layers outside the LDS-DMA kernel, BEV fusion, lift-splat and "
                      "the decoder heads on the exact f32 path.  Outputs within 1e-3 of the reference (tests/test_forward.py)",
        }.get(dtype, f"{'IEEE half' if dtype == 'f16' else 'bf16'} storage / f32 accumulate in the camera + LiDAR trunks "
                     "and the value projections; BEV fusion, depth softmax / lift-splat and the decoder heads in f32 "
                     "(speed mode: outputs NOT inside the 1e-3 tolerance, see tests/test_forward.py MODES)")
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
        # TT_BENCH_PIPELINE=n: batches in flight (n streams, round-robin; default 3: 88.1 / 87.3 / 84.7 ms per step for 1 / 2 / 3,
        # profiles/r06_pipeline_ab.txt); 1 = one batch at a time on the current stream (reported beside the headline as
        # `one_batch_at_a_time`)
        self.pipeline = int(os.environ.get("TT_BENCH_PIPELINE", "3"))
        self._streams, self._tick = None, 0
        if self.pipeline > 1:
            self.launch_note = f"eager launches, {self.pipeline} batches in flight on alternating streams"
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
        elif self.pipeline > 1:
            # consecutive batches on alternating streams: the latency-bound tail of batch i (the decoder's per-sample kernels:
            # 8 - 64 workgroups on 256 CUs) runs under the camera trunk of batch i + 1.  Every forward is complete inside the
            # timed region (device-wide synchronise at its end); activations of the two batches come from separate allocator pools
            if self._streams is None:
                self._streams = [torch.cuda.Stream(self.batch["img"].device) for _ in range(self.pipeline)]
                for st in self._streams:
                    st.wait_stream(torch.cuda.current_stream())
            st = self._streams[self._tick % self.pipeline]
            self._tick += 1
            with torch.cuda.stream(st):
                self.last = self.model.forward_inference(self.batch, channel_last_out=True)
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
        ops.CONV_BYTES = []
        ops.CONV_KERNELS = []
        self.model.use_side_stream = False     # per-launch timing needs the launches serialised on one stream
        # on a stream the timed steps have already run on: the first forward on a NEW stream pays tens of milliseconds of allocator
        # work inside whatever launch bracket it falls into (measured: 63 - 82 ms on one stride-2 layer, profiles/r06_roofline_warm.txt)
        st = self._streams[0] if self._streams else torch.cuda.current_stream()
        with torch.cuda.stream(st):
            self.model.forward_inference(self.batch, channel_last_out=True)   # eager: the graph replay bypasses the hook
        torch.cuda.synchronize()
        self.model.use_side_stream = True
        rec = []
        compulsory = 0
        for r, nb in zip(ops.CONV_PROFILE, ops.CONV_BYTES):
            if len(r) > 4:      # sparse launch: FLOPs of the EXISTING (row, tap) pairs of the live rows only
                rows_live = min(int(r[5]), int(r[4].item())) if r[4] is not None else int(r[5])
                compulsory += nb[0] * rows_live + nb[1]
                if len(r) > 6 and r[6] is not None:
                    taps_per_row, flop_per_pair = r[6]
                    pairs = int(taps_per_row[:rows_live].sum().item())
                    r = (flop_per_pair * pairs, r[1], r[2],
                         r[3].replace("M<=", f"M={rows_live} of <=") + f" pairs={pairs / max(rows_live, 1):.1f}/row")
                else:
                    r = (r[0] * rows_live, r[1], r[2], r[3].replace("M<=", f"M={rows_live} of <="))
            else:
                compulsory += nb
            rec.append(r)
        kernels = list(ops.CONV_KERNELS)
        ops.CONV_PROFILE = None
        ops.CONV_BYTES = None
        ops.CONV_KERNELS = None
        flops = sum(r[0] for r in rec)
        ms = sum(r[1].elapsed_time(r[2]) for r in rec)
        ach = flops / (ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TF[self.dtype]
        top = sorted(rec, key=lambda r: -r[1].elapsed_time(r[2]))[:5]
        dump = os.environ.get("TT_BENCH_DUMP")
        if dump and getattr(self, "is_leg", False):   # the secondary legs of the default run get their own table
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
        # the decoder's dense GEMMs: value_proj of the five refinement layers as ONE GEMM per FPN level
        # (M = B*4*H_l*W_l, N = 5*256, K = 256), thinktwice_decoder.py:392-393 / MSDA:431
        nv = 256 * self.cfg["cfg"]["refine_num"]
        dec = [r for r in rec if f" N={nv} K=256 k1x1s1" in r[3] and not r[3].startswith("sparse")]
        self._decoder_gemm = None
        if not dec:
            # "sample first, project after" (round 6): the value_proj GEMM does not exist.  The decoder's remaining dense GEMMs are
            # fpn_linear0-3 (thinktwice_decoder.py:330-333, 388-391): one 256 -> 256 linear per FPN level over the key sweep's
            # B*4 images -- HBM-shaped (K = N = 256: 128 FLOP per byte moved at f32)
            H, W = tuple(self.cfg["img_encoder"]["final_dim"])
            lv = {f"M={self.B * 4 * (H // s) * (W // s)} N=256 K=256 k1x1s1" for s in (4, 8, 16, 32)}
            fl = [r for r in rec if r[3] in lv]
            if fl:
                gf = sum(r[0] for r in fl)
                gms = sum(r[1].elapsed_time(r[2]) for r in fl)
                by = sum(int(r[3].split()[0][2:]) * (256 + 256) * 4 for r in fl)
                self._decoder_gemm = {"shape": max(fl, key=lambda r: r[0])[3], "launches": len(fl), "what": "fpn_linear0-3",
                                      "tflops": round(gf / (gms * 1e-3) / 1e12, 1), "mfma_frac": round(gf / (gms * 1e-3) / 1e12 / peak, 4),
                                      "ms": round(gms, 4), "hbm_gbs": round(by / (gms * 1e-3) / 1e9, 1),
                                      "hbm_frac": round(by / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      "note": "the value_proj GEMM of rounds 1-5 is gone (tt_msda_sample_proj_ln applies value_proj to "
                                              "the sampled sums); what is left of the decoder's dense GEMMs is bound by HBM: read M*256 + "
                                              "write M*256 f32 once"}
        if dec:
            esz = 4 if self.dtype in ("f32", "bf16x3", "bf16x3h") else 2
            gf = sum(r[0] for r in dec)
            gms = sum(r[1].elapsed_time(r[2]) for r in dec)
            tf = gf / (gms * 1e-3) / 1e12
            bytes_ = sum(int(r[3].split()[0][2:]) * (256 + nv) * esz for r in dec)   # read M*K, write M*N once
            big = max(dec, key=lambda r: int(r[3].split()[0][2:]))
            bms = big[1].elapsed_time(big[2])
            bM = int(big[3].split()[0][2:])
            self._decoder_gemm = {"shape": big[3], "launches": len(dec), "tflops": round(tf, 1),
                                  "mfma_frac": round(tf / peak, 4),
                                  "largest": {"ms": round(bms, 4), "tflops": round(big[0] / (bms * 1e-3) / 1e12, 1),
                                              "mfma_frac": round(big[0] / (bms * 1e-3) / 1e12 / peak, 4),
                                              "hbm_gbs": round(bM * (256 + nv) * esz / (bms * 1e-3) / 1e9, 1)},
                                  "hbm_gbs": round(bytes_ / (gms * 1e-3) / 1e9, 1),
                                  "hbm_frac": round(bytes_ / (gms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  "flop_per_hbm_byte": round(2.0 * 256 * nv / ((256 + nv) * esz), 1),
                                  "note": "all four levels' value_proj GEMMs (five layers fused along N); bound by "
                                          "whichever of mfma_frac / hbm_frac is larger"}
        # the dominant kernel (most time) on its own: what `rocprofv3 --kernel-trace --stats` lists under the same name
        by_k = {}
        for r, kname in zip(rec, kernels):
            a = by_k.setdefault(kname.replace(" + tail", ""), [0, 0.0, 0.0])
            a[0] += 1
            a[1] += r[1].elapsed_time(r[2])
            a[2] += r[0]
        dk, dv = max(by_k.items(), key=lambda kv: kv[1][1])
        mult = mfma_per_product(dk, self.dtype)
        dom_tf = dv[2] / (dv[1] * 1e-3) / 1e12
        dominant = {"kernel": dk, "launches": dv[0], "avg_launch_ms": round(dv[1] / dv[0], 4),
                    "ms_per_step": round(dv[1], 3), "algorithmic_tflops": round(dom_tf, 1), "frac": round(dom_tf / peak, 4),
                    "mfma_per_product": mult, "executed_mfma_frac": round(mult * dom_tf / peak, 4),
                    "note": "HIP events around each launch of this kernel in one SERIALISED forward (single stream); a "
                            "tail-split launch (256x64 tiles over the last row tiles) is timed with its main launch.  The "
                            f"rocprofv3 --stats average of the same kernel over the timed steps (profiles/{PROFILE_TAG}_forward_"
                            f"{self.dtype}_kernel_stats.csv) is higher because the LiDAR branch's kernels share the chip on a "
                            "second stream there"}
        traffic, traffic_note = self._pmc_traffic()
        x3 = {}
        if self.dtype in ("bf16x3", "bf16x3h"):
            # `achieved` / `frac` count ALGORITHMIC flops (2 M N K) against the dense 16-bit MFMA peak; a product costs three bf16
            # MFMAs on the bf16x3 kernels and two f16 MFMAs on the h2 kernel (1 on the exact-f32 latency kernels, whose peak is
            # lower: counted as 1), so frac is bounded by ~1/3 -- the matrix pipe itself is at `executed_mfma_frac`
            ex = sum(mfma_per_product(k, self.dtype) * r[0] for r, k in zip(rec, kernels)) / (ms * 1e-3) / 1e12
            x3 = {"mfma_per_product": round(ex / ach, 3), "executed_mfma_tflops": round(ex, 1),
                  "executed_mfma_frac": round(ex / peak, 4)}
        return {"kernel": "conv_igemm_glds_kernel (all conv/linear launches of one forward)", "bound": "mfma",
                "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), **x3,
                "dominant_kernel": dominant,
                "traffic": traffic, "traffic_note": traffic_note,
                "compulsory_bytes": int(compulsory),
                "traffic_over_compulsory": None if not traffic else round(traffic / compulsory, 3),
                "compulsory_note": "every operand of every launch moved once: input pixels it touches + output + weights "
                                   "+ residual rows (+ one rulebook row per sparse output row), in the storage dtype",
                "launches": len(rec),
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
        from . import build
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles",
                            f"{PROFILE_TAG}_forward_{self.dtype}_pmc.json")
        if self.B != 8 or not os.path.exists(path):
            return None, "no PMC summary for this dtype / batch"
        data = json.load(open(path))
        stamp = data.get("_stamp", {})
        if stamp.get("csrc_sha") != build.source_fingerprint():
            return None, (f"profiles/{os.path.basename(path)} was collected from another build (source fingerprint "
                          f"{stamp.get('csrc_sha')} != {build.source_fingerprint()}): counters are not quoted; re-run "
                          "tools/pmc_profile.sh + tools/summarize_pmc.py")
        forwards = 4     # the profiled command: bench.py --steps 2 --warmup 1 (+1 roofline pass)
        rd = wr = 0.0
        for name, c in data.items():
            if name != "_stamp" and ("conv_" in name or "igemm" in name or "splitk" in name):
                rd += c.get("FETCH_SIZE", {}).get("sum", 0.0)
                wr += c.get("WRITE_SIZE", {}).get("sum", 0.0)
        return int((2.0 * rd + wr) * 1024 / forwards), ("bytes per step over the same launches, from profiles/"
                                                         f"{os.path.basename(path)} (separate --pmc passes, FETCH_SIZE x2 "
                                                         f"corrected; build {stamp.get('csrc_sha')})")

    def extra(self):
        out = {}
        if self.pipeline > 1 and self.graph is None and os.environ.get("TT_BENCH_SERIAL", "1") != "0":
            # the same steps with ONE batch in flight (every forward starts when the previous one has been issued on the same
            # stream): what a caller that needs each result before the next frame sees
            import time
            keep, self.pipeline = self.pipeline, 1
            try:
                for _ in range(2):
                    self.step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 6
                for _ in range(n):
                    self.step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / n
            finally:
                self.pipeline = keep
            out["one_batch_at_a_time"] = {"value": round(self.B / dt, 3), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3),
                                          "steps": n}
        if getattr(self, "_decoder_gemm", None):
            out["decoder_gemm"] = self._decoder_gemm
        if os.environ.get("TT_BENCH_TICK", "1") != "0" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
            out["tick_latency"] = self.tick_latency()
        if os.environ.get("TT_BENCH_H2D", "1") != "0":
            out["h2d_inclusive"] = self.h2d_inclusive()
        def guarded(fn):       # a failing side leg must not take the headline line down with it
            try:
                return fn()
            except Exception as e:
                return {"error": f"{type(e).__name__}: {e}"[:300]}
        if os.environ.get("TT_BENCH_RAW", "1") != "0":
            out["raw_inclusive"] = guarded(self.raw_inclusive)
        if os.environ.get("TT_BENCH_VOXEL", "1") != "0" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
            out["voxel_pool_op"] = self.voxel_pool_op()
            out["lift_splat"] = guarded(self.lift_splat_op)
        single = int(os.environ.get("WORLD_SIZE", "1")) == 1
        # the same workload in the other precision modes, a few timed steps each: the exact-f32 parity mode and the
        # bf16-storage speed mode (whose outputs are NOT inside the 1e-3 tolerance: tests/test_forward.py MODES)
        # ... and the all-bf16x3 mode (the headline of rounds 2-5; the PAFPN's 3 x 3 layers on three bf16 MFMAs as well)
        legs = [("f32", "f32_parity_mode", "TT_BENCH_F32"), ("bf16x3", "bf16x3_mode", "TT_BENCH_X3"),
                ("bf16", "bf16_speed_mode", "TT_BENCH_BF16")]
        for dt_name, key, env in legs:
            if not single or dt_name == self.dtype or os.environ.get(env, "1") == "0":
                continue
            import time
            self.last = None
            torch.cuda.empty_cache()
            w = ForwardWorkload(self.B, self.batch["img"].device, dtype=dt_name)
            w.is_leg = True
            w.step()
            torch.cuda.synchronize()
            n = 3 if dt_name == "f32" else 8
            t0 = time.perf_counter()
            for _ in range(n):
                w.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            r = w.roofline()
            out[key] = {"value": round(self.B / dt, 3), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3),
                        "roofline": {k: r[k] for k in ("achieved", "peak", "unit", "frac", "conv_ms_per_step")},
                        "decoder_gemm": w._decoder_gemm}
            del w
        if os.environ.get("TT_BENCH_TRAIN", "1") != "0":
            # world > 1: the leg ran on EVERY rank in collect() (it issues the iteration's collectives); rank 0 reports it
            out["train_step"] = self.train_step_leg() if single else getattr(self, "_train_leg", None)
        return out

    def collect(self):
        """bench.py calls this on every rank after the timed steps.  With more than one rank the data-parallel training
        iteration (BASELINE config 4) runs here on all of them, so that the N-GPU record holds the ONE collective the design is
        built around: time, bytes and achieved xGMI bandwidth of the flat-gradient all-reduce over RCCL."""
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and os.environ.get("TT_BENCH_TRAIN", "1") != "0":
            self._train_leg = self.train_step_leg()

    def train_step_leg(self, iters=3):
        """BASELINE config 4 on ONE GPU, timed inside the default bench run so that the driver's record holds it (VERDICT r2
        item 4): trainer.Trainer.step at the same batch (forward_train + teacher pass + 23 losses, tape backward, flat
        all-reduce -- a no-op at world size 1 --, clip + AdamW, operand re-preparation) under model.train() semantics."""
        import time
        from .bench_train import TrainStepWorkload
        world = int(os.environ.get("WORLD_SIZE", "1"))
        self.last = None
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        try:
            w = TrainStepWorkload(self.B, self.batch["img"].device, dtype="bf16x3")
            w.step()                                    # warm-up (allocator, lazy buffers)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                w.step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / iters
            w.collect()
            r = w.roofline()
            leg = {"value": round(self.B / dt, 3), "unit": "samples/s", "ms_per_iteration": round(dt * 1e3, 2),
                   "batch": self.B, "iterations_timed": iters, "dtype": "bf16x3 forward / input gradients / weight gradients of the >= 64-channel layers, f32 for the rest",
                   "semantics": w._phases.get("batchnorm"), "wgrad_tflops": r["wgrad"]["tflops"], "wgrad_ms": r["wgrad"]["ms"],
                   "conv_launches": r["launches"], "conv_kernel_ms": r["kernel_ms"],
                   "all_reduce_ms": w._phases.get("all_reduce_ms"), "clip_adamw_ms": w._phases.get("clip_adamw_ms"),
                   "prepare_operands_ms": w._phases.get("prepare_operands_ms"), "peak_memory_gb": w._phases.get("peak_memory_gb"),
                   "loss": w._phases.get("loss"), "n_gpus": world,
                   "note": ("one process, one GPU: the gradient all-reduce is the world-size-1 short cut; the N-GPU record "
                            "(bench.py --gpus N) times the RCCL all-reduce in this leg") if world == 1 else
                           "data parallel, one rank per GPU, batch per GPU fixed (weak scaling); value = all ranks' samples / s"}
            if world > 1:
                leg["value"] = round(self.B * world / dt, 3)
                leg["all_reduce"] = getattr(w, "_all_reduce", None) or w.all_reduce_report()      # (collect() has timed it)
            del w
            torch.cuda.empty_cache()
            return leg
        except Exception as e:          # a failing side leg must not take the headline line down with it
            return {"error": f"{type(e).__name__}: {e}"[:300]}

    def h2d_inclusive(self, steps=5):
        """The same step with the batch handed over as HOST buffers (pinned): H2D copy of the f32 images / points +
        forward, per step.  Never the headline `value` (inputs resident in HBM), reported beside it."""
        keys = ("img", "points", "speed", "target_point", "target_command")
        host = {k: self.batch[k].cpu().pin_memory() for k in keys if torch.is_tensor(self.batch.get(k))}
        dev = self.batch["img"].device
        nbytes = sum(v.numel() * v.element_size() for v in host.values())

        def one():
            b = dict(self.batch)
            for k, v in host.items():
                b[k] = v.to(dev, non_blocking=True)
            return self.model.forward_inference(b, channel_last_out=True)
        one()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return {"value": round(self.B / dt, 3), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3),
                "h2d_bytes_per_step": nbytes, "note": "pinned host batch -> device copy inside the timed step"}

    def raw_inclusive(self, steps=5):
        """north_star's input: "synthetic 4 x 900 x 1600 RGB ... frames".  The same batch-8 step started from the RAW uint8
        camera frames resident in HBM (B x 2 sweeps x 4 cams x 900 x 1600 x 3 = 276 MB): tt_preprocess_images (undistort +
        resize + crop + normalise, thinktwice_agent.py's test_pipeline) inside the timed step, then the forward.  Reported
        beside the headline, whose timed region starts at the network input like the reference's forward_inference."""
        from .preprocess import ImagePreprocessor
        dev = self.batch["img"].device
        B, T = self.batch["img"].shape[:2]
        base = torch.from_numpy(synth.raw_camera_frames(seed=31, T=T)).to(dev)          # (T, 4, 900, 1600, 3) uint8
        raw = torch.stack([torch.roll(base, shifts=13 * b, dims=3) for b in range(B)]).contiguous()
        pre = ImagePreprocessor(final_dim=tuple(self.batch["img"].shape[-2:]), device=dev)

        def one():
            b = dict(self.batch)
            b["img"] = pre(raw)                                                         # (B, T, 4, 3, 448, 896) f32
            return self.model.forward_inference(b, channel_last_out=True)
        one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            pre(raw)
        e1.record()
        torch.cuda.synchronize()
        pre_ms = e0.elapsed_time(e1) / steps
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        nb = raw.numel()
        out_b = B * T * 4 * 3 * self.batch["img"].shape[-2] * self.batch["img"].shape[-1] * 4
        return {"value": round(B / dt, 3), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3),
                "raw_bytes_per_step": nb, "preprocess_ms": round(pre_ms, 3),
                "preprocess_gbps": round((nb + out_b) / (pre_ms * 1e-3) / 1e9, 1),
                "note": "uint8 4 x 900 x 1600 frames resident in HBM -> tt_preprocess_images -> forward, per step"}

    def lift_splat_op(self, launches=20):
        """The splat that actually runs inside the forward (tt_lift_splat_fwd_ws: softmax (x) context, permute and voxel pooling
        fused -- the reference's 514 MB / sample lift volume is never materialised), one launch per sweep, timed on its own
        with HIP events at the bench shape: HBM roofline on its ALGORITHMIC bytes (depth logits + context + geom + out)."""
        dev = self.batch["img"].device
        lss = self.model.img_encoder
        B, N = self.B, self.batch["img"].shape[2]
        D = lss.depth_channels
        fH, fW = lss.final_dim[0] // lss.downsample_factor, lss.final_dim[1] // lss.downsample_factor
        C = lss.output_channels
        g = torch.Generator(device="cpu").manual_seed(3)
        depth = (torch.randn(B * N, fH, fW, D, generator=g) * 2).to(dev)
        ctx = torch.randn(B * N, fH, fW, C, generator=g).to(dev)
        consts = lss.host_constants(self.batch["img_metas"], N)
        geom = lss.geometry(consts["gm"].to(dev), B, N)               # the forward's own frustum -> voxel index (int32 [B, Np, 3])
        vn = tuple(int(v) for v in lss.voxel_num)
        out = torch.zeros(B, vn[1], vn[0], C, dtype=torch.float32, device=dev)
        for _ in range(3):
            ops.lift_splat(depth, ctx, geom, vn, B, N, out=out, record=False)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
        ev[0].record()
        for i in range(launches):
            ops.lift_splat(depth, ctx, geom, vn, B, N, out=out, record=False)
            ev[i + 1].record()
        torch.cuda.synchronize()
        ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(launches))[launches // 2]
        nbytes = depth.numel() * 4 + ctx.numel() * 4 + geom.numel() * 4 + out.numel() * 4
        ach = nbytes / (ms * 1e-3) / 1e9
        return {"kernel": "lift_splat_strip_kernel + lift_splat_cells_kernel (tt_lift_splat_fwd_ws)", "bound": "hbm",
                "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
                "ms_per_launch": round(ms, 4), "algorithmic_bytes_per_launch": int(nbytes),
                "note": "one launch per sweep of the forward (B x 4 cams); 119 MB per launch: a latency / LDS-bound fusion, not a "
                        "stream -- the materialised formulation it replaces moves 4.1 GB per launch"}

    def voxel_pool_op(self, launches=20):
        """The B1 operator boundary (`voxel_pooling_forward_wrapper`, f32 rows) at the thinktwice.py size, next to the
        forward: HBM roofline on COMPULSORY bytes (SURVEY 8d target >= 60 % of 8 TB/s)."""
        import importlib.util
        spec = importlib.util.spec_from_file_location(
            "_tt_bench_main", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        w = mod.VoxelPoolWorkload(self.B, self.batch["img"].device)
        for _ in range(3):
            w.step()
        w.on_warm()
        for _ in range(launches // 2):
            w.step()
        r = w.roofline()
        del w
        torch.cuda.empty_cache()
        return r

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
            # the same tick issued from C: a compiled launch plan (thinktwice_amd/plan.py), tt_encoder_fwd + tt_decoder_fwd.
            # Timed BEFORE the graph legs: after a HIP graph has been captured and replayed in the process, plain multi-stream
            # launches of the same kernels measure ~2.5 ms slower per tick (22.6 vs 20.0 ms, profiles/r04_tick_notes.txt)
            from . import plan as P
            fp = P.compile_forward(self.model, b1, channel_last_out=True)
            out["c_plan_ms"] = timed(fp.run)
            out["c_plan_calls"], out["c_plan_streams"] = fp.calls, fp.nstreams
            out["c_plan_arena_gb"] = round(fp.arena.numel() / 1e9, 2)
            out["c_plan_arena_bump_gb"] = round(fp.recorded_arena_bytes / 1e9, 2)
            del fp
        except Exception as e:
            out["c_plan_error"] = f"{type(e).__name__}: {e}"[:300]
        try:
            # the WHOLE model-side tick from the raw sensors (thinktwice_agent.py:362-529, thinktwice_amd/agent_tick.py): H2D of the
            # four uint8 900 x 1600 frames + tt_preprocess_images + half-sweep merge + forward + tt_action_post + D2H of the control
            import numpy as np
            from .agent_tick import AgentTick
            raw = synth.raw_camera_frames(seed=41, T=1)[0]                               # (4, 900, 1600, 3) uint8, host
            raw_pinned = torch.from_numpy(raw).pin_memory()
            rng = np.random.default_rng(2)
            half = np.concatenate([rng.uniform(-8, 30, (32768, 1)), rng.uniform(-19, 19, (32768, 1)),
                                   rng.uniform(-4.5, 0.5, (32768, 1)), rng.uniform(0, 1, (32768, 1))], 1).astype(np.float32)
            for cache, key_ in ((False, "raw_ms"), (True, "raw_prev_sweep_cache_ms")):
                at = AgentTick(self.model, lag=2, queue_len=3, use_cache=cache)
                pos = np.array([3.0, 1.0])
                tick = lambda: at.run_step(raw_pinned, half, pos, 0.2, 4.0, pos + np.array([5.0, 20.0]), 2)   # noqa: E731
                for _ in range(6):
                    tick()                       # fills the queue (and the BEV ring); run_step synchronises itself (D2H copy)
                out[key_] = timed(tick)
                del at
            out["raw_note"] = ("host uint8 frames (17.3 MB, pinned) + 32,768-point half sweep per tick -> control: H2D, preprocess, "
                               "merge (65,536 points), eager forward, tt_action_post, D2H")
        except Exception as e:
            out["raw_error"] = f"{type(e).__name__}: {e}"[:300]
        try:
            g = InferenceGraph(self.model, b1, channel_last_out=True)
            out["graph_ms"] = timed(g.replay)
            gc = InferenceGraph(self.model, b1, channel_last_out=True, prev_bev=key)
            out["graph_prev_sweep_cache_ms"] = timed(gc.replay)
        except Exception as e:
            out["graph_error"] = f"{type(e).__name__}: {e}"
        return out

    def cpu_baseline(self):
        """The oracle on ONE frame in a bounded subprocess (bench_cpu_baseline.py beside bench.py: one thread per PHYSICAL core
        of the host -- SURVEY 8(d) --, 2 warm-ups + 5 timed passes, per-stage medians, CPU model string, one 32-thread pass
        beside it; TT_BENCH_CPU_PASSES=n for fewer timed passes, TT_BENCH_CPU_THREADS=n to fix the thread count)."""
        import json
        import subprocess
        import sys
        threads = int(os.environ.get("TT_BENCH_CPU_THREADS", "0"))       # 0: all physical cores (counted in the subprocess)
        timed = os.environ.get("TT_BENCH_CPU_PASSES", "5")
        root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
        try:
            r = subprocess.run([sys.executable, os.path.join(root, "bench_cpu_baseline.py"), str(threads), timed, "2"], cwd=root,
                               capture_output=True, text=True, timeout=420)
            return json.loads(r.stdout.strip().splitlines()[-1])
        except Exception as e:   # timeout / failure: report, never block the bench
            return {"value": None, "unit": "frames/s", "cores": threads, "kind": "port",
                    "sample": f"oracle forward did not finish within 420 s ({type(e).__name__})"}
