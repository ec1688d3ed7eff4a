#!/usr/bin/env python
"""bench.py -- ThinkTwice per-frame forward path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload forward|voxel_pool|train_step] [--batch B]

One "step" = one pass of the hot path over one batch of B synthetic frames per GPU
(4 cams x 2 sweeps x 448x896 + LiDAR).  Prints ONE JSON line (rank 0).  For N>1 the driver
launches one process per GPU through torch.distributed.run; frames shard data-parallel with
no data-path collective (inference: replicas only), `value` = all ranks' frames / max time.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3       # f32-in MFMA
MFMA_BF16_PEAK_TF = 2500.0     # dense bf16


# --------------------------------------------------------------------------- workloads
class VoxelPoolWorkload:
    """Op-boundary voxel pooling at the thinktwice.py size:
    per (sample, sweep) 501,760 points x 256 ch -> 21x21 BEV; B samples x 2 sweeps per step."""
    name = "voxel_pool_op_boundary"
    dtype = "f32"
    metric = "frames/sec through the voxel_pooling op alone (2 sweeps per frame; NOT the headline forward metric)"

    def __init__(self, batch, device):
        from thinktwice_amd import camera, ops, synth
        self.B = batch
        fr = camera.make_frustum((448, 896), 16, (1.0, 41.0, 0.5))
        grid = camera.VoxelGrid([-8.0, 30.4, 1.8285], [-19.2, 19.2, 1.8285], [-4, 10, 14])
        mats = camera.geometry_matrices(camera.stack_img_metas(synth.make_img_metas(batch))).to(device)
        self.voxel_num = grid.voxel_num
        self.geom = ops.frustum_voxel_index(fr.to(device), mats, grid.lower, grid.voxel_size.tolist(),
                                            batch, 4)
        self.Np = self.geom.shape[1]
        g = torch.Generator(device=device).manual_seed(1234)
        self.feats = [torch.randn(batch, self.Np, 256, device=device, generator=g) for _ in range(2)]
        self.kernel_ms = []
        self.planned_ms = []
        from thinktwice_amd.voxel_pooling import VoxelPoolPlan
        t0 = time.perf_counter()
        self.plan = VoxelPoolPlan(self.geom, self.voxel_num)     # static geometry: built once per calibration
        torch.cuda.synchronize()
        self.plan_build_ms = (time.perf_counter() - t0) * 1e3

        # algorithmic bytes per launch: geom + feats + out, per sample
        self.alg_bytes_per_launch = batch * (self.Np * 3 * 4 + self.Np * 256 * 4 + 441 * 256 * 4)
        # compulsory bytes: the rows of out-of-range points (74 % with the CARLA rig) need never be fetched, so the
        # HBM roofline of the op is geom (all points) + in-range feature rows + out
        g = self.geom
        inr = ((g[..., 0] >= 0) & (g[..., 0] < 21) & (g[..., 1] >= 0) & (g[..., 1] < 21) & (g[..., 2] >= 0) & (g[..., 2] < 1))
        self.in_range_rows = int(inr.sum().item())
        self.compulsory_bytes_per_launch = batch * (self.Np * 3 * 4 + 441 * 256 * 4) + self.in_range_rows * 256 * 4

    def step(self):
        from thinktwice_amd.voxel_pooling import voxel_pooling_forward_wrapper
        outs = []
        for sweep in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            out = torch.zeros(self.B, 21, 21, 256, device=self.feats[0].device)  # pre-zero (caller duty)
            e0.record()   # torch's current stream == the stream the kernel is launched on
            voxel_pooling_forward_wrapper(self.B, self.Np, 256, 21, 21, 1, self.geom, self.feats[sweep],
                                          out, None)
            e1.record()
            self.kernel_ms.append((e0, e1))
            outs.append(out.permute(0, 3, 1, 2))
            # the same op through the static-geometry plan (reported beside it, never part of `value`)
            out2 = torch.zeros(self.B, 21, 21, 256, device=self.feats[0].device)
            p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            p0.record()
            self.plan.forward_into(self.feats[sweep], out2)
            p1.record()
            self.planned_ms.append((p0, p1))
        return outs

    def frames_per_step(self):
        return self.B

    def on_warm(self):
        torch.cuda.synchronize()
        self.kernel_ms.clear()
        self.planned_ms.clear()

    def roofline(self):
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in self.kernel_ms)
        avg = ms[len(ms) // 2]                    # median launch (no best-half filter)
        ach = self.alg_bytes_per_launch / (avg * 1e-3) / 1e9
        comp = self.compulsory_bytes_per_launch / (avg * 1e-3) / 1e9
        pms = sorted(a.elapsed_time(b) for a, b in self.planned_ms)
        pavg = pms[len(pms) // 2]
        pcomp = (self.compulsory_bytes_per_launch - self.B * self.Np * 12 + self.in_range_rows * 4) / (pavg * 1e-3) / 1e9
        planned = {"kernel": "vp_planned_segments_kernel + vp_planned_cells_kernel", "avg_launch_ms": round(pavg, 4),
                   "achieved": round(pcomp, 1), "frac": round(pcomp / HBM_PEAK_GBS, 4), "unit": "GB/s",
                   "plan_build_ms": round(self.plan_build_ms, 3),
                   "note": "static camera rig: geom_xyz is identical every frame, so the (sample, cell) sort is built once "
                           "(tt_voxel_pool_plan_build, excluded) and a forward streams in-range rows + 4 B index per row + out"}
        sort = os.environ.get("TT_VP_SORT", "1") != "0"
        kern = ("vp_cs_count + vp_cs_scan + vp_cs_scatter (per-launch counting sort) + vp_planned_segments_kernel + "
                "vp_planned_cells_kernel") if sort else "voxel_pool_p1_kernel + voxel_pool_p2_kernel"
        return {"kernel": kern, "bound": "hbm", "static_geometry_plan": planned,
                "achieved": round(comp, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(comp / HBM_PEAK_GBS, 4),
                "traffic": self._pmc_traffic(), "avg_launch_ms": round(avg, 4),
                "compulsory_bytes_per_launch": self.compulsory_bytes_per_launch,
                "in_range_rows": self.in_range_rows, "rows": self.B * self.Np,
                "op_boundary_algorithmic": {"bytes_per_launch": self.alg_bytes_per_launch, "gbs": round(ach, 1),
                                            "frac": round(ach / HBM_PEAK_GBS, 4)},
                "note": "achieved/frac count COMPULSORY bytes (geom of every point + feature rows of in-range points "
                        "+ out); op_boundary_algorithmic is SURVEY 8(d)'s geom+feats+out figure, which exceeds what "
                        "HBM delivers because out-of-range rows are never fetched"}

    def _pmc_traffic(self):
        """HBM bytes per launch of the op's kernels from the committed rocprofv3 counter passes of this workload
        (profiles/r06_voxel_pool_pmc.json: FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KB), if they were taken from this build."""
        import json
        try:
            from thinktwice_amd import build
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06_voxel_pool_pmc.json")
            data = json.load(open(path))
            if data.get("_stamp", {}).get("csrc_sha") != build.source_fingerprint() or self.B != 8:
                return None
            kb = 0.0                        # one dispatch of each of the five kernels = one launch of the operator
            seen = 0
            for name, c in data.items():
                if name != "_stamp" and ("vp_cs_" in name or "vp_planned" in name):
                    kb += 2.0 * c.get("FETCH_SIZE", {}).get("per_dispatch", 0.0) + c.get("WRITE_SIZE", {}).get("per_dispatch", 0.0)
                    seen += 1
            return int(kb * 1024) if seen == 5 else None
        except Exception:
            return None

    def cpu_baseline(self):
        """oracle C restatement, 1 thread, one (sample, sweep) repeated for ~10 s."""
        from oracle import c_ref
        geom = self.geom[:1].cpu().numpy()
        feats = self.feats[0][:1].cpu().numpy()
        t0 = time.time()
        n = 0
        while time.time() - t0 < 10.0:
            c_ref.voxel_pool_fwd(geom, feats, self.voxel_num.tolist(), acc64=False, want_pos_memo=True)
            n += 1
        dt = time.time() - t0
        # a frame = 2 sweeps
        return {"value": round(n / 2 / dt, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": f"{n} x (1 sample, 1 sweep: 501,760 pts x 256 ch) voxel-pool only, oracle/voxel_pool_ref.c"}


def make_workload(name, batch, device):
    if name == "voxel_pool":
        return VoxelPoolWorkload(batch, device)
    if name == "forward":
        from thinktwice_amd import bench_forward
        return bench_forward.ForwardWorkload(batch, device)
    if name == "train_step":
        from thinktwice_amd import bench_train
        return bench_train.TrainStepWorkload(batch, device)
    raise SystemExit(f"unknown workload {name}")


# --------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frames per GPU per step")
    ap.add_argument("--workload", default=os.environ.get("TT_BENCH_WORKLOAD", "auto"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # invoked plainly with --gpus N: become the launcher -- one rank per GPU through torch.distributed.run on
        # 127.0.0.1 (the driver's own multi-GPU invocation sets WORLD_SIZE and lands in the branch below)
        import socket
        import subprocess
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU product path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # (a rank that dies inside a collective must not leave the others waiting for the default 10 minutes)
        dist.init_process_group(backend="nccl", device_id=device, timeout=datetime.timedelta(minutes=4))

    name = args.workload
    if name == "auto":
        name = "forward" if os.path.exists(os.path.join(ROOT, "thinktwice_amd", "bench_forward.py")) \
            else "voxel_pool"
    wl = make_workload(name, args.batch, device)

    from thinktwice_amd.bench_harness import run_timed
    res = run_timed(wl, args.steps, args.warmup, dist=dist, sync=torch.cuda.synchronize, device=device)
    dt = res["seconds"]
    if hasattr(wl, "collect"):      # report data that takes collectives to produce: on every rank
        wl.collect()

    if rank == 0:
        frames = wl.frames_per_step() * args.steps * world
        line = {
            "metric": getattr(wl, "metric", "frames/sec forward (4-cam+LiDAR, thinktwice.py cfg)"),
            "value": round(frames / dt, 3),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl.dtype,
            "data": "synthetic (seeded N(0,1) images / uniform LiDAR / CARLA calibration constants), random-init weights",
            "config": {"workload": wl.name, "precision": getattr(wl, "precision_note", wl.dtype),
                       "frames_per_gpu_per_step": args.batch,
                       "global_frames_per_step": args.batch * world,
                       "parallelism": f"replicas x{world} (no data-path collective)",
                       "launch": getattr(wl, "launch_note", "eager launches"),
                       "batches_in_flight": getattr(wl, "pipeline", 1)},
            "roofline": wl.roofline(),
        }
        extra = getattr(wl, "extra", None)
        if extra:
            line.update(extra())
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = wl.cpu_baseline()
        if name == "train_step":
            line["unit"] = "samples/s"
            line["config"]["parallelism"] = f"data parallel x{world}: one all-reduce of the flat gradient buffer per iteration"
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
