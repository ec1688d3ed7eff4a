"""bench.py `cpu_baseline` leg, run as a separate bounded process:   python bench_cpu_baseline.py [threads] [timed] [warmup]

The CPU path beside the GPU number (SURVEY.md 8(d) protocol): the oracle -- oracle/model_ref.py, the torch-CPU f32 restatement
that reproduces the reference modules bit for bit (tests/golden f7/f8) -- on ONE frame of the bench workload (B = 1, 2 sweeps x
4 cams x 448x896 + 65,536 points), `warmup` untimed passes then `timed` passes, MEDIAN reported, per stage (camera encoder /
LiDAR encoder / fusion / decoder) and end to end, with the host CPU model string and the thread count.  Test infrastructure:
this file lives beside bench.py, not in the product package; nothing under thinktwice_amd/ imports oracle/."""
import json
import os
import statistics
import sys
import time


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores():
    """Distinct (socket, core) pairs this process may run on ("all physical cores"); SMT siblings count once."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    cores = set()
    for cpu in allowed:
        try:
            base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
            cores.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            cores.add(("?", str(cpu)))
    return max(1, len(cores))


def main():
    auto = not (len(sys.argv) > 1 and int(sys.argv[1]) > 0)
    threads = int(sys.argv[1]) if not auto else physical_cores()
    timed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    warm = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    import torch
    torch.set_num_threads(threads)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle import model_ref as M
    from thinktwice_amd import config, params, synth
    cfg = config.model_config()
    sd = params.init_params(cfg, seed=0)
    batch = synth.make_batch(1)

    def one():
        """oracle.model_ref.forward_inference, statement for statement (EDF:194-210), with a clock between the stages."""
        t = [time.perf_counter()]
        cam = M.lss_forward(sd, "img_encoder", cfg, batch["img"], batch["img_metas"])
        cam_bev = M.rot_flip(cam["bev"])
        t.append(time.perf_counter())
        meas = M.measurement_feat(sd, batch)
        lid = [M.rot_flip(x) for x in M.lidar_net(sd, "lidar_encoder", cfg, batch["points"][:, -1])]
        t.append(time.perf_counter())
        flat, bev32, _ = M.fusion(sd, cam_bev, lid[0])
        t.append(time.perf_counter())
        M.decoder_forward(sd, cfg, flat, bev32, meas, cam["lidar2img"], cam["ida_mat"], cam["fpn_feats"])
        t.append(time.perf_counter())
        return [t[i + 1] - t[i] for i in range(4)] + [t[-1] - t[0]]

    rows = []
    sweep = {}
    with torch.no_grad():
        if auto:
            # SURVEY 8(d) asks for all physical cores; torch's CPU kernels do not scale across two sockets on this model (128
            # threads on a 2 x 64-core EPYC 9575F: 44.7 s per frame against 8.4 s on 32, profiles/r05_bench_default.json), and a
            # baseline slower than it need be flatters the GPU.  So: one pass per thread count, smallest first, stopping when a
            # count is slower than the best so far; the medians are then taken at the fastest count, and the sweep is reported.
            phys = physical_cores()
            best = None
            for n in sorted({min(32, phys), min(64, phys), phys}):
                torch.set_num_threads(n)
                if best is None:
                    one()                                   # (cold start: weights, allocator)
                t = one()[4]
                sweep[str(n)] = round(t * 1e3, 1)
                if best is None or t < best[1]:
                    best = (n, t)
                elif t > 1.1 * best[1]:
                    break
            threads = best[0]
            torch.set_num_threads(threads)
            warm = max(0, warm - 1)
        for _ in range(warm):
            one()
        for _ in range(timed):
            rows.append(one())
    med = [statistics.median(r[i] for r in rows) for i in range(5)]
    names = ["encoder_cam", "lidar", "fusion", "decoder"]
    print(json.dumps({
        "value": round(1.0 / med[4], 4), "unit": "frames/s", "cores": threads, "kind": "port",
        "cpu": cpu_model(), "logical_cpus": os.cpu_count(), "physical_cores": physical_cores(), "thread_sweep_frame_ms": sweep or None,
        "stage_ms": {n: round(med[i] * 1e3, 1) for i, n in enumerate(names)},
        "frame_ms": round(med[4] * 1e3, 1), "frame_ms_min_max": [round(min(r[4] for r in rows) * 1e3, 1),
                                                                  round(max(r[4] for r in rows) * 1e3, 1)],
        "sample": f"1 frame (B=1) full forward_inference, oracle/model_ref.py, torch {torch.__version__} CPU f32, {threads} "
                  f"threads{' (the fastest of the thread counts swept up to all physical cores)' if auto else ''}: median of {timed} timed passes after "
                  f"warm-up, per stage and end to end"}))


if __name__ == "__main__":
    main()
