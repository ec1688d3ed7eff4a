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
    threads = int(sys.argv[1]) if len(sys.argv) > 1 and int(sys.argv[1]) > 0 else physical_cores()
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
    with torch.no_grad():
        for _ in range(warm):
            one()
        for _ in range(timed):
            rows.append(one())
        # the round 1-4 protocol (<= 32 threads) beside it, one warm-up + one pass: says what the extra cores buy
        alt = None
        if threads > 32:
            torch.set_num_threads(32)
            one()
            alt = round(one()[4] * 1e3, 1)
            torch.set_num_threads(threads)
    med = [statistics.median(r[i] for r in rows) for i in range(5)]
    names = ["encoder_cam", "lidar", "fusion", "decoder"]
    print(json.dumps({
        "value": round(1.0 / med[4], 4), "unit": "frames/s", "cores": threads, "kind": "port",
        "cpu": cpu_model(), "logical_cpus": os.cpu_count(), "physical_cores": physical_cores(), "frame_ms_32_threads": alt,
        "stage_ms": {n: round(med[i] * 1e3, 1) for i, n in enumerate(names)},
        "frame_ms": round(med[4] * 1e3, 1), "frame_ms_min_max": [round(min(r[4] for r in rows) * 1e3, 1),
                                                                  round(max(r[4] for r in rows) * 1e3, 1)],
        "sample": f"1 frame (B=1) full forward_inference, oracle/model_ref.py, torch {torch.__version__} CPU f32, {threads} "
                  f"threads: median of {timed} timed passes after {warm} warm-ups, per stage and end to end"}))


if __name__ == "__main__":
    main()
