"""bench.py `cpu_baseline` leg, run as a separate bounded process:
    python -m thinktwice_amd.cpu_baseline [threads]
Times the oracle (torch-CPU restatement, bit-exact vs the reference modules) on ONE frame of the
bench workload and prints one JSON object.  This is the reported CPU baseline, never the product."""
import json
import os
import sys
import time


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else min(os.cpu_count() or 1, 32)
    os.environ.setdefault("OMP_NUM_THREADS", str(threads))
    import torch
    torch.set_num_threads(threads)
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    from oracle import model_ref as M
    from thinktwice_amd import config, params, synth
    cfg = config.model_config()
    sd = params.init_params(cfg, seed=0)
    batch = synth.make_batch(1)
    t0 = time.time()
    with torch.no_grad():
        M.forward_inference(sd, cfg, batch)
    dt = time.time() - t0
    print(json.dumps({"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
                      "sample": f"1 frame (B=1) full forward_inference, oracle/model_ref.py, torch {torch.__version__} "
                                f"CPU f32, {threads} threads, {dt:.1f} s"}))


if __name__ == "__main__":
    main()
