#!/bin/bash
# round 3, call C: where does the run-staged sparse conv spend its time?  DMA / MFMA ablations + L2 hit counters
cd /root/repo
mkdir -p gpurun_out
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0
for dbg in 0 3 4 7 8 16 24; do
  TT_SP_DEBUG=$dbg TT_BENCH_DUMP=gpurun_out/r3e_shapes_dbg$dbg.json timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r3e_bench_dbg$dbg.json 2> gpurun_out/r3e_bench_dbg$dbg.err
done
python - <<'PY'
import json
for k in (0, 3, 4, 7, 8, 16, 24):
    try:
        rows = json.load(open(f"gpurun_out/r3e_shapes_dbg{k}.json"))
        print("dbg", k, [(r["shape"].split(" of")[0].replace("sparse ", "") + " " + r["shape"].split("N=")[1].split(" pairs")[0], r["ms"]) for r in rows if "sparse" in r["shape"] and r["calls"] == 4])
    except Exception as e:
        print("dbg", k, "failed", e)
PY
