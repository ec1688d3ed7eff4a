#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv.py tests/test_lidar.py -m gpu -x -q 2>&1 | tail -2
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
for it in 1 2; do
for sp in 1 0; do
  TT_GLDS_X3_SPREAD=$sp TT_BENCH_DUMP=gpurun_out/r3_spread_shapes_$sp.json timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']; print('spread $sp run $it', b['value'], 'frames/s', b['ms_per_step'], 'ms; conv', r['conv_ms_per_step'], 'frac', r['frac'], 'dominant', r['dominant_kernel']['avg_launch_ms'], r['dominant_kernel']['frac'])"
done
done
python - <<'PY'
import json
a = {r["shape"]: r for r in json.load(open("gpurun_out/r3_spread_shapes_1.json"))}
b = {r["shape"]: r for r in json.load(open("gpurun_out/r3_spread_shapes_0.json"))}
rows = sorted(((b[k]["ms"] - a[k]["ms"], k) for k in a if k in b), reverse=True)
for dms, k in rows[:10] + rows[-5:]:
    print(f"{k:52s} {b[k]['ms']:7.3f} -> {a[k]['ms']:7.3f}  ({-dms:+.3f})")
print("sum", round(-sum(d for d, _ in rows), 3))
PY
