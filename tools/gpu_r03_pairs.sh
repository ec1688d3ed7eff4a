#!/bin/bash
# round 3: pair-format activations between bf16x3 convs (ResNet-50 bottlenecks) -- parity + A/B
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3_pairs_pytest_conv.txt
cat gpurun_out/r3_pairs_pytest_conv.txt
timeout 1200 python -m pytest tests/test_forward.py tests/test_lss.py tests/test_plan.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r3_pairs_pytest_fwd.txt
cat gpurun_out/r3_pairs_pytest_fwd.txt
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
for pr in 1 0; do
  TT_X3_PAIRS=$pr TT_BENCH_DUMP=gpurun_out/r3_pairs_shapes_$pr.json timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3_pairs_bench_$pr.json 2> gpurun_out/r3_pairs_bench_$pr.err
  tail -2 gpurun_out/r3_pairs_bench_$pr.err
  python - <<PY
import json
b = json.loads(open("gpurun_out/r3_pairs_bench_$pr.json").read().strip().splitlines()[-1])
d = json.load(open("gpurun_out/r3_pairs_shapes_$pr.json"))
print("pairs $pr", b["value"], "frames/s", b["ms_per_step"], "ms; conv total", round(sum(r["ms"] for r in d), 2), "roofline", b["roofline"]["frac"], b["roofline"].get("kernel_ms"))
PY
done
python - <<'PY'
import json
a = {r["shape"]: r for r in json.load(open("gpurun_out/r3_pairs_shapes_1.json"))}
b = {r["shape"]: r for r in json.load(open("gpurun_out/r3_pairs_shapes_0.json"))}
rows = sorted(((b[k]["ms"] - a[k]["ms"], k) for k in a if k in b), reverse=True)
for dms, k in rows[:14]:
    print(f"{k:46s} {b[k]['ms']:7.3f} -> {a[k]['ms']:7.3f}  ({dms:+.3f})")
print("sum of differences", round(sum(d for d, _ in rows), 3))
PY
timeout 1500 python -m pytest tests/test_train_step.py -m gpu -x -q -k "f13" 2>&1 | tail -6 > gpurun_out/r3_pairs_pytest_f13.txt
cat gpurun_out/r3_pairs_pytest_f13.txt
