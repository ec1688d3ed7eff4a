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
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d /root/repo/gpurun_out/r3e_pmc_l2 -o l2 --output-format csv -- python /root/repo/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/r3e_pmc_l2.log 2>&1
cd /root/repo
python - <<'PY'
import csv, glob, collections
files = glob.glob("gpurun_out/r3e_pmc_l2/**/*counter_collection.csv", recursive=True)
print(files)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda kv: -(kv[1].get("TCC_HIT_sum", 0) + kv[1].get("TCC_MISS_sum", 0)))[:14]:
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    print(f"{k:72s} n={cnt[k]//2:4d} hit={h:.3e} miss={m:.3e} rate={h/max(1,h+m):.3f}")
PY
