#!/bin/bash
# rocprofv3 kernel trace + stats of the default forward bench (legs off), and the train_step leg
cd /root/repo
mkdir -p gpurun_out
export TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_H2D=0 TT_BENCH_VOXEL=0 TT_BENCH_TRAIN=0
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r3_prof_fwd -o p --output-format csv -- python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/r3_prof_fwd.log 2>&1
cd /root/repo
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3_prof_fwd/**/*kernel_stats.csv", recursive=True)
print(f)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms (4 forwards + setup)", tot / 1e6)
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.3f} ms {int(r["Calls"]):6d} calls  avg {float(r["AverageNs"])/1e3:9.1f} us  {float(r["Percentage"]):5.2f}%  {r["Name"][:110]}')
PY
cp $(ls gpurun_out/r3_prof_fwd/*/*kernel_stats.csv | head -1) gpurun_out/r3_forward_bf16x3_kernel_stats.csv
