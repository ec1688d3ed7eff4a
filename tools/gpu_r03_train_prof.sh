#!/bin/bash
# rocprofv3 kernel stats of one training iteration at batch 8 (train-mode semantics), plus the bench line of the same workload
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python bench.py --workload train_step --steps 3 --warmup 1 > gpurun_out/r3_train_step_b8_bench.json 2> gpurun_out/r3_train_step_b8_bench.err
tail -2 gpurun_out/r3_train_step_b8_bench.err; cut -c1-700 gpurun_out/r3_train_step_b8_bench.json
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python $GRAFT_REPO_ROOT/bench.py --workload train_step --steps 2 --warmup 1 > /tmp/train_prof.log 2>&1
tail -3 /tmp/train_prof.log | cut -c1-300
f=$(find /tmp/prof_train -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r3_train_step_b8_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$f")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print("kernel time total ms (4 iterations incl. profiled one + setup)", tot / 1e6, "launches", calls)
for r in rows[:40]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):7d} calls avg {float(r["AverageNs"])/1e3:9.1f} us {float(r["Percentage"]):5.2f}% {r["Name"][:120]}')
PY
