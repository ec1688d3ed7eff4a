#!/bin/bash
# round 4: per-kernel totals of the last batch-1 tick (eager) with the wide chains, + the individual wide launches in order
ROOT="$GRAFT_REPO_ROOT"; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_chain.py -q -m gpu -x 2>&1 | tail -4 | tee $OUT/r04_chain_wide_b.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/r4t_eager
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/r4t_eager -o p -- python $ROOT/tools/tick_profile.py f32x3 5 > $OUT/r4t_eager.log 2>&1
cd $ROOT
( grep "^tick" $OUT/r4t_eager.log; python tools/last_tick_stats.py $OUT/r4t_eager 24 ) 2>&1 | cut -c1-150 | tee -a $OUT/r04_chain_wide_b.txt
python - <<'PY' 2>&1 | tee -a $OUT/r04_chain_wide_b.txt
import csv, glob
f = glob.glob("gpurun_out/r4t_eager/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", "")) for r in csv.DictReader(open(f))))
w = [r for r in rows if "chain" in r[2]][-40:]
prev = None
for s, e, n, gx, gy in w:
    print(f"{(e - s) / 1e3:7.1f} us  grid {gx}x{gy}  {n[:40]}")
PY
rm -rf $OUT/r4t_eager
