#!/bin/bash
# round 6, call AB: launches of ONE steady-state forward that are not this library's kernels (torch / rocclr / rocprim), batch 1 and 8
ROOT="$GRAFT_REPO_ROOT"; cd /tmp && export TMPDIR=/tmp; O=$ROOT/gpurun_out/r06_ab.txt; mkdir -p $ROOT/gpurun_out; rm -f $O
for B in 1 8; do
  rm -rf $ROOT/gpurun_out/r6nt
  TT_TICK_BATCH=$B timeout 600 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r6nt -o p -- python $ROOT/tools/tick_profile.py f32x3h 4 $B > /dev/null 2>&1
  python - "$ROOT/gpurun_out/r6nt" $B <<'PY' | tee -a $O
import csv, glob, sys
from collections import Counter
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 20_000_000:
        cut = i
tick = rows[cut:]
other = [r for r in tick if "tt::" not in r[2]]
print(f"batch {sys.argv[2]}: last forward {len(tick)} launches, {len(other)} not tt:: ({sum(e - s for s, e, _ in other) / 1e3:.1f} us of {sum(e - s for s, e, _ in tick) / 1e3:.1f} us of kernel time)")
for n, c in Counter(r[2].split("(")[0][:90] for r in other).most_common(12):
    print(f"   {c:4d} x {n}")
PY
done
rm -rf $ROOT/gpurun_out/r6nt
