"""Per-kernel totals of the LAST tick in a rocprofv3 kernel trace (ticks are separated by >= 20 ms of idle time, see
tools/tick_profile.py):  python tools/last_tick_stats.py <dir with *kernel_trace.csv> [top]"""
import csv
import glob
import sys
from collections import defaultdict

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 20_000_000:
        cut = i
tick = rows[cut:]
agg = defaultdict(lambda: [0, 0])
for s, e, n in tick:
    agg[n][0] += 1
    agg[n][1] += e - s
busy = sum(v[1] for v in agg.values())
print(f"last tick: {len(tick)} launches, {busy / 1e6:.3f} ms of kernel time, span {(tick[-1][1] - tick[0][0]) / 1e6:.3f} ms")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t / 1e3:9.1f} us {c:5d} x {t / c / 1e3:8.1f}  {n[:110]}")
