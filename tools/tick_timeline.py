"""Timeline of the decoder part of the LAST tick in a rocprofv3 kernel trace (tools/tick_profile.py):
python tools/tick_timeline.py <dir with *kernel_trace.csv> [layers]   -- start offset, duration, queue, kernel."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
nlayers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rd = list(csv.DictReader(open(f)))
qk = "Queue_Id" if "Queue_Id" in rd[0] else None
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(qk, "?") if qk else "?") for r in rd))
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 20_000_000:
        cut = i
tick = rows[cut:]
t0 = tick[0][0]
first = next(i for i, r in enumerate(tick) if "dec_gru_kernel" in r[2] or "look_project_pack" in r[2])
print(f"tick span {(tick[-1][1] - t0) / 1e3:.1f} us; decoder starts at {(tick[first][0] - t0) / 1e3:.1f} us")
seen = 0
for s, e, n, q in tick[max(0, first - 3):]:
    if "dec_gru_kernel" in n:
        seen += 1
        if seen > nlayers:
            break
    short = n.replace("void ", "").replace("tt::", "").split("(")[0][:60]
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:7.1f} us  q{q:>3}  {short}")
print(f"... last kernel ends at {(tick[-1][1] - t0) / 1e3:.1f} us")
