#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel stats CSV + steady-state step timeline.

usage: python tools/rocpd_stats.py <results.db> [out.csv] [--forwards N]
`--forwards` = number of forward passes the traced command ran (bench.py: warmup + steps + 1 roofline pass);
per-forward figures divide by it.
"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    out = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    nfwd = int(sys.argv[sys.argv.index("--forwards") + 1]) if "--forwards" in sys.argv else 1
    cur = db.cursor()
    rows = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
    agg = collections.OrderedDict()
    for n, s, e, _ in rows:
        a = agg.setdefault(n, [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += e - s; a[2] = min(a[2], e - s); a[3] = max(a[3], e - s)
    tot = sum(a[1] for a in agg.values())
    order = sorted(agg.items(), key=lambda kv: -kv[1][1])
    if out:
        with open(out, "w") as f:
            f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
            for n, a in order:
                f.write(f'"{n}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100 * a[1] / tot:.2f},{a[2]},{a[3]}\n')
    short = lambda n: re.sub(r"\(.*", "", n).replace("void ", "")[:100]
    print(f"total kernel time {tot / 1e6:.2f} ms over {nfwd} forwards = {tot / 1e6 / nfwd:.2f} ms/forward")
    for n, a in order[:30]:
        print(f"{short(n):100s} {a[0] / nfwd:8.1f}/fwd {a[1] / 1e6 / nfwd:8.3f} ms/fwd {a[1] / a[0] / 1e3:9.1f} us")
    # steady-state step: between the last two lidar_keys_kernel launches
    names = [short(r[0]) for r in rows]
    ks = [i for i, n in enumerate(names) if "lidar_keys_kernel" in n]
    if len(ks) >= 3:
        a, b = ks[-3], ks[-2]
        seg = rows[a:b]
        t0 = seg[0][1]
        span = (rows[b][1] - t0) / 1e6
        per_stream = collections.defaultdict(float)
        for r in seg:
            per_stream[r[3]] += (r[2] - r[1]) / 1e6
        iv = sorted((r[1], r[2]) for r in seg)
        busy, ce, idle = 0, iv[0][0], 0
        for s, e in iv:
            if s > ce:
                idle += s - ce
            if e > ce:
                busy += e - max(s, ce); ce = e
        print(f"steady-state step: span {span:.2f} ms, {len(seg)} kernels, union busy {busy / 1e6:.2f} ms, "
              f"idle {idle / 1e6:.2f} ms, per-stream busy {dict((k, round(v, 2)) for k, v in per_stream.items())}")


if __name__ == "__main__":
    main()
