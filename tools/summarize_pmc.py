#!/usr/bin/env python
"""Summarise rocprofv3 counter_collection CSVs (one --pmc pass each) per kernel name.
usage: tools/summarize_pmc.py gpurun_out/prof_<tag>_fetch_voxel gpurun_out/prof_<tag>_write_voxel ..."""
import csv
import glob
import json
import os
import sys


def load(d):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "?")
            c = r.get("Counter_Name")
            v = float(r.get("Counter_Value", 0))
            a = out.setdefault((name, c), [0, 0.0])
            a[0] += 1
            a[1] += v
    return out


def stamp():
    """Identity of the build the counters were taken from: source fingerprint (+ git HEAD when the tree
    has one); bench.py refuses counter data whose fingerprint differs from the tree it runs from."""
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, root)
    from thinktwice_amd import build
    out = {"csrc_sha": build.source_fingerprint()}
    try:
        out["git_head"] = subprocess.check_output(["git", "-C", root, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL,
                                                  text=True).strip()
    except Exception:
        out["git_head"] = None
    return out


def main():
    res = {"_stamp": stamp()}
    for d in sys.argv[1:]:
        for (name, c), (n, tot) in load(d).items():
            short = name.split("(")[0].replace("void ", "").replace("tt::", "")[:110]    # keep the kernel name itself
            res.setdefault(short, {})[c] = {"dispatches": n, "sum": tot, "per_dispatch": tot / max(n, 1)}
    res["_stamp"]["kernels"] = sorted(k for k in res if k != "_stamp")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
