"""Re-run the fixture generators (tests/golden/gen_golden.py, i.e. the REFERENCE's own modules) and compare what they produce now
with the committed .npz files, array by array, bit for bit:    python tests/golden/regen_check.py F3,F9,F12,F15,F17
Needs /root/reference (build container only).  Prints one JSON object; `mismatch` / `only_in_file` must be empty.
tests/test_goldens_regenerate.py runs the cheap fixtures on every CPU test run."""
import json, os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
import gen_golden as G
out = {}
def fake_save(name, **arrays):
    ref = np.load(os.path.join(G.HERE, name))
    bad = []
    for k, v in arrays.items():
        v = np.asarray(v)
        if k not in ref.files:
            bad.append(k + ": missing"); continue
        r = ref[k]
        if r.shape != v.shape or r.dtype != v.dtype:
            bad.append(f"{k}: shape/dtype {r.shape}{r.dtype} vs {v.shape}{v.dtype}"); continue
        if not np.array_equal(r, v, equal_nan=(r.dtype.kind in "fc")):
            if r.dtype.kind in "fiu" and r.size:
                a, b = r.astype(np.float64), v.astype(np.float64)
                d = float(np.max(np.abs(a - b)))
                rel = float(np.max(np.abs(a - b) / np.maximum(np.abs(a), 1e-30)))
                bad.append(f"{k}: max abs diff {d:.3e}, max rel diff {rel:.3e}, {int(np.sum(a != b))} of {a.size} entries")
            else:
                bad.append(f"{k}: differs")
    extra = [k for k in ref.files if k not in arrays]
    out[name] = {"arrays": len(arrays), "mismatch": bad, "only_in_file": extra}
G._save = fake_save
for n in sys.argv[1].split(","):
    t = time.time(); G.FIXTURES[n](); out[n + "_seconds"] = round(time.time() - t, 1)
print(json.dumps(out, indent=1))
