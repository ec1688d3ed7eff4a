"""CPU, build container only: the committed golden fixtures are exactly what the reference's own modules produce.
tests/golden/regen_check.py re-runs the generators (imports from /root/reference under the thin stubs of ref_stubs.py) in a
subprocess -- the stubs install stand-in mmcv / mmdet modules, which must not leak into this pytest process -- and compares every
array with the committed .npz bit for bit.  The cheap fixtures run here; the full-size ones (F8, F13b, F14: minutes of CPU
forward / backward) are checked by hand with the same script (tests/golden/README.md)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHEAP = "F3,F9,F12,F15,F17,F7,F10,F11"


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs /root/reference (build container only)")
def test_committed_goldens_are_what_the_reference_produces():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "regen_check.py"), CHEAP], cwd=ROOT,
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-2000:]
    report = json.loads(r.stdout[r.stdout.index("{\n"):])
    files = {k: v for k, v in report.items() if k.endswith(".npz")}
    assert len(files) == len(CHEAP.split(","))
    for name, rec in files.items():
        assert rec["arrays"] > 0 and rec["mismatch"] == [] and rec["only_in_file"] == [], (name, rec)
