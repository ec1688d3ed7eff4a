#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
TT_BENCH_F32=0 TT_BENCH_BF16=0 TT_BENCH_TICK=0 TT_BENCH_VOXEL=0 TT_BENCH_H2D=0 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['frac'], r['executed_mfma_frac']); print(json.dumps(r['dominant_kernel'])); print(r['traffic'], r['compulsory_bytes'], r['traffic_over_compulsory'])"
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
