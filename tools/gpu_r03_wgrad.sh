#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_conv_bwd.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python bench.py --workload train_step --steps 3 --warmup 1 2> /dev/null | python -c "
import json,sys
b=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=b['roofline']; print(b['value'],'samples/s', b['ms_per_step'],'ms; wgrad', r['wgrad'], 'top', json.dumps(b.get('wgrad_top_shapes'))[:900])"
