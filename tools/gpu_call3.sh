#!/bin/bash
# round 2, GPU call 3: composite decoder kernels (unit + end-to-end), x3 conv unit tests, x3 bench with the composite decoder
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_chain.py tests/test_decoder_fused.py -m gpu -q > gpurun_out/r2_pytest_c_units.log 2>&1; echo "units rc=$?"
grep -E "passed|failed|Error|assert|^E " gpurun_out/r2_pytest_c_units.log | cut -c1-300 | tail -25
timeout 900 python -m pytest tests/test_decoder.py -m gpu -q -s > gpurun_out/r2_pytest_c_dec.log 2>&1; echo "decoder rc=$?"
grep -E "passed|failed|Error|^E |^[0-9] \{" gpurun_out/r2_pytest_c_dec.log | cut -c1-700 | tail -20
timeout 600 python -m pytest tests/test_conv.py -m gpu -q -k "x3" > gpurun_out/r2_pytest_c_conv.log 2>&1; echo "conv rc=$?"
tail -3 gpurun_out/r2_pytest_c_conv.log
timeout 900 python -m pytest tests/test_forward.py -m gpu -q -s -k "bf16x3 or f32x3" > gpurun_out/r2_pytest_c_fwd.log 2>&1; echo "fwd rc=$?"
grep -E "passed|failed|rel errs|Error|^E " gpurun_out/r2_pytest_c_fwd.log | cut -c1-900 | tail -10
TT_BENCH_F32=0 TT_BENCH_DUMP=gpurun_out/r2_conv_shapes_x3_b.json timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_bench_x3_b.json 2> gpurun_out/r2_bench_x3_b.err; echo "bench x3 rc=$?"
tail -3 gpurun_out/r2_bench_x3_b.err
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r2_bench_x3_b.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'], d['roofline']['conv_ms_per_step'], d['roofline']['launches'])
    print(json.dumps(d.get('tick_latency')))
    print(json.dumps(d.get('bf16_speed_mode'))[:300])
except Exception as e:
    print('bench parse failed', e)
PY
