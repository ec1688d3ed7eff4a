#!/bin/bash
# round 5, call A: the new tests (wide-chain faults, counting-sort voxel pool, agent tick, optimizer, refactored LSS neck), the
# voxel-pool op with and without the per-launch sort, and the short-K A/B (tile widths, phase stagger)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; O=gpurun_out/r05_a
rm -f $O.*
timeout 1500 python -m pytest tests/test_chain.py tests/test_voxel_pool.py tests/test_agent_tick.py tests/test_control.py tests/test_optim.py \
    tests/test_lss.py tests/test_decoder.py tests/test_plan.py tests/test_decoder_fused.py \
    "tests/test_train_step.py::test_non_finite_gradient_norm_skips_the_update_on_the_device" \
    "tests/test_forward.py::test_forward_small_matches_reference_golden_and_oracle" \
    "tests/test_forward.py::test_inference_graph_replay_matches_eager_and_golden" \
    -q -m gpu --maxfail=12 --durations=6 2>&1 | tail -60 > $O.pytest.txt
cut -c1-400 $O.pytest.txt | tail -40
for sort in 1 0; do
for i in 1 2; do
TT_VP_SORT=$sort timeout 300 python bench.py --workload voxel_pool --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('TT_VP_SORT=$sort generic', r['avg_launch_ms'], 'ms frac', r['frac'], '| planned', r['static_geometry_plan']['avg_launch_ms'], 'ms frac', r['static_geometry_plan']['frac'])
" | tee -a $O.vp.txt
done; done
timeout 900 python tools/shortk_ab.py base t64 t128 stg20 stg40 stg80 2>&1 | tee $O.shortk.txt | cut -c1-200
