"""Per-tick glue of the closed-loop agent (SURVEY 8f-1 LiDAR side, 8f-3): pose matrices against the golden produced by
the reference's own functions (f15), the half-sweep merge kernel against the oracle restatement, and the brake / throttle
arbitration + stuck detector against the oracle over a long scripted sequence."""
import os

import numpy as np
import pytest
import torch


def test_pose_matrices_match_reference_golden(golden_dir):
    from oracle import agent_ref as R
    from thinktwice_amd import agent_tick as A
    f = np.load(os.path.join(golden_dir, "f15_agent_transforms.npz"))
    for p, fwd, inv in zip(f["poses"], f["fwd"], f["inv"]):
        np.testing.assert_allclose(A.ego_pose_matrix(*p), fwd, rtol=0, atol=1e-12)
        np.testing.assert_allclose(A.inv_ego_pose_matrix(*p), inv, rtol=0, atol=1e-9)
        np.testing.assert_allclose(R.transform_matrix(*p), fwd, rtol=0, atol=1e-12)
        np.testing.assert_allclose(R.inv_transform_matrix(*p), inv, rtol=0, atol=1e-9)
        np.testing.assert_allclose(A.inv_ego_pose_matrix(*p) @ A.ego_pose_matrix(*p), np.eye(4), atol=1e-9)


def test_agent_controller_matches_oracle_sequence():
    from oracle import agent_ref as R
    from thinktwice_amd.agent_tick import AgentController
    rng = np.random.default_rng(3)
    mine, ref = AgentController(stuck_threshold=40), R.Arbitration(stuck_threshold=40)
    stuck_seen = False
    for t in range(400):
        speed = 0.0 if 100 <= t < 200 else float(rng.uniform(0, 6))        # a long standstill trips the stuck detector
        args = (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(0, 1) * (rng.random() < 0.6)), float(rng.uniform(0, 1)),
                float(rng.uniform(0, 0.75) * (rng.random() < 0.7)), float(rng.uniform(0, 1) * (rng.random() < 0.5)), speed)
        s, th, b, info = mine.step(*args)
        rs, rth, rb = ref.step(*args)
        assert (s, th, b) == (rs, rth, rb), (t, args)
        assert mine.stuck_detector == ref.stuck_detector
        stuck_seen |= info["is_stuck"]
        assert 0.0 <= th <= 0.6 and b in (0.0, 1.0)
    assert stuck_seen


@pytest.mark.gpu
def test_lidar_half_sweep_merge_matches_oracle():
    from oracle import agent_ref as R
    from thinktwice_amd.agent_tick import LidarSweepMerger
    rng = np.random.default_rng(9)
    mine, ref = LidarSweepMerger(), R.SweepMerge()
    pos, compass = np.array([10.0, -4.0]), 0.3
    for t in range(5):
        n = int(rng.integers(1000, 30000)) if t != 3 else 0                 # an empty half sweep too
        now = np.concatenate([rng.uniform(-40, 40, (n, 3)), rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
        got = mine.merge(now, pos, compass).cpu().numpy()
        want = ref.step(now.astype(np.float64), pos, compass)
        assert got.shape == want.shape
        if len(want):
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)
        pos = pos + rng.uniform(-1.5, 1.5, 2)
        compass += float(rng.uniform(-0.2, 0.2))
