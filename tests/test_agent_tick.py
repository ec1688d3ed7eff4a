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


def _tick_inputs(n_ticks, seed=5):
    """A scripted drive: raw frames from synth.raw_camera_frames (F17's generator; tick t uses sweep t % 3 of a 3-sweep set,
    rolled by 7 t pixels so that no two ticks see the same image), random half sweeps, a moving pose, a fixed route target."""
    from thinktwice_amd import synth
    rng = np.random.default_rng(seed)
    base = synth.raw_camera_frames(seed=23, T=3)
    pos, compass = np.array([12.0, -3.0]), 0.4
    for t in range(n_ticks):
        frames = np.ascontiguousarray(np.roll(base[t % 3], 7 * t, axis=2))
        n = int(rng.integers(9000, 14000))
        half = np.concatenate([rng.uniform(-8, 30, (n, 1)), rng.uniform(-19, 19, (n, 1)), rng.uniform(-4.5, 0.5, (n, 1)),
                               rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
        speed = float(rng.uniform(0.0, 6.0))
        yield frames, half, pos.copy(), compass, speed, pos + np.array([9.0, 14.0]), int(rng.integers(1, 7)) if t % 5 else -1
        pos = pos + rng.uniform(-0.4, 0.4, 2)
        compass += float(rng.uniform(-0.05, 0.05))


def _check_stage_outputs(info, chain):
    """The tick's two input stages against the oracle's, each at its own tolerance: the image pipeline at golden F17's bounds
    (2e-3 max / 2e-5 mean on the normalised pixels; tests/test_preprocess.py), the merged cloud at 2e-4 m.  The oracle's later
    stages then consume the device's stage outputs (AgentChain.run_step)."""
    d = (info["img"].float().cpu() - chain.last["img"]).abs()
    assert float(d.max()) < 2e-3 and float(d.mean()) < 2e-5, (float(d.max()), float(d.mean()))
    got, want = info["cloud"].cpu().numpy(), chain.last["cloud"]
    assert got.shape == want.shape
    if len(want):
        np.testing.assert_allclose(got, want, rtol=0, atol=2e-4)


class _OracleHeads:
    """oracle.agent_ref's process_action + control_pid + Arbitration (pinned by F9 / F15) as one stateful object."""

    def __init__(self, cfg, stuck_threshold=800):
        from oracle import agent_ref as R
        c = cfg["cfg"]
        self.R, self.c = R, c
        self.turn = R.PID(c["turn_KP"], c["turn_KI"], c["turn_KD"], c["turn_n"])
        self.speed = R.PID(c["speed_KP"], c["speed_KI"], c["speed_KD"], c["speed_n"])
        self.arb = R.Arbitration(stuck_threshold)

    def step(self, pred, speed, target):
        v = torch.tensor([speed], dtype=torch.float32)
        sc, thc, bc = self.R.process_action(pred["mu_branches"].float().cpu(), pred["sigma_branches"].float().cpu())
        _, tht, bt, _, _ = self.R.control_pid(self.c, self.turn, self.speed, pred["pred_wp"][:, -1].float().cpu(), v,
                                              np.asarray(target, dtype=np.float64))
        return self.arb.step(sc, thc, bc, tht, bt, float(v))


@pytest.mark.gpu
@pytest.mark.parametrize("use_cache", [False, True], ids=["two-sweep", "prev-sweep-cache"])
def test_agent_tick_end_to_end_matches_the_oracle_chain(use_cache):
    """VERDICT r4 missing #2: the whole model-side tick (thinktwice_agent.py:362-529) chained -- uint8 4 x 900 x 1600 frames ->
    tt_preprocess_images -> half-sweep merge -> queue / sweep selection -> forward_inference (bf16x3, with and without the
    previous-sweep BEV cache) -> tt_action_post -- stage by stage against the ORACLE chain (oracle.agent_ref.AgentChain: preprocess_ref (F17)
    -> SweepMerge (F15) -> model_ref (F7 / F8) -> process_action / control_pid (F9) -> Arbitration (F15)) over 7 ticks (3 live).
    Shortened queue (lag 2, 4 frames) and a 128 x 256 network input so that the CPU oracle finishes in seconds per tick;
    the full-size tick is below."""
    from oracle import agent_ref as R
    from thinktwice_amd import calib, model as tm, params
    from thinktwice_amd.agent_tick import AgentTick
    hw = (128, 256)
    m, cfg = tm.build_thinktwice(dtype="f32x3", final_dim=hw)
    sd = params.init_params(cfg, seed=3)
    m.load_state_dict(sd)
    tick = AgentTick(m, lag=2, queue_len=4, use_cache=use_cache, stuck_threshold=5, final_dim=hw)
    mx, my = calib.undistort_rectify_map(calib.IMG_W, calib.IMG_H)
    chain = R.AgentChain(sd, cfg, mx, my, hw, tick.img_metas, lag=2, queue_len=4, stuck_threshold=5)
    heads = _OracleHeads(cfg, stuck_threshold=5)
    live, agree = 0, 0
    for t, args in enumerate(_tick_inputs(7)):
        s, th, b, info = tick.run_step(*args)
        rs, rth, rb, rpred = chain.run_step(*args, device_img=info["img"].cpu(), device_cloud=info["cloud"].cpu().numpy())
        _check_stage_outputs(info, chain)
        if t < 4:
            assert (s, th, b) == (0.0, 0.0, 0.0) and rpred is None
            continue
        live += 1
        for k in ("pred_wp", "mu_branches", "sigma_branches"):
            got, want = info["pred"][k].float().cpu(), rpred[k]
            e = float((got - want).abs().max() / want.abs().max().clamp_min(1e-6))
            assert e < 1e-3, (t, k, e)
        # post-processing: the oracle's heads + PID + arbitration driven by the DEVICE's outputs must give the device's controls
        # (stateful over the ticks); the chain's own controls differ by what 1e-3 on the heads does to them
        os_, oth, ob = heads.step(info["pred"], args[4], info["target_point"])
        assert abs(s - os_) < 1e-5 and abs(th - oth) < 1e-5 and b == ob, (t, (s, th, b), (os_, oth, ob))
        agree += int(abs(s - rs) < 5e-3 and abs(th - rth) < 5e-3 and b == rb)
    assert live == 3 and agree >= 2, agree


@pytest.mark.gpu
def test_agent_tick_full_size_one_live_tick():
    """The tick at the thinktwice.py size (4 x 900 x 1600 -> 448 x 896, ~20 k merged points), the shortest queue that has a
    previous sweep (lag 1, 2 frames): waypoints and both control heads within 1e-3 of the oracle chain, controls equal."""
    from oracle import agent_ref as R
    from thinktwice_amd import calib, model as tm, params
    from thinktwice_amd.agent_tick import AgentTick
    m, cfg = tm.build_thinktwice(dtype="f32x3")
    sd = params.init_params(cfg, seed=0)
    m.load_state_dict(sd)
    tick = AgentTick(m, lag=1, queue_len=2)
    mx, my = calib.undistort_rectify_map(calib.IMG_W, calib.IMG_H)
    chain = R.AgentChain(sd, cfg, mx, my, (calib.FINAL_H, calib.FINAL_W), tick.img_metas, lag=1, queue_len=2)
    for t, args in enumerate(_tick_inputs(3, seed=8)):
        s, th, b, info = tick.run_step(*args)
        rs, rth, rb, rpred = chain.run_step(*args, device_img=info["img"].cpu(), device_cloud=info["cloud"].cpu().numpy())
        _check_stage_outputs(info, chain)
    for k in ("pred_wp", "mu_branches", "sigma_branches", "pred_speed"):
        got, want = info["pred"][k].float().cpu(), rpred[k]
        e = float((got - want).abs().max() / want.abs().max().clamp_min(1e-6))
        assert e < 1e-3, (k, e)
    os_, oth, ob = _OracleHeads(cfg).step(info["pred"], args[4], info["target_point"])
    assert abs(s - os_) < 1e-5 and abs(th - oth) < 1e-5 and b == ob
