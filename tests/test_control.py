"""Closed-loop post-processing: the C ABI's tt_action_* entries (csrc/action_post.hip) through their binding
(thinktwice_amd/control.py) vs goldens recorded from the REFERENCE's own code: F9 = a 16-tick sequence of
`process_action` / `control_pid` (stateful PID windows included), F15 = 400 ticks of the agent's brake / throttle arbitration
+ stuck detector (statements lifted from thinktwice_agent.py:463-509 by tests/golden/gen_golden.py).

Tolerance 1e-6: the C source follows the reference's number types (f32 torch / numpy scalars, f64 PID); libm's atan2f vs
numpy's may differ in the last f32 bit."""
import ctypes
import os

import numpy as np
import pytest
import torch

from thinktwice_amd import config, control


def _f9(golden_dir):
    return np.load(os.path.join(golden_dir, "f9_control.npz"))


def test_host_stage_entries_match_reference_sequence(golden_dir):
    """process_action -> tt_action_ctrl_host, control_pid -> tt_action_pid_host (the reference's call structure)."""
    f = _f9(golden_dir)
    c = config.model_config()["cfg"]
    turn = control.PIDController(c["turn_KP"], c["turn_KI"], c["turn_KD"], c["turn_n"])
    spd = control.PIDController(c["speed_KP"], c["speed_KI"], c["speed_KD"], c["speed_n"])
    for t in range(f["mu"].shape[0]):
        pred = {"mu_branches": torch.from_numpy(f["mu"][t]).float(), "sigma_branches": torch.from_numpy(f["sigma"][t]).float()}
        speed = torch.from_numpy(f["speed"][t]).float()
        s1, th1, b1, meta = control.process_action(pred, 3, speed, f["target"][t])
        np.testing.assert_allclose([s1, th1, b1], f["pa"][t], rtol=0, atol=1e-6)
        s2, th2, b2, m2 = control.control_pid(c, turn, spd, torch.from_numpy(f["wp"][t]).float(), speed,
                                              f["target"][t].astype(np.float32).copy())
        got = [s2, th2, float(b2), m2["desired_speed"], m2["angle"], m2["angle_last"], m2["angle_target"],
               m2["angle_final"], m2["delta"]]
        np.testing.assert_allclose(got, f["pid"][t], rtol=0, atol=1e-6)
    assert set(meta) == {"speed", "steer", "throttle", "brake", "command", "target_point"}
    assert {"wp_1", "wp_4", "aim", "target", "desired_speed", "angle_final", "delta"} <= set(m2)


def test_oracle_action_heads_match_reference_golden_f9(golden_dir):
    """oracle/agent_ref.py's restatement of `process_action` / `control_pid` / `PIDController` (the checker of the end-to-end
    tick test) against the reference's own methods: bit for bit on the 16 stateful ticks of F9."""
    from oracle import agent_ref as R
    f = _f9(golden_dir)
    c = config.model_config()["cfg"]
    turn = R.PID(c["turn_KP"], c["turn_KI"], c["turn_KD"], c["turn_n"])
    spd = R.PID(c["speed_KP"], c["speed_KI"], c["speed_KD"], c["speed_n"])
    for t in range(f["mu"].shape[0]):
        pa = R.process_action(torch.from_numpy(f["mu"][t]).float(), torch.from_numpy(f["sigma"][t]).float())
        np.testing.assert_array_equal(np.array(pa), f["pa"][t])
        s, th, b, ds, af = R.control_pid(c, turn, spd, torch.from_numpy(f["wp"][t]).float(), torch.from_numpy(f["speed"][t]).float(),
                                         f["target"][t].astype(np.float32))
        np.testing.assert_array_equal(np.array([s, th, b, ds, af]), f["pid"][t][[0, 1, 2, 3, 7]])


def test_arbitration_matches_reference_statements_f15(golden_dir):
    f = np.load(os.path.join(golden_dir, "f15_agent_transforms.npz"))
    from thinktwice_amd.agent_tick import AgentController
    ctl = AgentController(stuck_threshold=int(f["arb_stuck_threshold"][0]))
    stuck = False
    for a, want in zip(f["arb_in"], f["arb_out"]):
        s, th, b, info = ctl.step(*[float(v) for v in a])
        np.testing.assert_allclose([s, th, b, ctl.stuck_detector], want, rtol=0, atol=1e-12)
        stuck |= info["is_stuck"]
    assert stuck


def _fused_expectation(golden_dir):
    """What the one-call entry must return on the F9 sequence: F9's two heads pushed through the arbitration stage
    (itself pinned by F15 above)."""
    f = _f9(golden_dir)
    from thinktwice_amd.agent_tick import AgentController
    ctl = AgentController()
    return f, [ctl.step(f["pa"][t][0], f["pa"][t][1], f["pa"][t][2], f["pid"][t][1], f["pid"][t][2], float(f["speed"][t][0]))[:3]
               for t in range(f["mu"].shape[0])]


def _check_fused(post, f, want, tick):
    for t in range(f["mu"].shape[0]):
        s, th, b, info = tick(post, t)
        np.testing.assert_allclose([s, th, b], want[t], rtol=0, atol=1e-6)
        np.testing.assert_allclose([info["steer_ctrl"], info["throttle_ctrl"], info["brake_ctrl"]], f["pa"][t], rtol=0, atol=1e-6)
        np.testing.assert_allclose([info["steer_traj"], info["throttle_traj"], info["brake_traj"], info["desired_speed"],
                                    info["angle"], info["angle_last"], info["angle_target"], info["angle_final"],
                                    info["delta"]], f["pid"][t], rtol=0, atol=1e-6)


def test_one_call_host_entry_matches_reference_sequence(golden_dir):
    f, want = _fused_expectation(golden_dir)
    post = control.ActionPost(config.model_config()["cfg"])
    _check_fused(post, f, want, lambda p, t: p.tick_host(f["mu"][t][0, -1], f["sigma"][t][0, -1], f["wp"][t][0],
                                                         float(f["speed"][t][0]), f["target"][t]))


def test_action_entry_rejects_bad_window_length():
    c = dict(config.model_config()["cfg"])
    c["turn_n"] = 65
    post = control.ActionPost(c)
    with pytest.raises(Exception, match="window length"):
        post.tick_host([1, 1], [1, 1], np.zeros(8), 1.0, [0, 10])


@pytest.mark.gpu
def test_one_call_device_entry_matches_reference_sequence(golden_dir):
    """tt_action_post on the model's output tensors in device memory (the full (1, L, ...) tensors: the entry reads the
    last refinement stage's slice), state in device memory, one D2H copy per tick."""
    f, want = _fused_expectation(golden_dir)
    post = control.ActionPost(config.model_config()["cfg"], device="cuda")

    def tick(p, t):
        pred = {"mu_branches": torch.from_numpy(f["mu"][t]).float().cuda(), "sigma_branches": torch.from_numpy(f["sigma"][t]).float().cuda(),
                "pred_wp": torch.from_numpy(np.concatenate([np.zeros((1, 5, 4, 2)), f["wp"][t][:, None]], 1)).float().cuda()}
        return p.tick(pred, float(f["speed"][t][0]), f["target"][t])
    _check_fused(post, f, want, tick)
    # device and host entries are the same source: a long random sequence must agree, state included
    host = control.ActionPost(config.model_config()["cfg"], stuck_threshold=30)
    dev = control.ActionPost(config.model_config()["cfg"], stuck_threshold=30, device="cuda")
    rng = np.random.default_rng(5)
    for t in range(300):
        mu, sg = rng.uniform(0.2, 3, (1, 6, 2)), rng.uniform(0.2, 3, (1, 6, 2))
        wp = np.cumsum(rng.uniform(0, 1, (1, 6, 4, 2)) * [0.6, 1.5] + [-0.3, 0.1], 2)
        speed = 0.0 if 60 <= t < 120 else float(rng.uniform(0, 7))
        target = rng.normal(size=2) * [3, 10] + [0, 12]
        pred = {k: torch.from_numpy(v).float().cuda() for k, v in (("mu_branches", mu), ("sigma_branches", sg), ("pred_wp", wp))}
        a = dev.tick(pred, speed, target)
        b = host.tick_host(np.float32(mu[0, -1]), np.float32(sg[0, -1]), np.float32(wp[0, -1]), speed, target)
        np.testing.assert_allclose(a[:3], b[:3], rtol=0, atol=1e-6)
        assert a[3]["stuck_detector"] == b[3]["stuck_detector"] and a[3]["is_stuck"] == b[3]["is_stuck"]
    st = (ctypes.c_char * ctypes.sizeof(control.ActionState)).from_buffer_copy(bytes(dev.state.cpu().numpy()))
    dstate = control.ActionState.from_buffer_copy(st)
    np.testing.assert_allclose(list(dstate.turn_window), list(host.state.turn_window), atol=1e-6)
    assert dstate.turn_head == host.state.turn_head and dstate.stuck_detector == host.state.stuck_detector
