"""Action post-processing of the closed-loop agent: a BINDING of the C ABI's `tt_action_*` entries
(csrc/action_post.hip, declared in include/thinktwice_hip.h).  The arithmetic -- Beta-mode control branch
(encoder_decoder_framework.py:268-304), waypoint PID with its two error windows (:309-390, utils.py:7-29) and the agent's
brake / throttle arbitration + stuck detector (leaderboard/team_code/thinktwice_agent.py:463-509) -- lives in that one C
source, compiled for the device and for the host.

* `ActionPost.tick(pred, speed, target)`: ONE kernel launch on the model's output tensors where they are, then one 192-byte
  device->host copy -- the whole post-processing of a tick (the reference copies mu / sigma / waypoints to the host and runs
  numpy on them).
* `process_action`, `control_pid`, `PIDController`: the reference's call structure (AGENT:458-461 calls them one by one) on
  the host entries, for callers that keep it.
"""
import ctypes

import numpy as np
import torch

from ._lib import check, cur_stream, lib, raise_on_device_fault

PID_WINDOW_MAX = 64
(ACT_STEER, ACT_THROTTLE, ACT_BRAKE, ACT_STEER_CTRL, ACT_THROTTLE_CTRL, ACT_BRAKE_CTRL, ACT_STEER_TRAJ, ACT_THROTTLE_TRAJ,
 ACT_BRAKE_TRAJ, ACT_DESIRED_SPEED, ACT_ANGLE, ACT_ANGLE_LAST, ACT_ANGLE_TARGET, ACT_ANGLE_FINAL, ACT_DELTA, ACT_AIM_X,
 ACT_AIM_Y, ACT_IS_TURN, ACT_IS_STUCK, ACT_STUCK_DETECTOR) = range(20)
ACTION_OUT = 24


class ActionCfg(ctypes.Structure):          # tt_action_cfg
    _fields_ = [(n, ctypes.c_double) for n in ("turn_KP", "turn_KI", "turn_KD", "speed_KP", "speed_KI", "speed_KD",
                                               "brake_speed", "brake_ratio", "clip_delta", "aim_dist", "angle_thresh",
                                               "dist_thresh")] + \
               [(n, ctypes.c_int) for n in ("turn_n", "speed_n", "stuck_threshold", "reserved")]


class ActionState(ctypes.Structure):        # tt_action_state
    _fields_ = [("turn_window", ctypes.c_double * PID_WINDOW_MAX), ("speed_window", ctypes.c_double * PID_WINDOW_MAX),
                ("turn_head", ctypes.c_int), ("speed_head", ctypes.c_int), ("stuck_detector", ctypes.c_int),
                ("reserved", ctypes.c_int)]


def make_cfg(cfg, stuck_threshold=800):
    """tt_action_cfg from the reference's `cfg` dict (configs/thinktwice.py:44-57)."""
    c = ActionCfg()
    for n, _ in ActionCfg._fields_[:12]:
        setattr(c, n, float(cfg[n]))
    c.turn_n, c.speed_n, c.stuck_threshold = int(cfg["turn_n"]), int(cfg["speed_n"]), int(stuck_threshold)
    return c


def _f32(v, n):
    a = np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(-1))
    assert a.size == n, (a.size, n)
    return a


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _info(out):
    return {"steer_ctrl": out[ACT_STEER_CTRL], "throttle_ctrl": out[ACT_THROTTLE_CTRL], "brake_ctrl": out[ACT_BRAKE_CTRL],
            "steer_traj": out[ACT_STEER_TRAJ], "throttle_traj": out[ACT_THROTTLE_TRAJ], "brake_traj": out[ACT_BRAKE_TRAJ],
            "desired_speed": out[ACT_DESIRED_SPEED], "angle": out[ACT_ANGLE], "angle_last": out[ACT_ANGLE_LAST],
            "angle_target": out[ACT_ANGLE_TARGET], "angle_final": out[ACT_ANGLE_FINAL], "delta": out[ACT_DELTA],
            "aim": (out[ACT_AIM_X], out[ACT_AIM_Y]), "is_turn": bool(out[ACT_IS_TURN]), "is_stuck": bool(out[ACT_IS_STUCK]),
            "stuck_detector": int(out[ACT_STUCK_DETECTOR])}


class ActionPost:
    """The whole post-processing of a tick as one call.  `device=None`: host entry (tt_action_post_host, state in a
    ctypes struct); a CUDA device: tt_action_post (state and output live in device memory, one D2H copy per tick)."""

    def __init__(self, cfg, stuck_threshold=800, device=None):
        self.cfg = make_cfg(cfg, stuck_threshold)
        self.device = None if device is None else torch.device(device)
        if self.device is None:
            self.state = ActionState()
        else:
            self.state = torch.zeros(ctypes.sizeof(ActionState), dtype=torch.uint8, device=self.device)
            self.out_dev = torch.zeros(ACTION_OUT, dtype=torch.float64, device=self.device)
            self.out_host = torch.zeros(ACTION_OUT, dtype=torch.float64).pin_memory()

    def reset(self):
        if self.device is None:
            self.state = ActionState()
        else:
            self.state.zero_()

    def tick_host(self, mu_last, sigma_last, wp_last, speed, target, stuck_desired_speed=-1.0):
        mu, sg, wp, tg = _f32(mu_last, 2), _f32(sigma_last, 2), _f32(wp_last, 8), _f32(target, 2)
        out = (ctypes.c_double * ACTION_OUT)()
        check(lib().tt_action_post_host(_fp(mu), _fp(sg), _fp(wp), ctypes.c_float(float(speed)), ctypes.c_float(tg[0]),
                                        ctypes.c_float(tg[1]), ctypes.c_float(stuck_desired_speed), ctypes.byref(self.cfg),
                                        ctypes.byref(self.state), out), "tt_action_post_host")
        o = list(out)
        return o[ACT_STEER], o[ACT_THROTTLE], o[ACT_BRAKE], _info(o)

    def tick(self, pred, speed, target, stuck_desired_speed=-1.0):
        """pred: the dict `forward_inference` returned (batch 1, device tensors); speed: float (m/s); target: (x, y)."""
        if self.device is None:
            return self.tick_host(pred["mu_branches"][0, -1].float().cpu().numpy(), pred["sigma_branches"][0, -1].float().cpu().numpy(),
                                  pred["pred_wp"][0, -1].float().cpu().numpy(), speed, target, stuck_desired_speed)
        # the LAST stage's rows of the three heads, where they are: [0, -1] of (1, stages, ...) tensors (views of larger output
        # buffers are fine as long as that row is dense)
        rows = []
        for t, n in ((pred["mu_branches"], 2), (pred["sigma_branches"], 2), (pred["pred_wp"], 8)):
            assert t.is_cuda and t.dtype == torch.float32 and t.shape[0] == 1, "batch-1 f32 outputs"
            r = t[0, -1]
            assert r.numel() == n and r.is_contiguous(), "the last stage's row of a head must be dense"
            rows.append(r)
        assert tuple(pred["pred_wp"].shape[-2:]) == (4, 2)
        last = lambda r, n: ctypes.c_void_p(r.data_ptr())                # noqa: E731
        mu, sg, wp = rows
        tg = _f32(target, 2)
        check(lib().tt_action_post(last(mu, 2), last(sg, 2), last(wp, 8), ctypes.c_float(float(speed)), ctypes.c_float(tg[0]),
                                   ctypes.c_float(tg[1]), ctypes.c_float(stuck_desired_speed), ctypes.byref(self.cfg),
                                   ctypes.c_void_p(self.state.data_ptr()), ctypes.c_void_p(self.out_dev.data_ptr()),
                                   cur_stream(self.device)), "tt_action_post")
        self.out_host.copy_(self.out_dev, non_blocking=True)          # the tick's one device -> host copy
        torch.cuda.current_stream(self.device).synchronize()
        # the forward that produced `pred` is complete here: a wide-chain barrier time-out (NaN outputs) is an error, not a
        # steering command (tt_device_faults reads host-mapped memory: no copy, no further synchronisation)
        if self.device.index is None or self.device.index == torch.cuda.current_device():
            raise_on_device_fault("tt_action_post")
        else:
            with torch.cuda.device(self.device):
                raise_on_device_fault("tt_action_post")
        o = self.out_host.tolist()
        return o[ACT_STEER], o[ACT_THROTTLE], o[ACT_BRAKE], _info(o)


# ------------------------------------------------------------------ the reference's call structure (host entries)
class PIDController:
    """State holder with the reference's constructor (utils.py:7-16); the stepping happens inside tt_action_pid_host."""

    def __init__(self, K_P=1.0, K_I=0.0, K_D=0.0, n=20):
        assert 1 <= n <= PID_WINDOW_MAX
        self.kp, self.ki, self.kd, self.n = K_P, K_I, K_D, n


class _PidPair:
    """cfg + state struct shared by the turn / speed controllers of one model (EDF:47-48)."""

    def __init__(self, cfg, turn, speed):
        c = dict(cfg)
        c.update(turn_KP=turn.kp, turn_KI=turn.ki, turn_KD=turn.kd, turn_n=turn.n, speed_KP=speed.kp, speed_KI=speed.ki,
                 speed_KD=speed.kd, speed_n=speed.n)
        self.cfg, self.state = make_cfg(c), ActionState()


def process_action(pred, command, speed, target_point):
    """-> steer, throttle, brake, metadata (EDF:268-288) through tt_action_ctrl_host."""
    mu = _f32(torch.as_tensor(pred["mu_branches"])[:, -1, :].reshape(-1)[:2].float().cpu().numpy(), 2)
    sg = _f32(torch.as_tensor(pred["sigma_branches"])[:, -1, :].reshape(-1)[:2].float().cpu().numpy(), 2)
    out = (ctypes.c_double * ACTION_OUT)()
    check(lib().tt_action_ctrl_host(_fp(mu), _fp(sg), out), "tt_action_ctrl_host")
    steer, throttle, brake = out[ACT_STEER_CTRL], out[ACT_THROTTLE_CTRL], out[ACT_BRAKE_CTRL]
    meta = {"speed": float(torch.as_tensor(speed).reshape(-1)[0]), "steer": steer, "throttle": throttle, "brake": brake,
            "command": command, "target_point": target_point}
    return steer, throttle, brake, meta


def control_pid(cfg, turn_controller, speed_controller, waypoints, velocity, target, stuck_desired_speed=-1):
    """Waypoint-following PID (EDF:309-390) through tt_action_pid_host.  waypoints (1,4,2), velocity (1,), target (2,)."""
    assert waypoints.shape[0] == 1
    pair = getattr(turn_controller, "_pair", None)
    if pair is None or pair is not getattr(speed_controller, "_pair", None):
        pair = turn_controller._pair = speed_controller._pair = _PidPair(cfg, turn_controller, speed_controller)
    wp = _f32(torch.as_tensor(waypoints)[0].detach().float().cpu().numpy(), 8)
    tg = _f32(target, 2)
    speed = float(torch.as_tensor(velocity).reshape(-1)[0])
    out = (ctypes.c_double * ACTION_OUT)()
    check(lib().tt_action_pid_host(_fp(wp), ctypes.c_float(speed), ctypes.c_float(tg[0]), ctypes.c_float(tg[1]),
                                   ctypes.c_float(stuck_desired_speed), ctypes.byref(pair.cfg), ctypes.byref(pair.state), out),
          "tt_action_pid_host")
    o = list(out)
    w = wp.reshape(4, 2).astype(np.float64)
    meta = {"speed": speed, "steer": o[ACT_STEER_TRAJ], "throttle": o[ACT_THROTTLE_TRAJ], "brake": o[ACT_BRAKE_TRAJ],
            "wp_4": tuple(w[3]), "wp_3": tuple(w[2]), "wp_2": tuple(w[1]), "wp_1": tuple(w[0]),
            "aim": (o[ACT_AIM_X], o[ACT_AIM_Y]), "target": tuple(tg.astype(np.float64)), "desired_speed": o[ACT_DESIRED_SPEED],
            "angle": o[ACT_ANGLE], "angle_last": o[ACT_ANGLE_LAST], "angle_target": o[ACT_ANGLE_TARGET],
            "angle_final": o[ACT_ANGLE_FINAL], "delta": o[ACT_DELTA]}
    return o[ACT_STEER_TRAJ], o[ACT_THROTTLE_TRAJ], bool(o[ACT_BRAKE_TRAJ]), meta


def arbitrate(cfg_struct, state, steer_ctrl, throttle_ctrl, brake_ctrl, throttle_traj, brake_traj, speed):
    """The agent's arbitration stage alone (AGENT:463-509) on a host state struct -> (steer, throttle, brake, info)."""
    out = (ctypes.c_double * ACTION_OUT)()
    check(lib().tt_action_arbitrate_host(ctypes.c_double(steer_ctrl), ctypes.c_double(throttle_ctrl), ctypes.c_double(brake_ctrl),
                                         ctypes.c_double(throttle_traj), ctypes.c_double(float(brake_traj)),
                                         ctypes.c_float(float(speed)), ctypes.byref(cfg_struct), ctypes.byref(state), out),
          "tt_action_arbitrate_host")
    o = list(out)
    return o[ACT_STEER], o[ACT_THROTTLE], o[ACT_BRAKE], {"is_turn": bool(o[ACT_IS_TURN]), "is_stuck": bool(o[ACT_IS_STUCK]),
                                                         "stuck_detector": int(o[ACT_STUCK_DETECTOR])}
