"""Action post-processing of the closed-loop agent: host-side scalar math, mirrored from
EncoderDecoder.process_action / _get_action_beta / control_pid (open_loop_training/code/
encoder_decoder_framework.py:268-390) and PIDController (code/utils.py:7-29).

These run once per simulator tick on a handful of scalars (the reference does them in numpy on the
CPU after a device->host copy); there is nothing to put on the GPU.  State (two 40-entry error windows)
lives in the controller objects, exactly like the reference's deques.
"""
from collections import deque

import numpy as np
import torch


class PIDController:
    def __init__(self, K_P=1.0, K_I=0.0, K_D=0.0, n=20):
        self.kp, self.ki, self.kd = K_P, K_I, K_D
        self.window = deque([0 for _ in range(n)], maxlen=n)
        self.peak = 0.0

    def step(self, error):
        self.window.append(error)
        self.peak = max(self.peak, abs(error))
        if len(self.window) >= 2:
            integral = np.mean(self.window)
            derivative = self.window[-1] - self.window[-2]
        else:
            integral = derivative = 0.0
        return self.kp * error + self.ki * integral + self.kd * derivative


def beta_mode_action(alpha, beta):
    """Mode (or mean when not unimodal) of Beta(alpha, beta) mapped to [-1, 1]; second column defaults
    to 0.5 before mapping (EDF:291-304).  alpha, beta: (1, 2) tensors."""
    a = torch.as_tensor(alpha, dtype=torch.float32).cpu()
    b = torch.as_tensor(beta, dtype=torch.float32).cpu()
    x = torch.zeros_like(a)
    x[:, 1] += 0.5
    both = (a > 1) & (b > 1)
    x[both] = (a[both] - 1) / (a[both] + b[both] - 2)
    x[(a <= 1) & (b > 1)] = 0.0
    x[(a > 1) & (b <= 1)] = 1.0
    flat = (a <= 1) & (b <= 1)
    x[flat] = a[flat] / torch.clamp(a[flat] + b[flat], min=1e-5)
    return x * 2 - 1


def process_action(pred, command, speed, target_point):
    """-> steer, throttle, brake, metadata (EDF:268-288)."""
    act = beta_mode_action(pred["mu_branches"][:, -1, :].reshape(1, 2), pred["sigma_branches"][:, -1, :].reshape(1, 2))
    acc, steer = act.numpy()[0].astype(np.float64)
    throttle, brake = (acc, 0.0) if acc >= 0.0 else (0.0, np.abs(acc))
    throttle, steer, brake = np.clip(throttle, 0, 1), np.clip(steer, -1, 1), np.clip(brake, 0, 1)
    meta = {"speed": float(torch.as_tensor(speed).cpu().numpy().astype(np.float64)), "steer": float(steer),
            "throttle": float(throttle), "brake": float(brake), "command": command, "target_point": target_point}
    return steer, throttle, brake, meta


def control_pid(cfg, turn_controller, speed_controller, waypoints, velocity, target, stuck_desired_speed=-1):
    """Waypoint-following PID (EDF:309-390).  waypoints (1,4,2) tensor, velocity (1,) tensor, target (2,) array."""
    assert waypoints.size(0) == 1
    wp = waypoints[0].detach().cpu().numpy()
    saved_wp, saved_target = wp.copy(), target.copy()
    wp = wp[:, ::-1]
    target = target[::-1]
    pairs = len(wp) - 1
    best, desired_speed, aim = 1e5, 0, wp[0]
    for i in range(pairs):
        desired_speed += np.linalg.norm(wp[i + 1] - wp[i]) * 2.0 / pairs
        mid = np.linalg.norm((wp[i + 1] + wp[i]) / 2.0)
        if abs(cfg["aim_dist"] - best) > abs(cfg["aim_dist"] - mid):
            aim, best = wp[i], mid
    desired_speed = desired_speed.astype(np.float64)
    if stuck_desired_speed > 0:
        desired_speed = stuck_desired_speed
    last = wp[-1] - wp[-2]

    def heading(v):
        return np.degrees(np.pi / 2 - np.arctan2(v[1], v[0])) / 90
    angle, angle_last, angle_target = heading(aim), heading(last), heading(target)
    use_target = np.abs(angle_target) < np.abs(angle)
    use_target = use_target or (np.abs(angle_target - angle_last) > cfg["angle_thresh"] and target[1] < cfg["dist_thresh"])
    angle_final = (angle_target if use_target else angle).astype(np.float64)
    speed = velocity[0].detach().cpu().numpy()
    if speed < 0.01:
        angle_final = 0.0
    steer = np.clip(turn_controller.step(angle_final), -1.0, 1.0)
    brake = desired_speed < cfg["brake_speed"] or (speed / desired_speed) > cfg["brake_ratio"]
    delta = np.clip(desired_speed - speed, 0.0, cfg["clip_delta"])
    throttle = np.clip(speed_controller.step(delta), 0.0, 1.0)
    throttle = throttle if not brake else 0.0
    meta = {"speed": float(speed.astype(np.float64)), "steer": float(steer), "throttle": float(throttle),
            "brake": float(brake), "wp_4": tuple(saved_wp[3].astype(np.float64)), "wp_3": tuple(saved_wp[2].astype(np.float64)),
            "wp_2": tuple(saved_wp[1].astype(np.float64)), "wp_1": tuple(saved_wp[0].astype(np.float64)),
            "aim": tuple(aim.astype(np.float64)), "target": tuple(saved_target.astype(np.float64)),
            "desired_speed": float(desired_speed), "angle": float(angle.astype(np.float64)),
            "angle_last": float(angle_last.astype(np.float64)), "angle_target": float(angle_target.astype(np.float64)),
            "angle_final": float(angle_final), "delta": float(delta.astype(np.float64))}
    return steer, throttle, brake, meta
