"""Per-tick glue of the closed-loop agent around `forward_inference` (SURVEY 8f-1 LiDAR side, 8f-3):

* `LidarSweepMerger`  -- the two 180-degree LiDAR half sweeps of consecutive ticks merged in the current ego frame
  (leaderboard/team_code/thinktwice_agent.py:340-352), on the device (tt_lidar_merge_half_sweeps);
* `AgentController`   -- the brake / throttle arbitration between the control branch and the trajectory PID, the stuck
  detector and the speed-dependent throttle cap (thinktwice_agent.py:463-509): host scalars, like the reference.
"""
import ctypes
import math

import numpy as np
import torch

from . import ops
from ._lib import check, lib, ptr


def ego_pose_matrix(x, y, yaw):
    """4x4 ego -> world transform of a planar pose (roll = pitch = 0): rotation about z by `yaw`, translation (x, y)
    (thinktwice_agent.py:47-60 with its roll / pitch terms, which are the constants 0 there, dropped)."""
    c, s = math.cos(yaw), math.sin(yaw)
    return np.array([[c, -s, 0.0, x], [s, c, 0.0, y], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])


def inv_ego_pose_matrix(x, y, yaw):
    """world -> ego: the inverse of `ego_pose_matrix` in closed form (thinktwice_agent.py:62-92)."""
    c, s = math.cos(yaw), math.sin(yaw)
    return np.array([[c, s, 0.0, -(c * x + s * y)], [-s, c, 0.0, -(-s * x + c * y)], [0.0, 0.0, 1.0, 0.0],
                     [0.0, 0.0, 0.0, 1.0]])


class LidarSweepMerger:
    """Keeps the previous half sweep (device) and its pose; `merge(now, pos, compass)` returns the (n_prev + n_now, 4)
    cloud the reference hands to the data pipeline: previous half moved into the current ego frame, sensor height added."""

    Z_SHIFT = 2.5

    def __init__(self, device="cuda"):
        self.device = torch.device(device)
        self.prev = None
        self.prev_matrix = None

    def reset(self):
        self.prev, self.prev_matrix = None, None

    @staticmethod
    def _pose_args(pos, compass):
        return pos[1], -pos[0], compass - np.pi / 2          # AGENT:342,351: (y, -x, compass - pi/2)

    def merge(self, now_lidar, pos, compass):
        now = torch.as_tensor(now_lidar, dtype=torch.float32).to(self.device).contiguous()
        assert now.dim() == 2 and now.shape[1] == 4
        n_now = now.shape[0]
        if self.prev is not None:
            rel = inv_ego_pose_matrix(*self._pose_args(pos, compass)) @ self.prev_matrix
            n_prev = self.prev.shape[0]
        else:
            rel, n_prev = np.eye(4), 0
        out = torch.empty(n_prev + n_now, 4, dtype=torch.float32, device=self.device)
        m = (ctypes.c_float * 12)(*[float(v) for v in rel[:3].reshape(-1)])
        check(lib().tt_lidar_merge_half_sweeps(ptr(self.prev), ctypes.c_int(n_prev), ptr(now), ctypes.c_int(n_now), m,
                                               ctypes.c_float(self.Z_SHIFT), ptr(out), ops.cur_stream(self.device)),
              "tt_lidar_merge_half_sweeps")
        self.prev = now
        self.prev_matrix = ego_pose_matrix(*self._pose_args(pos, compass))
        return out


class AgentController:
    """Final control of a tick from the two heads (AGENT:463-509).  Inputs are what `process_action` (control branch)
    and `control_pid` (last refinement stage's waypoints) return; state = the stuck counter."""

    def __init__(self, stuck_threshold=800):
        self.stuck_detector = 0
        self.stuck_threshold = stuck_threshold

    def step(self, steer_ctrl, throttle_ctrl, brake_ctrl, throttle_traj, brake_traj, speed):
        """-> (steer, throttle, brake, info).  `speed` in m/s (the measured velocity)."""
        if brake_traj < 0.05:
            brake_traj = 0.0
        if throttle_traj > brake_traj:
            brake_traj = 0.0
        wants_accel = (throttle_traj > 0) or (throttle_ctrl > 0) or (brake_traj < 0.95) or (brake_ctrl < 0.95)
        wants_brake = (brake_traj > 0.2) or (brake_ctrl > 0.2)
        steer = steer_ctrl
        is_turn = abs(steer) > 0.07
        speed_threshold = 1.5 if is_turn else 3.5            # less stuck in turns / fewer red-light infractions
        brake, throttle = (1.0, 0.0) if wants_brake else (0.0, 1.0)
        is_stuck = self.stuck_detector > self.stuck_threshold
        if is_stuck:                                         # crawl (TransFuser's rule)
            brake, throttle = (0.0, 1.0) if wants_accel else (1.0, 0.0)
        if float(speed) < 0.5:
            self.stuck_detector += 1
        elif float(speed) > 0.5:
            self.stuck_detector = 0
        max_throttle = 0.05 if float(speed) > speed_threshold else (0.4 if is_turn else 0.6)
        throttle = float(np.clip(throttle, 0.0, max_throttle))
        return float(steer), throttle, float(brake), {"is_turn": is_turn, "is_stuck": is_stuck,
                                                      "stuck_detector": self.stuck_detector}
