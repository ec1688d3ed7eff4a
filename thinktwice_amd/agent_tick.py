"""Per-tick glue of the closed-loop agent around `forward_inference` (SURVEY 8f-1 LiDAR side, 8f-3):

* `LidarSweepMerger`  -- the two 180-degree LiDAR half sweeps of consecutive ticks merged in the current ego frame
  (leaderboard/team_code/thinktwice_agent.py:340-352), on the device (tt_lidar_merge_half_sweeps);
* `AgentController`   -- the brake / throttle arbitration between the control branch and the trajectory PID, the stuck
  detector and the speed-dependent throttle cap (thinktwice_agent.py:463-509): a binding of tt_action_arbitrate_host.
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
    and `control_pid` (last refinement stage's waypoints) return; state = the stuck counter.  A binding: the arbitration
    itself is `tt_action_arbitrate_host` (csrc/action_post.hip), the same source the one-call device entry
    `tt_action_post` (`control.ActionPost.tick`) compiles."""

    def __init__(self, stuck_threshold=800):
        from . import config, control
        self._cfg = control.make_cfg(config.model_config()["cfg"], stuck_threshold)
        self._state = control.ActionState()
        self.stuck_threshold = stuck_threshold

    @property
    def stuck_detector(self):
        return int(self._state.stuck_detector)

    def step(self, steer_ctrl, throttle_ctrl, brake_ctrl, throttle_traj, brake_traj, speed):
        """-> (steer, throttle, brake, info).  `speed` in m/s (the measured velocity)."""
        from . import control
        return control.arbitrate(self._cfg, self._state, steer_ctrl, throttle_ctrl, brake_ctrl, throttle_traj, brake_traj,
                                 speed)
