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


def offset_then_rotate(target_xy, ref_xy, ref_yaw):
    """World point -> the ego frame at (ref_xy, ref_yaw): R(ref_yaw)^T (target - ref)  (thinktwice_agent.py:354-360)."""
    d = np.asarray(target_xy, dtype=np.float64) - np.asarray(ref_xy, dtype=np.float64)
    c, s = math.cos(ref_yaw), math.sin(ref_yaw)
    return np.array([c * d[0] + s * d[1], -s * d[0] + c * d[1]])


class AgentTick:
    """The model-side half of `ThinkTwiceAgent.run_step` (leaderboard/team_code/thinktwice_agent.py:362-529) as ONE object:

        raw uint8 frames (4 x 900 x 1600 x 3)  --ImagePreprocessor (tt_preprocess_images)-->  network input
        LiDAR half sweep (n, 4)                --LidarSweepMerger (tt_lidar_merge_half_sweeps)-->  merged cloud, +1 time column
        data queue / sweep selection (AGENT:426-444; `history_query_index_lis = [-1, 0]` at 20 Hz -> the frame `lag` = 10 ticks
        back is sweep 0, the current one the key sweep; zero control until `queue_len` = 31 frames have been seen)
        forward_inference (eager, or with the previous-sweep BEV cache: `use_cache`)
        process_action + control_pid + arbitration + stuck detector --ActionPost (tt_action_post, one 192 B D2H copy)--> control

    What stays outside is the simulator's side of the tick: BGR -> RGB of the CARLA buffers, the GPS filter, the route planner
    (`pos`, `next_wp`, `next_cmd` are arguments) and carla.VehicleControl.  `run_step` returns (steer, throttle, brake, info);
    `info["pred"]` holds the forward's output dict (None while the queue fills)."""

    def __init__(self, model, lag=10, queue_len=31, use_cache=False, stuck_threshold=800, final_dim=None, undistort=True):
        from collections import deque
        from . import calib, preprocess, synth
        from .encoder_decoder import PrevSweepCache
        assert queue_len > lag >= 1
        self.model, self.device = model, model.device
        self.lag, self.queue_len = lag, queue_len
        fd = tuple(final_dim) if final_dim is not None else (calib.FINAL_H, calib.FINAL_W)
        self.pre = preprocess.ImagePreprocessor(final_dim=fd, device=self.device, undistort=undistort)
        self.merger = LidarSweepMerger(self.device)
        self.post = model.action_post(stuck_threshold)                  # control.ActionPost on the model's cfg and device
        self.img_metas = synth.make_img_metas(1, final_dim=fd)         # the fixed evaluation rig (calib.camera_tables)
        self.frames = deque(maxlen=lag + 1)                              # preprocessed frames (4, 3, fh, fw) of the last ticks
        self.cache = PrevSweepCache(model, lag=lag) if use_cache else None
        self.step = -1

    def reset(self):
        self.frames.clear()
        self.merger.reset()
        self.post.reset()
        if self.cache is not None:
            self.cache.reset()
        self.step = -1

    def run_step(self, frames_rgb_u8, lidar_half, pos, compass, speed, next_wp, next_cmd):
        """frames_rgb_u8: uint8 (4, 900, 1600, 3) in camera_list order, host or device; lidar_half (n, 4) x y z intensity;
        pos (x, y) filtered GPS position; compass rad; speed m/s; next_wp (x, y) / next_cmd from the route planner."""
        self.step += 1
        if isinstance(compass, float) and math.isnan(compass):
            compass = 0.0                                                # AGENT:310-313
        raw = torch.as_tensor(frames_rgb_u8)
        if not raw.is_cuda:
            raw = raw.to(self.device, non_blocking=True)
        img = self.pre(raw.contiguous())                                 # (4, 3, fh, fw) f32
        cloud = self.merger.merge(lidar_half, pos, compass)              # (n_prev + n_now, 4)
        self.frames.append(img)
        info = {"step": self.step, "pred": None, "img": img, "cloud": cloud}
        live = self.step >= self.queue_len                               # AGENT:430-436: zero control while the queue fills
        # the previous-sweep cache needs the key-sweep BEVs of the `lag` ticks before the first live one
        warm = self.cache is not None and self.step >= self.queue_len - self.lag and len(self.frames) == self.lag + 1
        if not (live or warm):
            return 0.0, 0.0, 0.0, info
        ego_theta = compass - math.pi / 2                                # AGENT:376-377
        ego_xy = (pos[1], -pos[0])                                       # AGENT:379-381
        target = offset_then_rotate((next_wp[1], -next_wp[0]), ego_xy, ego_theta)       # AGENT:394
        command = int(next_cmd)
        if command < 0:
            command = 4
        command -= 1
        assert command in (0, 1, 2, 3, 4, 5)
        one_hot = torch.zeros(1, 6)
        one_hot[0, command] = 1.0
        pts = torch.zeros(1, 1, cloud.shape[0], 5, dtype=torch.float32, device=self.device)
        pts[0, 0, :, :4] = cloud                                         # carla_dataset.py:315-317: time column 0 for the key sweep
        batch = {"img": torch.stack([self.frames[0], self.frames[-1]])[None],           # [t - lag, t]: AGENT:439-444
                 "points": pts, "img_metas": self.img_metas,
                 "speed": torch.tensor([float(speed)], dtype=torch.float32, device=self.device),
                 "target_point": torch.tensor(target[None], dtype=torch.float32, device=self.device),
                 "target_command": one_hot.to(self.device)}
        with torch.no_grad():
            pred = self.cache.tick(batch) if self.cache is not None else self.model.forward_inference(batch)
        info["pred"] = pred
        if not live:
            return 0.0, 0.0, 0.0, info
        steer, throttle, brake, meta = self.post.tick(pred, float(speed), target)
        info.update(meta)
        info["target_point"] = target
        return steer, throttle, brake, info
