"""Oracle (TEST INFRASTRUCTURE): numpy restatement of the closed-loop agent's per-tick glue
(leaderboard/team_code/thinktwice_agent.py): LiDAR half-sweep merge :340-352 with the pose matrices :47-92, and the
brake / throttle arbitration :463-509.  The pose matrices are pinned by tests/golden/f15_agent_transforms.npz (generated
by executing the reference's own functions); the arbitration depends on carla.VehicleControl and the leaderboard runtime,
absent here: restated from the cited lines, parity unpinned."""
import math

import numpy as np


def transform_matrix(x, y, yaw):            # AGENT:47-60 (roll = pitch = 0 as in the reference call sites)
    cy, sy = math.cos(yaw), math.sin(yaw)
    return np.array([[cy, -sy, 0.0, x], [sy, cy, 0.0, y], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])


def inv_transform_matrix(x, y, yaw):        # AGENT:62-92
    cy, sy = math.cos(yaw), math.sin(yaw)
    nx, ny = -x, -y
    ox = nx * cy + ny * sy
    oy = -nx * sy + ny * cy
    return np.array([[cy, sy, 0.0, ox], [-sy, cy, 0.0, oy], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0]])


class SweepMerge:
    def __init__(self):
        self.prev, self.prev_matrix = None, None

    def step(self, now, pos, compass):      # AGENT:340-352
        if self.prev is not None:
            rel = inv_transform_matrix(pos[1], -pos[0], compass - np.pi / 2) @ self.prev_matrix
            hom = np.concatenate([self.prev[:, :3], np.ones((self.prev.shape[0], 1))], 1)
            moved = np.einsum("ij,kj->ki", rel, hom)
            moved = np.concatenate([moved[:, :3], self.prev[:, 3:4]], 1)
            out = np.concatenate([moved, now], 0).copy()
        else:
            out = now.copy()
        out[:, 2] += 2.5
        self.prev = now
        self.prev_matrix = transform_matrix(pos[1], -pos[0], compass - np.pi / 2)
        return out.astype(np.float32)


class Arbitration:                          # AGENT:463-509
    def __init__(self, stuck_threshold=800):
        self.stuck_detector, self.stuck_threshold = 0, stuck_threshold

    def step(self, steer_ctrl, throttle_ctrl, brake_ctrl, throttle_traj, brake_traj, speed):
        if brake_traj < 0.05:
            brake_traj = 0.0
        if throttle_traj > brake_traj:
            brake_traj = 0.0
        accel = (throttle_traj > 0) or (throttle_ctrl > 0) or (brake_traj < 0.95) or (brake_ctrl < 0.95)
        brk = (brake_traj > 0.2) or (brake_ctrl > 0.2)
        is_turn = abs(steer_ctrl) > 0.07
        thr = 1.5 if is_turn else 3.5
        brake, throttle = (1.0, 0.0) if brk else (0.0, 1.0)
        if self.stuck_detector > self.stuck_threshold:
            brake, throttle = (0.0, 1.0) if accel else (1.0, 0.0)
        if speed < 0.5:
            self.stuck_detector += 1
        elif speed > 0.5:
            self.stuck_detector = 0
        mx = 0.05 if speed > thr else (0.4 if is_turn else 0.6)
        return steer_ctrl, float(np.clip(throttle, 0.0, mx)), brake
