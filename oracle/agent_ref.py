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


# ----------------------------------------------------------------------------- action heads (pinned by golden F9)
# `process_action` / `_get_action_beta` / `control_pid` of open_loop_training/code/encoder_decoder_framework.py:267-390 and
# `PIDController` of code/utils.py:7-29, restated statement for statement on numpy / torch CPU scalars with the reference's
# number types.  tests/test_control.py::test_oracle_action_heads_match_reference_golden_f9 holds them to F9 (16 stateful
# ticks of the reference's own methods).
class PID:                                  # CU:7-29
    def __init__(self, K_P=1.0, K_I=0.0, K_D=0.0, n=20):
        from collections import deque
        self._K_P, self._K_I, self._K_D = K_P, K_I, K_D
        self._window = deque([0 for _ in range(n)], maxlen=n)
        self._max = 0.0

    def step(self, error):
        self._window.append(error)
        self._max = max(self._max, abs(error))
        if len(self._window) >= 2:
            integral = np.mean(self._window)
            derivative = self._window[-1] - self._window[-2]
        else:
            integral, derivative = 0.0, 0.0
        return self._K_P * error + self._K_I * integral + self._K_D * derivative


def action_beta(alpha, beta):               # EDF:294-309, torch f32 (1, 2)
    import torch
    x = torch.zeros_like(alpha)
    x[:, 1] += 0.5
    m1 = (alpha > 1) & (beta > 1)
    x[m1] = (alpha[m1] - 1) / (alpha[m1] + beta[m1] - 2)
    x[(alpha <= 1) & (beta > 1)] = 0.0
    x[(alpha > 1) & (beta <= 1)] = 1.0
    m4 = (alpha <= 1) & (beta <= 1)
    x[m4] = alpha[m4] / torch.clamp(alpha[m4] + beta[m4], min=1e-5)
    return x * 2 - 1


def process_action(mu_branches, sigma_branches):      # EDF:267-291 -> steer, throttle, brake
    action = action_beta(mu_branches[:, -1, :].view(1, 2), sigma_branches[:, -1, :].view(1, 2))
    acc, steer = action.cpu().numpy()[0].astype(np.float64)
    if acc >= 0.0:
        throttle, brake = acc, 0.0
    else:
        throttle, brake = 0.0, np.abs(acc)
    return float(np.clip(steer, -1, 1)), float(np.clip(throttle, 0, 1)), float(np.clip(brake, 0, 1))


def control_pid(cfg, turn_controller, speed_controller, waypoints, velocity, target, stuck_desired_speed=-1):   # EDF:314-390
    """waypoints torch (1, 4, 2), velocity torch (1,), target numpy (2,) -> steer, throttle, brake, desired_speed, angle_final."""
    waypoints = waypoints[0].data.cpu().numpy()
    waypoints = waypoints[:, ::-1]
    target = target[::-1]
    num_pairs = len(waypoints) - 1
    best_norm = 1e5
    desired_speed = 0
    aim = waypoints[0]
    for i in range(num_pairs):
        desired_speed += np.linalg.norm(waypoints[i + 1] - waypoints[i]) * 2.0 / num_pairs
        norm = np.linalg.norm((waypoints[i + 1] + waypoints[i]) / 2.0)
        if abs(cfg["aim_dist"] - best_norm) > abs(cfg["aim_dist"] - norm):
            aim = waypoints[i]
            best_norm = norm
    desired_speed = desired_speed.astype(np.float64)
    if stuck_desired_speed > 0:
        desired_speed = stuck_desired_speed
    aim_last = waypoints[-1] - waypoints[-2]
    angle = np.degrees(np.pi / 2 - np.arctan2(aim[1], aim[0])) / 90
    angle_last = np.degrees(np.pi / 2 - np.arctan2(aim_last[1], aim_last[0])) / 90
    angle_target = np.degrees(np.pi / 2 - np.arctan2(target[1], target[0])) / 90
    use_target_to_aim = np.abs(angle_target) < np.abs(angle)
    use_target_to_aim = use_target_to_aim or (np.abs(angle_target - angle_last) > cfg["angle_thresh"] and
                                              target[1] < cfg["dist_thresh"])
    angle_final = angle_target if use_target_to_aim else angle
    angle_final = angle_final.astype(np.float64)
    speed = velocity[0].data.cpu().numpy()
    if speed < 0.01:
        angle_final = 0.0
    steer = np.clip(turn_controller.step(angle_final), -1.0, 1.0)
    brake = desired_speed < cfg["brake_speed"] or (speed / desired_speed) > cfg["brake_ratio"]
    delta = np.clip(desired_speed - speed, 0.0, cfg["clip_delta"])
    throttle = np.clip(speed_controller.step(delta), 0.0, 1.0)
    throttle = throttle if not brake else 0.0
    return float(steer), float(throttle), float(brake), float(desired_speed), float(angle_final)


def offset_then_rotate(target_2d_world_coor, ref_2d_wolrd_coor, ref_yaw):      # AGENT:354-360
    final_coor = target_2d_world_coor - ref_2d_wolrd_coor
    R = np.array([[np.cos(ref_yaw), -np.sin(ref_yaw)], [np.sin(ref_yaw), np.cos(ref_yaw)]])
    return np.einsum("ij,kj->ki", R.T, final_coor)


class AgentChain:
    """The model-side half of `ThinkTwiceAgent.run_step` (AGENT:362-529) as an oracle chain: image pipeline
    (oracle/preprocess_ref.py, pinned by F17) -> half-sweep merge (SweepMerge, F15) -> the data queue and its sweep selection
    (AGENT:426-444: frame `lag` ticks back as sweep 0, the current one as the key sweep; zero control while the queue fills)
    -> `model_ref.forward_inference` (F7 / F8 / F14) -> process_action + control_pid (F9) -> arbitration (F15).
    The simulator-side inputs (averaged GPS position, route planner's next waypoint / command) are arguments."""

    def __init__(self, sd, cfg, mapx, mapy, final_dim, img_metas, lag=10, queue_len=31, stuck_threshold=800):
        from collections import deque
        self.sd, self.cfg, self.maps, self.final_dim, self.img_metas = sd, cfg, (mapx, mapy), tuple(final_dim), img_metas
        self.lag, self.queue_len = lag, queue_len
        self.queue = deque(maxlen=queue_len)
        self.merge = SweepMerge()
        self.arb = Arbitration(stuck_threshold)
        c = cfg["cfg"]
        self.turn = PID(c["turn_KP"], c["turn_KI"], c["turn_KD"], c["turn_n"])
        self.speed = PID(c["speed_KP"], c["speed_KI"], c["speed_KD"], c["speed_n"])
        self.step = -1

    def run_step(self, frames_u8, lidar_half, pos, compass, speed, next_wp, next_cmd, device_img=None, device_cloud=None):
        """`device_img` / `device_cloud`: the product's own stage outputs (preprocessed frames (4, 3, fh, fw), merged cloud (n, 4)).
        The chain always computes its own (kept in `self.last` for the caller's stage-level comparison at that stage's tolerance);
        when given, the DOWNSTREAM oracle stages consume the product's -- a 1e-4 difference in a point coordinate can move a
        point across a voxel boundary, a discrete change that no tolerance on the network outputs would describe."""
        import torch
        from . import model_ref, preprocess_ref
        self.step += 1
        if np.isnan(compass):
            compass = 0.0
        lidar = self.merge.step(np.asarray(lidar_half, dtype=np.float32), pos, compass)
        ego_theta = compass - np.pi / 2
        ego_xy = np.stack([pos[1], -pos[0]], axis=-1)
        target_point = offset_then_rotate(np.array([[next_wp[1], -next_wp[0]]]), ego_xy, ego_theta).squeeze(0)
        command = next_cmd
        if command < 0:
            command = 4
        command -= 1
        one_hot = [0] * 6
        one_hot[command] = 1
        img = preprocess_ref.preprocess(torch.from_numpy(np.asarray(frames_u8)), self.maps[0], self.maps[1], self.final_dim)
        self.last = dict(img=img, cloud=lidar)
        if device_img is not None:
            img = torch.as_tensor(device_img).float()
        if device_cloud is not None:
            lidar = np.asarray(device_cloud, dtype=np.float32)
        self.queue.append(dict(img=img, lidar=lidar))
        if self.step < self.queue_len:
            return 0.0, 0.0, 0.0, None
        sel = [self.queue[-1 * self.lag - 1], self.queue[-1]]                       # AGENT:439-444 with [-1, 0] -> [-11, -1]
        pts = torch.from_numpy(sel[-1]["lidar"])
        pts = torch.cat([pts, torch.zeros(pts.shape[0], 1)], 1)                      # carla_dataset.py:315-317 (time lag 0)
        batch = {"img": torch.stack([q["img"] for q in sel])[None], "points": pts[None, None], "img_metas": self.img_metas,
                 "speed": torch.tensor([speed], dtype=torch.float32),
                 "target_point": torch.from_numpy(target_point[None]).float(),
                 "target_command": torch.tensor([one_hot], dtype=torch.float32)}
        with torch.no_grad():
            pred = model_ref.forward_inference(self.sd, self.cfg, batch)
        v = torch.tensor([speed], dtype=torch.float32)
        s_c, th_c, b_c = process_action(pred["mu_branches"], pred["sigma_branches"])
        s_t, th_t, b_t, _, _ = control_pid(self.cfg["cfg"], self.turn, self.speed, pred["pred_wp"][:, -1], v, target_point)
        steer, throttle, brake = self.arb.step(s_c, th_c, b_c, th_t, b_t, float(v))
        return steer, throttle, brake, pred
