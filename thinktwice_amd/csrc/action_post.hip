// Action post-processing of a closed-loop tick: Beta-mode control branch, waypoint PID, brake / throttle
// arbitration and stuck detector as ONE entry, compiled once for the host and once for the device from the same
// function.  Reference: code/encoder_decoder_framework.py:268-390 (process_action, _get_action_beta, control_pid),
// code/utils.py:7-29 (PIDController), leaderboard/team_code/thinktwice_agent.py:463-509 (arbitration).
//
// Number types follow the reference's numpy / torch ones on the path (pinned by goldens F9 and F15): the Beta mode is f32
// torch math; waypoint differences, norms, the desired speed and the heading angles are f32 numpy scalars (NEP 50: a
// python float does not widen an f32 scalar); the PID windows, means and the final controls are f64.
#include <math.h>
#include <string.h>

#include "tt_common.h"

namespace tt {

struct ActionIn {
    float mu[2], sigma[2], wp[8];
    float speed, tx, ty, stuck_speed;
};

// PIDController.step (utils.py:17-29): window = deque(maxlen=n) pre-filled with zeros
__host__ __device__ inline double pid_step(double* win, int& head, int n, double kp, double ki, double kd, double err) {
    win[head] = err;                               // append (drops the oldest)
    const int newest = head;
    head = (head + 1) % n;
    double integral = 0.0, derivative = 0.0;
    if (n >= 2) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += win[(head + i) % n];      // oldest -> newest
        integral = s / n;
        derivative = win[newest] - win[(newest + n - 1) % n];
    }
    return kp * err + ki * integral + kd * derivative;
}

__host__ __device__ inline double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

// np.degrees(np.pi / 2 - np.arctan2(y, x)) / 90 on f32 scalars
__host__ __device__ inline float heading(float x, float y) {
    const float a = (float)(M_PI / 2) - atan2f(y, x);
    return (a * (float)(180.0 / M_PI)) / 90.f;
}

// brake / throttle arbitration between the two heads + stuck detector (AGENT:463-509); writes the final control and the
// IS_TURN / IS_STUCK / STUCK_DETECTOR slots of `out`
__host__ __device__ inline void arbitrate(double steer_ctrl, double throttle_ctrl, double brake_ctrl, double throttle_traj,
                                          double brake_traj, float speed_f, const tt_action_cfg& c, tt_action_state& st,
                                          double* out) {
    const double speed = (double)speed_f;
    if (brake_traj < 0.05) brake_traj = 0.0;
    if (throttle_traj > brake_traj) brake_traj = 0.0;
    const bool is_accel = throttle_traj > 0 || throttle_ctrl > 0 || brake_traj < 0.95 || brake_ctrl < 0.95;
    const bool is_brake = brake_traj > 0.2 || brake_ctrl > 0.2;
    const bool is_turn = fabs(steer_ctrl) > 0.07;
    const double speed_threshold = is_turn ? 1.5 : 3.5;
    double brake = is_brake ? 1.0 : 0.0, throttle = is_brake ? 0.0 : 1.0;
    const bool is_stuck = st.stuck_detector > c.stuck_threshold;
    if (is_stuck) {
        brake = is_accel ? 0.0 : 1.0;
        throttle = is_accel ? 1.0 : 0.0;
    }
    if (speed_f < 0.5f) st.stuck_detector += 1;
    else if (speed_f > 0.5f) st.stuck_detector = 0;
    const double max_throttle = speed > speed_threshold ? 0.05 : (is_turn ? 0.4 : 0.6);
    throttle = clipd(throttle, 0.0, max_throttle);
    out[TT_ACT_STEER] = steer_ctrl;
    out[TT_ACT_THROTTLE] = throttle;
    out[TT_ACT_BRAKE] = brake;
    out[TT_ACT_IS_TURN] = is_turn ? 1.0 : 0.0;
    out[TT_ACT_IS_STUCK] = is_stuck ? 1.0 : 0.0;
    out[TT_ACT_STUCK_DETECTOR] = (double)st.stuck_detector;
}

// process_action / _get_action_beta (EDF:268-304), f32 torch math: x = mode (or mean) of Beta(alpha, beta); action = 2x - 1
__host__ __device__ inline void ctrl_branch(const ActionIn& in, double* out) {
    float act[2];
    for (int i = 0; i < 2; ++i) {
        const float a = in.mu[i], b = in.sigma[i];
        float x;
        if (a > 1.f && b > 1.f) x = (a - 1.f) / (a + b - 2.f);
        else if (a <= 1.f && b > 1.f) x = 0.f;
        else if (a > 1.f && b <= 1.f) x = 1.f;
        else if (a <= 1.f && b <= 1.f) {
            const float d = a + b;
            x = a / (d < 1e-5f ? 1e-5f : d);
        } else x = (i == 1) ? 0.5f : 0.f;          // NaN inputs keep the initial value (zeros, second column + 0.5)
        act[i] = x * 2.f - 1.f;
    }
    const double acc = (double)act[0];
    out[TT_ACT_STEER_CTRL] = clipd((double)act[1], -1.0, 1.0);
    out[TT_ACT_THROTTLE_CTRL] = clipd(acc >= 0.0 ? acc : 0.0, 0.0, 1.0);
    out[TT_ACT_BRAKE_CTRL] = clipd(acc >= 0.0 ? 0.0 : fabs(acc), 0.0, 1.0);
}

// control_pid (EDF:309-390).  The reference flips (x, y) -> (y, x) first: w[i] = (wp[i].y, wp[i].x)
__host__ __device__ inline void waypoint_pid(const ActionIn& in, const tt_action_cfg& c, tt_action_state& st, double* out) {
    float wx[4], wy[4];
    for (int i = 0; i < 4; ++i) { wx[i] = in.wp[2 * i + 1]; wy[i] = in.wp[2 * i]; }
    const float tx = in.ty, ty = in.tx;             // target[::-1]
    float desired = 0.f, best = 1e5f, aimx = wx[0], aimy = wy[0];
    for (int i = 0; i < 3; ++i) {
        const float dx = wx[i + 1] - wx[i], dy = wy[i + 1] - wy[i];
        desired += sqrtf(dx * dx + dy * dy) * 2.0f / 3.f;
        const float mx = (wx[i + 1] + wx[i]) / 2.0f, my = (wy[i + 1] + wy[i]) / 2.0f;
        const float nrm = sqrtf(mx * mx + my * my);
        if (fabs(c.aim_dist - (double)best) > fabs(c.aim_dist - (double)nrm)) { aimx = wx[i]; aimy = wy[i]; best = nrm; }
    }
    double desired_speed = (double)desired;
    if (in.stuck_speed > 0.f) desired_speed = (double)in.stuck_speed;
    const float lx = wx[3] - wx[2], ly = wy[3] - wy[2];
    const float angle = heading(aimx, aimy), angle_last = heading(lx, ly), angle_target = heading(tx, ty);
    bool use_target = fabsf(angle_target) < fabsf(angle);
    use_target = use_target || (fabs((double)(angle_target - angle_last)) > c.angle_thresh && (double)ty < c.dist_thresh);
    double angle_final = (double)(use_target ? angle_target : angle);
    const double speed = (double)in.speed;
    if (in.speed < 0.01f) angle_final = 0.0;
    const double steer_traj =
        clipd(pid_step(st.turn_window, st.turn_head, c.turn_n, c.turn_KP, c.turn_KI, c.turn_KD, angle_final), -1.0, 1.0);
    const bool brake_b = desired_speed < c.brake_speed || (speed / desired_speed) > c.brake_ratio;
    const double delta = clipd(desired_speed - speed, 0.0, c.clip_delta);
    double throttle_traj =
        clipd(pid_step(st.speed_window, st.speed_head, c.speed_n, c.speed_KP, c.speed_KI, c.speed_KD, delta), 0.0, 1.0);
    if (brake_b) throttle_traj = 0.0;
    out[TT_ACT_STEER_TRAJ] = steer_traj;
    out[TT_ACT_THROTTLE_TRAJ] = throttle_traj;
    out[TT_ACT_BRAKE_TRAJ] = brake_b ? 1.0 : 0.0;
    out[TT_ACT_DESIRED_SPEED] = desired_speed;
    out[TT_ACT_ANGLE] = (double)angle;
    out[TT_ACT_ANGLE_LAST] = (double)angle_last;
    out[TT_ACT_ANGLE_TARGET] = (double)angle_target;
    out[TT_ACT_ANGLE_FINAL] = angle_final;
    out[TT_ACT_DELTA] = delta;
    out[TT_ACT_AIM_X] = (double)aimy;              // back in the model's (x, y) order
    out[TT_ACT_AIM_Y] = (double)aimx;
}

__host__ __device__ inline void action_post_core(const ActionIn& in, const tt_action_cfg& c, tt_action_state& st,
                                                 double* out) {
    for (int i = 0; i < TT_ACTION_OUT; ++i) out[i] = 0.0;
    ctrl_branch(in, out);
    waypoint_pid(in, c, st, out);
    arbitrate(out[TT_ACT_STEER_CTRL], out[TT_ACT_THROTTLE_CTRL], out[TT_ACT_BRAKE_CTRL], out[TT_ACT_THROTTLE_TRAJ],
              out[TT_ACT_BRAKE_TRAJ], in.speed, c, st, out);
}

__global__ void action_post_kernel(const float* __restrict__ mu, const float* __restrict__ sigma,
                                   const float* __restrict__ wp, float speed, float tx, float ty, float stuck_speed,
                                   tt_action_cfg cfg, tt_action_state* state, double* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ActionIn in;
    for (int i = 0; i < 2; ++i) { in.mu[i] = mu[i]; in.sigma[i] = sigma[i]; }
    for (int i = 0; i < 8; ++i) in.wp[i] = wp[i];
    in.speed = speed; in.tx = tx; in.ty = ty; in.stuck_speed = stuck_speed;
    action_post_core(in, cfg, *state, out);
}

static int check_cfg(const tt_action_cfg* c, const char* who) {
    TT_REQUIRE(c, "%s: null cfg", who);
    TT_REQUIRE(c->turn_n >= 1 && c->turn_n <= TT_PID_WINDOW_MAX && c->speed_n >= 1 && c->speed_n <= TT_PID_WINDOW_MAX,
               "%s: PID window length must be 1..%d (turn_n=%d speed_n=%d)", who, TT_PID_WINDOW_MAX, c->turn_n, c->speed_n);
    return 0;
}

}  // namespace tt

using namespace tt;

extern "C" int tt_action_post(const float* mu_last, const float* sigma_last, const float* wp_last, float speed,
                              float target_x, float target_y, float stuck_desired_speed, const tt_action_cfg* cfg_host,
                              tt_action_state* state, double* out, void* stream) {
    TT_REQUIRE(mu_last && sigma_last && wp_last && state && out, "tt_action_post: null");
    if (int rc = check_cfg(cfg_host, "tt_action_post")) return rc;
    hipLaunchKernelGGL(action_post_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, mu_last, sigma_last, wp_last, speed,
                       target_x, target_y, stuck_desired_speed, *cfg_host, state, out);
    return check_launch("tt_action_post");
}

extern "C" int tt_action_post_host(const float* mu_last, const float* sigma_last, const float* wp_last, float speed,
                                   float target_x, float target_y, float stuck_desired_speed, const tt_action_cfg* cfg,
                                   tt_action_state* state, double* out) {
    TT_REQUIRE(mu_last && sigma_last && wp_last && state && out, "tt_action_post_host: null");
    if (int rc = check_cfg(cfg, "tt_action_post_host")) return rc;
    TT_REQUIRE(state->turn_head >= 0 && state->turn_head < cfg->turn_n && state->speed_head >= 0 &&
                   state->speed_head < cfg->speed_n, "tt_action_post_host: corrupt state");
    ActionIn in;
    memcpy(in.mu, mu_last, sizeof(in.mu));
    memcpy(in.sigma, sigma_last, sizeof(in.sigma));
    memcpy(in.wp, wp_last, sizeof(in.wp));
    in.speed = speed; in.tx = target_x; in.ty = target_y; in.stuck_speed = stuck_desired_speed;
    action_post_core(in, *cfg, *state, out);
    return 0;
}

extern "C" int tt_action_arbitrate_host(double steer_ctrl, double throttle_ctrl, double brake_ctrl, double throttle_traj,
                                        double brake_traj, float speed, const tt_action_cfg* cfg, tt_action_state* state,
                                        double* out) {
    TT_REQUIRE(cfg && state && out, "tt_action_arbitrate_host: null");
    for (int i = 0; i < TT_ACTION_OUT; ++i) out[i] = 0.0;
    arbitrate(steer_ctrl, throttle_ctrl, brake_ctrl, throttle_traj, brake_traj, speed, *cfg, *state, out);
    return 0;
}

extern "C" int tt_action_ctrl_host(const float* mu_last, const float* sigma_last, double* out) {
    TT_REQUIRE(mu_last && sigma_last && out, "tt_action_ctrl_host: null");
    ActionIn in;
    memset(&in, 0, sizeof(in));
    memcpy(in.mu, mu_last, sizeof(in.mu));
    memcpy(in.sigma, sigma_last, sizeof(in.sigma));
    for (int i = 0; i < TT_ACTION_OUT; ++i) out[i] = 0.0;
    ctrl_branch(in, out);
    return 0;
}

extern "C" int tt_action_pid_host(const float* wp_last, float speed, float target_x, float target_y,
                                  float stuck_desired_speed, const tt_action_cfg* cfg, tt_action_state* state, double* out) {
    TT_REQUIRE(wp_last && state && out, "tt_action_pid_host: null");
    if (int rc = check_cfg(cfg, "tt_action_pid_host")) return rc;
    TT_REQUIRE(state->turn_head >= 0 && state->turn_head < cfg->turn_n && state->speed_head >= 0 &&
                   state->speed_head < cfg->speed_n, "tt_action_pid_host: corrupt state");
    ActionIn in;
    memset(&in, 0, sizeof(in));
    memcpy(in.wp, wp_last, sizeof(in.wp));
    in.speed = speed; in.tx = target_x; in.ty = target_y; in.stuck_speed = stuck_desired_speed;
    for (int i = 0; i < TT_ACTION_OUT; ++i) out[i] = 0.0;
    waypoint_pid(in, *cfg, *state, out);
    return 0;
}
