/*
 * grip_env.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Restates the reference's environment layer in plain C / double:
 *   RobotEnv.reset / step            simulation/environment/robot_env.py:56-241
 *   Actuator.*                       simulation/controller/actuator.py:21-293
 *   Reward.agent_reward              simulation/environment/reward.py:18-41
 *   project_to_target_direction      simulation/utils/utils.py:30-31
 *   euler/quaternion helpers         simulation/utils/transformations.py:789-839,
 *                                    972-1102,1179-1198
 * These rows are PINNED by tests/golden/controller_golden.json, generated from
 * the reference's own Python by tools/make_golden.py.
 */
#include "grip_oracle_int.h"

unsigned long orc_sizeof_env(void) { return sizeof(OrcEnv); }

void orc_env_config_default(OrcEnvConfig *c) {
    /* config/base_config.py:12-54 defaults */
    c->max_steps = 400; c->time_horizon = 400; c->include_roll = 1; c->full_observation = 1; c->her_buffer = 0;
    c->max_translation = 0.05; c->max_rotation = 0.15; c->pos_tolerance = 0.002; c->grasp_tolerance = 0.03;
    c->target_dir[0] = 1; c->target_dir[1] = 0;
}

/* utils.py:30-31 */
static double project(const double p[2], const double dir[2]) {
    double n = sqrt(dir[0]*dir[0] + dir[1]*dir[1]);
    return (p[0]*dir[0] + p[1]*dir[1]) / (n * n);
}

/* transformations.py:1179-1198 with the xyzw reorder of :1101 folded in (input is MuJoCo wxyz) */
static void quaternion_matrix_wxyz(const double qw[4], double R[9]) {
    double q[4] = {qw[1], qw[2], qw[3], qw[0]};
    double nq = q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3];
    if (nq < 4.0 * 2.220446049250313e-16) { R[0] = R[4] = R[8] = 1; R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0; return; }
    double s = sqrt(2.0 / nq);
    for (int i = 0; i < 4; i++) q[i] *= s;
    double o[4][4];
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) o[i][j] = q[i] * q[j];
    R[0] = 1.0 - o[1][1] - o[2][2]; R[1] = o[0][1] - o[2][3]; R[2] = o[0][2] + o[1][3];
    R[3] = o[0][1] + o[2][3]; R[4] = 1.0 - o[0][0] - o[2][2]; R[5] = o[1][2] - o[0][3];
    R[6] = o[0][2] - o[1][3]; R[7] = o[1][2] + o[0][3]; R[8] = 1.0 - o[0][0] - o[1][1];
}

/* transformations.py:1035-1090 for the non-repeating i=0,j=1,k=2 case ('sxyz'; 'rzyx' swaps ax/az) */
static void euler_sxyz_from_matrix(const double M[9], double e[3]) {
    const double eps = 4.0 * 2.220446049250313e-16;
    double cy = sqrt(M[0]*M[0] + M[3]*M[3]);
    if (cy > eps) { e[0] = atan2(M[7], M[8]); e[1] = atan2(-M[6], cy); e[2] = atan2(M[3], M[0]); }
    else { e[0] = atan2(-M[5], M[4]); e[1] = atan2(-M[6], cy); e[2] = 0.0; }
}

/* transformations.py:972-1032, axes 'sxyz' */
static void euler_matrix_sxyz(double ai, double aj, double ak, double M[9]) {
    double si = sin(ai), sj = sin(aj), sk = sin(ak), ci = cos(ai), cj = cos(aj), ck = cos(ak);
    double cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
    M[0] = cj * ck; M[1] = sj * sc - cs; M[2] = sj * cc + ss;
    M[3] = cj * sk; M[4] = sj * ss + cc; M[5] = sj * cs - sc;
    M[6] = -sj;     M[7] = cj * si;      M[8] = cj * ci;
}

/* actuator.py:266-293 (_enforce_constraints), in place */
void orc_enforce_constraints(int include_roll, double position[3], double orientation[3]) {
    if (!include_roll) orientation[0] = 0.0;
    else {
        if (orientation[0] > M_PI / 4) orientation[0] = M_PI / 4;
        if (orientation[0] < -M_PI / 4) orientation[0] = -M_PI / 4;
    }
    orientation[1] = 0.0;
    if (position[2] < 0.1) position[2] = 0.1;
    if (position[2] > 0.5) position[2] = 0.5;
}

/* test hooks for the transformations.py rows (a16): the static helpers above, exported one to one */
void orc_euler_matrix_sxyz(double ai, double aj, double ak, double M[9]) { euler_matrix_sxyz(ai, aj, ak, M); }
void orc_euler_sxyz_from_matrix(const double M[9], double e[3]) { euler_sxyz_from_matrix(M, e); }
/* transformations.euler_from_quaternion(q_wxyz, axes=(0, 0, 0, 1)) as actuator.py:53-54 calls it: returns (yaw, pitch, roll) */
void orc_euler_rzyx_from_quat_wxyz(const double q[4], double e[3]) {
    double R[9], s[3];
    quaternion_matrix_wxyz(q, R); euler_sxyz_from_matrix(R, s);
    e[0] = s[2]; e[1] = s[1]; e[2] = s[0];
}

/* actuator.py:58-102 (with :21-44, :249-264, :266-293 and Appendix B's closed-form pinv) */
void orc_target_pose(const OrcModel *m, const OrcEnvConfig *c, const OrcData *d, const double action[6], double target_qpos[5]) {
    (void)m;
    /* _normalise_action: MinMaxScaler((-1,1)) fitted on +-(max_translation x3, max_rotation x2, 1);
     * inverse_transform = (a - min_) / scale_ with min_ = 0. The reference feeds float32 actions and
     * sklearn keeps float32 through inverse_transform and the clip (dtype follows input). */
    float a[6];
    double lo[6] = {-c->max_translation, -c->max_translation, -c->max_translation, -c->max_rotation, -c->max_rotation, -1};
    int k = 0;
    for (int i = 0; i < 6; i++) {
        float v;
        if (!c->include_roll && i == 3) v = 0.f; else v = (float)action[k++];
        double hi = -lo[i], scale = 2.0 / (hi - lo[i]), mn = -1.0 - lo[i] * scale;
        v = (float)((double)v - mn);         /* X -= min_  (float32 array, float64 operand) */
        v = (float)((double)v / scale);      /* X /= scale_ */
        a[i] = v;
    }
    /* _clip_translation_vector: float32 norm, float32 in-place rescale */
    float len = sqrtf(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]);
    if ((double)len > c->max_translation) {
        float f = (float)(c->max_translation / (double)len);
        a[0] *= f; a[1] *= f; a[2] *= f;
    }
    for (int i = 3; i < 5; i++) {
        if ((double)a[i] < -c->max_rotation) a[i] = (float)(-c->max_rotation);
        if ((double)a[i] > c->max_rotation) a[i] = (float)c->max_rotation;
    }
    double translation[3] = {a[0], a[1], a[2]}, rotation[3] = {a[3], 0.0, a[4]};
    /* _get_current_pose */
    double Rq[9], e[3];
    quaternion_matrix_wxyz(d->xquat[B_EE], Rq);
    euler_sxyz_from_matrix(Rq, e);                       /* = (roll, pitch, yaw) after the 'rzyx' swap */
    double cur_ori[3] = {e[0], e[1], e[2]};
    const double *cur_pos = d->xpos[B_EE];
    /* T_world_new = compose(cur_ori, cur_pos) . compose(rotation, translation) */
    double Rold[9], Rrel[9], Rnew[9], t[3], position[3], orientation[3];
    euler_matrix_sxyz(cur_ori[0], cur_ori[1], cur_ori[2], Rold);
    euler_matrix_sxyz(rotation[0], rotation[1], rotation[2], Rrel);
    mulmm3(Rnew, Rold, Rrel);
    mulmv3(t, Rold, translation); add3(position, cur_pos, t);
    euler_sxyz_from_matrix(Rnew, orientation);
    orc_enforce_constraints(c->include_roll, position, orientation);
    double err_pos[3], err_ori[3];
    sub3(err_pos, position, cur_pos); sub3(err_ori, orientation, cur_ori);
    /* J = [jacp[:, :5]; jacr[:, :5]] of body ee at its origin has orthonormal columns: pinv = J^T */
    for (int i = 0; i < 3; i++) target_qpos[i] = d->qpos[i] + dot3(d->dof_axis[i], err_pos);
    target_qpos[3] = d->qpos[3] + dot3(d->dof_axis[3], err_ori);
    target_qpos[4] = d->qpos[4] + dot3(d->dof_axis[4], err_ori);
}

/* actuator.py:46-48: MinMaxScaler.transform => delta * scale_ (+ min_ = 0), first five entries */
void orc_scale_control(const OrcEnvConfig *c, const double dq[5], double ctrl[5]) {
    double st = 2.0 / (2 * c->max_translation), sr = 2.0 / (2 * c->max_rotation);
    double mt = -1.0 + c->max_translation * st, mr = -1.0 + c->max_rotation * sr;
    for (int i = 0; i < 3; i++) ctrl[i] = dq[i] * st + mt;
    for (int i = 3; i < 5; i++) ctrl[i] = dq[i] * sr + mr;
}

/* actuator.py:134-184 */
int orc_check_grasp(const OrcData *d) {
    int t1 = 0, t2 = 0;
    for (int i = 0; i < d->ncon; i++) {
        int g1 = d->con[i].g1, g2 = d->con[i].g2, other;
        if (g1 == G_OBJ) other = g2; else if (g2 == G_OBJ) other = g1; else continue;
        if (other == G_LK || other == G_LF) t1 = 1;
        if (other == G_RK || other == G_RF) t2 = 1;
    }
    return t1 + 2 * t2;
}

/* actuator.py:198-215 */
int orc_pheromone_level(const OrcData *d, const double dir[2]) {
    const double *ee = d->xpos[B_EE];
    double p = project(ee, dir);
    double dx = p * dir[0] - ee[0], dy = p * dir[1] - ee[1];
    double conc = 1.0 / exp(sqrt(dx*dx + dy*dy));
    if (conc > 0.82) return 3;
    if (conc > 0.6) return 2;
    if (conc > 0.37) return 1;
    return 0;
}

/* reward.py:18-41 */
double orc_agent_reward(const double init_obj[3], const double final_obj[3], const double dir[2],
                        int gripper_open, const double controls[2], int object_grasped) {
    double reward = 0.0;
    double p0 = project(init_obj, dir), p1 = project(final_obj, dir);
    double dx = p1 * dir[0] - final_obj[0], dy = p1 * dir[1] - final_obj[1];
    double lateral = sqrt(dx*dx + dy*dy), travel = p1 - p0;
    if (travel > 0.0 && travel < 0.1 && lateral < 0.1) {
        reward = travel;
        /* `np.all(controls) != 0`: true only when both controls are non-zero */
        if (!gripper_open && (controls[0] != 0 && controls[1] != 0) && object_grasped == 3) {
            reward *= 2;
            if (final_obj[2] > 0) reward *= 1.5;
        }
    }
    return reward * 30;
}

static void fill_goals_and_pad(const OrcEnvConfig *c, const OrcEnv *e, OrcStepOut *o) {
    const double *obj = e->d.xpos[B_OBJ];
    double p = project(obj, c->target_dir);
    o->desired_goal[0] = (float)(p * c->target_dir[0]); o->desired_goal[1] = (float)(p * c->target_dir[1]);
    o->achieved_goal[0] = (float)obj[0]; o->achieved_goal[1] = (float)obj[1];
    o->pad_grasp = orc_check_grasp(&e->d);
    o->pad_pheromone = orc_pheromone_level(&e->d, c->target_dir);
}

/* robot_env.py:56-75 */
void orc_env_reset(const OrcModel *m, const OrcEnvConfig *c, OrcEnv *e, OrcStepOut *o) {
    orc_reset_data(m, &e->d);
    e->d.xfrc[B_EE][2] = -(0.438 * m->gravity_z);        /* robot_env.py:64-65, constant verbatim */
    e->episode_step = 0; e->status = 0; e->gripper_open = 1;
    memset(o, 0, sizeof *o);
    fill_goals_and_pad(c, e, o);
    /* reset's desired_goal is the bare target direction (robot_env.py:72) */
    o->desired_goal[0] = (float)c->target_dir[0]; o->desired_goal[1] = (float)c->target_dir[1];
    o->gripper_open = 1;
    copy3(o->final_obj_pos, e->d.xpos[B_OBJ]); copy3(o->gripper_pos, e->d.xpos[B_EE]);
}

static double max_abs_diff(const double *a, const double *b, int n) {
    double mx = 0; for (int i = 0; i < n; i++) { double v = fabs(a[i] - b[i]); if (v > mx) mx = v; } return mx;
}

/* robot_env.py:77-241 */
void orc_env_step(const OrcModel *m, const OrcEnvConfig *c, OrcEnv *e, const double action[6], OrcStepOut *o) {
    OrcData *d = &e->d;
    int reached_target = 0, reached_initial = 0, nsub = 0;
    double init_obj[3], init_qpos[5], target[5], dq[5], ctrl5[5];
    memset(o, 0, sizeof *o);
    copy3(init_obj, d->xpos[B_OBJ]);
    memcpy(init_qpos, d->qpos, sizeof init_qpos);
    double open_close = action[c->include_roll ? 5 : 4];
    orc_target_pose(m, c, d, action, target);
    memcpy(o->target_qpos, target, sizeof target);
    int step_limit = c->max_steps;
    for (int i = 0; i < c->max_steps; i++) {                      /* :97-110 */
        for (int k = 0; k < 5; k++) dq[k] = target[k] - d->qpos[k];
        orc_scale_control(c, dq, ctrl5); memcpy(d->ctrl, ctrl5, sizeof ctrl5);
        orc_step(m, d); nsub++;
        step_limit--;
        if (max_abs_diff(d->qpos, target, 5) < c->pos_tolerance) {   /* post-step qpos (view, quirk Q4) */
            reached_target = 1; for (int k = 0; k < 5; k++) d->ctrl[k] = 0; break;
        }
    }
    if (step_limit == 0) {                                         /* :112-128 */
        memcpy(target, init_qpos, sizeof target);
        for (int i = 0; i < c->max_steps; i++) {
            for (int k = 0; k < 5; k++) dq[k] = target[k] - d->qpos[k];
            orc_scale_control(c, dq, ctrl5); memcpy(d->ctrl, ctrl5, sizeof ctrl5);
            orc_step(m, d); nsub++;
            if (max_abs_diff(d->qpos, target, 5) < c->pos_tolerance) {
                reached_initial = 1; for (int k = 0; k < 5; k++) d->ctrl[k] = 0; break;
            }
        }
    }
    if (!reached_target && !reached_initial) e->status = 1;         /* :130-132 */
    int object_grasped = 0;
    if (reached_target) {                                           /* :136-168 */
        if (open_close > 0. && !e->gripper_open) {
            double tq[2] = {0.4, 0.4};
            d->ctrl[5] = d->ctrl[6] = 0.5;                          /* actuator.py:117-132 */
            for (int i = 0; i < c->max_steps; i++) {
                double delta = max_abs_diff(tq, d->qpos + 5, 2);    /* pre-step */
                orc_step(m, d); nsub++;
                if (delta < c->grasp_tolerance || (d->qpos[5] > tq[0] && d->qpos[6] > tq[1])) {
                    e->gripper_open = 1; break;
                }
            }
            d->ctrl[5] = d->ctrl[6] = 0;
        } else if (open_close < 0. && e->gripper_open) {
            double tq[2] = {-0.4, -0.4};
            d->ctrl[5] = d->ctrl[6] = -1;                           /* actuator.py:104-115 */
            for (int i = 0; i < c->max_steps; i++) {
                double delta = max_abs_diff(tq, d->qpos + 5, 2);
                object_grasped = orc_check_grasp(d);                /* before the step (:155) */
                orc_step(m, d); nsub++;
                if (delta < c->grasp_tolerance) { e->gripper_open = 0; break; }
                if (object_grasped == 3) { e->gripper_open = 0; break; }
            }
            d->ctrl[5] = d->ctrl[6] = 0;
        }
    }
    const double *fobj = d->xpos[B_OBJ], *fee = d->xpos[B_EE];
    {
        double dx = fobj[0] - fee[0], dy = fobj[1] - fee[1];
        if (sqrt(dx*dx + dy*dy) > 1.) e->status = 1;                /* :172-173 */
    }
    fill_goals_and_pad(c, e, o);
    double controls[2] = {d->ctrl[5], d->ctrl[6]};
    double reward = orc_agent_reward(init_obj, fobj, c->target_dir, e->gripper_open, controls, object_grasped);
    if (c->her_buffer) {                                            /* :268-271, float32 goals */
        float gx = o->desired_goal[0] - o->achieved_goal[0], gy = o->desired_goal[1] - o->achieved_goal[1];
        float dist = sqrtf(gx*gx + gy*gy);
        reward += 1.0 / exp((double)dist);
    }
    int done;
    if (e->status != 0) done = 1;
    else if (e->episode_step == c->time_horizon - 1) { done = 1; e->status = 2; }
    else done = 0;
    {
        double dx = fobj[0] - init_obj[0], dy = fobj[1] - init_obj[1];
        o->total_distance = sqrt(dx*dx + dy*dy);
        double p0 = project(init_obj, c->target_dir), p1 = project(fobj, c->target_dir);
        double lx = p1 * c->target_dir[0] - fobj[0], ly = p1 * c->target_dir[1] - fobj[1];
        double lat = sqrt(lx*lx + ly*ly), travel = p1 - p0;
        o->line_distance = (travel > 0. && travel < 0.1 && lat < 0.1) ? travel : 0.;
    }
    e->episode_step += 1;
    o->reward = reward; o->done = done; o->status = e->status; o->episode_step = e->episode_step;
    o->gripper_open = e->gripper_open; o->object_grasped = object_grasped;
    o->reached_target = reached_target; o->reached_initial = reached_initial; o->reached_fail = !reached_target && !reached_initial;
    copy3(o->init_obj_pos, init_obj); copy3(o->final_obj_pos, fobj); copy3(o->gripper_pos, fee);
    o->n_substeps = nsub;
}
