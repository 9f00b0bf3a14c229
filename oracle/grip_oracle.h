/*
 * grip_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C, double-precision, one-environment-at-a-time restatement of the
 * reference's hot path: the dm_control/MuJoCo `physics.step()` that
 * simulation/environment/robot_env.py:100,119,142,157 calls, the controller of
 * simulation/controller/actuator.py, the macro step / reward / done logic of
 * robot_env.py:56-241 and reward.py:18-41, for the one model family of
 * xmls/<object>_env.xml.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / CPU baseline. The product
 * (mujoco_rl_manipulate_unknown_objects_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED for the physics: the arithmetic of `physics.step()` lives in
 * the third-party MuJoCo C library (pulled in by dm-control==1.0.3.post1,
 * reference setup.py:7; MuJoCo version not pinned by the reference), which is
 * absent from /root/reference and not installed anywhere in this environment,
 * and the reference holds no golden vectors for it. This
 * file restates MuJoCo's published computation model
 * from memory and is checked by analytic known-answer tests only. The
 * controller / reward / transformation rows ARE pinned: tests/golden/ holds
 * vectors generated from the reference's own Python (tools/make_golden.py).
 */
#ifndef GRIP_ORACLE_H
#define GRIP_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NB 8
#define ORC_NV 13
#define ORC_NQ 14
#define ORC_NU 7
#define ORC_NG 7
#define ORC_MAXCON 48
#define ORC_MAXEFC (7 + 4 * ORC_MAXCON)

typedef struct OrcModel OrcModel;

typedef struct {
    int g1, g2;          /* geom ids, g1 < g2 (0 = floor) */
    double pos[3];
    double frame[9];     /* rows: normal (g1 -> g2), tangent1, tangent2 */
    double dist;         /* signed distance, < margin */
    double friction[3];  /* sliding, torsional, (rolling unused: condim 4) */
    double mu;           /* regularised cone mu */
    int efc_adr;
} OrcContact;

typedef struct {
    /* state */
    double qpos[ORC_NQ], qvel[ORC_NV], ctrl[ORC_NU], qacc_warmstart[ORC_NV];
    double xfrc[ORC_NB][6];   /* force, torque at body COM (world frame) */
    double time;
    /* position-dependent */
    double xpos[ORC_NB][3], xmat[ORC_NB][9], xquat[ORC_NB][4], xipos[ORC_NB][3];
    double dof_axis[ORC_NV][3], dof_anchor[ORC_NV][3];
    double M[ORC_NV][ORC_NV];
    int ncon;
    OrcContact con[ORC_MAXCON];
    /* velocity / force */
    double qfrc_bias[ORC_NV], qfrc_passive[ORC_NV], qfrc_actuator[ORC_NV];
    double qfrc_applied[ORC_NV], qfrc_smooth[ORC_NV], qacc_smooth[ORC_NV];
    double qacc[ORC_NV], qfrc_constraint[ORC_NV];
    /* constraints */
    int nefc;
    int efc_type[ORC_MAXEFC];   /* 0 limit, 1 contact first row, 2 contact other row */
    int efc_id[ORC_MAXEFC];     /* dof (limit) or contact index */
    double efc_J[ORC_MAXEFC][ORC_NV];
    double efc_pos[ORC_MAXEFC], efc_margin[ORC_MAXEFC], efc_vel[ORC_MAXEFC];
    double efc_aref[ORC_MAXEFC], efc_R[ORC_MAXEFC], efc_D[ORC_MAXEFC];
    double efc_force[ORC_MAXEFC];
    int solver_iter;
    int mpr_calls, support_calls;   /* work counters */
} OrcData;

/* reference config (config/base_config.py:12-54) fields the path reads */
typedef struct {
    int max_steps;            /* :36 */
    int time_horizon;         /* :45 */
    int include_roll;         /* :33 */
    int full_observation;     /* :22 */
    int her_buffer;           /* :39 */
    double max_translation;   /* :30 */
    double max_rotation;      /* :29 */
    double pos_tolerance;     /* :32 */
    double grasp_tolerance;   /* :31 */
    double target_dir[2];     /* robot_env.py:30-33: (1,0) or (1,1), unnormalised */
} OrcEnvConfig;

typedef struct {
    OrcData d;
    int episode_step;
    int status;               /* 0 RUNNING, 1 FAIL, 2 TIME_LIMIT (robot_env.py:19-22) */
    int gripper_open;
} OrcEnv;

typedef struct {
    double reward;
    int done;
    int status, episode_step, gripper_open, object_grasped;
    int reached_target, reached_initial, reached_fail;
    double total_distance, line_distance;
    double init_obj_pos[3], final_obj_pos[3], gripper_pos[3];
    float achieved_goal[2], desired_goal[2];
    int pad_grasp, pad_pheromone;     /* sensor pad [0,0] and [0,1] */
    int n_substeps;
    double target_qpos[5];
} OrcStepOut;

/* ---- model ---- */
OrcModel *orc_model_load(const char *path);
void orc_model_free(OrcModel *m);
const char *orc_last_error(void);
double orc_model_scalar(const OrcModel *m, const char *name, int idx);

/* ---- physics (MuJoCo restatement) ---- */
void orc_reset_data(const OrcModel *m, OrcData *d);              /* mj_resetData + forward position */
void orc_fwd_position(const OrcModel *m, OrcData *d);            /* kinematics, M, collision */
void orc_forward(const OrcModel *m, OrcData *d);                 /* everything up to qacc */
void orc_step(const OrcModel *m, OrcData *d);                    /* dm_control Physics.step() */
void orc_set_step_noise(double amp, unsigned seed);               /* conditioning probe, see grip_physics.c; 0 = off (default) */
void orc_set_step_noise2(double amp_qpos, double amp_qvel, unsigned seed);
void orc_jac_body(const OrcModel *m, const OrcData *d, int body, const double point[3],
                  double jacp[3][ORC_NV], double jacr[3][ORC_NV]);
int orc_hull_hull(const OrcModel *m, OrcData *d, int g1, int g2, OrcContact *out);
int orc_plane_hull(const OrcModel *m, const OrcData *d, int g, OrcContact *out);

/* ---- environment (robot_env.py / actuator.py / reward.py restatement) ---- */
void orc_env_config_default(OrcEnvConfig *c);
void orc_env_reset(const OrcModel *m, const OrcEnvConfig *c, OrcEnv *e, OrcStepOut *o);
void orc_env_step(const OrcModel *m, const OrcEnvConfig *c, OrcEnv *e, const double action[6], OrcStepOut *o);
void orc_scale_control(const OrcEnvConfig *c, const double dq[5], double ctrl[5]);
int orc_check_grasp(const OrcData *d);
int orc_pheromone_level(const OrcData *d, const double dir[2]);
void orc_target_pose(const OrcModel *m, const OrcEnvConfig *c, const OrcData *d, const double action[6],
                     double target_qpos[5]);
void orc_enforce_constraints(int include_roll, double position[3], double orientation[3]);   /* actuator.py:266-293 */
void orc_euler_matrix_sxyz(double ai, double aj, double ak, double M[9]);                    /* transformations.py:972-1032 */
void orc_euler_sxyz_from_matrix(const double M[9], double e[3]);                             /* transformations.py:1035-1090 */
void orc_euler_rzyx_from_quat_wxyz(const double q[4], double e[3]);                          /* transformations.py:1093-1102 via :1179 */
double orc_agent_reward(const double init_obj[3], const double final_obj[3], const double dir[2],
                        int gripper_open, const double controls[2], int object_grasped);

/* ---- observation (sensor.py / utils.py restatement) ---- */
void orc_render(const OrcModel *m, const OrcData *d, int width, int height,
                unsigned char *rgb /*h*w*3*/, float *depth /*h*w metres*/);
void orc_transform_depth(float *depth, int n, unsigned char *out);
void orc_observation(const OrcModel *m, const OrcEnvConfig *c, const OrcData *d, unsigned char *obs /*5*64*64*/);
/* reward.py:57-77 (IntrinsicReward.intrinsic_reward) on two CHW uint8 observations */
double orc_intrinsic_reward(const unsigned char *old_obs, const unsigned char *new_obs, int full_observation);

unsigned long orc_sizeof_data(void);
unsigned long orc_sizeof_env(void);

#ifdef __cplusplus
}
#endif
#endif
