/*
 * grip_sim.h -- C ABI of the MI355X batched rollout engine (libgrip_sim.so).
 *
 * The reference has no FFI of its own: its hot path sits behind Python
 * surfaces and one implicit operator API, dm_control's `Physics`
 * (SURVEY.md §8b). Each entry point below names the reference call sites it
 * replaces. Plain pointers and sizes only; no torch types. All *_dev pointers
 * are HIP device pointers owned by the caller (torch tensors' data_ptr());
 * `stream` is a hipStream_t passed as void* (0 = the null stream). Every call
 * returns 0 on success, <0 on error (text via grip_last_error()); nothing
 * throws across the ABI. One GripBatch is not re-entrant; distinct batches
 * are independent. Per-lane numerical faults (NaN, contact overflow) are
 * reported in GripStepOut.fault, never as a call failure.
 */
#ifndef GRIP_SIM_H
#define GRIP_SIM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GRIP_NQ 14
#define GRIP_NV 13
#define GRIP_NU 7
#define GRIP_OBS_C 5
#define GRIP_OBS_H 64
#define GRIP_OBS_W 64
#define GRIP_MAXCON 14

typedef struct GripModel GripModel;
typedef struct GripBatch GripBatch;

/* config/base_config.py:12-54 -- the fields the hot path reads */
typedef struct {
    int32_t max_steps;          /* :36  inner-loop cap of robot_env.py:95 */
    int32_t time_horizon;       /* :45 */
    int32_t include_roll;       /* :33 */
    int32_t full_observation;   /* :22 */
    int32_t her_buffer;         /* :39 */
    int32_t auto_reset;         /* VecEnv semantics (DummyVecEnv, train_agent.py:22): reset a lane when done */
    float max_translation;      /* :30 */
    float max_rotation;         /* :29 */
    float pos_tolerance;        /* :32 */
    float grasp_tolerance;      /* :31 */
    float target_dir[2];        /* robot_env.py:30-33: (1,0) for direction 0, (1,1) for 45 (unnormalised) */
} GripEnvConfig;

/* Per-env outputs of one macro step: the SoA form of the tuple and `info` dict that
 * RobotEnv.step returns (robot_env.py:226-241). Any pointer may be NULL to skip it. */
typedef struct {
    float *reward;            /* [N]   robot_env.py:188-197 */
    uint8_t *done;            /* [N]   :201-206 */
    float *achieved_goal;     /* [N,2] :178 */
    float *desired_goal;      /* [N,2] :176-177 */
    int32_t *status;          /* [N]   Status enum :19-22 (value before auto-reset) */
    int32_t *episode_step;    /* [N]   :222,234 (value before auto-reset) */
    int32_t *gripper_open;    /* [N]   :231 */
    int32_t *object_grasped;  /* [N]   :233 */
    int32_t *position_reached;/* [N]   bit0 target, bit1 initial, bit2 fail (:83-85,239) */
    float *total_distance;    /* [N]   :209 */
    float *line_distance;     /* [N]   :213-220 */
    float *gripper_position;  /* [N,3] :237 */
    float *object_position;   /* [N,3] :238 */
    float *init_obj_pos;      /* [N,3] :87 */
    int32_t *n_substeps;      /* [N]   physics.step() calls this macro step */
    int32_t *fault;           /* [N]   bit0 diverged (NaN / runaway state: the episode ends as FAIL with reward 0 and the env is
                               *       reset -- dm_control's PhysicsError), bit1 contact-list overflow, bit2 solver hit iteration cap */
} GripStepOut;

const char *grip_last_error(void);

/* mujoco.Physics.from_xml_path (robot_env.py:26): load a compiled model blob
 * (mujoco_rl_manipulate_unknown_objects_amd/model/compiler.py output). Host only. */
int grip_model_load(const char *blob_path, GripModel **out);
void grip_model_free(GripModel *m);
int grip_model_nvert(const GripModel *m);

/* N environments of one model on one device. RobotEnv.__init__ (robot_env.py:24-44), N times. */
int grip_batch_create(const GripModel *m, int n_envs, int device_id, GripBatch **out);
void grip_batch_destroy(GripBatch *b);
int grip_batch_set_config(GripBatch *b, const GripEnvConfig *cfg);
int grip_batch_num_envs(const GripBatch *b);

/* RobotEnv.reset (robot_env.py:56-75): physics.reset(), gravity-compensation xfrc, counters.
 * mask_dev: uint8[N] (reset where != 0) or NULL for all. Goals/pad for the reset state are
 * written to `out` (achieved_goal, desired_goal = target direction, :71-72). */
int grip_batch_reset(GripBatch *b, const uint8_t *mask_dev, const GripStepOut *out, void *stream);

/* RobotEnv.step (robot_env.py:77-241) for every env: controller (actuator.py:58-102), the
 * MOVE / RETURN / OPEN / CLOSE loops of physics.step() (:97-168), reward (reward.py:18-41),
 * done and info. actions_dev: float32 [N,6] ([N,5] when !include_roll). */
int grip_batch_step(GripBatch *b, const float *actions_dev, const GripStepOut *out, void *stream);

/* RobotEnv.get_observation (robot_env.py:275-293): RGB + depth render of `gripper_camera`
 * (sensor.py:56-77), transform_depth (utils.py:11-19) and the sensor pad
 * (pad[0,0] = check_grasp, pad[0,1] = pheromone_level). obs_dev: uint8 [N,5,64,64] CHW
 * ([N,4,64,64] when !full_observation). */
int grip_batch_observe(GripBatch *b, uint8_t *obs_dev, void *stream);

/* ---- asynchronous stepping: the same RobotEnv.step, time-sliced -------------------------------------------------
 * A macro step takes 17..1200 calls of physics.step() depending on the action (robot_env.py:95-168), so lock-step
 * batches wait for their slowest env. grip_batch_advance instead gives every env at most `slice` calls of physics.step():
 *   - an env whose macro step is in flight resumes it;
 *   - an env that is waiting (after reset, or after finishing a macro step) and that the PREVIOUS call listed in
 *     ready_list at row r starts a new macro step with slot_actions_dev[r] (float32 [capacity, 6 or 5]);
 *   - an env that finishes writes its row of `out` (env-major, as grip_batch_step) and waits.
 * Then the waiting envs are listed: ready_list_dev int32 [capacity] (env ids, -1 padded), *ready_count_dev = how many.
 * The caller renders them (grip_batch_observe_list), runs the policy on those rows and passes the actions to the next
 * call. When more than `capacity` envs wait, the rest are listed by later calls (rotating start, nobody starves).
 * lag = 1: the envs listed by call k start in call k+1 (render + policy run between two launches).
 * lag = 2: they start in call k+2, with slot_actions_dev of that call, so that rendering and the policy for list k can run
 * on another stream WHILE call k+1 advances everybody else (the caller alternates two list / action buffers and orders
 * the streams with events). Listed envs stand still in between, so their state and `out` rows are stable to read.
 * budget_us > 0 additionally ends a wavefront's slice once that much wall-clock time has passed (checked between calls
 * of physics.step()), so that envs in expensive contact states do not hold the launch back: cheap envs then take up to
 * `slice` steps per tick, expensive ones fewer. The order in which envs finish then depends on timing; per env the
 * arithmetic and the results are identical to grip_batch_step either way -- only the schedule differs. */
int grip_batch_advance(GripBatch *b, const float *slot_actions_dev, int slice, int budget_us, int lag, int capacity, const GripStepOut *out,
                       int32_t *ready_list_dev, int32_t *ready_count_dev, void *stream);
/* get_observation for the listed envs only: row r of obs_dev (uint8 [capacity,5,64,64]) = env list_dev[r], r < *count_dev.
 * records_dev != NULL: the same bytes are also written to row record_row_dev[0] + r of records_dev (a trainer's observation
 * store; the row base is read on the device so that a captured hipGraph can advance it). obs_dev may then be NULL: the
 * observation is rendered once, into the record rows only (grip_conv1_u8_rows reads them there). */
int grip_batch_observe_list(GripBatch *b, const int32_t *list_dev, const int32_t *count_dev, int capacity, uint8_t *obs_dev,
                            uint8_t *records_dev, const int64_t *record_row_dev, void *stream);

/* RobotEnv.render (robot_env.py:302-340; sensor.py:56-77 with w_zoom / h_zoom): env `env` seen from any camera at any size -- the
 * reference's workbench / upper / gripper views for videos, GIFs and the human window (not on the training path). cam_pose_dev: 12 floats
 * on the device, optical centre (3) then rotation (9, row-major; columns = camera x, y, z axes in world coordinates, the camera looks
 * along -z), or NULL for the model's gripper camera of that env. fovy in degrees. rgb_dev uint8 [height, width, 3] and / or depth_dev
 * float32 [height, width] (metres along the optical axis), either may be NULL. */
int grip_batch_render_camera(GripBatch *b, int env, const float *cam_pose_dev, float fovy_deg, int width, int height, uint8_t *rgb_dev,
                             float *depth_dev, void *stream);

/* ---- rollout recorder: the trainer-side bookkeeping of asynchronous stepping, fused (csrc/grip_rollout.hip) ---------
 * Decision records live in caller-owned device arrays of n_records + 1 rows (row n_records is a dump row); per-env arrays
 * have n_envs + 1 rows (row n_envs: dump). One call per tick, after the policy ran on the listed envs:
 *   for every listed env: its previous record gets reward / done / next_rec and is marked completed (counters and episode
 *   statistics updated); a new record at row base[0] + r takes the policy outputs of row r; slot_actions[r] = the action
 *   clipped to [low, high] for the next grip_batch_advance. The stand-in for SB3's RolloutBuffer.add (no reference file:
 *   the reference trains through stable_baselines3, train_agent.py:82-92). */
typedef struct {
    int32_t n_envs, capacity, action_dim; int64_t n_records;
    const int32_t *ready_list, *ready_count;      /* from grip_batch_advance; rows >= *ready_count and negative entries are skipped
                                                   * (lists of several batches laid side by side leave holes) */
    const int64_t *base;                          /* [1] first record row of this tick */
    const float *reward; const uint8_t *done; const int32_t *n_substeps;   /* GripStepOut arrays [n_envs] (n_substeps may be NULL) */
    const float *actions, *values, *log_probs;    /* policy outputs [capacity, action_dim], [capacity], [capacity] */
    const float *low, *high;                      /* [action_dim] */
    float *slot_actions;                          /* [capacity, action_dim] out */
    int64_t *rec_of_env;                          /* [n_envs + 1] open record of each env, -1 = none */
    float *rewards, *dones; int64_t *next_rec, *prev_rec, *rec_env; uint8_t *completed, *is_rec;    /* [n_records + 1] */
    float *actions_buf, *log_probs_buf, *values_buf;                                               /* [n_records + 1, ...] */
    int64_t *n_completed, *substeps_total;        /* [1] counters */
    float *ep_ret, *ep_len;                       /* [n_envs + 1] running episode return / length */
    float *ep_ret_sum, *ep_len_sum, *ep_count;    /* [1] finished-episode statistics */
    /* optional fused Gaussian head: when noise != NULL, `actions` holds the policy MEAN [capacity, action_dim], noise is
     * N(0,1) of the same shape and log_std [action_dim] the policy's state-independent log standard deviation; the kernel
     * forms action = mean + exp(log_std) * noise and its log-probability itself (`log_probs` is then ignored) */
    const float *noise, *log_std;
    /* optional synthetic action stream (a ~ U(-1, 1)^action_dim from a counter-based generator keyed by (seed, rank,
     * env, t)): when rng_count != NULL (and noise != NULL: the fused head supplies mean and log_std) the action of a listed env is
     * u(rng_seed, env, rng_count[env], i) instead of a sample of the policy, its log-probability is that of u under the policy's
     * Gaussian (what PPO's ratio needs), and rng_count[env] -- [n_envs + 1], the env's decision counter t -- is incremented.
     * rng_seed carries seed and rank. The stream of an env does not depend on the order in which envs finish. */
    int64_t *rng_count; uint64_t rng_seed;
    /* Row strides, in floats, of `actions` and `values` (0 = dense: action_dim and 1): the merged policy | value head leaves [mean | value] as
     * the columns of ONE [capacity, action_dim + 1] matrix, read here in place instead of through two gathered copies. With the synthetic stream
     * (rng_count) `noise` may be NULL: the fused Gaussian head is switched on by log_std alone then. */
    int32_t mean_stride, value_stride;
} GripRolloutTick;
int grip_rollout_tick(const GripRolloutTick *args, void *stream);
/* GAE over every env's record chain, backwards from its open record (whose value bootstraps) along prev_rec. */
int grip_rollout_gae(int n_envs, const int64_t *rec_of_env, const int64_t *prev_rec, const float *rewards, const float *dones,
                     const float *values, float gamma, float gae_lambda, float *advantages, float *returns, void *stream);

/* IntrinsicReward.intrinsic_reward (reward.py:57-77, --im_reward): reward_dev[e] += sum rel_entr(hist(old), hist(new)) over
 * grey levels (and depth with full_observation, averaged). Pair r = (old_obs_dev row old_rows_dev[r] or r when NULL, a
 * negative row skips the pair; new_obs_dev row r), result added to reward_dev[list_dev[r]] (or [r] when list_dev is NULL);
 * with count_dev only pairs r < *count_dev are evaluated. Observations are uint8 [rows, channels, 64, 64]. */
int grip_intrinsic_reward(const uint8_t *old_obs_dev, const int64_t *old_rows_dev, const uint8_t *new_obs_dev, const int32_t *list_dev,
                          const int32_t *count_dev, int n_pairs, int channels, int full_observation, float *reward_dev, void *stream);

/* Policy input in one pass (models/feature_extractor.py:41-49): uint8 [n, channels, 64, 64] -> image channels as float32 / 255
 * in NHWC ([n, 64, 64, channels - 1], i.e. a channels-last [n, channels - 1, 64, 64] tensor) + the two sensor-pad scalars
 * of the last channel / 255 ([n, 2]). */
int grip_obs_preprocess(const uint8_t *obs_dev, int n, int channels, float *img_nhwc_dev, float *other_dev, void *stream);

/* Rollout-side first layer of AugmentedNatureCNN (models/feature_extractor.py:14-22,41-49) in one launch on the matrix cores
 * (fp32 products and sums, computed exactly on v_mfma_f32_32x32x16_bf16: a pixel 0..255 is a bf16, an fp32 weight is the sum of three bf16 terms, and a byte
 * times a bf16 is exact in the fp32 accumulator): obs_dev uint8 [n, 5, 64, 64] -> out_nhwc_dev float32 [n, 15, 15, 32] =
 * relu(conv2d(obs[:, :4] / 255, weight, bias, stride 4)) (i.e. a channels-last [n, 32, 15, 15] tensor) and other_dev float32
 * [n, 2] = obs[:, 4, 0, :2] / 255. weight_dev float32 [32, 4, 8, 8] with element strides weight_strides[4] (any layout),
 * bias_dev float32 [32], scratch_dev 12 288 floats (the weights / 255 as three bf16 terms, the GEMM's B operand, rewritten by every call -- or, with
 * weight_dev == NULL, taken as grip_conv1_prep left them: the rollouts split the weights once per policy update, not once per tick).
 * No autograd (the update's variant with the ReLU mask: grip_conv1_u8_train; its backward: grip_trunk_backward). */
int grip_conv1_prep(const float *weight_dev, const int64_t *weight_strides, float *scratch_dev, void *stream);
int grip_conv1_u8(const uint8_t *obs_dev, int n, int channels, const float *weight_dev, const int64_t *weight_strides, const float *bias_dev,
                  float *scratch_dev, float *out_nhwc_dev, float *other_dev, void *stream);
/* The same on rows row0_dev[0] .. row0_dev[0] + n - 1 of obs_dev (row0_dev: int64 [1] in device memory, NULL = 0): the time-sliced trainer renders
 * the observations of a tick once, straight into its record rows (grip_batch_observe_list with obs_dev = NULL), and the policy reads them there. */
int grip_conv1_u8_rows(const uint8_t *obs_dev, const int64_t *row0_dev, int n, int channels, const float *weight_dev, const int64_t *weight_strides,
                       const float *bias_dev, float *scratch_dev, float *out_nhwc_dev, float *other_dev, void *stream);

/* Rollout-side second and third layers of AugmentedNatureCNN (models/feature_extractor.py:17-21) in one launch on the matrix cores
 * (since round 5 v_mfma_f32_16x16x32_bf16 on three-term splits of both operands, fp32-equivalent; GRIP_CONV23_F32=1 keeps v_mfma_f32_16x16x4_f32): y1_nhwc_dev float32 [n, 15, 15, 32] (grip_conv1_u8's output) ->
 * out_nhwc_dev float32 [n, 4, 4, 64] = relu(conv2d(relu(conv2d(y1, w2, b2, stride 2)), w3, b3, stride 1)), i.e. a channels-last
 * [n, 64, 4, 4] tensor. The weights are passed as the GEMMs' B matrices, which grip_conv23_prep writes from w2 float32 [64, 32, 4, 4]
 * and w3 float32 [64, 64, 3, 3] (element strides w*_strides[4], any layout) into b2_mat_dev (2560 x 64 floats: the matrix k-major, 512 x 64, then
 * channel-major, then twice 196 608 bytes of operand fragments for the bf16 kernels -- every weight as three bf16 terms; the forward's, then the
 * data gradient's transposed ones) and b3_mat_dev (2880 x 64 floats: 576 x 64 twice, then twice 221 184 bytes of fragments): call it again whenever the weights change. No autograd (the update's variant: grip_conv23_train). */
int grip_conv23_prep(const float *w2_dev, const int64_t *w2_strides, const float *w3_dev, const int64_t *w3_strides, float *b2_mat_dev, float *b3_mat_dev,
                     void *stream);
int grip_conv23(const float *y1_nhwc_dev, int n, const float *b2_mat_dev, const float *bias2_dev, const float *b3_mat_dev, const float *bias3_dev,
                float *out_nhwc_dev, void *stream);

/* Update-side forward of the same layers: the kernels above with the extra outputs the backward pass wants. grip_conv1_u8_train also writes the
 * first layer's ReLU mask, mask_dev uint32 [n, 225]: bit c of word (image, position) = channel c is active (NULL: none), and can take its n images
 * from rows rows_dev[0 .. n - 1] (int64, device) of obs_dev -- a minibatch read where it lies in the rollout storage (NULL: consecutive rows from row0).
 * grip_conv23_train also writes y2_nhwc_dev float32 [n, 6, 6, 64] (the second layer's output, operand of the third layer's weight gradient) and
 * the two layers' masks, mask2_dev uint64 [n, 36] and mask3_dev uint64 [n, 16], bit c = channel c (all three NULL: plain grip_conv23). */
int grip_conv1_u8_train(const uint8_t *obs_dev, const int64_t *row0_dev, const int64_t *rows_dev, int n, int channels, const float *weight_dev,
                        const int64_t *weight_strides, const float *bias_dev, float *scratch_dev, float *out_nhwc_dev, float *other_dev, uint32_t *mask_dev, void *stream);
int grip_conv23_train(const float *y1_nhwc_dev, int n, const float *b2_mat_dev, const float *bias2_dev, const float *b3_mat_dev, const float *bias3_dev,
                      float *out_nhwc_dev, float *y2_nhwc_dev, uint64_t *mask2_dev, uint64_t *mask3_dev, void *stream);

/* Update-side backward of the three convolutions (csrc/grip_train.hip): what the tensor library's convolution_backward, threshold_backward and
 * bias-gradient sums compute for the reference's fp32 training path (stable_baselines3 PPO.train / SAC.train on models/feature_extractor.py:14-22),
 * in one launch plus a small reduction. g3_dev float32 [n, 4, 4, 64] = d loss / d y3 (NHWC); the three ReLU masks of the training forward; obs_dev
 * uint8 [n, channels = 5, 64, 64] (NULL: no first-layer weight gradient), or with obs_rows_dev (int64 [n], device) the rows of a larger store the
 * images are; the weight matrices of grip_conv23_prep. Outputs: g3m_dev = g3 masked (may
 * be NULL) and g2m_dev [n, 6, 6, 64] = d loss / d (second layer's pre-activation) -- the operands of those layers' weight gradients; g1m_dev
 * [n, 15, 15, 32] = d loss / d (first layer's pre-activation) (may be NULL when obs_dev is given: the tile is consumed on chip); grad_w1_dev =
 * d loss / d w1 as float32 [32, 4, 8, 8] with element strides grad_w1_strides[4] (w.r.t. the weight that multiplies obs / 255), grad_b1_dev [32].
 * With obs_dev also the other two layers' bias gradients, grad_b2_dev [64] and grad_b3_dev [64] (sums of g2m / g3m; either may be NULL).
 * partials_dev: scratch of grip_trunk_backward_parts(n) x 8352 floats. fp32-equivalent arithmetic on the matrix cores (since round 5 the two data
 * gradients on the bf16 pipe: gradient tile and weights as three bf16 terms, six products; GRIP_TRUNK_F32=1 in the environment keeps the fp32 instructions); sums in a
 * fixed order (no atomics): bit-identical from run to run on one device (the partial sums' grouping follows the CU count). */
int grip_trunk_backward_parts(int n);
int grip_trunk_backward(const float *g3_dev, const uint64_t *mask3_dev, const uint64_t *mask2_dev, const uint32_t *mask1_dev, const uint8_t *obs_dev,
                        const int64_t *obs_rows_dev, int channels,
                        const float *b3_mat_dev, const float *b2_mat_dev, int n, float *g3m_dev, float *g2m_dev, float *g1m_dev, float *partials_dev,
                        float *grad_w1_dev, const int64_t *grad_w1_strides, float *grad_b1_dev, float *grad_b2_dev, float *grad_b3_dev, void *stream);

/* The weight gradients of the second and third convolution (csrc/grip_train.hip, k_wgrad23_b3): what convolution_backward's weight output computes for
 * Conv2d(32, 64, 4, 2) and Conv2d(64, 64, 3, 1) of models/feature_extractor.py:14-22 in stable_baselines3's PPO.train / SAC.train.
 * y1_dev [n, 15, 15, 32] and y2_dev [n, 6, 6, 64]: the layers' inputs (post-ReLU activations of the training forward, NHWC float32); g2m_dev [n, 6, 6, 64] and
 * g3m_dev [n, 4, 4, 64]: d loss / d (pre-activation) of the two layers (grip_trunk_backward). grad_w2_dev = float32 [64, 32, 4, 4] and grad_w3_dev = [64, 64, 3, 3]
 * with element strides grad_w?_strides[4]. scratch_dev: grip_wgrad23_scratch_floats(n) floats (-1: no device). fp32-equivalent arithmetic on the bf16 matrix
 * pipe (both operands as three bf16 terms, six products); sums in a fixed order that depends on n and the device's CU count only. */
long long grip_wgrad23_scratch_floats(int n);
int grip_wgrad23(const float *y1_dev, const float *g2m_dev, const float *y2_dev, const float *g3m_dev, int n, float *scratch_dev, float *grad_w2_dev,
                 const int64_t *grad_w2_strides, float *grad_w3_dev, const int64_t *grad_w3_strides, void *stream);

/* Gradient clipping + Adam for a list of float32 tensors in two launches: what torch.nn.utils.clip_grad_norm_(parameters, max_norm) followed by
 * torch.optim.Adam.step() (no weight decay, no amsgrad) do in stable_baselines3's PPO.train for the reference's train_agent.py:33-47. Host arrays of
 * n_tensors (<= 48) device addresses: parameters, gradients (scaled in place by min(1, max_norm / (||g|| + 1e-6))), exp_avg, exp_avg_sq, and the step
 * counters (one float32 each, advanced by one); numel[t] elements each, all five of a tensor laid out alike. partials_dev: scratch of
 * grip_clip_adam_chunks(n_tensors, numel) floats; norm_out_dev: the gradient norm before clipping (may be NULL). The addresses travel in the kernel
 * arguments: the call can be captured in a graph. */
int grip_clip_adam_chunks(int n_tensors, const int64_t *numel);
int grip_clip_adam(int n_tensors, const int64_t *numel, float *const *params_dev, float *const *grads_dev, float *const *exp_avg_dev, float *const *exp_avg_sq_dev,
                   float *const *steps_dev, float lr, float beta1, float beta2, float eps, float max_norm, float *partials_dev, float *norm_out_dev, void *stream);

/* Backward of a tanh layer of the policy | value MLPs with its bias gradient, one pass: gz = g * (1 - h^2), grad_bias[b * cols + c] = sum over rows of gz.
 * g_dev float32 [batch, n, cols] (contiguous); h_dev (the layer's output) and gz_dev are addressed b * h_batch_stride + row * row_stride + c -- with
 * h_batch_stride = cols and row_stride = batch * cols the result is laid out row-major [n, batch * cols] (a transposition of the batch-major gradient
 * folded into the pass), with batch = 1 it is the plain case. scratch_dev: ceil(n / 32) * batch * cols floats. Sums in a fixed order. What torch's tanh_backward + sum(0) compute in stable_baselines3's PPO.train. */
int grip_tanh_backward_colsum(const float *g_dev, const float *h_dev, float *gz_dev, int batch, int n, int cols, int row_stride, int64_t h_batch_stride,
                              float *scratch_dev, float *grad_bias_dev, void *stream);
/* The ReLU counterpart for the extractor's linear layer (models/feature_extractor.py:22): gz = g * (h > 0) and its column sums; g_dev's rows are
 * g_row_stride >= cols floats apart (the leading columns of a wider gradient), h_dev / gz_dev float32 [n, cols]; scratch_dev: ceil(n / 32) * cols floats. */
int grip_relu_backward_colsum(const float *g_dev, int g_row_stride, const float *h_dev, float *gz_dev, int n, int cols, float *scratch_dev, float *grad_bias_dev,
                              void *stream);

/* z = tanh(z + bias) in place: z_dev float32 [batch, n, cols] contiguous, bias_dev [batch, cols], cols a multiple of 4 (a tanh layer of the policy | value MLPs,
 * models as stable_baselines3's MlpExtractor builds them for the reference's net_arch, after a batched GEMM that has no bias epilogue). */
int grip_bias_tanh(float *z_dev, const float *bias_dev, int batch, int n, int cols, void *stream);

/* PPO's clipped-surrogate loss of one minibatch and its gradients in one launch (the update stable_baselines3's PPO.train runs for the
 * reference's train_agent.py:33-47: advantages normalised per minibatch, clip_range, no value clipping, diagonal Gaussian with a
 * state-independent log_std): mean_dev / actions_dev float32 [n, action_dim], log_std_dev [action_dim], values / old_log_prob /
 * advantages / returns [n]. out_dev[3] = loss, policy loss, value loss; grad_mean_dev [n, action_dim], grad_values_dev [n],
 * grad_log_std_dev [action_dim] = d loss / d (mean, values, log_std). 2 <= n, 1 <= action_dim <= 8. */
int grip_ppo_loss(const float *mean_dev, const float *log_std_dev, const float *values_dev, const float *actions_dev, const float *old_log_prob_dev,
                  const float *advantages_dev, const float *returns_dev, int n, int action_dim, float clip_range, float ent_coef, float vf_coef,
                  float *out_dev, float *grad_mean_dev, float *grad_values_dev, float *grad_log_std_dev, void *stream);
/* The same loss for the update's explicit launch sequence (sb3/fused_update.py), two launches: (1) the minibatch's rows rows_dev[0..n) (int64) of the rollout's
 * sample arrays actions_dev [R, action_dim], old_log_prob_dev / advantages_dev / returns_dev [R] are gathered into samples_dev (n * (action_dim + 3) + 321 words:
 * actions | old_log_prob | advantages | returns | the loss kernel's 16 x 20 partial sums | its arrival counter, which the gather launch zeroes); (2) the loss reads mean and value in the layout of the merged heads' last batch-of-two GEMM --
 * heads_out_dev float32 [2, n, 8] WITHOUT the heads' biases, head_bias_dev [2, 8] beside it: [0, i, :action_dim] + bias[0] = mean_i, [1, i, 0] + bias[1][0] =
 * value_i -- and writes grad_heads_out_dev [2, n, 8] in that layout (padding = 0),
 * grad_head_bias_dev [2, 8] = its column sums (the action / value heads' bias gradients) and grad_log_std_dev [action_dim]; out_dev as above. The loss runs on 16 workgroups
 * whose partial sums the last one to finish adds in a fixed order; the arrival counter is part of the call's own samples_dev, so calls on different streams do not interfere. */
int grip_ppo_loss_heads(const float *heads_out_dev, const float *head_bias_dev, const float *log_std_dev, const float *actions_dev, const float *old_log_prob_dev, const float *advantages_dev,
                        const float *returns_dev, const int64_t *rows_dev, int n, int action_dim, float clip_range, float ent_coef, float vf_coef, float *samples_dev,
                        float *out_dev, float *grad_heads_out_dev, float *grad_head_bias_dev, float *grad_log_std_dev, void *stream);

/* ---- batch sets: several batches -- other object models, other target directions -- stepped by ONE launch per phase
 * (BASELINE.json configs[3]: {acorn, sand_ball, sugar_cube, bread_crumb} x direction {0, 45} in one process). Env ids are
 * global: batch g owns ids [sum of the sizes before it, ... + its size). `out` (may be NULL) holds result arrays over all
 * envs of the set; it is bound at creation. The batches stay usable on their own (reset, state hooks); after
 * grip_batch_set_config / grip_batch_set_state_storage on a member call grip_batchset_refresh. A single batch runs the same
 * kernels as a set of one, so a set is bit-identical to its batches stepped alone. */
typedef struct GripBatchSet GripBatchSet;
int  grip_batchset_create(GripBatch *const *batches, int n, const GripStepOut *out, GripBatchSet **out_set);
void grip_batchset_destroy(GripBatchSet *s);                 /* the member batches are not destroyed */
int  grip_batchset_refresh(GripBatchSet *s);
int  grip_batchset_num_envs(const GripBatchSet *s);
/* grip_batch_step over the set: actions_dev float32 [total envs, 6 or 5]. */
int  grip_batchset_step(GripBatchSet *s, const float *actions_dev, void *stream);
/* grip_batch_advance over the set. Batch g owns rows [g * capacity / n, (g + 1) * capacity / n) of slot_actions_dev and
 * ready_list_dev; its segment lists its waiting envs (global ids) packed to the front, -1 behind them, so the list has holes
 * and *ready_count_dev is set to capacity: consumers test the sign of an entry (grip_rollout_tick does). */
int  grip_batchset_advance(GripBatchSet *s, const float *slot_actions_dev, int slice, int budget_us, int lag, int capacity,
                           int32_t *ready_list_dev, int32_t *ready_count_dev, void *stream);
/* get_observation of every env of the set, obs_dev uint8 [total envs, 5 or 4, 64, 64]. */
int  grip_batchset_observe(GripBatchSet *s, uint8_t *obs_dev, void *stream);
/* row r of obs_dev (and of records_dev from row record_row_dev[0] on) = observation of env list_dev[r]; negative entries are skipped. */
int  grip_batchset_observe_list(GripBatchSet *s, const int32_t *list_dev, int capacity, uint8_t *obs_dev, uint8_t *records_dev,
                                const int64_t *record_row_dev, void *stream);

/* ---- low-level hooks (the dm_control Physics surface the reference touches; used by tests) ---- */
/* physics.data.qpos / qvel / ctrl / qacc_warmstart, env-major float32 [N,14],[N,13],[N,7],[N,13];
 * host_or_dev = 0: host pointers (synchronous copy), 1: device pointers. NULL skips a field. */
int grip_batch_get_state(GripBatch *b, float *qpos, float *qvel, float *ctrl, float *warm, int host_or_dev, void *stream);
int grip_batch_set_state(GripBatch *b, const float *qpos, const float *qvel, const float *ctrl, const float *warm,
                         int host_or_dev, void *stream);
int grip_batch_get_flags(GripBatch *b, int32_t *episode_step, int32_t *status, int32_t *gripper_open, void *stream); /* host */
int grip_batch_set_flags(GripBatch *b, const int32_t *episode_step, const int32_t *status, const int32_t *gripper_open, void *stream);
/* k calls of physics.step() (robot_env.py:100) with the current ctrl, every env. */
int grip_batch_substep(GripBatch *b, int k, void *stream);
/* data.ncon / data.contact[i] (actuator.py:157-176) and derived quantities of the CURRENT state, to host:
 * ncon int32[N]; con float32 [N,GRIP_MAXCON,10] = pos3, normal3, dist, geom1, geom2, pad;
 * xpos float32 [N,8,3] (bodies world, ee, base, lk, lf, rk, rf, object); qacc float32 [N,13] of a forward pass. */
int grip_batch_debug_forward(GripBatch *b, int32_t *ncon, float *con, float *xpos, float *qacc, float *qacc_smooth,
                             float *M, float *bias, void *stream);
/* mjlib.mj_jacBody for body `ee` (actuator.py:86-95) + pinv IK: target_qpos float32 [N,5] to host. */
int grip_batch_target_pose(GripBatch *b, const float *actions_dev, float *target_qpos_host, void *stream);

/* self-test of the solver's lane-distributed 13x13 Cholesky (the factorisation mj_solNewton does per iteration):
 * x = A^-1 b for n SPD systems, A float32 [n,13,13], b and x float32 [n,13], device pointers. */
int grip_selftest_cholesky(const float *A_dev, const float *b_dev, float *x_dev, int n, void *stream);

/* BASELINE.json configs[4] ("fp16 physics state"): keep qpos / qvel / ctrl of every env as IEEE half in HBM (half != 0) or as
 * fp32 (0, the default). Arithmetic stays fp32: values are rounded to nearest even when a kernel stores the state -- once per
 * grip_batch_step, once per time slice of grip_batch_advance -- and every reader (observation, state hooks) converts on load.
 * qacc_warmstart and the suspended macro-step context stay fp32. Converts the current contents; synchronises the stream.
 * This trades accuracy (positions to 2^-11 relative) for 94 B of the 348 B of state per env; it is off unless asked for. */
int grip_batch_set_state_storage(GripBatch *b, int half, void *stream);

/* timing of the macro-step kernel on its own stream: average ms per launch since the last call with reset != 0 */
int grip_batch_kernel_time(GripBatch *b, int reset, float *ms_avg, int *launches);

/* The same figure from the device's own clock, for EVERY time-slice launch since the last call with reset != 0 -- also the launches a
 * replayed hipGraph makes, which host-side events cannot bracket: the first workgroup of a grip_batch_advance launch stamps its start, every
 * wave its end (wall clock, 100 MHz), and the compaction that follows the launch on its stream adds end - start to a device counter.
 * (Measurement only; the reference has no counterpart: bench.py's `roofline` is quoted on it.) Synchronises `stream`. */
int grip_batch_device_time(GripBatch *b, int reset, double *ms_avg, long long *launches, void *stream);

#ifdef __cplusplus
}
#endif
#endif
