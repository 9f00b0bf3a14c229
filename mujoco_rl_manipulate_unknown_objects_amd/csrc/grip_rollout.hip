// grip_rollout.hip -- device-side bookkeeping of the asynchronous rollout (sb3/async_rollout.py).
//
// The time-sliced engine (grip_batch_advance) hands back, every tick, a list of envs that finished a macro step. The
// trainer keeps "decision records" in HBM: record r = (env, obs, action, log_prob, value), whose reward / done arrive
// when the env is listed the next time. Doing that with tensor-library ops costs ~60 launches of a few microseconds per
// tick; k_rollout_tick does all of it in one launch, k_rollout_gae walks every env's record chain backwards (generalised
// advantage estimation, Schulman et al. 2016, as SB3's RolloutBuffer.compute_returns_and_advantage does for lock-step
// buffers). Pure index/scalar work, HBM-resident, one thread per listed env / per env.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include "../../include/grip_sim.h"

int grip_fail(const char *msg);                     // grip_sim.hip: records the message grip_last_error() returns, yields -1
static int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    char buf[256]; snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    return grip_fail(buf);
}

// counter-based uniform in (-1, 1): splitmix64 finaliser over (seed, env, t, component); 24 random bits -> float
__device__ __forceinline__ float synth_action(unsigned long long seed, int env, long long t, int i) {
    unsigned long long x = seed + 0x9E3779B97F4A7C15ULL * ((unsigned long long)env + 1ULL) + 0xBF58476D1CE4E5B9ULL * (unsigned long long)t + 0x94D049BB133111EBULL * ((unsigned long long)i + 1ULL);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL; x ^= x >> 27; x *= 0x94D049BB133111EBULL; x ^= x >> 31;
    return ((float)(unsigned)(x >> 40) + 0.5f) * (2.0f / 16777216.0f) - 1.0f;
}

__global__ void k_rollout_tick(GripRolloutTick a) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.capacity) return;
    const int N = a.n_envs; const long long R = a.n_records;
    const bool valid = r < a.ready_count[0] && a.ready_list[r] >= 0;      // a negative entry is a hole (lists merged from several batches)
    const long long row = a.base[0] + r;
    const int env = valid ? a.ready_list[r] : N;                     // row N / record R: dump rows for masked writes
    // close the previous decision of this env
    const long long prev = a.rec_of_env[env];
    const bool had = valid && prev >= 0;
    if (had) {
        const float rew = a.reward[env]; const float dn = a.done[env] ? 1.f : 0.f;
        a.rewards[prev] = rew; a.dones[prev] = dn; a.next_rec[prev] = row; a.completed[prev] = 1;
        atomicAdd((unsigned long long *)a.n_completed, 1ULL);
        if (a.n_substeps) atomicAdd((unsigned long long *)a.substeps_total, (unsigned long long)a.n_substeps[env]);
        float er = a.ep_ret[env] + rew, el = a.ep_len[env] + 1.f;
        if (dn > 0.f) { atomicAdd(a.ep_ret_sum, er); atomicAdd(a.ep_len_sum, el); atomicAdd(a.ep_count, 1.f); er = 0.f; el = 0.f; }
        a.ep_ret[env] = er; a.ep_len[env] = el;
    }
    // open the new one (with the fused Gaussian head: sample = mean + std * noise, log N(sample | mean, std))
    const bool head = a.noise != nullptr || (a.rng_count != nullptr && a.log_std != nullptr);      // fused Gaussian head: `actions` holds the mean
    const int mstride = a.mean_stride > 0 ? a.mean_stride : a.action_dim, vstride = a.value_stride > 0 ? a.value_stride : 1;
    float logp = head ? 0.f : a.log_probs[r];
    for (int i = 0; i < a.action_dim; i++) {
        float v = a.actions[(size_t)r * mstride + i];
        if (head) {
            float z = a.noise ? a.noise[(size_t)r * a.action_dim + i] : 0.f; const float ls = a.log_std[i];
            if (a.rng_count) {                                             // synthetic stream: the action is given, z follows from it
                const float u = synth_action(a.rng_seed, env, a.rng_count[env], i);
                z = (u - v) * expf(-ls); v = u;
            } else v = fmaf(expf(ls), z, v);
            logp += -0.5f * z * z - ls - 0.91893853320467274178f;          // 0.5 log(2 pi)
        }
        a.actions_buf[(size_t)row * a.action_dim + i] = v;
        a.slot_actions[(size_t)r * a.action_dim + i] = fminf(fmaxf(v, a.low[i]), a.high[i]);
    }
    a.log_probs_buf[row] = logp; a.values_buf[row] = a.values[(size_t)r * vstride];
    a.is_rec[row] = valid ? 1 : 0; a.completed[row] = 0; a.next_rec[row] = -1;
    a.prev_rec[row] = had ? prev : -1;
    a.rec_env[row] = valid ? env : -1;
    if (valid) a.rec_of_env[env] = row;
    if (valid && head && a.rng_count) a.rng_count[env] += 1;
}

extern "C" int grip_rollout_tick(const GripRolloutTick *args, void *stream) {
    if (!args || args->capacity <= 0) return grip_fail("grip_rollout_tick: bad argument");
    int threads = 256, blocks = (args->capacity + threads - 1) / threads;
    hipLaunchKernelGGL(k_rollout_tick, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, *args);
    return launch_status("grip_rollout_tick");
}

// One thread per env: from the env's open record (not completed: it only supplies the bootstrap value) back along
// prev_rec to the start of the rollout.  A_r = delta_r + gamma lambda (1 - done_r) A_next,
// delta_r = reward_r + gamma (1 - done_r) V_next - V_r.
__global__ void k_rollout_gae(int n_envs, const long long *rec_of_env, const long long *prev_rec, const float *rewards, const float *dones,
                              const float *values, float gamma, float lam, float *advantages, float *returns) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_envs) return;
    long long cur = rec_of_env[e];
    if (cur < 0) return;
    float a_next = 0.f, v_next = values[cur];
    advantages[cur] = 0.f; returns[cur] = v_next;
    for (long long r = prev_rec[cur]; r >= 0; r = prev_rec[r]) {
        const float nonterm = 1.f - dones[r], v = values[r];
        const float delta = rewards[r] + gamma * v_next * nonterm - v;
        const float adv = delta + gamma * lam * nonterm * a_next;
        advantages[r] = adv; returns[r] = adv + v;
        a_next = adv; v_next = v;
    }
}

extern "C" int grip_rollout_gae(int n_envs, const int64_t *rec_of_env, const int64_t *prev_rec, const float *rewards, const float *dones,
                                const float *values, float gamma, float gae_lambda, float *advantages, float *returns, void *stream) {
    if (n_envs <= 0 || !rec_of_env || !prev_rec || !rewards || !dones || !values || !advantages || !returns) return grip_fail("grip_rollout_gae: bad argument");
    int threads = 128, blocks = (n_envs + threads - 1) / threads;
    hipLaunchKernelGGL(k_rollout_gae, dim3(blocks), dim3(threads), 0, (hipStream_t)stream, n_envs, (const long long *)rec_of_env, (const long long *)prev_rec,
                       rewards, dones, values, gamma, gae_lambda, advantages, returns);
    return launch_status("grip_rollout_gae");
}
