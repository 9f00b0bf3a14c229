// grip_policy.hip -- first layer of the rollout-side policy forward on the matrix cores.
//
// AugmentedNatureCNN (reference models/feature_extractor.py:14-22) starts with Conv2d(4, 32, kernel 8, stride 4) + ReLU on the
// image channels of the uint8 observation / 255 (SB3 preprocessing), and reads two scalars from the sensor-pad channel
// (:41-49). During rollouts that layer is 40 % of the policy's GPU time as separate launches (uint8 -> float NHWC pass, MIOpen
// implicit GEMM at 37 TFLOP/s, bias, ReLU). k_conv1_u8 does all of it in one launch as an implicit GEMM on
// v_mfma_f32_32x32x2_f32 (exact f32 products, f32 accumulation: the same arithmetic as the fp32 training path up to the
// summation order): M = 225 output positions of one image, N = 32 output channels, K = 4 x 8 x 8 = 256.
//
// A workgroup (4 waves) per image at a time. LDS: the image's four 64 x 64 uint8 planes (16 KB) and the
// weights as B[k][n] f32, already divided by 255 (32 KB). Wave w owns the 32-position tiles w and w + 4 (two independent accumulators; positions
// 225..255 of the last tile are padding). Per (ci, ky) a lane reads the 8 bytes of its position's kernel row once
// (two ds_read_b32, 4-byte aligned because the stride is 4) and converts the four bytes of its k-half with
// v_cvt_f32_ubyte; lanes 0-31 / 32-63 read consecutive 128-byte rows of B: conflict-free.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>

int grip_fail(const char *msg);                     // grip_sim.hip
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define C1_IN 4
#define C1_OUT 32
#define C1_K 8
#define C1_S 4
#define C1_HW 64
#define C1_OHW 15
#define C1_POS (C1_OHW * C1_OHW)          // 225
#define C1_KDIM (C1_IN * C1_K * C1_K)     // 256

__device__ __forceinline__ float ubyte_f32(unsigned v, int i) { return (float)((v >> (8 * i)) & 0xffu); }      // v_cvt_f32_ubyte{i}

// B[k][n] = w[n][ci][ky][kx] / 255, k = (ci * 8 + ky) * 8 + kx: once per call into a 32 KB scratch (the per-image workgroups
// then stage it with coalesced 16-byte loads instead of 8192 strided ones each)
__global__ void __launch_bounds__(256) k_conv1_prep(const float *__restrict__ w, long long so, long long sc, long long sy, long long sx, float *__restrict__ Bg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C1_KDIM * C1_OUT) return;
    const int n = i & (C1_OUT - 1), k = i >> 5, kx = k & 7, ky = (k >> 3) & 7, ci = k >> 6;
    Bg[i] = w[n * so + ci * sc + ky * sy + kx * sx] * (1.0f / 255.0f);
}

__global__ void __launch_bounds__(256) k_conv1_u8(const uint8_t *__restrict__ obs, int n_img, int channels, const float *__restrict__ Bg,
                                                  const float *__restrict__ bias, float *__restrict__ out, float *__restrict__ other) {
    __shared__ __attribute__((aligned(16))) uint8_t img[C1_IN * C1_HW * C1_HW];
    __shared__ __attribute__((aligned(16))) float B[C1_KDIM * C1_OUT];
    const int tid = threadIdx.x;
    {   const uint4 *src = reinterpret_cast<const uint4 *>(Bg); uint4 *dst = reinterpret_cast<uint4 *>(B);
        for (int i = tid; i < C1_KDIM * C1_OUT / 4; i += 256) dst[i] = src[i]; }
    const int wave = tid >> 6, l = tid & 63, m = l & 31, half = l >> 5;
    const float bn = bias[m];
  // a workgroup walks images blockIdx.x, + gridDim.x, ...: the launcher sizes the grid so that every workgroup gets the same
  // number (the weights are staged once per workgroup, and no partly filled last round of workgroups is left); while one
  // workgroup of a CU stages its next image the other keeps the matrix cores busy (prefetching the next image through
  // registers was slower: the compiler parks it in LDS before the MFMA loop, exposing the load)
  for (int b = blockIdx.x; b < n_img; b += gridDim.x) {
    const uint8_t *o = obs + (size_t)b * channels * C1_HW * C1_HW;
    __syncthreads();                                            // everybody is done with the previous image
    {   const uint4 *src = reinterpret_cast<const uint4 *>(o); uint4 *dst = reinterpret_cast<uint4 *>(img);
        for (int i = tid; i < C1_IN * C1_HW * C1_HW / 16; i += 256) dst[i] = src[i]; }
    if (tid < 2) other[(size_t)b * 2 + tid] = (float)o[(size_t)(channels - 1) * C1_HW * C1_HW + tid] * (1.0f / 255.0f);
    __syncthreads();
    int base[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int p = min((wave + 4 * t) * 32 + m, C1_POS - 1);
        base[t] = (p / C1_OHW) * C1_S * C1_HW + (p % C1_OHW) * C1_S;          // byte offset of the position's window in a plane
    }
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    const unsigned *img32 = reinterpret_cast<const unsigned *>(img);
#pragma unroll 4
    for (int row = 0; row < C1_IN * C1_K; row++) {                // (ci, ky)
        const int ci = row >> 3, ky = row & 7;
        unsigned a0[2], a1[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int off = (ci * C1_HW * C1_HW + ky * C1_HW + base[t]) >> 2;
            a0[t] = img32[off]; a1[t] = img32[off + 1];                 // kx 0..3, 4..7
        }
        const float *Bk = B + (row * 8 + half) * C1_OUT + m;
#pragma unroll
        for (int j = 0; j < 4; j++) {                                   // k = row * 8 + 2 j + half
            const float bv = Bk[2 * j * C1_OUT];
            const int kx = 2 * j + half;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const unsigned word = kx < 4 ? a0[t] : a1[t];
                const float av = ubyte_f32(word, kx & 3);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int p = (wave + 4 * t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (p < C1_POS) out[((size_t)b * C1_POS + p) * C1_OUT + m] = fmaxf(acc[t][r] + bn, 0.f);
        }
    }
  }
}

extern "C" int grip_conv1_u8(const uint8_t *obs_dev, int n, int channels, const float *weight_dev, const int64_t *weight_strides, const float *bias_dev,
                             float *scratch_dev, float *out_nhwc_dev, float *other_dev, void *stream) {
    if (!obs_dev || !weight_dev || !weight_strides || !bias_dev || !scratch_dev || !out_nhwc_dev || !other_dev || n <= 0 || channels != C1_IN + 1)
        return grip_fail("grip_conv1_u8: need uint8 [n, 5, 64, 64] observations, Conv2d(4, 32, 8, 4) weights and bias, a 32 KB scratch and the two outputs");
    hipLaunchKernelGGL(k_conv1_prep, dim3(C1_KDIM * C1_OUT / 256), dim3(256), 0, (hipStream_t)stream, weight_dev, (long long)weight_strides[0], (long long)weight_strides[1],
                       (long long)weight_strides[2], (long long)weight_strides[3], scratch_dev);
    const int resident = 256 * 3;                               // 48 KB of LDS per workgroup: three per CU
    const int per_wg = (n + resident - 1) / resident, grid = (n + per_wg - 1) / per_wg;
    hipLaunchKernelGGL(k_conv1_u8, dim3(grid), dim3(256), 0, (hipStream_t)stream, obs_dev, n, channels, (const float *)scratch_dev, bias_dev, out_nhwc_dev, other_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char buf[160]; snprintf(buf, sizeof buf, "grip_conv1_u8: %s", hipGetErrorString(e)); return grip_fail(buf); }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// PPO's clipped-surrogate loss of one minibatch with its gradients, in one launch (stable_baselines3 PPO.train as the reference's
// train_agent.py:33-47 configures it: normalize_advantage, clip_range, no value clipping, entropy of a state-independent diagonal
// Gaussian). As tensor-library operations the same arithmetic is ~60 launches of 4096 elements each, a tenth of the update's time.
//   A_i      = (adv_i - mean(adv)) / (std(adv, unbiased) + 1e-8)
//   logp_i   = sum_k [ -(a_ik - mu_ik)^2 / (2 sigma_k^2) - log sigma_k - log(2 pi) / 2 ]
//   ratio_i  = exp(clamp(logp_i - old_logp_i, -20, 20))
//   pl       = -mean_i min(A_i ratio_i, A_i clamp(ratio_i, 1 - clip, 1 + clip)),  vl = mean_i (ret_i - v_i)^2,
//   el       = -sum_k (1/2 + log(2 pi)/2 + log sigma_k),   loss = pl + ent_coef el + vf_coef vl
// out[0..2] = loss, pl, vl;  g_mean [n, A] = dloss / dmu,  g_values [n] = dloss / dv,  g_log_std [A] = dloss / dlog sigma.
// One workgroup (the minibatch is a few thousand rows; three block reductions).
#define PL_THREADS 1024
#define PL_MAXA 8
__device__ __forceinline__ float pl_block_sum(float v, float *red) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    __syncthreads();                                            // red may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PL_THREADS / 64; i++) s += red[i];
    return s;
}

__global__ void __launch_bounds__(PL_THREADS) k_ppo_loss(const float *__restrict__ mean, const float *__restrict__ log_std, const float *__restrict__ values,
                                                        const float *__restrict__ actions, const float *__restrict__ old_logp, const float *__restrict__ adv,
                                                        const float *__restrict__ ret, int n, int A, float clip, float ent_coef, float vf_coef,
                                                        float *__restrict__ out, float *__restrict__ g_mean, float *__restrict__ g_values, float *__restrict__ g_log_std) {
    __shared__ float red[PL_THREADS / 64];
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < n; i += PL_THREADS) s += adv[i];
    const float am = pl_block_sum(s, red) / (float)n;
    s = 0.f;
    for (int i = tid; i < n; i += PL_THREADS) { const float d = adv[i] - am; s += d * d; }
    const float asd = sqrtf(pl_block_sum(s, red) / (float)(n - 1));
    const float ainv = 1.0f / (asd + 1e-8f), invn = 1.0f / (float)n;
    float ls[PL_MAXA], isig[PL_MAXA], gls[PL_MAXA], lsum = 0.f;
#pragma unroll
    for (int k = 0; k < PL_MAXA; k++) { ls[k] = k < A ? log_std[k] : 0.f; isig[k] = expf(-ls[k]); gls[k] = 0.f; lsum += k < A ? ls[k] : 0.f; }
    float sobj = 0.f, sv = 0.f;
    for (int i = tid; i < n; i += PL_THREADS) {
        float z[PL_MAXA], lp = 0.f;
#pragma unroll
        for (int k = 0; k < PL_MAXA; k++) {
            z[k] = k < A ? (actions[(size_t)i * A + k] - mean[(size_t)i * A + k]) * isig[k] : 0.f;
            lp += k < A ? -0.5f * z[k] * z[k] - ls[k] - 0.9189385332046727f : 0.f;
        }
        const float lr_raw = lp - old_logp[i];
        const float lr = fminf(fmaxf(lr_raw, -20.f), 20.f);
        const float ratio = expf(lr), Ai = (adv[i] - am) * ainv;
        const float lo = 1.f - clip, hi = 1.f + clip;
        const float s1 = Ai * ratio, s2 = Ai * fminf(fmaxf(ratio, lo), hi);
        sobj += fminf(s1, s2);
        const bool inside = ratio >= lo && ratio <= hi;
        // d min(s1, s2) / d ratio as the tensor library differentiates it: the smaller branch, ties shared half and half
        const float w1 = s1 < s2 ? 1.f : (s1 == s2 ? 0.5f : 0.f), w2 = (s2 < s1 ? 1.f : (s1 == s2 ? 0.5f : 0.f)) * (inside ? 1.f : 0.f);
        const float pass = (lr_raw >= -20.f && lr_raw <= 20.f) ? 1.f : 0.f;
        const float coef = -invn * Ai * (w1 + w2) * ratio * pass;                  // dloss / dlogp_i
#pragma unroll
        for (int k = 0; k < PL_MAXA; k++) if (k < A) {
            g_mean[(size_t)i * A + k] = coef * z[k] * isig[k];
            gls[k] += coef * (z[k] * z[k] - 1.f);
        }
        const float dv = values[i] - ret[i];
        sv += dv * dv;
        g_values[i] = vf_coef * 2.f * dv * invn;
    }
    const float pl = -pl_block_sum(sobj, red) * invn, vl = pl_block_sum(sv, red) * invn;
#pragma unroll
    for (int k = 0; k < PL_MAXA; k++) {
        if (k < A) {                                                                // A is uniform: every thread takes the same reductions
            const float g = pl_block_sum(gls[k], red);
            if (tid == 0) g_log_std[k] = g - ent_coef;
        }
    }
    if (tid == 0) {
        const float el = -((float)A * (0.5f + 0.9189385332046727f) + lsum);
        out[0] = pl + ent_coef * el + vf_coef * vl; out[1] = pl; out[2] = vl;
    }
}

extern "C" int grip_ppo_loss(const float *mean_dev, const float *log_std_dev, const float *values_dev, const float *actions_dev, const float *old_log_prob_dev,
                             const float *advantages_dev, const float *returns_dev, int n, int action_dim, float clip_range, float ent_coef, float vf_coef,
                             float *out_dev, float *grad_mean_dev, float *grad_values_dev, float *grad_log_std_dev, void *stream) {
    if (!mean_dev || !log_std_dev || !values_dev || !actions_dev || !old_log_prob_dev || !advantages_dev || !returns_dev || !out_dev || !grad_mean_dev ||
        !grad_values_dev || !grad_log_std_dev || n < 2 || action_dim < 1 || action_dim > PL_MAXA)
        return grip_fail("grip_ppo_loss: need n >= 2 rows, 1..8 action dimensions and every array");
    hipLaunchKernelGGL(k_ppo_loss, dim3(1), dim3(PL_THREADS), 0, (hipStream_t)stream, mean_dev, log_std_dev, values_dev, actions_dev, old_log_prob_dev, advantages_dev,
                       returns_dev, n, action_dim, clip_range, ent_coef, vf_coef, out_dev, grad_mean_dev, grad_values_dev, grad_log_std_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char buf[160]; snprintf(buf, sizeof buf, "grip_ppo_loss: %s", hipGetErrorString(e)); return grip_fail(buf); }
    return 0;
}
