// grip_policy.hip -- the policy's convolutions on the matrix cores, the PPO minibatch loss.
//
// AugmentedNatureCNN (reference models/feature_extractor.py:14-22) starts with Conv2d(4, 32, kernel 8, stride 4) + ReLU on the
// image channels of the uint8 observation / 255 (SB3 preprocessing), and reads two scalars from the sensor-pad channel
// (:41-49). As tensor-library calls that layer is a uint8 -> float NHWC pass, an implicit GEMM at 37 TFLOP/s, bias, ReLU.
// k_conv1_u8 does all of it in one launch as an implicit GEMM: M = 225 output positions of one image, N = 32 output channels,
// K = 4 x 8 x 8 = 256 -- in fp32 arithmetic (exact products, fp32 sums: the training path's arithmetic up to the order of summation)
// at the bf16 matrix rate, because of what the operands ARE: a pixel is an integer 0..255, exactly a bf16; an fp32 weight is
// exactly the sum of three bf16 numbers (8 + 8 + 8 mantissa bits; w / 255 is split once per call). A byte times a bf16 has 16 significant
// bits: every product is exact in the fp32 accumulator of v_mfma_f32_32x32x16_bf16, and three of those instructions (one per weight
// term, 32 cycles each for 16 k) replace eight v_mfma_f32_32x32x2_f32 (64 cycles each): 5.3 x less matrix-pipe time, and the operand
// fragments are 16-byte LDS reads instead of a dword + a conversion per instruction. The fp32-MFMA version ran at 0.65 of the fp32 peak
// (167 us per 4096 images); this one is bound by the 200 MB it moves.
//
// A workgroup (4 waves) per image at a time, two workgroups per CU. LDS: the image's four planes as bf16 (32 KB, converted while staging:
// v_cvt_f32_ubyte + the upper halves packed) and the three weight terms as Bt[term][n][k] bf16 (48 KB, 16-byte chunks XOR-swizzled by the row: the B operand's
// 16-byte reads are conflict-free by construction, without padding -- 80 KB in all, exactly two workgroups per CU). The kernel as a whole is NOT conflict-free:
// SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 48 % (profiles/r03_update_kernels_pmc.txt) -- the A operand's ds_read2_b64 at a 4-pixel (8-byte) stride puts the 32 lanes
// of a group on 8-byte pieces of which several share a bank pair, and the staging stores count too; not separated per operand (it runs at the HBM rate of its 200 MB).
// Wave w owns the 32-position tiles w and w + 4 (positions 225..255 of the last tile are padding). k = (ci, ky, kx): an instruction's 16 k are the
// kernel rows ky, ky + 1 (lanes 0-31 / 32-63) x 8 kx = 8 consecutive pixels, one ds_read2_b64 (8-byte aligned: the stride is 4 pixels).
#include <hip/hip_runtime.h>
#include <atomic>
#include <algorithm>
#include <cstdlib>
#include <stdint.h>
#include <cstdio>

int grip_fail(const char *msg);                     // grip_sim.hip
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define C1_IN 4
#define C1_OUT 32
#define C1_K 8
#define C1_S 4
#define C1_HW 64
#define C1_OHW 15
#define C1_POS (C1_OHW * C1_OHW)          // 225
#define C1_KDIM (C1_IN * C1_K * C1_K)     // 256
#define C1_BROW 256                        // bf16 per row of Bt; the 16-byte chunk c of row n sits at chunk c ^ (n & 7): conflict-free 16-byte reads without padding
#define C1_BT_HALVES (3 * C1_OUT * C1_BROW)

__device__ __forceinline__ float ubyte_f32(unsigned v, int i) { return (float)((v >> (8 * i)) & 0xffu); }      // v_cvt_f32_ubyte{i}
__device__ __forceinline__ unsigned bf16_rne_bits(float x) { const unsigned u = __float_as_uint(x); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; }
__device__ __forceinline__ unsigned pack_hi16(float lo, float hi) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); }   // {hi[31:16], lo[31:16]}

// Bt[term][n][k] (bf16, rows of C1_BROW) = the three bf16 terms of w[n][ci][ky][kx] / 255, k = (ci * 8 + ky) * 8 + kx: once per call into a 48 KB scratch
__global__ void __launch_bounds__(256) k_conv1_prep(const float *__restrict__ w, long long so, long long sc, long long sy, long long sx, uint16_t *__restrict__ Bt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= C1_KDIM * C1_OUT) return;
    const int k = i & (C1_KDIM - 1), n = i >> 8, kx = k & 7, ky = (k >> 3) & 7, ci = k >> 6;
    float r = w[n * so + ci * sc + ky * sy + kx * sx] * (1.0f / 255.0f);
#pragma unroll
    for (int t = 0; t < 3; t++) {                                 // r - (its nearest bf16) is exact in fp32: after three terms at most the 25th bit is left
        const unsigned h = bf16_rne_bits(r);
        Bt[(t * C1_OUT + n) * C1_BROW + 8 * ((k >> 3) ^ (n & 7)) + (k & 7)] = (uint16_t)h;
        r -= __uint_as_float(h << 16);
    }
}

// row0 != NULL: image b is row row0[0] + b of obs (the trainer's record rows of this tick: the observation kernel renders straight into them);
// rows != NULL: image b is row rows[b] (a minibatch of the update, read where it lies instead of gathered first)
__global__ void __launch_bounds__(256, 2) k_conv1_u8(const uint8_t *__restrict__ obs, const long long *__restrict__ row0, const long long *__restrict__ rows, int n_img, int channels,
                                                     const uint16_t *__restrict__ Btg, const float *__restrict__ bias, float *__restrict__ out, float *__restrict__ other,
                                                     uint32_t *__restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) uint16_t c1_lds[];
    uint16_t *img = c1_lds, *Bt = c1_lds + C1_IN * C1_HW * C1_HW;                             // the planes as bf16 (32 KB), the weight terms (48 KB)
    const int tid = threadIdx.x;
    {   const uint4 *src = reinterpret_cast<const uint4 *>(Btg); uint4 *dst = reinterpret_cast<uint4 *>(Bt);
        for (int i = tid; i < C1_BT_HALVES / 8; i += 256) dst[i] = src[i]; }
    const int wave = tid >> 6, l = tid & 63, m = l & 31, half = l >> 5;
    const float bn = bias[m];
    int base[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int p = min((wave + 4 * t) * 32 + m, C1_POS - 1);
        base[t] = ((p / C1_OHW) * C1_S + half) * C1_HW + (p % C1_OHW) * C1_S;                 // pixel index of the lane's first k of a plane's first block
    }
  // a workgroup walks images blockIdx.x, + gridDim.x, ...; the NEXT image's bytes (and its two sensor-pad bytes) are requested before this image's MFMA loop,
  // which issues no global loads of its own: the latency of the 16 KB rides under the loop and the epilogue's stores
  auto image = [&](int b) { return obs + (size_t)(rows ? rows[b] : (row0 ? row0[0] : 0LL) + b) * channels * C1_HW * C1_HW; };
  uint4 px[4];
  unsigned pad = 0;
  if ((int)blockIdx.x < n_img) {
    const uint8_t *o = image(blockIdx.x);
#pragma unroll
    for (int u = 0; u < 4; u++) px[u] = reinterpret_cast<const uint4 *>(o)[tid + 256 * u];
    if (tid < 2) pad = o[(size_t)(channels - 1) * C1_HW * C1_HW + tid];
  }
  for (int b = blockIdx.x; b < n_img; b += gridDim.x) {
    __syncthreads();                                            // everybody is done with the previous image
#pragma unroll
    for (int u = 0; u < 4; u++) {                               // 16 pixels -> 16 bf16 = two 16-byte stores
        const unsigned wv[4] = {px[u].x, px[u].y, px[u].z, px[u].w};
        unsigned h[8];
#pragma unroll
        for (int q = 0; q < 4; q++) { h[2 * q] = pack_hi16(ubyte_f32(wv[q], 0), ubyte_f32(wv[q], 1)); h[2 * q + 1] = pack_hi16(ubyte_f32(wv[q], 2), ubyte_f32(wv[q], 3)); }
        uint4 *d = reinterpret_cast<uint4 *>(img) + 2 * (tid + 256 * u);
        d[0] = make_uint4(h[0], h[1], h[2], h[3]); d[1] = make_uint4(h[4], h[5], h[6], h[7]);
    }
    if (tid < 2) other[(size_t)b * 2 + tid] = (float)pad * (1.0f / 255.0f);
    __syncthreads();
    if (b + (int)gridDim.x < n_img) {
        const uint8_t *o = image(b + gridDim.x);
#pragma unroll
        for (int u = 0; u < 4; u++) px[u] = reinterpret_cast<const uint4 *>(o)[tid + 256 * u];
        if (tid < 2) pad = o[(size_t)(channels - 1) * C1_HW * C1_HW + tid];
    }
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][r] = 0.f;
    const uint16_t *Bl = Bt + m * C1_BROW;                      // + term * 32 * C1_BROW + 8 * ((2 kb + half) ^ (m & 7))
#pragma unroll 4
    for (int kb = 0; kb < 16; kb++) {                           // 16 k: plane kb >> 2, kernel rows 2 (kb & 3) + half
        const int poff = (kb >> 2) * C1_HW * C1_HW + 2 * (kb & 3) * C1_HW;
        bf16x8 af[2], bf[3];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const uint2 *ap = reinterpret_cast<const uint2 *>(img + poff + base[t]);         // 8 pixels = 16 bytes, 8-byte aligned
            const uint2 a0 = ap[0], a1 = ap[1];
            const uint4 av = make_uint4(a0.x, a0.y, a1.x, a1.y);
            af[t] = *reinterpret_cast<const bf16x8 *>(&av);
        }
#pragma unroll
        for (int term = 0; term < 3; term++) bf[term] = *reinterpret_cast<const bf16x8 *>(Bl + term * C1_OUT * C1_BROW + 8 * ((2 * kb + half) ^ (m & 7)));
#pragma unroll
        for (int term = 0; term < 3; term++)
#pragma unroll
            for (int t = 0; t < 2; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t], bf[term], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int p = (wave + 4 * t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float v = acc[t][r] + bn;
            if (p < C1_POS) out[((size_t)b * C1_POS + p) * C1_OUT + m] = fmaxf(v, 0.f);
            if (mask) {                                                 // training: bit c of word (image, position) = channel c is active (the backward's ReLU mask)
                const unsigned long long bal = __ballot(v > 0.f);
                if (m == 0 && p < C1_POS) mask[(size_t)b * C1_POS + p] = half ? (uint32_t)(bal >> 32) : (uint32_t)bal;
            }
        }
    }
  }
}

extern "C" int grip_conv1_u8_train(const uint8_t *obs_dev, const int64_t *row0_dev, const int64_t *rows_dev, int n, int channels, const float *weight_dev,
                                   const int64_t *weight_strides, const float *bias_dev, float *scratch_dev, float *out_nhwc_dev, float *other_dev, uint32_t *mask_dev, void *stream);
extern "C" int grip_conv1_u8_rows(const uint8_t *obs_dev, const int64_t *row0_dev, int n, int channels, const float *weight_dev, const int64_t *weight_strides,
                                  const float *bias_dev, float *scratch_dev, float *out_nhwc_dev, float *other_dev, void *stream) {
    return grip_conv1_u8_train(obs_dev, row0_dev, nullptr, n, channels, weight_dev, weight_strides, bias_dev, scratch_dev, out_nhwc_dev, other_dev, nullptr, stream);
}
extern "C" int grip_conv1_u8(const uint8_t *obs_dev, int n, int channels, const float *weight_dev, const int64_t *weight_strides, const float *bias_dev,
                             float *scratch_dev, float *out_nhwc_dev, float *other_dev, void *stream) {
    return grip_conv1_u8_rows(obs_dev, nullptr, n, channels, weight_dev, weight_strides, bias_dev, scratch_dev, out_nhwc_dev, other_dev, stream);
}
extern "C" int grip_conv1_prep(const float *weight_dev, const int64_t *weight_strides, float *scratch_dev, void *stream) {
    if (!weight_dev || !weight_strides || !scratch_dev) return grip_fail("grip_conv1_prep: need the Conv2d(4, 32, 8, 4) weight, its element strides and the 12 288-float scratch");
    hipLaunchKernelGGL(k_conv1_prep, dim3(C1_KDIM * C1_OUT / 256), dim3(256), 0, (hipStream_t)stream, weight_dev, (long long)weight_strides[0], (long long)weight_strides[1],
                       (long long)weight_strides[2], (long long)weight_strides[3], reinterpret_cast<uint16_t *>(scratch_dev));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char buf[160]; snprintf(buf, sizeof buf, "grip_conv1_prep: %s", hipGetErrorString(e)); return grip_fail(buf); }
    return 0;
}
extern "C" int grip_conv1_u8_train(const uint8_t *obs_dev, const int64_t *row0_dev, const int64_t *rows_dev, int n, int channels, const float *weight_dev,
                                   const int64_t *weight_strides, const float *bias_dev, float *scratch_dev, float *out_nhwc_dev, float *other_dev, uint32_t *mask_dev, void *stream) {
    if (!obs_dev || (weight_dev && !weight_strides) || !bias_dev || !scratch_dev || !out_nhwc_dev || !other_dev || n <= 0 || channels != C1_IN + 1)
        return grip_fail("grip_conv1_u8: need uint8 [n, 5, 64, 64] observations, Conv2d(4, 32, 8, 4) weights and bias, a 12 288-float scratch and the two outputs");
    // weight_dev == NULL: scratch_dev already holds the weight's three bf16 terms (grip_conv1_prep, or an earlier call with the same weights): the rollouts split
    // the weights once per policy update, not once per tick
    if (weight_dev)
        hipLaunchKernelGGL(k_conv1_prep, dim3(C1_KDIM * C1_OUT / 256), dim3(256), 0, (hipStream_t)stream, weight_dev, (long long)weight_strides[0], (long long)weight_strides[1],
                           (long long)weight_strides[2], (long long)weight_strides[3], reinterpret_cast<uint16_t *>(scratch_dev));
    // 80 KB of LDS per workgroup: exactly two per CU, 512 on the chip. Workgroup i takes images i, i + grid, ...: with all 512 launched a CU's two (i, i + 256 under
    // round-robin placement) share the remainder evenly -- sizing the grid for equal counts per WORKGROUP left some CUs with one workgroup more than others
    // (12 % off the balanced time at 4096 images)
    const size_t lds = (size_t)(C1_IN * C1_HW * C1_HW + C1_BT_HALVES) * sizeof(uint16_t);
    {   static std::atomic<unsigned long long> attr_set_mask{0ULL};       // the dynamic-LDS opt-in is a per-DEVICE function attribute
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return grip_fail("grip_conv1_u8: no current device");
        const unsigned long long bit = 1ULL << (dev & 63);
        if (!(attr_set_mask.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void *)k_conv1_u8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return grip_fail("grip_conv1_u8: cannot reserve LDS");
            attr_set_mask.fetch_or(bit, std::memory_order_release);
        }
    }
    const int grid = n < 512 ? n : 512;
    hipLaunchKernelGGL(k_conv1_u8, dim3(grid), dim3(256), lds, (hipStream_t)stream, obs_dev, (const long long *)row0_dev, (const long long *)rows_dev, n, channels,
                       (const uint16_t *)scratch_dev, bias_dev, out_nhwc_dev, other_dev, mask_dev);
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

// HEADS (the update's explicit path, sb3/fused_update.py): mean / values are read from, and their gradients written in, the layout of the merged heads' last
// batch-of-two GEMM -- o [2, n, 8]: o[0, i, :A] = mean_i, o[1, i, 0] = value_i, go likewise with the padding written as zeros -- and g_head_bias [2, 8] receives go's
// column sums (the action / value heads' bias gradients), so that nothing elementwise is left between the loss and the backward GEMMs.
template <bool HEADS>
__global__ void __launch_bounds__(PL_THREADS) k_ppo_loss(const float *__restrict__ mean, const float *__restrict__ log_std, const float *__restrict__ values,
                                                        const float *__restrict__ actions, const float *__restrict__ old_logp, const float *__restrict__ adv,
                                                        const float *__restrict__ ret, int n, int A, float clip, float ent_coef, float vf_coef,
                                                        float *__restrict__ out, float *__restrict__ g_mean, float *__restrict__ g_values, float *__restrict__ g_log_std,
                                                        const float *__restrict__ head_bias, float *__restrict__ g_head_bias) {
    const int MS = HEADS ? PL_MAXA : A, VS = HEADS ? PL_MAXA : 1;             // row strides of mean / g_mean and of values / g_values
    __shared__ float red[PL_THREADS / 64];
    __shared__ float redm[PL_THREADS / 64][2 * PL_MAXA + 4];                  // the end's sums, all at once: one barrier instead of one pair per sum
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < n; i += PL_THREADS) s += adv[i];
    const float am = pl_block_sum(s, red) / (float)n;
    s = 0.f;
    for (int i = tid; i < n; i += PL_THREADS) { const float d = adv[i] - am; s += d * d; }
    const float asd = sqrtf(pl_block_sum(s, red) / (float)(n - 1));
    const float ainv = 1.0f / (asd + 1e-8f), invn = 1.0f / (float)n;
    float ls[PL_MAXA], isig[PL_MAXA], gls[PL_MAXA], gms[PL_MAXA], hb[PL_MAXA], lsum = 0.f;
#pragma unroll
    for (int k = 0; k < PL_MAXA; k++) {
        ls[k] = k < A ? log_std[k] : 0.f; isig[k] = expf(-ls[k]); gls[k] = 0.f; gms[k] = 0.f; lsum += k < A ? ls[k] : 0.f;
        hb[k] = (HEADS && k < A) ? head_bias[k] : 0.f;                        // HEADS: the heads' GEMM ran without its bias
    }
    const float hbv = HEADS ? head_bias[PL_MAXA] : 0.f;
    float sobj = 0.f, sv = 0.f, gvs = 0.f;
    for (int i = tid; i < n; i += PL_THREADS) {
        float z[PL_MAXA], lp = 0.f;
#pragma unroll
        for (int k = 0; k < PL_MAXA; k++) {
            const float mu = k < A ? (HEADS ? mean[(size_t)i * MS + k] + hb[k] : mean[(size_t)i * MS + k]) : 0.f;
            z[k] = k < A ? (actions[(size_t)i * A + k] - mu) * isig[k] : 0.f;
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
        if (HEADS) {                                                               // the padded rows leave as two 16-byte stores each
            float gm[PL_MAXA];
#pragma unroll
            for (int k = 0; k < PL_MAXA; k++) {
                gm[k] = k < A ? coef * z[k] * isig[k] : 0.f;
                gms[k] += gm[k];
                gls[k] += k < A ? coef * (z[k] * z[k] - 1.f) : 0.f;
            }
            float4 *gp = reinterpret_cast<float4 *>(g_mean + (size_t)i * PL_MAXA);
            gp[0] = make_float4(gm[0], gm[1], gm[2], gm[3]); gp[1] = make_float4(gm[4], gm[5], gm[6], gm[7]);
        } else {
#pragma unroll
            for (int k = 0; k < PL_MAXA; k++) if (k < A) {
                g_mean[(size_t)i * A + k] = coef * z[k] * isig[k];
                gls[k] += coef * (z[k] * z[k] - 1.f);
            }
        }
        const float dv = (HEADS ? values[(size_t)i * VS] + hbv : values[i]) - ret[i];
        sv += dv * dv;
        const float gv = vf_coef * 2.f * dv * invn;
        if (HEADS) {
            float4 *gp = reinterpret_cast<float4 *>(g_values + (size_t)i * PL_MAXA);
            gp[0] = make_float4(gv, 0.f, 0.f, 0.f); gp[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            gvs += gv;
        } else g_values[i] = gv;
    }
    // every sum of the end in one pass: wave sums by shuffles, one barrier, then thread k adds the 16 wave sums of value k in wave order (the order
    // pl_block_sum adds them in: the results are the bits the one-sum-at-a-time version produced)
    auto wsum = [](float v) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
        return v;
    };
    const int w = tid >> 6;
    const bool lead = (tid & 63) == 0;
    { const float v = wsum(sobj); if (lead) redm[w][0] = v; }
    { const float v = wsum(sv); if (lead) redm[w][1] = v; }
    { const float v = wsum(gvs); if (lead) redm[w][2] = v; }
#pragma unroll
    for (int k = 0; k < PL_MAXA; k++) {
        { const float v = wsum(gls[k]); if (lead) redm[w][4 + k] = v; }
        if (HEADS) { const float v = wsum(gms[k]); if (lead) redm[w][4 + PL_MAXA + k] = v; }
    }
    __syncthreads();
    float tot = 0.f;
    if (tid < 2 * PL_MAXA + 4) {
#pragma unroll
        for (int i = 0; i < PL_THREADS / 64; i++) tot += redm[i][tid];
    }
    if (tid >= 4 && tid < 4 + A) g_log_std[tid - 4] = tot - ent_coef;
    if (HEADS) {
        if (tid >= 4 + PL_MAXA && tid < 4 + 2 * PL_MAXA) { const int k = tid - 4 - PL_MAXA; g_head_bias[k] = k < A ? tot : 0.f; if (k > 0) g_head_bias[PL_MAXA + k] = 0.f; }
        if (tid == 2) g_head_bias[PL_MAXA] = tot;
    }
    __syncthreads();
    if (tid == 0) {
        float so = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < PL_THREADS / 64; i++) { so += redm[i][0]; sq += redm[i][1]; }
        const float pl = -so * invn, vl = sq * invn;
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
    hipLaunchKernelGGL(k_ppo_loss<false>, dim3(1), dim3(PL_THREADS), 0, (hipStream_t)stream, mean_dev, log_std_dev, values_dev, actions_dev, old_log_prob_dev, advantages_dev,
                       returns_dev, n, action_dim, clip_range, ent_coef, vf_coef, out_dev, grad_mean_dev, grad_values_dev, grad_log_std_dev, (const float *)nullptr, (float *)nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char buf[160]; snprintf(buf, sizeof buf, "grip_ppo_loss: %s", hipGetErrorString(e)); return grip_fail(buf); }
    return 0;
}

// The HEADS loss over several workgroups (the one-workgroup kernel above is 31 us for 4096 rows: four dependent passes of a single workgroup, latency all the way). Every
// workgroup forms the advantages' mean and deviation itself, over all n rows and in the same order (16 KB from L2: cheaper than a second launch or a grid barrier), takes its share
// of the rows, leaves its partial sums in `part`, and the workgroup that arrives last adds them in workgroup order (fixed order: the result does not depend on timing) and resets
// the ticket. One launch of the loss at a time per device (the ticket is a module variable): the update's single stream.
#define PLM_THREADS 256
#define PLM_BLOCKS 16
#define PLM_SUMS (2 * PL_MAXA + 4)
// (the arrival counter of the last-workgroup reduction is a word of the CALL's scratch, zeroed by the gather launch in front of it on the same stream: two loss
// launches in flight on one device, or one that died half-way, cannot disturb each other)
__device__ __forceinline__ float plm_block_sum(float v, float *red) {           // 4 waves
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void __launch_bounds__(PLM_THREADS) k_ppo_loss_heads_mb(const float *__restrict__ o, const float *__restrict__ log_std, const float *__restrict__ actions,
                                                                   const float *__restrict__ old_logp, const float *__restrict__ adv, const float *__restrict__ ret, int n, int A,
                                                                   float clip, float ent_coef, float vf_coef, float *__restrict__ out, float *__restrict__ go,
                                                                   float *__restrict__ g_log_std, const float *__restrict__ head_bias, float *__restrict__ g_head_bias,
                                                                   float *__restrict__ part, unsigned *__restrict__ ticket) {
    __shared__ float red[PLM_THREADS / 64];
    __shared__ float redm[PLM_THREADS / 64][PLM_SUMS];
    __shared__ unsigned last;
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < n; i += PLM_THREADS) s += adv[i];
    const float am = plm_block_sum(s, red) / (float)n;
    s = 0.f;
    for (int i = tid; i < n; i += PLM_THREADS) { const float d = adv[i] - am; s += d * d; }
    const float asd = sqrtf(plm_block_sum(s, red) / (float)(n - 1));
    const float ainv = 1.0f / (asd + 1e-8f), invn = 1.0f / (float)n;
    float ls[PL_MAXA], isig[PL_MAXA], gls[PL_MAXA], gms[PL_MAXA], hb[PL_MAXA], lsum = 0.f;
#pragma unroll
    for (int k = 0; k < PL_MAXA; k++) {
        ls[k] = k < A ? log_std[k] : 0.f; isig[k] = expf(-ls[k]); gls[k] = 0.f; gms[k] = 0.f; lsum += k < A ? ls[k] : 0.f;
        hb[k] = k < A ? head_bias[k] : 0.f;
    }
    const float hbv = head_bias[PL_MAXA];
    const float *values = o + (size_t)n * PL_MAXA;
    float *g_values = go + (size_t)n * PL_MAXA;
    float sobj = 0.f, sv = 0.f, gvs = 0.f;
    const int per = (n + PLM_BLOCKS - 1) / PLM_BLOCKS, i0 = blockIdx.x * per, i1 = min(n, i0 + per);
    for (int i = i0 + tid; i < i1; i += PLM_THREADS) {
        float z[PL_MAXA], lp = 0.f;
#pragma unroll
        for (int k = 0; k < PL_MAXA; k++) {
            const float mu = k < A ? o[(size_t)i * PL_MAXA + k] + hb[k] : 0.f;
            z[k] = k < A ? (actions[(size_t)i * A + k] - mu) * isig[k] : 0.f;
            lp += k < A ? -0.5f * z[k] * z[k] - ls[k] - 0.9189385332046727f : 0.f;
        }
        const float lr_raw = lp - old_logp[i];
        const float lr = fminf(fmaxf(lr_raw, -20.f), 20.f);
        const float ratio = expf(lr), Ai = (adv[i] - am) * ainv;
        const float lo = 1.f - clip, hi = 1.f + clip;
        const float s1 = Ai * ratio, s2 = Ai * fminf(fmaxf(ratio, lo), hi);
        sobj += fminf(s1, s2);
        const bool inside = ratio >= lo && ratio <= hi;
        const float w1 = s1 < s2 ? 1.f : (s1 == s2 ? 0.5f : 0.f), w2 = (s2 < s1 ? 1.f : (s1 == s2 ? 0.5f : 0.f)) * (inside ? 1.f : 0.f);
        const float pass = (lr_raw >= -20.f && lr_raw <= 20.f) ? 1.f : 0.f;
        const float coef = -invn * Ai * (w1 + w2) * ratio * pass;
        float gm[PL_MAXA];
#pragma unroll
        for (int k = 0; k < PL_MAXA; k++) {
            gm[k] = k < A ? coef * z[k] * isig[k] : 0.f;
            gms[k] += gm[k];
            gls[k] += k < A ? coef * (z[k] * z[k] - 1.f) : 0.f;
        }
        float4 *gp = reinterpret_cast<float4 *>(go + (size_t)i * PL_MAXA);
        gp[0] = make_float4(gm[0], gm[1], gm[2], gm[3]); gp[1] = make_float4(gm[4], gm[5], gm[6], gm[7]);
        const float dv = values[(size_t)i * PL_MAXA] + hbv - ret[i];
        sv += dv * dv;
        const float gv = vf_coef * 2.f * dv * invn;
        float4 *gq = reinterpret_cast<float4 *>(g_values + (size_t)i * PL_MAXA);
        gq[0] = make_float4(gv, 0.f, 0.f, 0.f); gq[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        gvs += gv;
    }
    auto wsum = [](float v) {
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
        return v;
    };
    const int w = tid >> 6;
    const bool lead = (tid & 63) == 0;
    { const float v = wsum(sobj); if (lead) redm[w][0] = v; }
    { const float v = wsum(sv); if (lead) redm[w][1] = v; }
    { const float v = wsum(gvs); if (lead) redm[w][2] = v; }
#pragma unroll
    for (int k = 0; k < PL_MAXA; k++) {
        { const float v = wsum(gls[k]); if (lead) redm[w][4 + k] = v; }
        { const float v = wsum(gms[k]); if (lead) redm[w][4 + PL_MAXA + k] = v; }
    }
    __syncthreads();
    if (tid < PLM_SUMS) part[blockIdx.x * PLM_SUMS + tid] = (redm[0][tid] + redm[1][tid]) + (redm[2][tid] + redm[3][tid]);
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(ticket, 1u);
    __syncthreads();
    if (last != PLM_BLOCKS - 1) return;
    __threadfence();
    float tot = 0.f;
    if (tid < PLM_SUMS) {
#pragma unroll
        for (int b = 0; b < PLM_BLOCKS; b++) tot += __builtin_nontemporal_load(part + b * PLM_SUMS + tid);
    }
    if (tid >= 4 && tid < 4 + A) g_log_std[tid - 4] = tot - ent_coef;
    if (tid >= 4 + PL_MAXA && tid < 4 + 2 * PL_MAXA) { const int k = tid - 4 - PL_MAXA; g_head_bias[k] = k < A ? tot : 0.f; if (k > 0) g_head_bias[PL_MAXA + k] = 0.f; }
    if (tid == 2) g_head_bias[PL_MAXA] = tot;
    __shared__ float fin[2];
    if (tid < 2) fin[tid] = tot;
    __syncthreads();
    if (tid == 0) {
        const float pl = -fin[0] * invn, vl = fin[1] * invn;
        const float el = -((float)A * (0.5f + 0.9189385332046727f) + lsum);
        out[0] = pl + ent_coef * el + vf_coef * vl; out[1] = pl; out[2] = vl;
    }
}

// rows idx[0..n) of the rollout's sample arrays into one packed block: samples = actions [n, A] | old_log_prob [n] | advantages [n] | returns [n]  (one launch for
// the four gathers of a minibatch)
__global__ void __launch_bounds__(256) k_gather_samples(const float *__restrict__ actions, const float *__restrict__ logp, const float *__restrict__ adv,
                                                       const float *__restrict__ ret, const int64_t *__restrict__ idx, int n, int A, float *__restrict__ out, unsigned *__restrict__ ticket) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && ticket) *ticket = 0u;              // the loss kernel's arrival counter (per call)
    if (i >= n) return;
    const int64_t r = idx[i];
    for (int k = 0; k < A; k++) out[(size_t)i * A + k] = actions[r * A + k];
    out[(size_t)n * A + i] = logp[r]; out[(size_t)n * (A + 1) + i] = adv[r]; out[(size_t)n * (A + 2) + i] = ret[r];
}

extern "C" int grip_ppo_loss_heads(const float *heads_out_dev, const float *head_bias_dev, const float *log_std_dev, const float *actions_dev, const float *old_log_prob_dev,
                                   const float *advantages_dev, const float *returns_dev, const int64_t *rows_dev, int n, int action_dim, float clip_range,
                                   float ent_coef, float vf_coef, float *samples_dev, float *out_dev, float *grad_heads_out_dev, float *grad_head_bias_dev,
                                   float *grad_log_std_dev, void *stream) {
    if (!heads_out_dev || !head_bias_dev || !log_std_dev || !actions_dev || !old_log_prob_dev || !advantages_dev || !returns_dev || !rows_dev || !samples_dev || !out_dev ||
        !grad_heads_out_dev || !grad_head_bias_dev || !grad_log_std_dev || n < 2 || action_dim < 1 || action_dim > PL_MAXA)
        return grip_fail("grip_ppo_loss_heads: need n >= 2 rows, 1..8 action dimensions and every array");
    const int A = action_dim;
    hipLaunchKernelGGL(k_gather_samples, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, actions_dev, old_log_prob_dev, advantages_dev, returns_dev, rows_dev, n, A,
                       samples_dev, reinterpret_cast<unsigned *>(samples_dev + (size_t)n * (A + 3) + PLM_BLOCKS * PLM_SUMS));
    float *sa = samples_dev, *sl = samples_dev + (size_t)n * A, *sd = sl + n, *sr = sd + n;
#ifdef PLM_OFF            // comparison build: the one-workgroup kernel
    hipLaunchKernelGGL(k_ppo_loss<true>, dim3(1), dim3(PL_THREADS), 0, (hipStream_t)stream, heads_out_dev, log_std_dev, heads_out_dev + (size_t)n * PL_MAXA, sa, sl, sd, sr, n, A,
                       clip_range, ent_coef, vf_coef, out_dev, grad_heads_out_dev, grad_heads_out_dev + (size_t)n * PL_MAXA, grad_log_std_dev, head_bias_dev, grad_head_bias_dev);
#else
    // (samples_dev is followed by the workgroups' partial sums and the arrival counter: n * (A + 3) + 16 * 20 + 1 words in all)
    hipLaunchKernelGGL(k_ppo_loss_heads_mb, dim3(PLM_BLOCKS), dim3(PLM_THREADS), 0, (hipStream_t)stream, heads_out_dev, log_std_dev, (const float *)sa, (const float *)sl, (const float *)sd,
                       (const float *)sr, n, A, clip_range, ent_coef, vf_coef, out_dev, grad_heads_out_dev, grad_log_std_dev, head_bias_dev, grad_head_bias_dev, sr + n, reinterpret_cast<unsigned *>(sr + n + PLM_BLOCKS * PLM_SUMS));
#endif
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char buf[160]; snprintf(buf, sizeof buf, "grip_ppo_loss_heads: %s", hipGetErrorString(e)); return grip_fail(buf); }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Second and third layers of AugmentedNatureCNN (models/feature_extractor.py:17-21) in one launch:
//   y2 = relu(conv2d(y1, w2, b2, stride 2)),  y1 [n, 15, 15, 32] NHWC (what k_conv1_u8 writes), w2 [64, 32, 4, 4]  -> [n, 6, 6, 64]
//   y3 = relu(conv2d(y2, w3, b3, stride 1)),                                               w3 [64, 64, 3, 3]  -> [n, 4, 4, 64] NHWC
// as two implicit GEMMs on v_mfma_f32_16x16x4_f32 (exact f32 products, f32 accumulation), y2 staying in LDS between them. As tensor-library calls
// the pair is two implicit-GEMM launches plus bias and ReLU passes. A workgroup (4 waves) per C2_G = 2 images, two workgroups per CU (one loads
// while the other's MFMAs run) -- GEMM 1: M = 72 (image, oy, ox) rows = 4.5 tiles of 16, N = 64, K = 4 x 4 x 32 = 512; GEMM 2: M = 32 = 2 tiles (one per image),
// N = 64, K = 3 x 3 x 64 = 576. Wave w owns output channels 16 w .. 16 w + 15 and every M tile. Operand traffic is what an fp32 MFMA kernel is priced by
// (every LDS / VALU instruction next to an MFMA takes issue cycles of the same SIMD: tools/hiptests/t_mfma_peak.hip), so the reduction index is permuted:
// within a (ky, kx) block the k-step j of an instruction pairs input channels ci = C (l >> 4) + j (C = 8 or 16 per lane) -- a lane's A values of a block
// are then contiguous in LDS (b128 reads, one per four MFMAs) and its B values contiguous in the weights stored as Bt[co][k] (two or four 16-byte loads per
// block from L2, requested one block ahead). The first version read one dword per MFMA for each operand: 197 us per 4096 images against 92 of MFMA time.
// Training (y2_out != NULL): also y2 and both layers' ReLU masks as bits.
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define C2_G 2
#define C2_PS1 36                       // floats per y1 pixel in LDS (32 channels + 4: 16-byte aligned rows)
#define C2_PS2 68                       // floats per y2 pixel in LDS (64 channels + 4)
// The padding does NOT make the A reads conflict-free (measured 61 % of the LDS-array cycles are conflict cycles, profiles/r03_update_kernels_pmc.txt), and no padding
// can: a ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md, LDS), a group wants 16 distinct 16-byte slots of the 256-byte bank
// row, and a lane's slot is (pixel * PS1 / 4 + 2 kq) mod 16 in GEMM 1 -- every pixel of a tile row is EVEN (30 oy + 2 ox) and kq steps by two slots, so a group lands on at
// most 8 slots whatever PS1 is, and rows r16 = 0 / 14 (pixels 0 / 64) collide for every PS1 (GEMM 2: positions 18..21 wrap onto 2..5). Getting out needs the k chunks of
// the four lane groups one slot apart (channels 4 kq + 16 q instead of 8 kq + 4 q, A and B alike) AND another assignment of a tile's 16 rows to lanes; the prize is bounded by
// the kernel's MFMA-busy share (99 of 125 us): at most a fifth of 69 us per tick and 147 us per minibatch. Costed, not built this round.
#define C2_LDS_FLOATS (C2_G * 225 * C2_PS1)         // y1 of the group; y2 (C2_G * 36 * C2_PS2 floats) reuses the space once GEMM 1 is done

// Both layouts of both weight matrices, k = (ky, kx, ci): B2[k][n] (512 x 64, what the backward's scatter GEMMs read) then B2t[n][k] (64 x 512, the forward's);
// B3[k][n] (576 x 64) then B3t[n][k] (64 x 576)  (element strides of the weight tensors given)
__global__ void __launch_bounds__(256) k_conv23_prep(const float *__restrict__ w2, long long s2o, long long s2c, long long s2y, long long s2x,
                                                     const float *__restrict__ w3, long long s3o, long long s3c, long long s3y, long long s3x,
                                                     float *__restrict__ B2, float *__restrict__ B3) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    // ... and, behind the two fp32 layouts of each matrix, the operand FRAGMENTS of the bf16 kernel (k_conv23_b3 below): every weight as three bf16 terms (nearest,
    // exact remainder, twice), eight consecutive input channels of one tap per lane, in the order v_mfma_f32_16x16x32_bf16 wants its B operand -- lane l of wave w
    // holds output channel 16 w + (l & 15), channels 8 (l >> 4) .. + 7 of the k-group -- so that a fragment is ONE coalesced 16-byte load per lane:
    //   B2f [wave 4][tap 16][term 3][lane 64][8], B3f [wave 4][k-group 18 = tap x {channels 0..31, 32..63}][term 3][lane 64][8]
    if (i < 4 * 16 * 64 + 4 * 18 * 64) {
        const bool second = i >= 4 * 16 * 64;
        const int q = second ? i - 4 * 16 * 64 : i, lane = q & 63, grp = q >> 6, ngrp = second ? 18 : 16, w = grp / ngrp, kg = grp - w * ngrp;
        const int n = 16 * w + (lane & 15), c0 = 8 * (lane >> 4) + (second ? 32 * (kg & 1) : 0), tap = second ? kg >> 1 : kg;
        const int ky = second ? tap / 3 : tap >> 2, kx = second ? tap - 3 * (tap / 3) : tap & 3;
        uint16_t *dst = reinterpret_cast<uint16_t *>(second ? B3 + 2 * 576 * 64 : B2 + 2 * 512 * 64) + ((size_t)grp * 3 * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float r = second ? w3[n * s3o + (c0 + j) * s3c + ky * s3y + kx * s3x] : w2[n * s2o + (c0 + j) * s2c + ky * s2y + kx * s2x];
#pragma unroll
            for (int t = 0; t < 3; t++) { const unsigned h = bf16_rne_bits(r); dst[t * 64 * 8 + j] = (uint16_t)h; r -= __uint_as_float(h << 16); }
        }
    }
    // ... and the fragments of the TRANSPOSED products the backward's data-gradient GEMMs take (k_trunk_bwd_b3, grip_train.hip: reduction over the output channels):
    //   B2d [tap 16][input-channel half 2][k-step 2][term 3][lane 64][8] behind B2f, B3d [tap 9][input-channel quarter 4][k-step 2][term 3][lane 64][8] behind B3f;
    //   lane l holds input channel 16 block + (l & 15), output channels 32 k-step + 8 (l >> 4) .. + 7
    if (i < (16 * 2 * 2 + 9 * 4 * 2) * 64) {
        const bool second = i >= 16 * 2 * 2 * 64;
        const int q = second ? i - 16 * 2 * 2 * 64 : i, lane = q & 63, gd = q >> 6, ks = gd & 1, blk = second ? (gd >> 1) & 3 : (gd >> 1) & 1, tap = second ? gd >> 3 : gd >> 2;
        const int ci = 16 * blk + (lane & 15), o0 = 32 * ks + 8 * (lane >> 4);
        const int ky = second ? tap / 3 : tap >> 2, kx = second ? tap - 3 * (tap / 3) : tap & 3;
        uint16_t *dst = reinterpret_cast<uint16_t *>(second ? B3 + (2 * 576 + 864) * 64 : B2 + (2 * 512 + 768) * 64) + ((size_t)gd * 3 * 64 + lane) * 8;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float r = second ? w3[(o0 + j) * s3o + ci * s3c + ky * s3y + kx * s3x] : w2[(o0 + j) * s2o + ci * s2c + ky * s2y + kx * s2x];
#pragma unroll
            for (int t = 0; t < 3; t++) { const unsigned h = bf16_rne_bits(r); dst[t * 64 * 8 + j] = (uint16_t)h; r -= __uint_as_float(h << 16); }
        }
    }
    if (i < 512 * 64) {
        const int n = i & 63, k = i >> 6, ci = k & 31, kx = (k >> 5) & 3, ky = k >> 7;
        const float v = w2[n * s2o + ci * s2c + ky * s2y + kx * s2x];
        B2[i] = v; B2[512 * 64 + n * 512 + k] = v;
    }
    if (i < 576 * 64) {
        const int n = i & 63, k = i >> 6, ci = k & 63, kk = k >> 6, kx = kk % 3, ky = kk / 3;
        const float v = w3[n * s3o + ci * s3c + ky * s3y + kx * s3x];
        B3[i] = v; B3[576 * 64 + n * 576 + k] = v;
    }
}

// one implicit GEMM of the pair: MT row tiles, NBLK (ky, kx) blocks of 4 Q k-steps; the lane's A values of tile t and block c start at lds[abase[t] + aoff(c)]
// (4 Q contiguous floats), its B values at Bt[c * 16 Q] (4 Q contiguous floats); quarter-blocks of four k-steps, the next quarter's A read while this one's MFMAs issue
template <int MT, int NBLK, int Q, class AOff>
__device__ __forceinline__ void conv_gemm(const float *lds, const int (&abase)[MT], const float *__restrict__ Bt, f32x4 (&acc)[MT], AOff aoff) {
    float4 bcur[Q], bnext[Q], acur[MT], an[MT];
#pragma unroll
    for (int q = 0; q < Q; q++) bcur[q] = *reinterpret_cast<const float4 *>(Bt + 4 * q);
#pragma unroll
    for (int t = 0; t < MT; t++) acur[t] = *reinterpret_cast<const float4 *>(lds + abase[t] + aoff(0));
#pragma unroll
    for (int c = 0; c < NBLK; c++) {
        const int cn = c + 1 < NBLK ? c + 1 : c;
#pragma unroll
        for (int q = 0; q < Q; q++) bnext[q] = *reinterpret_cast<const float4 *>(Bt + cn * 16 * Q + 4 * q);      // a block = 4 lane groups x 4 Q channels
#pragma unroll
        for (int q = 0; q < Q; q++) {
#pragma unroll
            for (int t = 0; t < MT; t++) an[t] = *reinterpret_cast<const float4 *>(lds + abase[t] + (q + 1 < Q ? aoff(c) + 4 * (q + 1) : aoff(cn)));
            __builtin_amdgcn_sched_barrier(0);                       // keep the requests above the MFMAs (the scheduler sinks them to their uses)
#pragma unroll
            for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[t].x, bcur[q].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[t].y, bcur[q].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[t].z, bcur[q].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[t].w, bcur[q].w, acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < MT; t++) acur[t] = an[t];
        }
#pragma unroll
        for (int q = 0; q < Q; q++) bcur[q] = bnext[q];
    }
}

__global__ void __launch_bounds__(256, 2) k_conv23(const float *__restrict__ y1, int n_img, const float *__restrict__ B2t, const float *__restrict__ bias2,
                                                   const float *__restrict__ B3t, const float *__restrict__ bias3, float *__restrict__ out, float *__restrict__ y2_out,
                                                   uint16_t *__restrict__ mask2, uint16_t *__restrict__ mask3) {
    extern __shared__ __attribute__((aligned(16))) float c2_lds[];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r16 = l & 15, kq = l >> 4;
    const int img0 = blockIdx.x * C2_G, nimg = min(C2_G, n_img - img0);
    // y1 of the group -> LDS (a missing second image as zeros); eight 16-byte loads in flight per thread, then their stores
    for (int b0 = 0; b0 < C2_G * 1800; b0 += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = b0 + u * 256 + tid;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < nimg * 1800) v[u] = *reinterpret_cast<const float4 *>(y1 + (size_t)img0 * 7200 + (size_t)i * 4);
        }
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int i = b0 + u * 256 + tid;
            if (i < C2_G * 1800) *reinterpret_cast<float4 *>(c2_lds + (i >> 3) * C2_PS1 + (i & 7) * 4) = v[u];
        }
    }
    __syncthreads();
    const int ncol = 16 * w + r16;
    // ---- GEMM 1: rows m = image * 36 + position (72 = 4.5 tiles: the last half tile is computed from a clamped row and dropped)
    f32x4 acc[5];
    {
        int abase[5];
#pragma unroll
        for (int t = 0; t < 5; t++) {
            const int m = min(t * 16 + r16, C2_G * 36 - 1), g = m / 36, p = m - g * 36, oy = p / 6, ox = p - oy * 6;
            abase[t] = (g * 225 + 2 * oy * 15 + 2 * ox) * C2_PS1 + 8 * kq;
            acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        conv_gemm<5, 16, 2>(c2_lds, abase, B2t + (size_t)ncol * 512 + 8 * kq, acc, [](int c) { return ((c >> 2) * 15 + (c & 3)) * C2_PS1; });
    }
    __syncthreads();                                                 // every wave is done reading y1
    {   // y2 = relu(acc + bias) into LDS: lane holds rows 4 kq + r of tile t, column ncol
        const float bn = bias2[ncol];
#pragma unroll
        for (int t = 0; t < 5; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = t * 16 + 4 * kq + r;                    // = g * 36 + position
                const float v = fmaxf(acc[t][r] + bn, 0.f);
                if (m < C2_G * 36) c2_lds[m * C2_PS2 + ncol] = v;
                if (y2_out) {                                         // training: the activation for the weight gradient, and its ReLU mask -- bit c of the 64-bit word
                    const unsigned long long bal = __ballot(v > 0.f);  // (image, position) = channel c is active; this wave's 16 channels are one uint16 of it
                    if (m < nimg * 36) {
                        y2_out[((size_t)img0 * 36 + m) * 64 + ncol] = v;
                        if (r16 == 0) mask2[((size_t)img0 * 36 + m) * 4 + w] = (uint16_t)(bal >> (16 * kq));
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- GEMM 2: tile g = image g, row r16 = (oy, ox) of the 4 x 4 output
    f32x4 acc3[C2_G];
    {
        int a3[C2_G];
#pragma unroll
        for (int g = 0; g < C2_G; g++) {
            a3[g] = (g * 36 + (r16 >> 2) * 6 + (r16 & 3)) * C2_PS2 + 16 * kq;
            acc3[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        conv_gemm<C2_G, 9, 4>(c2_lds, a3, B3t + (size_t)ncol * 576 + 16 * kq, acc3, [](int c) { return ((c / 3) * 6 + (c % 3)) * C2_PS2; });
    }
    {
        const float bn = bias3[ncol];
#pragma unroll
        for (int g = 0; g < C2_G; g++) {
            if (g < nimg) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = fmaxf(acc3[g][r] + bn, 0.f);
                    out[((size_t)(img0 + g) * 16 + 4 * kq + r) * 64 + ncol] = v;
                    if (mask3) {
                        const unsigned long long bal = __ballot(v > 0.f);
                        if (r16 == 0) mask3[((size_t)(img0 + g) * 16 + 4 * kq + r) * 4 + w] = (uint16_t)(bal >> (16 * kq));
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same two layers on the bf16 matrix pipe, in fp32-equivalent arithmetic (round 5). An fp32 number is exactly the sum of three bf16 terms (nearest, exact
// remainder, twice: what is left is below its 25th bit); of the nine cross products of two such sums the six of weight >= 2^-16 are kept -- a1 b1, a1 b2, a2 b1, a1 b3,
// a2 b2, a3 b1 -- each a product of two 8-bit significands, EXACT in the fp32 accumulator of v_mfma_f32_16x16x32_bf16: six bf16 instructions (16 cycles for 32 k each)
// do the work of eight fp32 ones (32 cycles for 4 k each), 2.7 x less matrix-pipe time; the dropped terms are <= 2^-24 of a product.
// What that needs to pay: the operands split ONCE (the weights by k_conv23_prep, an activation when it is staged into LDS, as three bf16 planes), and the weights of
// the third layer -- two 16-row tiles per image pair give a fragment 12 instructions of work -- RESIDENT IN REGISTERS: its 18 k-groups x 3 terms x 4 registers = 216 per
// lane (CB_B3_RES; fewer would stream the rest from L2 two k-groups ahead), loaded once by a persistent workgroup (one per CU, four waves = one per SIMD, up to 512 registers each); the second layer's fragments (30 instructions of work
// each: five tiles) stream from L2 one tap ahead, 196 KB per image pair. The next pair's y1 is requested into registers before the MFMA loops and split after them.
// LDS: y1 as [term 3][image 2][pixel 225][32 channels] bf16 with 80-byte pixels, y2 as [term 3][row 72][64 channels] with 160-byte rows (with the row orders of
// CB_ROWS2 / CB_ROWS3 below the A operand's reads are free of bank conflicts): 139 KB.
#define CB_G 2
#define CB_PS1 80
#define CB_PL1 (CB_G * 225 * CB_PS1)
#define CB_PS2 160
#define CB_PL2 (CB_G * 36 * CB_PS2)
#define CB_LDS_BYTES (3 * CB_PL1 + 3 * CB_PL2)
#define CB_B3_RES 18             // k-groups of the third layer whose weight fragments stay in (accumulation) registers; the other 18 - CB_B3_RES stream from L2, two k-groups ahead

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
// Which GEMM row a lane's tile row is -- chosen so that the A operand's ds_read_b128 is free of bank conflicts. The LDS serves a b128 read in four groups of 16 lanes
// ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md), a group wants 16 distinct 16-byte slots of the 256-byte bank row, and a lane's slot is (5 pixel + kq) mod 16 with 80-byte
// pixels: with rows in their natural order (tile t = rows 16 t ...) a read took 8-10 LDS cycles instead of 4 and the four waves' reads, not the matrix pipe, set the
// pace. The order of a GEMM's rows is free: GEMM 1's tiles 0..3 hold 16 positions of ONE image each, every residue class of (3 oy + 5 ox) mod 8 once in lanes
// {0-3, 12-15} and once in lanes {4-11} (4 LDS cycles); tile 4 holds the eight rows left over (four per image) and duplicates of them (6 cycles; CB_REAL5: which lanes
// are real). GEMM 2's 32 rows (image g, position p = 16 g + p) with 160-byte rows are spread over its two tiles likewise (4 cycles each). tools: /tmp search, the tables.
__device__ const unsigned char CB_ROWS2[5][16] = {{0, 5, 2, 6, 7, 18, 9, 13, 11, 8, 19, 10, 4, 1, 12, 3}, {14, 25, 16, 20, 21, 32, 23, 27, 31, 22, 33, 30, 24, 15, 26, 17},
                                                  {36, 41, 38, 42, 43, 54, 45, 49, 47, 44, 55, 46, 40, 37, 48, 39}, {50, 61, 52, 56, 57, 68, 59, 63, 67, 58, 69, 66, 60, 51, 62, 53},
                                                  {71, 71, 71, 71, 64, 64, 64, 35, 29, 64, 65, 64, 70, 34, 71, 28}};
#define CB_REAL5 0xb591u            // bit i: tile 4's row i is a row of its own (lanes 0, 4, 7, 8, 10, 12, 13, 15), not a duplicate
__device__ const unsigned char CB_ROWS3[2][16] = {{11, 31, 22, 15, 9, 20, 19, 8, 1, 24, 4, 27, 6, 12, 21, 10}, {3, 14, 26, 25, 18, 0, 17, 13, 2, 29, 16, 7, 23, 5, 30, 28}};
__device__ __forceinline__ bf16x8 cb_bf(const u32x4 &v) { return __builtin_bit_cast(bf16x8, v); }
// the six products of one (row tile, k-group): smallest first
__device__ __forceinline__ f32x4 cb_mac6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb_bf(a[2]), cb_bf(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb_bf(a[1]), cb_bf(b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb_bf(a[0]), cb_bf(b[2]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb_bf(a[1]), cb_bf(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb_bf(a[0]), cb_bf(b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb_bf(a[0]), cb_bf(b[0]), c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ void cb_split4(const float4 v, bf16x4 &h, bf16x4 &m, bf16x4 &l) {
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const __bf16 a = (__bf16)x[i]; const float r = x[i] - (float)a;
        const __bf16 b = (__bf16)r; const float r2 = r - (float)b;
        h[i] = a; m[i] = b; l[i] = (__bf16)r2;
    }
}
#ifdef CB_STAMPS                 // diagnostic build only (tools/conv23_stamps.py): cycles of wave 0 of workgroup 0 per phase
__device__ unsigned long long g_cb_stamps[12];     // 0-7 phases, 8 the prologue; 9 / 10 / 11: latest workgroup start, earliest and latest workgroup end after the first start (100 MHz ticks, last launch)
__device__ unsigned long long g_cb_wg[512][2];
#define CB_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); if (blockIdx.x == 0 && threadIdx.x == 0) g_cb_stamps[i] += n_ - cb_t; cb_t = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int grip_debug_cb_stamps(unsigned long long *out8) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    static unsigned long long wg[512][2];
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_cb_stamps), sizeof(unsigned long long) * 9) != hipSuccess || hipMemcpyFromSymbol(wg, HIP_SYMBOL(g_cb_wg), sizeof wg) != hipSuccess) return -1;
    unsigned long long s0 = ~0ull, s1 = 0, e0 = ~0ull, e1 = 0;
    for (int i = 0; i < 512; i++) if (wg[i][0]) { s0 = wg[i][0] < s0 ? wg[i][0] : s0; s1 = wg[i][0] > s1 ? wg[i][0] : s1; e0 = wg[i][1] < e0 ? wg[i][1] : e0; e1 = wg[i][1] > e1 ? wg[i][1] : e1; }
    out8[9] = s1 - s0; out8[10] = e0 - s0; out8[11] = e1 - s0;
    static const unsigned long long zero[12] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_cb_stamps), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#else
#define CB_STAMP(i) do { } while (0)
#endif
__global__ void __launch_bounds__(256, 1) k_conv23_b3(const float *__restrict__ y1, int n_img, const u32x4 *__restrict__ B2f, const float *__restrict__ bias2,
                                                      const u32x4 *__restrict__ B3f, const float *__restrict__ bias3, float *__restrict__ out, float *__restrict__ y2_out,
                                                      uint16_t *__restrict__ mask2, uint16_t *__restrict__ mask3) {
    extern __shared__ __attribute__((aligned(16))) unsigned char cb_lds[];
#ifdef CB_STAMPS
    unsigned long long cb_t = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x < 512) g_cb_wg[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r16 = l & 15, kq = l >> 4;
    const int ncol = 16 * w + r16, npair = (n_img + CB_G - 1) / CB_G;
    const float bn2 = bias2[ncol], bn3 = bias3[ncol];
    // the third layer's weights for this wave's 16 channels, for the whole launch
    // (held in the ACCUMULATION half of the register file -- a wave's 512 registers are 256 v + 256 a, and left to itself the compiler keeps this array in v, spills
    // 150 registers to scratch and reloads them per k-group: written there once with v_accvgpr_write, read back four words ahead of the MFMAs that use them)
    unsigned b3a[CB_B3_RES][3][4];
    // (six k-groups = 18 loads in flight at a time: one load, its wait and its four register moves at a time were 54 round trips to L2, 34 k cycles = 14 us per launch)
#pragma unroll
    for (int k0 = 0; k0 < CB_B3_RES; k0 += 6) {
        u32x4 v[6][3];
#pragma unroll
        for (int kg = k0; kg < k0 + 6 && kg < CB_B3_RES; kg++)
#pragma unroll
            for (int t = 0; t < 3; t++) v[kg - k0][t] = B3f[((size_t)(w * 18 + kg) * 3 + t) * 64 + l];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kg = k0; kg < k0 + 6 && kg < CB_B3_RES; kg++)
#pragma unroll
            for (int t = 0; t < 3; t++) {
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(b3a[kg][t][0]) : "v"(v[kg - k0][t].x));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(b3a[kg][t][1]) : "v"(v[kg - k0][t].y));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(b3a[kg][t][2]) : "v"(v[kg - k0][t].z));
                asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(b3a[kg][t][3]) : "v"(v[kg - k0][t].w));
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    // row addresses of the two GEMMs' A fragments (bytes into a term plane)
    int a2[5], a3[CB_G];
#pragma unroll
    for (int t = 0; t < 5; t++) {
        const int m = CB_ROWS2[t][r16], g = m / 36, p = m - g * 36, oy = p / 6, ox = p - oy * 6;
        a2[t] = (g * 225 + 2 * oy * 15 + 2 * ox) * CB_PS1 + 16 * kq;
    }
#pragma unroll
    for (int g = 0; g < CB_G; g++) { const int m = CB_ROWS3[g][r16], gi = m >> 4, p = m & 15; a3[g] = 3 * CB_PL1 + (gi * 36 + (p >> 2) * 6 + (p & 3)) * CB_PS2 + 16 * kq; }
    // the GEMM rows of this lane's accumulator entries (tile row 4 kq + r, r = 0..3), four to a register: loop-invariant, seven registers -- looked up per trip they were
    // 28 dependent byte loads in the two epilogues (3.6 k cycles of a 33 k-cycle trip)
    unsigned rows2[5], rows3[CB_G];
#pragma unroll
    for (int t = 0; t < 5; t++) rows2[t] = CB_ROWS2[t][4 * kq] | (CB_ROWS2[t][4 * kq + 1] << 8) | (CB_ROWS2[t][4 * kq + 2] << 16) | ((unsigned)CB_ROWS2[t][4 * kq + 3] << 24);
#pragma unroll
    for (int g = 0; g < CB_G; g++) rows3[g] = CB_ROWS3[g][4 * kq] | (CB_ROWS3[g][4 * kq + 1] << 8) | (CB_ROWS3[g][4 * kq + 2] << 16) | ((unsigned)CB_ROWS3[g][4 * kq + 3] << 24);
    // y1 of a pair: 3600 float4, 15 per thread (the last round is partial); a missing second image reads as zeros
    float4 pre[15];
    auto request = [&](int pair, int tid_) {
        const int img0 = pair * CB_G, nimg = min(CB_G, n_img - img0);
#pragma unroll
        for (int u = 0; u < 15; u++) {
            const int q = u * 256 + tid_;
            pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < nimg * 1800) pre[u] = *reinterpret_cast<const float4 *>(y1 + (size_t)img0 * 7200 + (size_t)q * 4);
        }
    };
    int pair = blockIdx.x;
#ifdef CB_STAMPS
    { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); if (blockIdx.x == 0 && threadIdx.x == 0) g_cb_stamps[8] += n_ - cb_t; cb_t = n_; __builtin_amdgcn_sched_barrier(0); }
#endif
    if (pair < npair) request(pair, tid);
    for (; pair < npair; pair += gridDim.x) {
        const int img0 = pair * CB_G, nimg = min(CB_G, n_img - img0);
        // (every per-thread address below is a function of the thread id, i.e. loop-invariant: left alone, the optimiser hoists ~200 of them out of this persistent
        // loop and the kernel spills 150 registers. The id goes through an empty asm once per trip, so they are formed where they are used.)
        int tid_ = tid; asm volatile("" : "+v"(tid_));
        const int l_ = tid_ & 63, r16_ = l_ & 15, kq_ = l_ >> 4, w_ = tid_ >> 6, ncol_ = 16 * w_ + r16_;
        // (likewise the 48 + 18 fragment addresses of the streamed weights -- as 64-bit pointers held across the loop they alone were 130 registers -- are a uniform base plus
        // a 32-bit lane offset formed here, and the packed row tables are decoded where they are used)
        const unsigned boff2 = (unsigned)((w_ * 16 * 3) * 64 + l_) * 16u, boff3 = (unsigned)((w_ * 18 * 3) * 64 + l_) * 16u;
#pragma unroll
        for (int t = 0; t < 5; t++) asm volatile("" : "+v"(rows2[t]));
#pragma unroll
        for (int g = 0; g < CB_G; g++) asm volatile("" : "+v"(rows3[g]));
        CB_STAMP(0);            // loop top (+ the kernel's prologue on the first trip)
        // ---- the pair's y1, split, into its three planes
#pragma unroll
        for (int u = 0; u < 15; u++) {
            const int q = u * 256 + tid_;
            if (q < CB_G * 1800) {
                bf16x4 h, m, lo; cb_split4(pre[u], h, m, lo);
                unsigned char *d = cb_lds + (q >> 3) * CB_PS1 + (q & 7) * 8;
                *reinterpret_cast<bf16x4 *>(d) = h; *reinterpret_cast<bf16x4 *>(d + CB_PL1) = m; *reinterpret_cast<bf16x4 *>(d + 2 * CB_PL1) = lo;
            }
        }
        CB_STAMP(1);            // wait for the prefetched y1 + split + stores
        __syncthreads();
        CB_STAMP(2);            // barrier
        // ---- GEMM 1: rows m = image * 36 + position (72 = 4.5 tiles: the last half tile is computed from a clamped row and dropped), k-group = tap (ky, kx) x 32 channels
        f32x4 acc[5];
#pragma unroll
        for (int t = 0; t < 5; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            auto b2 = [&](int tap, int t) { return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(B2f) + (boff2 + (unsigned)((tap * 3 + t) * 1024))); };
            u32x4 bcur[3], bnext[3];
#pragma unroll
            for (int t = 0; t < 3; t++) bcur[t] = b2(0, t);
#pragma unroll
            for (int tap = 0; tap < 16; tap++) {
                const int tn = tap + 1 < 16 ? tap + 1 : tap;
#pragma unroll
                for (int t = 0; t < 3; t++) bnext[t] = b2(tn, t);
                const int toff = ((tap >> 2) * 15 + (tap & 3)) * CB_PS1;
#pragma unroll
                for (int t = 0; t < 5; t++) {
                    u32x4 a[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) a[k] = *reinterpret_cast<const u32x4 *>(cb_lds + k * CB_PL1 + a2[t] + toff);
                    acc[t] = cb_mac6(a, bcur, acc[t]);          // (six in a row on one accumulator: this instruction runs a single accumulation chain at full rate)
                    if (t & 1) __builtin_amdgcn_sched_barrier(0);          // (the ILP-first scheduler would pull every tile's reads of the tap up front: registers)
                }
#pragma unroll
                for (int t = 0; t < 3; t++) bcur[t] = bnext[t];
            }
        }
        CB_STAMP(3);            // GEMM 1
        if (pair + (int)gridDim.x < npair) request(pair + gridDim.x, tid_);           // the next pair's loads fly under the epilogue and GEMM 2 (~4 k cycles)
        // ---- y2 = relu(acc + bias), split, into its planes (nobody reads them before the barrier; y1's planes are left alone); training: y2 and its mask as bits
#pragma unroll
        for (int t = 0; t < 5; t++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = (int)((rows2[t] >> (8 * r)) & 0xffu);
                const bool own = t < 4 || ((CB_REAL5 >> (4 * kq_ + r)) & 1u);           // (a duplicate row of tile 4 computes the value of the row it copies: not stored twice)
                const float v = fmaxf(acc[t][r] + bn2, 0.f);
                if (own) {
                    const __bf16 a = (__bf16)v; const float r1 = v - (float)a; const __bf16 b = (__bf16)r1; const float r2 = r1 - (float)b;
                    unsigned char *d = cb_lds + 3 * CB_PL1 + m * CB_PS2 + ncol_ * 2;
                    *reinterpret_cast<__bf16 *>(d) = a; *reinterpret_cast<__bf16 *>(d + CB_PL2) = b; *reinterpret_cast<__bf16 *>(d + 2 * CB_PL2) = (__bf16)r2;
                }
                if (y2_out) {
                    const unsigned long long bal = __ballot(v > 0.f);
                    if (own && m < nimg * 36) {
                        y2_out[((size_t)img0 * 36 + m) * 64 + ncol_] = v;
                        if (r16_ == 0) mask2[((size_t)img0 * 36 + m) * 4 + w_] = (uint16_t)(bal >> (16 * kq_));
                    }
                }
            }
        }
        CB_STAMP(4);            // y2 epilogue
        __syncthreads();
        CB_STAMP(5);            // barrier
        // ---- GEMM 2: tile g = image g, row r16 = (oy, ox) of the 4 x 4 output; k-group = tap x half of the 64 channels
        f32x4 acc3[CB_G];
#pragma unroll
        for (int g = 0; g < CB_G; g++) acc3[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
        {
            auto b3g = [&](int kg, int t) { return *reinterpret_cast<const u32x4 *>(reinterpret_cast<const char *>(B3f) + (boff3 + (unsigned)((kg * 3 + t) * 1024))); };
            u32x4 bs[2][3];                         // the streamed k-groups' fragments: two in flight
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int t = 0; t < 3; t++) bs[j][t] = b3g(CB_B3_RES + j, t);
#pragma unroll
            for (int kg = 0; kg < 18; kg++) {
                const int tap = kg >> 1, koff = ((tap / 3) * 6 + (tap % 3)) * CB_PS2 + 64 * (kg & 1);
                u32x4 b3[3];
                if (kg < CB_B3_RES) {
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(b3[t].x) : "a"(b3a[kg < CB_B3_RES ? kg : 0][t][0]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(b3[t].y) : "a"(b3a[kg < CB_B3_RES ? kg : 0][t][1]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(b3[t].z) : "a"(b3a[kg < CB_B3_RES ? kg : 0][t][2]));
                        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(b3[t].w) : "a"(b3a[kg < CB_B3_RES ? kg : 0][t][3]));
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < 3; t++) b3[t] = bs[(kg - CB_B3_RES) & 1][t];
                    if (kg + 2 < 18) {
#pragma unroll
                        for (int t = 0; t < 3; t++) bs[(kg - CB_B3_RES) & 1][t] = b3g(kg + 2, t);
                    }
                }
#pragma unroll
                for (int g = 0; g < CB_G; g++) {
                    u32x4 a[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) a[k] = *reinterpret_cast<const u32x4 *>(cb_lds + k * CB_PL2 + a3[g] + koff);
                    acc3[g] = cb_mac6(a, b3, acc3[g]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        CB_STAMP(6);            // GEMM 2
#pragma unroll
        for (int g = 0; g < CB_G; g++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = (int)((rows3[g] >> (8 * r)) & 0xffu), gi = m >> 4, p = m & 15;
                const float v = fmaxf(acc3[g][r] + bn3, 0.f);
                const unsigned long long bal = __ballot(v > 0.f);
                if (gi < nimg) {
                    out[((size_t)(img0 + gi) * 16 + p) * 64 + ncol_] = v;
                    if (mask3 && r16_ == 0) mask3[((size_t)(img0 + gi) * 16 + p) * 4 + w_] = (uint16_t)(bal >> (16 * kq_));
                }
            }
        }
        CB_STAMP(7);            // output epilogue
        // (the next trip's staging writes y1's planes: every wave is past its GEMM 1 -- the barrier above -- and the y2 planes are rewritten only after the next barrier)
    }
#ifdef CB_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 512) g_cb_wg[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
#endif
}

extern "C" int grip_conv23_prep(const float *w2_dev, const int64_t *w2_strides, const float *w3_dev, const int64_t *w3_strides, float *b2_mat_dev, float *b3_mat_dev,
                                void *stream) {
    if (!w2_dev || !w2_strides || !w3_dev || !w3_strides || !b2_mat_dev || !b3_mat_dev) return grip_fail("grip_conv23_prep: need both weight tensors, their strides and the two outputs");
    hipLaunchKernelGGL(k_conv23_prep, dim3(576 * 64 / 256), dim3(256), 0, (hipStream_t)stream, w2_dev, (long long)w2_strides[0], (long long)w2_strides[1],
                       (long long)w2_strides[2], (long long)w2_strides[3], w3_dev, (long long)w3_strides[0], (long long)w3_strides[1], (long long)w3_strides[2],
                       (long long)w3_strides[3], b2_mat_dev, b3_mat_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char buf[160]; snprintf(buf, sizeof buf, "grip_conv23_prep: %s", hipGetErrorString(e)); return grip_fail(buf); }
    return 0;
}

extern "C" int grip_conv23_train(const float *y1_nhwc_dev, int n, const float *b2_mat_dev, const float *bias2_dev, const float *b3_mat_dev, const float *bias3_dev,
                                 float *out_nhwc_dev, float *y2_nhwc_dev, uint64_t *mask2_dev, uint64_t *mask3_dev, void *stream);
extern "C" int grip_conv23(const float *y1_nhwc_dev, int n, const float *b2_mat_dev, const float *bias2_dev, const float *b3_mat_dev, const float *bias3_dev,
                           float *out_nhwc_dev, void *stream) {
    return grip_conv23_train(y1_nhwc_dev, n, b2_mat_dev, bias2_dev, b3_mat_dev, bias3_dev, out_nhwc_dev, nullptr, nullptr, nullptr, stream);
}
extern "C" int grip_conv23_train(const float *y1_nhwc_dev, int n, const float *b2_mat_dev, const float *bias2_dev, const float *b3_mat_dev, const float *bias3_dev,
                                 float *out_nhwc_dev, float *y2_nhwc_dev, uint64_t *mask2_dev, uint64_t *mask3_dev, void *stream) {
    if ((y2_nhwc_dev != nullptr) != (mask2_dev != nullptr) || (y2_nhwc_dev != nullptr) != (mask3_dev != nullptr))
        return grip_fail("grip_conv23_train: the training outputs (y2, mask2, mask3) come together or not at all");
    if (!y1_nhwc_dev || !b2_mat_dev || !bias2_dev || !b3_mat_dev || !bias3_dev || !out_nhwc_dev || n <= 0)
        return grip_fail("grip_conv23: need y1 [n, 15, 15, 32], the two weight matrices of grip_conv23_prep, both biases and the output [n, 4, 4, 64]");
    const size_t lds = (size_t)C2_LDS_FLOATS * sizeof(float);
    {   // the dynamic-LDS opt-in is a per-DEVICE function attribute: remember it per device (a process may drive several GPUs, from several threads)
        static std::atomic<unsigned long long> attr_set_mask{0ULL};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return grip_fail("grip_conv23: no current device");
        const unsigned long long bit = 1ULL << (dev & 63);
        if (!(attr_set_mask.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void *)k_conv23, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return grip_fail("grip_conv23: cannot reserve LDS");
            attr_set_mask.fetch_or(bit, std::memory_order_release);
        }
    }
    // The shipped kernel computes on the bf16 matrix pipe in fp32-equivalent arithmetic (k_conv23_b3: persistent, one workgroup per CU); GRIP_CONV23_F32=1 (read once per
    // process) keeps rounds 2-4's fp32-MFMA kernel as the comparison.
    static const bool f32_kernel = [] { const char *e = getenv("GRIP_CONV23_F32"); return e && e[0] == '1'; }();
    if (!f32_kernel) {
        static std::atomic<unsigned long long> b3_attr_mask{0ULL};
        static std::atomic<int> cus[64];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return grip_fail("grip_conv23: no current device");
        const unsigned long long bit = 1ULL << (dev & 63);
        if (!(b3_attr_mask.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void *)k_conv23_b3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CB_LDS_BYTES) != hipSuccess) return grip_fail("grip_conv23: cannot reserve LDS");
            int n_cu = 0;
            if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) return grip_fail("grip_conv23: cannot read the CU count");
            cus[dev & 63].store(n_cu, std::memory_order_relaxed);
            b3_attr_mask.fetch_or(bit, std::memory_order_release);
        }
        const int npair = (n + CB_G - 1) / CB_G, grid = std::min(npair, cus[dev & 63].load(std::memory_order_relaxed));
        hipLaunchKernelGGL(k_conv23_b3, dim3(grid), dim3(256), CB_LDS_BYTES, (hipStream_t)stream, y1_nhwc_dev, n, reinterpret_cast<const u32x4 *>(b2_mat_dev + 2 * 512 * 64), bias2_dev,
                           reinterpret_cast<const u32x4 *>(b3_mat_dev + 2 * 576 * 64), bias3_dev, out_nhwc_dev, y2_nhwc_dev, (uint16_t *)mask2_dev, (uint16_t *)mask3_dev);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { char buf[160]; snprintf(buf, sizeof buf, "grip_conv23: %s", hipGetErrorString(e)); return grip_fail(buf); }
        return 0;
    }
    hipLaunchKernelGGL(k_conv23, dim3((n + C2_G - 1) / C2_G), dim3(256), lds, (hipStream_t)stream, y1_nhwc_dev, n, b2_mat_dev + 512 * 64, bias2_dev, b3_mat_dev + 576 * 64, bias3_dev, out_nhwc_dev, y2_nhwc_dev,
                       (uint16_t *)mask2_dev, (uint16_t *)mask3_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char buf[160]; snprintf(buf, sizeof buf, "grip_conv23: %s", hipGetErrorString(e)); return grip_fail(buf); }
    return 0;
}
