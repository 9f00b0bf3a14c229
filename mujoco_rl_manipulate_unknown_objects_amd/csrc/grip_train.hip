// grip_train.hip -- the update side of the policy (PPO / SAC minibatches; reference models/feature_extractor.py:14-22 trained by stable_baselines3 as
// train_agent.py:33-47 configures it), hand-written where the time went, in fp32 or fp32-EQUIVALENT arithmetic throughout (rounds 3-4: v_mfma_f32_16x16x4_f32 /
// v_mfma_f32_32x32x2_f32; round 5: the bf16 matrix pipe on three-term splits of both operands, six products, <= 2^-24 relative per product -- the tensor library's
// fp32 path up to the order of summation; tests/test_gpu_train_kernels.py).
//
// As tensor-library calls the backward of the three convolutions is, per 4096-sample minibatch on MI355X: data gradients 109 + 306 us (the stride-2 4 x 4
// layer at 32 TFLOP/s), weight gradients 55 + 93 + 211 us, three ReLU-mask passes (16 + 16 + 49 us), three bias-gradient reductions
// (14 + 18 + 39 us), a uint8 -> float image pass for the first layer's weight gradient (67 us) and ~40 us of zero-fills: ~1.0 ms of the
// update's 2.0 ms. Here:
//   k_trunk_bwd          g3 (gradient at the third layer's ReLU output) -> masked g3, masked g2 (the operands of the library's two remaining weight
//                        gradients), the first layer's weight gradient straight from the observation bytes and all three bias gradients, in ONE launch:
//                        both data gradients as scatter GEMMs whose column blocks are summed into an LDS tile, g2 and g1 never leaving LDS, the ReLU
//                        masks (bits written by the forward kernels) folded into the passes. 283 us; 214 us since the first layer's weight gradient runs on the bf16 pipe.
//   k_trunk_bwd_b3       (round 5, the shipped one) the same with both data-gradient GEMMs on the bf16 pipe, one 8-wave workgroup per CU: 176-191 us.
//   k_wgrad23_b3         (round 5) the weight gradients of the second and third layer on the bf16 pipe (transposed LDS reads), + k_wgrad23_reduce: 92 us for
//                        the tensor library's 152.
//   k_wgrad1_reduce      the workgroups' partial sums -> the gradient tensors (fixed order, no atomics).
//   k_gradnorm, k_clip_adam   clip_grad_norm_ + Adam.step() on the optimiser's own state tensors, two launches.
//   k_tanh_bwd_colsum, k_colsum_reduce   activation derivative + bias gradient of a dense layer in one pass (tanh: the policy | value MLPs; ReLU: the
//                        extractor's linear layer).
// The forward kernels the update shares with the rollouts (k_conv1_u8, k_conv23, with the training outputs) are in grip_policy.hip.
#include <hip/hip_runtime.h>
#include <atomic>
#include <algorithm>
#include <stdint.h>
#include <cstdio>

int grip_fail(const char *msg);                     // grip_sim.hip
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ------------------------------------------------------------------------------------------------------------------
// k_trunk_bwd. Shapes (NHWC): y1 [n, 15, 15, 32], y2 [n, 6, 6, 64], y3 [n, 4, 4, 64] (post-ReLU activations of the forward), g3 [n, 4, 4, 64] = d loss / d y3,
// obs uint8 [n, 5, 64, 64]. Weights as the forward's GEMM operands (grip_conv23_prep): B3[(ky, kx, ci)][co] (576 x 64), B2[(ky, kx, ci)][co] (512 x 64).
//   g3m = g3 * (y3 > 0)
//   col3[(img, oy, ox)][(ky, kx, ci)] = sum_co g3m[img, oy, ox, co] * w3[co, ci, ky, kx];  g2[img, oy + ky, ox + kx, ci] += col3;  g2m = g2 * (y2 > 0)
//   col2[(img, oy, ox)][(ky, kx, ci)] = sum_co g2m[img, oy, ox, co] * w2[co, ci, ky, kx];  g1[img, 2 oy + ky, 2 ox + kx, ci] += col2;  g1m = g1 * (y1 > 0)
//   dW1[(ci, ky, kx)][co] += sum_(oy, ox) obs[img, ci, 4 oy + ky, 4 ox + kx] * g1m[img, oy, ox, co];  db1[co] += sum g1m        (dW1 / 255 in the reduction)
// Persistent workgroups (4 waves) walk groups of DG = 2 images, two workgroups per CU (one stages / masks while the other's MFMAs run).
// The data-gradient GEMMs: M = (image, position) rows, K = 64 output channels, N = (ky, kx, ci) columns; the A operand (the masked gradient tile) sits in
// registers for the whole GEMM, the weights come from L2 (16 floats per lane and 16-column block, requested one block ahead). The reduction index is
// permuted -- k-step j of an instruction pairs co = 16 (l >> 4) + j -- so that a lane's 16 operand values are contiguous in memory (b128 reads) in both A
// and B; a sum does not care. After the 16 k-steps of a column block its 16 x 16 tile is ADDED into the LDS tile of the layer below, software-pipelined: the
// reads of the old values go out before the first half of the NEXT block's MFMAs, the adds and writes sit between its halves (an add pass issued on its own
// costs 60 cycles per word: 65 us of the first version's 225). No atomics: in GEMM 1 wave w owns input channels 16 w .. 16 w + 15 of every tap; in GEMM 2
// wave w owns the taps of one parity class (ky & 1, kx & 1) = (w >> 1, w & 1), whose target pixels have that parity -- waves never touch the same word,
// within a wave one block's rows are distinct pixels and blocks follow in program order.
// The first layer's weight gradient never sees g1 in memory: the masked tile stays in LDS and is the B operand of v_mfma_f32_32x32x2_f32 against the
// image bytes (A, converted on the fly; the planes of one image take the place of the dead g2 tile): M = 256 patch elements (wave w = plane w, two 32-row
// tiles = ky 0..3 / 4..7), N = 32 channels, K = 225 positions per image, accumulated in registers over all of the workgroup's images. (Round 4: on the bf16 pipe, exactly -- a byte
// is a bf16, the gradient three bf16 terms: see wgrad_image.)
#define DG 2
#define D_PS1 34                            // floats per g1 pixel in LDS
#define D_PS2 68                            // floats per g2 / g3 row in LDS
#define D_T1 (DG * 225 * D_PS1)             // g1 tile; the g3 tile (DG * 16 * D_PS2 floats) lives in its first part until GEMM 1 has its operands
#define D_T2 (DG * 36 * D_PS2)              // g2 tile; later the four 64 x 64 byte planes of one image (16 KB of its 19.1)
#define D_LDS_FLOATS (D_T1 + D_T2)
#define TB_PART (256 * 32 + 32 + 64 + 64)   // floats of a workgroup's partial sums: dW1 as [(ci, ky, kx)][co], then db1[32], db2[64], db3[64]

// MT row tiles x NB column blocks. The lane's A operand of tile t and k-steps 4 q .. 4 q + 3 is the float4 at TA[arow[t] + 4 q] (row t * 16 + (l & 15), co = 16 (l >> 4) + 4 q ..),
// read again for every column block (20 b128 reads per 80 MFMAs) rather than held (80 registers); bofs(nb) = float offset of the block's weights from Bp;
// tgt(nb, t, r) = LDS word (index into T) the lane's accumulator element r of tile t is added to, if live(t). The adds of block nb are pipelined into block
// nb + 1: the old sums are requested before its MFMAs, added and written back between its first and second k-quarter (ds_add_f32 would need no registers,
// but executes at a few lanes per cycle: 3x the whole kernel's time).
// A later block's reads see words OTHER LANES of the wave wrote in an earlier one: the hardware executes a wave's LDS instructions in order, but to the
// compiler those are different threads -- wave_order() keeps it from moving reads above earlier writes.
__device__ __forceinline__ void wave_order() { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
template <int MT, int NB, class BOfs, class Tgt, class Live>
__device__ __forceinline__ void scatter_gemm(const float *TA, const int (&arow)[MT], const float *__restrict__ Bp, float *T, BOfs bofs, Tgt tgt, Live live) {
    float4 bn[4];
#pragma unroll
    for (int q = 0; q < 4; q++) bn[q] = *reinterpret_cast<const float4 *>(Bp + bofs(0) + 4 * q);
    f32x4 prev[MT];
#pragma unroll
    for (int nb = 0; nb <= NB; nb++) {
        float4 b[4], aq[MT];
        float old[MT][4];
        if (nb < NB) {
#pragma unroll
            for (int q = 0; q < 4; q++) b[q] = bn[q];
            if (nb + 1 < NB) {
#pragma unroll
                for (int q = 0; q < 4; q++) bn[q] = *reinterpret_cast<const float4 *>(Bp + bofs(nb + 1) + 4 * q);
            }
#pragma unroll
            for (int t = 0; t < MT; t++) aq[t] = *reinterpret_cast<const float4 *>(TA + arow[t]);
        }
        if (nb > 0) {
#pragma unroll
            for (int t = 0; t < MT; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) old[t][r] = T[tgt(nb - 1, t, r)];
        }
        __builtin_amdgcn_sched_barrier(0);      // the requests (next block's weights, first operands, old sums) stay above the MFMAs: the scheduler would sink them to their uses
        f32x4 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (nb < NB) {
                if (q > 0) {                    // (not double-buffered: 20 registers this kernel does not have; the co-resident workgroup's wave covers the LDS latency)
#pragma unroll
                    for (int t = 0; t < MT; t++) aq[t] = *reinterpret_cast<const float4 *>(TA + arow[t] + 4 * q);
                }
#pragma unroll
                for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[t].x, b[q].x, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[t].y, b[q].y, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[t].z, b[q].z, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[t].w, b[q].w, acc[t], 0, 0, 0);
            }
            if (q == 0 && nb > 0) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < MT; t++)
                    if (live(t)) {
#pragma unroll
                        for (int r = 0; r < 4; r++) T[tgt(nb - 1, t, r)] = old[t][r] + prev[t][r];
                    }
                wave_order();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nb < NB) {
#pragma unroll
            for (int t = 0; t < MT; t++) prev[t] = acc[t];
        }
    }
}

__device__ __forceinline__ float4 bit_mask(unsigned bits, float4 v) { return make_float4(bits & 1u ? v.x : 0.f, bits & 2u ? v.y : 0.f, bits & 4u ? v.z : 0.f, bits & 8u ? v.w : 0.f); }

struct GroupIn { float4 g3[2]; unsigned m3[2]; unsigned long long m2; unsigned m1[2]; };     // a thread's share of a group's gradient tile and ReLU masks

__device__ __forceinline__ void group_in_load(GroupIn &gi, int tl, int img0, int nimg, const float *__restrict__ g3, const uint16_t *__restrict__ m3h,
                                              const unsigned long long *__restrict__ m2, const uint32_t *__restrict__ m1) {
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = tl + 256 * u, row = i >> 4, c4 = (i & 15) * 4;
        gi.g3[u] = make_float4(0.f, 0.f, 0.f, 0.f); gi.m3[u] = 0u;
        if ((row >> 4) < nimg) {
            gi.g3[u] = *reinterpret_cast<const float4 *>(g3 + ((size_t)img0 * 16 + row) * 64 + c4);
            gi.m3[u] = (unsigned)m3h[((size_t)img0 * 16 + row) * 4 + (c4 >> 4)] >> (c4 & 15);
        }
        const int pj = tl + 256 * u;
        gi.m1[u] = pj < nimg * 225 ? m1[(size_t)img0 * 225 + pj] : 0u;
    }
    gi.m2 = tl < nimg * 36 ? m2[(size_t)img0 * 36 + tl] : 0ULL;
}

__global__ void __launch_bounds__(256, 2) k_trunk_bwd(const float *__restrict__ g3, const uint64_t *__restrict__ m3, const uint64_t *__restrict__ m2, const uint32_t *__restrict__ m1,
                                                      const uint8_t *__restrict__ obs, const long long *__restrict__ obs_rows, int channels, const float *__restrict__ B3, const float *__restrict__ B2, int n_img,
                                                      float *__restrict__ g3m_out, float *__restrict__ g2m_out, float *__restrict__ g1m_out, float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float d_lds[];
    float *T1 = d_lds, *T2 = d_lds + D_T1, *T3 = d_lds;
    uint8_t *U = reinterpret_cast<uint8_t *>(T2);
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r16 = l & 15, kq = l >> 4;
    const int half = l >> 5, m32 = l & 31;                               // lane roles in the 32 x 32 x 2 instruction
    f32x16 cw[2];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int r = 0; r < 16; r++) cw[t][r] = 0.f;
    float4 bs4 = make_float4(0.f, 0.f, 0.f, 0.f), bs2 = bs4, bs3 = bs4;      // bias gradients: the thread's four channels of each layer, over its rows
    const int ngroups = (n_img + DG - 1) / DG;
    GroupIn gi;
    if ((int)blockIdx.x < ngroups) group_in_load(gi, tid, blockIdx.x * DG, min(DG, n_img - (int)blockIdx.x * DG), g3, reinterpret_cast<const uint16_t *>(m3),
                                                 reinterpret_cast<const unsigned long long *>(m2), m1);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int img0 = grp * DG, nimg = min(DG, n_img - img0);
    // the weights are the same for every group: hidden from the optimiser, which would otherwise hoist all 68 float4 loads per lane out of this loop (and spill them)
    const float *B3g = B3, *B2g = B2;
    asm volatile("" : "+s"(B3g), "+s"(B2g));
    int tl = tid;                                                           // likewise the staging passes' per-thread indices (all functions of tid): recomputed, not kept
    asm volatile("" : "+v"(tl));
    // ---- g3m = g3 * mask3 -> T3 (and memory: the third layer's weight gradient reads it); zero T2
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int i = tl + 256 * u, row = i >> 4, c4 = (i & 15) * 4;
        const float4 v = bit_mask(gi.m3[u], gi.g3[u]);
        bs3.x += v.x; bs3.y += v.y; bs3.z += v.z; bs3.w += v.w;
        if (g3m_out && (row >> 4) < nimg) *reinterpret_cast<float4 *>(g3m_out + ((size_t)img0 * 16 + row) * 64 + c4) = v;
        *reinterpret_cast<float4 *>(T3 + row * D_PS2 + c4) = v;
    }
#pragma unroll 1
    for (int i = tl; i < D_T2 / 4; i += 256) reinterpret_cast<float4 *>(T2)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tl < DG * 36) *reinterpret_cast<unsigned long long *>(T2 + tl * D_PS2 + 64) = gi.m2;      // the rows' ReLU masks ride in their padding words (the GEMMs never touch them)
    // ---- GEMM 1: rows (image t, position r16 = oy * 4 + ox), 9 column blocks (taps) of this wave's 16 input channels
    {
        int arow[DG];
#pragma unroll
        for (int t = 0; t < DG; t++) arow[t] = (t * 16 + r16) * D_PS2 + 16 * kq;
        const int tb = kq * 6 * D_PS2 + 16 * w + r16;                      // the lane holds rows 4 kq + r: (oy, ox) = (kq, r)
        scatter_gemm<DG, 9>(T3, arow, B3g + (size_t)(16 * w + r16) * 64 + 16 * kq, T2,
                            [](int c) { return c * 4096; },                  // column n = c * 64 + 16 w + r16
                            [&](int c, int t, int r) { return tb + (t * 36 + (c / 3) * 6 + r + (c % 3)) * D_PS2; }, [](int) { return true; });
    }
    __syncthreads();                                                        // T2 complete, T3 no longer needed
    // ---- g2m = g2 * mask2: to memory (the second layer's weight gradient reads it) and back into T2 as GEMM 2's operand; zero T1
#pragma unroll
    for (int u = 0; u < 5; u++) {
        const int i = tl + 256 * u, row = i >> 4, c4 = (i & 15) * 4;
        if (i < DG * 36 * 16) {
            const unsigned bits = reinterpret_cast<const unsigned *>(T2)[row * D_PS2 + 64 + (c4 >> 5)] >> (c4 & 31);
            const float4 v = bit_mask(bits, *reinterpret_cast<float4 *>(T2 + row * D_PS2 + c4));
            *reinterpret_cast<float4 *>(T2 + row * D_PS2 + c4) = v;
            bs2.x += v.x; bs2.y += v.y; bs2.z += v.z; bs2.w += v.w;
            if (row < nimg * 36) *reinterpret_cast<float4 *>(g2m_out + ((size_t)img0 * 36 + row) * 64 + c4) = v;
        }
    }
#pragma unroll 1
    for (int i = tl; i < D_T1 / 4; i += 256) reinterpret_cast<float4 *>(T1)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++)
        if (tl + 256 * u < DG * 225) reinterpret_cast<unsigned *>(T1)[(tl + 256 * u) * D_PS1 + 32] = gi.m1[u];
    // ---- GEMM 2: rows m = image * 36 + position (72 = 4.5 tiles: the last half tile is padding, computed from a clamped row and dropped), 8 column
    // blocks = this wave's 4 taps x 2 channel halves
    {
        int arow[5], base[5][4];
        const int py = w >> 1, px = w & 1;
#pragma unroll
        for (int t = 0; t < 5; t++) {
            arow[t] = min(t * 16 + r16, DG * 36 - 1) * D_PS2 + 16 * kq;
#pragma unroll
            for (int r = 0; r < 4; r++) {               // (the wave's parity offset goes into the base: what is left per block is a compile-time constant, i.e. an instruction offset)
                const int mo = min(t * 16 + 4 * kq + r, DG * 36 - 1), g = mo / 36, p = mo - g * 36, oy = p / 6, ox = p - oy * 6;
                base[t][r] = (g * 225 + (2 * oy + py) * 15 + 2 * ox + px) * D_PS1 + r16;
            }
        }
        const bool pad_ok = kq < 2;                     // tile 4: rows 64 + 4 kq + r exist for kq < 2 only
        // column block nt: tap (ky, kx) = (py + 2 (nt >> 2), px + 2 ((nt >> 1) & 1)), channel half nt & 1; n = (ky * 4 + kx) * 32 + 16 (nt & 1) + r16
        scatter_gemm<5, 8>(T2, arow, B2g + (size_t)((py * 4 + px) * 32 + r16) * 64 + 16 * kq, T1,
                           [](int nt) { return (((nt >> 2) * 8 + 2 * ((nt >> 1) & 1)) * 32 + 16 * (nt & 1)) * 64; },
                           [&](int nt, int t, int r) { return base[t][r] + ((nt >> 2) * 30 + 2 * ((nt >> 1) & 1)) * D_PS1 + 16 * (nt & 1); },
                           [&](int t) { return t < 4 || pad_ok; });
    }
    __syncthreads();                                                        // T1 complete, T2 free
    // ---- g1m = g1 * mask1, in place (and to memory when asked for), while the first image's byte planes are on their way (to registers, then U = T2's space)
    uint4 pa = make_uint4(0, 0, 0, 0), pb = pa, pc = pa, pd = pa;            // (scalars: as an array the four land in scratch memory)
    if (obs) {
        const uint4 *src = reinterpret_cast<const uint4 *>(obs + (size_t)(obs_rows ? obs_rows[img0] : img0) * channels * 4096) + tl;
        pa = src[0]; pb = src[256]; pc = src[512]; pd = src[768];
    }
#pragma unroll 2
    for (int i = tl; i < DG * 225 * 8; i += 256) {
        const int pix = i >> 3, c4 = (i & 7) * 4;
        float2 *tp = reinterpret_cast<float2 *>(T1 + pix * D_PS1 + c4);
        const float2 v0 = tp[0], v1 = tp[1];
        const float4 v = bit_mask(reinterpret_cast<const unsigned *>(T1)[pix * D_PS1 + 32] >> c4, make_float4(v0.x, v0.y, v1.x, v1.y));
        tp[0] = make_float2(v.x, v.y); tp[1] = make_float2(v.z, v.w);
        bs4.x += v.x; bs4.y += v.y; bs4.z += v.z; bs4.w += v.w;             // db1: the thread's four channels (c4 is the same in every pass: 256 % 8 == 0)
        if (g1m_out && i < nimg * 225 * 8) *reinterpret_cast<float4 *>(g1m_out + (size_t)img0 * 7200 + (size_t)i * 4) = v;
    }
    const int gnext = grp + gridDim.x;
    auto next_group = [&]() {
        if (gnext < ngroups)
            group_in_load(gi, tl, gnext * DG, min(DG, n_img - gnext * DG), g3, reinterpret_cast<const uint16_t *>(m3), reinterpret_cast<const unsigned long long *>(m2), m1);
    };
#ifndef WGRAD1_F32
    // K loop over an image's 225 positions on the bf16 pipe, EXACTLY as k_conv1_u8 computes the forward (grip_policy.hip): a pixel 0..255 is a bf16, the fp32 gradient is the
    // sum of three bf16 terms (nearest bf16, exact remainder, twice: what is left is below its 25th bit), a byte times a bf16 is exact in the fp32 accumulator -- three
    // v_mfma_f32_32x32x16_bf16 per tile and 16 positions where the fp32 instruction needed eight v_mfma_f32_32x32x2_f32 (96 against 512 matrix-pipe cycles). A chunk of 16
    // positions = one row oy of the image's 15 x 15: lanes 0..31 hold ox 0..7, lanes 32..63 ox 8..15 (ox = 15 does not exist: its B operand is forced to zero, its A bytes are
    // whatever follows the row -- finite). The gradient's split is redone by every wave (the four planes share it; LDS has no room for the three terms of a tile): the loop
    // is bound by those ~70 VALU and 24 LDS instructions per chunk, not by its six MFMAs. Measured: k_trunk_bwd 310 -> 214 us per 4096 samples, the captured update 1.055 ->
    // 0.992 ms per minibatch (same box; -DWGRAD1_F32 keeps the fp32 loop for the comparison).
    auto wgrad_image = [&](int g) {
        const uint8_t *ap = U + w * 4096 + (m32 >> 3) * 64 + (m32 & 7) + half * 32;          // + oy * 256 + 4 j; second tile (ky + 4): + 256
        const float *bp = T1 + (g * 225 + half * 8) * D_PS1 + m32;                            // + (oy * 15 + j) * D_PS1
#pragma unroll 1
        for (int oy = 0; oy < 15; oy++) {
            bf16x8 b0, b1, b2;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const float x = (j == 7 && half) ? 0.f : bp[j * D_PS1];
                const __bf16 h = (__bf16)x; const float r = x - (float)h;
                const __bf16 md = (__bf16)r; const float r2 = r - (float)md;
                b0[j] = h; b1[j] = md; b2[j] = (__bf16)r2;
            }
#pragma unroll
            for (int t = 0; t < 2; t++) {
                bf16x8 a;
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] = (__bf16)(float)ap[4 * j + 256 * t];
                cw[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, cw[t], 0, 0, 0);
                cw[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, cw[t], 0, 0, 0);
                cw[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b2, cw[t], 0, 0, 0);
            }
            ap += 256; bp += 15 * D_PS1;
        }
    };
#else
    // K loop over an image's 225 positions, two per instruction: lanes 0..31 walk rows 0..7, lanes 32..63 rows 8..15 (row 15 does not exist: its B operand
    // is forced to zero; 120 k-steps instead of 113, but every address is the row base plus an instruction offset -- with positions 2 j and 2 j + 1 in
    // the two halves the (oy, ox) bookkeeping was a dozen dependent VALU instructions per k-step, and the loop ran at 70 % of the MFMA rate)
    auto wgrad_image = [&](int g) {
        const uint8_t *ap = U + w * 4096 + (m32 >> 3) * 64 + (m32 & 7) + half * 8 * 256;      // + oy * 256 + 4 ox; second tile (ky + 4): + 256
        const float *bp = T1 + (g * 225 + half * 8 * 15) * D_PS1 + m32;
#pragma unroll 1
        for (int ro = 0; ro < 7; ro++) {
#pragma unroll
            for (int ox = 0; ox < 15; ox++) {
                const float b = bp[ox * D_PS1];
                cw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)ap[4 * ox], b, cw[0], 0, 0, 0);
                cw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)ap[4 * ox + 256], b, cw[1], 0, 0, 0);
            }
            ap += 256; bp += 15 * D_PS1;
        }
#pragma unroll
        for (int ox = 0; ox < 15; ox++) {                                   // rows 7 | 15: the upper half's row does not exist
            const float b = half ? 0.f : bp[ox * D_PS1];
            cw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)ap[4 * ox], b, cw[0], 0, 0, 0);
            cw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)ap[4 * ox + 256], b, cw[1], 0, 0, 0);
        }
    };
#endif
    if (obs) {
        uint4 *Uq = reinterpret_cast<uint4 *>(U) + tl;
        Uq[0] = pa; Uq[256] = pb; Uq[512] = pc; Uq[768] = pd;
        __syncthreads();
        // requests that fly during the K loop (it issues none of its own): the second image's planes, or the next group's gradient tile and masks
        if (nimg > 1) {
            const uint4 *src = reinterpret_cast<const uint4 *>(obs + (size_t)(obs_rows ? obs_rows[img0 + 1] : img0 + 1) * channels * 4096) + tl;
            pa = src[0]; pb = src[256]; pc = src[512]; pd = src[768];
        } else next_group();
        wgrad_image(0);
        if (nimg > 1) {
            __syncthreads();                                                // every wave is done with the first image's planes
            Uq[0] = pa; Uq[256] = pb; Uq[512] = pc; Uq[768] = pd;
            __syncthreads();
            next_group();
            wgrad_image(1);
        }
    } else next_group();
    __syncthreads();                                                        // before the next group's staging
  }
    if (part) {
        float *P = part + (size_t)blockIdx.x * TB_PART;
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;          // = ky_local * 8 + kx of tile t
                P[(w * 64 + t * 32 + row) * 32 + m32] = cw[t][r];
            }
        // db1: thread tid holds channels 4 (tid & 7) .. + 3 summed over its pixels: 32 threads per channel group, added in a fixed order through LDS
        __syncthreads();
        *reinterpret_cast<float4 *>(d_lds + tid * 4) = bs4;
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
            for (int k = 0; k < 32; k++) t += d_lds[((tid >> 2) + 8 * k) * 4 + (tid & 3)];
            P[256 * 32 + tid] = t;
        }
        // db2, db3: channels 4 (tid & 15) .. + 3, 16 threads per channel group
        __syncthreads();
        *reinterpret_cast<float4 *>(d_lds + tid * 4) = bs2; *reinterpret_cast<float4 *>(d_lds + 1024 + tid * 4) = bs3;
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, co = tid & 63;
            float t = 0.f;
            for (int k = 0; k < 16; k++) t += d_lds[which * 1024 + ((co >> 2) + 16 * k) * 4 + (co & 3)];
            P[256 * 32 + 32 + tid] = t;
        }
    }
}

// dW1[co][ci][ky][kx] = sum over the workgroups' partials / 255 (the forward multiplies bytes by w / 255), db1 / db2 / db3 = sums: 32 outputs per block,
// 32 slices of the partials each, added in a fixed order
__global__ void __launch_bounds__(1024) k_wgrad1_reduce(const float *__restrict__ part, int nparts, float *__restrict__ gw, long long so, long long sc, long long sy, long long sx,
                                                         float *__restrict__ gb, float *__restrict__ gb2, float *__restrict__ gb3) {
    __shared__ float red[32][32];
    const int o = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;         // 32 slices of the partials per output (17 MB to read: parallelism, not arithmetic)
    float s = 0.f;
    for (int p = sl; p < nparts; p += 32) s += part[(size_t)p * TB_PART + o];
    red[sl][threadIdx.x & 31] = s;
    __syncthreads();
    if (sl == 0) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int i = 1; i < 32; i++) t += red[i][threadIdx.x];
        if (o < 256 * 32) {
            const int mm = o >> 5, co = o & 31, ci = mm >> 6, ky = (mm >> 3) & 7, kx = mm & 7;
            gw[co * so + ci * sc + ky * sy + kx * sx] = t * (1.0f / 255.0f);
        } else if (o < 256 * 32 + 32) gb[o - 256 * 32] = t;
        else if (o < 256 * 32 + 96) { if (gb2) gb2[o - 256 * 32 - 32] = t; }
        else if (gb3) gb3[o - 256 * 32 - 96] = t;
    }
}

static int set_dyn_lds(const void *fn, size_t bytes, std::atomic<unsigned long long> &mask, const char *who) {
    // the dynamic-LDS opt-in is a per-DEVICE function attribute: remember it per device (a process may drive several GPUs, from several threads)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return grip_fail("grip_train: no current device");
    const unsigned long long bit = 1ULL << (dev & 63);
    if (!(mask.load(std::memory_order_acquire) & bit)) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) { char buf[96]; snprintf(buf, sizeof buf, "%s: cannot reserve LDS", who); return grip_fail(buf); }
        mask.fetch_or(bit, std::memory_order_release);
    }
    return 0;
}

static int launch_check(const char *who) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { char buf[160]; snprintf(buf, sizeof buf, "%s: %s", who, hipGetErrorString(e)); return grip_fail(buf); }
    return 0;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define E_RS 416                                       // bytes of a gradient row in the bf16 planes of k_trunk_bwd_b3: [term 3][64 channels] + 32
#define E_T2 (DG * 225 * D_PS1 * 4)
#define E_A3 (E_T2 + DG * 36 * D_PS2 * 4)
#define E_A2 (E_A3 + DG * 16 * E_RS)
#define E_LDS (E_A2 + 80 * E_RS)
#define E_U E_T2
#define E_G1P (E_T2 + 16384)
static_assert(E_G1P + 15 * 16 * 192 <= E_A2 + 72 * E_RS, "the first layer's split gradient must end before the padding rows of the g2m planes");
__global__ void k_trunk_bwd_b3(const float *__restrict__ g3, const uint64_t *__restrict__ m3, const uint64_t *__restrict__ m2, const uint32_t *__restrict__ m1, const uint8_t *__restrict__ obs,
                               const long long *__restrict__ obs_rows, int channels, const u32x4 *__restrict__ B3d, const u32x4 *__restrict__ B2d, int n_img, float *__restrict__ g3m_out,
                               float *__restrict__ g2m_out, float *__restrict__ g1m_out, float *__restrict__ part);

extern "C" int grip_trunk_backward_parts(int n) { const int groups = (n + DG - 1) / DG; return groups < 512 ? groups : 512; }

extern "C" int grip_trunk_backward(const float *g3_dev, const uint64_t *mask3_dev, const uint64_t *mask2_dev, const uint32_t *mask1_dev, const uint8_t *obs_dev,
                                   const int64_t *obs_rows_dev, int channels,
                                   const float *b3_mat_dev, const float *b2_mat_dev, int n, float *g3m_dev, float *g2m_dev, float *g1m_dev, float *partials_dev,
                                   float *grad_w1_dev, const int64_t *grad_w1_strides, float *grad_b1_dev, float *grad_b2_dev, float *grad_b3_dev, void *stream) {
    if (!g3_dev || !mask3_dev || !mask2_dev || !mask1_dev || !b3_mat_dev || !b2_mat_dev || !g2m_dev || n <= 0)
        return grip_fail("grip_trunk_backward: need g3 [n, 4, 4, 64], the three ReLU masks of the training forward (grip_conv1_u8_train, grip_conv23_train), the two weight "
                         "matrices of grip_conv23_prep and the output g2m");
    if (obs_dev && (channels != 5 || !partials_dev || !grad_w1_dev || !grad_w1_strides || !grad_b1_dev))
        return grip_fail("grip_trunk_backward: with observations (uint8 [n, 5, 64, 64]) the partial-sum scratch (grip_trunk_backward_parts(n) x 8352 floats) and the "
                         "first layer's gradient outputs are needed");
    if (!obs_dev && !g1m_dev) return grip_fail("grip_trunk_backward: nothing to do with the first layer's gradient (neither observations nor g1m)");
    // GRIP_TRUNK_F32=1 (read once): round 4's kernel -- the data gradients on the fp32 matrix instructions, two 4-wave workgroups per CU -- instead of k_trunk_bwd_b3
    static const bool f32_kernel = [] { const char *e = getenv("GRIP_TRUNK_F32"); return e && e[0] == '1'; }();
    int parts = grip_trunk_backward_parts(n);
    if (f32_kernel) {
        static std::atomic<unsigned long long> mask{0ULL};
        const size_t lds = (size_t)D_LDS_FLOATS * sizeof(float);
        if (set_dyn_lds((const void *)k_trunk_bwd, lds, mask, "grip_trunk_backward")) return -1;
        hipLaunchKernelGGL(k_trunk_bwd, dim3(parts), dim3(256), lds, (hipStream_t)stream, g3_dev, mask3_dev, mask2_dev, mask1_dev, obs_dev, (const long long *)obs_rows_dev, channels, b3_mat_dev, b2_mat_dev, n, g3m_dev, g2m_dev,
                           g1m_dev, obs_dev ? partials_dev : (float *)nullptr);
    } else {
        static std::atomic<unsigned long long> mask{0ULL};
        static std::atomic<int> cus[64];
        if (set_dyn_lds((const void *)k_trunk_bwd_b3, (size_t)E_LDS, mask, "grip_trunk_backward")) return -1;
        int dev = 0, n_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess) return grip_fail("grip_trunk_backward: no current device");
        n_cu = cus[dev & 63].load(std::memory_order_acquire);
        if (n_cu <= 0) {
            if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) return grip_fail("grip_trunk_backward: cannot read the CU count");
            cus[dev & 63].store(n_cu, std::memory_order_release);
        }
        parts = std::min(parts, n_cu);                   // one persistent 8-wave workgroup per CU
        hipLaunchKernelGGL(k_trunk_bwd_b3, dim3(parts), dim3(512), E_LDS, (hipStream_t)stream, g3_dev, mask3_dev, mask2_dev, mask1_dev, obs_dev, (const long long *)obs_rows_dev, channels,
                           reinterpret_cast<const u32x4 *>(b3_mat_dev + (2 * 576 + 864) * 64), reinterpret_cast<const u32x4 *>(b2_mat_dev + (2 * 512 + 768) * 64), n, g3m_dev, g2m_dev, g1m_dev,
                           obs_dev ? partials_dev : (float *)nullptr);
    }
    if (obs_dev)
        hipLaunchKernelGGL(k_wgrad1_reduce, dim3(TB_PART / 32), dim3(1024), 0, (hipStream_t)stream, (const float *)partials_dev, parts, grad_w1_dev, (long long)grad_w1_strides[0],
                           (long long)grad_w1_strides[1], (long long)grad_w1_strides[2], (long long)grad_w1_strides[3], grad_b1_dev, grad_b2_dev, grad_b3_dev);
    return launch_check("grip_trunk_backward");
}

// ------------------------------------------------------------------------------------------------------------------
// k_wgrad23_b3: the weight gradients of the second and third convolution on the bf16 matrix pipe (round 5; until then the tensor library's two kernels, 95 + 55 us per 4096
// samples and a kernel search in the warm-up).
//   dW2[co][ci][ky][kx] = sum_(img, oy, ox) g2m[img, oy, ox, co] * y1[img, 2 oy + ky, 2 ox + kx, ci]        (6 x 6 positions per image)
//   dW3[co][ci][ky][kx] = sum_(img, oy, ox) g3m[img, oy, ox, co] * y2[img, oy + ky, ox + kx, ci]            (4 x 4)
// Each is a GEMM out[m = (tap, ci)][n = co] whose REDUCTION index is the position -- both operands lie in memory position-major ([position][channel], NHWC), the matrix
// instruction wants a lane to hold eight consecutive k of one row / column. gfx950's ds_read_b64_tr_b16 is that transpose: per 16 lanes it reads a block of 4 rows
// (positions; any four addresses) x 16 columns (channels) and hands lane i column i. fp32-equivalent arithmetic as in k_conv23_b3 (grip_policy.hip): both operands split
// into three bf16 terms when they are staged into LDS, six products per k-step, smallest first, fp32 accumulators (v_mfma_f32_32x32x16_bf16).
// One persistent workgroup per CU (8 waves, two per SIMD: while one stages -- 5.5 VALU instructions per value to split, and a wave issues one per ~5 cycles -- or waits for
// LDS, the other's MFMAs run), in one of two ROLES by workgroup index -- the two gradients share no input:
//   role 2 (the first n_wg2 workgroups): a trip = one image PAIR: 72 positions = 4.5 k-chunks of 16 (the last half chunk multiplies zeros). Wave w owns tap row ky = w & 3,
//          columns kx = 2 (w >> 2), + 1: 2 m-tiles (32 input channels each) x 2 n-tiles = 64 accumulator registers. y1 in LDS as [image][pixel row][pixel][term 3][32 ci] bf16, 224-byte pixels,
//          3392-byte rows; g2m as [slot 80][term 3][64 co], 448-byte slots.
//   role 3 (the rest): a trip = four images, one k-chunk each. Wave w owns output channels 32 (w & 1) .., input channels 32 ((w >> 1) & 1) .., taps 0..4 (w < 4) or 5..8.
//          y2 as [image][pixel 36][term 3][64 ci] and g3m as [image][slot 16][term 3][64 co], 448-byte pixels / slots.
// The ORDER of the positions inside a chunk is free (a sum), and chosen so that no transposed read has a bank conflict: a 32-lane half reads 4 positions x 64 bytes, which
// must fall into the four different 64-byte quarters of the 256-byte bank row -- the positions of a 2 x 2 square do (pixel strides = -64, row strides = 128 mod 256), so a
// block of four k is a 2 x 2 square of positions, and the gradient rows are STORED in that order (slot = 4 block + 2 dy + dx). The next trip's operands are requested
// into registers before the MFMA loop. Every workgroup writes its partial sums; k_wgrad23_reduce adds them in a fixed order (no atomics) into the gradient tensors.
#define W2_PS 224
#define W2_RSY (15 * W2_PS + 32)
#define W2_IMG (15 * W2_RSY)
#define W2_GOFF (2 * W2_IMG)
#define W_RS 448
#define W2_LDS (W2_GOFF + 80 * W_RS)
#define W3_NI 4
#define W3_IMG (36 * W_RS)
#define W3_GOFF (W3_NI * W3_IMG)
#define W3_GIMG (16 * W_RS)
#define W3_LDS (W3_GOFF + W3_NI * W3_GIMG)
#define W2_PART (512 * 64)
#define W3_PART (576 * 64)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char lds_byte;
// eight consecutive k of the lane's row / column: two transposed reads (4 positions each)
__device__ __forceinline__ bf16x8 wg_tr8(lds_byte *p0, lds_byte *p1, int off) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(p0 + off));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3))) *)(p1 + off));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ f32x16 wg_mac6(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);
    return c;
}
// four fp32 values -> three bf16 terms each (nearest, exact remainder, twice), written as three 8-byte words `stride` bytes apart
__device__ __forceinline__ void wg_split_store(unsigned char *dst, int stride, const float4 v) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    bf16x4 h, m, l;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const __bf16 a = (__bf16)x[i]; const float r = x[i] - (float)a;
        const __bf16 b = (__bf16)r; const float r2 = r - (float)b;
        h[i] = a; m[i] = b; l[i] = (__bf16)r2;
    }
    *reinterpret_cast<bf16x4 *>(dst) = h; *reinterpret_cast<bf16x4 *>(dst + stride) = m; *reinterpret_cast<bf16x4 *>(dst + 2 * stride) = l;
}

#ifdef WG_STAMPS                 // diagnostic build only (tools/wgrad23_stamps.py): per role, cycles of wave 0 of the role's first workgroup per phase; every workgroup's start / end
__device__ unsigned long long g_wg_stamps[2][4];       // [role][prologue, staging (wait + split + store), barrier + requests, MFMA loop + barrier]
__device__ unsigned long long g_wg_span[512][2];
#define WG_STAMP(role, i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); if (first_of_role && threadIdx.x == 0) g_wg_stamps[role][i] += n_ - wg_t; wg_t = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int grip_debug_wg_stamps(unsigned long long *out8, unsigned long long *span1024) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_wg_stamps), sizeof(unsigned long long) * 8) != hipSuccess || hipMemcpyFromSymbol(span1024, HIP_SYMBOL(g_wg_span), sizeof(unsigned long long) * 1024) != hipSuccess) return -1;
    static const unsigned long long zero[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wg_stamps), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#else
#define WG_STAMP(role, i) do { } while (0)
#endif
__global__ void __launch_bounds__(512, 1) k_wgrad23_b3(const float *__restrict__ y1, const float *__restrict__ g2m, const float *__restrict__ y2, const float *__restrict__ g3m, int n_img,
                                                       int n_wg2, float *__restrict__ part2, float *__restrict__ part3) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
#ifdef WG_STAMPS
    unsigned long long wg_t = __builtin_amdgcn_s_memtime();
    const bool first_of_role = blockIdx.x == 0 || (int)blockIdx.x == n_wg2;
    if (threadIdx.x == 0 && blockIdx.x < 512) g_wg_span[blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int hf = l >> 5, gsel = (l >> 4) & 1, q = (l & 15) >> 2, p = l & 3;           // the lane in a transposed read: k-half, column half, the block's row it addresses, its 8-byte piece
    // zeros everywhere once: the padding slots of role 2's last half chunk stay zero, and nothing that is not a number is ever multiplied by them
    for (int i = tid; i < W2_LDS / 16; i += 512) reinterpret_cast<uint4 *>(wl)[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    lds_byte *L = (lds_byte *)wl;
    if ((int)blockIdx.x < n_wg2) {
        const int npair = (n_img + 1) / 2, ky = w & 3, kx0 = 2 * (w >> 2);
        // the lane's addresses in y1 for chunk c, half-fragment h: block b = 4 c + 2 hf + h = (image, 2 x 2 square), row q of it = position (2 sy + (q >> 1), 2 sx + (q & 1))
        lds_byte *pa[5][2];
#pragma unroll
        for (int c = 0; c < 5; c++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                int b = 4 * c + 2 * hf + h;
                if (b >= 18) b = 0;                                     // the half chunk of padding: any real pixels (their gradient rows are zero)
                const int g = b / 9, sq = b - 9 * g, oy = 2 * (sq / 3) + (q >> 1), ox = 2 * (sq % 3) + (q & 1);
                pa[c][h] = L + g * W2_IMG + (2 * oy + ky) * W2_RSY + (2 * ox + kx0) * W2_PS + 32 * gsel + 8 * p;
            }
        lds_byte *pb = L + W2_GOFF + (8 * hf + q) * W_RS + 32 * gsel + 8 * p;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
        // a pair's operands: 3600 + 1152 float4 = 8 + 3 requests per thread, issued in five parts between the chunks of the MFMA loop (the address path takes ~16 cycles
        // per request and wave: issued in one piece they held the matrix pipe up for 1.8 k cycles per trip)
        float4 pre[8], preg[3];
        auto request = [&](int pair, int part) {
            const int img0 = pair * 2, nimg = min(2, n_img - img0);
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (u * 5 / 8 == part) {
                    const int qq = u * 512 + tid;
                    pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (qq < nimg * 1800) pre[u] = *reinterpret_cast<const float4 *>(y1 + (size_t)img0 * 7200 + (size_t)qq * 4);
                }
#pragma unroll
            for (int u = 0; u < 3; u++)
                if (u + 2 == part) {
                    const int qq = u * 512 + tid;
                    preg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (qq < nimg * 576) preg[u] = *reinterpret_cast<const float4 *>(g2m + (size_t)img0 * 2304 + (size_t)qq * 4);
                }
        };
        int pair = blockIdx.x;
        if (pair < npair) {
#pragma unroll
            for (int part = 0; part < 5; part++) request(pair, part);
        }
        WG_STAMP(0, 0);
        for (; pair < npair; pair += n_wg2) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int qq = u * 512 + tid;
                if (qq < 3600) {
                    const int g = qq >= 1800, rem = qq - 1800 * g, pix = rem >> 3, c4 = (rem & 7) * 4, Y = pix / 15, X = pix - 15 * Y;
                    wg_split_store(wl + g * W2_IMG + Y * W2_RSY + X * W2_PS + 2 * c4, 64, pre[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 3; u++) {
                const int qq = u * 512 + tid;
                if (qq < 1152) {
                    const int pos = qq >> 4, c4 = (qq & 15) * 4, g = pos >= 36, pp = pos - 36 * g, oy = pp / 6, ox = pp - 6 * oy;
                    const int slot = (g * 9 + (oy >> 1) * 3 + (ox >> 1)) * 4 + (oy & 1) * 2 + (ox & 1);
                    wg_split_store(wl + W2_GOFF + slot * W_RS + 2 * c4, 128, preg[u]);
                }
            }
            WG_STAMP(0, 1);
            __syncthreads();
            WG_STAMP(0, 2);
            const bool more = pair + n_wg2 < npair;
#pragma unroll
            for (int c = 0; c < 5; c++) {
                if (more) request(pair + n_wg2, c);
                bf16x8 bfr[2][3];
#pragma unroll
                for (int nt = 0; nt < 2; nt++)
#pragma unroll
                    for (int t = 0; t < 3; t++) bfr[nt][t] = wg_tr8(pb, pb + 4 * W_RS, 16 * c * W_RS + nt * 64 + t * 128);
#pragma unroll
                for (int kx = 0; kx < 2; kx++) {
                    bf16x8 afr[3];
#pragma unroll
                    for (int t = 0; t < 3; t++) afr[t] = wg_tr8(pa[c][0], pa[c][1], kx * W2_PS + 64 * t);
                    acc[kx][0] = wg_mac6(afr, bfr[0], acc[kx][0]);
                    acc[kx][1] = wg_mac6(afr, bfr[1], acc[kx][1]);
                }
            }
            __syncthreads();                                            // before the next trip's staging overwrites the planes
            WG_STAMP(0, 3);
        }
        float *P = part2 + (size_t)blockIdx.x * W2_PART;
#pragma unroll
        for (int kx = 0; kx < 2; kx++)
#pragma unroll
            for (int nt = 0; nt < 2; nt++)
#pragma unroll
                for (int r = 0; r < 16; r++) P[((ky * 4 + kx0 + kx) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf) * 64 + nt * 32 + (l & 31)] = acc[kx][nt][r];
    } else {
        const int wg = blockIdx.x - n_wg2, n_wg3 = gridDim.x - n_wg2, ngrp = (n_img + W3_NI - 1) / W3_NI;
        const int nt = w & 1, cih = (w >> 1) & 1, tap0 = 5 * (w >> 2), ntap = 5 - (w >> 2);     // waves 0-3: taps 0..4, waves 4-7 (the same SIMDs): taps 5..8
        lds_byte *pa[5][2];
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int tap = min(tap0 + i, 8), b = 2 * hf + h, oy = 2 * (b >> 1) + (q >> 1) + tap / 3, ox = 2 * (b & 1) + (q & 1) + tap % 3;
                pa[i][h] = L + (oy * 6 + ox) * W_RS + 64 * cih + 32 * gsel + 8 * p;
            }
        lds_byte *pb = L + W3_GOFF + (8 * hf + q) * W_RS + 64 * nt + 32 * gsel + 8 * p;
        f32x16 acc[5];
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
        float4 pre[5], preg[2];                                          // four images' operands: 2304 + 1024 float4
        auto request = [&](int grp, int part) {
            const int img0 = grp * W3_NI, nimg = min(W3_NI, n_img - img0);
#pragma unroll
            for (int u = 0; u < 5; u++)
                if (u * 4 / 5 == part) {
                    const int qq = u * 512 + tid;
                    pre[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (qq < nimg * 576) pre[u] = *reinterpret_cast<const float4 *>(y2 + (size_t)img0 * 2304 + (size_t)qq * 4);
                }
#pragma unroll
            for (int u = 0; u < 2; u++)
                if (u + 2 == part) {
                    const int qq = u * 512 + tid;
                    preg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (qq < nimg * 256) preg[u] = *reinterpret_cast<const float4 *>(g3m + (size_t)img0 * 1024 + (size_t)qq * 4);
                }
        };
        int grp = wg;
        if (grp < ngrp) {
#pragma unroll
            for (int part = 0; part < 4; part++) request(grp, part);
        }
        WG_STAMP(1, 0);
        for (; grp < ngrp; grp += n_wg3) {
#pragma unroll
            for (int u = 0; u < 5; u++) {
                const int qq = u * 512 + tid;
                if (qq < 2304) {
                    const int g = qq / 576, rem = qq - 576 * g, pix = rem >> 4, c4 = (rem & 15) * 4;
                    wg_split_store(wl + g * W3_IMG + pix * W_RS + 2 * c4, 128, pre[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int qq = u * 512 + tid, g = qq >> 8, pos = (qq & 255) >> 4, c4 = (qq & 15) * 4, oy = pos >> 2, ox = pos & 3;
                const int slot = ((oy >> 1) * 2 + (ox >> 1)) * 4 + (oy & 1) * 2 + (ox & 1);
                wg_split_store(wl + W3_GOFF + g * W3_GIMG + slot * W_RS + 2 * c4, 128, preg[u]);
            }
            WG_STAMP(1, 1);
            __syncthreads();
            WG_STAMP(1, 2);
            const bool more = grp + n_wg3 < ngrp;
#pragma unroll
            for (int g = 0; g < W3_NI; g++) {
                if (more) request(grp + n_wg3, g);
                bf16x8 bfr[3];
#pragma unroll
                for (int t = 0; t < 3; t++) bfr[t] = wg_tr8(pb, pb + 4 * W_RS, g * W3_GIMG + 128 * t);
#pragma unroll
                for (int i = 0; i < 5; i++)
                    if (i < ntap) {                                      // (wave-uniform: the transposed reads need every lane)
                        bf16x8 afr[3];
#pragma unroll
                        for (int t = 0; t < 3; t++) afr[t] = wg_tr8(pa[i][0], pa[i][1], g * W3_IMG + 128 * t);
                        acc[i] = wg_mac6(afr, bfr, acc[i]);
                    }
            }
            __syncthreads();
            WG_STAMP(1, 3);
        }
        float *P = part3 + (size_t)wg * W3_PART;
#pragma unroll
        for (int i = 0; i < 5; i++)
            if (i < ntap) {
#pragma unroll
                for (int r = 0; r < 16; r++) P[((tap0 + i) * 64 + 32 * cih + (r & 3) + 8 * (r >> 2) + 4 * hf) * 64 + 32 * nt + (l & 31)] = acc[i][r];
            }
    }
#ifdef WG_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 512) g_wg_span[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ------------------------------------------------------------------------------------------------------------------
// k_trunk_bwd_b3 (round 5): k_trunk_bwd with the two data-gradient GEMMs on the bf16 matrix pipe -- gradient tile and (transposed) weights as three bf16 terms, six products
// per k-step, fp32 accumulators: 2 x 6 v_mfma_f32_16x16x32_bf16 (16 cycles each) per 16 x 16 tile and column block where the fp32 kernel issues 16 v_mfma_f32_16x16x4_f32
// (32 cycles each): 2.7 x less matrix-pipe time, which was 58 % of that kernel. What changes around them:
//   * one persistent workgroup of EIGHT waves per CU (two per SIMD, as the two 4-wave workgroups were). GEMM 1: wave = (input-channel quarter, image); GEMM 2: wave =
//     (parity class of the target pixel, input-channel half) -- as before no two waves ever add into the same word of the tile below;
//   * the masked gradient tiles are split ONCE, by the mask passes, into [row][term 3][64 channels] bf16 planes (416-byte rows: the A fragments' ds_read_b128 are free of bank
//     conflicts), GEMM 1's A fragments stay in registers, GEMM 2's five tiles are re-read per column block; the weights' fragments (k_conv23_prep: B3d, B2d) come from L2,
//     one column block ahead; the scatter into the fp32 tile below is k_trunk_bwd's (same accumulator layout), between the two k-steps of the next block;
//   * the first layer's weight gradient: the masked g1 of ONE image is split once into [row of 16 positions][term][32 channels] planes (in the space of the dead g2 tile and
//     planes) and every wave reads its B fragments from there with the transposed read (ds_read_b64_tr_b16: positions x channels -> channels x positions) -- the fp32
//     kernel's waves each re-split the tile, ~70 VALU instructions per 16 positions, the bound of that loop. Wave = (byte plane, half of the 8 x 8 patch): one 32 x 32 tile.
__device__ __forceinline__ f32x4 tb_mac6(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[2]), __builtin_bit_cast(bf16x8, b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[2]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[1]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, b[0]), c, 0, 0, 0);
    return c;
}
// MT row tiles x NB column blocks, K = 64 channels = two k-steps. TA: the A planes (bytes), arow[t] the lane's row of tile t (+ 64 k-step + 128 term); Bf: the lane's
// fragment pointer, bidx(nb) the block's first fragment ([k-step 2][term 3] follow, 64 lanes apart); T, tgt, live: as scatter_gemm. HOLD: A fragments stay in registers.
template <int MT, int NB, bool HOLD, int PF, class BIdx, class Tgt, class Live>
__device__ __forceinline__ void scatter_gemm_b3(const unsigned char *TA, const int (&arow)[MT], const u32x4 *__restrict__ Bf, float *T, BIdx bidx, Tgt tgt, Live live) {
    // the weights' fragments come from L2 (~1 k cycles away) into a ring PF column blocks deep; the A fragments of a tile are requested a tile ahead of their MFMAs
    u32x4 ring[PF][2][3], ah[HOLD ? MT : 1][2][3];
    auto fetch_b = [&](int nb) {
#pragma unroll
        for (int ks = 0; ks < 2; ks++)
#pragma unroll
            for (int tm = 0; tm < 3; tm++) ring[nb % PF][ks][tm] = Bf[(size_t)(bidx(nb) + ks * 3 + tm) * 64];
    };
    auto fetch_a = [&](int step, u32x4 (&a)[3]) {                          // step = k-step * MT + tile
#pragma unroll
        for (int tm = 0; tm < 3; tm++) a[tm] = *reinterpret_cast<const u32x4 *>(TA + arow[step % MT] + 64 * (step / MT) + 128 * tm);
    };
#pragma unroll
    for (int nb = 0; nb < PF && nb < NB; nb++) fetch_b(nb);
    if (HOLD) {
#pragma unroll
        for (int st = 0; st < 2 * MT; st++) fetch_a(st, ah[st % MT][st / MT]);
    }
    f32x4 prev[MT];
#pragma unroll
    for (int nb = 0; nb <= NB; nb++) {
        float old[MT][4];
        u32x4 acur[3], anext[3];
        if (nb > 0) {
#pragma unroll
            for (int t = 0; t < MT; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) old[t][r] = T[tgt(nb - 1, t, r)];
        }
        if (!HOLD && nb < NB) fetch_a(0, acur);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[MT];
#pragma unroll
        for (int t = 0; t < MT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < 2 * MT; st++) {
            const int ks = st / MT, t = st % MT;
            if (nb < NB) {
                if (HOLD) acc[t] = tb_mac6(ah[t][ks], ring[nb % PF][ks], acc[t]);
                else {
                    if (st + 1 < 2 * MT) fetch_a(st + 1, anext);
                    acc[t] = tb_mac6(acur, ring[nb % PF][ks], acc[t]);
#pragma unroll
                    for (int tm = 0; tm < 3; tm++) acur[tm] = anext[tm];
                }
            }
            if (st == MT - 1 && nb > 0) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t2 = 0; t2 < MT; t2++)
                    if (live(t2)) {
#pragma unroll
                        for (int r = 0; r < 4; r++) T[tgt(nb - 1, t2, r)] = old[t2][r] + prev[t2][r];
                    }
                wave_order();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nb < NB) {
            if (nb + PF < NB) fetch_b(nb + PF);
#pragma unroll
            for (int t = 0; t < MT; t++) prev[t] = acc[t];
        }
    }
}

#ifdef TB_STAMPS                 // diagnostic build only (tools/trunk_stamps.py): cycles of wave 0 of workgroup 0 per phase
__device__ unsigned long long g_tb_stamps[8];
#define TB_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); if (blockIdx.x == 0 && threadIdx.x == 0) g_tb_stamps[i] += n_ - tb_t; tb_t = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int grip_debug_tb_stamps(unsigned long long *out8) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tb_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    static const unsigned long long zero[8] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tb_stamps), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#else
#define TB_STAMP(i) do { } while (0)
#endif
struct GroupIn3 { float4 g3; unsigned m3; unsigned long long m2; unsigned m1; };           // a thread's share (of 512) of a group's gradient tile and ReLU masks
__device__ __forceinline__ void group_in_load3(GroupIn3 &gi, int tl, int img0, int nimg, const float *__restrict__ g3, const uint16_t *__restrict__ m3h,
                                               const unsigned long long *__restrict__ m2, const uint32_t *__restrict__ m1) {
    const int row = tl >> 4, c4 = (tl & 15) * 4;
    gi.g3 = make_float4(0.f, 0.f, 0.f, 0.f); gi.m3 = 0u;
    if ((row >> 4) < nimg) {
        gi.g3 = *reinterpret_cast<const float4 *>(g3 + ((size_t)img0 * 16 + row) * 64 + c4);
        gi.m3 = (unsigned)m3h[((size_t)img0 * 16 + row) * 4 + (c4 >> 4)] >> (c4 & 15);
    }
    gi.m1 = tl < nimg * 225 ? m1[(size_t)img0 * 225 + tl] : 0u;
    gi.m2 = tl < nimg * 36 ? m2[(size_t)img0 * 36 + tl] : 0ULL;
}

__global__ void __launch_bounds__(512, 1) k_trunk_bwd_b3(const float *__restrict__ g3, const uint64_t *__restrict__ m3, const uint64_t *__restrict__ m2, const uint32_t *__restrict__ m1,
                                                         const uint8_t *__restrict__ obs, const long long *__restrict__ obs_rows, int channels, const u32x4 *__restrict__ B3d,
                                                         const u32x4 *__restrict__ B2d, int n_img, float *__restrict__ g3m_out, float *__restrict__ g2m_out, float *__restrict__ g1m_out,
                                                         float *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char e_lds[];
    float *T1 = reinterpret_cast<float *>(e_lds), *T2 = reinterpret_cast<float *>(e_lds + E_T2);
    unsigned char *A3 = e_lds + E_A3, *A2 = e_lds + E_A2, *U = e_lds + E_U, *G1P = e_lds + E_G1P;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, r16 = l & 15, kq = l >> 4;
    const int half = l >> 5, m32 = l & 31, gsel = (l >> 4) & 1, q4 = (l & 15) >> 2, p4 = l & 3;      // lane roles in the 32 x 32 x 16 instruction and in its transposed read
    f32x16 cw;
#pragma unroll
    for (int r = 0; r < 16; r++) cw[r] = 0.f;
    float4 bs1 = make_float4(0.f, 0.f, 0.f, 0.f), bs2 = bs1, bs3 = bs1;      // bias gradients: the thread's four channels of each layer, over its rows
    const int ngroups = (n_img + DG - 1) / DG;
    // rows 72..79 of the g2m planes (the half tile of padding) are zeros for the whole launch: nothing else is ever written there
    for (int i = tid; i < 8 * E_RS / 16; i += 512) reinterpret_cast<uint4 *>(A2 + 72 * E_RS)[i] = make_uint4(0u, 0u, 0u, 0u);
#ifdef TB_STAMPS
    unsigned long long tb_t = __builtin_amdgcn_s_memtime();
#endif
    GroupIn3 gi;
    if ((int)blockIdx.x < ngroups) group_in_load3(gi, tid, blockIdx.x * DG, min(DG, n_img - (int)blockIdx.x * DG), g3, reinterpret_cast<const uint16_t *>(m3),
                                                  reinterpret_cast<const unsigned long long *>(m2), m1);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int img0 = grp * DG, nimg = min(DG, n_img - img0);
    const u32x4 *B3g = B3d, *B2g = B2d;                                    // (hidden from the optimiser, as in k_trunk_bwd: loop-invariant loads and addresses would be hoisted and spilled)
    asm volatile("" : "+s"(B3g), "+s"(B2g));
    int tl = tid;
    asm volatile("" : "+v"(tl));
    // ---- g3m = g3 * mask3 -> memory (the third layer's weight gradient reads it) and, split, the A planes of GEMM 1; zero T2
    {
        const int row = tl >> 4, c4 = (tl & 15) * 4;
        const float4 v = bit_mask(gi.m3, gi.g3);
        bs3.x += v.x; bs3.y += v.y; bs3.z += v.z; bs3.w += v.w;
        if (g3m_out && (row >> 4) < nimg) *reinterpret_cast<float4 *>(g3m_out + ((size_t)img0 * 16 + row) * 64 + c4) = v;
        wg_split_store(A3 + row * E_RS + 2 * c4, 128, v);
    }
#pragma unroll 1
    for (int i = tl; i < D_T2 / 4; i += 512) reinterpret_cast<float4 *>(T2)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tl < DG * 36) *reinterpret_cast<unsigned long long *>(T2 + tl * D_PS2 + 64) = gi.m2;      // the rows' ReLU masks ride in their padding words
    TB_STAMP(0);            // g3m pass, zero fill, barrier (+ loop top)
    // ---- GEMM 1: this wave's image (16 positions = one row tile), 9 column blocks (taps) of its 16 input channels
    {
        const int cb = w & 3, img = w >> 2;
        const int arow[1] = {(img * 16 + r16) * E_RS + 16 * kq};
        const int tb = kq * 6 * D_PS2 + 16 * cb + r16;                      // the lane holds rows 4 kq + r: (oy, ox) = (kq, r)
        scatter_gemm_b3<1, 9, true, 1>(A3, arow, B3g + l, T2, [&](int c) { return ((c * 4 + cb) * 2) * 3; },
                                    [&](int c, int, int r) { return tb + (img * 36 + (c / 3) * 6 + r + (c % 3)) * D_PS2; }, [](int) { return true; });
    }
    __syncthreads();
    TB_STAMP(1);            // GEMM 1 + barrier
    // ---- g2m = g2 * mask2: to memory (the second layer's weight gradient reads it) and, split, the A planes of GEMM 2; zero T1
#pragma unroll
    for (int u = 0; u < 3; u++) {
        const int i = tl + 512 * u, row = i >> 4, c4 = (i & 15) * 4;
        if (i < DG * 36 * 16) {
            const unsigned bits = reinterpret_cast<const unsigned *>(T2)[row * D_PS2 + 64 + (c4 >> 5)] >> (c4 & 31);
            const float4 v = bit_mask(bits, *reinterpret_cast<float4 *>(T2 + row * D_PS2 + c4));
            bs2.x += v.x; bs2.y += v.y; bs2.z += v.z; bs2.w += v.w;
            if (row < nimg * 36) *reinterpret_cast<float4 *>(g2m_out + ((size_t)img0 * 36 + row) * 64 + c4) = v;
            wg_split_store(A2 + row * E_RS + 2 * c4, 128, v);
        }
    }
#pragma unroll 1
    for (int i = tl; i < D_T1 / 4; i += 512) reinterpret_cast<float4 *>(T1)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    if (tl < DG * 225) reinterpret_cast<unsigned *>(T1)[tl * D_PS1 + 32] = gi.m1;
    TB_STAMP(2);            // g2m pass, zero fill, barrier
    // ---- GEMM 2: rows m = image * 36 + position (80 = 5 tiles, rows 72.. are zeros), this wave's 4 taps (one parity class) of its 16 input channels
    {
        const int par = w & 3, py = par >> 1, px = par & 1, chh = w >> 2;
        int arow[5], base[5][4];
#pragma unroll
        for (int t = 0; t < 5; t++) {
            arow[t] = (t * 16 + r16) * E_RS + 16 * kq;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int mo = min(t * 16 + 4 * kq + r, DG * 36 - 1), g = mo / 36, p = mo - g * 36, oy = p / 6, ox = p - oy * 6;
                base[t][r] = (g * 225 + (2 * oy + py) * 15 + 2 * ox + px) * D_PS1 + 16 * chh + r16;
            }
        }
        const bool pad_ok = kq < 2;                     // tile 4: rows 64 + 4 kq + r exist for kq < 2 only
        // column block a: tap (ky, kx) = (py + 2 (a >> 1), px + 2 (a & 1))
        scatter_gemm_b3<5, 4, false, 1>(A2, arow, B2g + l, T1, [&](int a) { return ((((py + 2 * (a >> 1)) * 4 + px + 2 * (a & 1)) * 2 + chh) * 2) * 3; },
                                     [&](int a, int t, int r) { return base[t][r] + ((a >> 1) * 30 + 2 * (a & 1)) * D_PS1; }, [&](int t) { return t < 4 || pad_ok; });
    }
    __syncthreads();                                                        // T1 complete; T2 and the planes are free
    TB_STAMP(3);            // GEMM 2 + barrier
    // ---- g1m = g1 * mask1 (to memory when asked for). With observations the masked tile is consumed on chip, image by image, while the byte planes are on their way:
    // the first layer's weight gradient, K = an image's 225 positions in 15 chunks of 16 (one row oy; position 15 of a row does not exist: zeros), A = the image bytes
    // (a pixel 0..255 is a bf16), B = the masked gradient as three bf16 terms, split ONCE per image into planes every wave reads by the transposed read
    const int gnext = grp + gridDim.x;
    auto next_group = [&]() {
        if (gnext < ngroups)
            group_in_load3(gi, tl, gnext * DG, min(DG, n_img - gnext * DG), g3, reinterpret_cast<const uint16_t *>(m3), reinterpret_cast<const unsigned long long *>(m2), m1);
    };
    auto masked = [&](int pix, int c4) {                                    // four channels of a pixel of the g1 tile, masked; db1 takes them (c4 is the same in every pass: 512 % 8 == 0)
        const float2 *tp = reinterpret_cast<const float2 *>(T1 + pix * D_PS1 + c4);
        const float2 v0 = tp[0], v1 = tp[1];
        const float4 v = bit_mask(reinterpret_cast<const unsigned *>(T1)[pix * D_PS1 + 32] >> c4, make_float4(v0.x, v0.y, v1.x, v1.y));
        bs1.x += v.x; bs1.y += v.y; bs1.z += v.z; bs1.w += v.w;
        return v;
    };
    if (obs) {
        uint4 pa, pb;
        {   const uint4 *src = reinterpret_cast<const uint4 *>(obs + (size_t)(obs_rows ? obs_rows[img0] : img0) * channels * 4096) + tl;
            pa = src[0]; pb = src[512]; }
        const int plane = w & 3, tile = w >> 2;
        const uint8_t *ap0 = U + plane * 4096 + ((m32 >> 3) + 4 * tile) * 64 + (m32 & 7) + half * 32;       // + oy * 256 + 4 j
        lds_byte *bp0 = (lds_byte *)G1P + (8 * half + q4) * 192 + 32 * gsel + 8 * p4;                       // + oy * 3072 + 64 term; the second four positions: + 768
#pragma unroll 1
        for (int g = 0; g < nimg; g++) {
            if (g > 0) __syncthreads();                                     // every wave is done with the last image's planes
#pragma unroll 1
            for (int i = tl; i < 225 * 8; i += 512) {
                const int pix = i >> 3, c4 = (i & 7) * 4, oy = pix / 15, ox = pix - 15 * oy;
                const float4 v = masked(g * 225 + pix, c4);
                if (g1m_out) *reinterpret_cast<float4 *>(g1m_out + ((size_t)(img0 + g) * 225 + pix) * 32 + c4) = v;
                wg_split_store(G1P + (oy * 16 + ox) * 192 + 2 * c4, 64, v);
            }
            if (tl < 15 * 12) reinterpret_cast<uint4 *>(G1P + ((tl / 12) * 16 + 15) * 192)[tl % 12] = make_uint4(0u, 0u, 0u, 0u);
            uint4 *Uq = reinterpret_cast<uint4 *>(U) + tl;
            Uq[0] = pa; Uq[512] = pb;
            __syncthreads();
            TB_STAMP(4);        // barrier, the image's mask + split pass, byte planes -> LDS, barrier
            // requests that fly during the K loop: the next image's planes, or the next group's gradient tile and masks
            if (g + 1 < nimg) {
                const uint4 *src = reinterpret_cast<const uint4 *>(obs + (size_t)(obs_rows ? obs_rows[img0 + g + 1] : img0 + g + 1) * channels * 4096) + tl;
                pa = src[0]; pb = src[512];
            } else next_group();
            const uint8_t *ap = ap0;
            lds_byte *bp = bp0;
#pragma unroll 3
            for (int oy = 0; oy < 15; oy++) {
                bf16x8 b[3], a;
#pragma unroll
                for (int tm = 0; tm < 3; tm++) b[tm] = wg_tr8(bp, bp + 768, 64 * tm);
#pragma unroll
                for (int j = 0; j < 8; j++) a[j] = (__bf16)(float)ap[4 * j];
                cw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[2], cw, 0, 0, 0);
                cw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[1], cw, 0, 0, 0);
                cw = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[0], cw, 0, 0, 0);
                ap += 256; bp += 3072;
            }
            TB_STAMP(5);        // requests + the K loop
        }
    } else {
#pragma unroll 2
        for (int i = tl; i < DG * 225 * 8; i += 512) {
            const float4 v = masked(i >> 3, (i & 7) * 4);
            if (g1m_out && i < nimg * 225 * 8) *reinterpret_cast<float4 *>(g1m_out + (size_t)img0 * 7200 + (size_t)i * 4) = v;
        }
        next_group();
    }
    __syncthreads();                                                        // before the next group's staging
  }
    if (part) {
        float *P = part + (size_t)blockIdx.x * TB_PART;
        const int plane = w & 3, tile = w >> 2;
#pragma unroll
        for (int r = 0; r < 16; r++) P[(plane * 64 + tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + m32] = cw[r];
        // db1: thread tid holds channels 4 (tid & 7) .. + 3 summed over its pixels: 64 threads per channel group, added in a fixed order through LDS
        float *red = reinterpret_cast<float *>(e_lds);
        __syncthreads();
        *reinterpret_cast<float4 *>(red + tid * 4) = bs1;
        __syncthreads();
        if (tid < 32) {
            float t = 0.f;
            for (int k = 0; k < 64; k++) t += red[((tid >> 2) + 8 * k) * 4 + (tid & 3)];
            P[256 * 32 + tid] = t;
        }
        // db2, db3: channels 4 (tid & 15) .. + 3, 32 threads per channel group
        __syncthreads();
        *reinterpret_cast<float4 *>(red + tid * 4) = bs2; *reinterpret_cast<float4 *>(red + 2048 + tid * 4) = bs3;
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, co = tid & 63;
            float t = 0.f;
            for (int k = 0; k < 32; k++) t += red[which * 2048 + ((co >> 2) + 16 * k) * 4 + (co & 3)];
            P[256 * 32 + 32 + tid] = t;
        }
    }
}

// the workgroups' partial sums -> the two gradient tensors (their strides), added in a fixed order: 32 outputs per block, 32 slices of the partials each
__global__ void __launch_bounds__(1024) k_wgrad23_reduce(const float *__restrict__ part2, int n2, const float *__restrict__ part3, int n3, float *__restrict__ gw2, long long so2, long long sc2,
                                                          long long sy2, long long sx2, float *__restrict__ gw3, long long so3, long long sc3, long long sy3, long long sx3) {
    __shared__ float red[32][32];
    const int o = blockIdx.x * 32 + (threadIdx.x & 31), sl = threadIdx.x >> 5;
    const bool second = o < W2_PART;
    const float *base = second ? part2 + o : part3 + (o - W2_PART);
    const size_t sz = second ? W2_PART : W3_PART;
    const int np = second ? n2 : n3;
    float s = 0.f;
    for (int pi = sl; pi < np; pi += 32) s += base[(size_t)pi * sz];
    red[sl][threadIdx.x & 31] = s;
    __syncthreads();
    if (sl == 0) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int i = 1; i < 32; i++) t += red[i][threadIdx.x];
        if (second) {
            const int m = o >> 6, co = o & 63, tap = m >> 5, ci = m & 31;
            gw2[co * so2 + ci * sc2 + (tap >> 2) * sy2 + (tap & 3) * sx2] = t;
        } else {
            const int oo = o - W2_PART, m = oo >> 6, co = oo & 63, tap = m >> 6, ci = m & 63;
            gw3[co * so3 + ci * sc3 + (tap / 3) * sy3 + (tap % 3) * sx3] = t;
        }
    }
}

// How the CUs are divided between the two roles: a trip of role 2 (an image pair) costs ~W_COST2, of role 3 (four images) ~W_COST3 (x 10 cycles; tools/wgrad23_stamps.py
// prints the measured trips); the split that minimises the slower role's trips x cost. The sums' order, and with it the last bits of the result, depends on it -- a function of
// (n, CU count) only.
#define W_COST2 1286
#define W_COST3 1177
static void wgrad23_split(int n, int n_cu, int &w2, int &w3) {
    const int pairs = (n + 1) / 2, grps = (n + W3_NI - 1) / W3_NI;
    long best = -1; w2 = 1; w3 = 1;
    for (int a = 1; a < n_cu; a++) {
        const int b = n_cu - a;
        const long t = std::max((long)((pairs + a - 1) / a) * W_COST2, (long)((grps + b - 1) / b) * W_COST3);
        if (best < 0 || t < best) { best = t; w2 = a; w3 = b; }
    }
    w2 = std::min(w2, pairs); w3 = std::min(w3, grps);
}
static int wgrad23_cus(int &n_cu) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return grip_fail("grip_wgrad23: no current device");
    static std::atomic<int> cus[64];
    int c = cus[dev & 63].load(std::memory_order_acquire);
    if (c <= 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 1) return grip_fail("grip_wgrad23: cannot read the CU count");
        cus[dev & 63].store(c, std::memory_order_release);
    }
    n_cu = c;
    return 0;
}
extern "C" long long grip_wgrad23_scratch_floats(int n) {
    int n_cu = 0, w2, w3;
    if (n <= 0 || wgrad23_cus(n_cu)) return -1;
    wgrad23_split(n, n_cu, w2, w3);
    return (long long)w2 * W2_PART + (long long)w3 * W3_PART;
}
extern "C" int grip_wgrad23(const float *y1_dev, const float *g2m_dev, const float *y2_dev, const float *g3m_dev, int n, float *scratch_dev, float *grad_w2_dev, const int64_t *grad_w2_strides,
                            float *grad_w3_dev, const int64_t *grad_w3_strides, void *stream) {
    if (!y1_dev || !g2m_dev || !y2_dev || !g3m_dev || !scratch_dev || !grad_w2_dev || !grad_w2_strides || !grad_w3_dev || !grad_w3_strides || n <= 0)
        return grip_fail("grip_wgrad23: need y1 [n, 15, 15, 32], g2m [n, 6, 6, 64], y2 [n, 6, 6, 64], g3m [n, 4, 4, 64] (NHWC float32), the scratch of grip_wgrad23_scratch_floats(n) "
                         "floats and the two gradient tensors with their strides");
    int n_cu = 0, w2, w3;
    if (wgrad23_cus(n_cu)) return -1;
    wgrad23_split(n, n_cu, w2, w3);
    static std::atomic<unsigned long long> mask{0ULL};
    if (set_dyn_lds((const void *)k_wgrad23_b3, (size_t)W2_LDS, mask, "grip_wgrad23")) return -1;
    float *part2 = scratch_dev, *part3 = scratch_dev + (size_t)w2 * W2_PART;
    hipLaunchKernelGGL(k_wgrad23_b3, dim3(w2 + w3), dim3(512), W2_LDS, (hipStream_t)stream, y1_dev, g2m_dev, y2_dev, g3m_dev, n, w2, part2, part3);
    hipLaunchKernelGGL(k_wgrad23_reduce, dim3((W2_PART + W3_PART) / 32), dim3(1024), 0, (hipStream_t)stream, (const float *)part2, w2, (const float *)part3, w3, grad_w2_dev,
                       (long long)grad_w2_strides[0], (long long)grad_w2_strides[1], (long long)grad_w2_strides[2], (long long)grad_w2_strides[3], grad_w3_dev,
                       (long long)grad_w3_strides[0], (long long)grad_w3_strides[1], (long long)grad_w3_strides[2], (long long)grad_w3_strides[3]);
    return launch_check("grip_wgrad23");
}

// ------------------------------------------------------------------------------------------------------------------
// Gradient clipping + Adam for the whole parameter list in two launches (stable_baselines3 PPO.train: clip_grad_norm_(max_grad_norm) then Adam.step(), as
// the reference's train_agent.py:33-47 configures them; torch.optim.Adam without weight decay / amsgrad). As tensor-library calls the pair is ~20 launches,
// ~105 us per optimiser step of 1.4 M parameters in 21 tensors (multi-tensor norm, its clean-up, six scalar kernels, the in-place scaling, the step counters,
// the fused Adam); the arithmetic moves 45 MB. The tensors' addresses ride in the kernel arguments (nothing to keep in device memory, capturable in a graph).
//   k_gradnorm   per 4096-element chunk: sum of g^2 -> partial[chunk]; the first chunk of every tensor advances that tensor's step counter
//   k_clip_adam  every block adds the partials in the same fixed order -> total norm, c = min(1, max_norm / (norm + 1e-6)); g *= c (written back, as the
//                reference leaves it), m = b1 m + (1 - b1) g, v = b2 v + (1 - b2) g^2, p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#define CA_MAXT 48
#define CA_CHUNK 4096
struct ClipAdamArgs {
    float *p[CA_MAXT], *g[CA_MAXT], *m[CA_MAXT], *v[CA_MAXT], *step[CA_MAXT];
    int chunk0[CA_MAXT + 1];                    // first chunk of tensor t (prefix sums of ceil(numel / CA_CHUNK))
    int numel[CA_MAXT];
    int n;
};

__device__ __forceinline__ int ca_find(const ClipAdamArgs &a, int chunk) { int t = 0; while (t + 1 < a.n && a.chunk0[t + 1] <= chunk) t++; return t; }

__global__ void __launch_bounds__(256) k_gradnorm(const ClipAdamArgs a, float *__restrict__ partial) {
    __shared__ float red[4];
    const int t = ca_find(a, blockIdx.x), c = blockIdx.x - a.chunk0[t], base = c * CA_CHUNK, n = a.numel[t];
    const float *g = a.g[t];
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < CA_CHUNK / 256; u++) { const int i = base + u * 256 + threadIdx.x; if (i < n) { const float x = g[i]; s += x * x; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        if (c == 0) a.step[t][0] += 1.0f;
    }
}

__global__ void __launch_bounds__(256) k_clip_adam(const ClipAdamArgs a, const float *__restrict__ partial, int nchunks, float lr, float b1, float b2, float eps, float max_norm,
                                                   float *__restrict__ norm_out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nchunks; i += 256) s += partial[i];                  // the same order in every block: every block gets the same bits
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float norm = sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    const float coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
    const int t = ca_find(a, blockIdx.x), base = (blockIdx.x - a.chunk0[t]) * CA_CHUNK, n = a.numel[t];
    const float step = a.step[t][0];                                                   // already advanced by k_gradnorm
    const float bc1 = 1.0f - powf(b1, step), bc2s = sqrtf(1.0f - powf(b2, step)), step_size = lr / bc1;
    float *p = a.p[t], *g = a.g[t], *m = a.m[t], *v = a.v[t];
#pragma unroll
    for (int u = 0; u < CA_CHUNK / 256; u++) {
        const int i = base + u * 256 + threadIdx.x;
        if (i < n) {
            const float gi = g[i] * coef;
            const float mi = b1 * m[i] + (1.0f - b1) * gi, vi = b2 * v[i] + (1.0f - b2) * gi * gi;
            g[i] = gi; m[i] = mi; v[i] = vi;
            p[i] -= step_size * mi / (sqrtf(vi) / bc2s + eps);
        }
    }
}

extern "C" int grip_clip_adam_chunks(int n_tensors, const int64_t *numel) {
    long long c = 0;
    for (int t = 0; t < n_tensors; t++) c += (numel[t] + CA_CHUNK - 1) / CA_CHUNK;
    return (int)c;
}

extern "C" int grip_clip_adam(int n_tensors, const int64_t *numel, float *const *params_dev, float *const *grads_dev, float *const *exp_avg_dev, float *const *exp_avg_sq_dev,
                              float *const *steps_dev, float lr, float beta1, float beta2, float eps, float max_norm, float *partials_dev, float *norm_out_dev, void *stream) {
    if (n_tensors <= 0 || n_tensors > CA_MAXT || !numel || !params_dev || !grads_dev || !exp_avg_dev || !exp_avg_sq_dev || !steps_dev || !partials_dev)
        return grip_fail("grip_clip_adam: 1..48 tensors, their element counts, five arrays of device addresses and the partial-sum scratch are needed");
    ClipAdamArgs a;
    a.n = n_tensors; a.chunk0[0] = 0;
    for (int t = 0; t < n_tensors; t++) {
        if (numel[t] <= 0 || numel[t] > 0x7fffffff || !params_dev[t] || !grads_dev[t] || !exp_avg_dev[t] || !exp_avg_sq_dev[t] || !steps_dev[t])
            return grip_fail("grip_clip_adam: an empty tensor or a missing address");
        a.p[t] = params_dev[t]; a.g[t] = grads_dev[t]; a.m[t] = exp_avg_dev[t]; a.v[t] = exp_avg_sq_dev[t]; a.step[t] = steps_dev[t];
        a.numel[t] = (int)numel[t]; a.chunk0[t + 1] = a.chunk0[t] + (int)((numel[t] + CA_CHUNK - 1) / CA_CHUNK);
    }
    const int nchunks = a.chunk0[n_tensors];
    hipLaunchKernelGGL(k_gradnorm, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a, partials_dev);
    hipLaunchKernelGGL(k_clip_adam, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, a, (const float *)partials_dev, nchunks, lr, beta1, beta2, eps, max_norm, norm_out_dev);
    return launch_check("grip_clip_adam");
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of a tanh layer of the policy | value MLPs (stable_baselines3's MlpExtractor, net_arch [256, 256] each, the reference's train_agent.py:33-47):
//   gz = g * (1 - h^2)  and  gb[c] = sum_rows gz[., c]   (the layer's bias gradient)
// in one pass over g and h instead of a tanh_backward pass plus a column reduction (13 + 16 .. 26 us per layer and 4096 rows as tensor-library calls).
// g is [B][n][C] (the batch-of-two GEMM's output layout); h and gz are addressed as [row][b * C + c] with hs floats per row: hs = B * C stores the result
// row-major [n][B * C] (what the first layer's concatenated GEMMs want, a transposition folded into the pass), B = 1 is the plain case.
// 32 rows per workgroup; a second small launch adds the partial column sums in a fixed order (bit-identical from run to run).
#define TB_ROWS 32
template <bool RELU>
__global__ void __launch_bounds__(256) k_tanh_bwd_colsum(const float *__restrict__ g, int g_row_stride, const float *__restrict__ h, float *__restrict__ gz, int B, int n, int C, int hs,
                                                         long long h_bstride, float *__restrict__ partial) {
    const int BC = B * C, r0 = blockIdx.x * TB_ROWS;
    for (int col = threadIdx.x; col < BC; col += 256) {
        const int b = col / C, c = col - b * C;
        float s = 0.f;
#pragma unroll 4
        for (int r = r0; r < min(r0 + TB_ROWS, n); r++) {
            const size_t hi = (size_t)b * h_bstride + (size_t)r * hs + c;
            const float hv = h[hi], gv = g[((size_t)b * n + r) * g_row_stride + c], v = RELU ? (hv > 0.f ? gv : 0.f) : gv * (1.0f - hv * hv);
            gz[hi] = v; s += v;
        }
        partial[(size_t)blockIdx.x * BC + col] = s;
    }
}

__global__ void __launch_bounds__(1024) k_colsum_reduce(const float *__restrict__ partial, int blocks, int BC, float *__restrict__ gb) {
    __shared__ float red[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;          // 64 columns per workgroup, the partials in 16 slices, added in a fixed order
    float s = 0.f;
    if (col < BC) for (int k = sl; k < blocks; k += 16) s += partial[(size_t)k * BC + col];
    red[sl][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl == 0 && col < BC) {
        float t = red[0][threadIdx.x];
#pragma unroll
        for (int i = 1; i < 16; i++) t += red[i][threadIdx.x];
        gb[col] = t;
    }
}

extern "C" int grip_tanh_backward_colsum(const float *g_dev, const float *h_dev, float *gz_dev, int batch, int n, int cols, int row_stride, int64_t h_batch_stride,
                                         float *scratch_dev, float *grad_bias_dev, void *stream) {
    if (!g_dev || !h_dev || !gz_dev || !scratch_dev || !grad_bias_dev || batch < 1 || n < 1 || cols < 1 || row_stride < cols)
        return grip_fail("grip_tanh_backward_colsum: need g [batch, n, cols], h / gz addressed [b * h_batch_stride + row * row_stride + c], the scratch "
                         "(ceil(n / 32) * batch * cols floats) and the bias-gradient output [batch * cols]");
    const int blocks = (n + TB_ROWS - 1) / TB_ROWS, BC = batch * cols;
    hipLaunchKernelGGL(k_tanh_bwd_colsum<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g_dev, cols, h_dev, gz_dev, batch, n, cols, row_stride, (long long)h_batch_stride, scratch_dev);
    hipLaunchKernelGGL(k_colsum_reduce, dim3((BC + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const float *)scratch_dev, blocks, BC, grad_bias_dev);
    return launch_check("grip_tanh_backward_colsum");
}

// The ReLU counterpart for the extractor's linear layer (models/feature_extractor.py:22): gz = g * (h > 0), grad_bias = column sums; g's rows may be
// g_row_stride floats apart (the first `cols` columns of the gradient of a wider concatenation), h and gz are [n, cols] contiguous.
extern "C" int grip_relu_backward_colsum(const float *g_dev, int g_row_stride, const float *h_dev, float *gz_dev, int n, int cols, float *scratch_dev, float *grad_bias_dev,
                                         void *stream) {
    if (!g_dev || !h_dev || !gz_dev || !scratch_dev || !grad_bias_dev || n < 1 || cols < 1 || g_row_stride < cols)
        return grip_fail("grip_relu_backward_colsum: need g (rows g_row_stride >= cols apart), h / gz [n, cols], the scratch (ceil(n / 32) * cols floats) and the bias-gradient output");
    const int blocks = (n + TB_ROWS - 1) / TB_ROWS;
    hipLaunchKernelGGL(k_tanh_bwd_colsum<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g_dev, g_row_stride, h_dev, gz_dev, 1, n, cols, cols, 0LL, scratch_dev);
    hipLaunchKernelGGL(k_colsum_reduce, dim3((cols + 63) / 64), dim3(1024), 0, (hipStream_t)stream, (const float *)scratch_dev, blocks, cols, grad_bias_dev);
    return launch_check("grip_relu_backward_colsum");
}

// z = tanh(z + bias) in place for a batch of matrices: z [batch, n, cols] contiguous, bias [batch, cols] (cols % 4 == 0). The second layers of the policy | value MLPs
// are a batch-of-two GEMM, which has no bias epilogue in the library: as tensor-library calls the bias costs a broadcast copy into the GEMM's output before it and the
// tanh a pass after it; here both are one pass.
__global__ void __launch_bounds__(256) k_bias_tanh(float *__restrict__ z, const float *__restrict__ bias, int n, int cols, long long total4) {
    const int c4 = cols >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long row = i / c4; const int c = (int)(i - row * c4);
        const int b = (int)(row / n);
        const float4 v = reinterpret_cast<const float4 *>(z)[i];
        const float4 bb = reinterpret_cast<const float4 *>(bias)[(size_t)b * c4 + c];
        reinterpret_cast<float4 *>(z)[i] = make_float4(tanhf(v.x + bb.x), tanhf(v.y + bb.y), tanhf(v.z + bb.z), tanhf(v.w + bb.w));
    }
}

extern "C" int grip_bias_tanh(float *z_dev, const float *bias_dev, int batch, int n, int cols, void *stream) {
    if (!z_dev || !bias_dev || batch < 1 || n < 1 || cols < 4 || (cols & 3))
        return grip_fail("grip_bias_tanh: need z [batch, n, cols] and bias [batch, cols] with cols a multiple of 4");
    const long long total4 = (long long)batch * n * (cols >> 2);
    const int blocks = (int)std::min<long long>((total4 + 255) / 256, 2048);
    hipLaunchKernelGGL(k_bias_tanh, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z_dev, bias_dev, n, cols, total4);
    return launch_check("grip_bias_tanh");
}
