// Issue cost of instruction kinds for ONE wave alone on its SIMD (the physics kernel's regime: 4096 envs = one 4-env wave per SIMD) and
// for two / four waves per SIMD: cycles per instruction of long unrolled streams of independent v_fma_f32, v_pk_fma_f32, v_mov (VGPR),
// v_accvgpr moves, ds_read_b128 (broadcast address), ds_swizzle, DPP adds. s_memtime around the stream, lane 0 of wave 0 reports.
//   hipcc --offload-arch=gfx950 -O3 tools/hiptests/t_issue.hip -o /tmp/t_issue && /tmp/t_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
#define REP 64
template <int KIND> __global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float a[16]; float2v p[8];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 8; i++) p[i] = float2v{a[2 * i], a[2 * i + 1]};
    const float s = out[0]; const float2v s2 = float2v{s, s};
    float4 acc4 = make_float4(0, 0, 0, 0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 16; r++) {
            if (KIND == 0) {
#pragma unroll
                for (int i = 0; i < 16; i++) a[i] = __builtin_fmaf(a[i], s, 1.0f);
            } else if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(s2));
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(s2));
            } else if (KIND == 2) {
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 15]));
            } else if (KIND == 3) {
#pragma unroll
                for (int i = 0; i < 16; i++) { float t; asm volatile("v_accvgpr_write_b32 a0, %1\n v_accvgpr_read_b32 %0, a0" : "=v"(t) : "v"(a[i]) : "a0"); a[i] = t; }
            } else if (KIND == 4) {
#pragma unroll
                for (int i = 0; i < 16; i++) { float4 t = *reinterpret_cast<const float4 *>(&lds[((it + i) & 63) * 4]); acc4.x += t.x; }
            } else if (KIND == 5) {
#pragma unroll
                for (int i = 0; i < 16; i++) a[i] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a[i]), (5 << 5) | 0x10));
            } else if (KIND == 6) {
#pragma unroll
                for (int i = 0; i < 16; i++) a[i] = a[i] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x128, 0xF, 0xF, false));
            } else if (KIND == 7) {     // dependent fma chain
#pragma unroll
                for (int i = 0; i < 16; i++) a[0] = __builtin_fmaf(a[0], s, 1.0f);
            } else if (KIND == 8) {     // dependent ds_read chain (pointer chase, broadcast)
#pragma unroll
                for (int i = 0; i < 16; i++) { int j = (int)a[0] & 1023; a[0] = lds[j] * 0.0f + (float)((j * 7 + 1) & 1023); }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = acc4.x;
    for (int i = 0; i < 16; i++) r += a[i];
    for (int i = 0; i < 8; i++) r += p[i].x + p[i].y;
    if (r == 12345.678f) out[1] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND> void run(const char *name, float *d, unsigned long long *c) {
    const int iters = 2000;
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 0, 0, d, c, iters);
        hipDeviceSynchronize();
        unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        int per = (KIND == 1 ? 2 : 1);
        printf("%-28s %d wave(s)/SIMD: %6.2f cycles (s_memtime) per instruction of one wave\n", name, threads / 256, (double)h / ((double)iters * REP * per / (KIND == 1 ? 2 : 1)));
    }
}
int main() {
    float *d; unsigned long long *c; hipMalloc(&d, 64); hipMalloc(&c, 8); hipMemset(d, 0, 64);
    run<0>("v_fma_f32 independent", d, c); run<1>("v_pk_fma_f32 independent", d, c); run<2>("v_mov_b32", d, c); run<3>("accvgpr write+read pair", d, c);
    run<4>("ds_read_b128 + add", d, c); run<5>("ds_swizzle (dependent x16)", d, c); run<6>("v_add + dpp row_ror:8", d, c); run<7>("v_fma_f32 dependent chain", d, c);
    run<8>("ds_read_b32 dependent chain", d, c);
    return 0;
}
