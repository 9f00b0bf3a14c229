// Is the fp32 MFMA rate the same when the operands come out of LDS? Two waves per SIMD, each a stream of v_mfma_f32_32x32x2_f32 on two accumulators with
// (a) register operands, (b) two ds_read_b32 per instruction pair feeding them (addresses spread over the banks). s_memtime (100 MHz) around the stream:
// 64 cycles per instruction / measured time per instruction = the clock the matrix pipe effectively ran at.   hipcc --offload-arch=gfx950 -O3 t_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND> __global__ void __launch_bounds__(256) k(float *out, unsigned long long *res, int iters) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    float a = threadIdx.x * 0.001f, b = out[0], v[8];
    for (int i = 0; i < 8; i++) v[i] = a + i;
    f32x16 c0, c1; for (int r = 0; r < 16; r++) { c0[r] = 0.f; c1[r] = 0.f; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) { c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0); }
            if (KIND == 1) { const float x = lds[(threadIdx.x + 64 * u + it) & 8191], y = lds[(threadIdx.x * 2 + 32 * u + it) & 8191];
                             c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, c1, 0, 0, 0); }
            if (KIND == 2) {
#pragma unroll
                for (int i = 0; i < 8; i++) v[i] = __builtin_fmaf(v[i], b, 1.0f);
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f; for (int r = 0; r < 16; r++) s += c0[r] + c1[r]; for (int i = 0; i < 8; i++) s += v[i];
    if (s == 1.2345f) out[1] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { res[0] = t1 - t0; res[1] = (unsigned long long)iters * 16 * 2; }
}
template <int KIND> void run(const char *name, float *d, unsigned long long *r, int iters) {
    k<KIND><<<256 * 2, 256>>>(d, r, iters);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, r, 16, hipMemcpyDeviceToHost);
    printf("%-24s %8.1f us   %8llu instructions per SIMD -> %.1f ns each = 64 cycles at %.0f MHz\n", name, h[0] / 100.0, h[1], h[0] * 10.0 / h[1], 64.0 * h[1] / (h[0] / 100.0));
}
int main() {
    float *d; unsigned long long *r; hipMalloc(&d, 64); hipMemset(d, 0, 64); hipMalloc(&r, 16);
    for (int rep = 0; rep < 2; rep++) {
        run<0>("MFMA only", d, r, 800); run<1>("MFMA + LDS operands", d, r, 800);
    }
    return 0;
}
