// What ONE SIMD sustains with 1, 2 and 4 resident waves, per instruction kind and encoding -- measured over ALL waves (every wave stamps
// s_memtime around its stream; the figure is cycles per instruction of a wave, mean over the waves, and from it how often the SIMD issues one), not
// on the oldest wave alone as t_issue.hip does (the oldest wave wins the arbitration and shows the lone-wave figure at any occupancy; and that file was
// built without -fno-slp-vectorize, so its "independent v_fma_f32" are v_pk_fma_f32 -- two per instruction).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize tools/hiptests/t_simd_rate.hip -o tools/hiptests/bin/t_simd_rate && tools/hiptests/bin/t_simd_rate
// Under rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU ... it calibrates what "VALU active / wave cycles" reads at a known issue rate
// (SQ_ACTIVE_INST_VALU is exactly one quad-cycle per VALU instruction: a count, not an occupancy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define REP 64
typedef float f4v __attribute__((ext_vector_type(4)));
// the stream: 64 instructions per loop trip over 16 registers; IND: every instruction on its own register (16 independent chains), else all on a[0]
#define X16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)
#define X64(I) X16(I) X16(I) X16(I) X16(I)
#define R(i) a[IND ? (i) : 0]
#define Q(i) a[IND ? (((i) + 5) & 15) : 1]
enum { FMA_VOP3_SGPR, FMA_VOP3_VGPR, FMAC_E32, ADD_E32, MUL_E32_SGPR, MOV_E32, CND_VCC_E32, CND_SGPR_E64, BFI, MAX_E32, AND_E32, LSHL_ADD, CMP_VCC, CMP_SGPR, CMP_CND_PAIR, RCP, SQRT, DPP_MOV, DPP_ADD, READLANE, WRITELANE,
       READFIRST, SWAP32, DS_SWIZZLE, DS_BPERMUTE, DS_READ_B32, DS_READ_B128, DS_WRITE_B32, SALU_MOV, FMA_MOV_MIX, CVT, MAD_U32, CMP1_CND15, SAND1_CND15, CMPS1_CNDS15, S_MOV32, S_AND64, S_SAVEEXEC, S_WAITCNT, S_NOP, S_CBR_NT, VADD_SADD, VADD3_SADD1, VADD_SNOP, VADD_WAIT, CMP1_CND3, CMP1_CND1_ADD2, CMP1_CND3_E64VCC, CMPS1_CNDS3, CND_E64_VCC, CMP1_CND2_INTERLEAVED, DS_WRITE_B32_SAME, DS_WRITE_B128_SAME, DS_WRITE_B128, DS_READ_B128_BCAST, DS_WRITE_B128_ROWSAME, NKIND };
template <int KIND, int IND> __global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 0.001f + i;
    const float s = out[0];
    const unsigned long long msk = ((unsigned long long *)out)[1] | 0x5555555555555555ull;
    const int ldsa = (threadIdx.x & 63) * 16, rowa = ((threadIdx.x & 63) >> 4) * 7296 * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_mov_b64 vcc, %0" :: "s"(msk) : "vcc");
    for (int it = 0; it < iters; it++) {
        if (KIND == FMA_VOP3_SGPR) {
#define I(i) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(R(i)) : "s"(s));
            X64(I)
#undef I
        } else if (KIND == FMA_VOP3_VGPR) {
#define I(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == FMAC_E32) {
#define I(i) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == ADD_E32) {
#define I(i) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == MUL_E32_SGPR) {
#define I(i) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(R(i)) : "s"(s));
            X64(I)
#undef I
        } else if (KIND == MOV_E32) {
#define I(i) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == CND_VCC_E32) {          // mask in vcc, written once by the scalar unit before the loop
#define I(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == CND_SGPR_E64) {
#define I(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(R(i)) : "v"(Q(i)), "s"(msk));
            X64(I)
#undef I
        } else if (KIND == BFI) {
#define I(i) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == MAX_E32) {
#define I(i) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == AND_E32) {
#define I(i) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == LSHL_ADD) {
#define I(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == CMP_VCC) {
#define I(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" :: "v"(R(i)), "v"(Q(i)) : "vcc");
            X64(I)
#undef I
        } else if (KIND == CMP_SGPR) {
#define I(i) { unsigned long long m_; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m_) : "v"(R(i)), "v"(Q(i))); asm volatile("" :: "s"(m_)); }
            X64(I)
#undef I
        } else if (KIND == CMP_CND_PAIR) {           // the select idiom: compare into vcc, select on it (32 pairs = 64 instructions)
#define I(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(R(i)) : "v"(Q(i)) : "vcc");
            X16(I) X16(I)
#undef I
        } else if (KIND == RCP) {
#define I(i) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(R(i)));
            X64(I)
#undef I
        } else if (KIND == SQRT) {
#define I(i) asm volatile("v_sqrt_f32_e32 %0, %0" : "+v"(R(i)));
            X64(I)
#undef I
        } else if (KIND == DPP_MOV) {
#define I(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0xf" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == DPP_ADD) {
#define I(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(R(i)));
            X64(I)
#undef I
        } else if (KIND == READLANE) {
#define I(i) { int t_; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(t_) : "v"(R(i))); asm volatile("" :: "s"(t_)); }
            X64(I)
#undef I
        } else if (KIND == WRITELANE) {
#define I(i) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(R(i)) : "s"(s));
            X64(I)
#undef I
        } else if (KIND == READFIRST) {
#define I(i) { int t_; asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(t_) : "v"(R(i))); asm volatile("" :: "s"(t_)); }
            X64(I)
#undef I
        } else if (KIND == SWAP32) {
#define I(i) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(R(i)), "+v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == DS_SWIZZLE) {
#define I(i) asm volatile("ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM,\"0010p\")\n s_waitcnt lgkmcnt(0)" : "+v"(R(i)));
            X16(I)
#undef I
        } else if (KIND == DS_BPERMUTE) {
#define I(i) { int ad_ = ldsa; asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(R(i)) : "v"(ad_)); }
            X16(I)
#undef I
        } else if (KIND == DS_READ_B32) {
#define I(i) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(R(i)) : "v"(ldsa), "n"(4 * (i)));
            X64(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == DS_READ_B128) {
#define I(i) { f4v t_; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t_) : "v"(ldsa), "n"(16 * (i))); asm volatile("" :: "v"(t_)); }
            X64(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == DS_WRITE_B32) {
#define I(i) asm volatile("ds_write_b32 %1, %0 offset:%2" :: "v"(R(i)), "v"(ldsa), "n"(4 * (i)));
            X64(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == SALU_MOV) {
#define I(i) { int t_; asm volatile("s_add_u32 %0, %1, 7" : "=s"(t_) : "s"(it) : "scc"); asm volatile("" :: "s"(t_)); }
            X64(I)
#undef I
        } else if (KIND == FMA_MOV_MIX) {            // half VOP3 multiply-adds, half plain moves
#define I(i) asm volatile("v_fma_f32 %0, %0, %1, %1\n v_mov_b32_e32 %2, %1" : "+v"(R(i)), "+v"(Q(i)), "=v"(a[15]));
            X16(I) X16(I)
#undef I
        } else if (KIND == CVT) {
#define I(i) asm volatile("v_cvt_f32_i32_e32 %0, %0" : "+v"(R(i)));
            X64(I)
#undef I
        } else if (KIND == MAD_U32) {
#define I(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == CMP1_CND15) {             // one compare into vcc, fifteen selects on it
#define I(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(R(i)) : "v"(Q(i)));
#define G asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" :: "v"(a[0]), "v"(a[1]) : "vcc"); I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)
            G G G G
#undef G
#undef I
        } else if (KIND == SAND1_CND15) {            // vcc written by the scalar unit, fifteen selects on it
#define I(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(R(i)) : "v"(Q(i)));
#define G asm volatile("s_and_b64 vcc, %0, exec" :: "s"(msk) : "vcc", "scc"); I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)
            G G G G
#undef G
#undef I
        } else if (KIND == CMPS1_CNDS15) {           // one compare into an SGPR pair, fifteen selects on it
#define I(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(R(i)) : "v"(Q(i)), "s"(m_));
#define G { unsigned long long m_; asm volatile("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(m_) : "v"(a[0]), "v"(a[1])); I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15) }
            G G G G
#undef G
#undef I
        } else if (KIND == S_MOV32) {
#define I(i) { int t_; asm volatile("s_mov_b32 %0, %1" : "=s"(t_) : "s"(it)); asm volatile("" :: "s"(t_)); }
            X64(I)
#undef I
        } else if (KIND == S_AND64) {
#define I(i) { unsigned long long t_; asm volatile("s_and_b64 %0, %1, exec" : "=s"(t_) : "s"(msk) : "scc"); asm volatile("" :: "s"(t_)); }
            X64(I)
#undef I
        } else if (KIND == S_SAVEEXEC) {
#define I(i) { unsigned long long t_; asm volatile("s_and_saveexec_b64 %0, exec" : "=s"(t_) :: "scc"); asm volatile("" :: "s"(t_)); }
            X64(I)
#undef I
        } else if (KIND == S_WAITCNT) {
#define I(i) asm volatile("s_waitcnt lgkmcnt(0)");
            X64(I)
#undef I
        } else if (KIND == S_NOP) {
#define I(i) asm volatile("s_nop 0");
            X64(I)
#undef I
        } else if (KIND == S_CBR_NT) {               // branch never taken (exec is never zero)
#define I(i) asm volatile("s_cbranch_execz 1f\n1:");
            X64(I)
#undef I
        } else if (KIND == VADD_SADD) {              // vector and scalar instructions alternating (32 + 32)
#define I(i) { int t_; asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(R(i)) : "v"(Q(i))); asm volatile("s_add_u32 %0, %1, 7" : "=s"(t_) : "s"(it) : "scc"); asm volatile("" :: "s"(t_)); }
            X16(I) X16(I)
#undef I
        } else if (KIND == VADD3_SADD1) {            // three vector, one scalar (48 + 16)
#define I(i) { int t_; asm volatile("v_add_f32_e32 %0, %0, %1\n v_add_f32_e32 %2, %2, %1\n v_add_f32_e32 %3, %3, %1" : "+v"(R(i)), "+v"(R((i + 1) & 15)), "+v"(R((i + 2) & 15)) : "v"(Q(i))); asm volatile("s_add_u32 %0, %1, 7" : "=s"(t_) : "s"(it) : "scc"); asm volatile("" :: "s"(t_)); }
            X16(I)
#undef I
        } else if (KIND == VADD_SNOP) {
#define I(i) asm volatile("v_add_f32_e32 %0, %0, %1\n s_nop 0" : "+v"(R(i)) : "v"(Q(i)));
            X16(I) X16(I)
#undef I
        } else if (KIND == VADD_WAIT) {
#define I(i) asm volatile("v_add_f32_e32 %0, %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(R(i)) : "v"(Q(i)));
            X16(I) X16(I)
#undef I
        } else if (KIND == CMP1_CND3) {              // one compare into vcc, three selects on it (the compiler's usual shape)
#define I(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %2, %2, %1, vcc\n v_cndmask_b32_e32 %3, %3, %1, vcc" : "+v"(R(i)), "+v"(Q(i)), "+v"(R((i + 1) & 15)), "+v"(R((i + 2) & 15)) :: "vcc");
            X16(I)
#undef I
        } else if (KIND == CMP1_CND1_ADD2) {         // one compare, ONE select, two adds
#define I(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_add_f32_e32 %2, %2, %1\n v_add_f32_e32 %3, %3, %1" : "+v"(R(i)), "+v"(Q(i)), "+v"(R((i + 1) & 15)), "+v"(R((i + 2) & 15)) :: "vcc");
            X16(I)
#undef I
        } else if (KIND == CMP1_CND3_E64VCC) {       // the same with the selects in their 64-bit encoding, vcc named as the mask operand
#define I(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %2, %2, %1, vcc\n v_cndmask_b32_e64 %3, %3, %1, vcc" : "+v"(R(i)), "+v"(Q(i)), "+v"(R((i + 1) & 15)), "+v"(R((i + 2) & 15)) :: "vcc");
            X16(I)
#undef I
        } else if (KIND == CMPS1_CNDS3) {            // compare into an SGPR pair, three selects on it
#define I(i) { unsigned long long m_; asm volatile("v_cmp_gt_f32_e64 %4, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, %4\n v_cndmask_b32_e64 %2, %2, %1, %4\n v_cndmask_b32_e64 %3, %3, %1, %4" : "+v"(R(i)), "+v"(Q(i)), "+v"(R((i + 1) & 15)), "+v"(R((i + 2) & 15)), "=&s"(m_)); }
            X16(I)
#undef I
        } else if (KIND == CND_E64_VCC) {            // 64-bit encoding, vcc (written once by the scalar unit) named as the mask
#define I(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(R(i)) : "v"(Q(i)));
            X64(I)
#undef I
        } else if (KIND == CMP1_CND2_INTERLEAVED) {  // compare, add, select, add, select, add ... : selects on vcc with other work between them
#define I(i) asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1\n v_add_f32_e32 %2, %2, %1\n v_cndmask_b32_e32 %0, %0, %1, vcc\n v_add_f32_e32 %3, %3, %1\n v_cndmask_b32_e32 %2, %2, %1, vcc\n v_add_f32_e32 %3, %3, %1\n v_add_f32_e32 %3, %3, %1\n v_add_f32_e32 %3, %3, %1" : "+v"(R(i)), "+v"(Q(i)), "+v"(R((i + 1) & 15)), "+v"(R((i + 2) & 15)) :: "vcc");
            I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#undef I
        } else if (KIND == DS_WRITE_B32_SAME) {      // every lane writes the SAME address (the value is the same too: what an unguarded "lane 0 stores" would be)
#define I(i) asm volatile("ds_write_b32 %1, %0 offset:%2" :: "v"(a[3]), "v"(0), "n"(4 * (i)));
            X64(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == DS_WRITE_B128_SAME) {
#define I(i) { f4v t_ = {a[0], a[1], a[2], a[3]}; asm volatile("ds_write_b128 %1, %0 offset:%2" :: "v"(t_), "v"(0), "n"(16 * (i))); }
            X64(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == DS_WRITE_B128) {          // every lane its own 16 bytes
#define I(i) { f4v t_ = {a[0], a[1], a[2], a[3]}; asm volatile("ds_write_b128 %1, %0 offset:%2" :: "v"(t_), "v"(ldsa), "n"(1024 * ((i) & 15))); }
            X64(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == DS_READ_B128_BCAST) {     // all 64 lanes read the same 16 bytes
#define I(i) { f4v t_; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t_) : "v"(0), "n"(16 * (i))); asm volatile("" :: "v"(t_)); }
            X64(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (KIND == DS_WRITE_B128_ROWSAME) {  // the 16 lanes of a row write the same 16 bytes, the four rows different ones (the kernel's "env's lanes store the env's vector")
#define I(i) { f4v t_ = {a[0], a[1], a[2], a[3]}; asm volatile("ds_write_b128 %1, %0 offset:%2" :: "v"(t_), "v"(rowa), "n"(16 * (i))); }
            X64(I)
#undef I
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
    for (int i = 0; i < 16; i++) r += a[i];
    if (r == 12345.678f) out[1] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
}
static int per_trip(int kind) { return (kind == DS_SWIZZLE || kind == DS_BPERMUTE) ? 16 : 64; }
template <int KIND, int IND> void run(const char *name, float *d, unsigned long long *c) {
    const int iters = 2000, nblk = 256;
    for (int threads : {256, 512, 1024}) {
        hipMemset(c, 0, nblk * 16 * 8);
        // 100 KB of dynamic LDS: one workgroup per CU, so threads / 256 IS the number of waves per SIMD
        hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND, IND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<KIND, IND>), dim3(nblk), dim3(threads), 100 * 1024, 0, d, c, 10);      // warm-up
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, IND>), dim3(nblk), dim3(threads), 100 * 1024, 0, d, c, iters);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nblk * 16); hipMemcpy(h.data(), c, nblk * 16 * 8, hipMemcpyDeviceToHost);
        const int nw = threads / 64; double sum = 0, mx = 0; int cnt = 0;
        for (int b = 0; b < nblk; b++) for (int w = 0; w < nw; w++) { double v = (double)h[b * 16 + w]; sum += v; mx = std::max(mx, v); cnt++; }
        const double ninstr = (double)iters * per_trip(KIND), mean = sum / cnt;
        printf("%-34s %s  %d wave(s)/SIMD: %6.2f cyc per instr of a wave (slowest wave %6.2f) -> the SIMD issues one per %5.2f cyc; kernel %.3f ms\n",
               name, IND ? "independent" : "dependent  ", threads / 256, mean / ninstr, mx / ninstr, mean / ninstr / (threads / 256), ms);
    }
}
#define BOTH(K, name) run<K, 1>(name, d, c); run<K, 0>(name, d, c);
int main() {
    float *d; unsigned long long *c; hipMalloc(&d, 64); hipMalloc(&c, 256 * 16 * 8); hipMemset(d, 0, 64);
    BOTH(FMA_VOP3_SGPR, "v_fma_f32 v,v,s,1.0 (VOP3)") BOTH(FMA_VOP3_VGPR, "v_fma_f32 v,v,v,v (VOP3)") BOTH(FMAC_E32, "v_fmac_f32_e32 (VOP2)") BOTH(ADD_E32, "v_add_f32_e32")
    BOTH(MUL_E32_SGPR, "v_mul_f32_e32 v,s,v") BOTH(MOV_E32, "v_mov_b32_e32") BOTH(CND_VCC_E32, "v_cndmask_b32_e32 vcc") BOTH(CND_SGPR_E64, "v_cndmask_b32_e64 s[n:n+1]")
    BOTH(BFI, "v_bfi_b32") BOTH(MAX_E32, "v_max_f32_e32") BOTH(AND_E32, "v_and_b32_e32") BOTH(LSHL_ADD, "v_lshl_add_u32") BOTH(CMP_VCC, "v_cmp_gt_f32_e32 vcc") BOTH(CMP_SGPR, "v_cmp_gt_f32_e64 sgpr")
    BOTH(CMP_CND_PAIR, "v_cmp + v_cndmask pair (per instr)") BOTH(RCP, "v_rcp_f32") BOTH(SQRT, "v_sqrt_f32") BOTH(DPP_MOV, "v_mov_b32_dpp row_ror") BOTH(DPP_ADD, "v_add_f32_dpp row_ror")
    BOTH(READLANE, "v_readlane_b32") BOTH(WRITELANE, "v_writelane_b32") BOTH(READFIRST, "v_readfirstlane_b32") BOTH(SWAP32, "v_permlane32_swap")
    run<DS_SWIZZLE, 1>("ds_swizzle + wait (round trip)", d, c); run<DS_BPERMUTE, 1>("ds_bpermute + wait (round trip)", d, c);
    run<DS_READ_B32, 1>("ds_read_b32 x64, one wait", d, c); run<DS_READ_B128, 1>("ds_read_b128 x64, one wait", d, c); run<DS_WRITE_B32, 1>("ds_write_b32 x64, one wait", d, c);
    run<SALU_MOV, 1>("s_add_u32", d, c); BOTH(FMA_MOV_MIX, "v_fma VOP3 + v_mov alternating") BOTH(CVT, "v_cvt_f32_i32") BOTH(MAD_U32, "v_mad_u32_u24")
    run<CMP1_CND15, 1>("1 v_cmp vcc + 15 v_cndmask vcc", d, c); run<SAND1_CND15, 1>("1 s_and vcc + 15 v_cndmask vcc", d, c); run<CMPS1_CNDS15, 1>("1 v_cmp sgpr + 15 v_cndmask sgpr", d, c);
    run<S_MOV32, 1>("s_mov_b32", d, c); run<S_AND64, 1>("s_and_b64", d, c); run<S_SAVEEXEC, 1>("s_and_saveexec_b64", d, c); run<S_WAITCNT, 1>("s_waitcnt (nothing pending)", d, c); run<S_NOP, 1>("s_nop 0", d, c);
    run<S_CBR_NT, 1>("s_cbranch_execz not taken", d, c); run<VADD_SADD, 1>("v_add + s_add alternating", d, c); run<VADD3_SADD1, 1>("3 v_add + 1 s_add", d, c); run<VADD_SNOP, 1>("v_add + s_nop alternating", d, c);
    run<VADD_WAIT, 1>("v_add + s_waitcnt alternating", d, c);
    run<CMP1_CND3, 1>("1 v_cmp vcc + 3 v_cndmask vcc", d, c); run<CMP1_CND1_ADD2, 1>("1 v_cmp vcc + 1 v_cndmask + 2 v_add", d, c); run<CMP1_CND3_E64VCC, 1>("1 v_cmp vcc + 3 v_cndmask_e64 vcc", d, c);
    run<CMPS1_CNDS3, 1>("1 v_cmp sgpr + 3 v_cndmask sgpr", d, c); run<CND_E64_VCC, 1>("v_cndmask_b32_e64 .., vcc", d, c); run<CMP1_CND2_INTERLEAVED, 1>("cmp add cnd add cnd add add add", d, c);
    run<DS_WRITE_B32_SAME, 1>("ds_write_b32, all lanes one address", d, c); run<DS_WRITE_B128_SAME, 1>("ds_write_b128, all lanes one address", d, c); run<DS_WRITE_B128, 1>("ds_write_b128, own 16 B per lane", d, c);
    run<DS_READ_B128_BCAST, 1>("ds_read_b128, all lanes one address", d, c); run<DS_WRITE_B128_ROWSAME, 1>("ds_write_b128, one address per row", d, c);
    return 0;
}
