// grip_physics.h -- fp32 device physics: SIXTEEN LANES COOPERATE ON ONE ENVIRONMENT, THEIR CLONES 32 LANES UP TAKE SECOND HALVES (gfx950).
//
// What one `physics.step()` of the reference computes (robot_env.py:100,119,142,157 -> dm_control
// Physics.step -> mj_step2 + mj_step1), laid out for CDNA4:
//
//   * the mapping: a wavefront holds TWO environments -- env A in lanes 0..15, env B in lanes 16..31 -- and lanes
//     32..63 are CLONES of lanes 0..31 (the same env, the same values, the same LDS addresses, no global writes). A 512-thread workgroup
//     holds 16 environments in 8 waves; 4096 envs are 2048 waves = TWO per SIMD of the chip, each covering the other's LDS round trips
//     (needs the kernel inside 256 registers: the env's state lives in LDS, not in registers);
//   * where ONE instruction stream can serve two data sets the clones take the second one and v_permlane32_swap hands the results across
//     (halves_f / halves_i): the support searches of a hull pair's two hulls, the two halves of the cooperative vertex scan, two of a
//     contact's four constraint rows, the odd-numbered contacts of a Hessian row. Everywhere else they repeat the lower half's arithmetic
//     (a wave instruction costs the same for 32 or 64 active lanes);
//   * the small dense per-env algebra (kinematics of four rigid groups, 7x7 + 6x6 mass matrix, bias forces,
//     Cholesky solves, integration) is computed redundantly by the env's lanes, bit-identically, so no
//     broadcast is ever needed for it (one redundancy is put to use: lanes 8..15 factorise M + h D while lanes 0..7 factorise M);
//   * everything with per-contact or per-geom-pair parallelism is spread over the lanes: the 17 narrow-phase
//     items (6 floor-hull, 11 hull-hull) run one per lane, each lane driving its own Minkowski-portal
//     state machine around the single support evaluation; a contact lives in the registers of ONE lane
//     (<= 14 contacts <= 16 lanes), which evaluates its rows, cone, force, its six rank-1 Hessian terms and
//     its share of every line-search derivative;
//   * lanes combine with DPP row rotations (row_ror 8/4/2/1 = a 16-lane all-reduce in four VALU ops,
//     commutative so every lane gets the bit-identical sum), never through memory;
//   * LDS holds what is shared: the convex hulls (vertices, edge graph, cube-map table of start vertices
//     for the hill-climbing support search) once per workgroup, and per env a 7.3 KB region: the state vectors, six geom frames, a
//     14-slot contact staging area, the mass matrix' block rows, the contacts' Jacobian rows / Hessian vectors (EF_* / ES_* below).
#pragma once
#include "grip_device.h"

#define KL 16                           // lanes per environment
// Environments a wave works on: two -- rows 0 and 1 of the wave are envs, rows 2 and 3 their clones (a wave instruction costs the same for 32 or 64 lanes). 4096
// envs are 2048 waves = TWO per SIMD, each covering the other's LDS round trips and issue gaps; needs the kernel inside 256 registers. (Rounds 2-4 kept round 2's
// mapping -- four envs per wave, 256-thread workgroups, one wave per SIMD, no clones -- behind -DGRIP_EPW=4 for comparison; it lost every A/B since round 3
// (86.4 against 93.9 M physics.step()/s then, before the clones took second halves of the work) and was retired in round 5: git history has it.)
#define EPW 2                           // environments per wave
#define EPB 16                          // environments per workgroup
#define WG_THREADS (EPB / EPW * WAVE)
#define WG_WAVES_PER_SIMD (WG_THREADS / 256)
// slot of the calling lane's environment within its workgroup; `active`: the lane belongs to an env row of its wave
DEVI int wg_env_slot(bool &active) {
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    active = lane < EPW * KL;
    return w * EPW + ((lane & (EPW * KL - 1)) / KL);
}

// per-env LDS region (floats)
#define EF_FRAMES 0                     // 6 geom frames x 12 (pos3, R9)
#define EF_STAGE 72                     // G_MAXC staged contacts x ST_STRIDE
#define ST_STRIDE 11                    // pos3, n3, dist, meta, fs, ft, tran
// The mass matrix is block diagonal for good (gripper 7 x 7, object 6 x 6): row r is kept as the eight words of ITS block's row
// ([g0..g6 | 0] for r < 7, [o0..o5 | 0 0] for the others) -- two 16-byte reads per dof lane, no zero blocks in LDS.
#define EF_M ((EF_STAGE + G_MAXC * ST_STRIDE + 3) & ~3)
#define M_STRIDE 8
// packed Cholesky factor (28 words, 1 / L_jj on the diagonal) of the gripper block of M + h D -- mj_Euler's implicit joint damping --
// factorised by the env's lanes 8..15 in the SAME instruction stream in which lanes 0..7 factorise M for qacc_smooth (forward_dense),
// parked here for integrate()
#define EF_LD (EF_M + 13 * M_STRIDE)
#define EF_U (EF_LD + 28)                // per-contact Hessian vectors (16-byte aligned): G_MAXC x 6 slots x U_STRIDE (13 entries + weight)
static_assert(EF_U % 4 == 0, "Hessian-vector slots are read with 16-byte loads");
#define U_STRIDE 16                    // a slot is four ds_read_b128: [ g0 .. g6 | w ][ o0 .. o5 | 0 | w ] -- the seven gripper columns, the six object
                                       // columns, the slot's weight at the end of both halves (a dof lane reads only its block's half when no
                                       // contact couples gripper and object, i.e. when H is block diagonal)
#define UPOS(j) ((j) + ((j) >= 7 ? 1 : 0))   // position of dof column j in a slot
// The env's STATE lives here too, not in registers: every value below is the same in the env's 16 lanes, so one copy in LDS replaces 16 x
// ~150 registers that were live across collide() and the solver (the kernel had 467 of them and ran one wave per SIMD). Each phase loads what
// it needs with 16-byte broadcast reads and lane 0 stores what it changes.
#define ES_QPOS (EF_U + G_MAXC * 6 * U_STRIDE)  // qpos[14]            (16 slots)
#define ES_QVEL (ES_QPOS + 16)          // qvel[13]            (16)
#define ES_CTRL (ES_QVEL + 16)          // ctrl[7]             (8)
#define ES_WARM (ES_CTRL + 8)           // qacc_warmstart[13]  (16)
#define ES_QFS (ES_WARM + 16)           // qfrc_smooth[13] of this step (16), dense block -> integrator
#define ES_QS (ES_QFS + 16)             // qacc_smooth[13] of this step (16), dense block -> solver
#define ES_MAC (ES_QS + 16)             // macro step: target[5], init_q[5], open_close, tq, init_obj[3] (16; same order as the suspended record MCF)
// ... and its integer words, the suspended record's order first (phase is kept in a register: the step loop turns on it):
//   [0] -, [1] cnt, [2] nsub, [3] grasped, [4] flags (bit 0 reached_target, bit 1 reached_initial), [5] fault bits,
//   [8] Newton iterations of this slice, [9] steps of this slice, [10] episode_step, [11] status, [12] gripper_open
#define ES_MACI (ES_MAC + 16)
#define ENV_FLOATS (ES_MACI + 16)
enum { MI_CNT = 1, MI_NSUB = 2, MI_GRASPED = 3, MI_FLAGS = 4, MI_FAULT = 5, MI_SUMIT = 8, MI_NSLICE = 9, MI_EPSTEP = 10, MI_STATUS = 11, MI_GOPEN = 12 };
#define MI(cx, i) (reinterpret_cast<int *>((cx).envl + ES_MACI)[i])
// what the constraint rows need of the kinematics (pe, a4, pk[2], ak[2], po, Ro: 30 floats): kinematics() -> make_constraints(). It sits in the
// last two Hessian-vector slots of the LAST contact, which nobody writes before hessian_vectors() of the following solve.
#define ES_KIN (EF_U + (G_MAXC - 1) * 6 * U_STRIDE + 4 * U_STRIDE)
// The narrow phase's portal memory (3 vertex-pair ids + 3 query directions = 12 words per hull-pair lane, lanes 0..10): in the Hessian-vector
// slots of contacts 12 and 13 (rows 0..3 of slot 13: its rows 4, 5 hold ES_KIN), which the solver touches only when an env has 13 or 14
// contacts -- collide() then forgets the portals (has = 0) and the next step starts them cold, exactly as a fresh macro step does.
#define ES_PORTAL (EF_U + (G_MAXC - 2) * 6 * U_STRIDE)
#define PORTAL_WORDS 12
#define PORTAL_SAFE_NCON (G_MAXC - 2)
static_assert(ES_PORTAL + 11 * PORTAL_WORDS <= ES_KIN && ES_PORTAL % 4 == 0, "portal memory must fit below ES_KIN");
// the integrator's exchange vector (qfrc_smooth + J^T f, one component per dof lane): the staging area is free once the solve is over
#define ES_ACC EF_STAGE
// The Newton step's linear system, gathered so that EVERY lane holds all of it: row r of H at EF_H + 16 r (13 entries), the negative
// gradient at EF_HG. It reuses the geom frames and the staging area (226 floats, dead between collide() and the next kinematics()) and
// overwrites EF_FORCE / EF_P, which the next pricing rewrites before anybody reads them.
#define EF_H EF_FRAMES
#define EF_HG (EF_FRAMES + 208)
static_assert(EF_HG + 13 <= EF_M, "the gathered Hessian must fit the frames + staging area");
static_assert(ES_QPOS % 4 == 0 && ENV_FLOATS % 4 == 0 && ES_KIN % 4 == 0, "state vectors are read with 16-byte loads");
// per-workgroup geom table (floats per geom): centre3, rbound, fs, ft, invweight, group, hull_vadr
#define GT_STRIDE 10
#define GT_FLOATS (GN_GEOM * GT_STRIDE)
// per-workgroup table of the two finger chains' model constants, one 36-float record per chain s (left, right): kn_pos[3], kn_R[9], fin_pos[3],
// fin_R[9], grp_com[1 + s][3], grp_inertia[1 + s][6], grp_mass[1 + s], pad[2]. Lanes 0..31 of a two-env wave work on the left chain, their
// clones 32..63 on the right one, in ONE instruction stream: the constants must come by lane (nine 16-byte reads), not as scalar operands.
#define LR_STRIDE 36
#define LR_FLOATS (4 * LR_STRIDE)       // records 2 (ee + base group) and 3 (object) carry only grp_com, grp_inertia, grp_mass (the same word positions): lanes 0..31 / clones
#define LR_OFF(hull_words) (((hull_words) + GT_FLOATS + 3) & ~3)             // 16-byte aligned
#define LDS_ENV_BASE(hull_words) (LR_OFF(hull_words) + LR_FLOATS)             // float offset of the first env region (16-byte aligned)

#define NEWTON_MAXIT 30
// A solve that is STALLED after this many iterations from qacc_warmstart -- its last exact line search ended at a step length below NEWTON_STALL_ALPHA -- starts again
// from qacc_smooth. What gets there (one physics.step() in ~5e7 of the bench workload, all of them objects squeezed between the fingers: tools/newton_cap_probe.py,
// profiles/r05_newton_cap/) is a warm start inside the wrong facet of the friction cones: the line search ends at the next kink of the piecewise-quadratic cost, step
// lengths of 1e-2 .. 1e-4 from the third iteration on, the gradient does not move for 15 .. 100 iterations -- the fp64 oracle, which runs MuJoCo's own iteration (100
// iterations, tolerance 1e-10), needs 5 .. 100 from the same warm start and 5 .. 8 from qacc_smooth. The problem is strictly convex: both starts end at the same qacc.
// (A healthy solve takes steps of length ~1; the few that need 10 .. 15 iterations do so from either start and are left alone.)
#ifndef NEWTON_RESTART
#define NEWTON_RESTART 6
#endif
#define NEWTON_STALL_ALPHA 0.02f
#ifndef NEWTON_TOL
#define NEWTON_TOL 1e-5f
#endif
#define NEWTON_GRAD_NOISE 3e-6f  // a gradient below this fraction of the two terms it is the difference of is fp32 rounding noise
#ifndef LS_MAXIT
#define LS_MAXIT 16
#endif
#ifndef LS_GTOL
#define LS_GTOL 1e-4f          // relative slope at which the exact line search stops. 1e-3 (MuJoCo's own ls_tolerance is 1e-2) measures +1 % with the same number of
#endif                       // exactly matching fixture lanes (372 of 379), but the solutions get ~75 x more sensitive to their inputs: the warm- against cold-portal
                             // builds end 2.6e-4 m apart after 25 steps instead of 3.5e-6 (tests/test_gpu_parity.py) -- measured in round 4, not taken; 1e-2: +2.5 %, three lanes lost
#define MPR_TOL_F 1e-6f
#define MPR_MAXIT 50
#define LUT_RES 8
#define LUT_CELLS (6 * LUT_RES * LUT_RES)

// diagnostic build only (-DGRIP_STAMPS): per-phase cycle accounting with s_memtime, never in the shipped library
#ifdef GRIP_STAMPS
#define NSTAMP 32
__device__ unsigned long long g_stamp_acc[NSTAMP];
struct Stamps { unsigned long long t; unsigned long long acc[NSTAMP]; };
DEVI unsigned long long stamp_now() { __builtin_amdgcn_sched_barrier(0); unsigned long long t = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); return t; }
#define STAMP(st, i) do { unsigned long long n_ = stamp_now(); (st).acc[i] += n_ - (st).t; (st).t = n_; } while (0)
// event counters of collide(): 0 calls (per env), 1 loop trips with per-lane supports, 2 loop trips in cooperative refinement,
// 3 hull-pair items past the sphere test, 4 of them ended by the remembered direction, 5 contacts from hull pairs,
// 6 hill-climb hops (all lanes), 7 support_vertex calls (all lanes)
__device__ unsigned long long g_dbg_cnt[16];   // 8..11: wave-level (first wave of a workgroup): cycles / trips with per-lane supports, cycles / trips of pure cooperative refinement; 12: cycles of collide() outside the loop
#define DBG_COUNT(i, n) do { if (blockIdx.x == 0) atomicAdd(&g_dbg_cnt[i], (unsigned long long)(n)); } while (0)      // first workgroup only: thousands of lanes on 16 counters would distort the timings
// per env and physics.step() (lane 0 of every env, all workgroups): 0..7 Newton iterations 0..6, 7+; 8..15 contacts 0, 1-2, 3-4, 5-6, 7-8, 9-10, 11-12, 13-14;
// 16 steps, 17 constrained, 18 gripper block constrained (13 x 13 solve), 19 any hull-hull contact, 20 active joint limit, 21 line-search evaluations,
// 24..31 wave trips by the number of Newton iterations the wave ran (max over its envs) 0..6, 7+
__device__ unsigned long long g_dbg_hist[96];   // 64..71 gradient decade at the chosen start; 72 warm start taken; 73 solves; 74..76 hull contacts by cone zone (top, middle, bottom) at the start; 77..79 at the end; 80 hull contacts whose zone changed; 81..83 / 84..86 / 87 the same for floor contacts   // 32 + 8 k + b: after Newton iteration k + 1 (k = 0..3) the scaled gradient was in decade b (< 1e-7, 1e-7.., ..., >= 1e-1)
#define DBG_HIST(i, n) atomicAdd(&g_dbg_hist[i], (unsigned long long)(n))
#ifdef GRIP_HIST            // the per-step histograms (atomics from every env: they distort the phase timings, so they are their own build)
#define HAVE_DBG_HIST 1
#endif
#elif defined(GRIP_MARKS)      // listing build only (hipcc -S -DGRIP_MARKS): phase boundaries as comments in the ISA, to count instructions per phase
#define DBG_COUNT(i, n) do { } while (0)
struct Stamps { int dummy; };
#define STAMP_STR2(x) #x
#define STAMP_STR(x) STAMP_STR2(x)
#define STAMP(st, i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; ==MARK " STAMP_STR(i) " line " STAMP_STR(__LINE__) ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DBG_COUNT(i, n) do { } while (0)
struct Stamps { int dummy; };
#define STAMP(st, i) do { } while (0)
#endif

// diagnostic build only (-DGRIP_CAPDUMP, tools/newton_cap_probe.py): every solve that runs into NEWTON_MAXIT leaves a record -- the state the step started
// from (qpos, qvel, ctrl, qacc_warmstart as the env's LDS vectors hold them while the solver runs), the last iterate, and per Newton iteration the scaled
// gradient, the step length, the cost and the line search's evaluations -- to be replayed on the oracle. Never in the shipped library.
#ifdef GRIP_CAPDUMP
#define CAPDUMP_RECORDS 256
#define CAPDUMP_WORDS 256                // 0 ncon, 1 iters, 2 coupled | grip << 1, 3 took the warm start, 4 kind (1 capped, 2 restarted from qacc_smooth and finished); 8.. qpos[14], 24.. qvel[13], 40.. ctrl[7], 48.. warm[13], 64.. last iterate[13], 80.. qacc_smooth[13], 96.. + 4 it: scaled gradient, alpha, cost, line-search evaluations
__device__ float g_capdump[CAPDUMP_RECORDS][CAPDUMP_WORDS];
__device__ unsigned g_capdump_n, g_capdump_restarts;
#endif

// ---------------------------------------------------------------- cross-lane primitives (16-lane rows)
template <int CTRL> DEVI float dpp_f(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false)); }
#define DPP_ROW_ROR(n) (0x120 + (n))
// All-reduce sum over the 16 lanes of a row. Every lane adds the same pairs and fp add is commutative, so all lanes get
// the bit-identical sum -- PROVIDED the adds stay adds: if the compiler contracts the caller's multiply into the first
// add, lane i computes fma(a_i, b_i, round(a_j b_j)) and lane j fma(a_j, b_j, round(a_i b_i)), which differ in the last
// bit, and the env's lanes then take different branches (line search, convergence tests). Hence contract(off) + opaque.
DEVI float sum16(float x) {
#pragma clang fp contract(off)
    opaque(x);
    x = x + dpp_f<DPP_ROW_ROR(8)>(x); x = x + dpp_f<DPP_ROW_ROR(4)>(x);
    x = x + dpp_f<DPP_ROW_ROR(2)>(x); x = x + dpp_f<DPP_ROW_ROR(1)>(x);
    return x;
}
// broadcast from lane J of each 16-lane row (ds_swizzle bit-mask mode: lane' = (lane & 0x10) | J inside each 32)
template <int J> DEVI float bcast16(float x) { return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), (J << 5) | 0x10)); }
// component `sub` of a vector that every lane holds (static select chain: no dynamic register indexing)
DEVI float pick13(const float (&v)[13], int sub) {
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 13; j++) r = sub == j ? v[j] : r;
    return r;
}
// every lane gets the whole vector whose component i lives in lane i
DEVI void gather13(float xi, float (&v)[13]) {
    v[0] = bcast16<0>(xi); v[1] = bcast16<1>(xi); v[2] = bcast16<2>(xi); v[3] = bcast16<3>(xi); v[4] = bcast16<4>(xi);
    v[5] = bcast16<5>(xi); v[6] = bcast16<6>(xi); v[7] = bcast16<7>(xi); v[8] = bcast16<8>(xi); v[9] = bcast16<9>(xi);
    v[10] = bcast16<10>(xi); v[11] = bcast16<11>(xi); v[12] = bcast16<12>(xi);
}
// ordering point for wave-private LDS traffic (LDS ops of one wave execute in order; this only pins the compiler)
DEVI void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
// 16-bit group of a 64-bit ballot that belongs to this lane's environment
DEVI unsigned group_bits(unsigned long long b, int lane) { return (unsigned)(b >> (lane & 48)) & 0xFFFFu; }
// Lanes 32..63 of a two-env wave are bit-identical clones of lanes 0..31: the same env, the same values, the same LDS traffic, no global
// writes. Where ONE instruction stream can serve two data sets (the support searches of a pair's two hulls, the two halves of a
// cooperative vertex scan, the two halves of a slot loop) the clones take the second set, and v_permlane32_swap hands each half's result
// to the other one: `lo` = what lanes 0..31 hold, `hi` = what lanes 32..63 hold, both in all 64 lanes afterwards -- the halves are clones
// again. (The instruction swaps rows 2, 3 of its first operand with rows 0, 1 of its second; with the same value in both, the first
// comes back as the lower half twice and the second as the upper half twice. tools/hiptests/t_swap.hip checks that on the device.)
DEVI void halves_u(unsigned x, unsigned &lo, unsigned &hi) { auto r_ = __builtin_amdgcn_permlane32_swap(x, x, false, false); lo = r_[0]; hi = r_[1]; }
DEVI void halves_f(float x, float &lo, float &hi) { unsigned a_, b_; halves_u(__float_as_uint(x), a_, b_); lo = __uint_as_float(a_); hi = __uint_as_float(b_); }
DEVI void halves_i(int x, int &lo, int &hi) { unsigned a_, b_; halves_u((unsigned)x, a_, b_); lo = (int)a_; hi = (int)b_; }
#define UPPER_HALF(cx) ((cx).lane >= 32)

struct Kin {
    V3 pe; M3 Re; V3 a4;
    V3 pk[2], ak[2];
    V3 po; M3 Ro;
    V3 c[4];
    float Ic[4][6];
    // the finger chain THIS half of the wave works on (lanes 0..31: left, s = 0; clones 32..63: right, s = 1): c[1 + s], Ic[1 + s], its mass.
    // c[1], c[2], Ic[1], Ic[2] are not filled in this mapping: bias_forces() and mass_matrix() take the own chain and swap results.
    V3 c_own; float Ic_own[6]; float m_own;
    // ... and the big body of the half: the ee + base group (lanes 0..31) or the object (clones): c[0] / c[3], Ic[0] / Ic[3], its mass (c[0], c[3], Ic[0], Ic[3] are not filled)
    V3 c_go; float Ic_go[6]; float m_go;
};

// tables staged in LDS, shared by the workgroup
struct Tables {
    const float *v;                 // hull vertices [nvert][4]
    const unsigned short *nbr;      // neighbour ids, local to the hull
    const unsigned short *lut;      // [6][LUT_CELLS] start vertices
    const float *gt;                // geom table [GN_GEOM][GT_STRIDE]
    const float *lr;                // finger-chain constants [2][LR_STRIDE]
};

struct Ctx {
    int lane, sub;                  // lane in wave, lane in env
    float *envl;                    // this env's LDS region
    Tables T;
};

// The lane's index inside its env as a value the optimiser cannot trace back to threadIdx: an address formed from it (slot, row, state word)
// is then computed where it is used -- two integer instructions -- instead of being hoisted out of the step loop and kept in a register for the
// whole launch (a dozen such addresses were the kernel's last spilled registers).
DEVI int local_sub(const Ctx &cx) { int s_ = cx.sub; asm volatile("" : "+v"(s_)); return s_; }

// A constant the optimiser cannot see through, for the same purpose: an LDS base address `cx.envl + CONSTANT` is loop-invariant, so it was hoisted out of the step loop,
// held in a register across everything and -- the kernel sits at 256 registers -- spilled to scratch and RELOADED inside the Newton loop (two scratch loads and an
// s_waitcnt vmcnt(0) per pricing: a memory round trip where one v_lshl_add would do). An offset that went through local_const() is a value of the current trip.
DEVI int local_const(int c) { asm volatile("" : "+v"(c)); return c; }

constexpr DEVI int pidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// N floats of a 16-byte aligned LDS vector (padded to a multiple of 4) into registers: ceil(N / 4) ds_read_b128, all lanes of an env read
// the same address (broadcast)
template <int N> DEVI void lds_ld(const float *p, float (&v)[N]) {
    const float4 *p4 = reinterpret_cast<const float4 *>(p);
#pragma unroll
    for (int i = 0; i < (N + 3) / 4; i++) {
        const float4 t = p4[i];
        v[4 * i] = t.x;
        if (4 * i + 1 < N) v[4 * i + 1] = t.y;
        if (4 * i + 2 < N) v[4 * i + 2] = t.z;
        if (4 * i + 3 < N) v[4 * i + 3] = t.w;
    }
}
// ... and back (the caller masks to one lane); the padding words are written too
template <int N> DEVI void lds_st(float *p, const float (&v)[N]) {
    float4 *p4 = reinterpret_cast<float4 *>(p);
#pragma unroll
    for (int i = 0; i < (N + 3) / 4; i++)
        p4[i] = make_float4(v[4 * i], 4 * i + 1 < N ? v[4 * i + 1] : 0.f, 4 * i + 2 < N ? v[4 * i + 2] : 0.f, 4 * i + 3 < N ? v[4 * i + 3] : 0.f);
}

// workgroup prologue: hulls + geom table into LDS; returns the context of the calling lane
DEVI Ctx stage_tables(const DevModel &m, float *lds) {
    unsigned *dst = reinterpret_cast<unsigned *>(lds);
    const unsigned *src = m.hull_blob;
    for (int i = threadIdx.x; i < m.hull_words; i += blockDim.x) dst[i] = src[i];
    float *gt = lds + m.hull_words;
    for (int i = threadIdx.x; i < GT_FLOATS; i += blockDim.x) {
        int g = i / GT_STRIDE, f = i % GT_STRIDE;
        float v;
        if (f < 3) v = m.geom_center[g][f];
        else if (f == 3) v = m.geom_rbound[g];
        else if (f == 4) v = m.geom_friction[g][0];
        else if (f == 5) v = m.geom_friction[g][1];
        else if (f == 6) v = m.geom_invweight[g];
        else if (f == 7) v = __int_as_float(m.geom_group[g]);
        else if (f == 8) v = __int_as_float(g >= 1 ? m.hull_vadr[g - 1] : 0);
        else v = __int_as_float(g >= 1 ? m.hull_vnum[g - 1] : 0);
        gt[i] = v;
    }
    float *lr = lds + LR_OFF(m.hull_words);
    for (int i = threadIdx.x; i < LR_FLOATS; i += blockDim.x) {
        const int s = i / LR_STRIDE, f = i % LR_STRIDE, grp = s < 2 ? 1 + s : (s == 2 ? 0 : 3);
        float v = 0.f;
        if (f < 24) {
            if (s >= 2) v = 0.f;
            else if (f < 3) v = m.kn_pos[s][f];
            else if (f < 12) v = m.kn_R[s][f - 3];
            else if (f < 15) v = m.fin_pos[s][f - 12];
            else v = m.fin_R[s][f - 15];
        }
        else if (f < 27) v = m.grp_com[grp][f - 24];
        else if (f < 33) v = m.grp_inertia[grp][f - 27];
        else if (f == 33) v = m.grp_mass[grp];
        lr[i] = v;
    }
    __syncthreads();
    Ctx c;
    c.lane = threadIdx.x & (WAVE - 1); c.sub = threadIdx.x & (KL - 1);
    c.T.v = reinterpret_cast<const float *>(dst);
    c.T.nbr = reinterpret_cast<const unsigned short *>(dst + m.hull_off_nbr);
    c.T.lut = reinterpret_cast<const unsigned short *>(dst + m.hull_off_lut);
    c.T.gt = gt; c.T.lr = lr;
    bool act_; c.envl = lds + LDS_ENV_BASE(m.hull_words) + wg_env_slot(act_) * ENV_FLOATS;
    return c;
}

// fault bits of the env (1 diverged, 2 contact overflow, 4 Newton iteration limit): a word of the env's LDS region, OR-ed by the lanes that see
// the condition (the env's 16 lanes always agree)
DEVI void env_fault_or(const Ctx &cx, int bits) { MI(cx, MI_FAULT) |= bits; }
DEVI int env_fault(const Ctx &cx) { return MI(cx, MI_FAULT); }

// ---------------------------------------------------------------- kinematics (redundant in the 16 lanes)
DEVI void store_frame(float *envl, int g, V3 p, const M3 &R) {
    float4 *f = reinterpret_cast<float4 *>(envl + EF_FRAMES + (g - 1) * 12);
    f[0] = make_float4(p.x, p.y, p.z, R.m[0]); f[1] = make_float4(R.m[1], R.m[2], R.m[3], R.m[4]); f[2] = make_float4(R.m[5], R.m[6], R.m[7], R.m[8]);
}
DEVI void load_frame(const float *envl, int g, V3 &p, M3 &R) {
    const float4 *f = reinterpret_cast<const float4 *>(envl + EF_FRAMES + (g - 1) * 12);
    const float4 a = f[0], b = f[1], c = f[2];
    p = v3(a.x, a.y, a.z);
    R.m[0] = a.w; R.m[1] = b.x; R.m[2] = b.y; R.m[3] = b.z; R.m[4] = b.w; R.m[5] = c.x; R.m[6] = c.y; R.m[7] = c.z; R.m[8] = c.w;
}

// what make_constraints() needs of the kinematics, parked in LDS across collide() (ES_KIN)
struct KinC { V3 pe, a4, pk[2], ak[2], po; M3 Ro; };
DEVI void kinc_store(float *envl, const V3 &pe, const V3 &a4, const V3 (&pk)[2], const V3 (&ak)[2], const V3 &po, const M3 &Ro) {
    float4 *f = reinterpret_cast<float4 *>(envl + ES_KIN);
    f[0] = make_float4(pe.x, pe.y, pe.z, a4.x); f[1] = make_float4(a4.y, a4.z, pk[0].x, pk[0].y); f[2] = make_float4(pk[0].z, pk[1].x, pk[1].y, pk[1].z);
    f[3] = make_float4(ak[0].x, ak[0].y, ak[0].z, ak[1].x); f[4] = make_float4(ak[1].y, ak[1].z, po.x, po.y);
    f[5] = make_float4(po.z, Ro.m[0], Ro.m[1], Ro.m[2]); f[6] = make_float4(Ro.m[3], Ro.m[4], Ro.m[5], Ro.m[6]); f[7] = make_float4(Ro.m[7], Ro.m[8], 0.f, 0.f);
}
DEVI void kinc_load(const float *envl, KinC &k) {
    const float4 *f = reinterpret_cast<const float4 *>(envl + ES_KIN);
    const float4 a = f[0], b = f[1], c = f[2], d = f[3], e = f[4], g = f[5], h = f[6], i = f[7];
    k.pe = v3(a.x, a.y, a.z); k.a4 = v3(a.w, b.x, b.y); k.pk[0] = v3(b.z, b.w, c.x); k.pk[1] = v3(c.y, c.z, c.w);
    k.ak[0] = v3(d.x, d.y, d.z); k.ak[1] = v3(d.w, e.x, e.y); k.po = v3(e.z, e.w, g.x);
    k.Ro.m[0] = g.y; k.Ro.m[1] = g.z; k.Ro.m[2] = g.w; k.Ro.m[3] = h.x; k.Ro.m[4] = h.y; k.Ro.m[5] = h.z; k.Ro.m[6] = h.w; k.Ro.m[7] = i.x; k.Ro.m[8] = i.y;
}

DEVI void kinematics(const DevModel &m, float (&qpos)[14], Kin &k, const Ctx &cx, bool store) {
    // free-joint quaternion normalised in place (mj_kinematics does the same)
    float qn = sqrtf(qpos[10] * qpos[10] + qpos[11] * qpos[11] + qpos[12] * qpos[12] + qpos[13] * qpos[13]);
    if (qn < 1e-15f) { qpos[10] = 1.f; qpos[11] = qpos[12] = qpos[13] = 0.f; }
    else { float iq = 1.0f / qn; qpos[10] *= iq; qpos[11] *= iq; qpos[12] *= iq; qpos[13] *= iq; }
    k.pe = v3(m.ee_pos0[0] + qpos[0], m.ee_pos0[1] + qpos[1], m.ee_pos0[2] + qpos[2]);
    // the four joint angles' sines and cosines up front, side by side (four independent ~40-instruction chains instead of one after the other
    // along the tree), and every LDS store in ONE block at the end: the whole tree is straight-line code for the instruction scheduler
    float sr, cr, sy, cy, sq[2], cq[2];
    sincos_joint(qpos[3], sr, cr); sincos_joint(qpos[4], sy, cy);
    sincos_joint(UPPER_HALF(cx) ? qpos[6] : qpos[5], sq[0], cq[0]); sq[1] = sq[0]; cq[1] = cq[0];      // the half's own knuckle angle
    // Re = Rx(roll) * Rz(yaw)
    k.Re.m[0] = cy;      k.Re.m[1] = -sy;     k.Re.m[2] = 0.f;
    k.Re.m[3] = cr * sy; k.Re.m[4] = cr * cy; k.Re.m[5] = -sr;
    k.Re.m[6] = sr * sy; k.Re.m[7] = sr * cy; k.Re.m[8] = cr;
    k.a4 = v3(0.f, -sr, cr);
    const V3 pb = k.pe + mulv(k.Re, ldv(m.base_pos));
    const M3 Rb = mulm(k.Re, ldm(m.base_R));
    // One finger chain per half of the wave: lanes 0..31 the left knuckle + finger, their clones 32..63 the right ones, the same instructions on
    // constants read by lane from the workgroup's table (stage_tables); pk / ak change hands (the constraint rows and the wrench projection want
    // both), the chain's COM and inertia stay with the half (bias_forces, mass_matrix).
    const bool up = UPPER_HALF(cx);
    V3 pk_o, pf_o; M3 Rk_o, Rf_o;
    {   const float4 *L4 = reinterpret_cast<const float4 *>(cx.T.lr + (up ? LR_STRIDE : 0));
        const float4 a0 = L4[0], a1 = L4[1], a2 = L4[2], a3 = L4[3], a4_ = L4[4], a5 = L4[5], a6 = L4[6], a7 = L4[7], a8 = L4[8];
        const V3 knp = v3(a0.x, a0.y, a0.z);
        M3 knR; knR.m[0] = a0.w; knR.m[1] = a1.x; knR.m[2] = a1.y; knR.m[3] = a1.z; knR.m[4] = a1.w; knR.m[5] = a2.x; knR.m[6] = a2.y; knR.m[7] = a2.z; knR.m[8] = a2.w;
        const V3 fnp = v3(a3.x, a3.y, a3.z);
        M3 fnR; fnR.m[0] = a3.w; fnR.m[1] = a4_.x; fnR.m[2] = a4_.y; fnR.m[3] = a4_.z; fnR.m[4] = a4_.w; fnR.m[5] = a5.x; fnR.m[6] = a5.y; fnR.m[7] = a5.z; fnR.m[8] = a5.w;
        const V3 com = v3(a6.x, a6.y, a6.z);
        const float inertia[6] = {a6.w, a7.x, a7.y, a7.z, a7.w, a8.x};
        k.m_own = a8.y;
        pk_o = pb + mulv(Rb, knp);
        const M3 Rk0 = mulm(Rb, knR);
        const V3 ak_o = col(Rk0, 1);
        const float sq_ = up ? sq[1] : sq[0], cq_ = up ? cq[1] : cq[0];
        M3 Ry; Ry.m[0] = cq_; Ry.m[1] = 0; Ry.m[2] = sq_; Ry.m[3] = 0; Ry.m[4] = 1; Ry.m[5] = 0; Ry.m[6] = -sq_; Ry.m[7] = 0; Ry.m[8] = cq_;
        Rk_o = mulm(Rk0, Ry);
        pf_o = pk_o + mulv(Rk_o, fnp);
        Rf_o = mulm(Rk_o, fnR);
        k.c_own = pk_o + mulv(Rk_o, com);
        rot_sym(Rk_o, inertia, k.Ic_own);
        halves_f(pk_o.x, k.pk[0].x, k.pk[1].x); halves_f(pk_o.y, k.pk[0].y, k.pk[1].y); halves_f(pk_o.z, k.pk[0].z, k.pk[1].z);
        halves_f(ak_o.x, k.ak[0].x, k.ak[1].x); halves_f(ak_o.y, k.ak[0].y, k.ak[1].y); halves_f(ak_o.z, k.ak[0].z, k.ak[1].z);
    }
    k.po = v3(qpos[7], qpos[8], qpos[9]);
    k.Ro = quat_mat(qpos[10], qpos[11], qpos[12], qpos[13]);
    {   // the ee + base group in lanes 0..31, the object in the clones: one COM and one rotated inertia per half, constants by lane from the table
        const float4 *G4 = reinterpret_cast<const float4 *>(cx.T.lr + (up ? 3 : 2) * LR_STRIDE + 24);
        const float4 g0 = G4[0], g1 = G4[1], g2 = G4[2];
        const float inertia[6] = {g0.w, g1.x, g1.y, g1.z, g1.w, g2.x};
        M3 Rs;
#pragma unroll
        for (int i = 0; i < 9; i++) Rs.m[i] = up ? k.Ro.m[i] : k.Re.m[i];
        const V3 ps = up ? k.po : k.pe;
        k.m_go = g2.y;
        k.c_go = ps + mulv(Rs, v3(g0.x, g0.y, g0.z));
        rot_sym(Rs, inertia, k.Ic_go);
    }
    if (store && cx.sub == 0) {                     // one lane of the env publishes the geom frames, what the constraint rows need, the state's quaternion
        store_frame(cx.envl, 1, pb, Rb);
        store_frame(cx.envl, up ? 4 : 2, pk_o, Rk_o); store_frame(cx.envl, up ? 5 : 3, pf_o, Rf_o);       // each half publishes its own chain's frames
        store_frame(cx.envl, 6, k.po, k.Ro);
        kinc_store(cx.envl, k.pe, k.a4, k.pk, k.ak, k.po, k.Ro);
        // the normalised quaternion is the state's (mj_kinematics normalises qpos in place)
        float4 *q4 = reinterpret_cast<float4 *>(cx.envl + ES_QPOS);
        q4[2] = make_float4(qpos[8], qpos[9], qpos[10], qpos[11]); q4[3] = make_float4(qpos[12], qpos[13], 0.f, 0.f);
    }
}

// ---------------------------------------------------------------- mass matrix: gripper 7x7 (Mg) and object 6x6 (Mo), packed lower
template <int NC, int NM>
DEVI void add_body(float (&Mp)[NM], const int (&dofs)[NC], const V3 (&jp)[NC], const V3 (&jr)[NC], float mass, const float *Ic) {
#pragma unroll
    for (int a = 0; a < NC; a++) {
        V3 Ia = symv(Ic, jr[a]);
#pragma unroll
        for (int b = 0; b <= a; b++) Mp[pidx(dofs[a], dofs[b])] += mass * dot(jp[a], jp[b]) + dot(Ia, jr[b]);
    }
}

DEVI void mass_matrix(const DevModel &m, const Kin &k, float (&Mg)[28], float (&Mo)[21], bool up) {
#pragma unroll
    for (int i = 0; i < 28; i++) Mg[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 21; i++) Mo[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 7; i++) Mg[pidx(i, i)] = m.armature[i];
#pragma unroll
    for (int i = 0; i < 6; i++) Mo[pidx(i, i)] = m.armature[7 + i];
    const V3 ex = v3(1, 0, 0), ey = v3(0, 1, 0), ez = v3(0, 0, 1), z0 = v3(0, 0, 0);
    {   // the half's big body as a 6 x 6 block: the ee + base group over (ee dofs 0..4, a zero sixth axis) in lanes 0..31, the object over its six dofs in the
        // clones; the halves swap blocks, the gripper's matrix takes the lower half's (first, as in the one-stream form: G, L, R), the object's the upper half's
        const int dofs[6] = {0, 1, 2, 3, 4, 5};
        const V3 ref = up ? k.po : k.pe;
        const V3 a3 = up ? col(k.Ro, 0) : ex, a4v = up ? col(k.Ro, 1) : k.a4, a5 = up ? col(k.Ro, 2) : z0;
        const V3 r = k.c_go - ref;
        const V3 jp[6] = {ex, ey, ez, cross(a3, r), cross(a4v, r), cross(a5, r)};
        const V3 jr[6] = {z0, z0, z0, a3, a4v, a5};
        float Mb2[21];
#pragma unroll
        for (int i = 0; i < 21; i++) Mb2[i] = 0.f;
        add_body<6, 21>(Mb2, dofs, jp, jr, k.m_go, k.Ic_go);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) {
                float lo, hi; halves_f(Mb2[pidx(a, b)], lo, hi);
                if (a < 5) Mg[pidx(a, b)] += lo;
                Mo[pidx(a, b)] += hi;
            }
    }
    {   // the half's own finger body as a 6 x 6 block over (ee dofs 0..4, own knuckle), the same expressions as add_body on the full matrix;
        // the halves swap blocks and both add left, then right, in the order of the one-stream form (G, L, R): the same sums
        const int dofs[6] = {0, 1, 2, 3, 4, 5};
        const V3 pk_o = up ? k.pk[1] : k.pk[0], ak_o = up ? k.ak[1] : k.ak[0];
        V3 r = k.c_own - k.pe, rk = k.c_own - pk_o;
        const V3 jp[6] = {ex, ey, ez, cross(ex, r), cross(k.a4, r), cross(ak_o, rk)};
        const V3 jr[6] = {z0, z0, z0, ex, k.a4, ak_o};
        float Mb[21];
#pragma unroll
        for (int i = 0; i < 21; i++) Mb[i] = 0.f;
        add_body<6, 21>(Mb, dofs, jp, jr, k.m_own, k.Ic_own);
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b <= a; b++) {
                float lo, hi; halves_f(Mb[pidx(a, b)], lo, hi);
                if (a < 5) { Mg[pidx(a, b)] += lo; Mg[pidx(a, b)] += hi; }
                else { Mg[pidx(5, b)] += lo; Mg[pidx(6, b == 5 ? 6 : b)] += hi; }
            }
    }
}

// y = M x with the block-diagonal mass matrix
DEVI void mass_mulv(const float (&Mg)[28], const float (&Mo)[21], const float (&x)[13], float (&y)[13]) {
#pragma unroll
    for (int i = 0; i < 7; i++) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 7; j++) s = fmaf(Mg[pidx(i, j)], x[j], s);
        y[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 6; j++) s = fmaf(Mo[pidx(i, j)], x[7 + j], s);
        y[7 + i] = s;
    }
}

// ---------------------------------------------------------------- group twists / wrenches
struct Twist { V3 vG, wG, vL, wL, vR, wR, vO, wO; };

DEVI void twists(const Kin &k, const float (&x)[13], Twist &t) {
    t.vG = v3(x[0], x[1], x[2]);
    t.wG = v3(x[3], 0.f, 0.f) + k.a4 * x[4];
    t.wL = t.wG + k.ak[0] * x[5]; t.vL = t.vG + cross(t.wG, k.pk[0] - k.pe);
    t.wR = t.wG + k.ak[1] * x[6]; t.vR = t.vG + cross(t.wG, k.pk[1] - k.pe);
    t.vO = v3(x[7], x[8], x[9]);
    t.wO = mulv(k.Ro, v3(x[10], x[11], x[12]));
}

DEVI void group_motion(const Kin &k, const Twist &t, int g, V3 p, V3 &v, V3 &w) {
    V3 ref, v0;
    if (g == GRP_G) { ref = k.pe; v0 = t.vG; w = t.wG; }
    else if (g == GRP_L) { ref = k.pk[0]; v0 = t.vL; w = t.wL; }
    else if (g == GRP_R) { ref = k.pk[1]; v0 = t.vR; w = t.wR; }
    else if (g == GRP_O) { ref = k.po; v0 = t.vO; w = t.wO; }
    else { ref = v3(0, 0, 0); v0 = v3(0, 0, 0); w = v3(0, 0, 0); }
    v = v0 + cross(w, p - ref);
}

struct Wrench { V3 FG, TG, FL, TL, FR, TR, FO, TO; };
DEVI void wrench_zero(Wrench &w) { w.FG = w.TG = w.FL = w.TL = w.FR = w.TR = w.FO = w.TO = v3(0, 0, 0); }
DEVI void wrench_add(const Kin &k, Wrench &w, int g, V3 p, V3 F, V3 T, float sgn) {
    F = F * sgn; T = T * sgn;
    if (g == GRP_G) { w.FG = w.FG + F; w.TG = w.TG + T + cross(p - k.pe, F); }
    else if (g == GRP_L) { w.FL = w.FL + F; w.TL = w.TL + T + cross(p - k.pk[0], F); }
    else if (g == GRP_R) { w.FR = w.FR + F; w.TR = w.TR + T + cross(p - k.pk[1], F); }
    else if (g == GRP_O) { w.FO = w.FO + F; w.TO = w.TO + T + cross(p - k.po, F); }
}
DEVI void wrench_project(const Kin &k, const Wrench &w, float (&q)[13]) {
    q[5] = dot(k.ak[0], w.TL); q[6] = dot(k.ak[1], w.TR);
    V3 F = w.FG + w.FL + w.FR;
    V3 T = w.TG + w.TL + cross(k.pk[0] - k.pe, w.FL) + w.TR + cross(k.pk[1] - k.pe, w.FR);
    q[0] = F.x; q[1] = F.y; q[2] = F.z; q[3] = T.x; q[4] = dot(k.a4, T);
    q[7] = w.FO.x; q[8] = w.FO.y; q[9] = w.FO.z;
    V3 tl = multv(k.Ro, w.TO);
    q[10] = tl.x; q[11] = tl.y; q[12] = tl.z;
}

// one Jacobian row expanded to 13 entries: linear (e . relative point velocity) or angular (e . relative w)
DEVI void row_add(const Kin &k, float (&j)[13], int g, V3 p, V3 e, float sgn, bool angular) {
    if (g == GRP_WORLD) return;
    if (g == GRP_O) {
        V3 a = angular ? e : cross(p - k.po, e);
        V3 al = multv(k.Ro, a);
        if (!angular) { j[7] += sgn * e.x; j[8] += sgn * e.y; j[9] += sgn * e.z; }
        j[10] += sgn * al.x; j[11] += sgn * al.y; j[12] += sgn * al.z;
        return;
    }
    V3 a = angular ? e : cross(p - k.pe, e);
    if (!angular) { j[0] += sgn * e.x; j[1] += sgn * e.y; j[2] += sgn * e.z; }
    j[3] += sgn * a.x; j[4] += sgn * dot(k.a4, a);
    if (g == GRP_L) { V3 b = angular ? e : cross(p - k.pk[0], e); j[5] += sgn * dot(k.ak[0], b); }
    if (g == GRP_R) { V3 b = angular ? e : cross(p - k.pk[1], e); j[6] += sgn * dot(k.ak[1], b); }
}

// ---------------------------------------------------------------- bias forces (gravity + Coriolis/centrifugal)
DEVI void bias_forces(const DevModel &m, const Kin &k, const float (&qvel)[13], float (&bias)[13], bool up) {
    Twist t; twists(k, qvel, t);
    V3 alG = cross(v3(qvel[3], 0, 0), k.a4 * qvel[4]);
    Wrench w; wrench_zero(w);
    V3 grav = v3(0, 0, m.gravity_z);
    {   // the half's own finger body (left in lanes 0..31, right in the clones), the expressions of the one-stream form; force and torque
        // about the knuckle change hands
        const V3 pk_o = up ? k.pk[1] : k.pk[0], ak_o = up ? k.ak[1] : k.ak[0], w_o = up ? t.wR : t.wL;
        const float qv_o = up ? qvel[6] : qvel[5];
        const V3 s_o = pk_o - k.pe;
        const V3 a_o = cross(alG, s_o) + cross(t.wG, cross(t.wG, s_o));
        const V3 al_o = alG + cross(t.wG, ak_o * qv_o);
        const V3 rc = k.c_own - pk_o;
        const V3 ac = a_o + cross(al_o, rc) + cross(w_o, cross(w_o, rc)) - grav;
        const V3 tau = symv(k.Ic_own, al_o) + cross(w_o, symv(k.Ic_own, w_o));
        const V3 F = ac * k.m_own;
        const V3 T = tau + cross(k.c_own - pk_o, F);
        halves_f(F.x, w.FL.x, w.FR.x); halves_f(F.y, w.FL.y, w.FR.y); halves_f(F.z, w.FL.z, w.FR.z);
        halves_f(T.x, w.TL.x, w.TR.x); halves_f(T.y, w.TL.y, w.TR.y); halves_f(T.z, w.TL.z, w.TR.z);
    }
    {   // the half's big body (ee + base group | object): the one-stream expressions with the object's zero angular acceleration written out
        const V3 ref = up ? k.po : k.pe, w_ = up ? t.wO : t.wG, al_ = up ? v3(0, 0, 0) : alG;
        const V3 rc = k.c_go - ref;
        const V3 ac = cross(al_, rc) + cross(w_, cross(w_, rc)) - grav;
        const V3 tau = symv(k.Ic_go, al_) + cross(w_, symv(k.Ic_go, w_));
        const V3 F = ac * k.m_go;
        const V3 T = tau + cross(k.c_go - ref, F);
        halves_f(F.x, w.FG.x, w.FO.x); halves_f(F.y, w.FG.y, w.FO.y); halves_f(F.z, w.FG.z, w.FO.z);
        halves_f(T.x, w.TG.x, w.TO.x); halves_f(T.y, w.TG.y, w.TO.y); halves_f(T.z, w.TG.z, w.TO.z);
    }
    wrench_project(k, w, bias);
}

// ---------------------------------------------------------------- small dense helpers in registers (packed lower, static indices)
template <int N>
DEVI void chol_packed(float *A) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        float s = A[pidx(j, j)];
#pragma unroll
        for (int q = 0; q < j; q++) s -= A[pidx(j, q)] * A[pidx(j, q)];
        float inv = rsqrtf(fmaxf(s, 1e-30f));
        A[pidx(j, j)] = inv;                       // the diagonal holds 1 / L_jj
#pragma unroll
        for (int i = j + 1; i < N; i++) {
            float t = A[pidx(i, j)];
#pragma unroll
            for (int q = 0; q < j; q++) t -= A[pidx(i, q)] * A[pidx(j, q)];
            A[pidx(i, j)] = t * inv;
        }
    }
}
template <int N>
DEVI void chol_solve_packed(const float *L, float *x) {
#pragma unroll
    for (int i = 0; i < N; i++) {
        float s = x[i];
#pragma unroll
        for (int q = 0; q < i; q++) s -= L[pidx(i, q)] * x[q];
        x[i] = s * L[pidx(i, i)];
    }
#pragma unroll
    for (int i = N - 1; i >= 0; i--) {
        float s = x[i];
#pragma unroll
        for (int q = i + 1; q < N; q++) s -= L[pidx(q, i)] * x[q];
        x[i] = s * L[pidx(i, i)];
    }
}
// x <- blockdiag(Ag, Ao)^-1 x for the gripper 7x7 and object 6x6 blocks (Ag, Ao are destroyed)
DEVI void block_solve(float (&Ag)[28], float (&Ao)[21], float (&x)[13]) {
    chol_packed<7>(Ag); chol_packed<6>(Ao);
    float xg[7], xo[6];
#pragma unroll
    for (int i = 0; i < 7; i++) xg[i] = x[i];
#pragma unroll
    for (int i = 0; i < 6; i++) xo[i] = x[7 + i];
    chol_solve_packed<7>(Ag, xg); chol_solve_packed<6>(Ao, xo);
#pragma unroll
    for (int i = 0; i < 7; i++) x[i] = xg[i];
#pragma unroll
    for (int i = 0; i < 6; i++) x[7 + i] = xo[i];
}

// ---------------------------------------------------------------- dense algebra, one matrix row per lane
// Lane i (i < 13) of an env holds row i of a symmetric positive definite 13 x 13 matrix. Cholesky in place: on exit
// row[q < i] = L[i][q] and row[i] = 1 / L[i][i]. Row j is handed to the other lanes with ds_swizzle broadcasts.
// LO > 0: the matrix is block diagonal with a leading LO x LO block that the caller does not need (rows >= LO have zeros in
// columns < LO): only the trailing block is factorised / solved, lanes < LO are left alone (their solution component is 0).
template <int J, int LO = 0>
DEVI void chol_col(float (&row)[13], int sub) {
    // one ds_swizzle round trip per column: lane J's L[J][0..J-1] and A[J][J] go to everyone, every lane then forms both its
    // own entry and the pivot 1 / L[J][J] (redundantly, bit-identically)
    float d = bcast16<J>(row[J]), t = row[J];
#pragma unroll
    for (int q = LO; q < J; q++) { float ljq = bcast16<J>(row[q]); d = fmaf(-ljq, ljq, d); t = fmaf(-row[q], ljq, t); }
    float inv = rsqrtf(fmaxf(d, 1e-30f));
    row[J] = sub == J ? inv : t * inv;                         // lanes below J: junk in the unused upper triangle
}
DEVI void chol_rows(float (&row)[13], int sub) {
    chol_col<0>(row, sub); chol_col<1>(row, sub); chol_col<2>(row, sub); chol_col<3>(row, sub); chol_col<4>(row, sub);
    chol_col<5>(row, sub); chol_col<6>(row, sub); chol_col<7>(row, sub); chol_col<8>(row, sub); chol_col<9>(row, sub);
    chol_col<10>(row, sub); chol_col<11>(row, sub); chol_col<12>(row, sub);
}
DEVI void chol_rows_obj(float (&row)[13], int sub) {            // trailing 6 x 6 (object) block only
    chol_col<7, 7>(row, sub); chol_col<8, 7>(row, sub); chol_col<9, 7>(row, sub); chol_col<10, 7>(row, sub); chol_col<11, 7>(row, sub);
    chol_col<12, 7>(row, sub);
}
// solves L L^T x = b with b_i in lane i; returns x_i in lane i (0 in lanes 13..15)
template <int J> DEVI void fwd_step(const float (&row)[13], int sub, float &acc, float &y) {
    float yj = bcast16<J>(acc * row[J]);
    y = sub == J ? yj : y;
    acc = sub > J ? fmaf(-row[J], yj, acc) : acc;
}
template <int I> DEVI void bwd_step(const float (&row)[13], int sub, float y, float &x) {
    float s = sum16(sub > I && sub < 13 ? row[I] * x : 0.f);   // sum over q > I of L[q][I] x_q
    x = sub == I ? (y - s) * row[I] : x;
}
DEVI float chol_solve_rows(const float (&row)[13], float b, int sub) {
    float acc = b, y = 0.f, x = 0.f;
    fwd_step<0>(row, sub, acc, y); fwd_step<1>(row, sub, acc, y); fwd_step<2>(row, sub, acc, y); fwd_step<3>(row, sub, acc, y);
    fwd_step<4>(row, sub, acc, y); fwd_step<5>(row, sub, acc, y); fwd_step<6>(row, sub, acc, y); fwd_step<7>(row, sub, acc, y);
    fwd_step<8>(row, sub, acc, y); fwd_step<9>(row, sub, acc, y); fwd_step<10>(row, sub, acc, y); fwd_step<11>(row, sub, acc, y);
    fwd_step<12>(row, sub, acc, y);
    bwd_step<12>(row, sub, y, x); bwd_step<11>(row, sub, y, x); bwd_step<10>(row, sub, y, x); bwd_step<9>(row, sub, y, x);
    bwd_step<8>(row, sub, y, x); bwd_step<7>(row, sub, y, x); bwd_step<6>(row, sub, y, x); bwd_step<5>(row, sub, y, x);
    bwd_step<4>(row, sub, y, x); bwd_step<3>(row, sub, y, x); bwd_step<2>(row, sub, y, x); bwd_step<1>(row, sub, y, x);
    bwd_step<0>(row, sub, y, x);
    return x;
}
DEVI float chol_solve_rows_obj(const float (&row)[13], float b, int sub) {
    float acc = b, y = 0.f, x = 0.f;
    fwd_step<7>(row, sub, acc, y); fwd_step<8>(row, sub, acc, y); fwd_step<9>(row, sub, acc, y); fwd_step<10>(row, sub, acc, y);
    fwd_step<11>(row, sub, acc, y); fwd_step<12>(row, sub, acc, y);
    bwd_step<12>(row, sub, y, x); bwd_step<11>(row, sub, y, x); bwd_step<10>(row, sub, y, x); bwd_step<9>(row, sub, y, x);
    bwd_step<8>(row, sub, y, x); bwd_step<7>(row, sub, y, x);
    return sub >= 7 ? x : 0.f;
}
// Newton direction by a REDUNDANT factorisation. The row-distributed Cholesky above pays one ds_swizzle round trip per column and per
// substitution step (39 dependent LDS-crossbar trips per solve) and a wave that is alone on its SIMD waits every one of them out. Here
// each lane publishes its row of H and its gradient component once, every lane reads the whole (lower triangle of the) system back
// -- one round trip -- and factorises it in registers, all 16 lanes bit-identically: about the same number of VALU instructions, no
// communication, and the direction comes out as a full vector in every lane (no second trip through EF_P).
// LO = 7: only the trailing 6 x 6 (object) block, the leading components of p are 0.
template <int LO>
DEVI void gathered_solve(const Ctx &cx, const float (&row)[13], float gi, float (&p)[13]) {
    constexpr int N = 13 - LO;
    float *H = cx.envl + EF_H;
    if (cx.sub < 13) {
        float *hr = H + cx.sub * 16;
#pragma unroll
        for (int j = LO; j < 13; j++) hr[j] = row[j];
        cx.envl[EF_HG + cx.sub] = -gi;
    }
    wave_sync();
    float A[N * (N + 1) / 2], x[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) A[pidx(i, j)] = H[(LO + i) * 16 + LO + j];
        x[i] = cx.envl[EF_HG + LO + i];
    }
    wave_sync();                                        // the area is rewritten by the next pricing: reads first (one wave, in order)
    chol_packed<N>(A);
    chol_solve_packed<N>(A, x);
#pragma unroll
    for (int i = 0; i < 13; i++) p[i] = 0.f;
#pragma unroll
    for (int i = 0; i < N; i++) p[LO + i] = x[i];
}

// The same for a block-diagonal H: every lane factorises only ITS block -- gripper 7 x 7 in lanes 0..6, object 6 x 6 (padded with an
// identity row) in the others -- 28 packed entries instead of 91, a fifth of the multiply-adds; both blocks at once, in different lanes.
// Lane i < 13 publishes the seven entries of its row and its gradient component as two 16-byte stores, reads its block back (14 16-byte
// loads) and returns the block's part of the Newton direction in x (x[6] = 0 in the object block).
DEVI void gathered_solve_block(const Ctx &cx, const float (&row)[7], float gi, float (&x)[7]) {
    float4 *H4 = reinterpret_cast<float4 *>(cx.envl + EF_H);
    const int lsub = local_sub(cx);
    if (lsub < 13) { H4[2 * lsub] = make_float4(row[0], row[1], row[2], row[3]); H4[2 * lsub + 1] = make_float4(row[4], row[5], row[6], -gi); }
    wave_sync();
    const bool grip = lsub < 7;
    const int base = grip ? 0 : 7;
    float A[28];
#pragma unroll
    for (int i = 0; i < 7; i++) {
        // row 13 does not exist: the object block's seventh row is the identity, selected per entry (reading it from an LDS row that lane 13 writes once per
        // solve instead was measured: -0.5 %, the extra store and its wait cost more than the ~30 selects)
        const int r = (i < 6 || grip) ? base + i : base;
        const float4 a = H4[2 * r], b = H4[2 * r + 1];
        const float e[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
        const bool pad = i == 6 && !grip;
#pragma unroll
        for (int j = 0; j <= i; j++) A[pidx(i, j)] = pad ? (j == 6 ? 1.f : 0.f) : e[j];
        x[i] = pad ? 0.f : b.w;
    }
    wave_sync();                                        // the area is rewritten by the next pricing: reads first (one wave, in order)
    chol_packed<7>(A);
    chol_solve_packed<7>(A, x);
}

// ---------------------------------------------------------------- collision
DEVI void make_tangents(V3 n, V3 &t1, V3 &t2) {
    t1 = fabsf(n.y) < 0.5f ? v3(0, 1, 0) : v3(0, 0, 1);
    t1 = normalized(t1 - n * dot(n, t1));
    t2 = cross(n, t1);
}

// Support vertex of the hull starting at vertex offset `base` (cube-map table `h`) for the LOCAL direction dl:
// start at the table entry and hill-climb the edge graph to the best neighbour until none improves. Six
// neighbour tests per iteration so that their dependent index -> vertex LDS hops overlap (2: -1.5 %, 4: -0.5 % against 6, same box).
#ifndef SUP_NB
#define SUP_NB 6                // neighbours tested per pass of the hill climb
#endif
// `adj` (out): the support vertex's adjacency range as packed in its 4th word (CSR offset | degree << 16).
DEVI int support_vertex(const Tables &T, int h, int base, V3 dl, V3 &vout, int hint, unsigned &adj) {
    float ax = fabsf(dl.x), ay = fabsf(dl.y), az = fabsf(dl.z);
    int axis = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    float mj = axis == 0 ? dl.x : axis == 1 ? dl.y : dl.z;
    float u = axis == 0 ? dl.y : dl.x, v = axis == 2 ? dl.y : dl.z;
    float im = rcp(fmaxf(fabsf(mj), 1e-30f));
    int iu = min(LUT_RES - 1, max(0, (int)((u * im + 1.f) * (0.5f * LUT_RES))));
    int iv = min(LUT_RES - 1, max(0, (int)((v * im + 1.f) * (0.5f * LUT_RES))));
    int cell = (2 * axis + (mj < 0.f ? 1 : 0)) * (LUT_RES * LUT_RES) + iu * LUT_RES + iv;
    const float4 *vb = reinterpret_cast<const float4 *>(T.v) + base;      // vertices are padded to 16 bytes: one ds_read_b128 each
    int cur = hint >= 0 ? hint : T.lut[h * LUT_CELLS + cell];       // any start climbs to the same maximum on a convex hull
    const float4 v0 = vb[cur];
    float bx = v0.x, by = v0.y, bz = v0.z;
    float bv = fmaf(bx, dl.x, fmaf(by, dl.y, bz * dl.z));
    // the vertex's own 4th word holds its adjacency range: moving to a neighbour needs no trip through the CSR offset table
    unsigned aw = __float_as_uint(v0.w);
    int e = (int)(aw & 0xffffu), eend = e + (int)(aw >> 16);
    int cand = cur; float cv = bv;
#ifdef GRIP_STAMPS
    const unsigned long long act_ = __ballot(1);
    const bool first_ = (__ffsll((long long)act_) - 1) == (int)(threadIdx.x & 63);
    const int nact_ = __popcll(act_);
    if (first_) { DBG_COUNT(15, 1); DBG_COUNT(12, nact_); }
#endif
    for (int guard = 0; guard < 2048; guard++) {
#ifdef GRIP_STAMPS
        if ((__ffsll((long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63)) DBG_COUNT(14, 1);
#endif
        int j[SUP_NB]; float x[SUP_NB], y[SUP_NB], z[SUP_NB];
#pragma unroll
        for (int q = 0; q < SUP_NB; q++) j[q] = T.nbr[min(e + q, eend - 1)];
#pragma unroll
        for (int q = 0; q < SUP_NB; q++) { const float4 vq = vb[j[q]]; x[q] = vq.x; y[q] = vq.y; z[q] = vq.z; }
#pragma unroll
        for (int q = 0; q < SUP_NB; q++) {                 // (only the value and the index of the best neighbour are tracked -- two selects per neighbour, not six:
            float s = fmaf(x[q], dl.x, fmaf(y[q], dl.y, z[q] * dl.z));      // the winner's record is read once more when the climb moves there)
            if (s > cv) { cv = s; cand = j[q]; }
        }
        e += SUP_NB;
        if (e >= eend) {                                   // this vertex's neighbours are all seen
            if (cand == cur) break;
            cur = cand;
            const float4 vn = vb[cur];
            bx = vn.x; by = vn.y; bz = vn.z; aw = __float_as_uint(vn.w);
            e = (int)(aw & 0xffffu); eend = e + (int)(aw >> 16);
            DBG_COUNT(6, 1);
        }
    }
    DBG_COUNT(7, 1);
    vout = v3(bx, by, bz); adj = aw;
    return cur;
}
DEVI int support_vertex(const Tables &T, int h, int base, V3 dl, V3 &vout, int hint = -1) {
    unsigned adj;
    return support_vertex(T, h, base, dl, vout, hint, adj);
}

// Support vertices of two hulls by all 16 lanes of the env (portal refinement asks for one on each): lane j scans vertices j,
// j + 16, ... of both LDS vertex tables, then two interleaved 16-lane arg-max reductions (ties: lowest index, as an exhaustive
// scan would give). ~n/16 LDS reads per lane and hull with no dependent chain, against ~3 dependent neighbour hops per step
// of the per-lane hill climb it replaces in collide().
template <int CTRL> DEVI int dpp_i(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, false); }
DEVI void coop_support2(const Tables &T, int baseA, int nA, V3 dA, int baseB, int nB, V3 dB, int sub, int &ia, int &ib) {
    float av = -3.0e38f, bv = -3.0e38f; int ai = 0x7fffffff, bi = 0x7fffffff;
    const float4 *vA = reinterpret_cast<const float4 *>(T.v) + baseA, *vB = reinterpret_cast<const float4 *>(T.v) + baseB;
    const int nmax = max(nA, nB);
    // the env's clone lanes 32..63 scan too: 32 lanes per env, vertices s, s + 32, ... (s = 0..31), the two halves' winners meet at the end
    const int first = sub + ((int)(threadIdx.x & 32) >> 1), stride = 2 * KL;
    for (int i = first; i < nmax; i += 2 * stride) {
        int ja[2], jb[2]; float4 a[2], b[2];
#pragma unroll
        for (int q = 0; q < 2; q++) { ja[q] = min(i + q * stride, nA - 1); jb[q] = min(i + q * stride, nB - 1); }
#pragma unroll
        for (int q = 0; q < 2; q++) { a[q] = vA[ja[q]]; b[q] = vB[jb[q]]; }
#pragma unroll
        for (int q = 0; q < 2; q++) {
            // (a lane's indices ascend, so the strict comparison alone keeps the lowest index among equal values; the index breaks ties only
            // where lanes meet, below)
            const float sa = fmaf(a[q].x, dA.x, fmaf(a[q].y, dA.y, a[q].z * dA.z));
            const bool ta = sa > av;
            av = ta ? sa : av; ai = ta ? ja[q] : ai;
            const float sb = fmaf(b[q].x, dB.x, fmaf(b[q].y, dB.y, b[q].z * dB.z));
            const bool tb = sb > bv;
            bv = tb ? sb : bv; bi = tb ? jb[q] : bi;
        }
    }
#define COOP_STEP2(R) { float oa = dpp_f<DPP_ROW_ROR(R)>(av); int oia = dpp_i<DPP_ROW_ROR(R)>(ai); float ob = dpp_f<DPP_ROW_ROR(R)>(bv); int oib = dpp_i<DPP_ROW_ROR(R)>(bi); \
        bool ta = oa > av || (oa == av && oia < ai); av = ta ? oa : av; ai = ta ? oia : ai; bool tb = ob > bv || (ob == bv && oib < bi); bv = tb ? ob : bv; bi = tb ? oib : bi; }
    COOP_STEP2(8) COOP_STEP2(4) COOP_STEP2(2) COOP_STEP2(1)
#undef COOP_STEP2
    {   float al, ah, bl, bh; int ail, aih, bil, bih;
        halves_f(av, al, ah); halves_i(ai, ail, aih); halves_f(bv, bl, bh); halves_i(bi, bil, bih);
        ai = (ah > al || (ah == al && aih < ail)) ? aih : ail;
        bi = (bh > bl || (bh == bl && bih < bil)) ? bih : bil; }
    ia = ai; ib = bi;
}

#ifndef GRIP_COLD_PORTAL      // support records also carry what is needed to rebuild them one physics.step() later (-DGRIP_COLD_PORTAL: the comparison build without the portal memory)
struct Sup { V3 v, v1, v2; int id; V3 qd; };
#define SUP_EXTRA_SET(d, s) (d).id = (s).id; (d).qd.x = (s).qd.x; (d).qd.y = (s).qd.y; (d).qd.z = (s).qd.z;
#define SUP_EXTRA_SEL(d, c, s) (d).id = (c) ? (s).id : (d).id; (d).qd.x = (c) ? (s).qd.x : (d).qd.x; (d).qd.y = (c) ? (s).qd.y : (d).qd.y; (d).qd.z = (c) ? (s).qd.z : (d).qd.z;
#else
struct Sup { V3 v, v1, v2; };
#define SUP_EXTRA_SET(d, s)
#define SUP_EXTRA_SEL(d, c, s)
#endif
// field-by-field copy: a whole-struct assignment becomes llvm.memcpy between stack slots, which SROA then leaves in scratch
// memory -- and the MPR loop pays a scratch round trip per portal update
DEVI void sup_set(Sup &d, const Sup &s) {
    d.v.x = s.v.x; d.v.y = s.v.y; d.v.z = s.v.z; d.v1.x = s.v1.x; d.v1.y = s.v1.y; d.v1.z = s.v1.z; d.v2.x = s.v2.x; d.v2.y = s.v2.y; d.v2.z = s.v2.z;
    SUP_EXTRA_SET(d, s)
}

DEVI void sup_sel(Sup &d, bool c, const Sup &s) {
    d.v.x = c ? s.v.x : d.v.x; d.v.y = c ? s.v.y : d.v.y; d.v.z = c ? s.v.z : d.v.z;
    d.v1.x = c ? s.v1.x : d.v1.x; d.v1.y = c ? s.v1.y : d.v1.y; d.v1.z = c ? s.v1.z : d.v1.z;
    d.v2.x = c ? s.v2.x : d.v2.x; d.v2.y = c ? s.v2.y : d.v2.y; d.v2.z = c ? s.v2.z : d.v2.z;
    SUP_EXTRA_SEL(d, c, s)
}
DEVI V3 portal_dir(const Sup &a, const Sup &b, const Sup &c) { return normalized(cross(b.v - a.v, c.v - a.v)); }
DEVI void expand_portal(const Sup &p0, Sup &p1, Sup &p2, Sup &p3, const Sup &p4) {
    V3 v4v0 = cross(p4.v, p0.v);
    // value selects, not "pick a destination, then store": selected pointers keep the portal in scratch memory
    const bool d1 = dot(p1.v, v4v0) > 0.f, d2 = dot(p2.v, v4v0) > 0.f, d3 = dot(p3.v, v4v0) > 0.f;
    const bool to1 = d1 ? d2 : !d3, to2 = !d1 && d3, to3 = d1 && !d2;
    sup_sel(p1, to1, p4); sup_sel(p2, to2, p4); sup_sel(p3, to3, p4);
}
DEVI bool reach_tol(const Sup &p1, const Sup &p2, const Sup &p3, const Sup &p4, V3 dir) {
    float d4 = dot(p4.v, dir);
    float mn = fminf(d4 - dot(p1.v, dir), fminf(d4 - dot(p2.v, dir), d4 - dot(p3.v, dir)));
    return mn <= MPR_TOL_F;
}
DEVI float origin_tri_dist2(V3 a, V3 b, V3 c, V3 &wit) {
    V3 ab = b - a, ac = c - a, ap = -a;
    float d1 = dot(ab, ap), d2 = dot(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) { wit = a; return dot(a, a); }
    V3 bp = -b; float d3 = dot(ab, bp), d4 = dot(ac, bp);
    if (d3 >= 0.f && d4 <= d3) { wit = b; return dot(b, b); }
    float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) { wit = a + ab * (d1 / (d1 - d3)); return dot(wit, wit); }
    V3 cp = -c; float d5 = dot(ab, cp), d6 = dot(ac, cp);
    if (d6 >= 0.f && d5 <= d6) { wit = c; return dot(c, c); }
    float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) { wit = a + ac * (d2 / (d2 - d6)); return dot(wit, wit); }
    float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) { wit = b + (c - b) * ((d4 - d3) / ((d4 - d3) + (d5 - d6))); return dot(wit, wit); }
    // face region: the closest point is the foot of the perpendicular from the origin. Formed as n (n . a) / |n|^2 its DIRECTION is the
    // facet normal exactly, whatever the depth; the barycentric form a + ab * v + ac * w gets it as a difference of vectors ~1e5 times
    // longer when two inflated hulls barely touch (depth ~1e-7 m at dist ~ margin) and fp32 then returns noise for the contact normal
    // (measured against the fp64 oracle along contact trajectories: one-step velocity errors of 0.4 m/s on exactly those states)
    V3 n = cross(ab, ac);
    float nn = dot(n, n);
    if (nn > 1e-30f) { wit = n * (dot(n, a) / nn); return dot(wit, wit); }
    float den = 1.0f / (va + vb + vc);
    wit = a + ab * (vb * den) + ac * (vc * den);
    return dot(wit, wit);
}
DEVI V3 find_pos(const Sup &p0, const Sup &p1, const Sup &p2, const Sup &p3) {
    V3 dir = portal_dir(p1, p2, p3);
    float b0 = dot(cross(p1.v, p2.v), p3.v), b1 = dot(cross(p3.v, p2.v), p0.v);
    float b2 = dot(cross(p0.v, p1.v), p3.v), b3 = dot(cross(p2.v, p1.v), p0.v);
    float sum = b0 + b1 + b2 + b3;
    if (sum <= 0.f) {
        b0 = 0.f; b1 = dot(cross(p2.v, p3.v), dir); b2 = dot(cross(p3.v, p1.v), dir); b3 = dot(cross(p1.v, p2.v), dir);
        sum = b1 + b2 + b3;
    }
    float inv = 0.5f / sum;
    return ((p0.v1 + p0.v2) * b0 + (p1.v1 + p1.v2) * b1 + (p2.v1 + p2.v2) * b2 + (p3.v1 + p3.v2) * b3) * inv;
}

// a contact owned by one lane
// what a pair lane of collide() remembers between calls: the direction that last separated its pair and the two support
// vertices that proved it (the hill climbs of the next check start there: same direction, bodies moved by one 2 ms step); and,
// for a pair in contact, the portal its refinement converged to -- three (vertex of hull 1, vertex of hull 2, query direction)
// records. The next call rebuilds that portal at the new poses and, if it still holds the origin ray, refines from there:
// 1-3 trips instead of 3 discovery + ~8 refinement trips. The converged facet is the same to the 1e-6 tolerance; the path is
// not, so results differ from a cold start at the 1e-6 m level over tens of steps (tests/test_gpu_parity.py).
#ifndef GRIP_COLD_PORTAL
struct PairMemo { V3 sep; int h1, h2; int has; };      // has: the last converged portal (vertex pairs and query directions) is in the lane's ES_PORTAL words
#else
struct PairMemo { V3 sep; int h1, h2; };
#endif

// (its constraint Jacobian rows -- normal, tangent 1, tangent 2, torsion -- are built once per step by make_constraints() and live in the
// contact's LDS slots, EF_U + 6 U_STRIDE c: 52 registers per lane less across the solver)
struct Contact {
    V3 p, n; float dist; int g1, g2, gA, gB; float fs, ft, tran, D0;
    float jar[4], jv[4];
};

// Narrow phase of one state, one item per lane and round. Items 0..5: floor vs hull geom item + 1 (support along -z,
// then graph neighbours inside the margin; mjc_PlaneConvex). Items >= 6: hull pair item - 6, Minkowski portal
// refinement (XenoCollide) on the margin-inflated hulls as a per-lane phase machine (0/1 seed the portal, 2 discover,
// 3 refine, 4 penetrate) around the ONE support evaluation. The contacts found by the 16 lanes are compacted through
// the env's LDS staging area (deterministic order: round, then lane); lane c then owns contact c.
// `sep` is this lane's memory between calls: the direction along which its hull pair was last proven apart (Minkowski
// support <= 0). Bodies move little in 2 ms, so the next call first tests that one direction (phase 6, one support pair) and
// usually is done -- the exact separating-axis argument MPR itself ends with, so no result changes; only when it fails does
// the portal search start from scratch.
DEVI int collide(const DevModel &m, const Ctx &cx, Contact &con, PairMemo &memo, Stamps &st) {
    V3 &sep = memo.sep;
    const float EPS2 = 1e-12f, EPSD = 1e-10f;
    const float infl = 0.5f * m.margin;
    const Tables &T = cx.T;
    int total = 0;
#pragma unroll 1
    for (int round = 0; round < 2; round++) {
        const int item = m.coop_items[cx.sub][round];
        const bool has = item >= 0;
        const bool plane = item < 6;
        if (round > 0) {
            // The later rounds hold what did not fit the 16 lanes: the floor test of the gripper base, which is out of reach of the floor
            // nearly always. One bounding-sphere height per lane decides whether anybody in the wave needs the round at all (its
            // prologue -- frames, table reads, portal set-up -- is ~370 instructions for all lanes).
            bool need = has && !plane;
            if (has && plane) {
                const float *tq = T.gt + (item + 1) * GT_STRIDE;
                V3 pq; M3 Rq; load_frame(cx.envl, item + 1, pq, Rq);
                const float cz = pq.z + (Rq.m[6] * tq[0] + Rq.m[7] * tq[1] + Rq.m[8] * tq[2]);
                need = cz <= tq[3] + m.margin;
            }
            if (!__any(need)) continue;
        }
        const int g1 = plane ? 0 : m.pairs[max(item - 6, 0)][0], g2 = !has ? 1 : plane ? item + 1 : m.pairs[max(item - 6, 0)][1];
        const float *t1 = T.gt + g1 * GT_STRIDE, *t2 = T.gt + g2 * GT_STRIDE;
        V3 p1 = v3(0, 0, 0), p2; M3 R1, R2;
#pragma unroll
        for (int i = 0; i < 9; i++) R1.m[i] = (i % 4 == 0) ? 1.f : 0.f;
        if (has && !plane) load_frame(cx.envl, g1, p1, R1);
        load_frame(cx.envl, g2, p2, R2);
        const int base2 = __float_as_int(t2[8]), base1 = __float_as_int(t1[8]);
        V3 c2 = p2 + mulv(R2, v3(t2[0], t2[1], t2[2]));
        Sup s0, s1, s2, s3;
        V3 dir = v3(0, 0, -1);
        int phase = -1;
        if (has && plane) {
            phase = (c2.z <= t2[3] + m.margin) ? 5 : -1;
        } else if (has) {
            V3 c1 = p1 + mulv(R1, v3(t1[0], t1[1], t1[2]));
            V3 dc = c2 - c1;
            float bound = t1[3] + t2[3] + m.margin;
            phase = dot(dc, dc) > bound * bound ? -1 : 0;
            if (phase < 0) sep = v3(0, 0, 0);
            if (phase == 0) DBG_COUNT(3, 1);
            s0.v1 = c1; s0.v2 = c2; s0.v = c1 - c2;
            if (dot(s0.v, s0.v) < EPS2) s0.v.x += 1e-5f;
            dir = normalized(-s0.v);
            if (phase == 0 && dot(sep, sep) > 0.5f) { dir = sep; phase = 6; }
#ifndef GRIP_COLD_PORTAL
            if (phase < 0) memo.has = 0;
            if (phase == 0 && memo.has) {           // rebuild the portal the last step converged to at the new poses; use it if it still holds the origin ray
                Sup w[3];
                int mpi[3]; V3 mpd[3];
                {   const float4 *pm = reinterpret_cast<const float4 *>(cx.envl + ES_PORTAL + min(local_sub(cx), 10) * PORTAL_WORDS);
                    const float4 m1 = pm[0], m2 = pm[1], m3 = pm[2];
                    mpi[0] = __float_as_int(m1.x); mpi[1] = __float_as_int(m1.y); mpi[2] = __float_as_int(m1.z);
                    mpd[0] = v3(m1.w, m2.x, m2.y); mpd[1] = v3(m2.z, m2.w, m3.x); mpd[2] = v3(m3.y, m3.z, m3.w); }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int i1 = mpi[k] & 0xffff, i2 = mpi[k] >> 16;
                    const float *a1 = T.v + 4 * (base1 + i1), *a2 = T.v + 4 * (base2 + i2);
                    w[k].v1 = p1 + mulv(R1, v3(a1[0], a1[1], a1[2])) + mpd[k] * infl;
                    w[k].v2 = p2 + mulv(R2, v3(a2[0], a2[1], a2[2])) - mpd[k] * infl;
                    w[k].v = w[k].v1 - w[k].v2; w[k].id = mpi[k]; w[k].qd = mpd[k];
                }
                V3 nn = cross(w[1].v - w[0].v, w[2].v - w[0].v);
                const float n2 = dot(nn, nn);
                bool ok = n2 > 1e-20f;
                V3 nd = nn * rsqrtf(fmaxf(n2, 1e-30f));
                const bool flip = dot(nd, s0.v) > 0.f;
                sup_set(s1, flip ? w[1] : w[0]); sup_set(s2, flip ? w[0] : w[1]); sup_set(s3, w[2]);
                if (flip) nd = -nd;
                ok = ok && dot(cross(s1.v, s3.v), s0.v) >= -EPSD && dot(cross(s3.v, s2.v), s0.v) >= -EPSD && dot(cross(s2.v, s1.v), s0.v) >= -EPSD;
                if (ok) { dir = nd; phase = dot(dir, s1.v) >= -EPSD ? 4 : 3; DBG_COUNT(13, 1); }
            }
#endif
        }
        // results of this lane's item: up to 4 contacts
        V3 rp0 = v3(0, 0, 0), rp1 = rp0, rp2 = rp0, rp3 = rp0; float rd0 = 0.f, rd1 = 0.f, rd2 = 0.f, rd3 = 0.f;
        V3 rn = v3(0, 0, 1); int rc = 0;
        int cnt = 0;
        // Two ways to evaluate supports. While any item of the env is in its opening moves (floor test, remembered direction,
        // portal seeding: a few trips, most pairs end there) every lane hill-climbs for its own item, in parallel. Once only
        // portal refinements are left (10-30 more trips for a touching pair, usually one or two pairs per env) the 16 lanes
        // serve them one at a time: the lowest-numbered refining lane (the "owner") publishes its two local directions and
        // hull ranges, everybody scans a sixteenth of the vertices, the owner takes the arg-max and advances its portal.
        const int nv2 = __float_as_int(t2[9]), nv1 = __float_as_int(t1[9]);
        if (round == 0 && cx.sub == 0) DBG_COUNT(0, 1);
#ifdef GRIP_STAMPS
        unsigned long long tw_ = stamp_now();
#endif
        STAMP(st, 20);
        while (__any(phase >= 0)) {
            const unsigned actm = group_bits(__ballot(phase >= 0), cx.lane);
            const unsigned early = group_bits(__ballot(phase >= 0 && phase != 3 && phase != 4), cx.lane);
            const bool coop = actm != 0u && early == 0u;
            if (cx.sub == 0 && actm != 0u) DBG_COUNT(coop ? 2 : 1, 1);
#ifdef GRIP_STAMPS
            const bool phase_was_early_ = phase >= 0 && !coop;
#endif
            const int owner = coop ? (__ffs((int)actm) - 1) : -1;
            const bool mine = coop ? owner == cx.sub : phase >= 0;
            V3 vl = v3(0, 0, 0), vl1 = v3(0, 0, 0); int vi2 = 0, vi1w = 0;
            if (__any(coop)) {
                // the owner's query (two local directions, hull ranges) goes to its 15 helpers by lane shuffles (ds_bpermute: no LDS
                // store, no barrier); every lane reads from the owner lane of ITS env
                const int src = 4 * ((cx.lane & 48) + max(owner, 0));
                V3 d2o = multv(R2, -dir), d1o = multv(R1, dir);
                auto bc = [&](float x) { return __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(x))); };
                auto bci = [&](int x) { return __builtin_amdgcn_ds_bpermute(src, x); };
                V3 d2 = v3(bc(d2o.x), bc(d2o.y), bc(d2o.z)), d1 = v3(bc(d1o.x), bc(d1o.y), bc(d1o.z));
                const int qb2 = bci(base2), qn2 = bci(nv2), qb1 = bci(base1), qn1 = bci(nv1);
                if (coop) {
                    int b2i, b1i;
                    coop_support2(T, qb2, qn2, d2, qb1, qn1, d1, cx.sub, b2i, b1i);
                    if (mine) {
                        const float *vp2 = T.v + 4 * (base2 + b2i), *vp1 = T.v + 4 * (base1 + b1i);
                        vl = v3(vp2[0], vp2[1], vp2[2]); vl1 = v3(vp1[0], vp1[1], vp1[2]); vi2 = b2i; vi1w = b1i;
                    }
                }
            }
            int vi1 = vi1w; unsigned adj2 = 0u;
            STAMP(st, 21);
            if (!coop && mine) {
                const bool rem = phase == 6;
                // the two hulls of a pair side by side: lanes 0..31 climb hull 2, their clones 32..63 hull 1 -- one hill-climb loop per trip
                // instead of two (floor items have one hull: the clones repeat it)
                const bool h1 = UPPER_HALF(cx) && !plane;
                const V3 dq2 = multv(R2, plane ? dir : -dir), dq1 = multv(R1, dir);
                V3 vo; unsigned adjo;
                const int vio = support_vertex(T, h1 ? g1 - 1 : g2 - 1, h1 ? base1 : base2, h1 ? dq1 : dq2, vo, rem ? (h1 ? memo.h1 : memo.h2) : -1, adjo);
                STAMP(st, 22);
                unsigned a_lo, a_hi; halves_u(adjo, a_lo, a_hi); adj2 = a_lo;
                int i_hi; halves_i(vio, vi2, i_hi);
                V3 v_hi; halves_f(vo.x, vl.x, v_hi.x); halves_f(vo.y, vl.y, v_hi.y); halves_f(vo.z, vl.z, v_hi.z);
                if (!plane) { vi1 = i_hi; vl1 = v_hi; }
            }
            STAMP(st, 23);
            if (mine) {
                Sup s;
                s.v2 = p2 + mulv(R2, vl);
                if (plane) {
                    if (s.v2.z <= m.margin) {
                        rp0 = v3(s.v2.x, s.v2.y, 0.5f * s.v2.z); rd0 = s.v2.z; rc = 1;
                        const int e0 = (int)(adj2 & 0xffffu), e1 = e0 + (int)(adj2 >> 16);       // floor items always take the per-lane path: adj2 is set
                        for (int e = e0; e < e1 && rc < 4; e++) {
                            int j = T.nbr[e];
                            const float *vp = T.v + 4 * (base2 + j);
                            V3 w = p2 + mulv(R2, v3(vp[0], vp[1], vp[2]));
                            if (w.z <= m.margin) {
                                V3 pw = v3(w.x, w.y, 0.5f * w.z);
                                if (rc == 1) { rp1 = pw; rd1 = w.z; } else if (rc == 2) { rp2 = pw; rd2 = w.z; } else { rp3 = pw; rd3 = w.z; }
                                rc++;
                            }
                        }
                    }
                    phase = -1;
                    STAMP(st, 24);
                } else {
                    s.v2 = s.v2 - dir * infl;
                    s.v1 = p1 + mulv(R1, vl1) + dir * infl;
                    s.v = s.v1 - s.v2;
#ifndef GRIP_COLD_PORTAL
                    s.id = vi1 | (vi2 << 16); s.qd = dir;
#endif
                    cnt++;
                    bool hit = false; float depth = 0.f; V3 nrm = v3(0, 0, 0), pos = v3(0, 0, 0);
                    if (phase == 6) {
                        if (dot(s.v, dir) <= 0.f) { phase = -1; memo.h1 = vi1; memo.h2 = vi2; DBG_COUNT(4, 1); }     // still apart along the remembered direction
                        else { sep = v3(0, 0, 0); dir = normalized(-s0.v); phase = 0; cnt = 0; }
                    } else if (phase == 0) {
                        sup_set(s1, s);
                        if (dot(s1.v, dir) <= 0.f) { phase = -1; sep = dir; memo.h1 = vi1; memo.h2 = vi2; }
                        else {
                            V3 d = cross(s0.v, s1.v);
                            if (dot(d, d) < EPS2 * 1e-2f) {      // origin on the ray v0 -> v1
                                hit = true; depth = norm(s1.v); nrm = depth < 1e-6f ? v3(0, 0, 0) : normalized(s1.v);
                                pos = (s1.v1 + s1.v2) * 0.5f;
                            } else { dir = normalized(d); phase = 1; }
                        }
                    } else if (phase == 1) {
                        sup_set(s2, s);
                        if (dot(s2.v, dir) <= 0.f) { phase = -1; sep = dir; memo.h1 = vi1; memo.h2 = vi2; }
                        else {
                            dir = normalized(cross(s1.v - s0.v, s2.v - s0.v));
                            if (dot(dir, s0.v) > 0.f) { Sup t; sup_set(t, s1); sup_set(s1, s2); sup_set(s2, t); dir = -dir; }
                            phase = 2; cnt = 0;
                        }
                    } else if (phase == 2) {
                        sup_set(s3, s);
                        if (dot(s3.v, dir) <= 0.f) { phase = -1; sep = dir; memo.h1 = vi1; memo.h2 = vi2; }
                        else if (cnt > 4 * MPR_MAXIT) phase = -1;
                        else if (dot(cross(s1.v, s3.v), s0.v) < -EPSD) { sup_set(s2, s3); dir = normalized(cross(s1.v - s0.v, s2.v - s0.v)); }
                        else if (dot(cross(s3.v, s2.v), s0.v) < -EPSD) { sup_set(s1, s3); dir = normalized(cross(s1.v - s0.v, s2.v - s0.v)); }
                        else {
                            dir = portal_dir(s1, s2, s3);
                            phase = dot(dir, s1.v) >= -EPSD ? 4 : 3; cnt = 0;
                        }
                    } else if (phase == 3) {
                        if (dot(s.v, dir) < 0.f) { phase = -1; sep = dir; memo.h1 = -1; memo.h2 = -1; }
                        else if (reach_tol(s1, s2, s3, s, dir) || cnt > MPR_MAXIT) phase = -1;
                        else {
                            expand_portal(s0, s1, s2, s3, s);
                            dir = portal_dir(s1, s2, s3);
                            if (dot(dir, s1.v) >= -EPSD) { phase = 4; cnt = 0; }
                        }
                    } else {   // phase 4
                        if (reach_tol(s1, s2, s3, s, dir) || cnt > MPR_MAXIT) {
                            V3 w; float d2 = origin_tri_dist2(s1.v, s2.v, s3.v, w);
                            depth = sqrtf(d2); nrm = depth < 1e-9f ? v3(0, 0, 0) : normalized(w);
                            pos = find_pos(s0, s1, s2, s3); hit = true;
#ifndef GRIP_COLD_PORTAL
                            memo.has = 1;
                            {   float4 *pm = reinterpret_cast<float4 *>(cx.envl + ES_PORTAL + min(local_sub(cx), 10) * PORTAL_WORDS);
                                pm[0] = make_float4(__int_as_float(s1.id), __int_as_float(s2.id), __int_as_float(s3.id), s1.qd.x);
                                pm[1] = make_float4(s1.qd.y, s1.qd.z, s2.qd.x, s2.qd.y); pm[2] = make_float4(s2.qd.z, s3.qd.x, s3.qd.y, s3.qd.z); }
#endif
                        } else {
                            expand_portal(s0, s1, s2, s3, s);
                            dir = portal_dir(s1, s2, s3);
                        }
                    }
                    if (hit) {
                        float dist = m.margin - depth;
                        if (dist < m.margin) {
                            if (dot(nrm, nrm) < 0.5f) nrm = normalized(s0.v2 - s0.v1);
                            rp0 = pos; rd0 = dist; rn = nrm; rc = 1; DBG_COUNT(5, 1);
                        }
                        phase = -1;
                    }
                }
            }
            STAMP(st, 25);
#ifdef GRIP_STAMPS
            {   const bool any_early = __any(phase_was_early_);
                unsigned long long tn_ = stamp_now();
                if (threadIdx.x == 0) { DBG_COUNT(any_early ? 8 : 10, tn_ - tw_); DBG_COUNT(any_early ? 9 : 11, 1); }
                tw_ = tn_; }
#endif
        }
        STAMP(st, 26);
        // ---- compaction: exclusive prefix of rc over the env's 16 lanes (rc <= 4: three ballots)
        unsigned b0 = group_bits(__ballot(rc & 1), cx.lane), b1 = group_bits(__ballot(rc & 2), cx.lane), b2 = group_bits(__ballot(rc & 4), cx.lane);
        unsigned below = (1u << cx.sub) - 1u;
        int off = total + __popc(b0 & below) + 2 * __popc(b1 & below) + 4 * __popc(b2 & below);
        total += __popc(b0) + 2 * __popc(b1) + 4 * __popc(b2);
        const float fs = fmaxf(t1[4], t2[4]), ft = fmaxf(t1[5], t2[5]), tran = t1[6] + t2[6];
        const int meta = g1 | (g2 << 8) | (__float_as_int(t1[7]) << 16) | (__float_as_int(t2[7]) << 24);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (q < rc) {
                int slot = off + q;
                if (slot < G_MAXC) {
                    float *st = cx.envl + EF_STAGE + slot * ST_STRIDE;
                    V3 pp = q == 0 ? rp0 : q == 1 ? rp1 : q == 2 ? rp2 : rp3;
                    float dd = q == 0 ? rd0 : q == 1 ? rd1 : q == 2 ? rd2 : rd3;
                    st[0] = pp.x; st[1] = pp.y; st[2] = pp.z; st[3] = rn.x; st[4] = rn.y; st[5] = rn.z;
                    st[6] = dd; st[7] = __int_as_float(meta); st[8] = fs; st[9] = ft; st[10] = tran;
                }
            }
        }
    }
    if (total > G_MAXC) { env_fault_or(cx, 2); total = G_MAXC; }
#ifndef GRIP_COLD_PORTAL
    if (total > PORTAL_SAFE_NCON) memo.has = 0;          // the solver is about to use the slots the portals live in: forget them (next step starts cold)
#endif
    wave_sync();
    // lane c takes contact c
    {   const float *st = cx.envl + EF_STAGE + min(cx.sub, G_MAXC - 1) * ST_STRIDE;
        con.p = v3(st[0], st[1], st[2]); con.n = v3(st[3], st[4], st[5]); con.dist = st[6];
        int meta = __float_as_int(st[7]);
        con.g1 = meta & 255; con.g2 = (meta >> 8) & 255; con.gA = (meta >> 16) & 255; con.gB = (meta >> 24) & 255;
        con.fs = st[8]; con.ft = st[9]; con.tran = st[10]; con.D0 = 0.f;
    }
    if (cx.sub >= total) { con.p = v3(0, 0, 0); con.n = v3(0, 0, 1); con.dist = 1.f; con.g1 = con.g2 = 0; con.gA = con.gB = GRP_WORLD; con.fs = con.ft = 1.f; con.tran = 1.f; }
    wave_sync();
    return total;
}

// actuator.py:134-184: does this env's contact list hold object-left-finger / object-right-finger contacts?
DEVI int check_grasp(const Ctx &cx, const Contact &con, int ncon) {
    bool mine = cx.sub < ncon && (con.g1 == 6 || con.g2 == 6);
    int other = con.g1 == 6 ? con.g2 : con.g1;
    unsigned t1 = group_bits(__ballot(mine && (other == 2 || other == 3)), cx.lane);
    unsigned t2 = group_bits(__ballot(mine && (other == 4 || other == 5)), cx.lane);
    return (t1 ? 1 : 0) + (t2 ? 2 : 0);
}

// ---------------------------------------------------------------- soft constraints
DEVI float impedance(const float *si, float pos, float margin) {
    float x = fabsf(pos - margin) * rcp(fmaxf(1e-15f, si[2]));
    if (x >= 1.f) return si[1];
    if (x <= 0.f) return si[0];
    float mid = si[3], power = si[4], y;
    if (power <= 1.f + 1e-6f) y = x;
    else if (power == 2.f) {            // the reference's solimp (robot xml :12: "0.9 0.95 0.001 0.5 2"): squares, not four pow() expansions
        const float u = 1.f - x;        // (a model-uniform branch: the generic path below costs ~80 instructions per pow)
        y = x <= mid ? x * x / mid : 1.f - u * u / (1.f - mid);
    }
    else if (x <= mid) y = __powf(x, power) / __powf(mid, power - 1.f);
    else y = 1.f - __powf(1.f - x, power) / __powf(1.f - mid, power - 1.f);
    return si[0] + y * (si[1] - si[0]);
}


// Elliptic condim-4 contact in jar space:  s(jar) = (D0 / 2 mu^2) dist^2(U, K), U = diag(mu, fs, fs, ft) jar,
// K = {U0 >= mu |U_t|}. Returns the cost and gradient, plus the Hessian in rank-structured form
//   H = diag(w) + ka a a^T - kb b b^T      (top zone: all zero; bottom zone: w = D, ka = kb = 0)
struct Cone { float cost, grad[4], w[4], a[4], b[4], ka, kb; };
DEVI void cone_eval(const float (&jar)[4], float D0, float impratio, float fs, float ft, Cone &c) {
    // (one exit, every field written once on each of the three paths: the early-return form zeroed all 19 fields up front and the compiler repeated those moves at every
    // nesting level of the zone tests -- ~50 v_mov per evaluation, a dozen evaluations per wave trip. Same expressions, same bits: +0.6 % physics rate, same box.
    // Splitting the record by consumer instead -- pricing: cost + gradient, line search: phi' / phi'' only, Hessian: the rank structure re-evaluated once per Newton
    // iteration -- removes the joins altogether and was measured at -2 %: the extra zone evaluations cost more than the moves they save. profiles/r05_physics_levers.txt)
    const float mu = fs * rsqrtf(impratio);
    const float D1 = D0 * impratio, D3 = D1 * ft * ft / fmaxf(1e-30f, fs * fs);
    const float S[4] = {mu, fs, fs, ft};
    float U[4];
#pragma unroll
    for (int i = 0; i < 4; i++) U[i] = S[i] * jar[i];
    const float N = U[0], T = sqrtf(U[1] * U[1] + U[2] * U[2] + U[3] * U[3]);
    const bool top = N >= mu * T || (T <= 0.f && N >= 0.f);
    const bool bottom = mu * N + T <= 0.f || (T <= 0.f && N < 0.f);
    if (top) {
#pragma unroll
        for (int i = 0; i < 4; i++) { c.grad[i] = 0.f; c.w[i] = 0.f; c.a[i] = 0.f; c.b[i] = 0.f; }
        c.cost = 0.f; c.ka = 0.f; c.kb = 0.f;
    } else if (bottom) {
        const float D[4] = {D0, D1, D1, D3};
        float cost = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) { cost += 0.5f * D[i] * jar[i] * jar[i]; c.grad[i] = D[i] * jar[i]; c.w[i] = D[i]; c.a[i] = 0.f; c.b[i] = 0.f; }
        c.cost = cost; c.ka = 0.f; c.kb = 0.f;
    } else {
        const float kap = D0 / fmaxf(1e-30f, mu * mu), s1 = rsqrtf(1.f + mu * mu);
        const float dist = (mu * T - N) * s1, invT = rcp(T);
        const float c2 = dist * mu * s1 * invT;
        c.a[0] = -s1 * S[0]; c.b[0] = 0.f; c.w[0] = 0.f;
#pragma unroll
        for (int i = 1; i < 4; i++) { const float t = U[i] * invT; c.a[i] = mu * s1 * t * S[i]; c.b[i] = t * S[i]; c.w[i] = kap * c2 * S[i] * S[i]; }
#pragma unroll
        for (int i = 0; i < 4; i++) c.grad[i] = kap * dist * c.a[i];
        c.ka = kap; c.kb = kap * c2;
        c.cost = 0.5f * kap * dist * dist;
    }
}
// jv^T H jv for the structured Hessian
DEVI float cone_quad(const Cone &c, const float (&jv)[4]) {
    float q = 0.f, da = 0.f, db = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { q = fmaf(c.w[i] * jv[i], jv[i], q); da = fmaf(c.a[i], jv[i], da); db = fmaf(c.b[i], jv[i], db); }
    return q + c.ka * da * da - c.kb * db * db;
}

// What the solver needs of this lane besides its contact: the joint limit it owns (lanes 0..6), the residuals jar = J x - aref of the two
// candidate starts (x = qacc_smooth, x = qacc_warmstart), its components of both and (M (warm - qacc_smooth))_i.
struct LaneCon { float lsgn, lD, laref; float jar_s[4], jar_w[4]; float Md_w, qsi, warmi; bool grip, constrained, coupled; };

// this lane's row of the env's mass matrix (zero in lanes 13..15)
DEVI void load_mrow(const Ctx &cx, float (&mrow)[13]) {
    const float4 *M4 = reinterpret_cast<const float4 *>(cx.envl + EF_M + min(cx.sub, 12) * M_STRIDE);
    const float4 a = M4[0], b = M4[1];
    const float e[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
    const bool g = cx.sub < 7, o = cx.sub >= 7 && cx.sub < 13;
#pragma unroll
    for (int j = 0; j < 7; j++) mrow[j] = g ? e[j] : 0.f;
#pragma unroll
    for (int j = 0; j < 6; j++) mrow[7 + j] = o ? e[j] : 0.f;
}
DEVI float row_dot(const float (&row)[13], const float (&v)[13]) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 13; j++) s = fmaf(row[j], v[j], s);
    return s;
}

// Reference accelerations and regularisation (mj_makeConstraint / mj_makeImpedance) and the solver's starting data. The joint limit of dof
// `sub` belongs to lane `sub`, contact c to lane c. A contact's four Jacobian rows are formed one at a time -- 13 registers, not 52 -- and go
// straight to its LDS slots (rows 0..3 of EF_U + 6 U_STRIDE c, where pricing, Hessian assembly and J p read them); on the way each row yields
// its velocity (-> aref) and its two start residuals. State comes from the env's LDS vectors, the kinematics from ES_KIN.
DEVI void make_constraints(const DevModel &m, const Ctx &cx, Contact &c, int ncon, LaneCon &lc) {
    float *S = cx.envl;
    const int sub = local_sub(cx);
    const bool live = sub < ncon;
    {   const int j = min(sub, 6);
        const float qj = S[ES_QPOS + j], vj = S[ES_QVEL + j];
        float r0 = 0.f, r1 = 0.f, iw = 0.f;
#pragma unroll
        for (int q = 0; q < 7; q++) { bool h = j == q; r0 = h ? m.range[q][0] : r0; r1 = h ? m.range[q][1] : r1; iw = h ? m.dof_invweight0[q] : iw; }
        float lo = qj - r0, hi = r1 - qj;
        float sgn = 0.f, dist = 0.f;
        if (lo < 0.f) { sgn = 1.f; dist = lo; } else if (hi < 0.f) { sgn = -1.f; dist = hi; }
        const bool own = sub < 7;
        lc.lsgn = own ? sgn : 0.f; lc.lD = 0.f; lc.laref = 0.f;
        if (__any(own && sgn != 0.f)) {                     // a joint at its limit is rare (0.1 % of the steps): impedance, regularisation and aref only then
            float imp = impedance(m.lim_solimp, dist, 0.f);     // (lD and laref are read only where lsgn != 0)
            float R = fmaxf(1e-15f, (1.f - imp) * iw / imp);
            lc.lD = own ? 1.0f / R : 0.f;
            lc.laref = own ? -m.b_lim * (sgn * vj) - m.k_lim * imp * dist : 0.f;
        }
    }
    const bool anylim = group_bits(__ballot(lc.lsgn != 0.f), cx.lane) != 0u;
    lc.constrained = ncon > 0 || anylim;
    // when no constraint touches the gripper (no joint limit, only floor-object contacts) its block of the problem is
    // unconstrained and decoupled: start it at qacc_smooth, which is then already optimal for those 7 dofs
    lc.grip = anylim || group_bits(__ballot(live && !(c.g1 == 0 && c.g2 == 6)), cx.lane) != 0u;
    // a gripper part against the object: the only contacts that couple the gripper's 7 x 7 block of the Hessian to the object's 6 x 6 block
    lc.coupled = group_bits(__ballot(live && c.g1 != 0 && (c.g1 == 6 || c.g2 == 6)), cx.lane) != 0u;
#pragma unroll
    for (int r = 0; r < 4; r++) { lc.jar_s[r] = 0.f; lc.jar_w[r] = 0.f; c.jar[r] = 0.f; c.jv[r] = 0.f; }
    lc.qsi = sub < 13 ? S[ES_QS + min(sub, 12)] : 0.f;
    lc.warmi = sub >= 13 ? 0.f : (sub < 7 && !lc.grip) ? lc.qsi : S[ES_WARM + min(sub, 12)];
    lc.Md_w = 0.f;
    if (!__any(lc.constrained)) return;                      // nobody in the wave solves: the rest is not needed
    float qs[13], dw[13];
    lds_ld<13>(S + ES_QS, qs);
    {   float warm[13]; lds_ld<13>(S + ES_WARM, warm);
#pragma unroll
        for (int i = 0; i < 13; i++) dw[i] = ((i < 7 && !lc.grip) ? qs[i] : warm[i]) - qs[i];
        // (M dw)_i from the lane's block row: the seven products of its own block, in the order the 13-wide dot product takes them
        const float4 *M4 = reinterpret_cast<const float4 *>(S + EF_M + min(sub, 12) * M_STRIDE);
        const float4 a = M4[0], b = M4[1];
        const bool g = sub < 7;
        float sacc = 0.f;
        sacc = fmaf(a.x, g ? dw[0] : dw[7], sacc); sacc = fmaf(a.y, g ? dw[1] : dw[8], sacc); sacc = fmaf(a.z, g ? dw[2] : dw[9], sacc);
        sacc = fmaf(a.w, g ? dw[3] : dw[10], sacc); sacc = fmaf(b.x, g ? dw[4] : dw[11], sacc); sacc = fmaf(b.y, g ? dw[5] : dw[12], sacc);
        sacc = fmaf(b.z, g ? dw[6] : 0.f, sacc);
        lc.Md_w = sub < 13 ? sacc : 0.f;
    }
    if (live) {
        float qvel[13]; lds_ld<13>(S + ES_QVEL, qvel);
        KinC k; kinc_load(S, k);
        V3 t1, t2; make_tangents(c.n, t1, t2);
        // Rows of (body B) - (body A) without a case analysis per body: a dof contributes with the signed difference of "B hangs on
        // it" and "A hangs on it" -- sG for the five ee dofs (any gripper group), sL / sR for the knuckle hinges, sO for the object's
        // six -- so every entry is one product, and a pair inside the gripper (finger against finger) gets its exact zeros from sG = 0.
        const float gB_grip = (c.gB == GRP_G || c.gB == GRP_L || c.gB == GRP_R) ? 1.f : 0.f, gA_grip = (c.gA == GRP_G || c.gA == GRP_L || c.gA == GRP_R) ? 1.f : 0.f;
        const float sG = gB_grip - gA_grip, sL = (c.gB == GRP_L ? 1.f : 0.f) - (c.gA == GRP_L ? 1.f : 0.f), sR = (c.gB == GRP_R ? 1.f : 0.f) - (c.gA == GRP_R ? 1.f : 0.f),
                    sO = (c.gB == GRP_O ? 1.f : 0.f) - (c.gA == GRP_O ? 1.f : 0.f);
        const V3 rE = c.p - k.pe, rL = c.p - k.pk[0], rR = c.p - k.pk[1], rO = c.p - k.po;
        const float imp = impedance(m.solimp, c.dist, m.margin);
        const float R0 = fmaxf(1e-15f, (1.f - imp) * c.tran / imp);
        c.D0 = 1.0f / R0;
        float4 *U = reinterpret_cast<float4 *>(S + EF_U + sub * 6 * U_STRIDE);
        // velocity of the row -> aref, residuals of the two starts, and the row itself into its LDS slot (weight word zero until hessian_vectors)
        auto finish_row = [&](int r, const float (&j)[13], float kterm) {
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 13; i++) v = fmaf(j[i], qvel[i], v);
            const float aref = -m.b_con * v - kterm;
            float vs = -aref, vw = 0.f;
#pragma unroll
            for (int i = 0; i < 13; i++) { vs = fmaf(j[i], qs[i], vs); vw = fmaf(j[i], dw[i], vw); }
            lc.jar_s[r] = vs; lc.jar_w[r] = vs + vw;
            U[4 * r] = make_float4(j[0], j[1], j[2], j[3]); U[4 * r + 1] = make_float4(j[4], j[5], j[6], 0.f);
            U[4 * r + 2] = make_float4(j[7], j[8], j[9], j[10]); U[4 * r + 3] = make_float4(j[11], j[12], 0.f, 0.f);
        };
        auto linear_row = [&](float (&j)[13], V3 e) {
            const V3 cg = cross(rE, e), al = multv(k.Ro, cross(rO, e));
            j[0] = sG * e.x; j[1] = sG * e.y; j[2] = sG * e.z; j[3] = sG * cg.x; j[4] = sG * dot(k.a4, cg);
            j[5] = sL * dot(k.ak[0], cross(rL, e)); j[6] = sR * dot(k.ak[1], cross(rR, e));
            j[7] = sO * e.x; j[8] = sO * e.y; j[9] = sO * e.z; j[10] = sO * al.x; j[11] = sO * al.y; j[12] = sO * al.z;
        };
        // The four rows in two passes of ONE instruction stream: lanes 0..31 build the normal row and the first tangent, their clones 32..63 the
        // second tangent and the torsion row (the same expressions per row as the one-row-at-a-time form below; the angular row takes e itself
        // where a linear row takes the lever arm's cross product, and has no translational part). Each half stores its rows to the contact's
        // slots; the start residuals change hands with v_permlane32_swap.
        {   const bool up = UPPER_HALF(cx);
            auto any_row = [&](float (&j)[13], V3 e, bool ang) {
                const V3 xE = cross(rE, e), xO = cross(rO, e), xL = cross(rL, e), xR = cross(rR, e);
                const V3 cg = ang ? e : xE, al = multv(k.Ro, ang ? e : xO), bl = ang ? e : xL, br = ang ? e : xR;
                j[0] = ang ? 0.f : sG * e.x; j[1] = ang ? 0.f : sG * e.y; j[2] = ang ? 0.f : sG * e.z; j[3] = sG * cg.x; j[4] = sG * dot(k.a4, cg);
                j[5] = sL * dot(k.ak[0], bl); j[6] = sR * dot(k.ak[1], br);
                j[7] = ang ? 0.f : sO * e.x; j[8] = ang ? 0.f : sO * e.y; j[9] = ang ? 0.f : sO * e.z; j[10] = sO * al.x; j[11] = sO * al.y; j[12] = sO * al.z;
            };
            auto finish2 = [&](int r, const float (&j)[13], float kterm, float &vs_out, float &vw_out) {
                float v = 0.f;
#pragma unroll
                for (int i = 0; i < 13; i++) v = fmaf(j[i], qvel[i], v);
                const float aref = -m.b_con * v - kterm;
                float vs = -aref, vw = 0.f;
#pragma unroll
                for (int i = 0; i < 13; i++) { vs = fmaf(j[i], qs[i], vs); vw = fmaf(j[i], dw[i], vw); }
                vs_out = vs; vw_out = vs + vw;
                U[4 * r] = make_float4(j[0], j[1], j[2], j[3]); U[4 * r + 1] = make_float4(j[4], j[5], j[6], 0.f);
                U[4 * r + 2] = make_float4(j[7], j[8], j[9], j[10]); U[4 * r + 3] = make_float4(j[11], j[12], 0.f, 0.f);
            };
            float sA, wA, sB, wB;
            {   float j[13]; any_row(j, up ? t2 : c.n, false); finish2(up ? 2 : 0, j, up ? 0.f : m.k_con * imp * (c.dist - m.margin), sA, wA); }
            {   float j[13]; any_row(j, up ? c.n : t1, up); finish2(up ? 3 : 1, j, 0.f, sB, wB); }
            halves_f(sA, lc.jar_s[0], lc.jar_s[2]); halves_f(wA, lc.jar_w[0], lc.jar_w[2]);
            halves_f(sB, lc.jar_s[1], lc.jar_s[3]); halves_f(wB, lc.jar_w[1], lc.jar_w[3]);
        }
    }
}

// Pricing of the constraints at the current point, row-distributed. Contact lane c evaluates its cone at jar (kept up to
// date incrementally: jar += alpha jv, as mj_solNewton does) and publishes its force f = -grad (4 floats) to LDS; dof lane
// i then reads the force of every contact and column i of its Jacobian rows (already in LDS for the Hessian) and gets
// (J^T f)_i directly -- a reduce-scatter through LDS instead of 13 sixteen-lane all-reduces. Lane j < 7 adds its joint limit.
#define EF_FORCE EF_STAGE               // [G_MAXC][4] contact forces; the staging area is free once collide() is done
#define EF_P (EF_STAGE + 64)            // [16] search direction, one component per dof lane

// `cn` comes in EVALUATED: the point being priced is where the line search stopped, and its last evaluation (line_eval) computed exactly
// this cone -- the same jar = fma(alpha, jv, jar) -- so it is not computed again (all zero in lanes without a contact, as the start pricing left it).
// `wlim` (wave-uniform): a joint of one of the wave's envs is at its limit (0.1 % of the steps); otherwise every limit term below is an exact zero and is skipped.
DEVI float price_constraints(const DevModel &m, const Ctx &cx, float lsgn, float lD, float laref, float xi, int ncon,
                             const Contact &c, bool live, const Cone &cn, float &jtfi, float &hdiag, bool wlim) {
    if (live)
        *reinterpret_cast<float4 *>(cx.envl + EF_FORCE + 4 * local_sub(cx)) = make_float4(-cn.grad[0], -cn.grad[1], -cn.grad[2], -cn.grad[3]);
    float cost = cn.cost, jt = 0.f;
    hdiag = 0.f;
    if (wlim) {                         // joint limit owned by this lane (lanes 0..6): J = sgn at dof `sub`
        float ljar = lsgn * xi - laref;
        bool lact = lsgn != 0.f && ljar < 0.f;
        jt = lact ? -lD * ljar * lsgn : 0.f;
        cost += lact ? 0.5f * lD * ljar * ljar : 0.f;
        hdiag = lact ? lD : 0.f;
    }
    wave_sync();
    const int isub = UPOS(min(local_sub(cx), 12)) + local_const(EF_U), fo_ = local_const(EF_FORCE);
    for (int k = 0; k < ncon; k++) {
        const float4 f = *reinterpret_cast<const float4 *>(cx.envl + fo_ + 4 * k);
        const float *u = cx.envl + k * 6 * U_STRIDE + isub;
        jt = fmaf(u[0], f.x, jt); jt = fmaf(u[U_STRIDE], f.y, jt); jt = fmaf(u[2 * U_STRIDE], f.z, jt); jt = fmaf(u[3 * U_STRIDE], f.w, jt);
    }
    jtfi = cx.sub < 13 ? jt : 0.f;
    return cost;
}

// The contact's six weighted Hessian vectors  J^T s'' J = sum_r w_r J_r J_r^T + ka ua ua^T - kb ub ub^T  go to the env's
// LDS slots (rows J_r themselves are already there: written once per step by make_constraints), from where every lane
// assembles its own row of H (assemble_rows). The rows are re-read from the lane's own slots, four columns at a time.
// `rank1`: somebody who is still iterating in this wave needs the two rank-one slots (a contact in the middle zone of its cone, or a coupled
// env whose full rows read every slot); in the top and bottom zones ka = kb = 0 and the slots' vectors are never read (assemble_rows_block).
DEVI void hessian_vectors(const Ctx &cx, bool live, const Cone &cn, bool rank1) {
    if (!live) return;
    float *Uf = cx.envl + EF_U + local_sub(cx) * 6 * U_STRIDE;
    float4 *U = reinterpret_cast<float4 *>(Uf);
#pragma unroll
    for (int r = 0; r < 4; r++) { Uf[r * U_STRIDE + 7] = cn.w[r]; Uf[r * U_STRIDE + 15] = cn.w[r]; }
    if (!rank1) return;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const float4 j0 = U[q], j1 = U[4 + q], j2 = U[8 + q], j3 = U[12 + q];
        float4 ua, ub;
        ua.x = cn.a[0] * j0.x + cn.a[1] * j1.x + cn.a[2] * j2.x + cn.a[3] * j3.x; ub.x = cn.b[1] * j1.x + cn.b[2] * j2.x + cn.b[3] * j3.x;
        ua.y = cn.a[0] * j0.y + cn.a[1] * j1.y + cn.a[2] * j2.y + cn.a[3] * j3.y; ub.y = cn.b[1] * j1.y + cn.b[2] * j2.y + cn.b[3] * j3.y;
        if (q != 3) {
            ua.z = cn.a[0] * j0.z + cn.a[1] * j1.z + cn.a[2] * j2.z + cn.a[3] * j3.z; ub.z = cn.b[1] * j1.z + cn.b[2] * j2.z + cn.b[3] * j3.z;
        } else { ua.z = 0.f; ub.z = 0.f; }                                                 // the object half's unused seventh column
        if (q == 0 || q == 2) {
            ua.w = cn.a[0] * j0.w + cn.a[1] * j1.w + cn.a[2] * j2.w + cn.a[3] * j3.w; ub.w = cn.b[1] * j1.w + cn.b[2] * j2.w + cn.b[3] * j3.w;
        } else { ua.w = cn.ka; ub.w = -cn.kb; }                                            // the slot's weight closes both halves
        U[16 + q] = ua; U[20 + q] = ub;
    }
}
// jv = J p for this lane's contact: rows from its LDS slots, the search direction from EF_P in the slots' column layout (zeros at the
// weight / padding positions, so the products there vanish)
DEVI void contact_jp(const Ctx &cx, bool live, float (&jv)[4]) {
    if (!live) return;
    const float4 *U = reinterpret_cast<const float4 *>(cx.envl + EF_U + local_sub(cx) * 6 * U_STRIDE);
    const float4 *P = reinterpret_cast<const float4 *>(cx.envl + EF_P);
    const float4 p0 = P[0], p1 = P[1], p2 = P[2], p3 = P[3];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float4 a = U[4 * r], b = U[4 * r + 1], c4 = U[4 * r + 2], d = U[4 * r + 3];
        float v = 0.f;
        v = fmaf(a.x, p0.x, v); v = fmaf(a.y, p0.y, v); v = fmaf(a.z, p0.z, v); v = fmaf(a.w, p0.w, v);
        v = fmaf(b.x, p1.x, v); v = fmaf(b.y, p1.y, v); v = fmaf(b.z, p1.z, v);
        v = fmaf(c4.x, p2.x, v); v = fmaf(c4.y, p2.y, v); v = fmaf(c4.z, p2.z, v); v = fmaf(c4.w, p2.w, v);
        v = fmaf(d.x, p3.x, v); v = fmaf(d.y, p3.y, v);
        jv[r] = v;
    }
}

// row `sub` of  H = M + sum over the env's contact slots of  w u u^T  (+ this lane's joint-limit term on the diagonal), all 13 columns:
// needed only when a gripper part touches the object (lc.coupled)
DEVI void assemble_rows(const Ctx &cx, int ncon, float hdiag, float (&row)[13]) {
    load_mrow(cx, row);
    const int isub = UPOS(min(cx.sub, 12));
    const float *U = cx.envl + EF_U;
    auto slot = [&](int s) {
        const float4 *u4 = reinterpret_cast<const float4 *>(U + s * U_STRIDE);
        float4 a = u4[0], b = u4[1], c4 = u4[2], d = u4[3];
        float u[13] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, c4.x, c4.y, c4.z, c4.w, d.x, d.y};
        float wi = b.w * U[s * U_STRIDE + isub];    // zero weights (inactive cone zones) simply add nothing
#pragma unroll
        for (int j = 0; j < 13; j++) row[j] = fmaf(wi, u[j], row[j]);
    };
    // the SAME order of summation as assemble_rows_block -- contacts 0, 2, 4, ... on top of the mass matrix' row in lanes 0..31, contacts
    // 1, 3, 5, ... on top of zeros in the clones, lo + hi -- so that an uncoupled env gets the same bits whether or not a wave-mate forces
    // the full rows on it (time-sliced == lock-step, mixed == single: the wave-mates differ)
    const bool up = UPPER_HALF(cx);
    if (up) {
#pragma unroll
        for (int j = 0; j < 13; j++) row[j] = 0.f;
    }
    for (int k = up ? 1 : 0; k < ncon; k += 2) {
#pragma unroll
        for (int q = 0; q < 6; q++) slot(6 * k + q);
    }
#pragma unroll
    for (int j = 0; j < 13; j++) { float lo, hi; halves_f(row[j], lo, hi); row[j] = lo + hi; }
    if (__any(hdiag != 0.f)) {
#pragma unroll
        for (int j = 0; j < 13; j++) row[j] += cx.sub == j ? hdiag : 0.f;
    }
    if (cx.sub >= 13) {                                    // lanes 13..15 carry identity rows
#pragma unroll
        for (int j = 0; j < 13; j++) row[j] = 0.f;
    }
}
// The same row when H is block diagonal (no gripper-object contact: 93 % of the wave trips of the bench workload): the seven entries of
// the lane's own block -- gripper dofs 0..6 for lanes 0..6, object dofs 7..12 (+ a zero) for the others -- from the matching half of
// every slot: two 16-byte reads and seven multiply-adds per slot instead of four and thirteen. A slot of the other block has an exact
// zero in this lane's column, so it adds nothing, as in the full row.
// `midmask` (wave-uniform): bit k = contact k of an env of this wave that is still iterating sits in the middle zone of its cone. Only
// then do the contact's two rank-one slots carry a weight; everywhere else ka = kb = 0 and the slots would add exact zeros (most contacts
// most of the time: 85 % of the floor contacts and 81 % of the hull contacts end in the bottom zone) -- they are not read.
DEVI void assemble_rows_block(const Ctx &cx, int ncon, const float (&mrow7)[7], float hdiag, unsigned midmask, float (&row)[7]) {
#pragma unroll
    for (int j = 0; j < 7; j++) row[j] = mrow7[j];
    const int lsub = local_sub(cx), isub = UPOS(min(lsub, 12)), half = lsub < 7 ? 0 : 2;
    const float *U = cx.envl + EF_U;
    auto slot = [&](int s) {
        const float4 *u4 = reinterpret_cast<const float4 *>(U + s * U_STRIDE) + half;
        const float4 a = u4[0], b = u4[1];
        const float wi = b.w * U[s * U_STRIDE + isub];
        row[0] = fmaf(wi, a.x, row[0]); row[1] = fmaf(wi, a.y, row[1]); row[2] = fmaf(wi, a.z, row[2]); row[3] = fmaf(wi, a.w, row[3]);
        row[4] = fmaf(wi, b.x, row[4]); row[5] = fmaf(wi, b.y, row[5]); row[6] = fmaf(wi, b.z, row[6]);
    };
    // contacts 0, 2, 4, ... in lanes 0..31 (on top of the mass matrix' row), contacts 1, 3, 5, ... in their clones 32..63 (on top of zeros);
    // the two partial rows meet through v_permlane32_swap: half the loop trips per lane, lo + hi in both halves (clones again)
    const bool up = UPPER_HALF(cx);
    if (up) {
#pragma unroll
        for (int j = 0; j < 7; j++) row[j] = 0.f;
    }
    for (int k = up ? 1 : 0; k < ncon; k += 2) {
        slot(6 * k); slot(6 * k + 1); slot(6 * k + 2); slot(6 * k + 3);
        if ((midmask >> k) & 1u) { slot(6 * k + 4); slot(6 * k + 5); }
    }
#pragma unroll
    for (int j = 0; j < 7; j++) { float lo, hi; halves_f(row[j], lo, hi); row[j] = lo + hi; }
    if (__any(hdiag != 0.f)) {                          // an active joint limit's term on the diagonal (rare)
        const int own = cx.sub < 7 ? cx.sub : cx.sub - 7;
#pragma unroll
        for (int j = 0; j < 7; j++) row[j] += own == j ? hdiag : 0.f;
    }
}

// phi'(alpha), phi''(alpha) of the total cost along the search direction: this lane's contact and limit, then all-reduce
// `cn` (out, contact lanes): the cone at jar + alpha jv -- the pricing of the new point reuses the last one (price_constraints)
DEVI void line_eval(const DevModel &m, float lsgn, float lD, float laref, float qi, float pi,
                    const Contact &c, bool live, float alpha, float g0, float g1, float &dphi, float &ddphi, Cone &cn, bool wlim) {
    float dp = 0.f, hp = 0.f;
    if (live) {
        float ja[4];
#pragma unroll
        for (int r = 0; r < 4; r++) ja[r] = fmaf(alpha, c.jv[r], c.jar[r]);       // (the very expression the solver advances jar with)
        cone_eval(ja, c.D0, m.impratio, c.fs, c.ft, cn);
#pragma unroll
        for (int r = 0; r < 4; r++) dp = fmaf(cn.grad[r], c.jv[r], dp);
        hp = cone_quad(cn, c.jv);
    }
    if (wlim) {
        float jv = lsgn * pi;
        float xx = lsgn * qi - laref + alpha * jv;
        bool act = lsgn != 0.f && xx < 0.f;
        dp += act ? lD * xx * jv : 0.f; hp += act ? lD * jv * jv : 0.f; }
    dphi = sum16(dp) + g0 + alpha * g1; ddphi = sum16(hp) + g1;
}

// Pricing of BOTH candidate starts of the Newton solve in one pass (qacc_smooth: x = a_s, M (x - a_s) = 0; qacc_warmstart): two cone
// evaluations per contact lane, two force sets through LDS, one exchange. Same arithmetic per point as price_constraints.
#define EF_FORCE2 (EF_STAGE + 80)       // [G_MAXC][4] forces of the second point
static_assert(EF_FORCE2 + 4 * G_MAXC <= EF_M, "second force set must fit the staging area");
DEVI void price_two_starts(const DevModel &m, const Ctx &cx, float lsgn, float lD, float laref, float xs, float xw, int ncon, const Contact &c, bool live,
                           const float (&jar_s)[4], const float (&jar_w)[4], Cone &cn_s, Cone &cn_w, float &lc_s, float &lc_w,
                           float &jt_s, float &jt_w, float &hd_s, float &hd_w, bool wlim) {
#pragma unroll
    for (int i = 0; i < 4; i++) { cn_s.grad[i] = cn_s.w[i] = cn_s.a[i] = cn_s.b[i] = 0.f; cn_w.grad[i] = cn_w.w[i] = cn_w.a[i] = cn_w.b[i] = 0.f; }
    cn_s.cost = cn_s.ka = cn_s.kb = 0.f; cn_w.cost = cn_w.ka = cn_w.kb = 0.f;
    if (live) {
        cone_eval(jar_s, c.D0, m.impratio, c.fs, c.ft, cn_s);
        cone_eval(jar_w, c.D0, m.impratio, c.fs, c.ft, cn_w);
        *reinterpret_cast<float4 *>(cx.envl + EF_FORCE + 4 * local_sub(cx)) = make_float4(-cn_s.grad[0], -cn_s.grad[1], -cn_s.grad[2], -cn_s.grad[3]);
        *reinterpret_cast<float4 *>(cx.envl + EF_FORCE2 + 4 * local_sub(cx)) = make_float4(-cn_w.grad[0], -cn_w.grad[1], -cn_w.grad[2], -cn_w.grad[3]);
    }
    lc_s = cn_s.cost; lc_w = cn_w.cost;
    jt_s = jt_w = 0.f; hd_s = hd_w = 0.f;
    if (wlim) {
        {   float lj = lsgn * xs - laref; bool la = lsgn != 0.f && lj < 0.f;
            jt_s = la ? -lD * lj * lsgn : 0.f; lc_s += la ? 0.5f * lD * lj * lj : 0.f; hd_s = la ? lD : 0.f; }
        {   float lj = lsgn * xw - laref; bool la = lsgn != 0.f && lj < 0.f;
            jt_w = la ? -lD * lj * lsgn : 0.f; lc_w += la ? 0.5f * lD * lj * lj : 0.f; hd_w = la ? lD : 0.f; }
    }
    wave_sync();
    const int isub = UPOS(min(local_sub(cx), 12)) + local_const(EF_U), fo_ = local_const(EF_FORCE);
    for (int k = 0; k < ncon; k++) {
        const float4 f = *reinterpret_cast<const float4 *>(cx.envl + fo_ + 4 * k);
        const float4 g = *reinterpret_cast<const float4 *>(cx.envl + fo_ + (EF_FORCE2 - EF_FORCE) + 4 * k);
        const float *u = cx.envl + k * 6 * U_STRIDE + isub;
        const float u0 = u[0], u1 = u[U_STRIDE], u2 = u[2 * U_STRIDE], u3 = u[3 * U_STRIDE];
        jt_s = fmaf(u0, f.x, jt_s); jt_s = fmaf(u1, f.y, jt_s); jt_s = fmaf(u2, f.z, jt_s); jt_s = fmaf(u3, f.w, jt_s);
        jt_w = fmaf(u0, g.x, jt_w); jt_w = fmaf(u1, g.y, jt_w); jt_w = fmaf(u2, g.z, jt_w); jt_w = fmaf(u3, g.w, jt_w);
    }
    if (cx.sub >= 13) { jt_s = 0.f; jt_w = 0.f; }
}
DEVI void cone_sel(Cone &d, bool take, const Cone &s) {
#pragma unroll
    for (int i = 0; i < 4; i++) { d.grad[i] = take ? s.grad[i] : d.grad[i]; d.w[i] = take ? s.w[i] : d.w[i]; d.a[i] = take ? s.a[i] : d.a[i]; d.b[i] = take ? s.b[i] : d.b[i]; }
    d.cost = take ? s.cost : d.cost; d.ka = take ? s.ka : d.ka; d.kb = take ? s.kb : d.kb;
}

// Primal Newton solve of  min 1/2 (a - a_s)^T M (a - a_s) + s(J a - aref)   (mj_solNewton's problem), cooperatively and
// row-distributed: dof lane i carries x_i, (M (x - a_s))_i and its gradient component, contact lane c carries jar_c; the search
// direction is a full vector in every lane (gathered_solve). Both candidate starts -- qacc_smooth and qacc_warmstart -- are priced
// in ONE pass (price_two_starts): qacc_smooth wins when it already satisfies the gradient tolerance (constraints inactive),
// otherwise the cheaper of the two is the start, as MuJoCo chooses it, and the solve ends right there when that start is converged.
// A Newton iteration: every lane assembles its own row of the Hessian, the system is gathered and factorised redundantly in
// registers, the exact line search all-reduces two scalars per evaluation (its first evaluation, at alpha = 0, reuses the cone the
// pricing just evaluated), then the new point is priced: one pricing per iteration, none repeated. All control flow depends only on
// all-reduced values, so the 16 lanes of an env always agree.
DEVI void solve_newton(const DevModel &m, const Ctx &cx, const LaneCon &lc, Contact &c, bool live, int ncon,
                       float &xi_out, float &jtfi_out, int &iters, Stamps &st, float *dbgH = nullptr) {
    const float lsgn = lc.lsgn, lD = lc.lD, laref = lc.laref, qsi = lc.qsi, warmi = lc.warmi, Md_w = lc.Md_w;
    const bool wlim = __any(lsgn != 0.f);               // wave-uniform: somebody's joint is at its limit (rare): every limit term is guarded by it
    const float scale = 1.0f / (m.meaninertia * 13.f);
    const float tol = fmaxf(m.tolerance, NEWTON_TOL);       // fp32 noise floor of the scaled gradient is ~1e-6
    // this lane's row of the (always block-diagonal) mass matrix, the seven entries of its own block
    float mrow[7];
    {   const int lsub = local_sub(cx);
        const float4 *M4 = reinterpret_cast<const float4 *>(cx.envl + EF_M + min(lsub, 12) * M_STRIDE);
        const float4 a = M4[0], b = M4[1];                  // (the padding words of a row are zero)
        const float e[7] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z};
#pragma unroll
        for (int j = 0; j < 7; j++) mrow[j] = cx.sub < 13 ? e[j] : 0.f;
    }
    STAMP(st, 14);
    // ---- both starts priced at once (their residuals jar = J x - aref and M (x - a_s) come from make_constraints)
    float xi, Mdi, jtfi, hdiag, cost; Cone cn; bool gconv, warm_taken;
#ifdef GRIP_CAPDUMP
    int cd_took_w = 0;
#endif
#ifdef HAVE_DBG_HIST
    bool take_w_dbg = false; float g2w_dbg = 0.f, g2s_dbg = 0.f;
#endif
    {   Cone cn_w; float lc_s, lc_w, jt_s, jt_w, hd_s, hd_w;
        price_two_starts(m, cx, lsgn, lD, laref, qsi, warmi, ncon, c, live, lc.jar_s, lc.jar_w, cn, cn_w, lc_s, lc_w, jt_s, jt_w, hd_s, hd_w, wlim);
        const float cost_s = sum16(lc_s);                                   // M (x - a_s) = 0 at x = a_s
        const float cost_w = sum16(0.5f * Md_w * (warmi - qsi) + lc_w);
        // converged when the scaled gradient is below the model's tolerance -- or below what fp32 can resolve: g = M(x - a_s) - J^T f
        // is a difference of two O(force) vectors, each carrying ~1e-6 relative rounding error
        const float gs = -jt_s, gw = Md_w - jt_w;
        const float g2s = sum16(gs * gs), t2s = sum16(jt_s * jt_s);
        const float g2w = sum16(gw * gw), t2w = sum16(Md_w * Md_w + jt_w * jt_w);
        const bool conv_s = scale * sqrtf(g2s) < tol || g2s < NEWTON_GRAD_NOISE * NEWTON_GRAD_NOISE * t2s;
        const bool conv_w = scale * sqrtf(g2w) < tol || g2w < NEWTON_GRAD_NOISE * NEWTON_GRAD_NOISE * t2w;
        const bool take_w = !conv_s && cost_w < cost_s;                     // qacc_smooth already optimal: keep it (constraints inactive)
#ifdef HAVE_DBG_HIST
        take_w_dbg = take_w; g2w_dbg = g2w; g2s_dbg = g2s;
#endif
#ifdef GRIP_CAPDUMP
        cd_took_w = take_w ? 1 : 0;
#endif
        xi = take_w ? warmi : qsi; Mdi = take_w ? Md_w : 0.f; jtfi = take_w ? jt_w : jt_s; hdiag = take_w ? hd_w : hd_s;
        cost = take_w ? cost_w : cost_s; gconv = take_w ? conv_w : conv_s; warm_taken = take_w;
        cone_sel(cn, take_w, cn_w);
#pragma unroll
        for (int r = 0; r < 4; r++) c.jar[r] = take_w ? lc.jar_w[r] : lc.jar_s[r];
    }
    STAMP(st, 6);
    bool done = gconv;
    iters = 0;
#ifdef GRIP_CAPDUMP
    float cd_hist[4 * NEWTON_MAXIT]; int cd_dump = 0;          // 1: ran into the cap, 2: restarted from qacc_smooth (and finished)
#pragma unroll
    for (int i = 0; i < 4 * NEWTON_MAXIT; i++) cd_hist[i] = 0.f;
#endif
#ifdef HAVE_DBG_HIST
    auto zone_of = [&](const float (&jar)[4]) {
        const float mu = c.fs * rsqrtf(m.impratio);
        const float N = mu * jar[0], T = sqrtf(c.fs * jar[1] * c.fs * jar[1] + c.fs * jar[2] * c.fs * jar[2] + c.ft * jar[3] * c.ft * jar[3]);
        return (N >= mu * T || (T <= 0.f && N >= 0.f)) ? 0 : (mu * N + T <= 0.f || (T <= 0.f && N < 0.f)) ? 2 : 1;
    };
    const int zone0 = zone_of(c.jar);
    if (cx.sub == 0) { const float sg = scale * sqrtf(take_w_dbg ? g2w_dbg : g2s_dbg); int b = 0; for (float t = 1e-7f; b < 7 && sg >= t; t *= 10.f) b++; DBG_HIST(64 + b, 1); DBG_HIST(72, take_w_dbg ? 1 : 0); DBG_HIST(73, 1); }
    if (live) DBG_HIST((c.g1 != 0 ? 74 : 81) + zone0, 1);
#endif
    while (__any(!done)) {
        if (!done) {
            // ---- one Newton iteration from the priced point (xi, Mdi, jtfi, cn, hdiag, cost)
            const float gi = Mdi - jtfi;                                    // gradient component of this lane
            STAMP(st, 15);
            // H couples the gripper's and the object's dofs only through a gripper-object contact: unless an env of the wave that is still
            // iterating has one, every lane assembles and factorises its own diagonal block only
            const bool full = __any(lc.coupled);
            // which contact positions hold a middle-zone contact in an env that is still iterating (wave-uniform, scalar registers)
            unsigned midmask;
            {   const unsigned long long mb = __ballot(live && cn.ka != 0.f);
                midmask = (unsigned)((mb | (mb >> 16) | (mb >> 32) | (mb >> 48)) & 0xffffull); }
            hessian_vectors(cx, live, cn, full || midmask != 0u);
            wave_sync();                                    // the contact lanes' Hessian vectors are in LDS
            STAMP(st, 16);
            float x[7];                                     // the Newton direction's components in this lane's block
            if (full) {
                float row[13];
                assemble_rows(cx, ncon, hdiag, row);
                STAMP(st, 8);
                if (dbgH && iters == 0 && cx.sub < 13) {
#pragma unroll
                    for (int j = 0; j < 13; j++) dbgH[cx.sub * 13 + j] = row[j];
                }
                // (gathered and factorised redundantly like the blocks: on a block-diagonal H its arithmetic is that of the block path plus exact
                // zeros, so an uncoupled env gets the same bits whether or not a wave-mate forces the full system -- results stay independent
                // of who shares the wave; the row-distributed Cholesky would save registers on this rare path but sums in another order)
                float p[13];
                gathered_solve<0>(cx, row, gi, p);
#pragma unroll
                for (int j = 0; j < 7; j++) x[j] = cx.sub < 7 ? p[j] : (j < 6 ? p[7 + j] : 0.f);
            } else {
                float row[7];
                assemble_rows_block(cx, ncon, mrow, hdiag, midmask, row);
                STAMP(st, 8);
                if (dbgH && iters == 0 && cx.sub < 13) {
#pragma unroll
                    for (int j = 0; j < 13; j++) {
                        const int k = j - (cx.sub < 7 ? 0 : 7);
                        float v = 0.f;
#pragma unroll
                        for (int q = 0; q < 7; q++) v = k == q ? row[q] : v;
                        dbgH[cx.sub * 13 + j] = (k >= 0 && k < (cx.sub < 7 ? 7 : 6)) ? v : 0.f;
                    }
                }
                gathered_solve_block(cx, row, gi, x);
            }
            STAMP(st, 9);
            // the direction goes to EF_P in the slots' column layout ([g0..g6, 0 | o0..o5, 0, 0]): the first lane of each block writes its half
            if (cx.sub == 0 || cx.sub == 7) {
                float4 *P = reinterpret_cast<float4 *>(cx.envl + EF_P) + (cx.sub == 0 ? 0 : 2);
                P[0] = make_float4(x[0], x[1], x[2], x[3]); P[1] = make_float4(x[4], x[5], cx.sub == 0 ? x[6] : 0.f, 0.f);
            }
            wave_sync();
            const float pi = cx.sub < 13 ? cx.envl[EF_P + UPOS(min(cx.sub, 12))] : 0.f;
            STAMP(st, 7);
            float Mpi = 0.f;
#pragma unroll
            for (int j = 0; j < 7; j++) Mpi = fmaf(mrow[j], x[j], Mpi);
            float g0 = sum16(Mpi * (xi - qsi)), g1 = sum16(Mpi * pi);
            contact_jp(cx, live, c.jv);
            STAMP(st, 17);
            // exact line search: safeguarded 1-D Newton on phi'(alpha); one evaluation site
            float lo = 0.f, hi = -1.f, alpha = 0.f, gtol = 0.f, dp0 = 0.f;
            bool lsdone = false, descent = true;
            {   // alpha = 0: the cone of the current point is the one the pricing evaluated (cn)
                float dp = 0.f, hp = 0.f;
                if (live) {
#pragma unroll
                    for (int r = 0; r < 4; r++) dp = fmaf(cn.grad[r], c.jv[r], dp);
                    hp = cone_quad(cn, c.jv);
                }
                if (wlim) {
                    float jv = lsgn * pi, xx = lsgn * xi - laref;
                    bool act = lsgn != 0.f && xx < 0.f;
                    dp += act ? lD * xx * jv : 0.f; hp += act ? lD * jv * jv : 0.f; }
                dp = sum16(dp) + g0; hp = sum16(hp) + g1;
                dp0 = dp;
                if (dp >= 0.f) { descent = false; lsdone = true; }
                gtol = LS_GTOL * fabsf(dp) + 1e-30f;
                if (!lsdone) alpha = -dp * rcp(fmaxf(hp, 1e-30f));
            }
#ifdef GRIP_CAPDUMP
            int cd_ls = 0;
#endif
            for (int ls = 1; ls <= LS_MAXIT; ls++) {
                if (!__any(!lsdone)) break;
#ifdef GRIP_CAPDUMP
                if (!lsdone) cd_ls++;
#endif
                if (!lsdone) {
                    float dp, hp;
                    line_eval(m, lsgn, lD, laref, xi, pi, c, live, alpha, g0, g1, dp, hp, cn, wlim);
                    if (fabsf(dp) < gtol) lsdone = true;
                    else {
                        if (dp < 0.f) lo = alpha; else hi = alpha;
                        if (hi > 0.f && (hi - lo) < 1e-6f * hi) lsdone = true;
                    }
                    if (!lsdone && ls < LS_MAXIT) {
                        float an = alpha - dp * rcp(fmaxf(hp, 1e-30f));
                        if (hi > 0.f && (an <= lo || an >= hi)) an = 0.5f * (lo + hi);
                        alpha = an;
                    }
                }
            }
            STAMP(st, 10);
            if (!descent) done = true;
            else {
                xi = fmaf(alpha, pi, xi); Mdi = fmaf(alpha, Mpi, Mdi);
#pragma unroll
                for (int r = 0; r < 4; r++) c.jar[r] = fmaf(alpha, c.jv[r], c.jar[r]);
                iters++;
                // ---- price the new point; stop on the gradient, on a stalled cost, or at the iteration limit
                STAMP(st, 4);
                const float lcst = price_constraints(m, cx, lsgn, lD, laref, xi, ncon, c, live, cn, jtfi, hdiag, wlim);
                const float gn = Mdi - jtfi;
                const float g2 = sum16(gn * gn), t2 = sum16(Mdi * Mdi + jtfi * jtfi);
                bool stop = scale * sqrtf(g2) < tol || g2 < NEWTON_GRAD_NOISE * NEWTON_GRAD_NOISE * t2;
                // mj_solNewton also stops when the step improved the cost by less than the tolerance. The difference of the two costs is
                // fp32 rounding noise here (costs of 1e2-1e3 against a threshold of ~1e-5: an ill-conditioned solve stopped or went on by the
                // last bit of a sum, with accelerations several m/s^2 apart). The exact line search gives the improvement without a
                // cancellation: phi' rises monotonically from phi'(0) = dp0 < 0 to 0 at alpha, so the decrease -int phi' is alpha |dp0| / 2
                // for a quadratic and within a factor of two of that across cone-zone kinks.
                if (scale * (-0.5f * alpha * dp0) < tol) stop = true;
                // ... and a solve that is still going after four iterations is allowed to stop on the measured difference too: what it gains
                // per iteration is then at the rounding level of the cost anyway (rare: ~0.3 % of the solves get this far)
                {   const float newcost = sum16(0.5f * Mdi * (xi - qsi) + lcst);
                    if (iters >= 4 && scale * fabsf(cost - newcost) < tol) stop = true;      // (a cost that ROSE is no stall: one of 512 restarted solves of round 5's probe ended that way, gradient 6.9)
                    cost = newcost; }
#ifdef HAVE_DBG_HIST
                if (cx.sub == 0 && iters <= 4) { const float sg = scale * sqrtf(g2); int b = 0; for (float t = 1e-7f; b < 7 && sg >= t; t *= 10.f) b++; DBG_HIST(32 + 8 * (iters - 1) + b, 1); }
#endif
#ifdef GRIP_CAPDUMP
#pragma unroll
                for (int q = 0; q < NEWTON_MAXIT; q++) if (q == iters - 1) { cd_hist[4 * q] = scale * sqrtf(g2); cd_hist[4 * q + 1] = alpha; cd_hist[4 * q + 2] = cost; cd_hist[4 * q + 3] = (float)cd_ls; }
                if (!stop && iters >= NEWTON_MAXIT) cd_dump = 1;
#endif
                if (!stop && iters >= NEWTON_MAXIT) { stop = true; env_fault_or(cx, 4); }
                if (!stop && iters >= NEWTON_RESTART && alpha < NEWTON_STALL_ALPHA && warm_taken) {
                    warm_taken = false;                     // once
#ifdef GRIP_CAPDUMP
                    if (cx.sub == 0 && cx.lane < 32) atomicAdd(&g_capdump_restarts, 1u);
                    cd_dump = 2;
#endif
                    // ---- cold restart (NEWTON_RESTART above): one step of length 1 along qacc_smooth - x, through the machinery of an ordinary step -- the direction
                    // goes to EF_P (every lane its own component, lanes 13..15 the padding words), jar += J (a_s - x), the cones at the new residuals, one pricing
                    const int lsub = local_sub(cx);
                    cx.envl[EF_P + (lsub < 7 ? lsub : lsub < 13 ? lsub + 1 : lsub == 13 ? 7 : lsub)] = lsub < 13 ? qsi - xi : 0.f;
                    wave_sync();
                    float jd[4] = {0.f, 0.f, 0.f, 0.f};
                    contact_jp(cx, live, jd);
#pragma unroll
                    for (int r = 0; r < 4; r++) c.jar[r] += jd[r];
                    xi = qsi; Mdi = 0.f;
                    if (live) cone_eval(c.jar, c.D0, m.impratio, c.fs, c.ft, cn);
                    wave_sync();                            // EF_P is read; the pricing below reuses the staging area
                    const float lcs = price_constraints(m, cx, lsgn, lD, laref, xi, ncon, c, live, cn, jtfi, hdiag, wlim);
                    cost = sum16(lcs);                                          // M (x - a_s) = 0 at x = a_s
                    const float g2r = sum16(jtfi * jtfi);
                    stop = scale * sqrtf(g2r) < tol;
                    iters++;
                }
                done = stop;
                STAMP(st, 6);
            }
        }
    }
#ifdef HAVE_DBG_HIST
    if (live) { const int z1 = zone_of(c.jar); DBG_HIST((c.g1 != 0 ? 77 : 84) + z1, 1); if (z1 != zone0) DBG_HIST(c.g1 != 0 ? 80 : 87, 1); }
    {   // would the cone zones of the PREVIOUS step's solution (same contact pattern) have predicted this solution's? (an active-set start: DESIGN.md section 9)
        const int z1 = live ? zone_of(c.jar) : 3, base = cx.lane - cx.sub;
        const unsigned mB = (unsigned)(__ballot(live && z1 == 2) >> base) & 0xffffu, mT = (unsigned)(__ballot(live && z1 == 0) >> base) & 0xffffu;
        const unsigned mM = (unsigned)(__ballot(live && z1 == 1) >> base) & 0xffffu, mH = (unsigned)(__ballot(live && c.g1 != 0) >> base) & 0xffffu;
        const unsigned mS = (unsigned)(__ballot(live && z1 != zone0) >> base) & 0xffffu;
        if (cx.sub == 0 && ncon > 0) {
            const int curZ = (int)(mB | (mT << 16)), curI = (int)(0x40000000u | ((unsigned)ncon << 16) | mH);
            DBG_HIST(89, 1);                                            // solves with contacts
            DBG_HIST(93, mS == 0 ? 1 : 0);                              // ... whose start zones were already the final ones
            DBG_HIST(94, mM == 0 ? 1 : 0);                              // ... without a middle-zone contact at the end
            if (MI(cx, 14) == curI) {
                DBG_HIST(90, 1);                                        // ... with the previous step's contact pattern (count, which are hull contacts)
                if (MI(cx, 13) == curZ && mM == 0) DBG_HIST(91, 1);     //     ... all in the zones the previous solution had them in, none in the middle zone
                if (MI(cx, 13) == curZ && mM == 0 && mS != 0) DBG_HIST(92, 1);   //     ... and the start classification was wrong (what the guess would have saved)
            }
            MI(cx, 13) = curZ; MI(cx, 14) = curI;
        }
    }
#endif
#ifdef GRIP_CAPDUMP
    if (cd_dump) {                          // (the restarted solves share the buffer with the capped ones: the first CAPDUMP_RECORDS of either kind are kept)
        unsigned slot = 0;
        if (cx.sub == 0 && cx.lane < 32) slot = atomicAdd(&g_capdump_n, 1u);
        slot = (unsigned)__builtin_amdgcn_ds_bpermute(4 * (cx.lane & 16), (int)slot);     // lane 0 of the env's lower row hands the record number to the row (the clones do not write)
        if (slot < CAPDUMP_RECORDS && cx.lane < 32) {
            float *r = g_capdump[slot]; const float *S = cx.envl;
            if (cx.sub == 0) { r[0] = (float)ncon; r[1] = (float)iters; r[2] = (float)((lc.coupled ? 1 : 0) | (lc.grip ? 2 : 0)); r[3] = (float)cd_took_w; r[4] = (float)cd_dump;
#pragma unroll
                for (int q = 0; q < 4 * NEWTON_MAXIT; q++) r[96 + q] = cd_hist[q]; }
            if (cx.sub < 14) r[8 + cx.sub] = S[ES_QPOS + cx.sub];
            if (cx.sub < 13) { r[24 + cx.sub] = S[ES_QVEL + cx.sub]; r[48 + cx.sub] = S[ES_WARM + cx.sub]; r[64 + cx.sub] = xi; r[80 + cx.sub] = S[ES_QS + cx.sub]; }
            if (cx.sub < 7) r[40 + cx.sub] = S[ES_CTRL + cx.sub];
        }
    }
#endif
    // the optimum and its constraint force stay distributed: lane i holds component i (the integrator exchanges them through LDS)
    STAMP(st, 18);
    xi_out = xi; jtfi_out = jtfi;
    STAMP(st, 19);
}

// ---------------------------------------------------------------- one physics.step()
// The env's state (qpos, qvel, ctrl, qacc_warmstart) lives in its LDS region; a step is
//   kinematics()  ->  [forward_dense()]  ->  collide()  ->  forward_acc()  ->  integrate()
// kinematics + collide are mj_step1's share (the state of "now": contacts as check_grasp sees them), the rest mj_step2's. The dense block
// (mass matrix, bias, qacc_smooth) does not depend on the contacts and runs BEFORE collide(), straight from the kinematics in registers, so
// that nothing of it is live across the narrow phase.

// position stage: kinematics from the LDS state. Returns qpos (the caller's control hooks want the first seven) and the full kinematics.
DEVI void forward_kin(const DevModel &m, const Ctx &cx, float (&qpos)[14], Kin &k) {
    lds_ld<14>(cx.envl + ES_QPOS, qpos);
    kinematics(m, qpos, k, cx, true);
    wave_sync();
}

// mass matrix -> EF_M, qfrc_smooth -> ES_QFS, qacc_smooth -> ES_QS (all redundantly in the 16 lanes, lane 0 stores)
DEVI void forward_dense(const DevModel &m, const Ctx &cx, const Kin &k, const float (&ctrl)[8], float xfrc_z, float *dbg_bias, Stamps &st) {
    float *S = cx.envl;
    // bias forces first, the mass matrix after them: the two together would be live next to the whole kinematics
    float qfs[13];
    {   float qvel[13]; lds_ld<13>(S + ES_QVEL, qvel);
        float bias[13];
        bias_forces(m, k, qvel, bias, cx.lane >= 32);
        if (dbg_bias) {
#pragma unroll
            for (int i = 0; i < 13; i++) dbg_bias[i] = bias[i];
        }
#pragma unroll
        for (int i = 0; i < 13; i++) qfs[i] = -m.damping[i] * qvel[i] - bias[i];
    }
#pragma unroll
    for (int u = 0; u < 7; u++) qfs[u] += m.gear[u] * fminf(fmaxf(ctrl[u], m.ctrlrange[u][0]), m.ctrlrange[u][1]);
    // xfrc_applied on body ee (force along z at its COM = frame origin): only the z slide sees it
    qfs[2] += xfrc_z;
#pragma unroll
    for (int i = 0; i < 13; i++) opaque(qfs[i]);         // (keeps the scheduler from pulling the mass matrix up into the bias block)
    float Mg[28], Mo[21];
    mass_matrix(m, k, Mg, Mo, cx.lane >= 32);
    if (cx.sub == 0) {                  // one lane publishes the env's mass matrix, every dof lane then owns a (block) row of it
        float4 *M4 = reinterpret_cast<float4 *>(S + EF_M);
#pragma unroll
        for (int i = 0; i < 7; i++) {
            M4[2 * i] = make_float4(Mg[pidx(i, 0)], Mg[pidx(i, 1)], Mg[pidx(i, 2)], Mg[pidx(i, 3)]);
            M4[2 * i + 1] = make_float4(Mg[pidx(i, 4)], Mg[pidx(i, 5)], Mg[pidx(i, 6)], 0.f);
        }
#pragma unroll
        for (int i = 0; i < 6; i++) {
            M4[2 * (7 + i)] = make_float4(Mo[pidx(i, 0)], Mo[pidx(i, 1)], Mo[pidx(i, 2)], Mo[pidx(i, 3)]);
            M4[2 * (7 + i) + 1] = make_float4(Mo[pidx(i, 4)], Mo[pidx(i, 5)], 0.f, 0.f);
        }
    }
    // qacc_smooth = M^-1 qfrc_smooth: block-diagonal (gripper 7x7, object 6x6), cheap enough to do redundantly in registers. Redundancy put to
    // use: the env's lanes 8..15 add mj_Euler's implicit damping h D to the gripper block's diagonal before the factorisation, so the one
    // instruction stream yields the factor of M (lanes 0..7: qacc_smooth) AND of M + h D (lanes 8..15), which integrate() would otherwise
    // build and factorise on its own. (x + 0 = x: lanes 0..7 factorise exactly M; a <freejoint/> object block has no damping.)
    {   const bool damped = cx.sub >= 8;
#pragma unroll
        for (int i = 0; i < 7; i++) Mg[pidx(i, i)] += damped ? m.timestep * m.damping[i] : 0.f; }
    float qs[13];
#pragma unroll
    for (int i = 0; i < 13; i++) qs[i] = qfs[i];
    block_solve(Mg, Mo, qs);
    if (cx.sub == 0) { lds_st<13>(S + ES_QFS, qfs); lds_st<13>(S + ES_QS, qs); }
    if (cx.sub == 8) lds_st<28>(S + EF_LD, Mg);
    wave_sync();
    STAMP(st, 2);
}
// once per kernel: the parts of the env's LDS region that no step rewrites
DEVI void env_lds_init(const Ctx &cx) {
    MI(cx, cx.sub) = 0;                                   // the macro step's integer words, fault bits included
    wave_sync();
}

// dynamics stage up to qacc: constraints + Newton solve. Lane i < 13 returns component i of qacc and of the constraint force J^T f.
DEVI void forward_acc(const DevModel &m, const Ctx &cx, Contact &con, int ncon, float &qacci, float &jtfi, int &iters, Stamps &st, float *dbgH = nullptr) {
    LaneCon lc;
    make_constraints(m, cx, con, ncon, lc);
    wave_sync();
    STAMP(st, 3);
    qacci = lc.qsi; jtfi = 0.f; iters = 0;
    if (__any(lc.constrained)) {
        if (lc.constrained) solve_newton(m, cx, lc, con, cx.sub < ncon, ncon, qacci, jtfi, iters, st, dbgH);
    }
#ifdef HAVE_DBG_HIST
    if (cx.sub == 0) {
        DBG_HIST(min(iters, 7), 1); DBG_HIST(8 + min((ncon + 1) / 2, 7), 1); DBG_HIST(16, 1); DBG_HIST(17, lc.constrained ? 1 : 0); DBG_HIST(18, lc.grip ? 1 : 0);
    }
    {   const bool hull = group_bits(__ballot(cx.sub < ncon && con.g1 != 0), cx.lane) != 0u, lim = group_bits(__ballot(lc.lsgn != 0.f), cx.lane) != 0u;
        if (cx.sub == 0) { DBG_HIST(19, hull ? 1 : 0); DBG_HIST(20, lim ? 1 : 0); }
        const bool coupled = group_bits(__ballot(cx.sub < ncon && con.g1 != 0 && (con.g1 == 6 || con.g2 == 6)), cx.lane) != 0u;     // a gripper part against the object
        if (cx.sub == 0) DBG_HIST(22, coupled ? 1 : 0);
        if (cx.lane == 0) { DBG_HIST(23, 1); }
        if (__any(coupled) && cx.lane == 0) DBG_HIST(88, 1);
        int mx = iters;
        for (int o = 16; o < 64; o <<= 1) mx = max(mx, __shfl_xor(mx, o));
        if (cx.lane == 0) DBG_HIST(24 + min(mx, 7), 1);
    }
#endif
    STAMP(st, 4);
}

// semi-implicit Euler with implicit joint damping: (M + h D) a' = qfrc_smooth + J^T f   (mj_Euler). The object block has
// no damping, so a' = qacc there. The dof lanes hand their components over through LDS (new warm start = qacc, and qfrc_smooth + J^T f),
// every lane integrates the whole state redundantly, lane 0 stores it. qnew: the gripper's new joint positions (control hooks).
DEVI void integrate(const DevModel &m, const Ctx &cx, float qacci, float jtfi, float (&qnew)[7], Stamps &st) {
    float *S = cx.envl;
    const float h = m.timestep;
    {   const int lsub = local_sub(cx);
        if (lsub < 13) { S[ES_WARM + lsub] = qacci; S[ES_ACC + lsub] = S[ES_QFS + lsub] + jtfi; } }
    wave_sync();
    float acc[13], qacc[13];
    lds_ld<13>(S + ES_ACC, acc); lds_ld<13>(S + ES_WARM, qacc);
    {   // gripper block: the factor of M + h D parked by forward_dense(). Object block: a <freejoint/> takes no joint defaults, so it has no
        // damping (the loader refuses a model that has) and M a = qfrc_smooth + J^T f is what qacc already solves: a' = qacc there.
        float Lg[28]; lds_ld<28>(S + EF_LD, Lg);
        float xg[7];
#pragma unroll
        for (int i = 0; i < 7; i++) xg[i] = acc[i];
        chol_solve_packed<7>(Lg, xg);
#pragma unroll
        for (int i = 0; i < 7; i++) acc[i] = xg[i];
#pragma unroll
        for (int i = 7; i < 13; i++) acc[i] = qacc[i];
    }
    float qvel[13], qpos[14];
    lds_ld<13>(S + ES_QVEL, qvel); lds_ld<14>(S + ES_QPOS, qpos);
#pragma unroll
    for (int i = 0; i < 13; i++) qvel[i] = fmaf(h, acc[i], qvel[i]);
#pragma unroll
    for (int i = 0; i < 10; i++) qpos[i] = fmaf(h, qvel[i], qpos[i]);
    V3 w = v3(qvel[10], qvel[11], qvel[12]);
    float ang = norm(w) * h;
    if (ang > 0.f) {
        V3 ax = normalized(w);
        float sh, ch; sincos_joint(0.5f * ang, sh, ch);
        float bw = ch, bx = ax.x * sh, by = ax.y * sh, bz = ax.z * sh;
        float aw = qpos[10], axx = qpos[11], ay = qpos[12], az = qpos[13];
        float rw = aw * bw - axx * bx - ay * by - az * bz;
        float rx = aw * bx + axx * bw + ay * bz - az * by;
        float ry = aw * by - axx * bz + ay * bw + az * bx;
        float rz = aw * bz + axx * by - ay * bx + az * bw;
        float inv = rsqrtf(rw * rw + rx * rx + ry * ry + rz * rz);
        qpos[10] = rw * inv; qpos[11] = rx * inv; qpos[12] = ry * inv; qpos[13] = rz * inv;
    }
    // diverged? One sum of magnitudes instead of 27 compares: it is NaN or huge as soon as any component is (a runaway state grows by orders of
    // magnitude per step, so "some component beyond 1e6" and "the sum beyond 1e6" fire on the same step)
    float mag = 0.f;
#pragma unroll
    for (int i = 0; i < 14; i++) mag += fabsf(qpos[i]);
#pragma unroll
    for (int i = 0; i < 13; i++) mag += fabsf(qvel[i]);
    if (!(mag < 1e6f)) env_fault_or(cx, 1);
    if (cx.sub == 0) { lds_st<14>(S + ES_QPOS, qpos); lds_st<13>(S + ES_QVEL, qvel); }
#pragma unroll
    for (int i = 0; i < 7; i++) qnew[i] = qpos[i];
    wave_sync();
    STAMP(st, 5);
}

// dynamics + integration of one physics.step(); kinematics, forward_dense and collide must have run on the current state
DEVI void physics_advance(const DevModel &m, const Ctx &cx, Contact &con, int ncon, Stamps &st, float (&qnew)[7], int *newton_iters = nullptr) {
    float qacci, jtfi; int iters;
    forward_acc(m, cx, con, ncon, qacci, jtfi, iters, st);
    if (newton_iters) *newton_iters = iters;
    integrate(m, cx, qacci, jtfi, qnew, st);
}

// one whole physics.step() on the LDS state with the stored controls (k_substep, k_reset's position stage uses the first half)
DEVI void physics_step(const DevModel &m, const Ctx &cx, float xfrc_z, Contact &con, int &ncon, PairMemo &memo, Stamps &st) {
    float qpos[14]; Kin k;
    forward_kin(m, cx, qpos, k);
    STAMP(st, 0);
    float ctrl[8]; lds_ld<8>(cx.envl + ES_CTRL, ctrl);
    forward_dense(m, cx, k, ctrl, xfrc_z, nullptr, st);
    ncon = collide(m, cx, con, memo, st);
    STAMP(st, 1);
    float qn[7];
    physics_advance(m, cx, con, ncon, st, qn);
}
