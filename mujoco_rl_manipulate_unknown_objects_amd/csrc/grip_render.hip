// grip_render.hip -- observation kernel: RobotEnv.get_observation (robot_env.py:275-293).
//
// One 1024-thread workgroup per environment. What the camera `gripper_camera` (robot xml :60) sees of the floor plane and the six convex hulls,
// the materials and lights of the scene (lit_colour) give RGB (sensor.py:64-66), the distance along the optical axis
// gives depth (sensor.py:69-72), which goes through transform_depth (utils.py:11-19) with its
// two whole-image reductions done in LDS. The uint8 CHW observation (5 x 64 x 64 = 20 480 B per
// env, 98 % of the macro step's algorithmic HBM bytes) is written once, coalesced, together with
// the sensor pad scalars pad[0,0] = check_grasp, pad[0,1] = pheromone_level (robot_env.py:281-283)
// that the step kernel left in pad_grasp / pad_pher.
// Two kernels find the visible surface: k_observe (observe_body_raster, the shipped one since round 4) RASTERISES the hulls' faces -- every camera-facing
// face tests the pixels of its own screen box, hits meet in a 64-bit LDS z-buffer; k_observe_rays (observe_body, rounds 1-4, GRIP_OBSERVE_RAYS=1) CLIPS every
// pixel's ray against the planes of every hull in its way (Cyrus-Beck), a 2 x 2 pixel tile per thread. Both shade, reduce and store alike; they agree up to
// the pixels whose ray passes within rounding of a face's edge (tools/render_ab.py, tests/test_gpu_parity.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include "grip_device.h"
#include <atomic>

int grip_fail(const char *msg);                     // grip_sim.hip: records the message grip_last_error() returns, yields -1
static int launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    char buf[256]; snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    return grip_fail(buf);
}

#define RW 64
#define RH 64
#define RTHREADS 1024
#define RPIX (RW * RH)

struct Frames { float p[6][3]; float R[6][9]; float cam_o[3]; float cam_R[9]; };

__device__ static void compute_frames(const DevModel &m, const float *qpos, int n, int e, int half, Frames &f) {
    float q[14];
    for (int i = 0; i < 14; i++) q[i] = ld_word(qpos, (size_t)i * n + e, half);
    V3 pe = v3(m.ee_pos0[0] + q[0], m.ee_pos0[1] + q[1], m.ee_pos0[2] + q[2]);
    float sr = sinf(q[3]), cr = cosf(q[3]), sy = sinf(q[4]), cy = cosf(q[4]);
    M3 Re;
    Re.m[0] = cy; Re.m[1] = -sy; Re.m[2] = 0.f; Re.m[3] = cr * sy; Re.m[4] = cr * cy; Re.m[5] = -sr; Re.m[6] = sr * sy; Re.m[7] = sr * cy; Re.m[8] = cr;
    auto put = [&](int g, V3 p, const M3 &R) { f.p[g][0] = p.x; f.p[g][1] = p.y; f.p[g][2] = p.z; for (int i = 0; i < 9; i++) f.R[g][i] = R.m[i]; };
    V3 pb = pe + mulv(Re, ldv(m.base_pos)); M3 Rb = mulm(Re, ldm(m.base_R));
    put(0, pb, Rb);
    for (int s = 0; s < 2; s++) {
        V3 pk = pb + mulv(Rb, ldv(m.kn_pos[s])); M3 Rk0 = mulm(Rb, ldm(m.kn_R[s]));
        float sq = sinf(q[5 + s]), cq = cosf(q[5 + s]);
        M3 Ry; Ry.m[0] = cq; Ry.m[1] = 0; Ry.m[2] = sq; Ry.m[3] = 0; Ry.m[4] = 1; Ry.m[5] = 0; Ry.m[6] = -sq; Ry.m[7] = 0; Ry.m[8] = cq;
        M3 Rk = mulm(Rk0, Ry);
        put(1 + 2 * s, pk, Rk);
        put(2 + 2 * s, pk + mulv(Rk, ldv(m.fin_pos[s])), mulm(Rk, ldm(m.fin_R[s])));
    }
    float qn = rsqrtf(q[10] * q[10] + q[11] * q[11] + q[12] * q[12] + q[13] * q[13]);
    put(5, v3(q[7], q[8], q[9]), quat_mat(q[10] * qn, q[11] * qn, q[12] * qn, q[13] * qn));
    V3 co = pe + mulv(Re, ldv(m.cam_pos)); M3 Rc = mulm(Re, ldm(m.cam_R));
    f.cam_o[0] = co.x; f.cam_o[1] = co.y; f.cam_o[2] = co.z;
    for (int i = 0; i < 9; i++) f.cam_R[i] = Rc.m[i];
}


// The same frames, one per thread: role 0..5 = geom frame role (base, left knuckle, left finger, right knuckle, right finger, object),
// role 6 = the gripper camera. Every thread walks only its own chain ee -> base -> knuckle -> finger with the expressions of
// compute_frames (bit-identical frames), so the observation kernel's prologue holds two matrices per thread instead of the whole tree
// (compute_frames needs 98 VGPRs and, called from k_observe, halved that kernel's occupancy).
__device__ __forceinline__ void frame_role(const DevModel &m, const float *qpos, int n, int e, int half, int role, Frames &f) {
    auto Q = [&](int i) { return ld_word(qpos, (size_t)i * n + e, half); };
    V3 P; M3 R;
    if (role == 5) {
        const float q0 = Q(10), q1 = Q(11), q2 = Q(12), q3 = Q(13);
        const float qn = rsqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
        P = v3(Q(7), Q(8), Q(9)); R = quat_mat(q0 * qn, q1 * qn, q2 * qn, q3 * qn);
    } else {
        const V3 pe = v3(m.ee_pos0[0] + Q(0), m.ee_pos0[1] + Q(1), m.ee_pos0[2] + Q(2));
        const float q3 = Q(3), q4 = Q(4);
        const float sr = sinf(q3), cr = cosf(q3), sy = sinf(q4), cy = cosf(q4);
        M3 Re;
        Re.m[0] = cy; Re.m[1] = -sy; Re.m[2] = 0.f; Re.m[3] = cr * sy; Re.m[4] = cr * cy; Re.m[5] = -sr; Re.m[6] = sr * sy; Re.m[7] = sr * cy; Re.m[8] = cr;
        if (role == 6) { P = pe + mulv(Re, ldv(m.cam_pos)); R = mulm(Re, ldm(m.cam_R)); }
        else {
            P = pe + mulv(Re, ldv(m.base_pos)); R = mulm(Re, ldm(m.base_R));
            if (role >= 1) {
                const int s = role >= 3 ? 1 : 0;
                const V3 pk = P + mulv(R, ldv(m.kn_pos[s])); const M3 Rk0 = mulm(R, ldm(m.kn_R[s]));
                const float qk = Q(5 + s);
                const float sq = sinf(qk), cq = cosf(qk);
                M3 Ry; Ry.m[0] = cq; Ry.m[1] = 0; Ry.m[2] = sq; Ry.m[3] = 0; Ry.m[4] = 1; Ry.m[5] = 0; Ry.m[6] = -sq; Ry.m[7] = 0; Ry.m[8] = cq;
                P = pk; R = mulm(Rk0, Ry);
                if (role == 2 || role == 4) { const V3 pf = P + mulv(R, ldv(m.fin_pos[s])); const M3 Rf = mulm(R, ldm(m.fin_R[s])); P = pf; R = Rf; }
            }
        }
    }
    float *op = role == 6 ? f.cam_o : f.p[role], *oR = role == 6 ? f.cam_R : f.R[role];
    op[0] = P.x; op[1] = P.y; op[2] = P.z;
#pragma unroll
    for (int i = 0; i < 9; i++) oR[i] = R.m[i];
}

__device__ static uint8_t to_u8(float v) { v *= 255.f; v = fminf(fmaxf(v, 0.f), 255.f); return (uint8_t)v; }

// Colour of one surface point under the fixed-function lighting MuJoCo's renderer applies to these scenes (robot xml :29-34 materials, :50-51 lights;
// [3P-recall], unpinned like the rest of the rendering): material ambient = diffuse = the geom's rgb, specular = (s, s, s), shininess x 128, emission x rgb;
// a headlight at the camera (ambient 0.1, diffuse 0.4, specular 0.5 by default) + the scene's lights -- a directional one and a spot (cutoff 45 degrees,
// exponent 10 by default) -- without attenuation or shadows; the sum is clamped per channel by the byte conversion. Flat face normals of the collision
// hulls (the reference shades its visual meshes). reflectance is 0 for every material a geom uses, so no mirror term.
__device__ __forceinline__ void lit_colour(const DevModel &m, int geom, float b0, float b1, float b2, V3 P, V3 N, V3 co, float &c0, float &c1, float &c2) {
    const float spec = m.geom_material[geom][0], shin = 128.f * m.geom_material[geom][1];
    const V3 V = normalized(co - P);
    float diff = m.geom_material[geom][2] + m.headlight[0], sp = 0.f;
    {   const float nv = dot(N, V);                                  // headlight: L = V, so the half vector is V too
        if (nv > 0.f) { diff += m.headlight[1] * nv; sp += m.headlight[2] * __powf(nv, shin); } }
#pragma unroll
    for (int l = 0; l < 2; l++) {
        const V3 ld = normalized(ldv(m.light_dir[l]));
        V3 L = -ld; float spot = 1.f;
        if (!m.light_directional[l]) {
            L = normalized(ldv(m.light_pos[l]) - P);
            const float c = -dot(L, ld);
            spot = c > m.light_params[l][3] ? __powf(c, m.light_params[l][4]) : 0.f;
        }
        diff += m.light_params[l][2];
        const float nl = dot(N, L);
        if (nl > 0.f && spot > 0.f) {
            diff += spot * m.light_params[l][0] * nl;
            const float nh = fmaxf(dot(N, normalized(L + V)), 0.f);
            sp += spot * m.light_params[l][1] * __powf(nh, shin);
        }
    }
    c0 = b0 * diff + spec * sp; c1 = b1 * diff + spec * sp; c2 = b2 * diff + spec * sp;
}

// Per env, every hull plane n.x <= d (body frame) is first rewritten for rays leaving the camera origin in CAMERA
// coordinates dc = (x, y, -1):  n.(ol + t dl) <= d  with  dl = Rg^T Rc dc, ol = Rg^T (co - pg)  becomes
// t (A.dc) <= B,  A = (Rg^T Rc)^T n,  B = d - n.ol.
// Hulls whose bounding sphere lies behind the camera plane are dropped for the whole env (the gripper base always is).
// Every thread owns a TW x TW pixel tile (1024 threads per env) and keeps its rays' (t_in, t_out, entering plane) in registers, so that a plane is
// read from LDS once per 16 rays and costs one FMA + rcp + a few selects per ray; depth never leaves registers.
typedef float f2 __attribute__((ext_vector_type(2)));         // two rays of a tile row: v_pk_fma_f32 / v_pk_mul_f32 work on both at once
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 bc2(float x) { return (f2){x, x}; }
#define TW 2                // tile width: every thread owns a TW x TW pixel tile
#define TPX (TW * TW)
#define NBOX 6              // bounding-box planes in front of every hull's plane list

// Renders env e of one batch into row blockIdx.x of obs (and, when obs2 != NULL, into row row2[0] + blockIdx.x of obs2: the
// trainer's record buffer, saving a 20 KB-per-env copy kernel).
__device__ __forceinline__ void observe_body(const DevModel &m, const DevConfig &cfg, const float *qpos, const int *pad_grasp, const int *pad_pher,
                                             int n, int e, uint8_t *obs, uint8_t *obs2, const long long *row2) {
    __shared__ Frames fr;
    __shared__ float gsph[GN_GEOM][4];          // bounding sphere centre in camera coordinates, radius^2
    __shared__ float gmx[GN_GEOM][12];          // plane transform of a hull: Rg^T Rc (row-major), Rg^T (co - pg)
    extern __shared__ float4 spl[];             // camera-space plane table of this env (sized by the launcher: planes x 16 B)
    __shared__ int gadr[GN_GEOM], gnum[GN_GEOM], gnf[GN_GEOM], gbf[GN_GEOM];   // table start, planes, planes / box planes facing the camera (listed first)
    __shared__ int wcnt[RTHREADS / 64];
    __shared__ float red[RTHREADS];
    __shared__ int redi[RTHREADS];
    __shared__ __align__(16) uint8_t img[5 * RPIX];    // the observation is composed here and leaves in 16-byte stores, once per destination
    const int tid = threadIdx.x;
    if (tid < 7) frame_role(m, qpos, n, e, cfg.state_half, tid, fr);
    __syncthreads();
    if (tid >= 1 && tid < GN_GEOM) {                                    // thread g: hull g's bounding sphere in camera coordinates
        const int g = tid;
        V3 co = ldv(fr.cam_o); M3 Rc = ldm(fr.cam_R);
        V3 p = ldv(fr.p[g - 1]); M3 R = ldm(fr.R[g - 1]);
        V3 c = multv(Rc, p + mulv(R, ldv(m.geom_center[g])) - co);
        float r = m.geom_rbound[g];
        gsph[g][0] = c.x; gsph[g][1] = c.y; gsph[g][2] = c.z; gsph[g][3] = r * r;
        gnum[g] = c.z - r < 0.f ? m.hull_pnum[g - 1] : 0;               // some of the sphere is in front of the camera
        // the hull's plane transform, once per env: Mx = Rg^T Rc and ol = Rg^T (co - pg); the table build reads them from LDS, which keeps it
        // inside the kernel's 64 VGPRs (held in registers by every thread they were the build's 18 spilled registers: 150 MB of scratch
        // traffic per launch)
        M3 Mx = mulm(M3{{R.m[0], R.m[3], R.m[6], R.m[1], R.m[4], R.m[7], R.m[2], R.m[5], R.m[8]}}, Rc);
        V3 ol = multv(R, co - p);
#pragma unroll
        for (int i = 0; i < 9; i++) gmx[g][i] = Mx.m[i];
        gmx[g][9] = ol.x; gmx[g][10] = ol.y; gmx[g][11] = ol.z;
    }
    __syncthreads();
    if (tid == 0) {
        int adr = 0;
        for (int g = 1; g < GN_GEOM; g++) { gadr[g] = adr; if (gnum[g]) adr += NBOX + gnum[g]; }
    }
    __syncthreads();
    {
    // plane n.x <= d of a hull (body frame), rewritten for rays from the camera origin in camera coordinates dc = (x, y, -1):
    // t (A.dc) <= B with A = (Rg^T Rc)^T n, B = d - n.(Rg^T (co - pg)); one float4 per plane of the visible hulls in LDS.
    // Each hull's table starts with the six planes of its vertices' bounding box (same form): a ray that misses the box misses
    // the hull, and a wave whose 256 rays all miss it skips the hull's whole plane list. The hull's own planes follow in two runs:
    // those the camera is outside of (B < 0: the only ones a ray can enter through), in their model order, then the others (they can
    // only end a ray's stay inside) -- a stable partition by ballot prefix sums, so the entering face of a ray does not depend on timing.
    const int wv = tid >> 6, ln = tid & 63;
    for (int g = 1; g < GN_GEOM; g++) {
        const int np = gnum[g];
        if (np == 0) continue;
        const M3 Mx = ldm(gmx[g]); const V3 ol = v3(gmx[g][9], gmx[g][10], gmx[g][11]);
        const float *pl = m.hull_planes + 4 * m.hull_padr[g - 1];
        float4 *tab = spl + gadr[g];
        if (wv == 0) {                                                  // the box planes, partitioned the same way inside the first wave
            float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < NBOX) {
                const int ax = tid >> 1; const bool hi = tid & 1;
                V3 nn = v3(ax == 0 ? 1.f : 0.f, ax == 1 ? 1.f : 0.f, ax == 2 ? 1.f : 0.f);
                if (!hi) nn = -nn;
                const float d = (hi ? m.hull_aabb[g - 1][3 + ax] : -m.hull_aabb[g - 1][ax]) + 1e-6f;
                V3 A = multv(Mx, nn);
                P = make_float4(A.x, A.y, A.z, d - dot(nn, ol));
            }
            const bool front = tid < NBOX && P.w < 0.f;
            const unsigned long long bal = __ballot(front);
            const int fpos = __popcll(bal & ((1ull << ln) - 1ull));
            if (tid < NBOX) tab[front ? fpos : NBOX - 1 - (tid - fpos)] = P;
            if (tid == 0) gbf[g] = __popcll(bal);
        }
        int nfront = 0;
        for (int j0 = 0; j0 < np; j0 += RTHREADS) {
            const int j = j0 + tid;
            float4 P = make_float4(0.f, 0.f, 0.f, 0.f);
            if (j < np) {
                V3 nn = v3(pl[4 * j], pl[4 * j + 1], pl[4 * j + 2]);
                V3 A = multv(Mx, nn);
                P = make_float4(A.x, A.y, A.z, pl[4 * j + 3] - dot(nn, ol));
            }
            const bool front = j < np && P.w < 0.f;
            const unsigned long long bal = __ballot(front);
            if (ln == 0) wcnt[wv] = __popcll(bal);
            __syncthreads();
            int before = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < RTHREADS / 64; k++) { const int c = wcnt[k]; before += k < wv ? c : 0; tot += c; }
            const int fpos = nfront + before + __popcll(bal & ((1ull << ln) - 1ull));       // camera-facing planes before this one
            if (j < np) tab[NBOX + (front ? fpos : np - 1 - (j - fpos))] = P;
            nfront += tot;
            __syncthreads();
        }
        if (tid == 0) gnf[g] = nfront;
    }
    }
    __syncthreads();
    // the camera frame is read again where it is needed (ray set-up, shading) rather than held in 12 registers across the plane loops
    const Frames *frl = &fr;
    asm volatile("" : "+v"(frl));
    const int nch = cfg.full_observation ? 5 : 4;
    uint8_t *o = obs ? obs + (size_t)blockIdx.x * nch * RPIX : nullptr;        // either destination may be absent (the trainer renders straight
    uint8_t *o2 = obs2 ? obs2 + (size_t)(row2[0] + blockIdx.x) * nch * RPIX : nullptr;   // into its record rows and reads them from there)
    const float tanh_ = tanf(0.5f * m.cam_fovy * 0.017453292519943295f);
    // tile of this thread: a wave covers 8 x 8 tiles = a square block of (8 TW)^2 pixels
    constexpr int WPR = RW / (8 * TW);          // waves per row of wave blocks
    const int w = tid >> 6, l = tid & 63;
    const int tx = (l & 7) + 8 * (w % WPR), ty = (l >> 3) + 8 * (w / WPR);
    float xs[TW], ys[TW];
#pragma unroll
    for (int a = 0; a < TW; a++) {
        xs[a] = (2.0f * (TW * tx + a + 0.5f) / RW - 1.0f) * tanh_;
        ys[a] = (1.0f - 2.0f * (TW * ty + a + 0.5f) / RH) * tanh_;
    }
    const f2 xs2 = {xs[0], xs[1]};
    float best[TPX]; int hitent[TPX];                                   // hit geom << 16 | entering plane; -1 = sky
#pragma unroll
    for (int q = 0; q < TPX; q++) {
        float x = xs[q % TW], y = ys[q / TW];
        float dz = frl->cam_R[6] * x + frl->cam_R[7] * y - frl->cam_R[8];
        best[q] = m.zfar; hitent[q] = -1;
        if (dz < 0.f) { float t = -frl->cam_o[2] / dz; if (t > m.znear && t < best[q]) { best[q] = t; hitent[q] = 0; } }
    }
    for (int g = 1; g < GN_GEOM; g++) {
        const int np = gnum[g];
        if (np == 0) continue;
        // bounding sphere in camera coordinates (oc = -c), per ray; a thread skips the hull when none of its rays can hit it
        const float cx_ = gsph[g][0], cy_ = gsph[g][1], cz_ = gsph[g][2];
        const float cq = cx_ * cx_ + cy_ * cy_ + cz_ * cz_ - gsph[g][3];
        unsigned mask = 0u;
#pragma unroll
        for (int q = 0; q < TPX; q++) {
            float x = xs[q % TW], y = ys[q / TW];
            float bq = -(cx_ * x + cy_ * y - cz_), dd = x * x + y * y + 1.f;
            bool pass = !(bq * bq - dd * cq < 0.f || (bq > 0.f && cq > 0.f));
            mask |= pass ? (1u << q) : 0u;
        }
        if (mask == 0u) continue;
        // Cyrus-Beck without a division per plane. Every ray starts at the camera, so a plane's B is one number for all rays: through a
        // plane with B < 0 (camera outside) a ray enters at t = B / den (den = A.dc < 0) or, when den >= 0, never reaches the hull at all;
        // a plane with B >= 0 bounds the stay from above (t den <= B). First run: the latest entry max B_i / den_i as an arg-max over
        // fractions compared by cross-multiplication (both denominators negative), one division at its end; second run: the entry
        // point must satisfy every other plane, B_i - t_in den_i >= 0. The rays of a tile row go through the packed fp32 pipe in pairs
        // (index q = 2 * row + column); "every denominator negative" and "no plane violated" are kept as sign bits (AND / OR of the raw words).
        const float4 *sb = spl + gadr[g];
        f2 bn[TW], bd[TW]; int dneg[TPX];
#pragma unroll
        for (int r = 0; r < TW; r++) { bn[r] = bc2(0.f); bd[r] = bc2(-1.f); dneg[2 * r] = dneg[2 * r + 1] = -1; }
        {   // bounding box first: six planes, no entering-plane bookkeeping
            const int bf = gbf[g];
            for (int pi = 0; pi < bf; pi++) {
                const float4 P = sb[pi];
                const f2 cxa = fma2(bc2(P.x), xs2, bc2(-P.z));
#pragma unroll
                for (int r = 0; r < TW; r++) {
                    const f2 den = fma2(bc2(P.y), bc2(ys[r]), cxa);
                    const f2 lhs = bc2(P.w) * bd[r], rhs = bn[r] * den;
                    const bool u0 = lhs.x > rhs.x, u1 = lhs.y > rhs.y;
                    bn[r].x = u0 ? P.w : bn[r].x; bd[r].x = u0 ? den.x : bd[r].x;
                    bn[r].y = u1 ? P.w : bn[r].y; bd[r].y = u1 ? den.y : bd[r].y;
                    dneg[2 * r] &= __float_as_int(den.x); dneg[2 * r + 1] &= __float_as_int(den.y);
                }
            }
            f2 tb[TW]; int vio[TPX];
#pragma unroll
            for (int r = 0; r < TW; r++) { tb[r] = bn[r] * (f2){rcp(bd[r].x), rcp(bd[r].y)}; vio[2 * r] = vio[2 * r + 1] = 0; }
            for (int pi = bf; pi < NBOX; pi++) {
                const float4 P = sb[pi];
                const f2 cxa = fma2(bc2(P.x), xs2, bc2(-P.z));
#pragma unroll
                for (int r = 0; r < TW; r++) {
                    const f2 den = fma2(bc2(P.y), bc2(ys[r]), cxa);
                    const f2 w = fma2(-tb[r], den, bc2(P.w));
                    vio[2 * r] |= __float_as_int(w.x); vio[2 * r + 1] |= __float_as_int(w.y);
                }
            }
            // ... and a box that starts behind what the ray has already hit (floor, nearer hull) cannot give a nearer hit
#pragma unroll
            for (int q = 0; q < TPX; q++) if (dneg[q] >= 0 || vio[q] < 0 || tb[q / TW][q % TW] > best[q]) mask &= ~(1u << q);
        }
        if (mask == 0u) continue;
        const float4 *sp = sb + NBOX;
        const int nf = gnf[g];
        int ent[TPX];
#pragma unroll
        for (int r = 0; r < TW; r++) { bn[r] = bc2(0.f); bd[r] = bc2(-1.f); dneg[2 * r] = dneg[2 * r + 1] = -1; ent[2 * r] = ent[2 * r + 1] = -1; }
        for (int pi = 0; pi < nf; pi++) {
            // every 16 planes: a wave none of whose rays can still reach the hull leaves the list (tiles next to a silhouette pass the box)
            if ((pi & 15) == 0 && pi > 0) {
                bool alive = false;
#pragma unroll
                for (int q = 0; q < TPX; q++) alive |= ((mask >> q) & 1u) && dneg[q] < 0;
                if (!__any(alive)) break;
            }
            const float4 P = sp[pi];
            const f2 cxa = fma2(bc2(P.x), xs2, bc2(-P.z));
#pragma unroll
            for (int r = 0; r < TW; r++) {
                const f2 den = fma2(bc2(P.y), bc2(ys[r]), cxa);
                const f2 lhs = bc2(P.w) * bd[r], rhs = bn[r] * den;     // B_i / den_i > B_b / den_b, both denominators negative
                const bool u0 = lhs.x > rhs.x, u1 = lhs.y > rhs.y;
                bn[r].x = u0 ? P.w : bn[r].x; bd[r].x = u0 ? den.x : bd[r].x; ent[2 * r] = u0 ? pi : ent[2 * r];
                bn[r].y = u1 ? P.w : bn[r].y; bd[r].y = u1 ? den.y : bd[r].y; ent[2 * r + 1] = u1 ? pi : ent[2 * r + 1];
                dneg[2 * r] &= __float_as_int(den.x); dneg[2 * r + 1] &= __float_as_int(den.y);
            }
        }
        f2 tin[TW]; int vio[TPX];
        bool any_left = false;
#pragma unroll
        for (int r = 0; r < TW; r++) tin[r] = bn[r] * (f2){rcp(bd[r].x), rcp(bd[r].y)};
#pragma unroll
        for (int q = 0; q < TPX; q++) {
            const float t = tin[q / TW][q % TW];
            const bool ok = ((mask >> q) & 1u) && dneg[q] < 0 && ent[q] >= 0 && t > 0.f && t > m.znear && t < best[q];
            vio[q] = ok ? 0 : (int)0x80000000;
            any_left |= ok;
        }
        if (!__any(any_left)) continue;
        for (int pi = nf; pi < np; pi++) {
            if (((pi - nf) & 15) == 0 && pi > nf) {
                bool alive = false;
#pragma unroll
                for (int q = 0; q < TPX; q++) alive |= vio[q] >= 0;
                if (!__any(alive)) break;
            }
            const float4 P = sp[pi];
            const f2 cxa = fma2(bc2(P.x), xs2, bc2(-P.z));
#pragma unroll
            for (int r = 0; r < TW; r++) {
                const f2 den = fma2(bc2(P.y), bc2(ys[r]), cxa);
                const f2 w = fma2(-tin[r], den, bc2(P.w));              // B_i - t_in den_i: negative = the entry point lies outside plane i
                vio[2 * r] |= __float_as_int(w.x); vio[2 * r + 1] |= __float_as_int(w.y);
            }
        }
#pragma unroll
        for (int q = 0; q < TPX; q++) {
            const bool ok = vio[q] >= 0;
            best[q] = ok ? tin[q / TW][q % TW] : best[q]; hitent[q] = ok ? ((g << 16) | ent[q]) : hitent[q];
        }
    }
    // shading (lit_colour: the materials' and lights' fixed-function model) and the RGB bytes
    asm volatile("" : "+v"(frl));
    const V3 co = ldv(frl->cam_o); const M3 Rc = ldm(frl->cam_R);
    float lmin = 3.0e38f;
    unsigned cb[3][TPX];
#pragma unroll
    for (int q = 0; q < TPX; q++) {
        const float x = xs[q % TW], y = ys[q / TW];
        const V3 dir = mulv(Rc, v3(x, y, -1.f));
        const int hit = hitent[q] < 0 ? -1 : (hitent[q] >> 16);
        float c0, c1, c2;
        if (hit < 0) {
            V3 dn = normalized(dir); float f = 0.5f * (dn.z + 1.0f);
            c0 = m.sky_rgb[3] + f * (m.sky_rgb[0] - m.sky_rgb[3]); c1 = m.sky_rgb[4] + f * (m.sky_rgb[1] - m.sky_rgb[4]); c2 = m.sky_rgb[5] + f * (m.sky_rgb[2] - m.sky_rgb[5]);
        } else {
            float b0, b1, b2; V3 nrm = v3(0, 0, 1);
            const V3 Pw = co + dir * best[q];
            if (hit == 0) {
                int cx = (int)floorf(Pw.x * 8.0f), cy = (int)floorf(Pw.y * 8.0f);
                int off = ((cx + cy) & 1) ? 3 : 0;
                b0 = m.floor_rgb[off]; b1 = m.floor_rgb[off + 1]; b2 = m.floor_rgb[off + 2];
            } else {
                b0 = m.geom_rgba[hit][0]; b1 = m.geom_rgba[hit][1]; b2 = m.geom_rgba[hit][2];
                const float4 P = spl[gadr[hit] + NBOX + (hitent[q] & 0xffff)];        // the entering plane's normal in camera coordinates
                nrm = normalized(mulv(Rc, v3(P.x, P.y, P.z)));
            }
            lit_colour(m, hit, b0, b1, b2, Pw, nrm, co, c0, c1, c2);
        }
        cb[0][q] = to_u8(c0); cb[1][q] = to_u8(c1); cb[2][q] = to_u8(c2);
        lmin = fminf(lmin, best[q]);
    }
    // the two pixels of a tile row leave as one 16-bit store per channel (the tile's column is even: 2-byte aligned)
    auto put2 = [&](int ch, int r, unsigned lo, unsigned hi) {
        const int px = (TW * ty + r) * RW + TW * tx;
        *reinterpret_cast<uint16_t *>(img + ch * RPIX + px) = (uint16_t)(lo | (hi << 8));
    };
#pragma unroll
    for (int r = 0; r < TW; r++) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) put2(ch, r, cb[ch][2 * r], cb[ch][2 * r + 1]);
        put2(nch - 1, r, 0u, 0u);
    }
    // transform_depth (utils.py:11-19): depth -= min; depth /= 2 * mean(depth[depth <= 1]); 255 * clip(depth, 0, 1)
    red[tid] = lmin; __syncthreads();
    for (int s = RTHREADS / 2; s > 0; s >>= 1) { if (tid < s) red[tid] = fminf(red[tid], red[tid + s]); __syncthreads(); }
    float dmin = red[0]; __syncthreads();
    float lsum = 0.f; int lcnt = 0;
#pragma unroll
    for (int q = 0; q < TPX; q++) { float d = best[q] - dmin; if (d <= 1.0f) { lsum += d; lcnt++; } }
    red[tid] = lsum; redi[tid] = lcnt; __syncthreads();
    for (int s = RTHREADS / 2; s > 0; s >>= 1) { if (tid < s) { red[tid] += red[tid + s]; redi[tid] += redi[tid + s]; } __syncthreads(); }
    float scale = 2.0f * (red[0] / (float)redi[0]);
    if (cfg.full_observation) {
        unsigned db[TPX];
#pragma unroll
        for (int q = 0; q < TPX; q++) {
            float v = (best[q] - dmin) / scale; v = fminf(fmaxf(v, 0.f), 1.f);
            float p = 255.0f * v;
            db[q] = (p != p) ? 0u : (unsigned)(uint8_t)p;
        }
#pragma unroll
        for (int r = 0; r < TW; r++) put2(3, r, db[2 * r], db[2 * r + 1]);
    }
    __syncthreads();
    if (tid == 0) {
        const int pg_ = GPTR(const int, pad_grasp)[e], pp_ = GPTR(const int, pad_pher)[e];
        img[(nch - 1) * RPIX] = (uint8_t)pg_; img[(nch - 1) * RPIX + 1] = (uint8_t)pp_;
    }
    __syncthreads();
    // rows are nch x 4096 bytes from 16-byte aligned bases: 1280 (1024) 16-byte stores per destination, coalesced
    const uint4 *src = reinterpret_cast<const uint4 *>(img);
    for (int i = tid; i < nch * RPIX / 16; i += RTHREADS) {
        const uint4 v = src[i];
        if (o) reinterpret_cast<uint4 *>(o)[i] = v;
        if (o2) reinterpret_cast<uint4 *>(o2)[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The same observation by RASTERISING the hulls' faces instead of clipping every ray against every plane (round 4, second half). The ray caster above
// spends 85 % of its time in the two plane loops -- a ray that passes a hull's bounding box is tested against all of the hull's planes, twice -- although a
// convex hull shows every pixel exactly one camera-facing face. Here the loop is turned round: the model carries every face's corner loop (DevModel::hull_loops,
// model/compiler.py build_hull), the hull vertices are taken to camera coordinates once per env, and every camera-facing face (B < 0 in t (A.dc) <= B) tests
// only the pixels of its own screen bounding box:
//   a ray t dc hits the face  <=>  dc . (v_i x v_{i+1}) <= 0 for every edge of the loop (counter-clockwise seen from outside; the edge planes pass through the
//   camera, so nothing is projected and vertices behind the camera need no clipping), and then t = B / (A.dc) -- the quotient the ray caster forms for its
//   entering plane, from the same A and B: the depth of a pixel is bit-identical wherever the same face wins.
// An edge's normal is always formed from its two vertices in the order of their indices and negated for the face that runs it backwards: the two faces of an
// edge see exactly opposite values, so a ray through an edge is claimed by both or by either, never by none (no holes), and the z-buffer -- 64-bit LDS words
// (depth bits | geom | plane) under an atomic minimum -- makes the outcome independent of the order of arrival: the image is reproducible bit for bit.
// Work distribution: one thread per face sets it up (plane, loop, screen box); faces of at most RS_SMALL box pixels and four corners (the bulk: the hulls are
// tessellated meshes) are rasterised by that thread; the others leave a record (plane, edge normals) and one work item per 8 x 8 pixel tile their box touches,
// which the waves then take in turn, a lane per pixel. Tables that overflow fall back to the setting-up thread walking the box itself (correct, slow, not seen).
// Differences from the ray caster: only pixels whose ray passes within rounding of a face's edge (tools/render_ab.py counts them).
#ifndef RS_SMALL
#define RS_SMALL 16
#endif
#ifndef RS_WAVES
#define RS_WAVES 8          // waves per SIMD the rasteriser is compiled for (64 registers; 2 workgroups per CU fit the LDS)
#endif
#ifndef RS_FACES
#define RS_FACES 255         // face records (measured per env: mean 78-93, max 188 on bread_crumb); an item is record << 8 | quad in 16 bits, 0xffff = none
#endif
#ifndef RS_QUAD
#define RS_QUAD 1            // work items are (face, 4 x 4 pixel quad), four per wave pass (0: (face, 8 x 8 tile), one per pass -- most faces cover a fraction of a tile)
#endif
// edge normals (mean 269-311 per env, max 614 measured) and items (8 x 8 tiles: mean 354-386, max 910) share what is left of 80 KB -- two workgroups per CU --
// after the static part, the vertices and the face records: the launcher sizes them (rs_caps)
#define RS_LDS_BUDGET (80 * 1024 - 34 * 1024)
#define RS_DYN_BYTES(nverts, edges, items) ((size_t)(((nverts) * 3 + 3) / 4 * 4) * 4 + (size_t)RS_FACES * 32 + (size_t)(edges) * 16 + (size_t)(items) * 2)
struct __align__(16) RsFace { float4 P; int ebase, k, id, pad; };
static_assert(!RS_QUAD || RS_FACES <= 255, "a quad item holds the face record in 8 bits, and 0xffff must stay free");

__device__ __forceinline__ void rs_claim(unsigned long long *zb, int pix, float t, int id) {
    atomicMin(zb + pix, ((unsigned long long)__float_as_uint(t) << 32) | (unsigned)id);
}

__device__ __forceinline__ void observe_body_raster(const DevModel &m, const DevConfig &cfg, const float *qpos, const int *pad_grasp, const int *pad_pher,
                                                    int n, int e, uint8_t *obs, uint8_t *obs2, const long long *row2, int nverts_max, const int RS_EDGES, const int RS_ITEMS) {
    __shared__ Frames fr;
    __shared__ float gmx[GN_GEOM][12];          // plane / vertex transform of a hull: Rg^T Rc (row-major), Rg^T (co - pg)
    __shared__ int gnum[GN_GEOM];               // a hull's planes if it can be seen, else 0
    __shared__ int nfaces, nedges, nitems;
    // the z-buffer (32 KB) while the faces are drawn; afterwards the composed observation (20 KB) and the two reductions' scratch
    __shared__ __align__(16) unsigned char pool[8 * RPIX];
    extern __shared__ float4 rs_dyn[];
    unsigned long long *zb = reinterpret_cast<unsigned long long *>(pool);
    uint8_t *img = pool; float *red = reinterpret_cast<float *>(pool + 5 * RPIX); int *redi = reinterpret_cast<int *>(pool + 6 * RPIX);
    float *cv = reinterpret_cast<float *>(rs_dyn);                                       // camera-space vertices of the visible hulls, xyz
    RsFace *frec = reinterpret_cast<RsFace *>(cv + (nverts_max * 3 + 3) / 4 * 4);
    float4 *en = reinterpret_cast<float4 *>(frec + RS_FACES);                             // edge normals of the recorded faces (one 16-byte read per edge and item)
    unsigned short *items = reinterpret_cast<unsigned short *>(en + RS_EDGES);            // record << 8 | quad (RS_QUAD = 0: record << 6 | tile)
    const int tid = threadIdx.x;
    // vertices and planes are numbered over all six hulls as the model lists them (uniform offsets, scalar compares): no per-env table of the visible ones, and
    // the vertices' global loads are issued before anything else -- they do not depend on the pose, so their latency hides behind the frames
    const int nvert_all = m.hull_vadr[GN_HULL - 1] + m.hull_vnum[GN_HULL - 1], nplane_all = m.hull_padr[GN_HULL - 1] + m.hull_pnum[GN_HULL - 1];
    auto hull_of_vertex = [&](int i) { int g = 1; for (int h = 2; h < GN_GEOM; h++) g = i >= m.hull_vadr[h - 1] ? h : g; return g; };
    auto hull_of_plane = [&](int i) { int g = 1; for (int h = 2; h < GN_GEOM; h++) g = i >= m.hull_padr[h - 1] ? h : g; return g; };
    constexpr int VPT = 2;                                              // vertices per thread held in registers over the prologue (more: loaded late, below)
    float4 vearly[VPT];
    {   const float4 *vb = reinterpret_cast<const float4 *>(m.hull_blob);
#pragma unroll
        for (int k = 0; k < VPT; k++) { const int i = tid + k * RTHREADS; vearly[k] = i < nvert_all ? vb[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    if (tid < 7) frame_role(m, qpos, n, e, cfg.state_half, tid, fr);
    if (tid == 0) { nfaces = 0; nedges = 0; nitems = 0; }
    for (int i = tid; i < RPIX; i += RTHREADS) zb[i] = ~0ull;
    for (int i = tid; i < RS_ITEMS; i += RTHREADS) items[i] = 0xffffu;
    __syncthreads();
    if (tid >= 1 && tid < GN_GEOM) {                                    // thread g: is hull g in front of the camera? its transform (as in the ray caster)
        const int g = tid;
        V3 co = ldv(fr.cam_o); M3 Rc = ldm(fr.cam_R);
        V3 p = ldv(fr.p[g - 1]); M3 R = ldm(fr.R[g - 1]);
        V3 c = multv(Rc, p + mulv(R, ldv(m.geom_center[g])) - co);
        float r = m.geom_rbound[g];
        M3 Mx = mulm(M3{{R.m[0], R.m[3], R.m[6], R.m[1], R.m[4], R.m[7], R.m[2], R.m[5], R.m[8]}}, Rc);
        V3 ol = multv(R, co - p);
#pragma unroll
        for (int i = 0; i < 9; i++) gmx[g][i] = Mx.m[i];
        gmx[g][9] = ol.x; gmx[g][10] = ol.y; gmx[g][11] = ol.z;
        // visible = the corners of the hull's vertex box are not all on the far side of one of the view pyramid's planes (near plane, four sides; the sides
        // pushed out by a hundredth of the box's depth against rounding): a hull that fails this covers no pixel's ray
        bool vis = c.z - r < 0.f;
        if (vis) {
            const float th = tanf(0.5f * m.cam_fovy * 0.017453292519943295f) * 1.0001f;
            unsigned out = 0x1fu;
            for (int k = 0; k < 8; k++) {
                const V3 q = multv(Mx, v3(m.hull_aabb[g - 1][(k & 1) ? 3 : 0] - ol.x, m.hull_aabb[g - 1][(k & 2) ? 4 : 1] - ol.y, m.hull_aabb[g - 1][(k & 4) ? 5 : 2] - ol.z));
                const float d = -q.z * th + 1e-6f;                                        // half-width of the pyramid at the corner's depth
                unsigned o_ = 0u;
                o_ |= q.z > -m.znear ? 1u : 0u; o_ |= q.x > d ? 2u : 0u; o_ |= q.x < -d ? 4u : 0u; o_ |= q.y > d ? 8u : 0u; o_ |= q.y < -d ? 16u : 0u;
                out &= o_;
            }
            vis = out == 0u;
        }
        gnum[g] = vis ? m.hull_pnum[g - 1] : 0;
    }
    __syncthreads();
    const float tanh_ = tanf(0.5f * m.cam_fovy * 0.017453292519943295f);
    auto ray_x = [&](int px) { return (2.0f * (px + 0.5f) / RW - 1.0f) * tanh_; };       // the ray caster's expressions: the same floats
    auto ray_y = [&](int py) { return (1.0f - 2.0f * (py + 0.5f) / RH) * tanh_; };
    // ---- vertices of the visible hulls -> camera coordinates: Mx^T (v - ol)
    {
        const float4 *vb = reinterpret_cast<const float4 *>(m.hull_blob);
        auto to_camera = [&](int i, float4 v) {
            const int g = hull_of_vertex(i);
            const int *gn = gnum; asm volatile("" : "+v"(gn));          // (read where it is used: hoisted, the table is registers held across the loops)
            if (!gn[g]) return;
            const M3 Mx = ldm(gmx[g]);
            const V3 c = multv(Mx, v3(v.x - gmx[g][9], v.y - gmx[g][10], v.z - gmx[g][11]));
            cv[3 * i] = c.x; cv[3 * i + 1] = c.y; cv[3 * i + 2] = c.z;
        };
#pragma unroll
        for (int k = 0; k < VPT; k++) if (tid + k * RTHREADS < nvert_all) to_camera(tid + k * RTHREADS, vearly[k]);
        for (int i = tid + VPT * RTHREADS; i < nvert_all; i += RTHREADS) to_camera(i, vb[i]);
    }
    __syncthreads();
    // ---- one thread per face: set-up, and the small ones drawn on the spot
#ifndef RS_STOP
#define RS_STOP 0          // diagnostic builds: 1 = leave after the vertex transform, 2 = after the per-face pass (tools/render_ab.py times the difference)
#endif
    if (RS_STOP != 1) {
        // (from the last plane down: the object and the fingers first -- the gripper base, whose 770 planes lead the list, is out of sight and its threads leave at once)
        for (int i = tid; i < nplane_all; i += RTHREADS) {
            const int jg = nplane_all - 1 - i, g = hull_of_plane(jg), j = jg - m.hull_padr[g - 1];
            {   const int *gn = gnum; asm volatile("" : "+v"(gn)); if (!gn[g]) continue; }
            const int l0 = m.hull_ladr[jg], K = m.hull_ladr[jg + 1] - l0;
            if (K < 3) continue;                                        // a duplicate of an earlier plane of the same face
            const float *pl = m.hull_planes + 4 * jg;
            const V3 nn = v3(pl[0], pl[1], pl[2]);
            const M3 Mx = ldm(gmx[g]);
            const V3 A = multv(Mx, nn);
            const float B = pl[3] - dot(nn, v3(gmx[g][9], gmx[g][10], gmx[g][11]));
            if (!(B < 0.f)) continue;                                   // the camera is not outside this plane: no ray enters through it
            const float *cvg = cv + 3 * m.hull_vadr[g - 1];
            const int *lp = m.hull_loops + l0;
            // screen box of the part of the loop beyond the near plane z = -znear (nothing nearer is drawn: t > znear): its corners there and the points where
            // its edges cross the plane. Conservative by a hundredth of a pixel; the edge tests below decide, and they need no projection or clipping.
            int id4[4]; V3 v4[4];
            float xmin = 3.0e38f, xmax = -3.0e38f, ymin = 3.0e38f, ymax = -3.0e38f;
            const float zn = -m.znear;
            auto take = [&](float x, float y, float z) {
                const float iz = rcp(-z), px = x * iz, py = y * iz;
                xmin = fminf(xmin, px); xmax = fmaxf(xmax, px); ymin = fminf(ymin, py); ymax = fmaxf(ymax, py);
            };
            auto corner = [&](V3 prev, V3 cur) {
                const bool ic = cur.z <= zn, ip = prev.z <= zn;
                if (ic) take(cur.x, cur.y, cur.z);
                if (ic != ip) { const float s_ = (zn - prev.z) * rcp(cur.z - prev.z); take(fmaf(s_, cur.x - prev.x, prev.x), fmaf(s_, cur.y - prev.y, prev.y), zn); }
            };
            auto vert = [&](int id) { return v3(cvg[3 * id], cvg[3 * id + 1], cvg[3 * id + 2]); };
#pragma unroll
            for (int k = 0; k < 4; k++) { id4[k] = lp[k < K ? k : 0]; v4[k] = vert(id4[k]); }
            V3 prev = K == 3 ? v4[2] : (K == 4 ? v4[3] : vert(lp[K - 1]));
#pragma unroll
            for (int k = 0; k < 4; k++) if (k < K) { corner(prev, v4[k]); prev = v4[k]; }
            for (int k = 4; k < K; k++) { const V3 c = vert(lp[k]); corner(prev, c); prev = c; }
            if (!(xmin <= xmax)) continue;                              // nothing of the face lies beyond the near plane
            int x0, x1, y0, y1;
            {   // pixel px's ray is ((2 (px + 0.5) / RW - 1) tanh): px + 0.5 = (x / tanh + 1) RW / 2
                const float sx = 0.5f * RW / tanh_, sy = 0.5f * RH / tanh_;
                const float fx0 = xmin * sx + 0.5f * RW - 0.5f, fx1 = xmax * sx + 0.5f * RW - 0.5f;
                const float fy0 = 0.5f * RH - ymax * sy - 0.5f, fy1 = 0.5f * RH - ymin * sy - 0.5f;
                if (fx1 < -0.01f || fy1 < -0.01f || fx0 > RW - 0.99f || fy0 > RH - 0.99f) continue;       // off screen
                x0 = (int)ceilf(fmaxf(fx0 - 0.01f, 0.f)); x1 = (int)floorf(fminf(fx1 + 0.01f, RW - 1.f));
                y0 = (int)ceilf(fmaxf(fy0 - 0.01f, 0.f)); y1 = (int)floorf(fminf(fy1 + 0.01f, RH - 1.f));
                if (x0 > x1 || y0 > y1) continue;                       // between pixel centres
            }
            const int idw = (g << 16) | j;
            const int area = (x1 - x0 + 1) * (y1 - y0 + 1);
            // the edge (a, b) of the loop: normal of the plane through the camera, formed in index order
            auto edge = [&](int ia, V3 a, int ib, V3 b) { const bool sw = ia > ib; const V3 c = cross(sw ? b : a, sw ? a : b); return sw ? -c : c; };
            if (K <= 4 && area <= RS_SMALL) {
                V3 ne[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int kn = (k + 1 < K) ? k + 1 : 0;
                    ne[k] = k < K ? edge(id4[k], v4[k], id4[kn], v4[kn]) : v3(0.f, 0.f, 0.f);
                }
                int px = x0, py = y0;
                for (int q = 0; q < area; q++) {
                    const float x = ray_x(px), y = ray_y(py);
                    bool in = true;
#pragma unroll
                    for (int k = 0; k < 4; k++) in &= !(fmaf(ne[k].x, x, fmaf(ne[k].y, y, -ne[k].z)) > 0.f);
                    const float den = fmaf(A.y, y, fmaf(A.x, x, -A.z));
                    const float t = B * rcp(den);
                    if (in && den < 0.f && t > 0.f && t > m.znear) rs_claim(zb, py * RW + px, t, idw);
                    if (++px > x1) { px = x0; ++py; }
                }
                continue;
            }
            // a larger face: a record, its edge normals, one item per tile (quad) of its box
            constexpr int TS = RS_QUAD ? 2 : 3, TB = RS_QUAD ? 8 : 6;              // log2 of the tile's side; bits of a tile number
            const int tx0 = x0 >> TS, tx1 = x1 >> TS, ty0 = y0 >> TS, ty1 = y1 >> TS, ntile = (tx1 - tx0 + 1) * (ty1 - ty0 + 1);
            int fi = atomicAdd(&nfaces, 1), eb = -1, ib = -1;
            if (fi < RS_FACES) { eb = atomicAdd(&nedges, K); if (eb + K <= RS_EDGES) { ib = atomicAdd(&nitems, ntile); if (ib + ntile > RS_ITEMS) ib = -1; } }
            if (ib >= 0) {
                RsFace f; f.P = make_float4(A.x, A.y, A.z, B); f.ebase = eb; f.k = K; f.id = idw; f.pad = 0;
                frec[fi] = f;
                int ia = lp[0]; V3 a = v3(cvg[3 * ia], cvg[3 * ia + 1], cvg[3 * ia + 2]);
                const int i0 = ia; const V3 a0 = a;
                for (int k = 0; k < K; k++) {
                    const int ib2 = k + 1 < K ? lp[k + 1] : i0;
                    const V3 b = k + 1 < K ? v3(cvg[3 * ib2], cvg[3 * ib2 + 1], cvg[3 * ib2 + 2]) : a0;
                    const V3 c = edge(ia, a, ib2, b);
                    en[eb + k] = make_float4(c.x, c.y, c.z, 0.f);
                    ia = ib2; a = b;
                }
                int w = ib;
                for (int ty = ty0; ty <= ty1; ty++) for (int tx = tx0; tx <= tx1; tx++) items[w++] = (unsigned short)((fi << TB) | ((ty << (6 - TS)) + tx));
                continue;
            }
            // tables full: this thread walks the box itself
            for (int py = y0; py <= y1; py++) for (int px = x0; px <= x1; px++) {
                const float x = ray_x(px), y = ray_y(py);
                bool in = true;
                int ia = lp[0]; V3 a = v3(cvg[3 * ia], cvg[3 * ia + 1], cvg[3 * ia + 2]);
                const int i0 = ia; const V3 a0 = a;
                for (int k = 0; k < K; k++) {
                    const int ib2 = k + 1 < K ? lp[k + 1] : i0;
                    const V3 b = k + 1 < K ? v3(cvg[3 * ib2], cvg[3 * ib2 + 1], cvg[3 * ib2 + 2]) : a0;
                    const V3 c = edge(ia, a, ib2, b);
                    in &= !(fmaf(c.x, x, fmaf(c.y, y, -c.z)) > 0.f);
                    ia = ib2; a = b;
                }
                const float den = fmaf(A.y, y, fmaf(A.x, x, -A.z));
                const float t = B * rcp(den);
                if (in && den < 0.f && t > 0.f && t > m.znear) rs_claim(zb, py * RW + px, t, idw);
            }
        }
    }
    __syncthreads();
    // ---- the recorded faces, tile by tile: a lane per pixel; a wave takes one 8 x 8 tile or four 4 x 4 quads (of any faces) per pass
    if (RS_STOP != 1 && RS_STOP != 2) {
        const int ni = min(nitems, RS_ITEMS), wv = tid >> 6, ln = tid & 63;
#if RS_QUAD
        // (the next pass's item and face record are requested while this pass works, and the edge normals come four at a time: one LDS round trip per
        // dependent step -- item -> record -> edge -> edge ... -- was most of this loop's time)
        auto fetch = [&](int it0, unsigned &item, RsFace &f) {
            const int it = it0 + (ln >> 4);
            item = it < ni ? (unsigned)items[it] : 0xffffu;                // (0xffff: past the end, or the hole a refused allocation leaves)
            f = frec[item != 0xffffu ? (item >> 8) : 0];
        };
        unsigned item, item_n = 0xffffu; RsFace f, f_n;
        fetch(4 * wv, item, f);
        for (int it0 = 4 * wv; it0 < ni; it0 += 4 * (RTHREADS / 64)) {
            if (it0 + 4 * (RTHREADS / 64) < ni) fetch(it0 + 4 * (RTHREADS / 64), item_n, f_n);
            const bool live = item != 0xffffu;
            const int K = live ? f.k : 0;
            const int quad = item & 255, px = (quad & 15) * 4 + (ln & 3), py = (quad >> 4) * 4 + ((ln >> 2) & 3);
            const float x = ray_x(px), y = ray_y(py);
            bool in = live;
            const float4 *ep = en + f.ebase;
            for (int k0 = 0; __any(k0 < K); k0 += 4) {
                float4 c[4];
#pragma unroll
                for (int j = 0; j < 4; j++) c[j] = ep[max(min(k0 + j, K - 1), 0)];       // (past the loop's end: its last edge again)
#pragma unroll
                for (int j = 0; j < 4; j++) in &= !(fmaf(c[j].x, x, fmaf(c[j].y, y, -c[j].z)) > 0.f);
            }
            const float den = fmaf(f.P.y, y, fmaf(f.P.x, x, -f.P.z));
            const float t = f.P.w * rcp(den);
            if (in && den < 0.f && t > 0.f && t > m.znear) rs_claim(zb, py * RW + px, t, f.id);
            item = item_n; f = f_n; item_n = 0xffffu;
        }
#else
        for (int it = wv; it < ni; it += RTHREADS / 64) {
            const unsigned item = __builtin_amdgcn_readfirstlane((unsigned)items[it]);
            if (item == 0xffffu) continue;                              // (the hole a refused allocation leaves)
            RsFace f = frec[item >> 6];
            f.k = __builtin_amdgcn_readfirstlane(f.k); f.ebase = __builtin_amdgcn_readfirstlane(f.ebase);
            const int tile = item & 63, px = (tile & 7) * 8 + (ln & 7), py = (tile >> 3) * 8 + (ln >> 3);
            const float x = ray_x(px), y = ray_y(py);
            bool in = true;
            const float4 *ep = en + f.ebase;
            for (int k = 0; k < f.k; k++) { const float4 c = ep[k]; in &= !(fmaf(c.x, x, fmaf(c.y, y, -c.z)) > 0.f); }
            const float den = fmaf(f.P.y, y, fmaf(f.P.x, x, -f.P.z));
            const float t = f.P.w * rcp(den);
            if (in && den < 0.f && t > 0.f && t > m.znear) rs_claim(zb, py * RW + px, t, f.id);
        }
#endif
    }
    __syncthreads();
    // ---- resolve: the thread's 2 x 2 tile against the floor, as the ray caster orders its hits
    const Frames *frl = &fr;
    asm volatile("" : "+v"(frl));
    const int nch = cfg.full_observation ? 5 : 4;
    uint8_t *o = obs ? obs + (size_t)blockIdx.x * nch * RPIX : nullptr;
    uint8_t *o2 = obs2 ? obs2 + (size_t)(row2[0] + blockIdx.x) * nch * RPIX : nullptr;
    constexpr int WPR = RW / (8 * TW);
    const int w = tid >> 6, l = tid & 63;
    const int tx = (l & 7) + 8 * (w % WPR), ty = (l >> 3) + 8 * (w / WPR);
    float xs[TW], ys[TW];
#pragma unroll
    for (int a = 0; a < TW; a++) { xs[a] = ray_x(TW * tx + a); ys[a] = ray_y(TW * ty + a); }
    float best[TPX]; int hitent[TPX];
#pragma unroll
    for (int q = 0; q < TPX; q++) {
        float x = xs[q % TW], y = ys[q / TW];
        float dz = frl->cam_R[6] * x + frl->cam_R[7] * y - frl->cam_R[8];
        best[q] = m.zfar; hitent[q] = -1;
        if (dz < 0.f) { float t = -frl->cam_o[2] / dz; if (t > m.znear && t < best[q]) { best[q] = t; hitent[q] = 0; } }
        const unsigned long long z = zb[(TW * ty + q / TW) * RW + TW * tx + q % TW];
        const float th = __uint_as_float((unsigned)(z >> 32));
        if (z != ~0ull && th < best[q]) { best[q] = th; hitent[q] = (int)(unsigned)z; }
    }
    __syncthreads();                                                    // the z-buffer's space becomes the image and the reductions' scratch
    // shading (lit_colour) and the RGB bytes
    asm volatile("" : "+v"(frl));
    const V3 co = ldv(frl->cam_o); const M3 Rc = ldm(frl->cam_R);
    float lmin = 3.0e38f;
    // (a pixel's three bytes go to the image as soon as they exist and the four pixels are kept apart -- sched_barrier -- so that one pixel's shading is live at a time:
    // held in cb[3][4] and interleaved, the tail was most of the kernel's spilled registers)
#pragma unroll
    for (int q = 0; q < TPX; q++) {
        const float x = xs[q % TW], y = ys[q / TW];
        const V3 dir = mulv(Rc, v3(x, y, -1.f));
        const int hit = hitent[q] < 0 ? -1 : (hitent[q] >> 16);
        float c0, c1, c2;
        if (hit < 0) {
            V3 dn = normalized(dir); float f = 0.5f * (dn.z + 1.0f);
            c0 = m.sky_rgb[3] + f * (m.sky_rgb[0] - m.sky_rgb[3]); c1 = m.sky_rgb[4] + f * (m.sky_rgb[1] - m.sky_rgb[4]); c2 = m.sky_rgb[5] + f * (m.sky_rgb[2] - m.sky_rgb[5]);
        } else {
            float b0, b1, b2; V3 nrm = v3(0, 0, 1);
            const V3 Pw = co + dir * best[q];
            if (hit == 0) {
                int cx = (int)floorf(Pw.x * 8.0f), cy = (int)floorf(Pw.y * 8.0f);
                int off = ((cx + cy) & 1) ? 3 : 0;
                b0 = m.floor_rgb[off]; b1 = m.floor_rgb[off + 1]; b2 = m.floor_rgb[off + 2];
            } else {
                b0 = m.geom_rgba[hit][0]; b1 = m.geom_rgba[hit][1]; b2 = m.geom_rgba[hit][2];
                const float *pl = m.hull_planes + 4 * (m.hull_padr[hit - 1] + (hitent[q] & 0xffff));     // the winning face's normal in camera coordinates, as the set-up formed it
                const V3 A = multv(ldm(gmx[hit]), v3(pl[0], pl[1], pl[2]));
                nrm = normalized(mulv(Rc, A));
            }
#if defined(RS_SKIP) && (RS_SKIP & 1)          // diagnostic builds (-DRS_STOP=1 -DRS_SKIP=1|2): no lighting / no depth reductions -- lit_colour is 0.030 of the 0.063 ms per 1024 rows that are not faces
            c0 = b0 * nrm.z; c1 = b1 * Pw.x; c2 = b2;
#else
            lit_colour(m, hit, b0, b1, b2, Pw, nrm, co, c0, c1, c2);
#endif
        }
        {   const int px = (TW * ty + q / TW) * RW + TW * tx + q % TW;
            img[px] = to_u8(c0); img[RPIX + px] = to_u8(c1); img[2 * RPIX + px] = to_u8(c2); }
        lmin = fminf(lmin, best[q]);
        __builtin_amdgcn_sched_barrier(0);
    }
    auto put2 = [&](int ch, int r, unsigned lo, unsigned hi) {
        const int px = (TW * ty + r) * RW + TW * tx;
        *reinterpret_cast<uint16_t *>(img + ch * RPIX + px) = (uint16_t)(lo | (hi << 8));
    };
#pragma unroll
    for (int r = 0; r < TW; r++) put2(nch - 1, r, 0u, 0u);
#if defined(RS_SKIP) && (RS_SKIP & 2)
    float dmin = lmin; red[0] = 1.f; redi[0] = 1;
#else
    // transform_depth (utils.py:11-19), as above
    red[tid] = lmin; __syncthreads();
    for (int s = RTHREADS / 2; s > 0; s >>= 1) { if (tid < s) red[tid] = fminf(red[tid], red[tid + s]); __syncthreads(); }
    float dmin = red[0]; __syncthreads();
    float lsum = 0.f; int lcnt = 0;
#pragma unroll
    for (int q = 0; q < TPX; q++) { float d = best[q] - dmin; if (d <= 1.0f) { lsum += d; lcnt++; } }
    red[tid] = lsum; redi[tid] = lcnt; __syncthreads();
    for (int s = RTHREADS / 2; s > 0; s >>= 1) { if (tid < s) { red[tid] += red[tid + s]; redi[tid] += redi[tid + s]; } __syncthreads(); }
#endif
    float scale = 2.0f * (red[0] / (float)redi[0]);
    if (cfg.full_observation) {
        unsigned db[TPX];
#pragma unroll
        for (int q = 0; q < TPX; q++) {
            float v = (best[q] - dmin) / scale; v = fminf(fmaxf(v, 0.f), 1.f);
            float p = 255.0f * v;
            db[q] = (p != p) ? 0u : (unsigned)(uint8_t)p;
        }
#pragma unroll
        for (int r = 0; r < TW; r++) put2(3, r, db[2 * r], db[2 * r + 1]);
    }
    __syncthreads();
    if (tid == 0) {
        const int pg_ = GPTR(const int, pad_grasp)[e], pp_ = GPTR(const int, pad_pher)[e];
        img[(nch - 1) * RPIX] = (uint8_t)pg_; img[(nch - 1) * RPIX + 1] = (uint8_t)pp_;
    }
    __syncthreads();
#ifdef RS_DEBUG_COUNTS          // diagnostic build: the row starts with the env's table counts (faces, edges, items, visible planes) instead of pixels
    if (tid == 0) { int *d = reinterpret_cast<int *>(img); d[0] = nfaces; d[1] = nedges; d[2] = nitems; d[3] = gnum[1] + gnum[2] + gnum[3] + gnum[4] + gnum[5] + gnum[6]; }
    __syncthreads();
#endif
    const uint4 *src = reinterpret_cast<const uint4 *>(img);
    for (int i = tid; i < nch * RPIX / 16; i += RTHREADS) {
        const uint4 v = src[i];
        if (o) reinterpret_cast<uint4 *>(o)[i] = v;
        if (o2) reinterpret_cast<uint4 *>(o2)[i] = v;
    }
}

// One kernel for a batch and for a set of batches: per-batch arguments in device memory behind a const __restrict__ pointer
// (workgroup-uniform index -> scalar loads). list != NULL: row b shows env list[b] (global id over the set; a negative entry is a
// hole, rows >= *count -- when count is given -- are skipped too: their rows are left alone); list == NULL: row b = env b.
__global__ void __launch_bounds__(RTHREADS, 8) k_observe_rays(const RenderGroup *__restrict__ groups, int ngroups, const int *list, const int *count,
                                                           uint8_t *obs, uint8_t *obs2, const long long *row2) {
    if (list && count && (int)blockIdx.x >= *count) return;
    const int ge = list ? list[blockIdx.x] : (int)blockIdx.x;
    if (ge < 0) return;
    int g = 0;
    for (int i = 1; i < ngroups; i++) g = ge >= groups[i].env0 ? i : g;
    const RenderGroup &rg = groups[g];
    const DevConfig cfg = rg.cfg;
    observe_body(rg.m, cfg, rg.qpos, rg.pad_grasp, rg.pad_pher, rg.n, ge - rg.env0, obs, obs2, row2);
}

__global__ void __launch_bounds__(RTHREADS, RS_WAVES) k_observe(const RenderGroup *__restrict__ groups, int ngroups, const int *list, const int *count,
                                                      uint8_t *obs, uint8_t *obs2, const long long *row2, int nverts_max, int edges_cap, int items_cap) {
    if (list && count && (int)blockIdx.x >= *count) return;
    const int ge = list ? list[blockIdx.x] : (int)blockIdx.x;
    if (ge < 0) return;
    int g = 0;
    for (int i = 1; i < ngroups; i++) g = ge >= groups[i].env0 ? i : g;
    const RenderGroup &rg = groups[g];
    const DevConfig cfg = rg.cfg;
    observe_body_raster(rg.m, cfg, rg.qpos, rg.pad_grasp, rg.pad_pher, rg.n, ge - rg.env0, obs, obs2, row2, nverts_max, edges_cap, items_cap);
}

// GRIP_OBSERVE_RAYS=1 in the environment (read once): the ray-casting kernel of rounds 1-4 instead of the rasteriser, for tools/render_ab.py's pixel comparison
static bool observe_by_rays() {
    static const bool v = [] { const char *e = getenv("GRIP_OBSERVE_RAYS"); return e && e[0] == '1'; }();
    return v;
}
#define RS_MAX_VERTS 4096

extern "C" int grip_render_launch(const RenderGroup *groups_dev, int ngroups, const int *list, const int *count, int nblocks, int nplanes_max, int nverts_max,
                                  uint8_t *obs, uint8_t *obs2, const long long *row2, hipStream_t s) {
    const bool rays = observe_by_rays();
    if (!rays && (nverts_max < 1 || nverts_max > RS_MAX_VERTS)) return grip_fail("grip_render_launch: the model's hulls have more vertices than the observation kernel's LDS table holds (RS_MAX_VERTS)");
    {   // static + dynamic LDS pass 64 KB (ray caster: ~29 KB + up to 41 KB of planes; rasteriser: ~34 KB + vertices, face records, items): opt in, once per device
        static std::atomic<unsigned long long> attr_set_mask{0ULL};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return -1;
        const unsigned long long bit = 1ULL << (dev & 63);
        if (!(attr_set_mask.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void *)k_observe_rays, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((RMAXPL + NBOX * GN_HULL) * sizeof(float4))) != hipSuccess) return -1;
            if (hipFuncSetAttribute((const void *)k_observe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RS_DYN_BYTES(RS_MAX_VERTS, 1024, 8192)) != hipSuccess) return -1;
            attr_set_mask.fetch_or(bit, std::memory_order_release);
        }
    }
    if (rays)
        hipLaunchKernelGGL(k_observe_rays, dim3(nblocks), dim3(RTHREADS), (size_t)(nplanes_max + NBOX * GN_HULL) * sizeof(float4), s, groups_dev, ngroups, list, count, obs, obs2, row2);
    else {
        // 1024 edge normals where they fit (512 at least), the rest of the budget for items (at least 1024: beyond the budget a workgroup has the CU to itself)
        const long long left = (long long)RS_LDS_BUDGET - (long long)RS_DYN_BYTES(nverts_max, 0, 0);
        const int edges_cap = left >= 1024 * 16 + 2048 * 2 ? 1024 : 512;
        const int items_cap = (int)std::min<long long>(8192, std::max<long long>(1024, (left - edges_cap * 16) / 2)) & ~7;
        hipLaunchKernelGGL(k_observe, dim3(nblocks), dim3(RTHREADS), RS_DYN_BYTES(nverts_max, edges_cap, items_cap), s, groups_dev, ngroups, list, count, obs, obs2, row2, nverts_max,
                           edges_cap, items_cap);
    }
    return launch_status("grip_render_launch");
}

// ------------------------------------------------------------------------------------------------------------------
// RobotEnv.render (robot_env.py:302-340): any camera, any size -- the workbench / upper / gripper views of the reference at their
// zoomed size (64 * rendering_zoom_width x 64 * rendering_zoom_height), for videos, GIFs and the human window. Not on the training path:
// one thread per pixel, the hull planes are read from global memory in the hull's own frame (the ray is taken there), the same
// scene, materials and shading as the observation kernel. camera = optical centre cam_o and rotation cam_R (columns: camera x, y, z in
// world coordinates, looking along -z) -- NULL: the model's gripper camera of env e; fovy in degrees (vertical).
// Output: rgb uint8 [h, w, 3] (HWC, as dm_control's physics.render returns it) and/or depth float32 [h, w], metres along the optical
// axis (zfar where nothing is hit).
__global__ void __launch_bounds__(256) k_render_camera(const RenderGroup *__restrict__ groups, int e, const float *cam, float fovy_deg, int w, int h,
                                                       uint8_t *rgb, float *depth) {
    __shared__ Frames fr;
    const RenderGroup &rg = groups[0];
    const DevModel &m = rg.m;
    if (threadIdx.x == 0) {
        compute_frames(m, rg.qpos, rg.n, e, rg.cfg.state_half, fr);
        if (cam) { for (int i = 0; i < 3; i++) fr.cam_o[i] = cam[i]; for (int i = 0; i < 9; i++) fr.cam_R[i] = cam[3 + i]; }
    }
    __syncthreads();
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    if (px >= w * h) return;
    const int i = px / w, j = px % w;
    const V3 co = ldv(fr.cam_o); const M3 Rc = ldm(fr.cam_R);
    const float th = tanf(0.5f * fovy_deg * 0.017453292519943295f), tw = th * (float)w / (float)h;
    const float x = (2.0f * (j + 0.5f) / w - 1.0f) * tw, y = (1.0f - 2.0f * (i + 0.5f) / h) * th;
    const V3 dir = mulv(Rc, v3(x, y, -1.f));                       // parameter t = distance along the optical axis
    float best = m.zfar; int hit = -1; V3 nrm = v3(0, 0, 1);
    if (dir.z < 0.f) { float t = -co.z / dir.z; if (t > m.znear && t < best) { best = t; hit = 0; } }
    for (int g = 1; g < GN_GEOM; g++) {
        const V3 p = ldv(fr.p[g - 1]); const M3 R = ldm(fr.R[g - 1]);
        const V3 ol = multv(R, co - p), dl = multv(R, dir);
        const float *pl = m.hull_planes + 4 * m.hull_padr[g - 1];
        float tin = -3.0e38f, tout = 3.0e38f; int ent = -1;
        for (int k = 0; k < m.hull_pnum[g - 1]; k++) {
            const V3 nn = v3(pl[4 * k], pl[4 * k + 1], pl[4 * k + 2]);
            const float den = dot(nn, dl), num = pl[4 * k + 3] - dot(nn, ol);
            if (den < 0.f) { const float t = num / den; if (t > tin) { tin = t; ent = k; } }
            else if (den > 0.f) tout = fminf(tout, num / den);
            else if (num < 0.f) { tin = 3.0e38f; }                 // parallel, origin outside: miss
            if (tin > tout) break;
        }
        if (ent >= 0 && !(tin > tout) && tin > m.znear && tin < best) {
            best = tin; hit = g; nrm = mulv(R, v3(pl[4 * ent], pl[4 * ent + 1], pl[4 * ent + 2]));
        }
    }
    if (depth) depth[px] = best;
    if (!rgb) return;
    float c0, c1, c2;
    if (hit < 0) {
        V3 dn = normalized(dir); float f = 0.5f * (dn.z + 1.0f);
        c0 = m.sky_rgb[3] + f * (m.sky_rgb[0] - m.sky_rgb[3]); c1 = m.sky_rgb[4] + f * (m.sky_rgb[1] - m.sky_rgb[4]); c2 = m.sky_rgb[5] + f * (m.sky_rgb[2] - m.sky_rgb[5]);
    } else {
        float b0, b1, b2;
        const V3 Pw = co + dir * best;
        if (hit == 0) {
            int cx = (int)floorf(Pw.x * 8.0f), cy = (int)floorf(Pw.y * 8.0f);
            int off = ((cx + cy) & 1) ? 3 : 0;
            b0 = m.floor_rgb[off]; b1 = m.floor_rgb[off + 1]; b2 = m.floor_rgb[off + 2];
        } else { b0 = m.geom_rgba[hit][0]; b1 = m.geom_rgba[hit][1]; b2 = m.geom_rgba[hit][2]; }
        lit_colour(m, hit, b0, b1, b2, Pw, normalized(nrm), co, c0, c1, c2);
    }
    rgb[3 * px] = to_u8(c0); rgb[3 * px + 1] = to_u8(c1); rgb[3 * px + 2] = to_u8(c2);
}

extern "C" int grip_render_camera_launch(const RenderGroup *group_dev, int env, const float *cam_dev, float fovy_deg, int w, int h, uint8_t *rgb_dev,
                                         float *depth_dev, hipStream_t s) {
    const int total = w * h;
    hipLaunchKernelGGL(k_render_camera, dim3((total + 255) / 256), dim3(256), 0, s, group_dev, env, cam_dev, fovy_deg, w, h, rgb_dev, depth_dev);
    return launch_status("grip_render_camera_launch");
}

// ------------------------------------------------------------------------------------------------------------------
// IntrinsicReward.intrinsic_reward (reward.py:57-77): sum of rel_entr between the 256-bin grey-level histograms of the
// previous and the new observation (and, with --full_observation, of their depth channels; the two sums averaged), added to
// the env's reward. One 256-thread block per pair: four LDS histograms by atomics, then one bin per thread and a block sum.
// cv2.cvtColor(BGR2GRAY) on uint8 as OpenCV 4.8.1 does it (fixed point, 15 bits: (B*3735 + G*19235 + R*9798 + 2^14) >> 15,
// channel 0 taken as B); make_pdf (utils.py:5-8) = counts / 4096 in float32.
__global__ void __launch_bounds__(256) k_intrinsic_reward(const uint8_t *old_obs, const long long *old_rows, const uint8_t *new_obs,
                                                          const int *list, const int *count, int channels, int full_observation, float *reward) {
    const int r = blockIdx.x, tid = threadIdx.x;
    if (count && r >= *count) return;
    if (list && list[r] < 0) return;                        // a hole of a merged list
    const long long orow = old_rows ? old_rows[r] : (long long)r;
    if (orow < 0) return;                                   // no previous observation (first decision of this env)
    __shared__ int h[4][256];
    __shared__ float red[256];
    for (int k = 0; k < 4; k++) h[k][tid] = 0;
    __syncthreads();
    const uint8_t *a = old_obs + (size_t)orow * channels * RPIX, *b = new_obs + (size_t)r * channels * RPIX;
    for (int px = tid; px < RPIX; px += 256) {
        int ga = (a[px] * 3735 + a[RPIX + px] * 19235 + a[2 * RPIX + px] * 9798 + (1 << 14)) >> 15;
        int gb = (b[px] * 3735 + b[RPIX + px] * 19235 + b[2 * RPIX + px] * 9798 + (1 << 14)) >> 15;
        atomicAdd(&h[0][ga], 1); atomicAdd(&h[1][gb], 1);
        if (full_observation) { atomicAdd(&h[2][a[3 * RPIX + px]], 1); atomicAdd(&h[3][b[3 * RPIX + px]], 1); }
    }
    __syncthreads();
    auto term = [](int co, int cn) {
        float p = (float)co / (float)RPIX, q = (float)cn / (float)RPIX;
        return (p > 0.f && q > 0.f) ? p * logf(p / q) : 0.f;           // rel_entr; inf (q == 0 < p) is set to 0 by reward.py:66
    };
    float t_rgb = term(h[0][tid], h[1][tid]), t_dep = full_observation ? term(h[2][tid], h[3][tid]) : 0.f;
    red[tid] = full_observation ? 0.5f * (t_rgb + t_dep) : t_rgb;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    if (tid == 0) reward[list ? list[r] : r] += red[0];
}

extern "C" int grip_intrinsic_reward(const uint8_t *old_obs_dev, const int64_t *old_rows_dev, const uint8_t *new_obs_dev, const int32_t *list_dev,
                                     const int32_t *count_dev, int n_pairs, int channels, int full_observation, float *reward_dev, void *stream) {
    if (!old_obs_dev || !new_obs_dev || !reward_dev || n_pairs <= 0 || (channels != 4 && channels != 5)) return grip_fail("grip_intrinsic_reward: bad argument");
    hipLaunchKernelGGL(k_intrinsic_reward, dim3(n_pairs), dim3(256), 0, (hipStream_t)stream, old_obs_dev, (const long long *)old_rows_dev, new_obs_dev,
                       list_dev, count_dev, channels, full_observation, reward_dev);
    return launch_status("grip_intrinsic_reward");
}

// ------------------------------------------------------------------------------------------------------------------
// Policy input in one pass: uint8 CHW observation -> float32 / 255 image channels in NHWC (the layout MIOpen's implicit-GEMM
// convolutions want: one 16-byte store per pixel for the 4 image channels) + the two sensor-pad scalars / 255
// (models/feature_extractor.py:41-49 reads them from the last channel). Replaces cast, scale and layout copy (three
// full passes over 4x the bytes).
__global__ void __launch_bounds__(256) k_obs_preprocess(const uint8_t *obs, int n, int channels, float *img, float *other) {
    const int nimg = channels - 1;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;              // pixel index over n * 4096
    if (i >= (size_t)n * RPIX) return;
    const size_t b = i / RPIX; const int px = (int)(i % RPIX);
    const uint8_t *o = obs + b * channels * RPIX;
    const float s = 1.0f / 255.0f;
    if (nimg == 4) {
        float4 v = make_float4(o[px] * s, o[RPIX + px] * s, o[2 * RPIX + px] * s, o[3 * RPIX + px] * s);
        reinterpret_cast<float4 *>(img)[i] = v;
    } else {
        for (int c = 0; c < nimg; c++) img[i * nimg + c] = o[c * RPIX + px] * s;
    }
    if (px < 2) other[b * 2 + px] = o[(size_t)nimg * RPIX + px] * s;
}

extern "C" int grip_obs_preprocess(const uint8_t *obs_dev, int n, int channels, float *img_nhwc_dev, float *other_dev, void *stream) {
    if (!obs_dev || !img_nhwc_dev || !other_dev || n <= 0 || (channels != 4 && channels != 5)) return grip_fail("grip_obs_preprocess: bad argument");
    size_t total = (size_t)n * RPIX;
    hipLaunchKernelGGL(k_obs_preprocess, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, obs_dev, n, channels, img_nhwc_dev, other_dev);
    return launch_status("grip_obs_preprocess");
}
