// grip_physics.h -- fp32 device physics for one environment per wavefront lane (gfx950).
//
// What one `physics.step()` of the reference computes (robot_env.py:100,119,142,157 ->
// dm_control Physics.step -> mj_step2 + mj_step1), restructured
// for the lane-per-env execution model:
//   * four rigid groups instead of seven bodies (grip_device.h);
//   * the constraint Jacobian is never materialised: J x and J^T f are evaluated from group
//     twists / wrenches at each contact point; for the Newton Hessian one row at a time is
//     expanded into registers and folded in as weighted rank-1 updates;
//   * per-lane variable-length data (geom frames, contact list, per-row solver scratch) sits in
//     LDS as [slot][lane], so a wave's access to one slot is one conflict-free 256-byte row;
//   * the convex hulls (vertices, edge graph, a cube-map table of start vertices) are staged once
//     per launch into LDS; support queries hill-climb the edge graph from the table entry
//     (about 9 neighbour tests per query instead of 70-573 vertex tests);
//   * every heavy routine has ONE call site inside a loop (MPR is a per-lane state machine around a
//     single support evaluation, the Newton solver a staged loop around a single constraint pass),
//     which keeps the kernel inside the instruction cache and lets lanes in different phases share
//     the same instructions.
#pragma once
#include "grip_device.h"

// ---------------------------------------------------------------- LDS layout (floats per lane)
#define GF_BASE 0                       // 6 geom frames x 12 (pos3, R9)
#define CB_BASE (GF_BASE + 72)
#define C_POS 0
#define C_N 3
#define C_DIST 6
#define C_META 7                        // g1 | g2 << 8
#define C_FS 8
#define C_FT 9
#define C_D0 10
#define C_AREF 11
#define C_JAR 15
#define C_JV 19
#define C_STRIDE 23
#define LDS_FLOATS_PER_LANE (CB_BASE + G_MAXC * C_STRIDE)
#define LDS_LANE_WORDS (LDS_FLOATS_PER_LANE * WAVE)

#define NEWTON_MAXIT 20
#define LS_MAXIT 16
#define MPR_TOL_F 1e-6f
#define MPR_MAXIT 50
#define LUT_RES 8
#define LUT_CELLS (6 * LUT_RES * LUT_RES)

#define LD(slot) lds[(slot) * WAVE + lane]

struct Kin {
    V3 pe; M3 Re; V3 a4;
    V3 pk[2], ak[2];
    V3 po; M3 Ro;
    V3 c[4];
    float Ic[4][6];
};

// hull tables staged in LDS behind the per-lane region (word offsets from DevModel)
struct Hulls {
    const float *v;                 // [nvert][4]
    const unsigned short *nadr;     // CSR over all hull vertices
    const unsigned short *nbr;      // neighbour ids, local to the hull
    const unsigned short *lut;      // [6][LUT_CELLS] start vertices
};

constexpr DEVI int pidx(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

DEVI Hulls stage_hulls(const DevModel &m, float *lds, int lane) {
    unsigned *dst = reinterpret_cast<unsigned *>(lds + LDS_LANE_WORDS);
    const unsigned *src = m.hull_blob;
    for (int i = lane; i < m.hull_words; i += WAVE) dst[i] = src[i];
    __syncthreads();
    Hulls h;
    h.v = reinterpret_cast<const float *>(dst);
    h.nadr = reinterpret_cast<const unsigned short *>(dst + m.hull_off_nadr);
    h.nbr = reinterpret_cast<const unsigned short *>(dst + m.hull_off_nbr);
    h.lut = reinterpret_cast<const unsigned short *>(dst + m.hull_off_lut);
    return h;
}

// ---------------------------------------------------------------- kinematics
DEVI void store_frame(float *lds, int lane, int g, V3 p, const M3 &R) {
    int s = GF_BASE + (g - 1) * 12;
    LD(s) = p.x; LD(s + 1) = p.y; LD(s + 2) = p.z;
#pragma unroll
    for (int i = 0; i < 9; i++) LD(s + 3 + i) = R.m[i];
}
DEVI void load_frame(const float *lds, int lane, int g, V3 &p, M3 &R) {
    int s = GF_BASE + (g - 1) * 12;
    p = v3(LD(s), LD(s + 1), LD(s + 2));
#pragma unroll
    for (int i = 0; i < 9; i++) R.m[i] = LD(s + 3 + i);
}

DEVI void kinematics(const DevModel &m, float (&qpos)[14], Kin &k, float *lds, int lane) {
    // free-joint quaternion normalised in place (mj_kinematics does the same)
    float qn = sqrtf(qpos[10] * qpos[10] + qpos[11] * qpos[11] + qpos[12] * qpos[12] + qpos[13] * qpos[13]);
    if (qn < 1e-15f) { qpos[10] = 1.f; qpos[11] = qpos[12] = qpos[13] = 0.f; }
    else { float iq = 1.0f / qn; qpos[10] *= iq; qpos[11] *= iq; qpos[12] *= iq; qpos[13] *= iq; }
    k.pe = v3(m.ee_pos0[0] + qpos[0], m.ee_pos0[1] + qpos[1], m.ee_pos0[2] + qpos[2]);
    float sr = sinf(qpos[3]), cr = cosf(qpos[3]), sy = sinf(qpos[4]), cy = cosf(qpos[4]);
    // Re = Rx(roll) * Rz(yaw)
    k.Re.m[0] = cy;      k.Re.m[1] = -sy;     k.Re.m[2] = 0.f;
    k.Re.m[3] = cr * sy; k.Re.m[4] = cr * cy; k.Re.m[5] = -sr;
    k.Re.m[6] = sr * sy; k.Re.m[7] = sr * cy; k.Re.m[8] = cr;
    k.a4 = v3(0.f, -sr, cr);
    V3 pb = k.pe + mulv(k.Re, ldv(m.base_pos));
    M3 Rb = mulm(k.Re, ldm(m.base_R));
    store_frame(lds, lane, 1, pb, Rb);
#pragma unroll
    for (int s = 0; s < 2; s++) {
        V3 pk = pb + mulv(Rb, ldv(m.kn_pos[s]));
        M3 Rk0 = mulm(Rb, ldm(m.kn_R[s]));
        k.pk[s] = pk; k.ak[s] = col(Rk0, 1);
        float sq = sinf(qpos[5 + s]), cq = cosf(qpos[5 + s]);
        M3 Ry; Ry.m[0] = cq; Ry.m[1] = 0; Ry.m[2] = sq; Ry.m[3] = 0; Ry.m[4] = 1; Ry.m[5] = 0; Ry.m[6] = -sq; Ry.m[7] = 0; Ry.m[8] = cq;
        M3 Rk = mulm(Rk0, Ry);
        store_frame(lds, lane, 2 + 2 * s, pk, Rk);
        V3 pf = pk + mulv(Rk, ldv(m.fin_pos[s]));
        M3 Rf = mulm(Rk, ldm(m.fin_R[s]));
        store_frame(lds, lane, 3 + 2 * s, pf, Rf);
        k.c[1 + s] = pk + mulv(Rk, ldv(m.grp_com[1 + s]));
        rot_sym(Rk, m.grp_inertia[1 + s], k.Ic[1 + s]);
    }
    k.po = v3(qpos[7], qpos[8], qpos[9]);
    k.Ro = quat_mat(qpos[10], qpos[11], qpos[12], qpos[13]);
    store_frame(lds, lane, 6, k.po, k.Ro);
    k.c[0] = k.pe + mulv(k.Re, ldv(m.grp_com[0]));
    rot_sym(k.Re, m.grp_inertia[0], k.Ic[0]);
    k.c[3] = k.po + mulv(k.Ro, ldv(m.grp_com[3]));
    rot_sym(k.Ro, m.grp_inertia[3], k.Ic[3]);
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

DEVI void mass_matrix(const DevModel &m, const Kin &k, float (&Mg)[28], float (&Mo)[21]) {
#pragma unroll
    for (int i = 0; i < 28; i++) Mg[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 21; i++) Mo[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 7; i++) Mg[pidx(i, i)] = m.armature[i];
#pragma unroll
    for (int i = 0; i < 6; i++) Mo[pidx(i, i)] = m.armature[7 + i];
    const V3 ex = v3(1, 0, 0), ey = v3(0, 1, 0), ez = v3(0, 0, 1), z0 = v3(0, 0, 0);
    {   const int dofs[5] = {0, 1, 2, 3, 4};
        V3 r = k.c[0] - k.pe;
        const V3 jp[5] = {ex, ey, ez, cross(ex, r), cross(k.a4, r)};
        const V3 jr[5] = {z0, z0, z0, ex, k.a4};
        add_body<5, 28>(Mg, dofs, jp, jr, m.grp_mass[0], k.Ic[0]); }
    {   const int dofs[6] = {0, 1, 2, 3, 4, 5};
        V3 r = k.c[1] - k.pe, rk = k.c[1] - k.pk[0];
        const V3 jp[6] = {ex, ey, ez, cross(ex, r), cross(k.a4, r), cross(k.ak[0], rk)};
        const V3 jr[6] = {z0, z0, z0, ex, k.a4, k.ak[0]};
        add_body<6, 28>(Mg, dofs, jp, jr, m.grp_mass[1], k.Ic[1]); }
    {   const int dofs[6] = {0, 1, 2, 3, 4, 6};
        V3 r = k.c[2] - k.pe, rk = k.c[2] - k.pk[1];
        const V3 jp[6] = {ex, ey, ez, cross(ex, r), cross(k.a4, r), cross(k.ak[1], rk)};
        const V3 jr[6] = {z0, z0, z0, ex, k.a4, k.ak[1]};
        add_body<6, 28>(Mg, dofs, jp, jr, m.grp_mass[2], k.Ic[2]); }
    {   const int dofs[6] = {0, 1, 2, 3, 4, 5};
        V3 r = k.c[3] - k.po;
        V3 c0 = col(k.Ro, 0), c1 = col(k.Ro, 1), c2 = col(k.Ro, 2);
        const V3 jp[6] = {ex, ey, ez, cross(c0, r), cross(c1, r), cross(c2, r)};
        const V3 jr[6] = {z0, z0, z0, c0, c1, c2};
        add_body<6, 21>(Mo, dofs, jp, jr, m.grp_mass[3], k.Ic[3]); }
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
DEVI void bias_forces(const DevModel &m, const Kin &k, const float (&qvel)[13], float (&bias)[13]) {
    Twist t; twists(k, qvel, t);
    V3 alG = cross(v3(qvel[3], 0, 0), k.a4 * qvel[4]);
    V3 sL = k.pk[0] - k.pe, sR = k.pk[1] - k.pe;
    V3 aL = cross(alG, sL) + cross(t.wG, cross(t.wG, sL));
    V3 aR = cross(alG, sR) + cross(t.wG, cross(t.wG, sR));
    V3 alL = alG + cross(t.wG, k.ak[0] * qvel[5]);
    V3 alR = alG + cross(t.wG, k.ak[1] * qvel[6]);
    Wrench w; wrench_zero(w);
    V3 grav = v3(0, 0, m.gravity_z);
    {   V3 rc = k.c[0] - k.pe;
        V3 ac = cross(alG, rc) + cross(t.wG, cross(t.wG, rc)) - grav;
        V3 tau = symv(k.Ic[0], alG) + cross(t.wG, symv(k.Ic[0], t.wG));
        wrench_add(k, w, GRP_G, k.c[0], ac * m.grp_mass[0], tau, 1.f); }
    {   V3 rc = k.c[1] - k.pk[0];
        V3 ac = aL + cross(alL, rc) + cross(t.wL, cross(t.wL, rc)) - grav;
        V3 tau = symv(k.Ic[1], alL) + cross(t.wL, symv(k.Ic[1], t.wL));
        wrench_add(k, w, GRP_L, k.c[1], ac * m.grp_mass[1], tau, 1.f); }
    {   V3 rc = k.c[2] - k.pk[1];
        V3 ac = aR + cross(alR, rc) + cross(t.wR, cross(t.wR, rc)) - grav;
        V3 tau = symv(k.Ic[2], alR) + cross(t.wR, symv(k.Ic[2], t.wR));
        wrench_add(k, w, GRP_R, k.c[2], ac * m.grp_mass[2], tau, 1.f); }
    {   V3 rc = k.c[3] - k.po;
        V3 ac = cross(t.wO, cross(t.wO, rc)) - grav;
        V3 tau = cross(t.wO, symv(k.Ic[3], t.wO));
        wrench_add(k, w, GRP_O, k.c[3], ac * m.grp_mass[3], tau, 1.f); }
    wrench_project(k, w, bias);
}

// ---------------------------------------------------------------- dense helpers (packed lower, static indices)
template <int N>
DEVI void chol_packed(float *A) {
#pragma unroll
    for (int j = 0; j < N; j++) {
        float s = A[pidx(j, j)];
#pragma unroll
        for (int q = 0; q < j; q++) s -= A[pidx(j, q)] * A[pidx(j, q)];
        s = fmaxf(s, 1e-30f);
        float inv = rsqrtf(s);
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

// ---------------------------------------------------------------- collision
DEVI void make_tangents(V3 n, V3 &t1, V3 &t2) {
    t1 = fabsf(n.y) < 0.5f ? v3(0, 1, 0) : v3(0, 0, 1);
    t1 = normalized(t1 - n * dot(n, t1));
    t2 = cross(n, t1);
}

// Support vertex of hull `h` (0..5) for the LOCAL direction dl: start at the cube-map table entry and
// hill-climb the edge graph to the best neighbour until no neighbour improves. Lanes walk different
// vertices; the loop is flattened to one neighbour test per iteration so lanes stay busy.
DEVI int support_vertex(const DevModel &m, const Hulls &H, int h, V3 dl, V3 &vout) {
    float ax = fabsf(dl.x), ay = fabsf(dl.y), az = fabsf(dl.z);
    int axis = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    float mj = axis == 0 ? dl.x : axis == 1 ? dl.y : dl.z;
    float u = axis == 0 ? dl.y : dl.x, v = axis == 2 ? dl.y : dl.z;
    float im = 1.0f / fmaxf(fabsf(mj), 1e-30f);
    int iu = min(LUT_RES - 1, max(0, (int)((u * im + 1.f) * (0.5f * LUT_RES))));
    int iv = min(LUT_RES - 1, max(0, (int)((v * im + 1.f) * (0.5f * LUT_RES))));
    int cell = (2 * axis + (mj < 0.f ? 1 : 0)) * (LUT_RES * LUT_RES) + iu * LUT_RES + iv;
    const int base = m.hull_vadr[h];
    const float *vb = H.v + 4 * base;
    int cur = H.lut[h * LUT_CELLS + cell];
    float bx = vb[4 * cur], by = vb[4 * cur + 1], bz = vb[4 * cur + 2];
    float bv = fmaf(bx, dl.x, fmaf(by, dl.y, bz * dl.z));
    int e = H.nadr[base + cur], eend = H.nadr[base + cur + 1];
    int cand = cur; float cv = bv, cx = bx, cy = by, cz = bz;
    for (int guard = 0; guard < 4096; guard++) {
        if (e < eend) {
            int j = H.nbr[e]; e++;
            float x = vb[4 * j], y = vb[4 * j + 1], z = vb[4 * j + 2];
            float s = fmaf(x, dl.x, fmaf(y, dl.y, z * dl.z));
            if (s > cv) { cv = s; cand = j; cx = x; cy = y; cz = z; }
        } else {
            if (cand == cur) break;
            cur = cand; bx = cx; by = cy; bz = cz;
            e = H.nadr[base + cur]; eend = H.nadr[base + cur + 1];
        }
    }
    vout = v3(bx, by, bz);
    return cur;
}

DEVI void push_contact(float *lds, int lane, int &ncon, int &fault, V3 pos, V3 n, float dist, int g1, int g2, const DevModel &m) {
    if (ncon >= G_MAXC) { fault |= 2; return; }
    int s = CB_BASE + ncon * C_STRIDE;
    LD(s + C_POS) = pos.x; LD(s + C_POS + 1) = pos.y; LD(s + C_POS + 2) = pos.z;
    LD(s + C_N) = n.x; LD(s + C_N + 1) = n.y; LD(s + C_N + 2) = n.z;
    LD(s + C_DIST) = dist;
    LD(s + C_META) = __int_as_float(g1 | (g2 << 8));
    LD(s + C_FS) = fmaxf(m.geom_friction[g1][0], m.geom_friction[g2][0]);
    LD(s + C_FT) = fmaxf(m.geom_friction[g1][1], m.geom_friction[g2][1]);
    ncon++;
}

struct Sup { V3 v, v1, v2; };

DEVI V3 portal_dir(const Sup &a, const Sup &b, const Sup &c) { return normalized(cross(b.v - a.v, c.v - a.v)); }
DEVI void expand_portal(const Sup &p0, Sup &p1, Sup &p2, Sup &p3, const Sup &p4) {
    V3 v4v0 = cross(p4.v, p0.v);
    if (dot(p1.v, v4v0) > 0.f) { if (dot(p2.v, v4v0) > 0.f) p1 = p4; else p3 = p4; }
    else { if (dot(p3.v, v4v0) > 0.f) p2 = p4; else p1 = p4; }
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

// All narrow-phase work of one state: floor-hull and hull-hull tests run through ONE loop whose body holds
// the single support evaluation. Item q < 6: floor vs hull geom q + 1 (support along -z, then graph
// neighbours inside the margin). Item q >= 6: hull pair q - 6, Minkowski portal refinement (XenoCollide) on the
// margin-inflated hulls as a per-lane phase machine: 0/1 seed the portal, 2 discover, 3 refine, 4 penetrate.
DEVI void collide(const DevModel &m, const Hulls &H, float *lds, int lane, int &ncon, int &fault) {
    ncon = 0;
    const float EPS2 = 1e-12f, EPSD = 1e-10f;
    const float infl = 0.5f * m.margin;
    const int nitem = 6 + m.npair;
    for (int q = 0; q < nitem; q++) {
        const bool plane = q < 6;
        const int g1 = plane ? 0 : m.pairs[q - 6][0], g2 = plane ? q + 1 : m.pairs[q - 6][1];
        V3 p1 = v3(0, 0, 0), p2; M3 R1, R2;
#pragma unroll
        for (int i = 0; i < 9; i++) R1.m[i] = (i % 4 == 0) ? 1.f : 0.f;
        if (!plane) load_frame(lds, lane, g1, p1, R1);
        load_frame(lds, lane, g2, p2, R2);
        V3 c2 = p2 + mulv(R2, ldv(m.geom_center[g2]));
        Sup s0, s1, s2, s3;
        V3 dir;
        int phase;                 // -1 = finished
        if (plane) {
            phase = (c2.z <= m.geom_rbound[g2] + m.margin) ? 5 : -1;
            dir = v3(0, 0, -1);
        } else {
            V3 c1 = p1 + mulv(R1, ldv(m.geom_center[g1]));
            V3 dc = c2 - c1;
            float bound = m.geom_rbound[g1] + m.geom_rbound[g2] + m.margin;
            phase = dot(dc, dc) > bound * bound ? -1 : 0;
            s0.v1 = c1; s0.v2 = c2; s0.v = c1 - c2;
            if (dot(s0.v, s0.v) < EPS2) s0.v.x += 1e-5f;
            dir = normalized(-s0.v);
        }
        int cnt = 0;
        while (__any(phase >= 0)) {
            if (phase >= 0) {
                // ---- the one support evaluation: hull g2 along -dir (and hull g1 along +dir for pairs)
                Sup s;
                V3 vl; int vi2 = support_vertex(m, H, g2 - 1, multv(R2, plane ? dir : -dir), vl);
                s.v2 = p2 + mulv(R2, vl);
                if (plane) {
                    // deepest vertex and up to three graph neighbours inside the margin (mjc_PlaneConvex)
                    if (s.v2.z <= m.margin) {
                        push_contact(lds, lane, ncon, fault, v3(s.v2.x, s.v2.y, 0.5f * s.v2.z), v3(0, 0, 1), s.v2.z, 0, g2, m);
                        const int base = m.hull_vadr[g2 - 1];
                        int e0 = H.nadr[base + vi2], e1 = H.nadr[base + vi2 + 1], extra = 0;
                        for (int e = e0; e < e1 && extra < 3; e++) {
                            int j = H.nbr[e];
                            const float *vp = H.v + 4 * (base + j);
                            V3 w = p2 + mulv(R2, v3(vp[0], vp[1], vp[2]));
                            if (w.z <= m.margin) { push_contact(lds, lane, ncon, fault, v3(w.x, w.y, 0.5f * w.z), v3(0, 0, 1), w.z, 0, g2, m); extra++; }
                        }
                    }
                    phase = -1;
                } else {
                    s.v2 = s.v2 - dir * infl;
                    V3 vl1; support_vertex(m, H, g1 - 1, multv(R1, dir), vl1);
                    s.v1 = p1 + mulv(R1, vl1) + dir * infl;
                    s.v = s.v1 - s.v2;
                    cnt++;
                    bool hit = false; float depth = 0.f; V3 nrm = v3(0, 0, 0), pos = v3(0, 0, 0);
                    if (phase == 0) {
                        s1 = s;
                        if (dot(s1.v, dir) <= 0.f) phase = -1;
                        else {
                            V3 d = cross(s0.v, s1.v);
                            if (dot(d, d) < EPS2 * 1e-2f) {      // origin on the ray v0 -> v1
                                hit = true; depth = norm(s1.v); nrm = depth < 1e-6f ? v3(0, 0, 0) : normalized(s1.v);
                                pos = (s1.v1 + s1.v2) * 0.5f;
                            } else { dir = normalized(d); phase = 1; }
                        }
                    } else if (phase == 1) {
                        s2 = s;
                        if (dot(s2.v, dir) <= 0.f) phase = -1;
                        else {
                            dir = normalized(cross(s1.v - s0.v, s2.v - s0.v));
                            if (dot(dir, s0.v) > 0.f) { Sup t = s1; s1 = s2; s2 = t; dir = -dir; }
                            phase = 2; cnt = 0;
                        }
                    } else if (phase == 2) {
                        s3 = s;
                        if (dot(s3.v, dir) <= 0.f || cnt > 4 * MPR_MAXIT) phase = -1;
                        else if (dot(cross(s1.v, s3.v), s0.v) < -EPSD) { s2 = s3; dir = normalized(cross(s1.v - s0.v, s2.v - s0.v)); }
                        else if (dot(cross(s3.v, s2.v), s0.v) < -EPSD) { s1 = s3; dir = normalized(cross(s1.v - s0.v, s2.v - s0.v)); }
                        else {
                            dir = portal_dir(s1, s2, s3);
                            phase = dot(dir, s1.v) >= -EPSD ? 4 : 3; cnt = 0;
                        }
                    } else if (phase == 3) {
                        if (dot(s.v, dir) < 0.f || reach_tol(s1, s2, s3, s, dir) || cnt > MPR_MAXIT) phase = -1;
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
                        } else {
                            expand_portal(s0, s1, s2, s3, s);
                            dir = portal_dir(s1, s2, s3);
                        }
                    }
                    if (hit) {
                        float dist = m.margin - depth;
                        if (dist < m.margin) {
                            if (dot(nrm, nrm) < 0.5f) nrm = normalized(s0.v2 - s0.v1);
                            push_contact(lds, lane, ncon, fault, pos, nrm, dist, g1, g2, m);
                        }
                        phase = -1;
                    }
                }
            }
        }
    }
}

// actuator.py:134-184 on the device contact list
DEVI int check_grasp(const float *lds, int lane, int ncon) {
    int t1 = 0, t2 = 0;
    for (int c = 0; c < ncon; c++) {
        int meta = __float_as_int(LD(CB_BASE + c * C_STRIDE + C_META));
        int g1 = meta & 255, g2 = (meta >> 8) & 255, other;
        if (g1 == 6) other = g2; else if (g2 == 6) other = g1; else continue;
        if (other == 2 || other == 3) t1 = 1;
        if (other == 4 || other == 5) t2 = 1;
    }
    return t1 + 2 * t2;
}

// ---------------------------------------------------------------- soft constraints
DEVI float impedance(const float *si, float pos, float margin) {
    float x = fabsf(pos - margin) / fmaxf(1e-15f, si[2]);
    if (x >= 1.f) return si[1];
    if (x <= 0.f) return si[0];
    float mid = si[3], power = si[4], y;
    if (power <= 1.f + 1e-6f) y = x;
    else if (x <= mid) y = __powf(x, power) / __powf(mid, power - 1.f);
    else y = 1.f - __powf(1.f - x, power) / __powf(1.f - mid, power - 1.f);
    return si[0] + y * (si[1] - si[0]);
}

struct Limits { float sgn[7], D[7], aref[7]; };

// Elliptic condim-4 contact in jar space:  s(jar) = (D0 / 2 mu^2) dist^2(U, K), U = diag(mu, fs, fs, ft) jar,
// K = {U0 >= mu |U_t|}. Returns the cost and gradient, plus the Hessian in rank-structured form
//   H = diag(w) + ka a a^T - kb b b^T      (top zone: all zero; bottom zone: w = D, ka = kb = 0)
struct Cone { float cost, grad[4], w[4], a[4], b[4], ka, kb; };
DEVI void cone_eval(const float (&jar)[4], float D0, float impratio, float fs, float ft, Cone &c) {
    float mu = fs * rsqrtf(impratio);
    float D1 = D0 * impratio, D3 = D1 * ft * ft / fmaxf(1e-30f, fs * fs);
    float S[4] = {mu, fs, fs, ft}, U[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { U[i] = S[i] * jar[i]; c.grad[i] = 0.f; c.w[i] = 0.f; c.a[i] = 0.f; c.b[i] = 0.f; }
    c.cost = 0.f; c.ka = 0.f; c.kb = 0.f;
    float N = U[0], T = sqrtf(U[1] * U[1] + U[2] * U[2] + U[3] * U[3]);
    if (N >= mu * T || (T <= 0.f && N >= 0.f)) return;
    if (mu * N + T <= 0.f || (T <= 0.f && N < 0.f)) {
        float D[4] = {D0, D1, D1, D3};
#pragma unroll
        for (int i = 0; i < 4; i++) { c.cost += 0.5f * D[i] * jar[i] * jar[i]; c.grad[i] = D[i] * jar[i]; c.w[i] = D[i]; }
        return;
    }
    float kap = D0 / fmaxf(1e-30f, mu * mu), s1 = rsqrtf(1.f + mu * mu);
    float dist = (mu * T - N) * s1, invT = 1.0f / T;
    float c2 = dist * mu * s1 * invT;
    c.a[0] = -s1 * S[0];
#pragma unroll
    for (int i = 1; i < 4; i++) { float t = U[i] * invT; c.a[i] = mu * s1 * t * S[i]; c.b[i] = t * S[i]; c.w[i] = kap * c2 * S[i] * S[i]; }
#pragma unroll
    for (int i = 0; i < 4; i++) c.grad[i] = kap * dist * c.a[i];
    c.ka = kap; c.kb = kap * c2;
    c.cost = 0.5f * kap * dist * dist;
}
// jv^T H jv for the structured Hessian
DEVI float cone_quad(const Cone &c, const float (&jv)[4]) {
    float q = 0.f, da = 0.f, db = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) { q = fmaf(c.w[i] * jv[i], jv[i], q); da = fmaf(c.a[i], jv[i], da); db = fmaf(c.b[i], jv[i], db); }
    return q + c.ka * da * da - c.kb * db * db;
}

struct ContactGeo { V3 p, n, t1, t2; int gA, gB; float fs, ft, D0; };
DEVI void load_contact(const DevModel &m, const float *lds, int lane, int c, ContactGeo &g) {
    int s = CB_BASE + c * C_STRIDE;
    g.p = v3(LD(s + C_POS), LD(s + C_POS + 1), LD(s + C_POS + 2));
    g.n = v3(LD(s + C_N), LD(s + C_N + 1), LD(s + C_N + 2));
    make_tangents(g.n, g.t1, g.t2);
    int meta = __float_as_int(LD(s + C_META));
    g.gA = m.geom_group[meta & 255]; g.gB = m.geom_group[(meta >> 8) & 255];
    g.fs = LD(s + C_FS); g.ft = LD(s + C_FT); g.D0 = LD(s + C_D0);
}
// relative motion rows of contact g for group twists t: (n.v, t1.v, t2.v, n.w)
DEVI void contact_rows(const Kin &k, const Twist &t, const ContactGeo &g, float (&r)[4]) {
    V3 vA, wA, vB, wB;
    group_motion(k, t, g.gA, g.p, vA, wA); group_motion(k, t, g.gB, g.p, vB, wB);
    V3 dv = vB - vA, dw = wB - wA;
    r[0] = dot(g.n, dv); r[1] = dot(g.t1, dv); r[2] = dot(g.t2, dv); r[3] = dot(g.n, dw);
}

// reference accelerations and regularisation for limits and contacts (mj_makeConstraint / mj_makeImpedance)
DEVI void make_constraints(const DevModel &m, const Kin &k, const float (&qpos)[14], const float (&qvel)[13],
                           Limits &lim, float *lds, int lane, int ncon) {
#pragma unroll
    for (int j = 0; j < 7; j++) {
        float lo = qpos[j] - m.range[j][0], hi = m.range[j][1] - qpos[j];
        float sgn = 0.f, dist = 0.f;
        if (lo < 0.f) { sgn = 1.f; dist = lo; } else if (hi < 0.f) { sgn = -1.f; dist = hi; }
        float imp = impedance(m.lim_solimp, dist, 0.f);
        float R = fmaxf(1e-15f, (1.f - imp) * m.dof_invweight0[j] / imp);
        lim.sgn[j] = sgn; lim.D[j] = 1.0f / R;
        lim.aref[j] = -m.b_lim * (sgn * qvel[j]) - m.k_lim * imp * dist;
    }
    Twist tv; twists(k, qvel, tv);
    for (int c = 0; c < ncon; c++) {
        int s = CB_BASE + c * C_STRIDE;
        ContactGeo g; load_contact(m, lds, lane, c, g);
        float dist = LD(s + C_DIST);
        int meta = __float_as_int(LD(s + C_META));
        float imp = impedance(m.solimp, dist, m.margin);
        float tran = m.geom_invweight[meta & 255] + m.geom_invweight[(meta >> 8) & 255];
        float R0 = fmaxf(1e-15f, (1.f - imp) * tran / imp);
        LD(s + C_D0) = 1.0f / R0;
        float vel[4]; contact_rows(k, tv, g, vel);
        LD(s + C_AREF) = -m.b_con * vel[0] - m.k_con * imp * (dist - m.margin);
        LD(s + C_AREF + 1) = -m.b_con * vel[1];
        LD(s + C_AREF + 2) = -m.b_con * vel[2];
        LD(s + C_AREF + 3) = -m.b_con * vel[3];
    }
}

// Hessian storage: the full problem keeps 13x13 packed lower (91); the object-only problem (no constraint touches
// the gripper: the usual case, object resting on the floor while the gripper moves freely) keeps the 6x6 object block (21).
template <bool OBJ> struct HessT { static constexpr int N = OBJ ? 21 : 91; };
template <bool OBJ> constexpr DEVI int hidx(int i, int j) { return OBJ ? pidx(i - 7, j - 7) : pidx(i, j); }

// H += w * u u^T restricted to dofs [lo, 13)
template <bool OBJ, int LO>
DEVI void rank1(float *Hp, const float (&u)[13], float w) {
#pragma unroll
    for (int i = LO; i < 13; i++) {
        float wi = w * u[i];
#pragma unroll
        for (int j = LO; j <= i; j++) Hp[hidx<OBJ>(i, j)] = fmaf(wi, u[j], Hp[hidx<OBJ>(i, j)]);
    }
}

// One pass over all constraints at acceleration x: stores jar in LDS, returns the constraint cost,
// accumulates J^T force into `jtf` and, if wantH, adds J^T s'' J to the packed Hessian.
template <bool OBJ>
DEVI float constraint_pass(const DevModel &m, const Kin &k, const Limits &lim, const float (&x)[13],
                           float *lds, int lane, int ncon, float (&jtf)[13], float *Hp, bool wantH) {
    float cost = 0.f;
    Wrench w; wrench_zero(w);
    float flim[7];
#pragma unroll
    for (int j = 0; j < 7; j++) flim[j] = 0.f;
    if (!OBJ) {
#pragma unroll
        for (int j = 0; j < 7; j++) {
            float jar = lim.sgn[j] * x[j] - lim.aref[j];
            bool act = lim.sgn[j] != 0.f && jar < 0.f;
            flim[j] = act ? -lim.D[j] * jar * lim.sgn[j] : 0.f;
            cost += act ? 0.5f * lim.D[j] * jar * jar : 0.f;
            if (wantH) Hp[hidx<OBJ>(OBJ ? 7 : j, OBJ ? 7 : j)] += act ? lim.D[j] : 0.f;
        }
    }
    Twist t; twists(k, x, t);
    for (int c = 0; c < ncon; c++) {
        int s = CB_BASE + c * C_STRIDE;
        ContactGeo g; load_contact(m, lds, lane, c, g);
        float jar[4];
        if (OBJ) {          // floor-object contact: only the object moves
            V3 dv = t.vO + cross(t.wO, g.p - k.po);
            jar[0] = dot(g.n, dv); jar[1] = dot(g.t1, dv); jar[2] = dot(g.t2, dv); jar[3] = dot(g.n, t.wO);
        } else contact_rows(k, t, g, jar);
#pragma unroll
        for (int r = 0; r < 4; r++) { jar[r] -= LD(s + C_AREF + r); LD(s + C_JAR + r) = jar[r]; }
        Cone cn; cone_eval(jar, g.D0, m.impratio, g.fs, g.ft, cn);
        cost += cn.cost;
        V3 F = g.n * (-cn.grad[0]) + g.t1 * (-cn.grad[1]) + g.t2 * (-cn.grad[2]);
        V3 Tq = g.n * (-cn.grad[3]);
        if (OBJ) { w.FO = w.FO + F; w.TO = w.TO + Tq + cross(g.p - k.po, F); }
        else { wrench_add(k, w, g.gB, g.p, F, Tq, 1.f); wrench_add(k, w, g.gA, g.p, F, Tq, -1.f); }
        bool any = cn.w[0] != 0.f || cn.w[1] != 0.f || cn.ka != 0.f;
        if (wantH && any) {
            const bool objonly = OBJ || (g.gA == GRP_WORLD && g.gB == GRP_O);     // rows touch dofs 7..12 only
            float ua[13], ub[13];
#pragma unroll
            for (int i = 0; i < 13; i++) { ua[i] = 0.f; ub[i] = 0.f; }
#pragma unroll 1
            for (int r = 0; r < 4; r++) {
                float j[13];
#pragma unroll
                for (int i = 0; i < 13; i++) j[i] = 0.f;
                V3 e = r == 1 ? g.t1 : r == 2 ? g.t2 : g.n;
                if (OBJ) row_add(k, j, GRP_O, g.p, e, 1.f, r == 3);
                else { row_add(k, j, g.gB, g.p, e, 1.f, r == 3); row_add(k, j, g.gA, g.p, e, -1.f, r == 3); }
                float ar = r == 0 ? cn.a[0] : r == 1 ? cn.a[1] : r == 2 ? cn.a[2] : cn.a[3];
                float br = r == 0 ? cn.b[0] : r == 1 ? cn.b[1] : r == 2 ? cn.b[2] : cn.b[3];
                float wr = r == 0 ? cn.w[0] : r == 1 ? cn.w[1] : r == 2 ? cn.w[2] : cn.w[3];
#pragma unroll
                for (int i = OBJ ? 7 : 0; i < 13; i++) { ua[i] = fmaf(ar, j[i], ua[i]); ub[i] = fmaf(br, j[i], ub[i]); }
                if (objonly) rank1<OBJ, 7>(Hp, j, wr); else rank1<OBJ, OBJ ? 7 : 0>(Hp, j, wr);
            }
            if (cn.ka != 0.f) {
#pragma unroll 1
                for (int q = 0; q < 2; q++) {
                    float u[13];
#pragma unroll
                    for (int i = 0; i < 13; i++) u[i] = q == 0 ? ua[i] : ub[i];
                    float wq = q == 0 ? cn.ka : -cn.kb;
                    if (objonly) rank1<OBJ, 7>(Hp, u, wq); else rank1<OBJ, OBJ ? 7 : 0>(Hp, u, wq);
                }
            }
        }
    }
    if (OBJ) {
#pragma unroll
        for (int i = 0; i < 7; i++) jtf[i] = 0.f;
        jtf[7] = w.FO.x; jtf[8] = w.FO.y; jtf[9] = w.FO.z;
        V3 tl = multv(k.Ro, w.TO);
        jtf[10] = tl.x; jtf[11] = tl.y; jtf[12] = tl.z;
    } else {
        wrench_project(k, w, jtf);
#pragma unroll
        for (int j = 0; j < 7; j++) jtf[j] += flim[j];
    }
    return cost;
}

// phi'(alpha), phi''(alpha) of the total cost along the search direction (jar, jv cached in LDS)
template <bool OBJ>
DEVI void line_eval(const DevModel &m, const Limits &lim, const float (&qacc)[13], const float (&p)[13],
                    const float *lds, int lane, int ncon, float alpha, float g0, float g1, float &dphi, float &ddphi) {
    float dp = g0 + alpha * g1, hp = g1;
    if (!OBJ) {
#pragma unroll
        for (int j = 0; j < 7; j++) {
            float jv = lim.sgn[j] * p[j];
            float x = lim.sgn[j] * qacc[j] - lim.aref[j] + alpha * jv;
            bool act = lim.sgn[j] != 0.f && x < 0.f;
            dp += act ? lim.D[j] * x * jv : 0.f; hp += act ? lim.D[j] * jv * jv : 0.f;
        }
    }
    for (int c = 0; c < ncon; c++) {
        int s = CB_BASE + c * C_STRIDE;
        float ja[4], jv[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { jv[r] = LD(s + C_JV + r); ja[r] = LD(s + C_JAR + r) + alpha * jv[r]; }
        Cone cn; cone_eval(ja, LD(s + C_D0), m.impratio, LD(s + C_FS), LD(s + C_FT), cn);
#pragma unroll
        for (int r = 0; r < 4; r++) dp = fmaf(cn.grad[r], jv[r], dp);
        hp += cone_quad(cn, jv);
    }
    dphi = dp; ddphi = hp;
}

// Primal Newton solve of  min 1/2 (a - a_s)^T M (a - a_s) + s(J a - aref)   (mj_solNewton's problem).
// Staged loop with ONE constraint-pass site: stage 0 prices qacc_smooth, stage 1 prices qacc_warmstart (the better
// one is the start, as MuJoCo does) and stops right there when the start already satisfies the gradient
// tolerance -- the common case of a persisting contact, which then never builds a Hessian; stages >= 2 are
// Newton iterations (Hessian, Cholesky, exact line search). OBJ = only the 6 object dofs are constrained.
template <bool OBJ>
DEVI void solve_newton(const DevModel &m, const Kin &k, const float (&Mg)[28], const float (&Mo)[21], const Limits &lim,
                       const float (&qs)[13], const float (&warm)[13], float *lds, int lane, int ncon,
                       float (&qacc)[13], float (&jtf)[13], int &fault, int &iters) {
    constexpr int LO = OBJ ? 7 : 0;
    constexpr int NH = HessT<OBJ>::N;
    const float scale = 1.0f / (m.meaninertia * 13.f);
    const float tol = fmaxf(m.tolerance, 1e-5f);            // fp32 noise floor of the scaled gradient is ~1e-6
    float H[NH], Md[13], x[13];
    float cost = 0.f, cs = 0.f;
    int stage = 0; bool done = false;
    iters = 0;
#pragma unroll
    for (int i = 0; i < 13; i++) { qacc[i] = qs[i]; x[i] = qs[i]; Md[i] = 0.f; }
    while (__any(!done)) {
        if (!done) {
            const bool wantH = stage >= 2;
            if (wantH) {
#pragma unroll
                for (int i = 0; i < NH; i++) H[i] = 0.f;
                if (!OBJ) {
#pragma unroll
                    for (int i = 0; i < 7; i++)
#pragma unroll
                        for (int j = 0; j <= i; j++) H[hidx<OBJ>(OBJ ? 7 : i, OBJ ? 7 : j)] = Mg[pidx(i, j)];
                }
#pragma unroll
                for (int i = 0; i < 6; i++)
#pragma unroll
                    for (int j = 0; j <= i; j++) H[hidx<OBJ>(7 + i, 7 + j)] = Mo[pidx(i, j)];
            }
            float dq[13];
#pragma unroll
            for (int i = 0; i < 13; i++) dq[i] = x[i] - qs[i];
            mass_mulv(Mg, Mo, dq, Md);
            float newcost = 0.f;
#pragma unroll
            for (int i = LO; i < 13; i++) newcost = fmaf(0.5f * Md[i], dq[i], newcost);
            newcost += constraint_pass<OBJ>(m, k, lim, x, lds, lane, ncon, jtf, H, wantH);
            float gn = 0.f;
#pragma unroll
            for (int i = LO; i < 13; i++) { float g = Md[i] - jtf[i]; gn = fmaf(g, g, gn); }
            const bool gconv = scale * sqrtf(gn) < tol;
            if (stage == 0) {
                cs = newcost; stage = 1;
#pragma unroll
                for (int i = LO; i < 13; i++) x[i] = warm[i];
                if (gconv) done = true;                     // qacc_smooth already optimal (constraints inactive)
            } else if (stage == 1) {
                if (newcost < cs) {
#pragma unroll
                    for (int i = LO; i < 13; i++) qacc[i] = warm[i];
                    if (gconv) done = true;
                } else {
#pragma unroll
                    for (int i = LO; i < 13; i++) x[i] = qs[i];
                }
                stage = 2;
            } else {
                bool stop = gconv;
                if (stage > 2 && scale * (cost - newcost) < tol) stop = true;
                cost = newcost;
                if (!stop && iters >= NEWTON_MAXIT) { stop = true; fault |= 4; }
                if (stop) done = true;
                else {
                    float p[13];
#pragma unroll
                    for (int i = 0; i < 13; i++) p[i] = 0.f;
                    chol_packed<13 - LO>(H);
                    {   float pr[13 - LO];
#pragma unroll
                        for (int i = LO; i < 13; i++) pr[i - LO] = -(Md[i] - jtf[i]);
                        chol_solve_packed<13 - LO>(H, pr);
#pragma unroll
                        for (int i = LO; i < 13; i++) p[i] = pr[i - LO]; }
                    float Mpv[13]; mass_mulv(Mg, Mo, p, Mpv);
                    float g0 = 0.f, g1 = 0.f;
#pragma unroll
                    for (int i = LO; i < 13; i++) { g0 = fmaf(Mpv[i], qacc[i] - qs[i], g0); g1 = fmaf(Mpv[i], p[i], g1); }
                    {   Twist tp; twists(k, p, tp);
                        for (int c = 0; c < ncon; c++) {
                            ContactGeo g; load_contact(m, lds, lane, c, g);
                            float jv[4];
                            if (OBJ) {
                                V3 dv = tp.vO + cross(tp.wO, g.p - k.po);
                                jv[0] = dot(g.n, dv); jv[1] = dot(g.t1, dv); jv[2] = dot(g.t2, dv); jv[3] = dot(g.n, tp.wO);
                            } else contact_rows(k, tp, g, jv);
#pragma unroll
                            for (int r = 0; r < 4; r++) LD(CB_BASE + c * C_STRIDE + C_JV + r) = jv[r];
                        } }
                    // exact line search: safeguarded 1-D Newton on phi'(alpha); one evaluation site
                    float lo = 0.f, hi = -1.f, alpha = 0.f, gtol = 0.f;
                    bool lsdone = false, descent = true;
                    for (int ls = 0; ls <= LS_MAXIT; ls++) {
                        if (!__any(!lsdone)) break;
                        if (!lsdone) {
                            float dp, hp;
                            line_eval<OBJ>(m, lim, qacc, p, lds, lane, ncon, alpha, g0, g1, dp, hp);
                            if (ls == 0) {
                                if (dp >= 0.f) { descent = false; lsdone = true; }
                                gtol = 1e-4f * fabsf(dp) + 1e-30f;
                            } else {
                                if (fabsf(dp) < gtol) lsdone = true;
                                else {
                                    if (dp < 0.f) lo = alpha; else hi = alpha;
                                    if (hi > 0.f && (hi - lo) < 1e-6f * hi) lsdone = true;
                                }
                            }
                            if (!lsdone && ls < LS_MAXIT) {
                                float an = alpha - dp / fmaxf(hp, 1e-30f);
                                if (hi > 0.f && (an <= lo || an >= hi)) an = 0.5f * (lo + hi);
                                alpha = an;
                            }
                        }
                    }
                    if (!descent) done = true;
                    else {
#pragma unroll
                        for (int i = LO; i < 13; i++) { qacc[i] = fmaf(alpha, p[i], qacc[i]); x[i] = qacc[i]; }
                        iters++; stage++;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------- one physics.step()
struct LaneState { float qpos[14], qvel[13], ctrl[7], warm[13]; };

// position stage (mj_step1's share): kinematics + collision of the current state; contacts stay in LDS.
DEVI void forward_pos(const DevModel &m, const Hulls &H, LaneState &s, float *lds, int lane, Kin &k, int &ncon, int &fault) {
    kinematics(m, s.qpos, k, lds, lane);
    collide(m, H, lds, lane, ncon, fault);
}

// dynamics stage (mj_step2's share up to qacc) on top of forward_pos
DEVI void forward_acc(const DevModel &m, LaneState &s, float xfrc_z, float *lds, int lane, const Kin &k, int ncon, int &fault,
                      float (&Mg)[28], float (&Mo)[21], float (&qfrc_smooth)[13], float (&qacc)[13], float (&jtf)[13], int &iters,
                      float *dbg_qs, float *dbg_bias) {
    mass_matrix(m, k, Mg, Mo);
    float bias[13];
    bias_forces(m, k, s.qvel, bias);
    if (dbg_bias) {
#pragma unroll
        for (int i = 0; i < 13; i++) dbg_bias[i] = bias[i];
    }
#pragma unroll
    for (int i = 0; i < 13; i++) qfrc_smooth[i] = -m.damping[i] * s.qvel[i] - bias[i];
#pragma unroll
    for (int u = 0; u < 7; u++) qfrc_smooth[u] += m.gear[u] * fminf(fmaxf(s.ctrl[u], m.ctrlrange[u][0]), m.ctrlrange[u][1]);
    // xfrc_applied on body ee (force along z at its COM = frame origin): only the z slide sees it
    qfrc_smooth[2] += xfrc_z;
    float qs[13];
    {   // block solves M qacc_smooth = qfrc_smooth  (gripper 7x7, object 6x6; the cross block is zero)
        float Lg[28], Lo[21], xg[7], xo[6];
#pragma unroll
        for (int i = 0; i < 28; i++) Lg[i] = Mg[i];
#pragma unroll
        for (int i = 0; i < 21; i++) Lo[i] = Mo[i];
        chol_packed<7>(Lg); chol_packed<6>(Lo);
#pragma unroll
        for (int i = 0; i < 7; i++) xg[i] = qfrc_smooth[i];
#pragma unroll
        for (int i = 0; i < 6; i++) xo[i] = qfrc_smooth[7 + i];
        chol_solve_packed<7>(Lg, xg); chol_solve_packed<6>(Lo, xo);
#pragma unroll
        for (int i = 0; i < 7; i++) qs[i] = xg[i];
#pragma unroll
        for (int i = 0; i < 6; i++) qs[7 + i] = xo[i];
    }
    if (dbg_qs) {
#pragma unroll
        for (int i = 0; i < 13; i++) dbg_qs[i] = qs[i];
    }
    Limits lim;
    make_constraints(m, k, s.qpos, s.qvel, lim, lds, lane, ncon);
    bool anylim = false;
#pragma unroll
    for (int j = 0; j < 7; j++) anylim |= lim.sgn[j] != 0.f;
    iters = 0;
    // does any constraint touch the gripper? (joint limit, or a contact whose geoms are not floor + object)
    bool grip = anylim;
    for (int c = 0; c < ncon; c++) {
        int meta = __float_as_int(LD(CB_BASE + c * C_STRIDE + C_META));
        grip |= !((meta & 255) == 0 && ((meta >> 8) & 255) == 6);
    }
    const bool constrained = ncon > 0 || anylim;
#pragma unroll
    for (int i = 0; i < 13; i++) { qacc[i] = qs[i]; jtf[i] = 0.f; }
    if (__any(constrained && !grip)) {
        if (constrained && !grip) solve_newton<true>(m, k, Mg, Mo, lim, qs, s.warm, lds, lane, ncon, qacc, jtf, fault, iters);
    }
    if (__any(constrained && grip)) {
        if (constrained && grip) solve_newton<false>(m, k, Mg, Mo, lim, qs, s.warm, lds, lane, ncon, qacc, jtf, fault, iters);
    }
}

// dynamics + integration of one physics.step(); forward_pos must have run on the current state
DEVI void physics_advance(const DevModel &m, LaneState &s, float xfrc_z, float *lds, int lane, const Kin &k, int ncon, int &fault) {
    float Mg[28], Mo[21], qfs[13], qacc[13], jtf[13]; int iters;
    forward_acc(m, s, xfrc_z, lds, lane, k, ncon, fault, Mg, Mo, qfs, qacc, jtf, iters, nullptr, nullptr);
    const float h = m.timestep;
    // semi-implicit Euler with implicit joint damping: (M + h D) a' = qfrc_smooth + J^T f.
    // Only the gripper block carries damping; for the object block a' = qacc.
    float xg[7];
#pragma unroll
    for (int i = 0; i < 7; i++) { Mg[pidx(i, i)] += h * m.damping[i]; xg[i] = qfs[i] + jtf[i]; }
    chol_packed<7>(Mg);
    chol_solve_packed<7>(Mg, xg);
#pragma unroll
    for (int i = 0; i < 13; i++) s.warm[i] = qacc[i];
#pragma unroll
    for (int i = 0; i < 7; i++) s.qvel[i] = fmaf(h, xg[i], s.qvel[i]);
#pragma unroll
    for (int i = 7; i < 13; i++) s.qvel[i] = fmaf(h, qacc[i], s.qvel[i]);
#pragma unroll
    for (int i = 0; i < 10; i++) s.qpos[i] = fmaf(h, s.qvel[i], s.qpos[i]);
    V3 w = v3(s.qvel[10], s.qvel[11], s.qvel[12]);
    float ang = norm(w) * h;
    if (ang > 0.f) {
        V3 ax = normalized(w);
        float sh = sinf(0.5f * ang), ch = cosf(0.5f * ang);
        float bw = ch, bx = ax.x * sh, by = ax.y * sh, bz = ax.z * sh;
        float aw = s.qpos[10], axx = s.qpos[11], ay = s.qpos[12], az = s.qpos[13];
        float rw = aw * bw - axx * bx - ay * by - az * bz;
        float rx = aw * bx + axx * bw + ay * bz - az * by;
        float ry = aw * by - axx * bz + ay * bw + az * bx;
        float rz = aw * bz + axx * by - ay * bx + az * bw;
        float inv = rsqrtf(rw * rw + rx * rx + ry * ry + rz * rz);
        s.qpos[10] = rw * inv; s.qpos[11] = rx * inv; s.qpos[12] = ry * inv; s.qpos[13] = rz * inv;
    }
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 14; i++) bad |= !(fabsf(s.qpos[i]) < 1e6f);
#pragma unroll
    for (int i = 0; i < 13; i++) bad |= !(fabsf(s.qvel[i]) < 1e6f);
    if (bad) fault |= 1;
}
