/*
 * grip_physics.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Restates what one `physics.step()` of the reference does
 * (simulation/environment/robot_env.py:100,119,142,157 -> dm_control
 * Physics.step -> MuJoCo mj_step2 then mj_step1; SURVEY.md §3.2-note and
 * Appendix C) for the fixed gripper + free object topology of xmls/<object>_env.xml.
 * MuJoCo is a third-party dependency absent from /root/reference: the
 * pipeline below follows its published computation model [3P-recall] and is
 * PARITY UNPINNED (see grip_oracle.h).
 *
 *   step2: actuation (ctrl clamp, gear)  -> qfrc_smooth -> qacc_smooth = M^-1 ..
 *          -> soft constraints (joint limits + elliptic condim-4 contacts)
 *          -> primal Newton solve for qacc -> semi-implicit Euler with implicit
 *          joint damping
 *   step1: kinematics, mass matrix, collision (plane-hull, hull-hull MPR)
 */
#include "grip_oracle_int.h"

unsigned long orc_sizeof_data(void) { return sizeof(OrcData); }

/* ------------------------------------------------------------------ kinematics */
static void set_body(OrcData *d, int b, const double p[3], const double q[4]) {
    copy3(d->xpos[b], p);
    memcpy(d->xquat[b], q, sizeof(double) * 4);
    quat_to_mat(d->xmat[b], q);
}

static void kinematics(const OrcModel *m, OrcData *d) {
    static const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ez[3] = {0, 0, 1};
    const double *axes[3] = {ex, ey, ez};
    double p[3], q[4], R[9], t[3], qj[4], qn[4];
    /* quaternion of the free joint is normalised in place, as mj_kinematics does */
    quat_normalize(d->qpos + 10);
    double p0[3] = {0, 0, 0}, q0[4] = {1, 0, 0, 0};
    set_body(d, B_WORLD, p0, q0);
    /* ee: three slides, roll, yaw, applied in MJCF order */
    copy3(p, m->body_pos[B_EE]); memcpy(q, m->body_quat[B_EE], sizeof q);
    quat_to_mat(R, q);
    for (int i = 0; i < 3; i++) {
        mulmv3(d->dof_axis[i], R, axes[i]);
        copy3(d->dof_anchor[i], p);
        addscl3(p, p, d->dof_axis[i], d->qpos[i]);
    }
    mulmv3(d->dof_axis[3], R, ex); copy3(d->dof_anchor[3], p);
    axis_quat(qj, ex, d->qpos[3]); quat_mul(qn, q, qj); memcpy(q, qn, sizeof q); quat_to_mat(R, q);
    mulmv3(d->dof_axis[4], R, ez); copy3(d->dof_anchor[4], p);
    axis_quat(qj, ez, d->qpos[4]); quat_mul(qn, q, qj); memcpy(q, qn, sizeof q);
    quat_normalize(q);
    set_body(d, B_EE, p, q);
    /* base: welded to ee */
    mulmv3(t, d->xmat[B_EE], m->body_pos[B_BASE]); add3(p, d->xpos[B_EE], t);
    quat_mul(q, d->xquat[B_EE], m->body_quat[B_BASE]); quat_normalize(q);
    set_body(d, B_BASE, p, q);
    /* knuckles (hinge about local y) and welded fingers */
    const int kb[2] = {B_LK, B_RK}, fb[2] = {B_LF, B_RF}, kd[2] = {5, 6};
    for (int s = 0; s < 2; s++) {
        mulmv3(t, d->xmat[B_BASE], m->body_pos[kb[s]]); add3(p, d->xpos[B_BASE], t);
        quat_mul(q, d->xquat[B_BASE], m->body_quat[kb[s]]); quat_normalize(q);
        quat_to_mat(R, q);
        mulmv3(d->dof_axis[kd[s]], R, ey); copy3(d->dof_anchor[kd[s]], p);
        axis_quat(qj, ey, d->qpos[kd[s]]); quat_mul(qn, q, qj); quat_normalize(qn);
        set_body(d, kb[s], p, qn);
        mulmv3(t, d->xmat[kb[s]], m->body_pos[fb[s]]); add3(p, d->xpos[kb[s]], t);
        quat_mul(q, d->xquat[kb[s]], m->body_quat[fb[s]]); quat_normalize(q);
        set_body(d, fb[s], p, q);
    }
    /* object: free joint */
    set_body(d, B_OBJ, d->qpos + 7, d->qpos + 10);
    for (int i = 0; i < 3; i++) {
        copy3(d->dof_axis[7 + i], axes[i]); copy3(d->dof_anchor[7 + i], d->xpos[B_OBJ]);
        mulmv3(d->dof_axis[10 + i], d->xmat[B_OBJ], axes[i]); copy3(d->dof_anchor[10 + i], d->xpos[B_OBJ]);
    }
    for (int b = 0; b < NB; b++) {
        mulmv3(t, d->xmat[b], m->body_ipos[b]); add3(d->xipos[b], d->xpos[b], t);
    }
}

static int dof_is_linear(int dof) { return dof < 3 || (dof >= 7 && dof < 10); }

void orc_jac_body(const OrcModel *m, const OrcData *d, int body, const double point[3],
                  double jacp[3][NV], double jacr[3][NV]) {
    (void)m;
    for (int r = 0; r < 3; r++) for (int c = 0; c < NV; c++) { jacp[r][c] = 0; jacr[r][c] = 0; }
    for (int k = 0; k < k_body_ndof[body]; k++) {
        int dof = k_body_dofs[body][k];
        const double *ax = d->dof_axis[dof];
        if (dof_is_linear(dof)) {
            for (int r = 0; r < 3; r++) jacp[r][dof] = ax[r];
        } else {
            double rel[3], c[3];
            sub3(rel, point, d->dof_anchor[dof]); cross3(c, ax, rel);
            for (int r = 0; r < 3; r++) { jacr[r][dof] = ax[r]; jacp[r][dof] = c[r]; }
        }
    }
}

static void world_inertia(const OrcModel *m, const OrcData *d, int b, double Iw[9]) {
    double Ri[9], Rl[9], T[9];
    quat_to_mat(Rl, m->body_iquat[b]);
    mulmm3(Ri, d->xmat[b], Rl);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) T[3*i+j] = Ri[3*i+j] * m->body_inertia[b][j];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        Iw[3*i+j] = T[3*i]*Ri[3*j] + T[3*i+1]*Ri[3*j+1] + T[3*i+2]*Ri[3*j+2];
}

static void mass_matrix(const OrcModel *m, OrcData *d) {
    double jp[3][NV], jr[3][NV], Iw[9];
    for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) d->M[i][j] = (i == j) ? m->dof_armature[i] : 0.0;
    for (int b = 1; b < NB; b++) {
        orc_jac_body(m, d, b, d->xipos[b], jp, jr);
        world_inertia(m, d, b, Iw);
        for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) {
            double s = 0;
            for (int r = 0; r < 3; r++) s += m->body_mass[b] * jp[r][i] * jp[r][j];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) s += jr[r][i] * Iw[3*r+c] * jr[c][j];
            d->M[i][j] += s;
        }
    }
}

/* ------------------------------------------------------------------ collision */
static void hull_info(const OrcModel *m, int g, const double **verts, int *n) {
    *verts = m->hull_verts + 3 * m->hull_vadr[g - 1]; *n = m->hull_vnum[g - 1];
}

/* support point of hull geom g in world direction dir (unit), inflated by `inflate` */
static int hull_support(const OrcModel *m, OrcData *d, int g, const double dir[3], double inflate, double out[3]) {
    const double *v; int n; hull_info(m, g, &v, &n);
    int b = m->geom_body[g];
    double dl[3]; mulmtv3(dl, d->xmat[b], dir);
    int best = 0; double bv = -1e300;
    for (int i = 0; i < n; i++) {
        double s = dot3(v + 3*i, dl);
        if (s > bv) { bv = s; best = i; }
    }
    double w[3]; mulmv3(w, d->xmat[b], v + 3*best);
    for (int k = 0; k < 3; k++) out[k] = d->xpos[b][k] + w[k] + dir[k] * inflate;
    d->support_calls++;
    return best;
}

static void geom_center_world(const OrcModel *m, const OrcData *d, int g, double c[3]) {
    int b = m->geom_body[g]; double t[3];
    mulmv3(t, d->xmat[b], m->geom_center[g]); add3(c, d->xpos[b], t);
}

/* tangent basis as mju_makeFrame [3P-recall] */
static void make_frame(double frame[9]) {
    double *n = frame, *t1 = frame + 3, *t2 = frame + 6;
    if (fabs(n[1]) < 0.5) { t1[0] = 0; t1[1] = 1; t1[2] = 0; } else { t1[0] = 0; t1[1] = 0; t1[2] = 1; }
    double s = dot3(n, t1); addscl3(t1, t1, n, -s); normalize3(t1);
    cross3(t2, n, t1);
}

static void mix_friction(const OrcModel *m, int g1, int g2, double f[3]) {
    for (int k = 0; k < 3; k++) f[k] = fmax(m->geom_friction[g1][k], m->geom_friction[g2][k]);
}

/* floor (z = 0, normal +z) against hull geom g: the deepest vertex plus up to three of
 * its hull-graph neighbours that are also within the margin [3P-recall: mjc_PlaneConvex]. */
int orc_plane_hull(const OrcModel *m, const OrcData *d, int g, OrcContact *out) {
    int b = m->geom_body[g];
    double c[3]; geom_center_world(m, d, g, c);
    if (c[2] > m->geom_rbound[g] + m->margin) return 0;         /* bounding-sphere cull */
    const double *v; int n; hull_info(m, g, &v, &n);
    const double *R = d->xmat[b];
    int best = 0; double hmin = 1e300;
    for (int i = 0; i < n; i++) {
        double h = d->xpos[b][2] + R[6]*v[3*i] + R[7]*v[3*i+1] + R[8]*v[3*i+2];
        if (h < hmin) { hmin = h; best = i; }
    }
    if (hmin > m->margin) return 0;
    int cand[4], nc = 0; cand[nc++] = best;
    int base = m->hull_vadr[g - 1];
    for (int e = m->hull_nadr[base + best]; e < m->hull_nadr[base + best + 1] && nc < 4; e++) {
        int j = m->hull_nbr[e];
        double h = d->xpos[b][2] + R[6]*v[3*j] + R[7]*v[3*j+1] + R[8]*v[3*j+2];
        if (h <= m->margin) cand[nc++] = j;
    }
    for (int k = 0; k < nc; k++) {
        double w[3]; mulmv3(w, R, v + 3*cand[k]); add3(w, w, d->xpos[b]);
        OrcContact *o = out + k;
        memset(o, 0, sizeof *o);
        o->g1 = G_FLOOR; o->g2 = g; o->dist = w[2];
        o->pos[0] = w[0]; o->pos[1] = w[1]; o->pos[2] = 0.5 * w[2];
        o->frame[0] = 0; o->frame[1] = 0; o->frame[2] = 1; make_frame(o->frame);
        mix_friction(m, G_FLOOR, g, o->friction);
    }
    return nc;
}

/* ---- Minkowski Portal Refinement (XenoCollide; what libccd's ccdMPRPenetration, the
 * routine MuJoCo 2.2 calls for mesh-mesh pairs, implements) [3P-recall]. Each hull is
 * inflated by margin/2 so that proximity inside the margin reads as penetration. ---- */
typedef struct { double v[3], v1[3], v2[3]; } SupPt;

static void mpr_support(const OrcModel *m, OrcData *d, int g1, int g2, const double dir[3], double infl, SupPt *s) {
    double nd[3] = {-dir[0], -dir[1], -dir[2]};
    hull_support(m, d, g1, dir, infl, s->v1);
    hull_support(m, d, g2, nd, infl, s->v2);
    sub3(s->v, s->v1, s->v2);
}

static void portal_dir(const SupPt *p1, const SupPt *p2, const SupPt *p3, double dir[3]) {
    double a[3], b[3]; sub3(a, p2->v, p1->v); sub3(b, p3->v, p1->v); cross3(dir, a, b); normalize3(dir);
}

static void expand_portal(SupPt *p0, SupPt *p1, SupPt *p2, SupPt *p3, const SupPt *p4) {
    double v4v0[3]; cross3(v4v0, p4->v, p0->v);
    if (dot3(p1->v, v4v0) > 0) {
        if (dot3(p2->v, v4v0) > 0) *p1 = *p4; else *p3 = *p4;
    } else {
        if (dot3(p3->v, v4v0) > 0) *p2 = *p4; else *p1 = *p4;
    }
}

static int reach_tolerance(const SupPt *p1, const SupPt *p2, const SupPt *p3, const SupPt *p4, const double dir[3], double tol) {
    double d4 = dot3(p4->v, dir);
    double m1 = d4 - dot3(p1->v, dir), m2 = d4 - dot3(p2->v, dir), m3 = d4 - dot3(p3->v, dir);
    double mn = fmin(m1, fmin(m2, m3));
    return mn <= tol;
}

/* squared distance from the origin to triangle (a,b,c); witness = closest point */
static double origin_tri_dist2(const double a[3], const double b[3], const double c[3], double witness[3]) {
    double ab[3], ac[3], ap[3]; sub3(ab, b, a); sub3(ac, c, a); scl3(ap, a, -1);
    double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0 && d2 <= 0) { copy3(witness, a); return dot3(a, a); }
    double bp[3]; scl3(bp, b, -1);
    double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0 && d4 <= d3) { copy3(witness, b); return dot3(b, b); }
    double vc = d1*d4 - d3*d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) { double t = d1 / (d1 - d3); addscl3(witness, a, ab, t); return dot3(witness, witness); }
    double cp[3]; scl3(cp, c, -1);
    double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0 && d5 <= d6) { copy3(witness, c); return dot3(c, c); }
    double vb = d5*d2 - d1*d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { double t = d2 / (d2 - d6); addscl3(witness, a, ac, t); return dot3(witness, witness); }
    double va = d3*d6 - d5*d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        double t = (d4 - d3) / ((d4 - d3) + (d5 - d6)); double bc[3]; sub3(bc, c, b); addscl3(witness, b, bc, t);
        return dot3(witness, witness);
    }
    double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
    for (int k = 0; k < 3; k++) witness[k] = a[k] + ab[k]*v + ac[k]*w;
    return dot3(witness, witness);
}

static void find_pos(const SupPt *p0, const SupPt *p1, const SupPt *p2, const SupPt *p3, double pos[3]) {
    double dir[3], t[3], b[4];
    portal_dir(p1, p2, p3, dir);
    cross3(t, p1->v, p2->v); b[0] = dot3(t, p3->v);
    cross3(t, p3->v, p2->v); b[1] = dot3(t, p0->v);
    cross3(t, p0->v, p1->v); b[2] = dot3(t, p3->v);
    cross3(t, p2->v, p1->v); b[3] = dot3(t, p0->v);
    double sum = b[0] + b[1] + b[2] + b[3];
    if (sum <= 0) {
        b[0] = 0;
        cross3(t, p2->v, p3->v); b[1] = dot3(t, dir);
        cross3(t, p3->v, p1->v); b[2] = dot3(t, dir);
        cross3(t, p1->v, p2->v); b[3] = dot3(t, dir);
        sum = b[1] + b[2] + b[3];
    }
    double inv = 1.0 / sum;
    const SupPt *pp[4] = {p0, p1, p2, p3};
    for (int k = 0; k < 3; k++) {
        double s1 = 0, s2 = 0;
        for (int i = 0; i < 4; i++) { s1 += b[i] * pp[i]->v1[k]; s2 += b[i] * pp[i]->v2[k]; }
        pos[k] = 0.5 * (s1 + s2) * inv;
    }
}

#define MPR_TOL 1e-6
#define MPR_ITER 50
#define MPR_EPS 1e-14

/* returns 1 and fills depth/dir/pos when the inflated hulls intersect */
static int mpr_penetration(const OrcModel *m, OrcData *d, int g1, int g2, double infl, double *depth, double dir_out[3], double pos[3]) {
    SupPt p0, p1, p2, p3, p4;
    double dir[3], va[3], vb[3], t[3];
    d->mpr_calls++;
    geom_center_world(m, d, g1, p0.v1); geom_center_world(m, d, g2, p0.v2); sub3(p0.v, p0.v1, p0.v2);
    if (dot3(p0.v, p0.v) < MPR_EPS) p0.v[0] += 1e-5;
    /* --- discover portal --- */
    scl3(dir, p0.v, -1); normalize3(dir);
    mpr_support(m, d, g1, g2, dir, infl, &p1);
    if (dot3(p1.v, dir) <= 0) return 0;
    cross3(dir, p0.v, p1.v);
    if (dot3(dir, dir) < MPR_EPS) {
        /* origin on the segment v0-v1 (or at v1) */
        if (dot3(p1.v, p1.v) < MPR_EPS) { *depth = 0; dir_out[0] = dir_out[1] = dir_out[2] = 0; }
        else { *depth = norm3(p1.v); copy3(dir_out, p1.v); normalize3(dir_out); }
        for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p1.v1[k] + p1.v2[k]);
        return 1;
    }
    normalize3(dir);
    mpr_support(m, d, g1, g2, dir, infl, &p2);
    if (dot3(p2.v, dir) <= 0) return 0;
    sub3(va, p1.v, p0.v); sub3(vb, p2.v, p0.v); cross3(dir, va, vb); normalize3(dir);
    if (dot3(dir, p0.v) > 0) { SupPt s = p1; p1 = p2; p2 = s; scl3(dir, dir, -1); }
    for (int it = 0;; it++) {
        if (it > 4 * MPR_ITER) return 0;
        mpr_support(m, d, g1, g2, dir, infl, &p3);
        if (dot3(p3.v, dir) <= 0) return 0;
        int cont = 0;
        cross3(t, p1.v, p3.v);
        if (dot3(t, p0.v) < -MPR_EPS) { p2 = p3; cont = 1; }
        if (!cont) {
            cross3(t, p3.v, p2.v);
            if (dot3(t, p0.v) < -MPR_EPS) { p1 = p3; cont = 1; }
        }
        if (!cont) break;
        sub3(va, p1.v, p0.v); sub3(vb, p2.v, p0.v); cross3(dir, va, vb); normalize3(dir);
    }
    /* --- refine portal until the origin is enclosed --- */
    for (int it = 0;; it++) {
        portal_dir(&p1, &p2, &p3, dir);
        if (dot3(dir, p1.v) >= -MPR_EPS) break;                /* portal encapsules origin */
        mpr_support(m, d, g1, g2, dir, infl, &p4);
        if (dot3(p4.v, dir) < 0 || reach_tolerance(&p1, &p2, &p3, &p4, dir, MPR_TOL) || it > MPR_ITER) return 0;
        expand_portal(&p0, &p1, &p2, &p3, &p4);
    }
    /* --- penetration: push the portal to the surface --- */
    for (int it = 0;; it++) {
        portal_dir(&p1, &p2, &p3, dir);
        mpr_support(m, d, g1, g2, dir, infl, &p4);
        if (reach_tolerance(&p1, &p2, &p3, &p4, dir, MPR_TOL) || it > MPR_ITER) {
            double w[3];
            double d2 = origin_tri_dist2(p1.v, p2.v, p3.v, w);
            *depth = sqrt(d2);
            if (*depth < MPR_EPS) { dir_out[0] = dir_out[1] = dir_out[2] = 0; }
            else { copy3(dir_out, w); normalize3(dir_out); }
            find_pos(&p0, &p1, &p2, &p3, pos);
            return 1;
        }
        expand_portal(&p0, &p1, &p2, &p3, &p4);
    }
}

int orc_hull_hull(const OrcModel *m, OrcData *d, int g1, int g2, OrcContact *out) {
    double c1[3], c2[3], dc[3];
    geom_center_world(m, d, g1, c1); geom_center_world(m, d, g2, c2); sub3(dc, c2, c1);
    double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + m->margin;
    if (dot3(dc, dc) > bound * bound) return 0;                 /* bounding-sphere cull */
    double depth, dir[3], pos[3];
    if (!mpr_penetration(m, d, g1, g2, 0.5 * m->margin, &depth, dir, pos)) return 0;
    double dist = m->margin - depth;
    if (dist >= m->margin) return 0;
    memset(out, 0, sizeof *out);
    out->g1 = g1; out->g2 = g2; out->dist = dist;
    copy3(out->pos, pos);
    /* Minkowski difference is geom1 - geom2; the closest boundary point w to the origin is the
     * translation of geom2 that separates the pair, i.e. w/|w| points from geom1 to geom2 --
     * libccd's `dir`, which MuJoCo uses as the contact normal. */
    copy3(out->frame, dir);
    if (dot3(out->frame, out->frame) < 0.5) { copy3(out->frame, dc); normalize3(out->frame); }
    make_frame(out->frame);
    mix_friction(m, g1, g2, out->friction);
    return 1;
}

static void collision(const OrcModel *m, OrcData *d) {
    d->ncon = 0;
    for (int g = 1; g < NG; g++) {
        if (d->ncon + 4 > ORC_MAXCON) break;
        d->ncon += orc_plane_hull(m, d, g, d->con + d->ncon);
    }
    for (int k = 0; k < m->npair; k++) {
        if (d->ncon + 1 > ORC_MAXCON) break;
        d->ncon += orc_hull_hull(m, d, m->hull_pairs[2*k], m->hull_pairs[2*k+1], d->con + d->ncon);
    }
}

void orc_fwd_position(const OrcModel *m, OrcData *d) {
    kinematics(m, d);
    mass_matrix(m, d);
    collision(m, d);
}

/* ------------------------------------------------------------------ dense helpers */
static int cholesky(double A[NV][NV], int n) {      /* in place, lower */
    for (int j = 0; j < n; j++) {
        double s = A[j][j];
        for (int k = 0; k < j; k++) s -= A[j][k] * A[j][k];
        if (s < MINVAL) s = MINVAL;
        A[j][j] = sqrt(s);
        for (int i = j + 1; i < n; i++) {
            double t = A[i][j];
            for (int k = 0; k < j; k++) t -= A[i][k] * A[j][k];
            A[i][j] = t / A[j][j];
        }
    }
    return 0;
}
static void chol_solve(double L[NV][NV], int n, double x[NV]) {
    for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i][k] * x[k]; x[i] = s / L[i][i]; }
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k]; x[i] = s / L[i][i]; }
}

/* ------------------------------------------------------------------ velocity / bias */
static void bias_forces(const OrcModel *m, OrcData *d) {
    double w[NB][3], v[NB][3], al[NB][3], a[NB][3];
    memset(w, 0, sizeof w); memset(v, 0, sizeof v); memset(al, 0, sizeof al); memset(a, 0, sizeof a);
    double t[3], u[3], r[3];
    /* ee: slides (static parent: no bias), then roll, then yaw */
    for (int i = 0; i < 3; i++) addscl3(v[B_EE], v[B_EE], d->dof_axis[i], d->qvel[i]);
    for (int j = 3; j <= 4; j++) {
        scl3(u, d->dof_axis[j], d->qvel[j]);
        cross3(t, w[B_EE], u); add3(al[B_EE], al[B_EE], t);
        add3(w[B_EE], w[B_EE], u);
    }
    const int order[5] = {B_BASE, B_LK, B_LF, B_RK, B_RF};
    const int hinge[5] = {-1, 5, -1, 6, -1};
    for (int k = 0; k < 5; k++) {
        int b = order[k], p = m->body_parent[b];
        sub3(r, d->xpos[b], d->xpos[p]);
        cross3(t, w[p], r); add3(v[b], v[p], t);
        cross3(u, w[p], t);                      /* w x (w x r) */
        cross3(t, al[p], r); add3(a[b], a[p], t); add3(a[b], a[b], u);
        copy3(w[b], w[p]); copy3(al[b], al[p]);
        if (hinge[k] >= 0) {
            scl3(u, d->dof_axis[hinge[k]], d->qvel[hinge[k]]);
            cross3(t, w[b], u); add3(al[b], al[b], t);
            add3(w[b], w[b], u);
        }
    }
    copy3(v[B_OBJ], d->qvel + 7);
    mulmv3(w[B_OBJ], d->xmat[B_OBJ], d->qvel + 10);
    for (int i = 0; i < NV; i++) d->qfrc_bias[i] = 0;
    double jp[3][NV], jr[3][NV], Iw[9];
    for (int b = 1; b < NB; b++) {
        double rc[3], ac[3], f[3], tau[3], Iwv[3];
        sub3(rc, d->xipos[b], d->xpos[b]);
        cross3(t, w[b], rc); cross3(u, w[b], t);
        cross3(t, al[b], rc); add3(ac, a[b], t); add3(ac, ac, u);
        ac[2] -= m->gravity_z;
        scl3(f, ac, m->body_mass[b]);
        world_inertia(m, d, b, Iw);
        mulmv3(tau, Iw, al[b]); mulmv3(Iwv, Iw, w[b]); cross3(t, w[b], Iwv); add3(tau, tau, t);
        orc_jac_body(m, d, b, d->xipos[b], jp, jr);
        for (int i = 0; i < NV; i++)
            for (int k = 0; k < 3; k++) d->qfrc_bias[i] += jp[k][i] * f[k] + jr[k][i] * tau[k];
    }
}

/* ------------------------------------------------------------------ constraints */
static void sol_kb(const double solref[2], const double solimp[5], double timestep, double *k, double *b) {
    double tc = fmax(solref[0], 2 * timestep), dr = solref[1], dmax = solimp[1];   /* refsafe */
    *k = 1.0 / fmax(MINVAL, dmax * dmax * tc * tc * dr * dr);
    *b = 2.0 / fmax(MINVAL, dmax * tc);
}

static double impedance(const double si[5], double pos, double margin) {
    double dmin = si[0], dmax = si[1], width = si[2], mid = si[3], power = si[4];
    double x = fabs(pos - margin) / fmax(MINVAL, width);
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    double y;
    if (power <= 1 + 1e-12) y = x;
    else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
    else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
}

static void make_constraints(const OrcModel *m, OrcData *d) {
    int n = 0;
    double k, b;
    /* joint limits on the 7 gripper joints (slides/hinges: dof index = joint index) */
    sol_kb(m->lim_solref, m->lim_solimp, m->timestep, &k, &b);
    for (int j = 0; j < NU; j++) {
        for (int side = 0; side < 2; side++) {
            double dist = side == 0 ? d->qpos[j] - m->jnt_range[j][0] : m->jnt_range[j][1] - d->qpos[j];
            if (dist >= 0) continue;
            memset(d->efc_J[n], 0, sizeof d->efc_J[n]);
            d->efc_J[n][j] = side == 0 ? 1.0 : -1.0;
            d->efc_type[n] = 0; d->efc_id[n] = j; d->efc_pos[n] = dist; d->efc_margin[n] = 0;
            double imp = impedance(m->lim_solimp, dist, 0);
            d->efc_R[n] = fmax(MINVAL, (1 - imp) * m->dof_invweight0[j] / imp);
            d->efc_vel[n] = d->efc_J[n][j] * d->qvel[j];
            d->efc_aref[n] = -b * d->efc_vel[n] - k * imp * dist;
            n++;
        }
    }
    sol_kb(m->solref, m->solimp, m->timestep, &k, &b);
    double jp1[3][NV], jr1[3][NV], jp2[3][NV], jr2[3][NV];
    for (int c = 0; c < d->ncon; c++) {
        OrcContact *con = d->con + c;
        con->efc_adr = -1;
        if (con->dist >= m->margin) continue;
        int b1 = m->geom_body[con->g1], b2 = m->geom_body[con->g2];
        orc_jac_body(m, d, b1, con->pos, jp1, jr1);
        orc_jac_body(m, d, b2, con->pos, jp2, jr2);
        con->efc_adr = n;
        for (int r = 0; r < 4; r++) {
            const double *ax = con->frame + 3 * (r < 3 ? r : 0);
            for (int i = 0; i < NV; i++) {
                double s = 0;
                if (r < 3) for (int q = 0; q < 3; q++) s += ax[q] * (jp2[q][i] - jp1[q][i]);
                else for (int q = 0; q < 3; q++) s += ax[q] * (jr2[q][i] - jr1[q][i]);
                d->efc_J[n + r][i] = s;
            }
            d->efc_type[n + r] = r == 0 ? 1 : 2; d->efc_id[n + r] = c;
            d->efc_pos[n + r] = r == 0 ? con->dist : 0; d->efc_margin[n + r] = r == 0 ? m->margin : 0;
            double vel = 0; for (int i = 0; i < NV; i++) vel += d->efc_J[n + r][i] * d->qvel[i];
            d->efc_vel[n + r] = vel;
        }
        double imp = impedance(m->solimp, con->dist, m->margin);
        double tran = m->body_invweight0[b1][0] + m->body_invweight0[b2][0];
        double R0 = fmax(MINVAL, (1 - imp) * tran / imp);
        double R1 = R0 / fmax(MINVAL, m->impratio);
        double fs = con->friction[0], ft = con->friction[1];
        d->efc_R[n] = R0; d->efc_R[n + 1] = R1; d->efc_R[n + 2] = R1;
        d->efc_R[n + 3] = R1 * fs * fs / fmax(MINVAL, ft * ft);
        con->mu = fs * sqrt(R1 / R0);
        d->efc_aref[n] = -b * d->efc_vel[n] - k * imp * (con->dist - m->margin);
        for (int r = 1; r < 4; r++) d->efc_aref[n + r] = -b * d->efc_vel[n + r];
        n += 4;
    }
    d->nefc = n;
    for (int i = 0; i < n; i++) d->efc_D[i] = 1.0 / d->efc_R[i];
}

/* cost, gradient (= -force) and Hessian of one elliptic contact in jar space.
 * s(jar) = (D0 / 2 mu^2) * dist^2(U, K),  U = diag(mu, f1, f1, f3) jar,  K = {U0 >= mu |U_t|}. */
static double cone_eval(const double jar[4], const double D[4], double mu, const double fr[3],
                        double grad[4], double H[16]) {
    double S[4] = {mu, fr[0], fr[0], fr[1]};
    double U[4]; for (int i = 0; i < 4; i++) U[i] = S[i] * jar[i];
    double N = U[0], T = sqrt(U[1]*U[1] + U[2]*U[2] + U[3]*U[3]);
    for (int i = 0; i < 4; i++) grad[i] = 0;
    if (H) for (int i = 0; i < 16; i++) H[i] = 0;
    if (N >= mu * T || (T <= 0 && N >= 0)) return 0;                       /* top zone */
    if (mu * N + T <= 0 || (T <= 0 && N < 0)) {                            /* bottom zone */
        double c = 0;
        for (int i = 0; i < 4; i++) { c += 0.5 * D[i] * jar[i] * jar[i]; grad[i] = D[i] * jar[i]; if (H) H[5*i] = D[i]; }
        return c;
    }
    double kap = D[0] / fmax(MINVAL, mu * mu), s1 = 1.0 / sqrt(1 + mu * mu);
    double dist = (mu * T - N) * s1;
    double nU[4] = {-s1, mu * s1 * U[1] / T, mu * s1 * U[2] / T, mu * s1 * U[3] / T};
    for (int i = 0; i < 4; i++) grad[i] = kap * S[i] * dist * nU[i];
    if (H) {
        double c2 = dist * mu * s1 / T;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) {
            double h = nU[i] * nU[j];
            if (i > 0 && j > 0) h += c2 * ((i == j ? 1.0 : 0.0) - (U[i] / T) * (U[j] / T));
            H[4*i+j] = kap * S[i] * S[j] * h;
        }
    }
    return 0.5 * kap * dist * dist;
}

typedef struct { double cost, gauss; } CostOut;

/* evaluates constraint cost at jar; fills force = -grad and, if Hout, H = M + J^T s'' J */
static double constraint_cost(const OrcData *d, const double jar[], double force[], double Hout[NV][NV]) {
    double cost = 0;
    for (int i = 0; i < d->nefc;) {
        if (d->efc_type[i] == 0) {
            if (jar[i] < 0) {
                cost += 0.5 * d->efc_D[i] * jar[i] * jar[i]; force[i] = -d->efc_D[i] * jar[i];
                if (Hout) for (int a = 0; a < NV; a++) for (int b = 0; b < NV; b++)
                    Hout[a][b] += d->efc_D[i] * d->efc_J[i][a] * d->efc_J[i][b];
            } else force[i] = 0;
            i++;
        } else {
            const OrcContact *con = d->con + d->efc_id[i];
            double g[4], H[16];
            cost += cone_eval(jar + i, d->efc_D + i, con->mu, con->friction, g, Hout ? H : NULL);
            for (int r = 0; r < 4; r++) force[i + r] = -g[r];
            if (Hout) {
                for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) {
                    double h = H[4*r+c]; if (h == 0) continue;
                    for (int a = 0; a < NV; a++) { double ja = h * d->efc_J[i + r][a]; if (ja == 0) continue;
                        for (int b = 0; b < NV; b++) Hout[a][b] += ja * d->efc_J[i + c][b]; }
                }
            }
            i += 4;
        }
    }
    return cost;
}

static double total_cost(const OrcData *d, const double qacc[NV], double jar[], double force[], double *gauss_out) {
    double dq[NV], gauss = 0;
    for (int i = 0; i < NV; i++) dq[i] = qacc[i] - d->qacc_smooth[i];
    for (int i = 0; i < NV; i++) { double s = 0; for (int j = 0; j < NV; j++) s += d->M[i][j] * dq[j]; gauss += 0.5 * s * dq[i]; }
    for (int i = 0; i < d->nefc; i++) { double s = -d->efc_aref[i]; for (int j = 0; j < NV; j++) s += d->efc_J[i][j] * qacc[j]; jar[i] = s; }
    if (gauss_out) *gauss_out = gauss;
    return gauss + constraint_cost(d, jar, force, NULL);
}

/* 1-D derivative pair of phi(alpha) = cost(qacc + alpha * p) */
static void line_eval(const OrcData *d, const double jar[], const double jv[], double alpha,
                      double g0, double g1, double *dphi, double *ddphi) {
    double dp = g0 + alpha * g1, hp = g1;       /* Gauss part: g0 = p^T M (a - a_s), g1 = p^T M p */
    double ja[4];
    for (int i = 0; i < d->nefc;) {
        if (d->efc_type[i] == 0) {
            double x = jar[i] + alpha * jv[i];
            if (x < 0) { dp += d->efc_D[i] * x * jv[i]; hp += d->efc_D[i] * jv[i] * jv[i]; }
            i++;
        } else {
            const OrcContact *con = d->con + d->efc_id[i];
            double g[4], H[16];
            for (int r = 0; r < 4; r++) ja[r] = jar[i + r] + alpha * jv[i + r];
            cone_eval(ja, d->efc_D + i, con->mu, con->friction, g, H);
            for (int r = 0; r < 4; r++) { dp += g[r] * jv[i + r]; for (int c = 0; c < 4; c++) hp += jv[i + r] * H[4*r+c] * jv[i + c]; }
            i += 4;
        }
    }
    *dphi = dp; *ddphi = hp;
}

static void solve_newton(const OrcModel *m, OrcData *d) {
    double jar[ORC_MAXEFC], force[ORC_MAXEFC], jv[ORC_MAXEFC], jar2[ORC_MAXEFC], force2[ORC_MAXEFC];
    double qacc[NV], grad[NV], p[NV], Mp[NV], H[NV][NV];
    const double scale = 1.0 / (m->meaninertia * NV);
    /* warm start: the better of qacc_warmstart and qacc_smooth [3P-recall] */
    double c_w = total_cost(d, d->qacc_warmstart, jar, force, NULL);
    double c_s = total_cost(d, d->qacc_smooth, jar2, force2, NULL);
    if (c_w < c_s) memcpy(qacc, d->qacc_warmstart, sizeof qacc); else memcpy(qacc, d->qacc_smooth, sizeof qacc);
    double cost = total_cost(d, qacc, jar, force, NULL);
    d->solver_iter = 0;
    for (int it = 0; it < m->iterations; it++) {
        /* gradient and Hessian */
        for (int i = 0; i < NV; i++) {
            double s = 0; for (int j = 0; j < NV; j++) s += d->M[i][j] * (qacc[j] - d->qacc_smooth[j]);
            for (int r = 0; r < d->nefc; r++) s -= d->efc_J[r][i] * force[r];
            grad[i] = s;
        }
        double gn = 0; for (int i = 0; i < NV; i++) gn += grad[i] * grad[i];
        if (scale * sqrt(gn) < m->tolerance) break;
        memcpy(H, d->M, sizeof H);
        constraint_cost(d, jar, force2, H);
        cholesky(H, NV);
        for (int i = 0; i < NV; i++) p[i] = -grad[i];
        chol_solve(H, NV, p);
        /* exact line search: safeguarded 1-D Newton on phi'(alpha) */
        double g0 = 0, g1 = 0;
        for (int i = 0; i < NV; i++) { double s = 0; for (int j = 0; j < NV; j++) s += d->M[i][j] * p[j]; Mp[i] = s; }
        for (int i = 0; i < NV; i++) { g0 += Mp[i] * (qacc[i] - d->qacc_smooth[i]); g1 += Mp[i] * p[i]; }
        for (int r = 0; r < d->nefc; r++) { double s = 0; for (int j = 0; j < NV; j++) s += d->efc_J[r][j] * p[j]; jv[r] = s; }
        double lo = 0, hi = -1, dlo, dhi = 0, alpha = 0, dp, hp;
        line_eval(d, jar, jv, 0, g0, g1, &dp, &hp); dlo = dp;
        if (dp >= 0) break;                                        /* not a descent direction: converged */
        double gtol = 1e-12 * fabs(dlo) + 1e-300;
        for (int ls = 0; ls < 50; ls++) {
            double step = -dp / fmax(hp, MINVAL);
            double an = alpha + step;
            if (hi > 0 && (an <= lo || an >= hi)) an = 0.5 * (lo + hi);
            alpha = an;
            line_eval(d, jar, jv, alpha, g0, g1, &dp, &hp);
            if (fabs(dp) < gtol) break;
            if (dp < 0) { lo = alpha; dlo = dp; } else { hi = alpha; dhi = dp; }
            if (hi > 0 && (hi - lo) < 1e-14 * hi) break;
        }
        (void)dlo; (void)dhi;
        for (int i = 0; i < NV; i++) qacc[i] += alpha * p[i];
        double newcost = total_cost(d, qacc, jar, force, NULL);
        d->solver_iter = it + 1;
        double improvement = scale * (cost - newcost);
        cost = newcost;
        if (improvement < m->tolerance) break;
    }
    memcpy(d->qacc, qacc, sizeof qacc);
    for (int i = 0; i < d->nefc; i++) d->efc_force[i] = force[i];
    for (int i = 0; i < NV; i++) { double s = 0; for (int r = 0; r < d->nefc; r++) s += d->efc_J[r][i] * force[r]; d->qfrc_constraint[i] = s; }
}

/* ------------------------------------------------------------------ forward / step */
static void fwd_velocity_to_acc(const OrcModel *m, OrcData *d) {
    bias_forces(m, d);
    for (int i = 0; i < NV; i++) d->qfrc_passive[i] = -m->dof_damping[i] * d->qvel[i];
    /* actuation: motors, ctrl clamped to ctrlrange, force = gear * ctrl */
    for (int i = 0; i < NV; i++) d->qfrc_actuator[i] = 0;
    for (int u = 0; u < NU; u++) {
        double c = fmin(fmax(d->ctrl[u], m->ctrlrange[u][0]), m->ctrlrange[u][1]);
        d->qfrc_actuator[u] = m->gear[u] * c;
    }
    /* xfrc_applied: force/torque at the body COM, mapped through the body Jacobian */
    double jp[3][NV], jr[3][NV];
    for (int i = 0; i < NV; i++) d->qfrc_applied[i] = 0;
    for (int b = 1; b < NB; b++) {
        int any = 0; for (int k = 0; k < 6; k++) any |= d->xfrc[b][k] != 0;
        if (!any) continue;
        orc_jac_body(m, d, b, d->xipos[b], jp, jr);
        for (int i = 0; i < NV; i++) for (int k = 0; k < 3; k++)
            d->qfrc_applied[i] += jp[k][i] * d->xfrc[b][k] + jr[k][i] * d->xfrc[b][3 + k];
    }
    double L[NV][NV];
    memcpy(L, d->M, sizeof L); cholesky(L, NV);
    for (int i = 0; i < NV; i++) {
        d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
        d->qacc_smooth[i] = d->qfrc_smooth[i];
    }
    chol_solve(L, NV, d->qacc_smooth);
    make_constraints(m, d);
    if (d->nefc == 0) {
        memcpy(d->qacc, d->qacc_smooth, sizeof d->qacc);
        for (int i = 0; i < NV; i++) d->qfrc_constraint[i] = 0;
        d->solver_iter = 0;
    } else {
        solve_newton(m, d);
    }
}

void orc_forward(const OrcModel *m, OrcData *d) {
    orc_fwd_position(m, d);
    fwd_velocity_to_acc(m, d);
}

static void euler_advance(const OrcModel *m, OrcData *d) {
    const double h = m->timestep;
    double qacc[NV], L[NV][NV];
    /* implicit joint damping: (M + h D) qacc' = qfrc_smooth + qfrc_constraint  [3P-recall: mj_Euler] */
    memcpy(L, d->M, sizeof L);
    for (int i = 0; i < NV; i++) { L[i][i] += h * m->dof_damping[i]; qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    cholesky(L, NV); chol_solve(L, NV, qacc);
    memcpy(d->qacc_warmstart, d->qacc, sizeof d->qacc);
    for (int i = 0; i < NV; i++) d->qvel[i] += h * qacc[i];
    for (int i = 0; i < 10; i++) d->qpos[i] += h * d->qvel[i];
    /* free-joint quaternion: q <- q * exp(h w), w in the body frame */
    double w[3] = {d->qvel[10], d->qvel[11], d->qvel[12]};
    double ang = norm3(w) * h;
    if (ang > 0) {
        double ax[3] = {w[0], w[1], w[2]}, dq[4], qn[4];
        normalize3(ax); axis_quat(dq, ax, ang);
        quat_mul(qn, d->qpos + 10, dq); quat_normalize(qn);
        memcpy(d->qpos + 10, qn, sizeof qn);
    }
    d->time += h;
}

/* Conditioning probe (tests / fixture generation only; off by default). An fp32 implementation injects ~1e-7 relative rounding
 * noise into the state at EVERY physics.step(); whether a macro step's integer outputs survive that is a property of the state, not of
 * the implementation. With amp > 0 every orc_step() ends by perturbing qpos / qvel by amp * (1 + |x|) * u, u uniform in [-1, 1] from a
 * counter hash of (seed, step time, index): deterministic and thread safe. */
static double g_step_noise_amp = 0.0, g_step_noise_vamp = 0.0;
static unsigned g_step_noise_seed = 0;
void orc_set_step_noise(double amp, unsigned seed) { g_step_noise_amp = amp; g_step_noise_vamp = amp; g_step_noise_seed = seed; }
/* separate amplitudes for positions and velocities: an fp32 solver's acceleration error (Newton stopped at the fp32 noise floor of
 * its gradient) shows up in qvel first */
void orc_set_step_noise2(double amp_qpos, double amp_qvel, unsigned seed) { g_step_noise_amp = amp_qpos; g_step_noise_vamp = amp_qvel; g_step_noise_seed = seed; }
static double noise_u(unsigned a, unsigned b, unsigned c) {
    unsigned long long x = ((unsigned long long)a << 32) ^ ((unsigned long long)b * 0x9E3779B97F4A7C15ULL) ^ ((unsigned long long)c << 17);
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (double)(x >> 11) / 4503599627370496.0 - 1.0;      /* 53 bits -> [-1, 1) */
}

void orc_step(const OrcModel *m, OrcData *d) {
    /* mj_step2 on the state whose position stage is already current, then mj_step1 */
    fwd_velocity_to_acc(m, d);
    euler_advance(m, d);
    if (g_step_noise_amp > 0.0 || g_step_noise_vamp > 0.0) {
        unsigned t = (unsigned)(d->time / m->timestep + 0.5);
        for (int i = 0; i < ORC_NQ; i++) d->qpos[i] += g_step_noise_amp * (1.0 + fabs(d->qpos[i])) * noise_u(g_step_noise_seed, t, (unsigned)i);
        for (int i = 0; i < ORC_NV; i++) d->qvel[i] += g_step_noise_vamp * (1.0 + fabs(d->qvel[i])) * noise_u(g_step_noise_seed, t, 100u + (unsigned)i);
    }
    orc_fwd_position(m, d);
}

void orc_reset_data(const OrcModel *m, OrcData *d) {
    memset(d, 0, sizeof *d);
    memcpy(d->qpos, m->qpos0, sizeof d->qpos);
    orc_fwd_position(m, d);
}
