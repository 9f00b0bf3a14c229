// grip_sim.hip -- kernels and C ABI (include/grip_sim.h) of the MI355X batched rollout engine.
//
// Data layout in HBM: every per-env quantity is a structure of arrays, [field][env] fp32
// (qpos 14 x N, qvel 13 x N, ctrl 7 x N, qacc_warmstart 13 x N, flags), so that the 64 lanes of a
// wave load 64 consecutive floats per field: one 256-byte coalesced row per field per wave.
// One macro step (RobotEnv.step, robot_env.py:77-241) is ONE kernel launch: each lane loads its
// state once (188 B), runs the controller, the whole MOVE / RETURN / OPEN / CLOSE state machine
// of physics.step() calls with the state in registers and the contact list in LDS, and writes
// the state back once (160 B) together with reward / done / info.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/grip_sim.h"
#include "grip_physics.h"

// ------------------------------------------------------------------------------------------------
// error handling
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
extern "C" const char *grip_last_error(void) { return g_err.c_str(); }
static int fail(const std::string &s) { g_err = s; return -1; }
int grip_fail(const char *msg) { return fail(msg); }      // for the other translation units of the library
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------------------------------------
// host model
// ------------------------------------------------------------------------------------------------
struct Blob {
    struct Entry { std::string name; unsigned dtype, ndim, dims[4]; size_t off, count; };
    std::vector<unsigned char> buf; std::vector<Entry> entries;
    bool load(const char *path) {
        FILE *f = fopen(path, "rb"); if (!f) return false;
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        if (n < 12) { fclose(f); return false; }
        buf.resize(n); bool ok = fread(buf.data(), 1, n, f) == (size_t)n; fclose(f);
        if (!ok || memcmp(buf.data(), "GRPM", 4) != 0) return false;
        unsigned cnt; memcpy(&cnt, buf.data() + 8, 4);
        size_t off = 12;
        for (unsigned i = 0; i < cnt; i++) {
            if (off + 48 > buf.size()) return false;
            Entry e; char nm[25] = {0}; memcpy(nm, buf.data() + off, 24); e.name = nm;
            memcpy(&e.dtype, buf.data() + off + 24, 4); memcpy(&e.ndim, buf.data() + off + 28, 4); memcpy(e.dims, buf.data() + off + 32, 16);
            if (e.ndim > 4 || e.dtype > 1) return false;
            off += 48; e.off = off; e.count = 1; for (unsigned k = 0; k < e.ndim; k++) e.count *= e.dims[k];
            size_t nb = e.count * (e.dtype == 0 ? 8 : 4);
            if (nb > buf.size() - off) return false;                     // truncated or corrupt entry
            off += nb + ((8 - nb % 8) % 8);
            entries.push_back(e);
        }
        return true;
    }
    const Entry *find(const char *name) const { for (auto &e : entries) if (e.name == name) return &e; return nullptr; }
    bool f64(const char *name, std::vector<double> &out, size_t expect = 0) const {
        const Entry *e = find(name); if (!e || e->dtype != 0 || (expect && e->count != expect)) return false;
        out.resize(e->count); memcpy(out.data(), buf.data() + e->off, e->count * 8); return true;
    }
    bool i32(const char *name, std::vector<int> &out, size_t expect = 0) const {
        const Entry *e = find(name); if (!e || e->dtype != 1 || (expect && e->count != expect)) return false;
        out.resize(e->count); memcpy(out.data(), buf.data() + e->off, e->count * 4); return true;
    }
};

struct GripModel {
    DevModel host;                       // pointer members are filled per device at batch creation
    std::vector<unsigned> hull_blob;     // vertices | nadr | nbr | lut (see DevModel)
    std::vector<float> planes;           // [nplane][4]
    std::vector<int> loops;              // face polygons: CSR offsets [nplane + 1] | corner vertex ids (see DevModel)
    int nvert = 0;
};

namespace hm {   // tiny double-precision helpers for the host-side composite-body build
struct D3 { double x, y, z; };
static void qmat(const double *q, double R[9]) {
    double n = std::sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    double w = q[0]/n, x = q[1]/n, y = q[2]/n, z = q[3]/n;
    R[0] = 1-2*(y*y+z*z); R[1] = 2*(x*y-w*z); R[2] = 2*(x*z+w*y);
    R[3] = 2*(x*y+w*z); R[4] = 1-2*(x*x+z*z); R[5] = 2*(y*z-w*x);
    R[6] = 2*(x*z-w*y); R[7] = 2*(y*z+w*x); R[8] = 1-2*(x*x+y*y);
}
static void mm(const double A[9], const double B[9], double C[9]) {
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3*i+j] = A[3*i]*B[j] + A[3*i+1]*B[3+j] + A[3*i+2]*B[6+j];
    memcpy(C, t, sizeof t);
}
static void mv(const double A[9], const double v[3], double r[3]) {
    double t[3] = {A[0]*v[0]+A[1]*v[1]+A[2]*v[2], A[3]*v[0]+A[4]*v[1]+A[5]*v[2], A[6]*v[0]+A[7]*v[1]+A[8]*v[2]};
    memcpy(r, t, sizeof t);
}
// inertia of body (iquat, diag) expressed in the frame rotated by Rb: Rb Ri diag Ri^T Rb^T
static void body_inertia(const double Rb[9], const double *iquat, const double *diag, double I[9]) {
    double Ri[9], R[9]; qmat(iquat, Ri); mm(Rb, Ri, R);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
        double s = 0; for (int k = 0; k < 3; k++) s += R[3*i+k] * diag[k] * R[3*j+k];
        I[3*i+j] = s;
    }
}
}  // namespace hm

// composite of bodies welded together: parts given as (mass, com in group frame, inertia about own com in group frame)
static void composite(int n, const double *mass, const double (*com)[3], const double (*I)[9], float &m_out, float *com_out, float *I_out) {
    double M = 0, c[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) { M += mass[i]; for (int k = 0; k < 3; k++) c[k] += mass[i] * com[i][k]; }
    for (int k = 0; k < 3; k++) c[k] /= M;
    double T[9] = {0};
    for (int i = 0; i < n; i++) {
        double d[3] = {com[i][0] - c[0], com[i][1] - c[1], com[i][2] - c[2]};
        double d2 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) T[3*a+b] += I[i][3*a+b] + mass[i] * ((a == b ? d2 : 0.0) - d[a] * d[b]);
    }
    m_out = (float)M; for (int k = 0; k < 3; k++) com_out[k] = (float)c[k];
    I_out[0] = (float)T[0]; I_out[1] = (float)T[4]; I_out[2] = (float)T[8]; I_out[3] = (float)T[1]; I_out[4] = (float)T[2]; I_out[5] = (float)T[5];
}

extern "C" int grip_model_load(const char *blob_path, GripModel **out) {
    Blob b;
    if (!b.load(blob_path)) return fail(std::string("cannot read model blob ") + blob_path);
    std::vector<double> opt, margin, solref, solimp, lsolref, lsolimp, bpos, bquat, bmass, bipos, biquat, binert, arm, damp, rng, gear,
        crange, qpos0, gfric, gcen, grb, grgba, hverts, hplanes, biw, diw, mi, campos, camquat, camfovy, visual, frgb, srgb, ldir, lpos, lpar, gmat, hlight;
    std::vector<int> gbody, hvadr, hvnum, hpadr, hpnum, nadr, nbr, pairs, lut, ldirl;
    bool ok = b.f64("opt", opt, 5) && b.f64("geom_margin", margin, 1) && b.f64("geom_solref", solref, 2) && b.f64("geom_solimp", solimp, 5) &&
              b.f64("lim_solref", lsolref, 2) && b.f64("lim_solimp", lsolimp, 5) && b.f64("body_pos", bpos, 24) && b.f64("body_quat", bquat, 32) &&
              b.f64("body_mass", bmass, 8) && b.f64("body_ipos", bipos, 24) && b.f64("body_iquat", biquat, 32) && b.f64("body_inertia", binert, 24) &&
              b.f64("dof_armature", arm, 13) && b.f64("dof_damping", damp, 13) && b.f64("jnt_range", rng, 14) && b.f64("gear", gear, 7) &&
              b.f64("ctrlrange", crange, 14) && b.f64("qpos0", qpos0, 14) && b.f64("geom_friction", gfric, 21) && b.f64("geom_center", gcen, 21) &&
              b.f64("geom_rbound", grb, 7) && b.f64("geom_rgba", grgba, 28) && b.f64("hull_verts", hverts) && b.f64("hull_planes", hplanes) &&
              b.f64("body_invweight0", biw, 16) && b.f64("dof_invweight0", diw, 13) && b.f64("meaninertia", mi, 1) && b.f64("cam_pos", campos, 3) &&
              b.f64("cam_quat", camquat, 4) && b.f64("cam_fovy", camfovy, 1) && b.f64("visual", visual, 3) && b.f64("floor_rgb", frgb, 6) &&
              b.f64("sky_rgb", srgb, 6) && b.f64("light_dir", ldir, 6) && b.i32("geom_body", gbody, 7) && b.i32("hull_vadr", hvadr, 6) &&
              b.i32("hull_vnum", hvnum, 6) && b.i32("hull_padr", hpadr, 6) && b.i32("hull_pnum", hpnum, 6) && b.i32("hull_nadr", nadr) &&
              b.i32("hull_nbr", nbr) && b.i32("hull_pairs", pairs) && b.i32("hull_lut", lut, 6 * 6 * 64);
    if (!ok) return fail(std::string("model blob ") + blob_path + " is missing fields or has the wrong version");
    GripModel *gm = new GripModel();
    DevModel &m = gm->host; memset(&m, 0, sizeof m);
    m.timestep = (float)opt[0]; m.gravity_z = (float)opt[1]; m.impratio = (float)opt[2]; m.iterations = (int)opt[3]; m.tolerance = (float)opt[4];
    m.margin = (float)margin[0];
    auto kb = [&](const std::vector<double> &sr, const std::vector<double> &si, float &k, float &bb) {
        double tc = std::max(sr[0], 2.0 * opt[0]), dr = sr[1], dmax = si[1];
        k = (float)(1.0 / (dmax * dmax * tc * tc * dr * dr)); bb = (float)(2.0 / (dmax * tc)); };
    kb(solref, solimp, m.k_con, m.b_con); kb(lsolref, lsolimp, m.k_lim, m.b_lim);
    for (int i = 0; i < 5; i++) { m.solimp[i] = (float)solimp[i]; m.lim_solimp[i] = (float)lsolimp[i]; }
    m.meaninertia = (float)mi[0];
    // ---- composite rigid groups (double), bodies: 1 ee, 2 base, 3 lk, 4 lf, 5 rk, 6 rf, 7 object
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    auto part = [&](int body, const double Rb[9], const double pb[3], double &mass, double com[3], double I[9]) {
        mass = bmass[body]; double t[3]; hm::mv(Rb, &bipos[3 * body], t);
        for (int k = 0; k < 3; k++) com[k] = pb[k] + t[k];
        hm::body_inertia(Rb, &biquat[4 * body], &binert[3 * body], I); };
    {   // G = ee (frame) + base
        double mass[2], com[2][3], I[2][9], Rb[9], zero[3] = {0, 0, 0};
        part(1, I3, zero, mass[0], com[0], I[0]);
        hm::qmat(&bquat[4 * 2], Rb); part(2, Rb, &bpos[3 * 2], mass[1], com[1], I[1]);
        composite(2, mass, com, I, m.grp_mass[0], m.grp_com[0], m.grp_inertia[0]);
        for (int k = 0; k < 3; k++) m.base_pos[k] = (float)bpos[6 + k];
        for (int k = 0; k < 9; k++) m.base_R[k] = (float)Rb[k];
    }
    for (int s = 0; s < 2; s++) {   // L / R = knuckle (frame) + finger
        int kbody = 3 + 2 * s, fbody = 4 + 2 * s;
        double mass[2], com[2][3], I[2][9], Rf[9], Rk[9], zero[3] = {0, 0, 0};
        part(kbody, I3, zero, mass[0], com[0], I[0]);
        hm::qmat(&bquat[4 * fbody], Rf); part(fbody, Rf, &bpos[3 * fbody], mass[1], com[1], I[1]);
        composite(2, mass, com, I, m.grp_mass[1 + s], m.grp_com[1 + s], m.grp_inertia[1 + s]);
        hm::qmat(&bquat[4 * kbody], Rk);
        for (int k = 0; k < 3; k++) { m.kn_pos[s][k] = (float)bpos[3 * kbody + k]; m.fin_pos[s][k] = (float)bpos[3 * fbody + k]; }
        for (int k = 0; k < 9; k++) { m.kn_R[s][k] = (float)Rk[k]; m.fin_R[s][k] = (float)Rf[k]; }
    }
    {   double mass[1], com[1][3], I[1][9], zero[3] = {0, 0, 0};
        part(7, I3, zero, mass[0], com[0], I[0]);
        composite(1, mass, com, I, m.grp_mass[3], m.grp_com[3], m.grp_inertia[3]); }
    {   double Re[9]; hm::qmat(&bquat[4], Re);
        for (int k = 0; k < 9; k++) if (std::fabs(Re[k] - I3[k]) > 1e-12) { delete gm; return fail("ee body quat must be identity"); }
        for (int k = 0; k < 3; k++) m.ee_pos0[k] = (float)bpos[3 + k]; }
    for (int i = 0; i < 13; i++) { m.armature[i] = (float)arm[i]; m.damping[i] = (float)damp[i]; }
    // the object hangs on a <freejoint/>, which takes no joint defaults: no damping there (integrate() relies on it: M a = qfrc_smooth + J^T f
    // is then what qacc already solves for the object block)
    for (int i = 7; i < 13; i++) if (damp[i] != 0.0) { delete gm; return fail("the object's free joint must be undamped"); }
    for (int j = 0; j < 7; j++) {
        m.range[j][0] = (float)rng[2*j]; m.range[j][1] = (float)rng[2*j+1]; m.gear[j] = (float)gear[j];
        m.ctrlrange[j][0] = (float)crange[2*j]; m.ctrlrange[j][1] = (float)crange[2*j+1]; m.dof_invweight0[j] = (float)diw[j];
    }
    for (int i = 0; i < 14; i++) m.qpos0[i] = (float)qpos0[i];
    const int body_group[8] = {GRP_WORLD, GRP_G, GRP_G, GRP_L, GRP_L, GRP_R, GRP_R, GRP_O};
    for (int g = 0; g < 7; g++) {
        for (int k = 0; k < 3; k++) m.geom_center[g][k] = (float)gcen[3*g+k];
        m.geom_rbound[g] = (float)grb[g];
        m.geom_friction[g][0] = (float)gfric[3*g]; m.geom_friction[g][1] = (float)gfric[3*g+1];
        m.geom_invweight[g] = (float)biw[2 * gbody[g]];
        m.geom_group[g] = body_group[gbody[g]];
        for (int k = 0; k < 4; k++) m.geom_rgba[g][k] = (float)grgba[4*g+k];
    }
    for (int h = 0; h < 6; h++) { m.hull_vadr[h] = hvadr[h]; m.hull_vnum[h] = hvnum[h]; m.hull_padr[h] = hpadr[h]; m.hull_pnum[h] = hpnum[h]; }
    for (int h = 0; h < 6; h++) {
        if (hvadr[h] < 0 || hvnum[h] <= 0 || (size_t)(hvadr[h] + hvnum[h]) * 3 > hverts.size()) { delete gm; return fail("hull vertex ranges do not fit hull_verts"); }
        for (int k = 0; k < 3; k++) { m.hull_aabb[h][k] = 3.0e38f; m.hull_aabb[h][3 + k] = -3.0e38f; }
        for (int v = hvadr[h]; v < hvadr[h] + hvnum[h]; v++)
            for (int k = 0; k < 3; k++) {
                m.hull_aabb[h][k] = std::min(m.hull_aabb[h][k], (float)hverts[3 * v + k]);
                m.hull_aabb[h][3 + k] = std::max(m.hull_aabb[h][3 + k], (float)hverts[3 * v + k]);
            }
    }
    m.npair = (int)pairs.size() / 2;
    if (m.npair > GN_PAIR_MAX) { delete gm; return fail("too many collision pairs"); }
    for (int q = 0; q < m.npair; q++) { m.pairs[q][0] = pairs[2*q]; m.pairs[q][1] = pairs[2*q+1]; }
    // narrow-phase items over an env's 16 lanes: lanes 0..10 the hull pairs, lanes 11..15 the floor tests of geoms 2..6,
    // lane 11 also takes the (almost always sphere-culled) floor test of the base in a second round
    if (m.npair > 11) { delete gm; return fail("the 16-lane item map holds at most 11 hull pairs"); }
    for (int l = 0; l < 16; l++) { m.coop_items[l][0] = -1; m.coop_items[l][1] = -1; }
    for (int q = 0; q < m.npair; q++) m.coop_items[q][0] = 6 + q;
    for (int g = 2; g <= 6; g++) m.coop_items[11 + (g - 2)][0] = g - 1;
    m.coop_items[11][1] = 0;
    gm->nvert = (int)hverts.size() / 3;
    {   // pack the hull tables into one word blob
        auto pack16 = [](std::vector<unsigned> &dst, const std::vector<int> &src) {
            size_t w0 = dst.size(); dst.resize(w0 + (src.size() + 1) / 2, 0u);
            unsigned short *p = reinterpret_cast<unsigned short *>(dst.data() + w0);
            for (size_t i = 0; i < src.size(); i++) p[i] = (unsigned short)src[i];
        };
        for (int v : nadr) if (v < 0 || v > 65535) { delete gm; return fail("hull graph too large for 16-bit offsets"); }
        std::vector<unsigned> &hb = gm->hull_blob;
        hb.resize(4 * (size_t)gm->nvert);
        float *vf = reinterpret_cast<float *>(hb.data());
        if ((int)nadr.size() != gm->nvert + 1) { delete gm; return fail("hull_nadr must hold one CSR offset per hull vertex plus the end"); }
        // the 4th word of a vertex carries its own adjacency range (CSR offset | degree << 16): a hill climb that moves to a neighbour
        // has that neighbour's range from the vertex read it already made -- no dependent trip through the offset table
        for (int i = 0; i < gm->nvert; i++) {
            for (int k = 0; k < 3; k++) vf[4*i+k] = (float)hverts[3*i+k];
            const int deg = nadr[i + 1] - nadr[i];
            if (deg < 0 || deg > 65535) { delete gm; return fail("hull vertex degree does not fit 16 bits"); }
            const unsigned packed = (unsigned)nadr[i] | ((unsigned)deg << 16);
            memcpy(&vf[4*i+3], &packed, 4);
        }
        m.hull_off_nadr = 0;                             // (the CSR offsets travel in the vertex records: the table itself is not staged)
        m.hull_off_nbr = (int)hb.size(); pack16(hb, nbr);
        m.hull_off_lut = (int)hb.size(); pack16(hb, lut);
        m.hull_words = (int)hb.size();
    }
    gm->planes.resize(hplanes.size()); for (size_t i = 0; i < hplanes.size(); i++) gm->planes[i] = (float)hplanes[i];
    for (int k = 0; k < 3; k++) m.cam_pos[k] = (float)campos[k];
    {   double Rc[9]; hm::qmat(camquat.data(), Rc); for (int k = 0; k < 9; k++) m.cam_R[k] = (float)Rc[k]; }
    m.cam_fovy = (float)camfovy[0]; m.znear = (float)visual[1]; m.zfar = (float)visual[2];
    for (int k = 0; k < 6; k++) { m.floor_rgb[k] = (float)frgb[k]; m.sky_rgb[k] = (float)srgb[k]; m.light_dir[k / 3][k % 3] = (float)ldir[k]; }
    // materials / lights (round 4): blobs compiled before them lack the arrays
    if (!(b.f64("geom_material", gmat, 21) && b.f64("light_params", lpar, 10) && b.f64("headlight", hlight, 3) && b.f64("light_pos", lpos, 6) && b.i32("light_directional", ldirl, 2))) {
        delete gm; return fail(std::string("model blob lacks the material / light arrays: recompile it with model/compiler.py (") + blob_path + ")");
    }
    for (int g = 0; g < GN_GEOM; g++) for (int k = 0; k < 3; k++) m.geom_material[g][k] = (float)gmat[3 * g + k];
    for (int l = 0; l < 2; l++) {
        for (int k = 0; k < 3; k++) m.light_pos[l][k] = (float)lpos[3 * l + k];
        m.light_directional[l] = ldirl[l];
        m.light_params[l][0] = (float)lpar[5 * l]; m.light_params[l][1] = (float)lpar[5 * l + 1]; m.light_params[l][2] = (float)lpar[5 * l + 2];
        m.light_params[l][3] = (float)cos(lpar[5 * l + 3] * 0.017453292519943295); m.light_params[l][4] = (float)lpar[5 * l + 4];
    }
    for (int k = 0; k < 3; k++) m.headlight[k] = (float)hlight[k];
    {   // face polygons (round 4, second half): the observation kernel rasterises the hulls' faces
        std::vector<int> ladr, lids;
        const size_t np = gm->planes.size() / 4;
        if (!(b.i32("hull_ladr", ladr, np + 1) && b.i32("hull_loops", lids))) {
            delete gm; return fail(std::string("model blob lacks the face polygons (hull_ladr / hull_loops): recompile it with model/compiler.py (") + blob_path + ")");
        }
        bool ok = ladr[0] == 0 && (size_t)ladr[np] == lids.size();
        for (size_t j = 0; ok && j < np; j++) ok = ladr[j + 1] >= ladr[j] && (ladr[j + 1] == ladr[j] || ladr[j + 1] - ladr[j] >= 3);
        for (int h = 0; ok && h < 6; h++)
            for (int j = hpadr[h]; ok && (ok = hpadr[h] >= 0 && (size_t)(hpadr[h] + hpnum[h]) <= np) && j < hpadr[h] + hpnum[h]; j++)
                for (int k = ladr[j]; ok && k < ladr[j + 1]; k++) ok = lids[k] >= 0 && lids[k] < hvnum[h];
        if (!ok) { delete gm; return fail("hull_ladr / hull_loops are not face polygons over the hulls' vertices"); }
        gm->loops = ladr; gm->loops.insert(gm->loops.end(), lids.begin(), lids.end());
    }
    *out = gm;
    return 0;
}
extern "C" void grip_model_free(GripModel *m) { delete m; }
extern "C" int grip_model_nvert(const GripModel *m) { return m ? m->nvert : -1; }

// ------------------------------------------------------------------------------------------------
// batch
// ------------------------------------------------------------------------------------------------
struct StepOutDev {     // device copy of GripStepOut (kernel argument)
    float *reward; uint8_t *done; float *achieved_goal; float *desired_goal; int *status; int *episode_step; int *gripper_open;
    int *object_grasped; int *position_reached; float *total_distance; float *line_distance; float *gripper_position;
    float *object_position; float *init_obj_pos; int *n_substeps; int *fault;
};

struct GripBatch {
    int n = 0, device = 0;
    DevModel hmodel; unsigned *d_hull = nullptr; float *d_planes = nullptr; int *d_loops = nullptr; size_t lds_bytes = 0;
    DevConfig cfg;
    float *qpos = nullptr, *qvel = nullptr, *ctrl = nullptr, *warm = nullptr;     // SoA [field][N]
    int *episode_step = nullptr, *status = nullptr, *gripper_open = nullptr;
    int *pad_grasp = nullptr, *pad_pher = nullptr;                              // sensor-pad scalars of the current state
    int nplanes = 0, nverts = 0;
    int *mc_ints = nullptr, *mc_astate = nullptr, *mc_slot = nullptr, *mc_order = nullptr, *mc_heavy = nullptr, *mc_tick = nullptr, *mc_gen = nullptr; unsigned long long *mc_t0 = nullptr; float *mc_flts = nullptr, *mc_memo = nullptr;   // suspended macro steps
    float *reset_info = nullptr;                                                // grasp0, pher0, objx0, objy0 of the reset state
    float *scratch = nullptr; size_t scratch_bytes = 0;
    float xfrc_z = 0.f;
    std::vector<hipEvent_t> ev0, ev1; int ev_used = 0;
    bool reset_info_valid = false;
    void *d_self = nullptr, *d_rself = nullptr;    // this batch as a set of one: GroupArgs / RenderGroup records in device memory
};

static constexpr int EV_RING = 1024;
static constexpr int LDS_MAX_BYTES = 160 * 1024;

// ---- state load / store (SoA; the 16 lanes of an env read the same words)
struct StatePtrs { float *qpos, *qvel, *ctrl, *warm; int *episode_step, *status, *gripper_open, *pad_grasp, *pad_pher; int n; int half; };

// A macro step suspended between two time slices (grip_batch_advance): everything k_macro_step keeps in registers across
// its physics.step() loop, one record per env (MCI / MCF below: 32 + 64 bytes, so a scattered env of the cost-sorted work order
// touches whole sectors of its own instead of one 4-byte word in each of 21 sectors). astate, slot, heavy, gen stay [N] arrays: k_compact
// scans them over all envs. astate: 0 = macro step in flight, 1 = finished, waiting for an action;
// slot: row of the compact action / observation arrays this waiting env was given by the last k_compact (-1 = none yet);
// heavy: hull-hull contacts the env had at its last physics.step() -- the cost class k_compact sorts the work order by;
// gen: number of the compaction that gave the slot (tick[0] counts compactions): a slot-holder starts `lag` launches later.
struct MacroCtx { int *ints; float *flts; int *astate; int *slot; int *heavy; int *tick; int *gen; unsigned long long *t0; float *memo; };
// per lane: word 0 = has | has_sep << 1 | (h1 + 1) << 2 | (h2 + 1) << 14 (portal flag, separating-direction flag, the two support-vertex
// hints), 3 vertex-pair ids, 3 query directions (the portal memory), the separating direction: EVERYTHING collide() remembers between
// calls, so that a resumed macro step takes the same branches -- per-lane or cooperative support evaluation included -- as the
// uninterrupted one and time-sliced results equal lock-step results bit for bit in contact too (tests/test_gpu_contact.py)
#define MC_MEMO_WORDS 16
// memory layout of the pair memory: [env][block of 4 words][lane][4]: block 0 = {word 0, separating direction} of the 16 lanes (256
// contiguous bytes per env, one 16-byte access per lane), blocks 1..3 = the portal (vertex-pair ids, query directions) of the lanes in contact
#define MEMO4(mc, e, blk, sub) (reinterpret_cast<float4 *>((mc).memo) + (((size_t)(e) * 4 + (blk)) * 16 + (sub)))
#define MC_NINT_PAD 8
#define MC_NFLT_PAD 16
#define MCI(mc, f, e) (mc).ints[(size_t)(e) * MC_NINT_PAD + (f)]
#define MCF(mc, f, e) (mc).flts[(size_t)(e) * MC_NFLT_PAD + (f)]
enum { MC_PHASE = 0, MC_CNT, MC_NSUB, MC_GRASPED, MC_FLAGS, MC_FAULT, MC_NINT };
static_assert(MC_CNT == MI_CNT && MC_NSUB == MI_NSUB && MC_GRASPED == MI_GRASPED && MC_FLAGS == MI_FLAGS && MC_FAULT == MI_FAULT, "suspended record = layout of ES_MACI");
enum { MC_TARGET = 0, MC_INITQ = 5, MC_OPENCLOSE = 10, MC_TQ = 11, MC_INITOBJ = 12, MC_NFLT = 15 };
static_assert(MC_NINT <= MC_NINT_PAD && MC_NFLT <= MC_NFLT_PAD, "suspended-context record too small");

// global state arrays <-> the env's LDS state vectors: lane j moves component j (one load / store instruction per array)
DEVI void ld_state(const StatePtrs &p, int e, const Ctx &cx) {
    float *S = cx.envl; const int j = cx.sub;
    if (j < 14) S[ES_QPOS + j] = ld_word(p.qpos, (size_t)j * p.n + e, p.half);
    if (j < 13) S[ES_QVEL + j] = ld_word(p.qvel, (size_t)j * p.n + e, p.half);
    if (j < 7) S[ES_CTRL + j] = ld_word(p.ctrl, (size_t)j * p.n + e, p.half);
    if (j < 13) S[ES_WARM + j] = p.warm[(size_t)j * p.n + e];
    wave_sync();
}
// `doit`: this lane's env is stored (all 16 lanes of an env agree)
DEVI void st_state(const StatePtrs &p, int e, const Ctx &cx, bool doit) {
    wave_sync();
    const float *S = cx.envl; const int j = cx.sub;
    if (doit) {
        if (j < 14) st_word(p.qpos, (size_t)j * p.n + e, S[ES_QPOS + j], p.half);
        if (j < 13) st_word(p.qvel, (size_t)j * p.n + e, S[ES_QVEL + j], p.half);
        if (j < 7) st_word(p.ctrl, (size_t)j * p.n + e, S[ES_CTRL + j], p.half);
        if (j < 13) p.warm[(size_t)j * p.n + e] = S[ES_WARM + j];
    }
}
// the reset state into the env's LDS vectors (lane 0 of the env)
DEVI void reset_state(const DevModel &m, const Ctx &cx) {
    float *S = cx.envl;
    float q[14], z13[13] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, z7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 14; i++) q[i] = m.qpos0[i];
    lds_st<14>(S + ES_QPOS, q); lds_st<13>(S + ES_QVEL, z13); lds_st<7>(S + ES_CTRL, z7); lds_st<13>(S + ES_WARM, z13);
}

// utils.py:30-31
DEVI float project_dir(float x, float y, const DevConfig &c) {
    float n = sqrtf(c.dir_x * c.dir_x + c.dir_y * c.dir_y);
    return (x * c.dir_x + y * c.dir_y) / (n * n);
}
// actuator.py:198-215
DEVI int pheromone_level(V3 ee, const DevConfig &c) {
    float p = project_dir(ee.x, ee.y, c);
    float dx = p * c.dir_x - ee.x, dy = p * c.dir_y - ee.y;
    float conc = 1.0f / expf(sqrtf(dx * dx + dy * dy));
    return conc > 0.82f ? 3 : conc > 0.6f ? 2 : conc > 0.37f ? 1 : 0;
}

// transformations.py:1035-1090 ('sxyz' non-repeating branch) on a row-major 3x3
DEVI void euler_from_matrix(const M3 &M, float &ax, float &ay, float &az) {
    float cy = sqrtf(M.m[0] * M.m[0] + M.m[3] * M.m[3]);
    if (cy > 4.0f * 1.1920929e-7f) { ax = atan2f(M.m[7], M.m[8]); ay = atan2f(-M.m[6], cy); az = atan2f(M.m[3], M.m[0]); }
    else { ax = atan2f(-M.m[5], M.m[4]); ay = atan2f(-M.m[6], cy); az = 0.f; }
}
// transformations.py:972-1032 ('sxyz')
DEVI M3 euler_matrix(float ai, float aj, float ak) {
    float si = sinf(ai), sj = sinf(aj), sk = sinf(ak), ci = cosf(ai), cj = cosf(aj), ck = cosf(ak);
    float cc = ci * ck, cs = ci * sk, sc = si * ck, ss = si * sk;
    M3 M;
    M.m[0] = cj * ck; M.m[1] = sj * sc - cs; M.m[2] = sj * cc + ss;
    M.m[3] = cj * sk; M.m[4] = sj * ss + cc; M.m[5] = sj * cs - sc;
    M.m[6] = -sj; M.m[7] = cj * si; M.m[8] = cj * ci;
    return M;
}

// Actuator.get_target_pose (actuator.py:58-102) with _normalise_action (:21-44), _clip_translation_vector
// (:249-264), _enforce_constraints (:266-293); the 6x5 Jacobian pseudo-inverse in closed form (its columns
// are orthonormal: three world axes, the roll axis x and the yaw axis Rx(roll) z).
DEVI void target_pose(const DevConfig &c, const float *act, const float (&qpos)[14], const Kin &k, float (&target)[5], float &open_close) {
    float a[6]; int q = 0;
#pragma unroll
    for (int i = 0; i < 6; i++) a[i] = (!c.include_roll && i == 3) ? 0.f : act[q++];
    open_close = a[5];
    float st = 2.0f / (2.0f * c.max_translation), sr = 2.0f / (2.0f * c.max_rotation);
    float t0 = a[0] / st, t1 = a[1] / st, t2 = a[2] / st, r0 = a[3] / sr, r1 = a[4] / sr;
    float len = sqrtf(t0 * t0 + t1 * t1 + t2 * t2);
    if (len > c.max_translation) { float f = c.max_translation / len; t0 *= f; t1 *= f; t2 *= f; }
    r0 = fminf(fmaxf(r0, -c.max_rotation), c.max_rotation); r1 = fminf(fmaxf(r1, -c.max_rotation), c.max_rotation);
    float roll, pitch, yaw;
    euler_from_matrix(k.Re, roll, pitch, yaw);
    M3 Rold = euler_matrix(roll, pitch, yaw), Rrel = euler_matrix(r0, 0.f, r1);
    M3 Rnew = mulm(Rold, Rrel);
    V3 pos = k.pe + mulv(Rold, v3(t0, t1, t2));
    float o0, o1, o2; euler_from_matrix(Rnew, o0, o1, o2);
    const float PI4 = 0.78539816339744830962f;
    if (!c.include_roll) o0 = 0.f; else o0 = fminf(fmaxf(o0, -PI4), PI4);
    o1 = 0.f;
    pos.z = fminf(fmaxf(pos.z, 0.1f), 0.5f);
    V3 ep = pos - k.pe, eo = v3(o0 - roll, o1 - pitch, o2 - yaw);
    target[0] = qpos[0] + ep.x; target[1] = qpos[1] + ep.y; target[2] = qpos[2] + ep.z;
    target[3] = qpos[3] + eo.x; target[4] = qpos[4] + dot(k.a4, eo);
}

// reward.py:18-41
DEVI float agent_reward(V3 o0, V3 o1, const DevConfig &c, int gripper_open, float c5, float c6, int grasped, float &line_distance) {
    float p0 = project_dir(o0.x, o0.y, c), p1 = project_dir(o1.x, o1.y, c);
    float dx = p1 * c.dir_x - o1.x, dy = p1 * c.dir_y - o1.y;
    float lat = sqrtf(dx * dx + dy * dy), travel = p1 - p0, r = 0.f;
    bool ok = travel > 0.f && travel < 0.1f && lat < 0.1f;
    line_distance = ok ? travel : 0.f;
    if (ok) {
        r = travel;
        if (!gripper_open && (c5 != 0.f && c6 != 0.f) && grasped == 3) { r *= 2.f; if (o1.z > 0.f) r *= 1.5f; }
    }
    return r * 30.f;
}

enum { PH_MOVE = 0, PH_RETURN, PH_OPEN, PH_CLOSE, PH_FINAL, PH_DONE };
#define CP_CLASSES 16       // work-order classes of k_compact: cost 0..14 of running envs, 15 = idle

// ------------------------------------------------------------------------------------------------
// kernels: 512-thread workgroups = 16 environments x (16 cooperating lanes + 16 clone lanes) (grip_physics.h)
// ------------------------------------------------------------------------------------------------
extern __shared__ float lds_dyn[];
#ifndef GRIP_COLD_PORTAL
#define PAIRMEMO_EXTRA_INIT(m) (m).has = 0;
#else
#define PAIRMEMO_EXTRA_INIT(m)
#endif

#define STAMPS_DECL Stamps stm; STAMPS_INIT
#ifdef GRIP_STAMPS
#define STAMPS_INIT for (int i_ = 0; i_ < NSTAMP; i_++) stm.acc[i_] = 0; stm.t = stamp_now();
#else
#define STAMPS_INIT
#endif

__global__ void __launch_bounds__(WG_THREADS, WG_WAVES_PER_SIMD) k_reset(const DevModel m, DevConfig cfg, StatePtrs st, const uint8_t *mask, StepOutDev out, float *reset_info, MacroCtx mc) {
    const Ctx cx = stage_tables(m, lds_dyn);
    STAMPS_DECL
    bool act_; int e = blockIdx.x * EPB + wg_env_slot(act_);
    bool valid = act_ && e < st.n;
    if (e >= st.n) e = st.n - 1;          // (idle rows of a 2-env wave mirror the env of the row 32 lanes below: same reads, no writes)
    bool doit = valid && (mask == nullptr || mask[e] != 0);
    env_lds_init(cx);
    if (cx.sub == 0) reset_state(m, cx);
    wave_sync();
    Kin k; Contact con; int ncon = 0;
    PairMemo sep; sep.sep = v3(0, 0, 0); sep.h1 = sep.h2 = -1; PAIRMEMO_EXTRA_INIT(sep)
    float q0[14];
    forward_kin(m, cx, q0, k);
    ncon = collide(m, cx, con, sep, stm);
    const int fault = env_fault(cx);
    int grasp = check_grasp(cx, con, ncon), pher = pheromone_level(k.pe, cfg);
    if (blockIdx.x == 0 && threadIdx.x == 0 && reset_info) { reset_info[0] = (float)grasp; reset_info[1] = (float)pher; reset_info[2] = k.po.x; reset_info[3] = k.po.y; }
    st_state(st, e, cx, doit);
    if (!doit || cx.sub != 0) return;
    st.episode_step[e] = 0; st.status[e] = 0; st.gripper_open[e] = 1;
    st.pad_grasp[e] = grasp; st.pad_pher[e] = pher;
    mc.astate[e] = 1; mc.slot[e] = -1;                      // any macro step in flight is dropped
    if (out.achieved_goal) { out.achieved_goal[2 * e] = k.po.x; out.achieved_goal[2 * e + 1] = k.po.y; }
    if (out.desired_goal) { out.desired_goal[2 * e] = cfg.dir_x; out.desired_goal[2 * e + 1] = cfg.dir_y; }
    if (out.object_position) { out.object_position[3 * e] = k.po.x; out.object_position[3 * e + 1] = k.po.y; out.object_position[3 * e + 2] = k.po.z; }
    if (out.gripper_position) { out.gripper_position[3 * e] = k.pe.x; out.gripper_position[3 * e + 1] = k.pe.y; out.gripper_position[3 * e + 2] = k.pe.z; }
    if (out.status) out.status[e] = 0;
    if (out.episode_step) out.episode_step[e] = 0;
    if (out.gripper_open) out.gripper_open[e] = 1;
    if (out.fault) out.fault[e] = fault;
}

// RobotEnv.step (robot_env.py:77-241): 2 envs (+ their clone lanes) per wave, 16 lanes per env
//
// slice <= 0: lock-step -- every env starts a macro step with actions[e] and the launch returns when the slowest is done.
// slice  > 0: time slice of grip_batch_advance -- an env in flight resumes from its MacroCtx, a waiting env that holds a
// slot starts a macro step with actions[slot], both run at most `slice` calls of physics.step(); envs that finish write
// their outputs and wait, the others are suspended. The arithmetic per env is the same in both modes.
// `block`: index of the workgroup within its batch (blockIdx.x for a single batch; a set of batches shares one launch)
DEVI void macro_step_body(const DevModel &m, const DevConfig &cfg, const StatePtrs &st, const float *actions, const StepOutDev &out,
                          const float *reset_info, float xfrc_z, const MacroCtx &mc, int slice, const int *order, long long budget_ticks, int lag, int block) {
    const Ctx cx = stage_tables(m, lds_dyn);
    STAMPS_DECL
    bool act_; int e = block * EPB + wg_env_slot(act_);
    // Lanes 32..63 of a two-env wave are CLONES of lanes 0..31: the same env, the same arithmetic, the same LDS traffic (identical values to
    // identical addresses), no global writes (`valid` is false there). collide(), make_constraints() and the solver hand them the second
    // half of work that one instruction stream can do for two data sets (grip_physics.h, halves_*).
    const bool inb = e < st.n;                          // the env exists (clones included)
    const bool valid = act_ && inb;                     // ... and this lane may write its results
    if (e >= st.n) e = st.n - 1;
    if (order) e = order[e];
    const bool writer = valid && cx.sub == 0;
    const bool sliced = slice > 0;
    float *S = cx.envl;
    env_lds_init(cx);
    ld_state(st, e, cx);
    const int adim = cfg.include_roll ? 6 : 5;

    // What the macro step carries lives in the env's LDS region (ES_MAC floats, ES_MACI integers); registers keep what the loop turns on:
    // phase, the step budget, the "first step" flag.
    Contact con; int ncon = 0;
    int phase = inb ? PH_MOVE : PH_DONE;
    bool first = true;
    size_t arow = (size_t)e;
    if (cx.sub == 0) { MI(cx, MI_EPSTEP) = st.episode_step[e]; MI(cx, MI_STATUS) = st.status[e]; MI(cx, MI_GOPEN) = st.gripper_open[e]; }
    if (sliced && inb) {
        if (mc.astate[e] != 0) {                        // waiting: start only when the last compaction gave this env a slot
            int sl = mc.slot[e];
            if (sl >= 0 && mc.gen[e] <= mc.tick[0] - lag) arow = (size_t)sl; else phase = PH_DONE;
        } else {                                        // in flight: resume
            phase = MCI(mc, MC_PHASE, e); first = false;
            if (cx.sub >= 1 && cx.sub < MC_NINT) MI(cx, cx.sub) = MCI(mc, cx.sub, e);   // cnt, nsub, grasped, flags, fault: the record has the layout of ES_MACI
            if (cx.sub < MC_NFLT) S[ES_MAC + cx.sub] = MCF(mc, cx.sub, e);         // targets etc.: the record has the layout of ES_MAC
        }
    }
    wave_sync();
    int budget = sliced ? slice : 0x7fffffff;
    PairMemo sep; sep.sep = v3(0, 0, 0); sep.h1 = sep.h2 = -1; PAIRMEMO_EXTRA_INIT(sep)   // collide()'s per-lane memory of its pair's separating direction
#ifndef GRIP_COLD_PORTAL
    // the portal memory lives as long as the macro step: a resumed env gets it back, so that the result does not depend on where
    // the time slices end (lock-step keeps it in registers for the whole macro step)
    if (sliced && inb && !first) {
        const float4 m0 = *MEMO4(mc, e, 0, cx.sub);
        const int w0 = __float_as_int(m0.x);
        sep.has = w0 & 1; sep.h1 = ((w0 >> 2) & 0xfff) - 1; sep.h2 = ((w0 >> 14) & 0xfff) - 1;
        if (w0 & 2) sep.sep = v3(m0.y, m0.z, m0.w);
        if (sep.has) {                                      // only lanes whose pair was in contact carry more than the first block
            float4 *pm = reinterpret_cast<float4 *>(S + ES_PORTAL + min(cx.sub, 10) * PORTAL_WORDS);      // (only hull-pair lanes 0..10 ever have a portal)
            pm[0] = *MEMO4(mc, e, 1, cx.sub); pm[1] = *MEMO4(mc, e, 2, cx.sub); pm[2] = *MEMO4(mc, e, 3, cx.sub);
        }
    }
#endif
    // the wall-clock budget runs from the moment the FIRST workgroup of the launch started (k_compact clears the stamp), so
    // that a workgroup that was placed late -- other streams' kernels were using its CU -- does not stretch the launch
    // (the same stamp times the launch: t0[0] = start of the first workgroup, t0[1] = end of the last wave; k_compact, which follows every
    // slice launch on its stream, adds the difference to t0[2] and counts the launch in t0[3] -- every launch, a replayed graph's too, where
    // host-side events cannot be recorded: grip_batch_device_time)
    unsigned long long t0v = 0ULL;
    if (sliced) {
        if (cx.lane == 0) {
            unsigned long long now = (unsigned long long)wall_clock64();          // 100 MHz
            unsigned long long old = atomicCAS(mc.t0, 0ULL, now);
            t0v = old ? old : now;
            // more workgroups than the chip holds at once (> 4096 envs): the later rounds start when the first ones are done and
            // get a budget of their own; a workgroup that is merely placed a little late still ends with the launch
            if (budget_ticks > 0 && now - t0v > (unsigned long long)(budget_ticks / 2)) t0v = now;
        }
        // wave-uniform: keep it in scalar registers
        t0v = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(t0v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(t0v & 0xffffffffULL));
    }
    const long long t_start = (long long)t0v;

    while (__any(phase != PH_DONE && (budget > 0 || phase == PH_FINAL))) {
#ifdef GRIP_STAMPS
        stm.acc[12] += __popcll(__ballot(cx.sub == 0 && cx.lane < EPW * KL && phase != PH_DONE && (budget > 0 || phase == PH_FINAL)));   // envs of the wave still working
        stm.acc[13] += 1;
#endif
        if (phase != PH_DONE && (budget > 0 || phase == PH_FINAL)) {
            float delta_pre = 0.f;
            {   // ---- position stage of "now" + everything that needs the full kinematics in registers: the first step's target pose,
                // the control hooks (they read qpos and the targets only), the dense dynamics block
                float q[14]; Kin k;
                forward_kin(m, cx, q, k);
                STAMP(stm, 0);
                if (first) {
                    first = false;
                    float act[6], target[5], open_close;
#pragma unroll
                    for (int i = 0; i < 6; i++) act[i] = i < adim ? actions[arow * adim + i] : 0.f;
                    target_pose(cfg, act, q, k, target, open_close);
                    if (cx.sub == 0) {                  // targets, the joint positions to return to, the object's start (ES_MAC)
                        float4 *M4 = reinterpret_cast<float4 *>(S + ES_MAC);
                        M4[0] = make_float4(target[0], target[1], target[2], target[3]); M4[1] = make_float4(target[4], q[0], q[1], q[2]);
                        M4[2] = make_float4(q[3], q[4], open_close, 0.f); M4[3] = make_float4(k.po.x, k.po.y, k.po.z, 0.f);
                    }
                    wave_sync();
                }
                if (phase != PH_FINAL) {
                    // ---- pre-step hooks
                    float ctrl[8]; lds_ld<8>(S + ES_CTRL, ctrl);
                    if (phase == PH_MOVE || phase == PH_RETURN) {
                        // Actuator.scale_control (actuator.py:46-48): MinMaxScaler.transform, no clipping
                        float tg[8]; lds_ld<8>(S + ES_MAC + (phase == PH_RETURN ? MC_INITQ : MC_TARGET) / 4 * 4, tg);
                        const int o = phase == PH_RETURN ? MC_INITQ % 4 : 0;
                        float st_ = 2.0f / (2.0f * cfg.max_translation), sr_ = 2.0f / (2.0f * cfg.max_rotation);
#pragma unroll
                        for (int i = 0; i < 3; i++) ctrl[i] = ((o ? tg[i + 1] : tg[i]) - q[i]) * st_;
#pragma unroll
                        for (int i = 3; i < 5; i++) ctrl[i] = ((o ? tg[i + 1] : tg[i]) - q[i]) * sr_;
                        if (cx.sub == 0) lds_st<8>(S + ES_CTRL, ctrl);
                    } else {
                        const float tq = S[ES_MAC + MC_TQ];
                        delta_pre = fmaxf(fabsf(tq - q[5]), fabsf(tq - q[6]));
                    }
                    STAMP(stm, 28);
                    forward_dense(m, cx, k, ctrl, xfrc_z, nullptr, stm);
                }
            }
            ncon = collide(m, cx, con, sep, stm);       // contacts as check_grasp sees them
            STAMP(stm, 1);
            if (phase == PH_FINAL) {
                // ---- robot_env.py:170-241 on the final state
                int ef = e; asm volatile("" : "+v"(ef));         // the result addresses are formed here, not before the loop (and kept)
                KinC kc; kinc_load(S, kc);
                V3 fo = kc.po, fe = kc.pe;
                V3 init_obj = v3(S[ES_MAC + MC_INITOBJ], S[ES_MAC + MC_INITOBJ + 1], S[ES_MAC + MC_INITOBJ + 2]);
                float dxy = sqrtf((fo.x - fe.x) * (fo.x - fe.x) + (fo.y - fe.y) * (fo.y - fe.y));
                int status = MI(cx, MI_STATUS), gripper_open = MI(cx, MI_GOPEN), episode_step = MI(cx, MI_EPSTEP), fault = MI(cx, MI_FAULT);
                const int grasped = MI(cx, MI_GRASPED), flags = MI(cx, MI_FLAGS), nsub = MI(cx, MI_NSUB);
                const bool reached_target = flags & 1, reached_initial = flags & 2;
                if (dxy > 1.f) status = 1;
                float p1 = project_dir(fo.x, fo.y, cfg);
                float dgx = p1 * cfg.dir_x, dgy = p1 * cfg.dir_y;
                float line;
                float reward = agent_reward(init_obj, fo, cfg, gripper_open, S[ES_CTRL + 5], S[ES_CTRL + 6], grasped, line);
                if (cfg.her_buffer) { float gx = dgx - fo.x, gy = dgy - fo.y; reward += 1.0f / expf(sqrtf(gx * gx + gy * gy)); }
                // a diverged env (NaN / runaway state, fault bit 0) is what dm_control reports as PhysicsError: the episode ends as a
                // failure with zero reward and finite outputs, and the state is reset even without auto_reset -- one bad env
                // must not poison a batch of thousands through a NaN reward
                const bool diverged = (fault & 1) != 0 || !(reward == reward) || !(dxy == dxy);
                if (diverged) { fault |= 1; status = 1; reward = 0.f; line = 0.f; fo = v3(0, 0, 0); fe = v3(0, 0, 0); dgx = cfg.dir_x; dgy = cfg.dir_y; init_obj = v3(0, 0, 0); }
                int done;
                if (status != 0) done = 1;
                else if (episode_step == cfg.time_horizon - 1) { done = 1; status = 2; }
                else done = 0;
                episode_step += 1;
                int pg = check_grasp(cx, con, ncon), ph = pheromone_level(fe, cfg);
                if (writer) {
                    if (out.reward) out.reward[ef] = reward;
                    if (out.done) out.done[ef] = (uint8_t)done;
                    if (out.status) out.status[ef] = status;
                    if (out.episode_step) out.episode_step[ef] = episode_step;
                    if (out.gripper_open) out.gripper_open[ef] = gripper_open;
                    if (out.object_grasped) out.object_grasped[ef] = grasped;
                    if (out.position_reached) out.position_reached[ef] = (reached_target ? 1 : 0) | (reached_initial ? 2 : 0) | ((!reached_target && !reached_initial) ? 4 : 0);
                    if (out.total_distance) out.total_distance[ef] = sqrtf((fo.x - init_obj.x) * (fo.x - init_obj.x) + (fo.y - init_obj.y) * (fo.y - init_obj.y));
                    if (out.line_distance) out.line_distance[ef] = line;
                    if (out.gripper_position) { out.gripper_position[3 * ef] = fe.x; out.gripper_position[3 * ef + 1] = fe.y; out.gripper_position[3 * ef + 2] = fe.z; }
                    if (out.object_position) { out.object_position[3 * ef] = fo.x; out.object_position[3 * ef + 1] = fo.y; out.object_position[3 * ef + 2] = fo.z; }
                    if (out.init_obj_pos) { out.init_obj_pos[3 * ef] = init_obj.x; out.init_obj_pos[3 * ef + 1] = init_obj.y; out.init_obj_pos[3 * ef + 2] = init_obj.z; }
                    if (out.n_substeps) out.n_substeps[ef] = nsub;
                    if (out.fault) out.fault[ef] = fault;
                }
                float agx = fo.x, agy = fo.y;
                if (done && (cfg.auto_reset || diverged)) {
                    if (cx.sub == 0) reset_state(m, cx);
                    episode_step = 0; status = 0; gripper_open = 1;
                    pg = (int)reset_info[0]; ph = (int)reset_info[1]; agx = reset_info[2]; agy = reset_info[3]; dgx = cfg.dir_x; dgy = cfg.dir_y;
                }
                st_state(st, ef, cx, valid);
                if (writer) {
                    if (out.achieved_goal) { out.achieved_goal[2 * ef] = agx; out.achieved_goal[2 * ef + 1] = agy; }
                    if (out.desired_goal) { out.desired_goal[2 * ef] = dgx; out.desired_goal[2 * ef + 1] = dgy; }
                    st.pad_grasp[ef] = pg; st.pad_pher[ef] = ph;
                    st.episode_step[ef] = episode_step; st.status[ef] = status; st.gripper_open[ef] = gripper_open;
                    if (mc.astate) { mc.astate[ef] = 1; mc.slot[ef] = -1; }
                }
                phase = PH_DONE;
            } else {
                int grasped = 0;
                if (phase == PH_CLOSE) { grasped = check_grasp(cx, con, ncon); if (cx.sub == 0) MI(cx, MI_GRASPED) = grasped; }   // robot_env.py:155, before the step
                float qn[7]; int last_iters = 0;
                physics_advance(m, cx, con, ncon, stm, qn, &last_iters);
                const int cnt = MI(cx, MI_CNT) + 1;
                if (cx.sub == 0) { MI(cx, MI_CNT) = cnt; MI(cx, MI_NSUB) += 1; MI(cx, MI_SUMIT) += last_iters; MI(cx, MI_NSLICE) += 1; }
                budget--;
#ifdef GRIP_STAMPS
                stm.acc[11] += 1;               // env-substeps of this lane (lane 0 of each wave is reported)
#endif
                // the wave's share of the tick is spent? (every step: looking only every fourth step saves a scalar memory round trip per step but
                // lets the waves overrun the budget by different amounts -- the launch then waits for the latest: measured -4.6 %)
                if (budget_ticks > 0 && wall_clock64() - t_start > budget_ticks) budget = 0;
                // ---- post-step transitions
                bool to_gripper = false, to_final = false, set56 = false, zero_cnt = false;
                float c56 = 0.f;                            // new value of ctrl[5] = ctrl[6] (knuckle motors) when a transition sets them
                if (phase == PH_MOVE || phase == PH_RETURN) {
                    float tg[8]; lds_ld<8>(S + ES_MAC + (phase == PH_RETURN ? MC_INITQ : MC_TARGET) / 4 * 4, tg);
                    const int o = phase == PH_RETURN ? MC_INITQ % 4 : 0;
                    float dmax = 0.f;
#pragma unroll
                    for (int i = 0; i < 5; i++) dmax = fmaxf(dmax, fabsf(qn[i] - (o ? tg[i + 1] : tg[i])));
                    bool reached = dmax < cfg.pos_tolerance;            // post-step qpos (quirk Q4)
                    if (reached && cx.sub == 0) {                       // ctrl[0..4] = 0
                        float z = 0.f; asm volatile("" : "+v"(z));      // (a zero made here: a hoisted constant vector was being kept -- spilled -- across the loop)
                        *reinterpret_cast<float4 *>(S + ES_CTRL) = make_float4(z, z, z, z); S[ES_CTRL + 4] = z;
                    }
                    const int flags = MI(cx, MI_FLAGS);
                    if (phase == PH_MOVE) {
                        if (reached && cx.sub == 0) MI(cx, MI_FLAGS) = flags | 1;      // reached_target
                        if (cnt == cfg.max_steps) {                     // step_limit == 0 (:112), also when reached on the last try (Q5)
                            phase = PH_RETURN; zero_cnt = true;         // (the RETURN loop steers towards init_q: MC_INITQ of ES_MAC)
                        } else if (reached) to_gripper = true;
                    } else {
                        if (reached && cx.sub == 0) MI(cx, MI_FLAGS) = flags | 2;      // reached_initial
                        if (reached || cnt == cfg.max_steps) { if (flags & 1) to_gripper = true; else to_final = true; }
                    }
                } else if (phase == PH_OPEN) {
                    const float tq = S[ES_MAC + MC_TQ];
                    bool stop = delta_pre < cfg.grasp_tolerance || (qn[5] > tq && qn[6] > tq);
                    if (stop && cx.sub == 0) MI(cx, MI_GOPEN) = 1;
                    if (stop || cnt == cfg.max_steps) { c56 = 0.f; set56 = true; to_final = true; }
                } else {   // PH_CLOSE
                    bool stop = delta_pre < cfg.grasp_tolerance || grasped == 3;
                    if (stop && cx.sub == 0) MI(cx, MI_GOPEN) = 0;
                    if (stop || cnt == cfg.max_steps) { c56 = 0.f; set56 = true; to_final = true; }
                }
                if (to_gripper) {
                    const float open_close = S[ES_MAC + MC_OPENCLOSE];
                    const int gripper_open = MI(cx, MI_GOPEN);
                    float tq = 0.f; bool set_tq = false;
                    if (open_close > 0.f && !gripper_open) { phase = PH_OPEN; zero_cnt = true; tq = 0.4f; set_tq = true; c56 = 0.5f; set56 = true; }
                    else if (open_close < 0.f && gripper_open) { phase = PH_CLOSE; zero_cnt = true; tq = -0.4f; set_tq = true; c56 = -1.f; set56 = true; }
                    else to_final = true;
                    if (set_tq && cx.sub == 0) S[ES_MAC + MC_TQ] = tq;
                }
                if (zero_cnt && cx.sub == 0) MI(cx, MI_CNT) = 0;
                if (to_final) {
                    const int flags = MI(cx, MI_FLAGS);
                    if (!(flags & 3) && cx.sub == 0) MI(cx, MI_STATUS) = 1;       // robot_env.py:130-132: neither the target nor the initial pose reached
                    phase = PH_FINAL;
                }
                if (set56 && cx.sub == 0) { S[ES_CTRL + 5] = c56; S[ES_CTRL + 6] = c56; }
                wave_sync();
                STAMP(stm, 27);
            }
        }
    }
#ifdef GRIP_STAMPS
    if (cx.lane == 0) for (int i = 0; i < NSTAMP; i++) atomicAdd(&g_stamp_acc[i], stm.acc[i]);      // whole-GPU phase totals (diagnostic build)
#endif
    int ee = e; asm volatile("" : "+v"(ee));                 // (the suspend addresses are formed here, after the loop)
    if (sliced && cx.lane == 0) atomicMax(mc.t0 + 1, (unsigned long long)wall_clock64());      // end stamp of the launch (no return value: fire and forget)
    if (sliced) {
        // cost estimate of this env's next physics.step(), in units of roughly half a plain step: Newton iterations of the
        // last solve plus 1.5 per hull-hull contact (MPR refinement); k_compact sorts the work order by it
        int hv = __popc(group_bits(__ballot(cx.sub < ncon && con.g1 != 0), cx.lane));
        const int sum_iters = MI(cx, MI_SUMIT), nsub_slice = MI(cx, MI_NSLICE);
        // (twice the mean Newton iteration count of this slice: single steps alternate between 1 and 2 iterations)
        if (writer && nsub_slice > 0) mc.heavy[ee] = min(CP_CLASSES - 2, (2 * sum_iters + nsub_slice / 2) / nsub_slice + 3 * min(hv, 3));
    }
#ifndef GRIP_COLD_PORTAL
    if (sliced && valid && phase != PH_DONE) {              // every lane parks its pair's portal memory (the flag; the portal if there is one)
        const bool has_sep = dot(sep.sep, sep.sep) > 0.5f;
        const int w0 = (sep.has & 1) | (has_sep ? 2 : 0) | (((sep.h1 + 1) & 0xfff) << 2) | (((sep.h2 + 1) & 0xfff) << 14);
        *MEMO4(mc, ee, 0, cx.sub) = make_float4(__int_as_float(w0), sep.sep.x, sep.sep.y, sep.sep.z);     // the direction is read back only under its flag
        if (sep.has) {
            const float4 *pm = reinterpret_cast<const float4 *>(S + ES_PORTAL + min(cx.sub, 10) * PORTAL_WORDS);
            *MEMO4(mc, ee, 1, cx.sub) = pm[0]; *MEMO4(mc, ee, 2, cx.sub) = pm[1]; *MEMO4(mc, ee, 3, cx.sub) = pm[2];
        }
    }
#endif
    if (sliced && phase != PH_DONE) {                       // out of budget mid-step: suspend
        st_state(st, ee, cx, valid);
        if (valid && cx.sub < MC_NFLT) MCF(mc, cx.sub, ee) = S[ES_MAC + cx.sub];
        if (valid && cx.sub >= 1 && cx.sub < MC_NINT) MCI(mc, cx.sub, ee) = MI(cx, cx.sub);      // cnt, nsub, grasped, flags, fault
        if (writer) {
            st.status[ee] = MI(cx, MI_STATUS); st.gripper_open[ee] = MI(cx, MI_GOPEN);
            mc.astate[ee] = 0;
            MCI(mc, MC_PHASE, ee) = phase;
        }
    }
}

// ---- ONE kernel for a batch and for a SET of batches (different object models / target directions): the per-batch arguments
// (model constants, config, array pointers) are GroupArgs records in device memory, read through a const __restrict__ pointer
// with a workgroup-uniform index -> scalar loads, exactly what kernel arguments cost. Workgroup b belongs to the group whose
// [wg0, wg0 + workgroups) range holds it. A single batch is a set of one, so both run the same machine code and a mixed batch
// is bit-identical to its groups run alone; mixed-object training (BASELINE.json configs[3]) fills the chip with ONE grid.
struct GroupArgs {
    DevModel m; DevConfig cfg; StatePtrs st; StepOutDev out; const float *reset_info; float xfrc_z; MacroCtx mc; int *order;
    int wg0, env0;                 // first workgroup / first global env id of the group
};

DEVI int group_of_block(const GroupArgs *__restrict__ groups, int ngroups, int block) {
    int g = 0;
    for (int i = 1; i < ngroups; i++) g = block >= groups[i].wg0 ? i : g;
    return g;
}

// out1 (use_out1 != 0): result arrays given with the call (grip_batch_step / _advance of a single batch) instead of the ones
// bound to the set. slice <= 0: actions float32 [total envs, adim] (lock-step); slice > 0: [ngroups * seg_rows, adim].
__global__ void __launch_bounds__(WG_THREADS, WG_WAVES_PER_SIMD) k_macro_step(const GroupArgs *__restrict__ groups, int ngroups, const StepOutDev out1, int use_out1, const float *actions,
                                                              int seg_rows, int slice, long long budget_ticks, int lag) {
    const int g = group_of_block(groups, ngroups, (int)blockIdx.x);
    const GroupArgs &ga = groups[g];
    const int adim = ga.cfg.include_roll ? 6 : 5;
    const float *act = actions + (slice > 0 ? (size_t)g * seg_rows : (size_t)ga.env0) * adim;
    // the 16 result pointers are needed only when a macro step ends: handed on by address (the call's own arrays sit in the kernel-argument
    // segment, the set's in the group record) and loaded there, instead of being selected here and kept in 32 registers for the whole launch
    const StepOutDev *outp = use_out1 ? &out1 : &ga.out;
    macro_step_body(ga.m, ga.cfg, ga.st, act, *outp, ga.reset_info, ga.xfrc_z, ga.mc, slice, slice > 0 ? ga.order : nullptr, budget_ticks, lag,
                    (int)blockIdx.x - ga.wg0);
}

// Deterministic compaction after a time slice (one 1024-thread block): the waiting envs that hold no slot yet, scanned from env `rot` on so that
// nobody starves when more wait than `capacity`, get slots 0..count-1 (list[slot] = env, -1 beyond count). order[] is the
// work order of the next slice: envs that will run (in flight, or holding a slot) sorted by cost class -- no hull contact,
// one or two, more -- so that the 4 envs of a wave and the 16 of a workgroup cost about the same per physics.step(), then
// the idle envs, whose workgroups retire at once.
#define CP_THREADS 1024
#define CP_FAST_CHUNK 16    // batches up to 16384 envs sort their work order by the ballot / single-scan path
// id_offset: added to the env ids written to `list` (0 for a single batch, the group's first global env id in a set)
// inclusive prefix sum of one int per thread over the 1024-thread block: wave-level shuffles, then the 16 wave totals through LDS
// (two barriers per scan instead of the twenty of a Hillis-Steele sweep through LDS: the compaction runs between every two slices)
DEVI int block_scan_incl(int v, int *wsum, int &total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int x = __shfl_up(v, d); if (lane >= d) v += x; }
    __syncthreads();                                    // wsum may still be read by the previous scan
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < CP_THREADS / 64; i++) { int x = wsum[i]; before += i < w ? x : 0; tot += x; }
    total = tot;
    return v + before;
}

DEVI void compact_body(const MacroCtx &mc, int n, int capacity, int *list, int *count, int *order, int id_offset) {
    // the rotation advances with a device-side tick counter, so that a captured (hipGraph) tick keeps rotating
    const int rot = (int)(((long long)mc.tick[0] * capacity) % n);
    __shared__ int wsum[CP_THREADS / 64];
    __shared__ int cls_total[CP_CLASSES];
    const int t = threadIdx.x, chunk = (n + CP_THREADS - 1) / CP_THREADS;
    // pass 1: slots for the waiting envs, in rotated order
    int c = 0;
    const int tick = mc.tick[0];
    for (int i = 0; i < chunk; i++) { int v = t * chunk + i; if (v < n) { int e = v + rot; if (e >= n) e -= n; c += (mc.astate[e] != 0 && mc.slot[e] < 0); } }
    int total;
    int base = block_scan_incl(c, wsum, total) - c;
    for (int i = 0; i < chunk; i++) {
        int v = t * chunk + i;
        if (v < n) {
            int e = v + rot; if (e >= n) e -= n;
            if (mc.astate[e] != 0 && mc.slot[e] < 0) { int p = base++; if (p < capacity) { mc.slot[e] = p; mc.gen[e] = tick; list[p] = e + id_offset; } }
        }
    }
    int cnt = total < capacity ? total : capacity;
    for (int p = cnt + t; p < capacity; p += CP_THREADS) list[p] = -1;
    if (t == 0) *count = cnt;
    __syncthreads();
    // pass 2: counting sort of the envs by class (0..14 = running by cost, 15 = idle), stable within a class
    auto cls_of = [&](int e) { bool run = mc.astate[e] == 0 || mc.slot[e] >= 0; return !run ? CP_CLASSES - 1 : min(max(mc.heavy[e], 0), CP_CLASSES - 2); };
    if (chunk <= CP_FAST_CHUNK) {
        // env e = i * 1024 + t: per chunk row i and wave w the class counts come from 16 ballots; one block scan over the table
        // [class][row][wave] gives every (class, row, wave) cell its first position, a lane's rank inside its cell is a popcount.
        // Same order[] as the general path below (class-major, env-ascending), four barriers instead of thirty-four.
        __shared__ int hist[CP_CLASSES * CP_FAST_CHUNK * (CP_THREADS / 64)];
        const int lane = t & 63, w = t >> 6;
        for (int i = 0; i < chunk; i++) {
            const int e = i * CP_THREADS + t;
            const int k = e < n ? cls_of(e) : -1;
            for (int kk = 0; kk < CP_CLASSES; kk++) {
                const unsigned long long mk = __ballot(k == kk);
                if (lane == kk) hist[(kk * chunk + i) * (CP_THREADS / 64) + w] = __popcll(mk);
            }
        }
        __syncthreads();
        const int cells = CP_CLASSES * chunk * (CP_THREADS / 64), per = (cells + CP_THREADS - 1) / CP_THREADS;
        int sum = 0;
        for (int j = 0; j < per; j++) { const int c2 = t * per + j; if (c2 < cells) sum += hist[c2]; }
        int tot;
        int run = block_scan_incl(sum, wsum, tot) - sum;
        for (int j = 0; j < per; j++) { const int c2 = t * per + j; if (c2 < cells) { const int v = hist[c2]; hist[c2] = run; run += v; } }
        __syncthreads();
        for (int i = 0; i < chunk; i++) {
            const int e = i * CP_THREADS + t;
            const int k = e < n ? cls_of(e) : -1;
            int rank = 0;
            for (int kk = 0; kk < CP_CLASSES; kk++) {
                const unsigned long long mk = __ballot(k == kk);
                if (k == kk) rank = __popcll(mk & ((1ull << lane) - 1ull));
            }
            if (k >= 0) order[hist[(k * chunk + i) * (CP_THREADS / 64) + w] + rank] = e;
        }
    } else {
        int mine[CP_CLASSES], before[CP_CLASSES];
        for (int k = 0; k < CP_CLASSES; k++) mine[k] = 0;
        for (int i = 0; i < chunk; i++) { int e = t * chunk + i; if (e < n) mine[cls_of(e)]++; }
        for (int k = 0; k < CP_CLASSES; k++) {
            int tot;
            before[k] = block_scan_incl(mine[k], wsum, tot) - mine[k];
            if (t == 0) cls_total[k] = tot;
        }
        __syncthreads();
        int start = 0;
        for (int k = 0; k < CP_CLASSES; k++) { before[k] += start; start += cls_total[k]; }
        for (int i = 0; i < chunk; i++) { int e = t * chunk + i; if (e < n) order[before[cls_of(e)]++] = e; }
    }
    if (t == 0) {                                       // every thread read the old tick before the first barrier
        mc.tick[0] = mc.tick[0] + 1;
        const unsigned long long a = mc.t0[0], z = mc.t0[1];          // the slice launch this compaction follows: 100 MHz stamps
        if (a != 0ULL && z > a) { mc.t0[2] += z - a; mc.t0[3] += 1ULL; }
        mc.t0[0] = 0ULL; mc.t0[1] = 0ULL;
    }
}
// one block per group: segment g of the list gets the group's waiting envs (global ids), -1 beyond its count (counts[g]);
// total != NULL receives the number of rows (the merged list of a set has holes: validity is the sign of an entry)
__global__ void __launch_bounds__(CP_THREADS) k_compact(const GroupArgs *__restrict__ groups, int seg_rows, int *list, int *counts, int *total) {
    const GroupArgs &ga = groups[blockIdx.x];
    compact_body(ga.mc, ga.st.n, seg_rows, list + (size_t)blockIdx.x * seg_rows, counts + blockIdx.x, ga.order, ga.env0);
    if (total && blockIdx.x == 0 && threadIdx.x == 0) *total = (int)gridDim.x * seg_rows;
}

// k calls of physics.step() with the stored ctrl (test hook / micro-benchmark)
__global__ void __launch_bounds__(WG_THREADS, WG_WAVES_PER_SIMD) k_substep(const DevModel m, StatePtrs st, int nsteps, float xfrc_z, int *fault_out) {
    const Ctx cx = stage_tables(m, lds_dyn);
    STAMPS_DECL
    bool act_; int e = blockIdx.x * EPB + wg_env_slot(act_);
    const bool valid = act_ && e < st.n;
    if (e >= st.n) e = st.n - 1;          // (idle rows of a 2-env wave mirror the env of the row 32 lanes below: same reads, no writes)
    env_lds_init(cx);
    ld_state(st, e, cx);
    Contact con; int ncon = 0;
    PairMemo sep; sep.sep = v3(0, 0, 0); sep.h1 = sep.h2 = -1; PAIRMEMO_EXTRA_INIT(sep)
    for (int i = 0; i < nsteps; i++) physics_step(m, cx, xfrc_z, con, ncon, sep, stm);
    st_state(st, e, cx, valid);
    if (valid && cx.sub == 0 && fault_out) fault_out[e] = env_fault(cx);
#ifdef GRIP_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < NSTAMP; i++) g_stamp_acc[i] = stm.acc[i];
#endif
}

__global__ void __launch_bounds__(WG_THREADS, WG_WAVES_PER_SIMD) k_debug_forward(const DevModel m, StatePtrs st, float xfrc_z, int getenv_dbgH, int *ncon_out, float *con_out, float *xpos_out,
                                                                 float *qacc_out, float *qs_out, float *M_out, float *bias_out) {
    const Ctx cx = stage_tables(m, lds_dyn);
    STAMPS_DECL
    bool act_; int e = blockIdx.x * EPB + wg_env_slot(act_);
    const bool valid = act_ && e < st.n;
    if (e >= st.n) e = st.n - 1;          // (idle rows of a 2-env wave mirror the env of the row 32 lanes below: same reads, no writes)
    env_lds_init(cx);
    ld_state(st, e, cx);
    Kin k; Contact con; int ncon = 0, iters = 0;
    PairMemo sep; sep.sep = v3(0, 0, 0); sep.h1 = sep.h2 = -1; PAIRMEMO_EXTRA_INIT(sep)
    float q0[14], bias[13], ctrl[8];
    forward_kin(m, cx, q0, k);
    lds_ld<8>(cx.envl + ES_CTRL, ctrl);
    forward_dense(m, cx, k, ctrl, xfrc_z, bias, stm);
    ncon = collide(m, cx, con, sep, stm);
    float gpos[18];                             // geom frame origins, read now: the solver reuses the frames' LDS area (EF_H)
    for (int g = 1; g <= 6; g++) { V3 p; M3 R; load_frame(cx.envl, g, p, R); gpos[3 * (g - 1)] = p.x; gpos[3 * (g - 1) + 1] = p.y; gpos[3 * (g - 1) + 2] = p.z; }
    float qs[13]; lds_ld<13>(cx.envl + ES_QS, qs);
    wave_sync();
    float qacci, jtfi, qacc[13];
    forward_acc(m, cx, con, ncon, qacci, jtfi, iters, stm, getenv_dbgH ? M_out + (size_t)e * 169 : nullptr);
    gather13(qacci, qacc);
    if (!valid) return;
    if (cx.sub < G_MAXC) {                      // lane c reports contact c
        float *o = con_out + ((size_t)e * G_MAXC + cx.sub) * 10;
        bool live = cx.sub < ncon;
        o[0] = live ? con.p.x : 0.f; o[1] = live ? con.p.y : 0.f; o[2] = live ? con.p.z : 0.f;
        o[3] = live ? con.n.x : 0.f; o[4] = live ? con.n.y : 0.f; o[5] = live ? con.n.z : 0.f;
        o[6] = live ? con.dist : 0.f; o[7] = live ? (float)con.g1 : 0.f; o[8] = live ? (float)con.g2 : 0.f; o[9] = (float)iters;
    }
    if (cx.sub != 0) return;
    ncon_out[e] = ncon;
    float xp_dummy[24];
    float *xp = getenv_dbgH ? xp_dummy : xpos_out + (size_t)e * 24;
    xp[0] = xp[1] = xp[2] = 0.f;
    xp[3] = k.pe.x; xp[4] = k.pe.y; xp[5] = k.pe.z;
    for (int g = 1; g <= 6; g++) { xp[3 * (g + 1)] = gpos[3 * (g - 1)]; xp[3 * (g + 1) + 1] = gpos[3 * (g - 1) + 1]; xp[3 * (g + 1) + 2] = gpos[3 * (g - 1) + 2]; }
    for (int i = 0; i < 13; i++) { qacc_out[(size_t)e * 13 + i] = qacc[i]; qs_out[(size_t)e * 13 + i] = qs[i]; if (!getenv_dbgH) bias_out[(size_t)e * 13 + i] = bias[i]; }
    if (!getenv_dbgH) for (int i = 0; i < 169; i++) {        // the 13 x 13 matrix from the block rows kept in LDS
        const int r = i / 13, c = i % 13, k = c - (r < 7 ? 0 : 7);
        M_out[(size_t)e * 169 + i] = (k >= 0 && k < (r < 7 ? 7 : 6)) ? cx.envl[EF_M + r * M_STRIDE + k] : 0.f;
    }
}

__global__ void __launch_bounds__(WG_THREADS, WG_WAVES_PER_SIMD) k_target_pose(const DevModel m, DevConfig cfg, StatePtrs st, const float *actions, float *target_out) {
    const Ctx cx = stage_tables(m, lds_dyn);
    bool act_; int e = blockIdx.x * EPB + wg_env_slot(act_);
    const bool valid = act_ && e < st.n;
    if (e >= st.n) e = st.n - 1;          // (idle rows of a 2-env wave mirror the env of the row 32 lanes below: same reads, no writes)
    ld_state(st, e, cx);
    float q[14]; lds_ld<14>(cx.envl + ES_QPOS, q);
    Kin k; kinematics(m, q, k, cx, false);
    const int adim = cfg.include_roll ? 6 : 5;
    float act[6], target[5], oc;
    for (int i = 0; i < 6; i++) act[i] = i < adim ? actions[(size_t)e * adim + i] : 0.f;
    target_pose(cfg, act, q, k, target, oc);
    if (valid && cx.sub == 0) for (int i = 0; i < 5; i++) target_out[(size_t)e * 5 + i] = target[i];
}

// [rows][cols] -> [cols][rows]; used by the env-major <-> SoA state hooks
__global__ void k_transpose(const float *src, float *dst, int rows, int cols, int src_half, int dst_half) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    int r = i / cols, c = i % cols;
    st_word(dst, (size_t)c * rows + r, ld_word(src, i, src_half), dst_half);
}
// elementwise fp32 <-> half change of a state array's storage (src and dst are different buffers)
__global__ void k_restore(const float *src, float *dst, size_t cnt, int src_half, int dst_half) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cnt) st_word(dst, i, ld_word(src, i, src_half), dst_half);
}

// ------------------------------------------------------------------------------------------------
// host API
// ------------------------------------------------------------------------------------------------
static StatePtrs state_ptrs(GripBatch *b) {
    StatePtrs p; p.qpos = b->qpos; p.qvel = b->qvel; p.ctrl = b->ctrl; p.warm = b->warm; p.episode_step = b->episode_step;
    p.status = b->status; p.gripper_open = b->gripper_open; p.pad_grasp = b->pad_grasp; p.pad_pher = b->pad_pher; p.n = b->n;
    p.half = b->cfg.state_half;
    return p;
}
static StepOutDev to_dev(const GripStepOut *o) {
    StepOutDev d; memset(&d, 0, sizeof d);
    if (!o) return d;
    d.reward = o->reward; d.done = o->done; d.achieved_goal = o->achieved_goal; d.desired_goal = o->desired_goal; d.status = o->status;
    d.episode_step = o->episode_step; d.gripper_open = o->gripper_open; d.object_grasped = o->object_grasped;
    d.position_reached = o->position_reached; d.total_distance = o->total_distance; d.line_distance = o->line_distance;
    d.gripper_position = o->gripper_position; d.object_position = o->object_position; d.init_obj_pos = o->init_obj_pos;
    d.n_substeps = o->n_substeps; d.fault = o->fault;
    return d;
}
static MacroCtx macro_ctx(GripBatch *b) { MacroCtx c; c.ints = b->mc_ints; c.flts = b->mc_flts; c.astate = b->mc_astate; c.slot = b->mc_slot; c.heavy = b->mc_heavy; c.tick = b->mc_tick; c.gen = b->mc_gen; c.t0 = b->mc_t0; c.memo = b->mc_memo; return c; }
static int grid_of(const GripBatch *b) { return (b->n + EPB - 1) / EPB; }

// observation kernel (grip_render.hip)
extern "C" int grip_render_launch(const RenderGroup *groups_dev, int ngroups, const int *list, const int *count, int nblocks, int nplanes_max, int nverts_max,
                                  uint8_t *obs, uint8_t *obs2, const long long *row2, hipStream_t s);

template <class T> static T *off(T *p, size_t n) { return p ? p + n : nullptr; }

// the GroupArgs / RenderGroup records of batch b as member of a set: first workgroup wg0, first global env id env0, result arrays
// `out` over all envs of the set (NULL: none bound)
static void fill_group(GripBatch *b, int wg0, int env0, const GripStepOut *out, GroupArgs &ga, RenderGroup &rg) {
    memset(&ga, 0, sizeof ga);
    ga.m = b->hmodel; ga.cfg = b->cfg; ga.st = state_ptrs(b); ga.reset_info = b->reset_info; ga.xfrc_z = b->xfrc_z;
    ga.mc = macro_ctx(b); ga.order = b->mc_order; ga.wg0 = wg0; ga.env0 = env0;
    if (out) {
        const GripStepOut &o = *out; StepOutDev &d = ga.out; const size_t e0 = (size_t)env0;
        d.reward = off(o.reward, e0); d.done = off(o.done, e0); d.achieved_goal = off(o.achieved_goal, 2 * e0); d.desired_goal = off(o.desired_goal, 2 * e0);
        d.status = off(o.status, e0); d.episode_step = off(o.episode_step, e0); d.gripper_open = off(o.gripper_open, e0);
        d.object_grasped = off(o.object_grasped, e0); d.position_reached = off(o.position_reached, e0); d.total_distance = off(o.total_distance, e0);
        d.line_distance = off(o.line_distance, e0); d.gripper_position = off(o.gripper_position, 3 * e0); d.object_position = off(o.object_position, 3 * e0);
        d.init_obj_pos = off(o.init_obj_pos, 3 * e0); d.n_substeps = off(o.n_substeps, e0); d.fault = off(o.fault, e0);
    }
    rg.m = b->hmodel; rg.cfg = b->cfg; rg.qpos = b->qpos; rg.pad_grasp = b->pad_grasp; rg.pad_pher = b->pad_pher; rg.n = b->n; rg.env0 = env0;
}
// (re)write the batch's own records; synchronous -- callers are creation and the configuration setters, which synchronise anyway
static int upload_self(GripBatch *b) {
    GroupArgs ga; RenderGroup rg;
    fill_group(b, 0, 0, nullptr, ga, rg);
    HIPCHK(hipMemcpy(b->d_self, &ga, sizeof ga, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->d_rself, &rg, sizeof rg, hipMemcpyHostToDevice));
    return 0;
}

static int ensure_lds_attr(int device) {      // function attributes are per device
    static std::mutex mu; static std::vector<char> set_on;
    std::lock_guard<std::mutex> lock(mu);
    if (device >= 0 && (size_t)device < set_on.size() && set_on[device]) return 0;
    HIPCHK(hipFuncSetAttribute((const void *)k_reset, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)k_macro_step, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)k_substep, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)k_debug_forward, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_BYTES));
    HIPCHK(hipFuncSetAttribute((const void *)k_target_pose, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX_BYTES));
    if (device >= 0) { if (set_on.size() <= (size_t)device) set_on.resize(device + 1, 0); set_on[device] = 1; }
    return 0;
}

static int batch_build(GripBatch *b, const GripModel *m);
extern "C" int grip_batch_create(const GripModel *m, int n_envs, int device_id, GripBatch **out) {
    if (!m || n_envs <= 0 || !out) return fail("grip_batch_create: bad arguments");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("grip_batch_create: no HIP device (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= ndev) return fail("grip_batch_create: device_id out of range");
    HIPCHK(hipSetDevice(device_id));
    if (ensure_lds_attr(device_id)) return -1;
    GripBatch *b = new GripBatch(); b->n = n_envs; b->device = device_id;
    if (batch_build(b, m) || grip_batch_reset(b, nullptr, nullptr, nullptr)) {      // g_err is set; release what was allocated
        std::string why = g_err; grip_batch_destroy(b); g_err = why; return -1;
    }
    *out = b;
    return 0;
}

static int batch_build(GripBatch *b, const GripModel *m) {
    size_t N = (size_t)b->n;
    HIPCHK(hipMalloc(&b->d_hull, m->hull_blob.size() * sizeof(unsigned)));
    HIPCHK(hipMemcpy(b->d_hull, m->hull_blob.data(), m->hull_blob.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    b->lds_bytes = ((size_t)LDS_ENV_BASE(m->hull_blob.size()) + (size_t)EPB * ENV_FLOATS) * sizeof(float);
    static_assert(ENV_FLOATS % 4 == 0 && EF_U % 4 == 0, "Hessian-vector slots must stay 16-byte aligned");
    if (b->lds_bytes > 160 * 1024) return fail("model hull tables do not fit the 160 KiB LDS next to the per-lane contact storage");
    if (m->planes.size() / 4 > RMAXPL) return fail("model has more hull face planes than the observation kernel's LDS table holds (RMAXPL)");
    b->nplanes = (int)(m->planes.size() / 4); b->nverts = m->nvert;
    HIPCHK(hipMalloc(&b->d_planes, m->planes.size() * sizeof(float)));
    HIPCHK(hipMemcpy(b->d_planes, m->planes.data(), m->planes.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc(&b->d_loops, m->loops.size() * sizeof(int)));
    HIPCHK(hipMemcpy(b->d_loops, m->loops.data(), m->loops.size() * sizeof(int), hipMemcpyHostToDevice));
    DevModel hm_ = m->host; hm_.hull_blob = b->d_hull; hm_.hull_planes = b->d_planes; hm_.hull_ladr = b->d_loops; hm_.hull_loops = b->d_loops + b->nplanes + 1;
    b->hmodel = hm_;
    HIPCHK(hipMalloc(&b->qpos, 14 * N * sizeof(float))); HIPCHK(hipMalloc(&b->qvel, 13 * N * sizeof(float)));
    HIPCHK(hipMalloc(&b->ctrl, 7 * N * sizeof(float))); HIPCHK(hipMalloc(&b->warm, 13 * N * sizeof(float)));
    HIPCHK(hipMalloc(&b->episode_step, N * sizeof(int))); HIPCHK(hipMalloc(&b->status, N * sizeof(int)));
    HIPCHK(hipMalloc(&b->gripper_open, N * sizeof(int))); HIPCHK(hipMalloc(&b->pad_grasp, N * sizeof(int)));
    HIPCHK(hipMalloc(&b->pad_pher, N * sizeof(int))); HIPCHK(hipMalloc(&b->reset_info, 4 * sizeof(float)));
    HIPCHK(hipMalloc(&b->mc_ints, (size_t)MC_NINT_PAD * N * sizeof(int))); HIPCHK(hipMalloc(&b->mc_flts, (size_t)MC_NFLT_PAD * N * sizeof(float)));
    HIPCHK(hipMalloc(&b->mc_astate, N * sizeof(int))); HIPCHK(hipMalloc(&b->mc_slot, N * sizeof(int))); HIPCHK(hipMalloc(&b->mc_order, N * sizeof(int)));
    HIPCHK(hipMalloc(&b->mc_heavy, N * sizeof(int))); HIPCHK(hipMemset(b->mc_heavy, 0, N * sizeof(int)));
    HIPCHK(hipMalloc(&b->mc_tick, sizeof(int))); HIPCHK(hipMemset(b->mc_tick, 0, sizeof(int)));
    HIPCHK(hipMalloc(&b->mc_gen, N * sizeof(int))); HIPCHK(hipMemset(b->mc_gen, 0, N * sizeof(int)));
    HIPCHK(hipMalloc(&b->mc_t0, 4 * sizeof(unsigned long long))); HIPCHK(hipMemset(b->mc_t0, 0, 4 * sizeof(unsigned long long)));     // start, end, sum, launches
    HIPCHK(hipMalloc(&b->mc_memo, (size_t)MC_MEMO_WORDS * 16 * N * sizeof(float))); HIPCHK(hipMemset(b->mc_memo, 0, (size_t)MC_MEMO_WORDS * 16 * N * sizeof(float)));
    {   std::vector<int> ident(N); for (size_t i = 0; i < N; i++) ident[i] = (int)i;          // work order: identity until the first compaction
        HIPCHK(hipMemcpy(b->mc_order, ident.data(), N * sizeof(int), hipMemcpyHostToDevice)); }
    b->scratch_bytes = 169 * N * sizeof(float) + 1024;
    HIPCHK(hipMalloc(&b->scratch, b->scratch_bytes));
    // config defaults (config/base_config.py:12-54)
    b->cfg.max_steps = 400; b->cfg.time_horizon = 400; b->cfg.include_roll = 1; b->cfg.full_observation = 1; b->cfg.her_buffer = 0;
    b->cfg.auto_reset = 0; b->cfg.max_translation = 0.05f; b->cfg.max_rotation = 0.15f; b->cfg.pos_tolerance = 0.002f;
    b->cfg.grasp_tolerance = 0.03f; b->cfg.dir_x = 1.f; b->cfg.dir_y = 0.f; b->cfg.state_half = 0;
    b->xfrc_z = -(0.438f * m->host.gravity_z);      // robot_env.py:64-65, constant verbatim
    b->ev0.assign(EV_RING, nullptr); b->ev1.assign(EV_RING, nullptr);
    for (int i = 0; i < EV_RING; i++) { HIPCHK(hipEventCreate(&b->ev0[i])); HIPCHK(hipEventCreate(&b->ev1[i])); }
    HIPCHK(hipMalloc(&b->d_self, sizeof(GroupArgs))); HIPCHK(hipMalloc(&b->d_rself, sizeof(RenderGroup)));
    return upload_self(b);
}

extern "C" void grip_batch_destroy(GripBatch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    (void)hipDeviceSynchronize();
    void *ptrs[] = {b->d_hull, b->d_planes, b->d_loops, b->qpos, b->qvel, b->ctrl, b->warm, b->episode_step, b->status,
                    b->gripper_open, b->pad_grasp, b->pad_pher, b->reset_info, b->scratch, b->mc_ints, b->mc_flts, b->mc_astate,
                    b->mc_slot, b->mc_order, b->mc_heavy, b->mc_tick, b->mc_gen, b->mc_t0, b->mc_memo, b->d_self, b->d_rself};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    for (auto &e : b->ev0) if (e) (void)hipEventDestroy(e);
    for (auto &e : b->ev1) if (e) (void)hipEventDestroy(e);
    delete b;
}

extern "C" int grip_batch_num_envs(const GripBatch *b) { return b ? b->n : -1; }

extern "C" int grip_batch_set_config(GripBatch *b, const GripEnvConfig *c) {
    if (!b || !c) return fail("grip_batch_set_config: bad arguments");
    if (c->max_steps <= 0 || c->time_horizon <= 0) return fail("grip_batch_set_config: max_steps and time_horizon must be positive");
    b->cfg.max_steps = c->max_steps; b->cfg.time_horizon = c->time_horizon; b->cfg.include_roll = c->include_roll;
    b->cfg.full_observation = c->full_observation; b->cfg.her_buffer = c->her_buffer; b->cfg.auto_reset = c->auto_reset;
    b->cfg.max_translation = c->max_translation; b->cfg.max_rotation = c->max_rotation; b->cfg.pos_tolerance = c->pos_tolerance;
    b->cfg.grasp_tolerance = c->grasp_tolerance; b->cfg.dir_x = c->target_dir[0]; b->cfg.dir_y = c->target_dir[1];
    // the reset-state pad scalars depend on the target direction: refresh them (no lane is reset: mask of zeros)
    HIPCHK(hipSetDevice(b->device));
    uint8_t *zero_mask = (uint8_t *)b->scratch;
    HIPCHK(hipMemsetAsync(zero_mask, 0, (size_t)b->n, nullptr));
    StepOutDev none; memset(&none, 0, sizeof none);
    hipLaunchKernelGGL(k_reset, dim3(1), dim3(WG_THREADS), b->lds_bytes, nullptr, b->hmodel, b->cfg, state_ptrs(b), zero_mask, none, b->reset_info, macro_ctx(b));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(nullptr));
    return upload_self(b);
}

extern "C" int grip_batch_reset(GripBatch *b, const uint8_t *mask_dev, const GripStepOut *out, void *stream) {
    if (!b) return fail("grip_batch_reset: null batch");
    HIPCHK(hipSetDevice(b->device));
    hipLaunchKernelGGL(k_reset, dim3(grid_of(b)), dim3(WG_THREADS), b->lds_bytes, (hipStream_t)stream, b->hmodel, b->cfg, state_ptrs(b), mask_dev,
                       to_dev(out), b->reset_info, macro_ctx(b));
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int grip_batch_step(GripBatch *b, const float *actions_dev, const GripStepOut *out, void *stream) {
    if (!b || !actions_dev) return fail("grip_batch_step: null argument");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)stream;
    int slot = b->ev_used % EV_RING;
    HIPCHK(hipEventRecord(b->ev0[slot], s));
    hipLaunchKernelGGL(k_macro_step, dim3(grid_of(b)), dim3(WG_THREADS), b->lds_bytes, s, (const GroupArgs *)b->d_self, 1, to_dev(out), 1, actions_dev, 0, 0, 0LL, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(b->ev1[slot], s));
    b->ev_used++;
    return 0;
}

extern "C" int grip_batch_advance(GripBatch *b, const float *slot_actions_dev, int slice, int budget_us, int lag, int capacity, const GripStepOut *out,
                                  int32_t *ready_list_dev, int32_t *ready_count_dev, void *stream) {
    if (!b || !slot_actions_dev || !ready_list_dev || !ready_count_dev) return fail("grip_batch_advance: null argument");
    if (slice <= 0 || capacity <= 0 || capacity > b->n) return fail("grip_batch_advance: need slice > 0 and 0 < capacity <= num_envs");
    if (lag != 1 && lag != 2) return fail("grip_batch_advance: lag must be 1 (decide between launches) or 2 (decide during the next launch)");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)stream;
    // launch timing uses events, which a stream under hipGraph capture cannot take: captured ticks are not timed
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s) HIPCHK(hipStreamIsCapturing(s, &cap));
    const bool timed = cap == hipStreamCaptureStatusNone;
    int slot = b->ev_used % EV_RING;
    if (timed) HIPCHK(hipEventRecord(b->ev0[slot], s));
    hipLaunchKernelGGL(k_macro_step, dim3(grid_of(b)), dim3(WG_THREADS), b->lds_bytes, s, (const GroupArgs *)b->d_self, 1, to_dev(out), 1, slot_actions_dev, capacity, slice,
                       (long long)(budget_us > 0 ? budget_us : 0) * 100LL, lag);
    HIPCHK(hipGetLastError());
    if (timed) { HIPCHK(hipEventRecord(b->ev1[slot], s)); b->ev_used++; }
    hipLaunchKernelGGL(k_compact, dim3(1), dim3(CP_THREADS), 0, s, (const GroupArgs *)b->d_self, capacity, ready_list_dev, ready_count_dev, (int *)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int grip_batch_kernel_time(GripBatch *b, int reset, float *ms_avg, int *launches) {
    if (!b) return fail("grip_batch_kernel_time: null batch");
    HIPCHK(hipSetDevice(b->device));
    int n = b->ev_used < EV_RING ? b->ev_used : EV_RING;
    double tot = 0;
    for (int i = 0; i < n; i++) {
        HIPCHK(hipEventSynchronize(b->ev1[i]));
        float ms = 0; HIPCHK(hipEventElapsedTime(&ms, b->ev0[i], b->ev1[i])); tot += ms;
    }
    if (ms_avg) *ms_avg = n ? (float)(tot / n) : 0.f;
    if (launches) *launches = n;
    if (reset) b->ev_used = 0;
    return 0;
}

extern "C" int grip_batch_device_time(GripBatch *b, int reset, double *ms_avg, long long *launches, void *stream) {
    if (!b) return fail("grip_batch_device_time: null batch");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long acc[2] = {0ULL, 0ULL};
    // (no early return between the asynchronous copy into this stack frame and the synchronisation: the errors are looked at afterwards)
    const hipError_t e1 = hipMemcpyAsync(acc, b->mc_t0 + 2, sizeof acc, hipMemcpyDeviceToHost, s);
    const hipError_t e2 = (reset && e1 == hipSuccess) ? hipMemsetAsync(b->mc_t0 + 2, 0, sizeof acc, s) : hipSuccess;
    const hipError_t e3 = hipStreamSynchronize(s);
    HIPCHK(e1); HIPCHK(e2); HIPCHK(e3);
    if (ms_avg) *ms_avg = acc[1] ? (double)acc[0] / (double)acc[1] * 1e-5 : 0.0;       // 100 MHz ticks -> ms
    if (launches) *launches = (long long)acc[1];
    return 0;
}

extern "C" int grip_batch_substep(GripBatch *b, int k, void *stream) {
    if (!b || k < 0) return fail("grip_batch_substep: bad arguments");
    HIPCHK(hipSetDevice(b->device));
    hipLaunchKernelGGL(k_substep, dim3(grid_of(b)), dim3(WG_THREADS), b->lds_bytes, (hipStream_t)stream, b->hmodel, state_ptrs(b), k, b->xfrc_z, (int *)nullptr);
    HIPCHK(hipGetLastError());
    return 0;
}

struct DevBuf {      // device allocation released on every return path
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    template <class T> T *as() const { return (T *)p; }
};

static int xfer_field(GripBatch *b, float *soa, float *user, int width, bool to_user, int host_or_dev, hipStream_t s, int half = 0) {
    if (!user) return 0;
    size_t cnt = (size_t)width * b->n;
    int threads = 256, blocks = (int)((cnt + threads - 1) / threads);
    if (cnt * sizeof(float) > b->scratch_bytes) return fail("state scratch too small");
    if (to_user) {
        float *dst = host_or_dev ? user : b->scratch;
        hipLaunchKernelGGL(k_transpose, dim3(blocks), dim3(threads), 0, s, soa, dst, width, b->n, half, 0);       // [width][N] -> [N][width]
        HIPCHK(hipGetLastError());
        if (!host_or_dev) { HIPCHK(hipMemcpyAsync(user, b->scratch, cnt * sizeof(float), hipMemcpyDeviceToHost, s)); HIPCHK(hipStreamSynchronize(s)); }
    } else {
        const float *src = user;
        if (!host_or_dev) { HIPCHK(hipMemcpyAsync(b->scratch, user, cnt * sizeof(float), hipMemcpyHostToDevice, s)); src = b->scratch; }
        hipLaunchKernelGGL(k_transpose, dim3(blocks), dim3(threads), 0, s, src, soa, b->n, width, 0, half);        // [N][width] -> [width][N]
        HIPCHK(hipGetLastError());
        if (!host_or_dev) HIPCHK(hipStreamSynchronize(s));
    }
    return 0;
}

extern "C" int grip_batch_get_state(GripBatch *b, float *qpos, float *qvel, float *ctrl, float *warm, int host_or_dev, void *stream) {
    if (!b) return fail("grip_batch_get_state: null batch");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)stream;
    const int h = b->cfg.state_half;
    if (xfer_field(b, b->qpos, qpos, 14, true, host_or_dev, s, h) || xfer_field(b, b->qvel, qvel, 13, true, host_or_dev, s, h) ||
        xfer_field(b, b->ctrl, ctrl, 7, true, host_or_dev, s, h) || xfer_field(b, b->warm, warm, 13, true, host_or_dev, s)) return -1;
    return 0;
}
extern "C" int grip_batch_set_state(GripBatch *b, const float *qpos, const float *qvel, const float *ctrl, const float *warm, int host_or_dev, void *stream) {
    if (!b) return fail("grip_batch_set_state: null batch");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)stream;
    const int h = b->cfg.state_half;
    if (xfer_field(b, b->qpos, (float *)qpos, 14, false, host_or_dev, s, h) || xfer_field(b, b->qvel, (float *)qvel, 13, false, host_or_dev, s, h) ||
        xfer_field(b, b->ctrl, (float *)ctrl, 7, false, host_or_dev, s, h) || xfer_field(b, b->warm, (float *)warm, 13, false, host_or_dev, s)) return -1;
    return 0;
}
extern "C" int grip_batch_set_state_storage(GripBatch *b, int half, void *stream) {
    if (!b || (half != 0 && half != 1)) return fail("grip_batch_set_state_storage: need a batch and half in {0, 1}");
    HIPCHK(hipSetDevice(b->device));
    if (b->cfg.state_half == half) return 0;
    hipStream_t s = (hipStream_t)stream;
    float *arr[3] = {b->qpos, b->qvel, b->ctrl}; const int width[3] = {14, 13, 7};
    for (int k = 0; k < 3; k++) {       // through the scratch buffer: the two layouts of one array overlap
        size_t cnt = (size_t)width[k] * b->n;
        if (cnt * sizeof(float) > b->scratch_bytes) return fail("state scratch too small");
        HIPCHK(hipMemcpyAsync(b->scratch, arr[k], cnt * (b->cfg.state_half ? sizeof(__half) : sizeof(float)), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_restore, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, (const float *)b->scratch, arr[k], cnt, b->cfg.state_half, half);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(s));
    b->cfg.state_half = half;
    return upload_self(b);
}

extern "C" int grip_batch_get_flags(GripBatch *b, int32_t *episode_step, int32_t *status, int32_t *gripper_open, void *stream) {
    if (!b) return fail("grip_batch_get_flags: null batch");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)stream; size_t nb = (size_t)b->n * sizeof(int);
    if (episode_step) HIPCHK(hipMemcpyAsync(episode_step, b->episode_step, nb, hipMemcpyDeviceToHost, s));
    if (status) HIPCHK(hipMemcpyAsync(status, b->status, nb, hipMemcpyDeviceToHost, s));
    if (gripper_open) HIPCHK(hipMemcpyAsync(gripper_open, b->gripper_open, nb, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}
extern "C" int grip_batch_set_flags(GripBatch *b, const int32_t *episode_step, const int32_t *status, const int32_t *gripper_open, void *stream) {
    if (!b) return fail("grip_batch_set_flags: null batch");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)stream; size_t nb = (size_t)b->n * sizeof(int);
    if (episode_step) HIPCHK(hipMemcpyAsync(b->episode_step, episode_step, nb, hipMemcpyHostToDevice, s));
    if (status) HIPCHK(hipMemcpyAsync(b->status, status, nb, hipMemcpyHostToDevice, s));
    if (gripper_open) HIPCHK(hipMemcpyAsync(b->gripper_open, gripper_open, nb, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

extern "C" int grip_batch_debug_forward(GripBatch *b, int32_t *ncon, float *con, float *xpos, float *qacc, float *qacc_smooth, float *M, float *bias, void *stream) {
    if (!b || !ncon || !con || !xpos || !qacc || !qacc_smooth || !M || !bias) return fail("grip_batch_debug_forward: null argument");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)stream; size_t N = (size_t)b->n;
    DevBuf b_ncon, b_con, b_xpos, b_qacc, b_qs, b_M, b_bias;
    HIPCHK(b_ncon.alloc(N * sizeof(int))); HIPCHK(b_con.alloc(N * G_MAXC * 10 * sizeof(float))); HIPCHK(b_xpos.alloc(N * 24 * sizeof(float)));
    HIPCHK(b_qacc.alloc(N * 13 * sizeof(float))); HIPCHK(b_qs.alloc(N * 13 * sizeof(float))); HIPCHK(b_M.alloc(N * 169 * sizeof(float)));
    HIPCHK(b_bias.alloc(N * 13 * sizeof(float)));
    int *d_ncon = b_ncon.as<int>(); float *d_con = b_con.as<float>(), *d_xpos = b_xpos.as<float>(), *d_qacc = b_qacc.as<float>(),
          *d_qs = b_qs.as<float>(), *d_M = b_M.as<float>(), *d_bias = b_bias.as<float>();
    hipLaunchKernelGGL(k_debug_forward, dim3(grid_of(b)), dim3(WG_THREADS), b->lds_bytes, s, b->hmodel, state_ptrs(b), b->xfrc_z, getenv("GRIP_DEBUG_H") ? 1 : 0, d_ncon, d_con, d_xpos, d_qacc, d_qs, d_M, d_bias);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(ncon, d_ncon, N * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(con, d_con, N * G_MAXC * 10 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(xpos, d_xpos, N * 24 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(qacc, d_qacc, N * 13 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(qacc_smooth, d_qs, N * 13 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(M, d_M, N * 169 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(bias, d_bias, N * 13 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

extern "C" int grip_batch_target_pose(GripBatch *b, const float *actions_dev, float *target_qpos_host, void *stream) {
    if (!b || !actions_dev || !target_qpos_host) return fail("grip_batch_target_pose: null argument");
    HIPCHK(hipSetDevice(b->device));
    hipStream_t s = (hipStream_t)stream; size_t N = (size_t)b->n;
    DevBuf b_t; HIPCHK(b_t.alloc(N * 5 * sizeof(float))); float *d_t = b_t.as<float>();
    hipLaunchKernelGGL(k_target_pose, dim3(grid_of(b)), dim3(WG_THREADS), b->lds_bytes, s, b->hmodel, b->cfg, state_ptrs(b), actions_dev, d_t);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(target_qpos_host, d_t, N * 5 * sizeof(float), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

extern "C" int grip_batch_observe(GripBatch *b, uint8_t *obs_dev, void *stream) {
    if (!b || !obs_dev) return fail("grip_batch_observe: null argument");
    HIPCHK(hipSetDevice(b->device));
    if (grip_render_launch((const RenderGroup *)b->d_rself, 1, nullptr, nullptr, b->n, b->nplanes, b->nverts, obs_dev, nullptr, nullptr, (hipStream_t)stream)) return -1;
    return 0;
}
extern "C" int grip_render_camera_launch(const RenderGroup *group_dev, int env, const float *cam_dev, float fovy_deg, int w, int h, uint8_t *rgb_dev,
                                         float *depth_dev, hipStream_t s);
extern "C" int grip_batch_render_camera(GripBatch *b, int env, const float *cam_pose_dev, float fovy_deg, int width, int height, uint8_t *rgb_dev,
                                        float *depth_dev, void *stream) {
    if (!b || env < 0 || env >= b->n || width <= 0 || height <= 0 || width > 4096 || height > 4096 || (!rgb_dev && !depth_dev) || !(fovy_deg > 0.f && fovy_deg < 180.f))
        return fail("grip_batch_render_camera: bad argument");
    HIPCHK(hipSetDevice(b->device));
    return grip_render_camera_launch((const RenderGroup *)b->d_rself, env, cam_pose_dev, fovy_deg, width, height, rgb_dev, depth_dev, (hipStream_t)stream);
}
extern "C" int grip_batch_observe_list(GripBatch *b, const int32_t *list_dev, const int32_t *count_dev, int capacity, uint8_t *obs_dev,
                                       uint8_t *records_dev, const int64_t *record_row_dev, void *stream) {
    if (!b || (!obs_dev && !records_dev) || !list_dev || !count_dev || capacity <= 0 || (records_dev && !record_row_dev)) return fail("grip_batch_observe_list: bad argument");
    HIPCHK(hipSetDevice(b->device));
    if (grip_render_launch((const RenderGroup *)b->d_rself, 1, list_dev, count_dev, capacity, b->nplanes, b->nverts, obs_dev, records_dev, (const long long *)record_row_dev, (hipStream_t)stream)) return -1;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// batch sets: several batches (object models / target directions) behind one launch per phase
// ------------------------------------------------------------------------------------------------
struct GripBatchSet {
    std::vector<GripBatch *> b; int device = 0, total_envs = 0, total_wgs = 0, nplanes_max = 0, nverts_max = 0; size_t lds_max = 0;
    std::vector<int> env0, wg0;
    GripStepOut out; bool has_out = false;
    GroupArgs *d_groups = nullptr; RenderGroup *d_rgroups = nullptr;
    int *counts = nullptr;                 // per-group ready counts of the last advance (device)
};

static int set_upload(GripBatchSet *s) {
    const int G = (int)s->b.size();
    std::vector<GroupArgs> ga(G); std::vector<RenderGroup> rg(G);
    for (int g = 0; g < G; g++) fill_group(s->b[g], s->wg0[g], s->env0[g], s->has_out ? &s->out : nullptr, ga[g], rg[g]);
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemcpy(s->d_groups, ga.data(), sizeof(GroupArgs) * G, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->d_rgroups, rg.data(), sizeof(RenderGroup) * G, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int grip_batchset_create(GripBatch *const *batches, int n, const GripStepOut *out, GripBatchSet **out_set) {
    if (!batches || n <= 0 || n > 64 || !out_set) return fail("grip_batchset_create: need 1..64 batches");
    for (int g = 0; g < n; g++) {
        if (!batches[g]) return fail("grip_batchset_create: null batch");
        if (batches[g]->device != batches[0]->device) return fail("grip_batchset_create: the batches of a set live on one device");
        if (batches[g]->cfg.include_roll != batches[0]->cfg.include_roll || batches[g]->cfg.full_observation != batches[0]->cfg.full_observation)
            return fail("grip_batchset_create: the batches of a set share the action and observation layout (include_roll, full_observation)");
    }
    GripBatchSet *s = new GripBatchSet(); s->device = batches[0]->device;
    for (int g = 0; g < n; g++) {
        GripBatch *b = batches[g];
        s->b.push_back(b); s->env0.push_back(s->total_envs); s->wg0.push_back(s->total_wgs);
        s->total_envs += b->n; s->total_wgs += grid_of(b);
        s->lds_max = std::max(s->lds_max, b->lds_bytes); s->nplanes_max = std::max(s->nplanes_max, b->nplanes); s->nverts_max = std::max(s->nverts_max, b->nverts);
    }
    if (out) { s->out = *out; s->has_out = true; }
    if (hipSetDevice(s->device) != hipSuccess || hipMalloc(&s->counts, n * sizeof(int)) != hipSuccess || hipMemset(s->counts, 0, n * sizeof(int)) != hipSuccess ||
        hipMalloc(&s->d_groups, n * sizeof(GroupArgs)) != hipSuccess || hipMalloc(&s->d_rgroups, n * sizeof(RenderGroup)) != hipSuccess || set_upload(s)) {
        std::string why = g_err.empty() ? std::string("grip_batchset_create: device setup failed") : g_err;
        grip_batchset_destroy(s); g_err = why; return -1;
    }
    *out_set = s;
    return 0;
}

extern "C" void grip_batchset_destroy(GripBatchSet *s) {
    if (!s) return;
    (void)hipSetDevice(s->device); (void)hipDeviceSynchronize();
    if (s->counts) (void)hipFree(s->counts);
    if (s->d_groups) (void)hipFree(s->d_groups);
    if (s->d_rgroups) (void)hipFree(s->d_rgroups);
    delete s;
}

extern "C" int grip_batchset_refresh(GripBatchSet *s) {
    if (!s) return fail("grip_batchset_refresh: null set");
    HIPCHK(hipSetDevice(s->device)); HIPCHK(hipDeviceSynchronize());
    return set_upload(s);
}

extern "C" int grip_batchset_num_envs(const GripBatchSet *s) { return s ? s->total_envs : -1; }

static int set_launch(GripBatchSet *s, const float *actions, int seg_rows, int slice, int budget_us, int lag, hipStream_t st) {
    GripBatch *b0 = s->b[0];                         // launch timing goes to the first batch's event ring (grip_batch_kernel_time)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st) HIPCHK(hipStreamIsCapturing(st, &cap));
    const bool timed = cap == hipStreamCaptureStatusNone;
    int slot = b0->ev_used % EV_RING;
    if (timed) HIPCHK(hipEventRecord(b0->ev0[slot], st));
    StepOutDev none; memset(&none, 0, sizeof none);
    hipLaunchKernelGGL(k_macro_step, dim3(s->total_wgs), dim3(WG_THREADS), s->lds_max, st, (const GroupArgs *)s->d_groups, (int)s->b.size(), none, 0, actions, seg_rows, slice,
                       (long long)(budget_us > 0 ? budget_us : 0) * 100LL, lag);
    HIPCHK(hipGetLastError());
    if (timed) { HIPCHK(hipEventRecord(b0->ev1[slot], st)); b0->ev_used++; }
    return 0;
}

extern "C" int grip_batchset_step(GripBatchSet *s, const float *actions_dev, void *stream) {
    if (!s || !actions_dev) return fail("grip_batchset_step: null argument");
    HIPCHK(hipSetDevice(s->device));
    return set_launch(s, actions_dev, 0, 0, 0, 1, (hipStream_t)stream);
}

extern "C" int grip_batchset_advance(GripBatchSet *s, const float *slot_actions_dev, int slice, int budget_us, int lag, int capacity,
                                     int32_t *ready_list_dev, int32_t *ready_count_dev, void *stream) {
    if (!s || !slot_actions_dev || !ready_list_dev || !ready_count_dev) return fail("grip_batchset_advance: null argument");
    const int G = (int)s->b.size();
    if (slice <= 0 || capacity <= 0 || capacity % G) return fail("grip_batchset_advance: need slice > 0 and a capacity that is a multiple of the number of batches");
    const int seg = capacity / G;
    for (GripBatch *b : s->b) if (seg > b->n) return fail("grip_batchset_advance: capacity / batches exceeds a batch's env count");
    if (lag != 1 && lag != 2) return fail("grip_batchset_advance: lag must be 1 or 2");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    if (set_launch(s, slot_actions_dev, seg, slice, budget_us, lag, st)) return -1;
    hipLaunchKernelGGL(k_compact, dim3(G), dim3(CP_THREADS), 0, st, (const GroupArgs *)s->d_groups, seg, ready_list_dev, s->counts, ready_count_dev);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int grip_batchset_observe(GripBatchSet *s, uint8_t *obs_dev, void *stream) {
    if (!s || !obs_dev) return fail("grip_batchset_observe: null argument");
    HIPCHK(hipSetDevice(s->device));
    return grip_render_launch(s->d_rgroups, (int)s->b.size(), nullptr, nullptr, s->total_envs, s->nplanes_max, s->nverts_max, obs_dev, nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int grip_batchset_observe_list(GripBatchSet *s, const int32_t *list_dev, int capacity, uint8_t *obs_dev, uint8_t *records_dev,
                                          const int64_t *record_row_dev, void *stream) {
    if (!s || !list_dev || (!obs_dev && !records_dev) || capacity <= 0 || (records_dev && !record_row_dev)) return fail("grip_batchset_observe_list: bad argument");
    HIPCHK(hipSetDevice(s->device));
    return grip_render_launch(s->d_rgroups, (int)s->b.size(), list_dev, nullptr, capacity, s->nplanes_max, s->nverts_max, obs_dev, records_dev, (const long long *)record_row_dev, (hipStream_t)stream);
}

#ifdef GRIP_STAMPS
extern "C" int grip_debug_stamps(unsigned long long *out8) {   // NSTAMP entries
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamp_acc), sizeof(unsigned long long) * NSTAMP) != hipSuccess) return -1;
    unsigned long long zero[NSTAMP] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_acc), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
extern "C" int grip_debug_hist(unsigned long long *out64) {   // per-step histograms of the diagnostic build (grip_physics.h: g_dbg_hist), reset on read
    static const unsigned long long zero[96] = {0};
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dbg_hist), sizeof(unsigned long long) * 96) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_hist), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
extern "C" int grip_debug_counters(unsigned long long *out8) {   // 16 collide() event counters (grip_physics.h), reset on read
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dbg_cnt), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    unsigned long long zero[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_cnt), zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif

#ifdef GRIP_CAPDUMP
extern "C" int grip_debug_capdump(float *out, unsigned *count) {   // records of the solves that hit NEWTON_MAXIT (grip_physics.h: g_capdump), reset on read
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(g_capdump_n), sizeof(unsigned)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_capdump), sizeof(float) * CAPDUMP_RECORDS * CAPDUMP_WORDS) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(count + 1, HIP_SYMBOL(g_capdump_restarts), sizeof(unsigned)) != hipSuccess) return -1;       // count[1]: solves restarted from qacc_smooth
    const unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_capdump_restarts), &zero, sizeof zero) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_capdump_n), &zero, sizeof zero) == hipSuccess ? 0 : -1;
}
#endif

// ---- self-test hook of the lane-distributed Cholesky (tests only): x = A^-1 b for n SPD 13x13 systems
__global__ void __launch_bounds__(WG_THREADS, WG_WAVES_PER_SIMD) k_test_chol(const float *A, const float *b, float *x, int n) {
    bool act_; int e = blockIdx.x * EPB + wg_env_slot(act_), sub = threadIdx.x & (KL - 1);
    bool valid = act_ && e < n; if (e >= n) e = n - 1;
    float row[13];
    for (int j = 0; j < 13; j++) row[j] = sub < 13 ? A[(size_t)e * 169 + min(sub, 12) * 13 + j] : 0.f;
    if (sub >= 13) row[12] = 1.f;
    chol_rows(row, sub);
    float xi = chol_solve_rows(row, sub < 13 ? b[(size_t)e * 13 + min(sub, 12)] : 0.f, sub);
    float v[13]; gather13(xi, v);
    if (valid && sub == 0) for (int j = 0; j < 13; j++) x[(size_t)e * 13 + j] = v[j];
}
extern "C" int grip_selftest_cholesky(const float *A_dev, const float *b_dev, float *x_dev, int n, void *stream) {
    hipLaunchKernelGGL(k_test_chol, dim3((n + EPB - 1) / EPB), dim3(WG_THREADS), 0, (hipStream_t)stream, A_dev, b_dev, x_dev, n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
