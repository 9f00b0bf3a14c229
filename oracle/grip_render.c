/*
 * grip_render.c -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Observation path of the reference:
 *   RGBDSensor.render_images      simulation/controller/sensor.py:56-77
 *   transform_depth               simulation/utils/utils.py:11-19   (PINNED by tests/golden)
 *   RobotEnv.get_observation      simulation/environment/robot_env.py:275-293
 *
 * `physics.render` is MuJoCo's OpenGL rasteriser (third party, absent here): pixel parity with it
 * is not attainable and this renderer is UNPINNED. It is a ray caster over the same
 * scene: `gripper_camera` (robot xml :60, fovy 135 deg, on body ee), the floor plane with the
 * 2x2-checker `grid` material (:30-31,54), the six convex hulls with their rgba, the skybox
 * gradient (:29), the materials (:32-34) and both lights (:50-51) under the fixed-function lighting model (lit_colour), flat face normals. Depth is the distance along
 * the optical axis in metres, as dm_control's depth render returns it.
 */
#include "grip_oracle_int.h"

typedef struct { double o[3], R[9]; double tanh; } Cam;

static void camera_pose(const OrcModel *m, const OrcData *d, Cam *c) {
    double t[3], Rc[9];
    mulmv3(t, d->xmat[B_EE], m->cam_pos); add3(c->o, d->xpos[B_EE], t);
    quat_to_mat(Rc, m->cam_quat); mulmm3(c->R, d->xmat[B_EE], Rc);
    c->tanh = tan(0.5 * m->cam_fovy * M_PI / 180.0);
}

/* ray (o + t dir) against hull geom g; returns entering t and outward normal (world) */
static int ray_hull(const OrcModel *m, const OrcData *d, int g, const double o[3], const double dir[3], double *thit, double nrm[3]) {
    int b = m->geom_body[g];
    double c[3], oc[3], t[3];
    mulmv3(t, d->xmat[b], m->geom_center[g]); add3(c, d->xpos[b], t); sub3(oc, o, c);
    /* bounding sphere */
    double dd = dot3(dir, dir), bq = dot3(oc, dir), cq = dot3(oc, oc) - m->geom_rbound[g] * m->geom_rbound[g];
    if (bq * bq - dd * cq < 0) return 0;
    double ol[3], dl[3], rel[3];
    sub3(rel, o, d->xpos[b]); mulmtv3(ol, d->xmat[b], rel); mulmtv3(dl, d->xmat[b], dir);
    const double *pl = m->hull_planes + 4 * m->hull_padr[g - 1];
    int np = m->hull_pnum[g - 1], ent = -1;
    double tin = -1e300, tout = 1e300;
    for (int i = 0; i < np; i++) {
        const double *p = pl + 4 * i;
        double den = dot3(p, dl), num = p[3] - dot3(p, ol);
        if (den < 0) { double tt = num / den; if (tt > tin) { tin = tt; ent = i; } }
        else if (den > 0) { double tt = num / den; if (tt < tout) tout = tt; }
        else if (num < 0) return 0;
        if (tin > tout) return 0;
    }
    if (ent < 0 || tin <= 0) return 0;
    *thit = tin;
    mulmv3(nrm, d->xmat[b], pl + 4 * ent);
    return 1;
}

/* MuJoCo's fixed-function lighting of one surface point [3P-recall; unpinned like the rest of the rendering] (robot xml :29-34 materials, :50-51 lights):
 * material ambient = diffuse = the geom's rgb, specular (s, s, s), shininess x 128, emission x rgb; a headlight at the camera + the scene's lights
 * (a directional one and a spot with cutoff and exponent), no attenuation, no shadows; clamped per channel by the byte conversion. */
static void lit_colour(const OrcModel *m, int geom, const double *base, const double *P, const double *N, const double *co, double *col) {
    const double spec = m->geom_material[geom][0], shin = 128.0 * m->geom_material[geom][1];
    double V[3] = {co[0] - P[0], co[1] - P[1], co[2] - P[2]}; normalize3(V);
    double diff = m->geom_material[geom][2] + m->headlight[0], sp = 0.0;
    double nv = dot3(N, V);
    if (nv > 0) { diff += m->headlight[1] * nv; sp += m->headlight[2] * pow(nv, shin); }
    for (int l = 0; l < 2; l++) {
        double ld[3] = {m->light_dir[l][0], m->light_dir[l][1], m->light_dir[l][2]}; normalize3(ld);
        double L[3] = {-ld[0], -ld[1], -ld[2]}, spot = 1.0;
        if (!m->light_directional[l]) {
            for (int k = 0; k < 3; k++) L[k] = m->light_pos[l][k] - P[k];
            normalize3(L);
            double c = -dot3(L, ld);
            spot = c > cos(m->light_params[l][3] * 0.017453292519943295) ? pow(c, m->light_params[l][4]) : 0.0;
        }
        diff += m->light_params[l][2];
        double nl = dot3(N, L);
        if (nl > 0 && spot > 0) {
            diff += spot * m->light_params[l][0] * nl;
            double H[3] = {L[0] + V[0], L[1] + V[1], L[2] + V[2]}; normalize3(H);
            double nh = dot3(N, H); if (nh < 0) nh = 0;
            sp += spot * m->light_params[l][1] * pow(nh, shin);
        }
    }
    for (int k = 0; k < 3; k++) col[k] = base[k] * diff + spec * sp;
}

static unsigned char to_u8(double v) { v = v * 255.0; if (v < 0) v = 0; if (v > 255) v = 255; return (unsigned char)v; }

void orc_render(const OrcModel *m, const OrcData *d, int width, int height, unsigned char *rgb, float *depth) {
    Cam cam; camera_pose(m, d, &cam);
    const double znear = m->visual[1], zfar = m->visual[2];
    for (int i = 0; i < height; i++) for (int j = 0; j < width; j++) {
        double x = (2.0 * (j + 0.5) / width - 1.0) * cam.tanh * ((double)width / height);
        double y = (1.0 - 2.0 * (i + 0.5) / height) * cam.tanh;
        double dc[3] = {x, y, -1.0}, dir[3];
        mulmv3(dir, cam.R, dc);
        double best = zfar, col[3], nrm[3] = {0, 0, 1};
        int hit = -1;
        /* floor */
        if (dir[2] < 0) { double t = -cam.o[2] / dir[2]; if (t > znear && t < best) { best = t; hit = 0; } }
        for (int g = 1; g < NG; g++) {
            double t, n[3];
            if (ray_hull(m, d, g, cam.o, dir, &t, n) && t > znear && t < best) { best = t; hit = g; copy3(nrm, n); }
        }
        if (hit < 0) {
            /* skybox gradient by elevation */
            double dn[3] = {dir[0], dir[1], dir[2]}; normalize3(dn);
            double f = 0.5 * (dn[2] + 1.0);
            for (int k = 0; k < 3; k++) col[k] = m->sky_rgb[3 + k] + f * (m->sky_rgb[k] - m->sky_rgb[3 + k]);
        } else {
            double base[3], P[3] = {cam.o[0] + best * dir[0], cam.o[1] + best * dir[1], cam.o[2] + best * dir[2]};
            if (hit == 0) {
                int cx = (int)floor(P[0] * 8.0), cy = (int)floor(P[1] * 8.0);     /* texrepeat 4, 2x2 checker per repeat */
                const double *c = ((cx + cy) & 1) ? m->floor_rgb + 3 : m->floor_rgb;
                copy3(base, c);
            } else copy3(base, m->geom_rgba[hit]);
            normalize3(nrm);
            lit_colour(m, hit, base, P, nrm, cam.o, col);
        }
        int px = i * width + j;
        rgb[3 * px] = to_u8(col[0]); rgb[3 * px + 1] = to_u8(col[1]); rgb[3 * px + 2] = to_u8(col[2]);
        depth[px] = (float)best;
    }
}

/* utils.py:11-19, float32 arithmetic like the numpy original on a float32 image */
void orc_transform_depth(float *depth, int n, unsigned char *out) {
    float mn = depth[0];
    for (int i = 1; i < n; i++) if (depth[i] < mn) mn = depth[i];
    double sum = 0; int cnt = 0;
    for (int i = 0; i < n; i++) { depth[i] -= mn; if (depth[i] <= 1.0f) { sum += depth[i]; cnt++; } }
    float mean = (float)(sum / cnt);                   /* cnt == 0 -> NaN, as in the reference (quirk Q7) */
    float scale = 2.0f * mean;
    for (int i = 0; i < n; i++) {
        float v = depth[i] / scale;
        v = v < 0.f ? 0.f : v > 1.f ? 1.f : v;
        float p = 255.0f * v;
        out[i] = (p != p) ? 0 : (unsigned char)p;
    }
}

/* robot_env.py:275-293: dstack(rgb, depth, pad).astype(uint8) -> CHW */
void orc_observation(const OrcModel *m, const OrcEnvConfig *c, const OrcData *d, unsigned char *obs) {
    enum { W = 64, H = 64 };
    static _Thread_local unsigned char rgb[W * H * 3], dep8[W * H];
    static _Thread_local float depth[W * H];
    orc_render(m, d, W, H, rgb, depth);
    orc_transform_depth(depth, W * H, dep8);
    int nch = c->full_observation ? 5 : 4;
    for (int px = 0; px < W * H; px++) {
        for (int k = 0; k < 3; k++) obs[k * W * H + px] = rgb[3 * px + k];
        if (c->full_observation) obs[3 * W * H + px] = dep8[px];
        obs[(nch - 1) * W * H + px] = 0;
    }
    obs[(nch - 1) * W * H + 0] = (unsigned char)orc_check_grasp(d);
    obs[(nch - 1) * W * H + 1] = (unsigned char)orc_pheromone_level(d, c->target_dir);
}


/* reward.py:57-77 IntrinsicReward.intrinsic_reward on two CHW uint8 observations: grey-level (and, with
 * --full_observation, depth) histograms of 256 bins as probability vectors, sum of scipy.special.rel_entr(p_old, p_new)
 * with infinities zeroed, the two sums averaged. cv2 is not installed here: cv2.cvtColor(BGR2GRAY) on uint8 is restated
 * from OpenCV 4.8.1 (setup.py pins opencv-python==4.8.1.78; modules/imgproc/src/color_rgb.simd.hpp, RGB2Gray<uchar>:
 * (B*3735 + G*19235 + R*9798 + (1 << 14)) >> 15 with channel 0 taken as B), cv2.calcHist as exact counts in float32,
 * make_pdf (utils.py:5-8) as hist / hist.sum() in float32 -- parity of this row is UNPINNED for lack of cv2. */
static double hist_rel_entr(const int *ho, const int *hn, int npx) {
    double s = 0;
    for (int i = 0; i < 256; i++) {
        float p = (float)ho[i] / (float)npx, q = (float)hn[i] / (float)npx;
        if (p > 0.f && q > 0.f) s += (double)(float)(p * logf(p / q));      /* rel_entr on float32 arrays */
        /* p == 0: 0;  p > 0 and q == 0: inf, set to 0 by reward.py:66 */
    }
    return s;
}
double orc_intrinsic_reward(const unsigned char *old_obs, const unsigned char *new_obs, int full_observation) {
    enum { NPX = 64 * 64 };
    int ho[256] = {0}, hn[256] = {0};
    for (int px = 0; px < NPX; px++) {
        ho[(old_obs[px] * 3735 + old_obs[NPX + px] * 19235 + old_obs[2 * NPX + px] * 9798 + (1 << 14)) >> 15]++;
        hn[(new_obs[px] * 3735 + new_obs[NPX + px] * 19235 + new_obs[2 * NPX + px] * 9798 + (1 << 14)) >> 15]++;
    }
    double r = hist_rel_entr(ho, hn, NPX);
    if (full_observation) {
        int dO[256] = {0}, dN[256] = {0};
        for (int px = 0; px < NPX; px++) { dO[old_obs[3 * NPX + px]]++; dN[new_obs[3 * NPX + px]]++; }
        r = (r + hist_rel_entr(dO, dN, NPX)) / 2;
    }
    return r;
}
