/* grip_oracle_int.h -- CPU ORACLE (test infrastructure): internal definitions. */
#ifndef GRIP_ORACLE_INT_H
#define GRIP_ORACLE_INT_H

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "grip_oracle.h"

#define NB ORC_NB
#define NV ORC_NV
#define NQ ORC_NQ
#define NU ORC_NU
#define NG ORC_NG
#define MINVAL 1e-15

/* fixed topology (model/compiler.py) */
enum { B_WORLD = 0, B_EE, B_BASE, B_LK, B_LF, B_RK, B_RF, B_OBJ };
enum { G_FLOOR = 0, G_BASE, G_LK, G_LF, G_RK, G_RF, G_OBJ };

struct OrcModel {
    double timestep, gravity_z, impratio, tolerance; int iterations;
    double margin, solref[2], solimp[5], lim_solref[2], lim_solimp[5];
    int body_parent[NB];
    double body_pos[NB][3], body_quat[NB][4], body_mass[NB], body_ipos[NB][3], body_iquat[NB][4], body_inertia[NB][3];
    double dof_armature[NV], dof_damping[NV], jnt_range[NU][2], gear[NU], ctrlrange[NU][2];
    double qpos0[NQ];
    int geom_body[NG];
    double geom_friction[NG][3], geom_center[NG][3], geom_rbound[NG], geom_rgba[NG][4];
    int hull_vadr[NG - 1], hull_vnum[NG - 1], hull_padr[NG - 1], hull_pnum[NG - 1];
    int nvert, nplane, nnbr, npair;
    double *hull_verts, *hull_planes; int *hull_nadr, *hull_nbr, *hull_pairs;
    double body_invweight0[NB][2], dof_invweight0[NV], meaninertia;
    double cam_pos[3], cam_quat[4], cam_fovy, visual[3], floor_rgb[6], sky_rgb[6];
    double light_dir[2][3], light_pos[2][3]; int light_directional[2];
    /* materials and light colours of MuJoCo's fixed-function lighting (robot xml :29-34, :50-51): per geom specular, shininess, emission; per light
     * diffuse, specular, ambient, spot cutoff (degrees), spot exponent; the headlight's ambient, diffuse, specular */
    double geom_material[NG][3], light_params[2][5], headlight[3];
};

/* ---- tiny vector helpers ---- */
static inline double dot3(const double a[3], const double b[3]) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static inline void cross3(double r[3], const double a[3], const double b[3]) {
    double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
    r[0] = x; r[1] = y; r[2] = z;
}
static inline void copy3(double r[3], const double a[3]) { r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; }
static inline void sub3(double r[3], const double a[3], const double b[3]) { r[0] = a[0]-b[0]; r[1] = a[1]-b[1]; r[2] = a[2]-b[2]; }
static inline void add3(double r[3], const double a[3], const double b[3]) { r[0] = a[0]+b[0]; r[1] = a[1]+b[1]; r[2] = a[2]+b[2]; }
static inline void scl3(double r[3], const double a[3], double s) { r[0] = a[0]*s; r[1] = a[1]*s; r[2] = a[2]*s; }
static inline void addscl3(double r[3], const double a[3], const double b[3], double s) { r[0] = a[0]+b[0]*s; r[1] = a[1]+b[1]*s; r[2] = a[2]+b[2]*s; }
static inline double norm3(const double a[3]) { return sqrt(dot3(a, a)); }
static inline double normalize3(double a[3]) {
    double n = norm3(a);
    if (n < MINVAL) { a[0] = 1; a[1] = 0; a[2] = 0; return 0; }
    a[0] /= n; a[1] /= n; a[2] /= n; return n;
}
/* r = R * v, R row-major 3x3 */
static inline void mulmv3(double r[3], const double R[9], const double v[3]) {
    double x = R[0]*v[0] + R[1]*v[1] + R[2]*v[2], y = R[3]*v[0] + R[4]*v[1] + R[5]*v[2], z = R[6]*v[0] + R[7]*v[1] + R[8]*v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
/* r = R^T * v */
static inline void mulmtv3(double r[3], const double R[9], const double v[3]) {
    double x = R[0]*v[0] + R[3]*v[1] + R[6]*v[2], y = R[1]*v[0] + R[4]*v[1] + R[7]*v[2], z = R[2]*v[0] + R[5]*v[1] + R[8]*v[2];
    r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmm3(double r[9], const double A[9], const double B[9]) {
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3*i+j] = A[3*i]*B[j] + A[3*i+1]*B[3+j] + A[3*i+2]*B[6+j];
    memcpy(r, t, sizeof t);
}
static inline void quat_mul(double r[4], const double a[4], const double b[4]) {
    double w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
    double x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
    double y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
    double z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
    r[0] = w; r[1] = x; r[2] = y; r[3] = z;
}
static inline void quat_normalize(double q[4]) {
    double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static inline void quat_to_mat(double R[9], const double q[4]) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2*(y*y + z*z); R[1] = 2*(x*y - w*z); R[2] = 2*(x*z + w*y);
    R[3] = 2*(x*y + w*z); R[4] = 1 - 2*(x*x + z*z); R[5] = 2*(y*z - w*x);
    R[6] = 2*(x*z - w*y); R[7] = 2*(y*z + w*x); R[8] = 1 - 2*(x*x + y*y);
}
static inline void axis_quat(double q[4], const double axis[3], double ang) {
    double s = sin(0.5 * ang);
    q[0] = cos(0.5 * ang); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}

/* dofs that move each body (fixed topology) */
static const int k_body_ndof[NB] = {0, 5, 5, 6, 6, 6, 6, 6};
static const int k_body_dofs[NB][6] = {
    {0}, {0,1,2,3,4,0}, {0,1,2,3,4,0}, {0,1,2,3,4,5}, {0,1,2,3,4,5}, {0,1,2,3,4,6}, {0,1,2,3,4,6}, {7,8,9,10,11,12}};

#endif
