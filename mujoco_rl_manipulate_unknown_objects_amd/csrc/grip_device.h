// grip_device.h -- device-side model layout and small fp32 vector algebra for the gfx950 kernels.
//
// One environment is worked on by 16 lanes (two envs and their clone lanes per wavefront: grip_physics.h). The articulated system of
// xmls/<object>_env.xml (reference robot xml :58-99) is folded, at model-load time on the host,
// into FOUR rigid groups -- G = ee + welded base, L = left knuckle + welded finger, R = right
// knuckle + finger, O = object -- because welded bodies never move relative to each other.
// This is a restructuring of the same mechanics the reference gets from MuJoCo, not a port.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#define GN_GEOM 7          // 0 floor, 1 base, 2 lk, 3 lf, 4 rk, 5 rf, 6 object
#define GN_HULL 6
#define GN_PAIR_MAX 16
#define G_MAXC 14          // contact slots per env (GRIP_MAXCON)
#define WAVE 64
#define RMAXPL 2560        // hull face planes of one env the observation kernel keeps in LDS (40 KiB); bread_crumb_env has 2458

enum { GRP_G = 0, GRP_L = 1, GRP_R = 2, GRP_O = 3, GRP_WORLD = 4 };

struct DevModel {
    float timestep, gravity_z, impratio, tolerance;
    int iterations;
    float margin;
    float k_con, b_con, k_lim, b_lim;          // spring/damper of the reference acceleration (solref)
    float solimp[5], lim_solimp[5];
    float meaninertia;
    // rigid groups
    float grp_mass[4];
    float grp_com[4][3];                        // COM in the group frame
    float grp_inertia[4][6];                    // about the COM, group frame: xx yy zz xy xz yz
    float ee_pos0[3];
    float base_pos[3], base_R[9];               // base frame in the ee frame
    float kn_pos[2][3], kn_R[2][9];             // knuckle frames in the base frame (before the hinge)
    float fin_pos[2][3], fin_R[2][9];           // finger frames in the knuckle frames
    float armature[13], damping[13];
    float range[7][2], gear[7], ctrlrange[7][2];
    float dof_invweight0[7];
    float qpos0[14];
    // geoms
    float geom_center[GN_GEOM][3];              // in the geom's body frame
    float geom_rbound[GN_GEOM];
    float geom_friction[GN_GEOM][2];            // sliding, torsional
    float geom_invweight[GN_GEOM];              // body_invweight0[geom body][0]
    int geom_group[GN_GEOM];                    // rigid group of the geom's body (GRP_WORLD for the floor)
    int hull_vadr[GN_HULL], hull_vnum[GN_HULL];
    int npair;
    int pairs[GN_PAIR_MAX][2];
    int coop_items[16][2];                      // narrow-phase item of each of an env's 16 lanes, per round (-1 = none)
    // hull tables as one word blob (staged into LDS by every launch): [nvert][4] f32 vertices in the body
    // frame | u16 CSR offsets over all hull vertices | u16 neighbour ids (local) | u16 cube-map start vertices
    const unsigned *hull_blob;
    int hull_words, hull_off_nadr, hull_off_nbr, hull_off_lut;
    // camera / rendering
    float cam_pos[3], cam_R[9], cam_fovy;
    float znear, zfar;
    float geom_rgba[GN_GEOM][4];
    float floor_rgb[6], sky_rgb[6];
    float light_dir[2][3];
    // materials and lights of MuJoCo's fixed-function lighting (robot xml :29-34, :50-51): per geom specular / shininess / emission of its material;
    // per light position, directional flag, diffuse / specular / ambient intensity, spot cutoff (cosine) and exponent; the headlight's ambient /
    // diffuse / specular. (reflectance is 0 for every material a geom of these scenes uses)
    float geom_material[GN_GEOM][3];
    float light_pos[2][3]; int light_directional[2];
    float light_params[2][5];                   // diffuse, specular, ambient, cos(cutoff), exponent
    float headlight[3];
    int hull_padr[GN_HULL], hull_pnum[GN_HULL];
    float hull_aabb[GN_HULL][6];                // min xyz, max xyz of the hull's vertices in its geom frame (observation kernel: ray cull)
    const float *hull_planes;                   // [nplane][4] n.x <= d, body frame
    // face polygons (observation kernel's rasteriser): plane j's corners are hull_loops[hull_ladr[j] .. hull_ladr[j + 1]) -- vertex ids local to the plane's
    // hull, counter-clockwise seen from outside; an empty range = the plane is a duplicate of an earlier one of the same face
    const int *hull_ladr, *hull_loops;
};

struct DevConfig {
    int max_steps, time_horizon, include_roll, full_observation, her_buffer, auto_reset;
    float max_translation, max_rotation, pos_tolerance, grasp_tolerance;
    float dir_x, dir_y;
    int state_half;      // qpos / qvel / ctrl are stored as IEEE half in HBM (grip_batch_set_state_storage); arithmetic stays fp32
};

// per-batch arguments of the observation kernel (one record per batch of a set; a single batch is a set of one). The model is
// held BY VALUE: its constants are then read through the kernel's const __restrict__ record pointer with a uniform index,
// i.e. by scalar loads (through a pointer stored in the record they would be per-lane flat loads and double the VGPR count).
struct RenderGroup { DevModel m; DevConfig cfg; const float *qpos; const int *pad_grasp, *pad_pher; int n, env0; };

// A pointer read from a record in memory has no known address space, and every access through it becomes a flat_* instruction
// with a 64-bit VGPR address. The state arrays are global memory: GPTR(T, p) says so at the point of use.
#define GPTR(T, p) ((__attribute__((address_space(1))) T *)(p))

// one word of the SoA state arrays: fp32, or IEEE half (round to nearest even on store) when the batch keeps qpos / qvel /
// ctrl in half precision (BASELINE.json configs[4]); `half` is uniform over the launch
__device__ __forceinline__ float ld_word(const float *base, size_t idx, int half) {
    return half ? __half2float(__ushort_as_half(GPTR(const unsigned short, base)[idx])) : GPTR(const float, base)[idx];
}
__device__ __forceinline__ void st_word(float *base, size_t idx, float v, int half) {
    if (half) GPTR(unsigned short, base)[idx] = __half_as_ushort(__float2half_rn(v)); else GPTR(float, base)[idx] = v;
}

// ---------------------------------------------------------------- fp32 helpers
struct V3 { float x, y, z; };
struct M3 { float m[9]; };                      // row-major

#define DEVI __device__ __forceinline__

DEVI V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEVI V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
DEVI V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
DEVI V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
DEVI V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
DEVI V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
DEVI float dot(V3 a, V3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
DEVI V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEVI float norm(V3 a) { return sqrtf(dot(a, a)); }
// Hides a value's provenance from the optimiser at zero run-time cost. Used at the top of loop bodies so that LLVM's
// (register-pressure-blind) loop-invariant code motion does not hoist whole Jacobian rows out of the solver loops
// and keep them live across everything else.
DEVI void opaque(float &x) { asm volatile("" : "+v"(x)); }
DEVI float rcp(float x) { return __builtin_amdgcn_rcpf(x); }          // v_rcp_f32, 1 ulp
// sin and cos of a joint angle (|x| of a few pi at most: hinge ranges are +-pi, qpos never winds up). Quadrant reduction with
// a three-term Cody-Waite pi/2 and the Cephes single-precision kernels on [-pi/4, pi/4]: within 1-2 ulp there, ~40
// instructions instead of the ~250 of the library's full-range sincosf (Payne-Hanek path included).
DEVI void sincos_joint(float x, float &s, float &c) {
    float q = rintf(x * 0.63661977236758134308f);                      // nearest multiple of pi/2
    float r = fmaf(q, -1.5703125f, x);                                   // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188216e-8
    r = fmaf(q, -4.837512969970703125e-4f, r);
    r = fmaf(q, -7.54978995489188216e-8f, r);
    float z = r * r;
    float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
    int n = (int)q;
    float ss = (n & 1) ? cp : sp, cc = (n & 1) ? sp : cp;
    s = (n & 2) ? -ss : ss;
    c = ((n + 1) & 2) ? -cc : cc;
}
DEVI V3 normalized(V3 a) {
    float d = dot(a, a);
    if (d < 1e-38f) return v3(1.f, 0.f, 0.f);
    return a * rsqrtf(d);
}
DEVI V3 ldv(const float *p) { return v3(p[0], p[1], p[2]); }
DEVI V3 mulv(const M3 &R, V3 v) {
    return v3(R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z,
              R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z);
}
DEVI V3 multv(const M3 &R, V3 v) {               // R^T v
    return v3(R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z,
              R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z);
}
DEVI M3 mulm(const M3 &A, const M3 &B) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return r;
}
DEVI M3 ldm(const float *p) { M3 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.m[i] = p[i];
    return r; }
DEVI V3 col(const M3 &R, int j) { return v3(R.m[j], R.m[3 + j], R.m[6 + j]); }
DEVI M3 quat_mat(float w, float x, float y, float z) {
    M3 R;
    R.m[0] = 1 - 2 * (y * y + z * z); R.m[1] = 2 * (x * y - w * z); R.m[2] = 2 * (x * z + w * y);
    R.m[3] = 2 * (x * y + w * z); R.m[4] = 1 - 2 * (x * x + z * z); R.m[5] = 2 * (y * z - w * x);
    R.m[6] = 2 * (x * z - w * y); R.m[7] = 2 * (y * z + w * x); R.m[8] = 1 - 2 * (x * x + y * y);
    return R;
}
// symmetric 3x3 (xx yy zz xy xz yz) times vector
DEVI V3 symv(const float *s, V3 v) {
    return v3(s[0] * v.x + s[3] * v.y + s[4] * v.z, s[3] * v.x + s[1] * v.y + s[5] * v.z, s[4] * v.x + s[5] * v.y + s[2] * v.z);
}
// world inertia R S R^T of a symmetric local tensor, returned symmetric-packed
DEVI void rot_sym(const M3 &R, const float *s, float *out) {
    // T = R * S
    float T[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float a = R.m[3 * i], b = R.m[3 * i + 1], c = R.m[3 * i + 2];
        T[3 * i] = a * s[0] + b * s[3] + c * s[4];
        T[3 * i + 1] = a * s[3] + b * s[1] + c * s[5];
        T[3 * i + 2] = a * s[4] + b * s[5] + c * s[2];
    }
    out[0] = T[0] * R.m[0] + T[1] * R.m[1] + T[2] * R.m[2];
    out[1] = T[3] * R.m[3] + T[4] * R.m[4] + T[5] * R.m[5];
    out[2] = T[6] * R.m[6] + T[7] * R.m[7] + T[8] * R.m[8];
    out[3] = T[0] * R.m[3] + T[1] * R.m[4] + T[2] * R.m[5];
    out[4] = T[0] * R.m[6] + T[1] * R.m[7] + T[2] * R.m[8];
    out[5] = T[3] * R.m[6] + T[4] * R.m[7] + T[5] * R.m[8];
}
