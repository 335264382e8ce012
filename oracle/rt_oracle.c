/*
 * rt_oracle.c -- CPU restatement of the Soulztorm/Raytracer per-pixel tracing path.
 *
 * TEST INFRASTRUCTURE ONLY (see rt_oracle.h).  Citations are relative to /root/reference.
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fPIC -shared -pthread (oracle/Makefile).
 * No FMA is ever emitted (x86-64 baseline has none and contraction is off), every float op is one
 * IEEE binary32 round-to-nearest-even operation in the order glm 0.9.9.9 evaluates it.
 */
#include "rt_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * vec3 in glm's evaluation order
 * ---------------------------------------------------------------------------------------------- */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v_add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v_sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v_mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v_div(v3 a, v3 b) { return V(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 v_scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 v_divs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }   /* type_vec3.inl:717-723 */
static inline v3 v_adds(v3 a, float s) { return V(a.x + s, a.y + s, a.z + s); }
static inline v3 v_neg(v3 a) { return V(-a.x, -a.y, -a.z); }
/* glm::dot(vec3): tmp = a*b; tmp.x + tmp.y + tmp.z  (detail/func_geometric.inl:48-55) */
static inline float v_dot(v3 a, v3 b) { v3 t = v_mul(a, b); return (t.x + t.y) + t.z; }
/* glm::cross (detail/func_geometric.inl:68-79) */
static inline v3 v_cross(v3 x, v3 y)
{
    return V(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm::normalize = v * inversesqrt(dot(v,v)), inversesqrt = 1/sqrt (func_geometric.inl:82-90,
   func_exponential.inl:134-139) */
static inline v3 v_normalize(v3 v) { return v_scale(v, 1.0f / sqrtf(v_dot(v, v))); }
/* glm::reflect = I - N * dot(N, I) * 2 (func_geometric.inl:104-110) */
static inline v3 v_reflect(v3 I, v3 N) { return v_sub(I, v_scale(v_scale(N, v_dot(N, I)), 2.0f)); }
/* glm::mix(x, y, a) = x * (1 - a) + y * a (func_common.inl:104-112) */
static inline v3 v_mix(v3 x, v3 y, float a) { return v_add(v_scale(x, 1.0f - a), v_scale(y, a)); }
/* glm::min(x,y) = (y < x) ? y : x ; glm::max(x,y) = (x < y) ? y : x (func_common.inl:17-30);
   std::min / std::max have the same select form. */
static inline float f_min(float x, float y) { return (y < x) ? y : x; }
static inline float f_max(float x, float y) { return (x < y) ? y : x; }

/* ------------------------------------------------------------------------------------------------
 * pcg32 = setseq_xsh_rr_64_32 with the default stream (pcg_random.hpp:158-159,413-437,484-489,845-873)
 * ---------------------------------------------------------------------------------------------- */
#define PCG_MULT 6364136223846793005ULL
#define PCG_INC  1442695040888963407ULL
#define PCG_DEFAULT_SEED 0xcafef00dd15ea5e5ULL

uint64_t orc_pcg32_seed(uint64_t seed) { return (seed + PCG_INC) * PCG_MULT + PCG_INC; }

uint32_t orc_pcg32_next(uint64_t* state)
{
    uint64_t old = *state;                       /* output_previous: permute the OLD state */
    *state = old * PCG_MULT + PCG_INC;
    uint32_t rot = (uint32_t)(old >> 59);
    old ^= old >> 18;
    uint32_t x = (uint32_t)(old >> 27);
    return (x >> rot) | (x << ((32u - rot) & 31u));
}

/* Walnut::Random::Float (Random.h:11,35-38): (float)u32 / (float)UINT32_MAX, the divisor rounds to 2^32. */
float orc_u32_to_float(uint32_t u) { return (float)u / 4294967296.0f; }

/* Per-(pixel, frame) seed of the parity RNG: splitmix64 finaliser over (frame:pixel) offset by the
   global seed; the result seeds a pcg32 exactly as the engine constructor would. */
uint64_t orc_pixel_seed(uint32_t pixel, uint32_t frame, uint32_t seed)
{
    uint64_t z = (((uint64_t)frame << 32) | (uint64_t)pixel) + (uint64_t)seed * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return orc_pcg32_seed(z);
}

static uint64_t g_serial_state = 0;
static int g_serial_init = 0;
void orc_serial_rng_reset(void) { g_serial_state = orc_pcg32_seed(PCG_DEFAULT_SEED); g_serial_init = 1; }

/* ------------------------------------------------------------------------------------------------
 * Portable sincos / pow(x, 1/2.4): only + - * / and int<->float conversions in a fixed order, so a
 * device without FMA contraction reproduces them bit for bit.
 * ---------------------------------------------------------------------------------------------- */
void orc_sincos(float a, float* s_out, float* c_out)
{
    /* quadrant k = round(a * 2/pi), a >= 0 */
    int k = (int)(a * 0.636619772f + 0.5f);
    float kf = (float)k;
    /* Cody-Waite: pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188216e-8 */
    float r = a - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float z = r * r;
    /* sin(r) on [-pi/4, pi/4] */
    float ps = -1.9515295891e-4f * z;
    ps = ps + 8.3321608736e-3f;
    ps = ps * z;
    ps = ps - 1.6666654611e-1f;
    ps = ps * z;
    ps = ps * r;
    float sn = ps + r;
    /* cos(r) on [-pi/4, pi/4] */
    float pc = 2.443315711809948e-5f * z;
    pc = pc - 1.388731625493765e-3f;
    pc = pc * z;
    pc = pc + 4.166664568298827e-2f;
    pc = pc * z;
    pc = pc * z;
    float hz = 0.5f * z;
    float cs = pc - hz;
    cs = cs + 1.0f;
    switch (k & 3) {
    case 0:  *s_out = sn;  *c_out = cs;  break;
    case 1:  *s_out = cs;  *c_out = -sn; break;
    case 2:  *s_out = -sn; *c_out = -cs; break;
    default: *s_out = -cs; *c_out = sn;  break;
    }
}

static inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* x^(1/2.4) for x in [0, 1]; the value for x below the sRGB toe (0.0031308) only needs to be finite. */
float orc_pow_inv24(float x)
{
    uint32_t ux = float_to_bits(x);
    int e = (int)(ux >> 23) - 127;
    float m = bits_to_float((ux & 0x007FFFFFu) | 0x3F800000u);      /* [1, 2) */
    if (m > 1.41421354f) { m = m * 0.5f; e = e + 1; }                /* [sqrt(.5), sqrt(2)] */
    /* log2(m) = (2/ln2) * atanh(s), s = (m-1)/(m+1) */
    float s = (m - 1.0f) / (m + 1.0f);
    float s2 = s * s;
    float p = 0.111111111f * s2;
    p = p + 0.142857143f;
    p = p * s2;
    p = p + 0.2f;
    p = p * s2;
    p = p + 0.333333333f;
    p = p * s2;
    p = p * s;
    p = p + s;
    float l2 = p * 2.88539008f;                                      /* 2/ln 2 */
    float y = ((float)e + l2) * 0.416666667f;                        /* / 2.4 */
    /* 2^y, y in about [-53, 0.2] */
    int n = (int)(y + 1024.5f) - 1024;                               /* round to nearest */
    float g = y - (float)n;                                          /* [-0.5, 0.5] */
    float t = g * 0.693147181f;
    float q = 1.98412698e-4f * t;                                    /* 1/5040 */
    q = q + 1.38888889e-3f;                                          /* 1/720  */
    q = q * t;
    q = q + 8.33333333e-3f;                                          /* 1/120  */
    q = q * t;
    q = q + 4.16666667e-2f;                                          /* 1/24   */
    q = q * t;
    q = q + 1.66666667e-1f;                                          /* 1/6    */
    q = q * t;
    q = q + 0.5f;
    q = q * t;
    q = q + 1.0f;
    q = q * t;
    q = q + 1.0f;
    if (n < -126) n = -126;
    return q * bits_to_float((uint32_t)(n + 127) << 23);
}

/* ------------------------------------------------------------------------------------------------
 * Tone map: Util::ACESFilm, LessThan, LinearToSRGB (Renderer.h:57-98)
 * ---------------------------------------------------------------------------------------------- */
static inline float f_clamp01(float x) { return f_min(f_max(x, 0.0f), 1.0f); }   /* func_common.inl:240-246 */

static inline float aces1(float x)
{
    /* clamp((x*(a*x+b)) / (x*(c*x+d)+e), 0, 1) */
    float num = x * (2.51f * x + 0.03f);
    float den = x * (2.43f * x + 0.59f) + 0.14f;
    return f_clamp01(num / den);
}

static inline float srgb1(float c, int portable_pow)
{
    c = f_clamp01(c);
    float pw = portable_pow ? orc_pow_inv24(c) : powf(c, 1.0f / 2.4f);
    float hi = pw * 1.055f - 0.055f;
    float lo = c * 12.92f;
    float a = (c < 0.0031308f) ? 1.0f : 0.0f;
    return hi * (1.0f - a) + lo * a;              /* glm::mix with a vec3 interpolator (func_common.inl:81-89) */
}

void orc_tonemap(const float in[3], int use_ace, int portable_pow, float out[3])
{
    for (int i = 0; i < 3; ++i) {
        float c = in[i];
        if (use_ace) c = srgb1(aces1(c), portable_pow);
        out[i] = c;
    }
}

/* FNV-1a folded over 32-bit words (the form the reference-run checksums were recorded in). */
uint32_t orc_fnv1a(const void* data, uint64_t nbytes)
{
    const uint32_t* p = (const uint32_t*)data;
    uint32_t h = 2166136261u;
    for (uint64_t i = 0; i < nbytes / 4; ++i) { h ^= p[i]; h *= 16777619u; }
    return h;
}

/* ------------------------------------------------------------------------------------------------
 * Camera (Camera.cpp:11-19, 122-150) and the glm matrix routines it calls
 * ---------------------------------------------------------------------------------------------- */
#define M(a, c, r) ((a)[(c) * 4 + (r)])

/* detail/func_matrix.inl:347-405 (compute_inverse<4,4>) */
static void mat4_inverse(const float* m, float* out)
{
    float Coef00 = M(m,2,2) * M(m,3,3) - M(m,3,2) * M(m,2,3);
    float Coef02 = M(m,1,2) * M(m,3,3) - M(m,3,2) * M(m,1,3);
    float Coef03 = M(m,1,2) * M(m,2,3) - M(m,2,2) * M(m,1,3);
    float Coef04 = M(m,2,1) * M(m,3,3) - M(m,3,1) * M(m,2,3);
    float Coef06 = M(m,1,1) * M(m,3,3) - M(m,3,1) * M(m,1,3);
    float Coef07 = M(m,1,1) * M(m,2,3) - M(m,2,1) * M(m,1,3);
    float Coef08 = M(m,2,1) * M(m,3,2) - M(m,3,1) * M(m,2,2);
    float Coef10 = M(m,1,1) * M(m,3,2) - M(m,3,1) * M(m,1,2);
    float Coef11 = M(m,1,1) * M(m,2,2) - M(m,2,1) * M(m,1,2);
    float Coef12 = M(m,2,0) * M(m,3,3) - M(m,3,0) * M(m,2,3);
    float Coef14 = M(m,1,0) * M(m,3,3) - M(m,3,0) * M(m,1,3);
    float Coef15 = M(m,1,0) * M(m,2,3) - M(m,2,0) * M(m,1,3);
    float Coef16 = M(m,2,0) * M(m,3,2) - M(m,3,0) * M(m,2,2);
    float Coef18 = M(m,1,0) * M(m,3,2) - M(m,3,0) * M(m,1,2);
    float Coef19 = M(m,1,0) * M(m,2,2) - M(m,2,0) * M(m,1,2);
    float Coef20 = M(m,2,0) * M(m,3,1) - M(m,3,0) * M(m,2,1);
    float Coef22 = M(m,1,0) * M(m,3,1) - M(m,3,0) * M(m,1,1);
    float Coef23 = M(m,1,0) * M(m,2,1) - M(m,2,0) * M(m,1,1);

    float Fac0[4] = { Coef00, Coef00, Coef02, Coef03 };
    float Fac1[4] = { Coef04, Coef04, Coef06, Coef07 };
    float Fac2[4] = { Coef08, Coef08, Coef10, Coef11 };
    float Fac3[4] = { Coef12, Coef12, Coef14, Coef15 };
    float Fac4[4] = { Coef16, Coef16, Coef18, Coef19 };
    float Fac5[4] = { Coef20, Coef20, Coef22, Coef23 };
    float Vec0[4] = { M(m,1,0), M(m,0,0), M(m,0,0), M(m,0,0) };
    float Vec1[4] = { M(m,1,1), M(m,0,1), M(m,0,1), M(m,0,1) };
    float Vec2[4] = { M(m,1,2), M(m,0,2), M(m,0,2), M(m,0,2) };
    float Vec3[4] = { M(m,1,3), M(m,0,3), M(m,0,3), M(m,0,3) };
    static const float SignA[4] = { +1.0f, -1.0f, +1.0f, -1.0f };
    static const float SignB[4] = { -1.0f, +1.0f, -1.0f, +1.0f };
    float inv[16];
    for (int i = 0; i < 4; ++i) {
        float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
        float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
        float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
        float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
        M(inv,0,i) = Inv0 * SignA[i];
        M(inv,1,i) = Inv1 * SignB[i];
        M(inv,2,i) = Inv2 * SignA[i];
        M(inv,3,i) = Inv3 * SignB[i];
    }
    float d0 = M(m,0,0) * M(inv,0,0);
    float d1 = M(m,0,1) * M(inv,1,0);
    float d2 = M(m,0,2) * M(inv,2,0);
    float d3 = M(m,0,3) * M(inv,3,0);
    float Dot1 = (d0 + d1) + (d2 + d3);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * OneOverDeterminant;
}

/* mat4 * vec4 (detail/type_mat4x4.inl:561-572): (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
static inline void mat4_mul_vec4(const float* m, const float v[4], float out[4])
{
    for (int r = 0; r < 4; ++r) {
        float Mul0 = M(m,0,r) * v[0];
        float Mul1 = M(m,1,r) * v[1];
        float Add0 = Mul0 + Mul1;
        float Mul2 = M(m,2,r) * v[2];
        float Mul3 = M(m,3,r) * v[3];
        float Add1 = Mul2 + Mul3;
        out[r] = Add0 + Add1;
    }
}

/* the vector/matrix helpers themselves, exported for known-answer tests against real glm */
void orc_glm_normalize(const float a[3], float o[3]) { v3 r = v_normalize(V(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void orc_glm_reflect(const float i[3], const float n[3], float o[3]) { v3 r = v_reflect(V(i[0], i[1], i[2]), V(n[0], n[1], n[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void orc_glm_mix(const float x[3], const float y[3], float a, float o[3]) { v3 r = v_mix(V(x[0], x[1], x[2]), V(y[0], y[1], y[2]), a); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
float orc_glm_dot(const float a[3], const float b[3]) { return v_dot(V(a[0], a[1], a[2]), V(b[0], b[1], b[2])); }
void orc_glm_cross(const float a[3], const float b[3], float o[3]) { v3 r = v_cross(V(a[0], a[1], a[2]), V(b[0], b[1], b[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; }
void orc_glm_inverse(const float m[16], float o[16]) { mat4_inverse(m, o); }
void orc_glm_mat_vec(const float m[16], const float v[4], float o[4]) { mat4_mul_vec4(m, v, o); }

void orc_camera_forward_default(float fwd[3])
{
    /* Camera.cpp:14: glm::normalize(glm::vec3(0, -0.1, -1)); the double literals are cast to float */
    v3 f = v_normalize(V((float)0, (float)-0.1, (float)-1));
    fwd[0] = f.x; fwd[1] = f.y; fwd[2] = f.z;
}

void orc_camera_projection(float fov_deg, float w, float h, float znear, float zfar, float proj[16], float inv_proj[16])
{
    /* Camera.cpp:124: perspectiveFov(radians(fov), w, h, near, far) -> perspectiveFovRH_NO
       (ext/matrix_clip_space.inl:372-389); radians = deg * 0.0174532925... (func_trigonometric.inl) */
    float rad = fov_deg * (float)0.01745329251994329576923690768489;
    float hh = cosf(0.5f * rad) / sinf(0.5f * rad);
    float ww = hh * h / w;
    memset(proj, 0, 16 * sizeof(float));
    M(proj,0,0) = ww;
    M(proj,1,1) = hh;
    M(proj,2,2) = -(zfar + znear) / (zfar - znear);
    M(proj,2,3) = -1.0f;
    M(proj,3,2) = -(2.0f * zfar * znear) / (zfar - znear);
    mat4_inverse(proj, inv_proj);
}

void orc_camera_view(const float pos[3], const float fwd[3], float view[16], float inv_view[16])
{
    /* Camera.cpp:130: lookAt(pos, pos + fwd, (0,1,0)) -> lookAtRH (ext/matrix_transform.inl:99-119) */
    v3 eye = V(pos[0], pos[1], pos[2]);
    v3 center = v_add(eye, V(fwd[0], fwd[1], fwd[2]));
    v3 up = V(0.0f, 1.0f, 0.0f);
    v3 f = v_normalize(v_sub(center, eye));
    v3 s = v_normalize(v_cross(f, up));
    v3 u = v_cross(s, f);
    memset(view, 0, 16 * sizeof(float));
    M(view,0,0) = s.x;  M(view,1,0) = s.y;  M(view,2,0) = s.z;
    M(view,0,1) = u.x;  M(view,1,1) = u.y;  M(view,2,1) = u.z;
    M(view,0,2) = -f.x; M(view,1,2) = -f.y; M(view,2,2) = -f.z;
    M(view,3,0) = -v_dot(s, eye);
    M(view,3,1) = -v_dot(u, eye);
    M(view,3,2) = v_dot(f, eye);
    M(view,3,3) = 1.0f;
    mat4_inverse(view, inv_view);
}

void orc_camera_ray_dirs(const float inv_proj[16], const float inv_view[16], uint32_t w, uint32_t h, float* dirs3)
{
    /* Camera.cpp:138-149 */
    for (uint32_t y = 0; y < h; ++y) {
        for (uint32_t x = 0; x < w; ++x) {
            float cx = (float)x / (float)w;
            float cy = (float)y / (float)h;
            cx = cx * 2.0f - 1.0f;
            cy = cy * 2.0f - 1.0f;
            float in4[4] = { cx, cy, 1.0f, 1.0f }, target[4];
            mat4_mul_vec4(inv_proj, in4, target);
            v3 t3 = v_divs(V(target[0], target[1], target[2]), target[3]);
            v3 n = v_normalize(t3);
            float n4[4] = { n.x, n.y, n.z, 0.0f }, rd[4];
            mat4_mul_vec4(inv_view, n4, rd);
            v3 d = v_normalize(V(rd[0], rd[1], rd[2]));
            float* o = dirs3 + 3 * ((size_t)x + (size_t)y * w);
            o[0] = d.x; o[1] = d.y; o[2] = d.z;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Scene
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    float c[3], e[3];         /* boundingBox center / extends (KDTreeStructs.h:44-48) */
    int32_t left, right;      /* -1 = none */
    int32_t first, count;     /* the node's own triangle list (interior nodes keep theirs) */
    int32_t depth;            /* KDTreeNode::id as the stackless builder uses it */
} kd_node;

struct orc_scene {
    orc_sphere* spheres;   uint32_t n_spheres;
    orc_material* mats;    uint32_t n_mats;
    float* verts;          uint32_t n_verts;
    uint32_t* tris;        uint32_t n_tris;
    float* normals;        /* 9 per triangle */
    uint32_t* tri_mat;
    kd_node* nodes;        uint32_t n_nodes;
    int32_t* leaf_idx;     uint32_t n_leaf_idx;
    uint32_t kd_depth;
};

orc_scene* orc_scene_create(void) { return (orc_scene*)calloc(1, sizeof(orc_scene)); }

static void free_kd(orc_scene* s)
{
    free(s->nodes); free(s->leaf_idx);
    s->nodes = NULL; s->leaf_idx = NULL; s->n_nodes = 0; s->n_leaf_idx = 0; s->kd_depth = 0;
}

void orc_scene_destroy(orc_scene* s)
{
    if (!s) return;
    free(s->spheres); free(s->mats); free(s->verts); free(s->tris); free(s->normals); free(s->tri_mat);
    free_kd(s);
    free(s);
}

static void* dup_mem(const void* p, size_t n) { void* q = malloc(n ? n : 1); if (q && n) memcpy(q, p, n); return q; }

int orc_scene_set_spheres(orc_scene* s, const orc_sphere* sp, uint32_t n)
{
    free(s->spheres); s->spheres = (orc_sphere*)dup_mem(sp, (size_t)n * sizeof(*sp)); s->n_spheres = n; return 0;
}
int orc_scene_set_materials(orc_scene* s, const orc_material* m, uint32_t n)
{
    free(s->mats); s->mats = (orc_material*)dup_mem(m, (size_t)n * sizeof(*m)); s->n_mats = n; return 0;
}
int orc_scene_set_mesh(orc_scene* s, const float* verts, uint32_t nverts, const uint32_t* tri_idx,
                       const float* normals9, const uint32_t* tri_mat, uint32_t ntris)
{
    free(s->verts); free(s->tris); free(s->normals); free(s->tri_mat); free_kd(s);
    s->verts = (float*)dup_mem(verts, (size_t)nverts * 12); s->n_verts = nverts;
    s->tris = (uint32_t*)dup_mem(tri_idx, (size_t)ntris * 12); s->n_tris = ntris;
    s->normals = (float*)dup_mem(normals9, (size_t)ntris * 36);
    s->tri_mat = (uint32_t*)dup_mem(tri_mat, (size_t)ntris * 4);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * KD build: naive restatement of KDTreeCPU::KDTreeCPU + constructTreeStackless
 * (KDTreeCPU.cpp:17-52, 93-132, 159-191, 424-632; constants KDTreeCPU.h:14-17)
 * ---------------------------------------------------------------------------------------------- */
#define NUM_TRIS_PER_NODE 10
#define MAX_DEPTH 40

typedef struct bnode {
    float c[3], e[3];
    struct bnode *left, *right;
    int num_tris; int* tri_indices; int id;
} bnode;

static inline v3 vert_at(const orc_scene* s, uint32_t i) { const float* p = s->verts + 3 * (size_t)i; return V(p[0], p[1], p[2]); }

/* getMinTriValue / getMaxTriValue (KDTreeCPU.cpp:98-132): the index triple goes through a glm::vec3
   (uint -> float -> int), so indices above 2^24 are rounded -- reproduced, not fixed. */
static inline void tri_axis_vals(const orc_scene* s, int tri, int axis, float* mn, float* mx)
{
    const uint32_t* t = s->tris + 3 * (size_t)tri;
    float a = s->verts[3 * (size_t)(int)(float)t[0] + axis];
    float b = s->verts[3 * (size_t)(int)(float)t[1] + axis];
    float c = s->verts[3 * (size_t)(int)(float)t[2] + axis];
    *mn = (a < b && a < c) ? a : (b < c ? b : c);
    *mx = (a > b && a > c) ? a : (b > c ? b : c);
}

static bnode* bnode_new(void) { return (bnode*)calloc(1, sizeof(bnode)); }

static bnode* build_tree(const orc_scene* s)
{
    /* computeTightFittingBoundingBox over all vertices (KDTreeCPU.cpp:159-191) */
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint32_t i = 0; i < s->n_verts; ++i)
        for (int k = 0; k < 3; ++k) {
            float v = s->verts[3 * (size_t)i + k];
            if (v < mn[k]) mn[k] = v;
            if (v > mx[k]) mx[k] = v;
        }
    bnode* root = bnode_new();
    for (int k = 0; k < 3; ++k) { root->c[k] = 0.5f * (mn[k] + mx[k]); root->e[k] = 0.5f * (mx[k] - mn[k]); }
    root->num_tris = (int)s->n_tris;
    root->tri_indices = (int*)malloc(sizeof(int) * (s->n_tris ? s->n_tris : 1));
    for (uint32_t i = 0; i < s->n_tris; ++i) root->tri_indices[i] = (int)i;
    root->id = 0;
    if (root->num_tris <= NUM_TRIS_PER_NODE) return root;

    size_t cap = 64, top = 0;
    bnode** stack = (bnode**)malloc(cap * sizeof(bnode*));
    stack[top++] = root;
    while (top) {
        bnode* node = stack[--top];
        int depth = node->id;
        if (node->num_tris <= NUM_TRIS_PER_NODE || node->id >= MAX_DEPTH) continue;

        const float cost_traversal = 1.5f, cost_intersect = 1.0f, split_delta = 0.01f;
        int min_cost_side = 0;   /* the reference initialises it to the longest side; every sweep visits
                                    all three sides and the first finite cost replaces it */
        {
            /* getLongestBoundingBoxSide (KDTreeCPU.cpp:93-96) */
            min_cost_side = (node->e[0] > node->e[1] && node->e[0] > node->e[2]) ? 0 : (node->e[1] > node->e[2] ? 1 : 2);
        }
        float min_cost = FLT_MAX, min_splitting_plane = 0.0f;
        for (int side = 0; side < 3; ++side) {
            float min_s = node->c[side] - node->e[side];
            float max_s = node->c[side] + node->e[side];
            float s_length = max_s - min_s;
            for (float current_delta = split_delta; current_delta < 1.0f; current_delta += split_delta) {
                float current_plane = min_s + s_length * current_delta;
                float le[3] = { node->e[0], node->e[1], node->e[2] };
                float re[3] = { node->e[0], node->e[1], node->e[2] };
                le[side] = (current_plane - min_s) * 0.5f;
                re[side] = (max_s - current_plane) * 0.5f;
                float area_left = 8.0f * ((le[0] * le[1] + le[1] * le[2]) + le[0] * le[2]);
                float area_right = 8.0f * ((re[0] * re[1] + re[1] * re[2]) + re[0] * re[2]);
                uint32_t count_left = 0, count_right = 0;
                for (int i = 0; i < node->num_tris; ++i) {
                    float tmin, tmax;
                    tri_axis_vals(s, node->tri_indices[i], side, &tmin, &tmax);
                    if (tmin < current_plane) count_left++;
                    if (tmax >= current_plane) count_right++;
                }
                float cost = (cost_traversal + area_left * ((float)count_left) * cost_intersect)
                             + area_right * ((float)count_right) * cost_intersect;
                if (cost < min_cost) { min_cost = cost; min_splitting_plane = current_plane; min_cost_side = side; }
            }
        }
        int ax = min_cost_side;
        float min_before = node->c[ax] - node->e[ax];
        float max_before = node->c[ax] + node->e[ax];
        float median_val = min_splitting_plane;
        float left_extend = (median_val - min_before) * 0.5f;
        float left_center = median_val - left_extend;
        float right_extend = (max_before - median_val) * 0.5f;
        float right_center = median_val + right_extend;

        int* li = (int*)malloc(sizeof(int) * (size_t)node->num_tris);
        int* ri = (int*)malloc(sizeof(int) * (size_t)node->num_tris);
        int nl = 0, nr = 0;
        for (int i = 0; i < node->num_tris; ++i) {
            float tmin, tmax;
            tri_axis_vals(s, node->tri_indices[i], ax, &tmin, &tmax);
            if (tmin < median_val) li[nl++] = node->tri_indices[i];
            if (tmax >= median_val) ri[nr++] = node->tri_indices[i];
        }
        if (top + 2 > cap) { cap *= 2; stack = (bnode**)realloc(stack, cap * sizeof(bnode*)); }
        if (nl > 0) {
            bnode* c = bnode_new();
            memcpy(c->c, node->c, 12); memcpy(c->e, node->e, 12);
            c->c[ax] = left_center; c->e[ax] = left_extend;
            c->num_tris = nl; c->tri_indices = li; c->id = depth + 1;
            node->left = c; stack[top++] = c;
        } else free(li);
        if (nr > 0) {
            bnode* c = bnode_new();
            memcpy(c->c, node->c, 12); memcpy(c->e, node->e, 12);
            c->c[ax] = right_center; c->e[ax] = right_extend;
            c->num_tris = nr; c->tri_indices = ri; c->id = depth + 1;
            node->right = c; stack[top++] = c;
        } else free(ri);
    }
    free(stack);
    return root;
}

static void count_tree(const bnode* n, uint32_t* nodes, uint32_t* idx, uint32_t* depth)
{
    /* explicit stack: depth <= 41 but keep recursion out of it anyway */
    size_t cap = 128, top = 0;
    const bnode** st = (const bnode**)malloc(cap * sizeof(*st));
    st[top++] = n;
    while (top) {
        const bnode* c = st[--top];
        (*nodes)++; *idx += (uint32_t)c->num_tris;
        if ((uint32_t)c->id > *depth) *depth = (uint32_t)c->id;
        if (top + 2 > cap) { cap *= 2; st = (const bnode**)realloc(st, cap * sizeof(*st)); }
        if (c->right) st[top++] = c->right;
        if (c->left) st[top++] = c->left;
    }
    free(st);
}

static int32_t flatten_tree(bnode* n, orc_scene* s, uint32_t* next_node, uint32_t* next_idx)
{
    int32_t me = (int32_t)(*next_node)++;
    kd_node* k = &s->nodes[me];
    memcpy(k->c, n->c, 12); memcpy(k->e, n->e, 12);
    k->depth = n->id;
    k->first = (int32_t)*next_idx; k->count = n->num_tris;
    memcpy(s->leaf_idx + *next_idx, n->tri_indices, sizeof(int) * (size_t)n->num_tris);
    *next_idx += (uint32_t)n->num_tris;
    int32_t l = n->left ? flatten_tree(n->left, s, next_node, next_idx) : -1;
    int32_t r = n->right ? flatten_tree(n->right, s, next_node, next_idx) : -1;
    s->nodes[me].left = l; s->nodes[me].right = r;
    free(n->tri_indices); free(n);
    return me;
}

int orc_scene_build_kd(orc_scene* s)
{
    free_kd(s);
    if (!s->n_tris) return -1;
    bnode* root = build_tree(s);
    uint32_t nn = 0, ni = 0, depth = 0;
    count_tree(root, &nn, &ni, &depth);
    s->nodes = (kd_node*)calloc(nn, sizeof(kd_node));
    s->leaf_idx = (int32_t*)malloc(sizeof(int32_t) * (ni ? ni : 1));
    s->n_nodes = nn; s->n_leaf_idx = ni; s->kd_depth = depth;
    uint32_t a = 0, b = 0;
    flatten_tree(root, s, &a, &b);
    return 0;
}

int orc_scene_import_kd(orc_scene* s, uint32_t n_nodes, const float* center3, const float* extent3,
                        const int32_t* left, const int32_t* right, const int32_t* first, const int32_t* count,
                        uint32_t n_leaf_idx, const int32_t* leaf_idx)
{
    free_kd(s);
    s->nodes = (kd_node*)calloc(n_nodes ? n_nodes : 1, sizeof(kd_node));
    s->leaf_idx = (int32_t*)dup_mem(leaf_idx, sizeof(int32_t) * (size_t)n_leaf_idx);
    s->n_nodes = n_nodes; s->n_leaf_idx = n_leaf_idx;
    for (uint32_t i = 0; i < n_nodes; ++i) {
        memcpy(s->nodes[i].c, center3 + 3 * (size_t)i, 12);
        memcpy(s->nodes[i].e, extent3 + 3 * (size_t)i, 12);
        s->nodes[i].left = left[i]; s->nodes[i].right = right[i];
        s->nodes[i].first = first[i]; s->nodes[i].count = count[i];
        if ((left[i] >= 0 && (uint32_t)left[i] >= n_nodes) || (right[i] >= 0 && (uint32_t)right[i] >= n_nodes)) return -2;
        if (left[i] < 0 && right[i] < 0 && ((uint32_t)first[i] + (uint32_t)count[i] > n_leaf_idx)) return -3;
    }
    return 0;
}

int orc_scene_kd_sizes(const orc_scene* s, uint32_t* n_nodes, uint32_t* n_leaf_idx, uint32_t* depth)
{
    *n_nodes = s->n_nodes; *n_leaf_idx = s->n_leaf_idx; *depth = s->kd_depth; return 0;
}

int orc_scene_export_kd(const orc_scene* s, float* center3, float* extent3, int32_t* left, int32_t* right,
                        int32_t* first, int32_t* count, int32_t* leaf_idx)
{
    for (uint32_t i = 0; i < s->n_nodes; ++i) {
        memcpy(center3 + 3 * (size_t)i, s->nodes[i].c, 12);
        memcpy(extent3 + 3 * (size_t)i, s->nodes[i].e, 12);
        left[i] = s->nodes[i].left; right[i] = s->nodes[i].right;
        first[i] = s->nodes[i].first; count[i] = s->nodes[i].count;
    }
    memcpy(leaf_idx, s->leaf_idx, sizeof(int32_t) * (size_t)s->n_leaf_idx);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Intersections (KDAccel/Intersections.cpp:23-94)
 * ---------------------------------------------------------------------------------------------- */
static inline int aabb_intersect(const float c[3], const float e[3], v3 O, v3 invD, float* t_near)
{
    v3 l1 = v_mul(v_sub(V(c[0], c[1], c[2]), O), invD);
    v3 l2 = v_mul(V(e[0], e[1], e[2]), invD);
    v3 tMin = v_sub(l1, l2);
    v3 tMax = v_add(l1, l2);
    v3 t2 = V(f_max(tMin.x, tMax.x), f_max(tMin.y, tMax.y), f_max(tMin.z, tMax.z));
    float tFar = f_min(f_min(t2.x, t2.y), t2.z);
    if (tFar < 0.0f) return 0;
    v3 t1 = V(f_min(tMin.x, tMax.x), f_min(tMin.y, tMax.y), f_min(tMin.z, tMax.z));
    float tNear = f_max(f_max(t1.x, t1.y), t1.z);
    if (tNear > tFar) return 0;
    *t_near = tNear;
    return 1;
}

static inline int tri_intersect(v3 O, v3 D, v3 v0, v3 v1, v3 v2, float* t, float* u_out, float* v_out)
{
    v3 e1 = v_sub(v1, v0);
    v3 e2 = v_sub(v2, v0);
    v3 h = v_cross(D, e2);
    float a = v_dot(e1, h);
    if (a > -0.00001f && a < 0.00001f) return 0;
    float f = 1.0f / a;
    v3 s = v_sub(O, v0);
    float u = f * v_dot(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = v_cross(s, e1);
    float v = f * v_dot(D, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    *t = f * v_dot(e2, q);          /* written even when the final test below fails, as the reference does */
    if (*t > 0.00001f) { *u_out = u; *v_out = v; return 1; }
    return 0;
}

int orc_aabb(const float center[3], const float extent[3], const float o[3], const float inv_d[3], float* t_near)
{
    return aabb_intersect(center, extent, V(o[0], o[1], o[2]), V(inv_d[0], inv_d[1], inv_d[2]), t_near);
}
int orc_tri(const float o[3], const float d[3], const float v0[3], const float v1[3], const float v2[3],
            float* t, float* u, float* v)
{
    return tri_intersect(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), V(v0[0], v0[1], v0[2]),
                         V(v1[0], v1[1], v1[2]), V(v2[0], v2[1], v2[2]), t, u, v);
}

/* ------------------------------------------------------------------------------------------------
 * KD traversal: KDTreeCPU::intersect (KDTreeCPU.cpp:640-708), recursion kept as written
 * ---------------------------------------------------------------------------------------------- */
typedef struct { v3 O, D, invD; } ray_t;

static int kd_intersect_node(const orc_scene* s, int32_t ni, const ray_t* ray, float* t, uint32_t* tri_index,
                             float* u, float* v, orc_counters* cnt)
{
    const kd_node* node = &s->nodes[ni];
    float dist_aabb_near = FLT_MAX;
    cnt->nodes_visited++;
    int intersects_aabb = aabb_intersect(node->c, node->e, ray->O, ray->invD, &dist_aabb_near);
    if (dist_aabb_near > *t) return 0;
    if (intersects_aabb) {
        if (node->left < 0 && node->right < 0) {
            int detected = 0;
            for (int i = 0; i < node->count; ++i) {
                int triIndex = s->leaf_idx[node->first + i];
                const uint32_t* tri = s->tris + 3 * (size_t)triIndex;
                float tmp_t = FLT_MAX, tmp_u = 0.0f, tmp_v = 0.0f;
                cnt->tris_tested++;
                int hit = tri_intersect(ray->O, ray->D, vert_at(s, tri[0]), vert_at(s, tri[1]), vert_at(s, tri[2]),
                                        &tmp_t, &tmp_u, &tmp_v);
                if (hit) {
                    detected = 1;
                    if (tmp_t < *t) { *t = tmp_t; *tri_index = (uint32_t)triIndex; *u = tmp_u; *v = tmp_v; }
                }
            }
            return detected;
        } else {
            int hl = 0, hr = 0;
            if (node->left >= 0) hl = kd_intersect_node(s, node->left, ray, t, tri_index, u, v, cnt);
            if (node->right >= 0) hr = kd_intersect_node(s, node->right, ray, t, tri_index, u, v, cnt);
            return hl || hr;
        }
    }
    return 0;
}

static int kd_intersect(const orc_scene* s, const ray_t* ray, float* t, uint32_t* tri_index, float* u, float* v,
                        orc_counters* cnt)
{
    *t = FLT_MAX;
    if (!s->n_nodes) return 0;
    return kd_intersect_node(s, 0, ray, t, tri_index, u, v, cnt);
}

/* ------------------------------------------------------------------------------------------------
 * Renderer::TraceRay / ClosestHit* / RefractionRay / PerPixel (Renderer.cpp:138-372)
 * ---------------------------------------------------------------------------------------------- */
#define EPSILON 0.0002f              /* Renderer.cpp:14 */
#define ID_TRI_BIT 0x40000000

typedef struct { float Distance; v3 Position, Normal; uint32_t MaterialIndex; int32_t id; } hit_t;

static hit_t trace_ray(const orc_scene* s, int scene_mode, const ray_t* ray, orc_counters* cnt)
{
    float closestDistSpheres = FLT_MAX;  int closestSphereIndex = -1;
    float closestDistTriangles = FLT_MAX; int closestTriangleIndex = -1;
    float cu = 0.0f, cv = 0.0f;
    cnt->rays++;

    if (scene_mode == 1 || scene_mode == 2) {               /* Renderer.cpp:243-271 */
        float a = v_dot(ray->D, ray->D);
        float dbl_a = 2.0f * a;
        for (uint32_t i = 0; i < s->n_spheres; ++i) {
            const orc_sphere* sp = &s->spheres[i];
            cnt->sphere_tests++;
            v3 ro = v_sub(ray->O, V(sp->x, sp->y, sp->z));
            float b = 2.0f * v_dot(ro, ray->D);
            float c = v_dot(ro, ro) - sp->radius * sp->radius;
            float disc = b * b - 4.0f * a * c;
            if (disc >= 0.0f) {
                float t = (-b - sqrtf(disc)) / dbl_a;
                if (t > 0.0f && t < closestDistSpheres) { closestDistSpheres = t; closestSphereIndex = (int)i; }
            }
        }
    }
    if (scene_mode == 0 || scene_mode == 2) {               /* Renderer.cpp:272-286 */
        float t = 0.0f, u = 0.0f, v = 0.0f; uint32_t hitIndex = 0;
        if (kd_intersect(s, ray, &t, &hitIndex, &u, &v, cnt)) {
            closestDistTriangles = t; closestTriangleIndex = (int)hitIndex; cu = u; cv = v;
        }
    }

    hit_t h; memset(&h, 0, sizeof h);
    if (closestSphereIndex < 0 && closestTriangleIndex < 0) { h.Distance = -1.0f; h.id = -1; return h; }  /* Miss() */

    if (closestDistSpheres < closestDistTriangles) {        /* ClosestHitSphere, Renderer.cpp:303-317 */
        const orc_sphere* sp = &s->spheres[closestSphereIndex];
        v3 P = v_add(ray->O, v_scale(ray->D, closestDistSpheres));
        h.Distance = closestDistSpheres; h.Position = P;
        h.Normal = v_normalize(v_sub(P, V(sp->x, sp->y, sp->z)));
        h.MaterialIndex = sp->material; h.id = closestSphereIndex;
        return h;
    }
    /* ClosestHitTriangle, Renderer.cpp:319-332 */
    const float* n = s->normals + 9 * (size_t)closestTriangleIndex;
    v3 P = v_add(ray->O, v_scale(ray->D, closestDistTriangles));
    v3 nrm = v_add(v_add(v_scale(V(n[0], n[1], n[2]), (1.0f - cu) - cv), v_scale(V(n[3], n[4], n[5]), cu)),
                   v_scale(V(n[6], n[7], n[8]), cv));
    h.Distance = closestDistTriangles; h.Position = P; h.Normal = v_normalize(nrm);
    h.MaterialIndex = s->tri_mat[closestTriangleIndex]; h.id = ID_TRI_BIT | closestTriangleIndex;
    return h;
}

static int refraction_ray(v3 D, v3 N, v3 P, float IOR, v3* outO, v3* outD)
{
    v3 ref_n = N;
    float eta_t = IOR, eta_i = 1.0f;
    float i_dot_n = v_dot(D, N);
    if (i_dot_n < 0.0f) i_dot_n = -i_dot_n;
    else { ref_n = v_neg(N); eta_t = 1.0f; eta_i = IOR; }
    float eta = eta_i / eta_t;
    float k = 1.0f - (eta * eta) * (1.0f - i_dot_n * i_dot_n);
    if (k < 0.0f) return 0;
    *outO = v_add(P, v_scale(ref_n, -EPSILON));
    *outD = v_normalize(v_sub(v_scale(v_add(D, v_scale(ref_n, i_dot_n)), eta), v_scale(ref_n, sqrtf(k))));
    return 1;
}

int orc_refract(const float d[3], const float n[3], const float p[3], float ior, float out_o[3], float out_d[3])
{
    v3 o = V(0, 0, 0), dd = V(0, 0, 0);
    int ok = refraction_ray(V(d[0], d[1], d[2]), V(n[0], n[1], n[2]), V(p[0], p[1], p[2]), ior, &o, &dd);
    out_o[0] = o.x; out_o[1] = o.y; out_o[2] = o.z; out_d[0] = dd.x; out_d[1] = dd.y; out_d[2] = dd.z;
    return ok;
}

typedef struct { int mode; uint64_t* state; } rng_t;

static inline float rng_float(rng_t* r) { return orc_u32_to_float(orc_pcg32_next(r->state)); }

static const float kPi = 3.14159274f;     /* std::atan(1.0f) * 4.0f, bits 0x40490fdb (Renderer.h:12) */

/* Util::RandomUnitVector (Renderer.h:46-54) */
static inline v3 random_unit_vector(rng_t* rng)
{
    const float TwoPi = 2.0f * kPi;
    float z = rng_float(rng) * 2.0f - 1.0f;
    float a = rng_float(rng) * TwoPi;
    if (rng->mode == ORC_RNG_PIXEL_HASH) {
        float r = sqrtf(1.0f - z * z), sn, cs;
        orc_sincos(a, &sn, &cs);
        return V(r * cs, r * sn, z);
    } else if (rng->mode == ORC_RNG_SERIAL_LIBM_DOUBLE) {
        /* unqualified sqrt/cos/sin binding to the double overloads: float operands promote, the
           products are formed in double and rounded once on assignment */
        float r = (float)sqrt((double)(1.0f - z * z));
        float x = (float)((double)r * cos((double)a));
        float y = (float)((double)r * sin((double)a));
        return V(x, y, z);
    } else {
        float r = sqrtf(1.0f - z * z);
        return V(r * cosf(a), r * sinf(a), z);
    }
}

static v3 per_pixel(const orc_scene* s, const orc_settings* st, v3 camPos, v3 dir, rng_t* rng, int32_t* primary_id,
                    orc_counters* cnt)
{
    ray_t ray; ray.O = camPos; ray.D = dir;
    v3 ambientColor = V(0.0f, 0.0f, 0.0f), finalColor = V(0.0f, 0.0f, 0.0f), contribution = V(1.0f, 1.0f, 1.0f);
    if (primary_id) *primary_id = -1;

    for (uint32_t i = 0; i < st->bounces; ++i) {
        ray.invD = V(1.0f / ray.D.x, 1.0f / ray.D.y, 1.0f / ray.D.z);
        hit_t hit = trace_ray(s, st->scene_mode, &ray, cnt);
        if (i == 0 && primary_id) *primary_id = hit.id;
        if (hit.Distance < 0.0f) { finalColor = v_add(finalColor, v_mul(contribution, ambientColor)); break; }

        const orc_material* mat = &s->mats[hit.MaterialIndex];
        int doTransmission = 0;
        int hitInside = v_dot(ray.D, hit.Normal) > 0.0f;
        v3 normalSurface = hitInside ? v_neg(hit.Normal) : hit.Normal;
        ray.O = v_add(hit.Position, v_scale(normalSurface, EPSILON));
        if (mat->transparency > 0.0f) doTransmission = 1;

        if (doTransmission) {
            v3 o2, d2;
            if (refraction_ray(ray.D, hit.Normal, hit.Position, mat->ior, &o2, &d2)) { ray.O = o2; ray.D = d2; continue; }
        } else {
            v3 diffuseRayDir = v_normalize(v_add(hit.Normal, random_unit_vector(rng)));
            v3 reflectedVector = v_reflect(ray.D, hit.Normal);
            reflectedVector = v_normalize(v_mix(reflectedVector, diffuseRayDir, mat->roughness * mat->roughness));
            ray.D = reflectedVector;
        }
        finalColor = v_add(finalColor, v_mul(V(mat->emission[0], mat->emission[1], mat->emission[2]), contribution));
        contribution = v_mul(contribution, V(mat->albedo[0], mat->albedo[1], mat->albedo[2]));
        {
            float p = f_max(contribution.x, f_max(contribution.y, contribution.z));
            if (rng_float(rng) > p) break;
            contribution = v_scale(contribution, 1.0f / p);
        }
    }
    return finalColor;
}

/* ------------------------------------------------------------------------------------------------
 * Renderer::Render (Renderer.cpp:45-135)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const orc_scene* s; const orc_settings* st; v3 camPos; const float* dirs; uint32_t w, h, frame, seed;
    int mode; float* accum; float* image; int32_t* ids; orc_counters cnt; int tid, nthreads; uint32_t row_step;
} job_t;

static void resolve_pixel(const job_t* j, const float acc[3], float* px)
{
    float c[3] = { acc[0] / (float)j->frame, acc[1] / (float)j->frame, acc[2] / (float)j->frame };
    float o[3];
    orc_tonemap(c, j->st->use_ace, j->mode == ORC_RNG_PIXEL_HASH, o);
    px[0] = o[0]; px[1] = o[1]; px[2] = o[2]; px[3] = 1.0f;
}

static void* render_rows(void* arg)
{
    job_t* j = (job_t*)arg;
    for (uint32_t y = (uint32_t)j->tid * j->row_step; y < j->h; y += (uint32_t)j->nthreads * j->row_step) {
        for (uint32_t x = 0; x < j->w; ++x) {
            uint32_t pixelIndex = y * j->w + x;
            uint64_t local_state;
            rng_t rng; rng.mode = j->mode;
            if (j->mode == ORC_RNG_PIXEL_HASH) { local_state = orc_pixel_seed(pixelIndex, j->frame, j->seed); rng.state = &local_state; }
            else rng.state = &g_serial_state;
            const float* d = j->dirs + 3 * (size_t)pixelIndex;
            v3 color = per_pixel(j->s, j->st, j->camPos, V(d[0], d[1], d[2]), &rng, j->ids ? &j->ids[pixelIndex] : NULL, &j->cnt);
            float* a = j->accum + 3 * (size_t)pixelIndex;
            a[0] += color.x; a[1] += color.y; a[2] += color.z;
            resolve_pixel(j, a, j->image + 4 * (size_t)pixelIndex);
        }
    }
    return NULL;
}

static void aa_pass(const job_t* j)      /* Renderer.cpp:84-124 */
{
    const int kernelSize = 1;
    for (uint32_t y = 0; y < j->h; ++y)
        for (uint32_t x = 0; x < j->w; ++x) {
            float acc[3] = { 0.0f, 0.0f, 0.0f }, weight = 0.0f;
            for (int yo = -kernelSize; yo <= kernelSize; ++yo)
                for (int xo = -kernelSize; xo <= kernelSize; ++xo) {
                    int64_t xx = (int64_t)x + xo, yy = (int64_t)y + yo;
                    if (abs(xo + yo) > kernelSize || xx < 0 || xx >= (int64_t)j->w || yy < 0 || yy >= (int64_t)j->h) continue;
                    float wgt = 1.0f - ((float)(abs(yo) + abs(xo)) * 0.5f);
                    const float* a = j->accum + 3 * ((size_t)yy * j->w + (size_t)xx);
                    acc[0] += a[0] * wgt; acc[1] += a[1] * wgt; acc[2] += a[2] * wgt;
                    weight += wgt;
                }
            if (weight == 0.0f) continue;
            float div = weight * (float)j->frame;
            float c[3] = { acc[0] / div, acc[1] / div, acc[2] / div }, o[3];
            orc_tonemap(c, j->st->use_ace, j->mode == ORC_RNG_PIXEL_HASH, o);
            float* px = j->image + 4 * ((size_t)y * j->w + x);
            px[0] = o[0]; px[1] = o[1]; px[2] = o[2]; px[3] = 1.0f;
        }
}

int orc_render_frame(const orc_scene* s, const orc_settings* st, const float cam_pos[3], const float* dirs3,
                     uint32_t w, uint32_t h, uint32_t frame_index, int rng_mode, uint32_t seed,
                     float* accum3, float* image4, int32_t* primary_ids, orc_counters* counters,
                     int nthreads, uint32_t row_step)
{
    if (!frame_index || !w || !h) return -1;
    if ((st->scene_mode == 0 || st->scene_mode == 2) && !s->n_nodes) return -2;
    if (row_step == 0) row_step = 1;
    if (rng_mode != ORC_RNG_PIXEL_HASH) { nthreads = 1; if (!g_serial_init) orc_serial_rng_reset(); }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if (frame_index == 1) memset(accum3, 0, (size_t)w * h * 3 * sizeof(float));

    job_t jobs[256]; pthread_t th[256];
    for (int t = 0; t < nthreads; ++t) {
        job_t* j = &jobs[t];
        memset(j, 0, sizeof *j);
        j->s = s; j->st = st; j->camPos = V(cam_pos[0], cam_pos[1], cam_pos[2]); j->dirs = dirs3; j->w = w; j->h = h;
        j->frame = frame_index; j->seed = seed; j->mode = rng_mode; j->accum = accum3; j->image = image4;
        j->ids = primary_ids; j->tid = t; j->nthreads = nthreads; j->row_step = row_step;
    }
    if (nthreads == 1) render_rows(&jobs[0]);
    else {
        for (int t = 0; t < nthreads; ++t) pthread_create(&th[t], NULL, render_rows, &jobs[t]);
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    if (st->anti_aliasing) aa_pass(&jobs[0]);
    if (counters) {
        for (int t = 0; t < nthreads; ++t) {
            counters->rays += jobs[t].cnt.rays; counters->nodes_visited += jobs[t].cnt.nodes_visited;
            counters->tris_tested += jobs[t].cnt.tris_tested; counters->sphere_tests += jobs[t].cnt.sphere_tests;
        }
    }
    return 0;
}

int orc_trace_primary(const orc_scene* s, int scene_mode, const float cam_pos[3], const float* dirs3, uint32_t n,
                      int32_t* ids, float* t, float* u, float* v, orc_counters* counters)
{
    orc_counters c; memset(&c, 0, sizeof c);
    if ((scene_mode == 0 || scene_mode == 2) && !s->n_nodes) return -2;
    for (uint32_t i = 0; i < n; ++i) {
        ray_t ray; ray.O = V(cam_pos[0], cam_pos[1], cam_pos[2]);
        ray.D = V(dirs3[3 * (size_t)i], dirs3[3 * (size_t)i + 1], dirs3[3 * (size_t)i + 2]);
        ray.invD = V(1.0f / ray.D.x, 1.0f / ray.D.y, 1.0f / ray.D.z);
        float tt = FLT_MAX, uu = 0.0f, vv = 0.0f; uint32_t idx = 0;
        if (scene_mode == 0) {
            int hit = kd_intersect(s, &ray, &tt, &idx, &uu, &vv, &c);
            c.rays++;
            ids[i] = hit ? (int32_t)(ID_TRI_BIT | idx) : -1;
            if (t) t[i] = tt; if (u) u[i] = uu; if (v) v[i] = vv;
        } else {
            hit_t h = trace_ray(s, scene_mode, &ray, &c);
            ids[i] = h.id;
            if (t) t[i] = h.Distance; if (u) u[i] = 0.0f; if (v) v[i] = 0.0f;
        }
    }
    if (counters) *counters = c;
    return 0;
}
