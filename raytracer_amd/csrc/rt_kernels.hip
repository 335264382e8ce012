// rt_kernels.hip -- gfx950 kernels of the per-pixel tracing path.
//
// One lane owns one pixel for a whole path: camera ray generation (Camera.cpp:134-150), the bounce loop
// (Renderer.cpp:138-229), closest hit over spheres (Renderer.cpp:243-271) and/or the flattened KD tree
// (KDTreeCPU.cpp:640-708, Intersections.cpp:23-94), shading (Renderer.cpp:303-361), accumulation and the tone-mapped
// RGBA32F frame (Renderer.cpp:64-79).  fp32 VALU work, no MFMA (nothing here is a dense contraction).
//
// Numerics contract: every float operation is a single IEEE binary32 op in the order glm 0.9.9.9 evaluates it;
// the file is compiled with -ffp-contract=off (no FMA fusion) and hipcc's default correctly rounded fp32 divide
// and sqrt, so results are bit-identical to a CPU that follows the same order.  min/max are compare+select in the
// reference's operand order (NOT v_min/v_max, which drop NaNs): with a zero direction component the slab test
// produces NaNs that the reference treats as "box hit, never pruned", and the centre pixel column of the default
// camera has direction.x == 0 exactly.
#include <hip/hip_runtime.h>
#include <float.h>
#include "rt_device.h"

namespace {

struct F3 { float x, y, z; };

__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
// glm::dot(vec3) = (x + y) + z of the component products (detail/func_geometric.inl:48-55)
__device__ __forceinline__ float dot3(F3 a, F3 b) { F3 t = a * b; return (t.x + t.y) + t.z; }
// glm::cross (detail/func_geometric.inl:68-79)
__device__ __forceinline__ F3 cross3(F3 x, F3 y)
{
    return f3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// Correctly rounded square root, bit for bit sqrtf.  For x in [2^-96, FLT_MAX] it is the core of the compiler's own expansion
// of sqrtf -- the hardware estimate (within one ulp), then the neighbours one ulp below and above tried with exact FMA
// residuals -- without the rescaling that only smaller inputs need and without the zero / infinity / NaN pass-through; any
// other input takes sqrtf itself.  The path calls this 14 times per bounce; the plain expansion is twice as long.
__device__ __forceinline__ float sqrt_rn(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_below = __builtin_fmaf(-below, s, x), r_above = __builtin_fmaf(-above, s, x);
    float r = (0.0f >= r_below) ? below : s;
    r = (0.0f < r_above) ? above : r;
    // outside the range (tiny, zero, negative, infinite, NaN): rare, so one wave-wide test guards the general routine
    const bool general = !(x >= 0x1p-96f && x <= FLT_MAX);
    if (__ballot(general) != 0ull) {
        float y = x;
        asm volatile("" : "+v"(y));                         // keeps the general routine inside the rare branch (it would otherwise be
        if (general) r = sqrtf(y);                          // computed speculatively for every call and merged with a select)
    }
    return r;
}

// 1.0f / sqrt_rn(x): both operations correctly rounded, bit for bit what the two separate routines give.  For x in
// [2^-96, FLT_MAX] the root lies in [2^-48, 2^64], where the compiler's own expansion of an fp32 division neither rescales its
// operands nor needs its final special-case pass: what is left of it for a numerator of 1 is the hardware reciprocal and six
// FMAs (its first quotient 1 * r is r itself).  Any other x takes the general routines behind the same wave-wide test.
#ifndef RT_FAST_RSQRT
#define RT_FAST_RSQRT 1
#endif
// 1.0f / d, correctly rounded, for 2^-64 <= |d| <= 2^64: the compiler's expansion of that division with the operand scaling and
// the special-case pass left out, neither of which does anything in this range (its first quotient 1 * r is r itself).
__device__ __forceinline__ float rcp_unscaled(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return __builtin_fmaf(__builtin_fmaf(-d, q, 1.0f), r, q);
}

// a / d, correctly rounded, for operands of moderate size: the compiler's expansion of the fp32 division (reciprocal estimate, one
// Newton step on it, quotient, two residual corrections) with the operand scaling and the special-case pass left out.  Neither
// does anything while a is zero or 2^-100 <= |a| <= 2^100, 2^-100 <= |d| <= 2^100 and the quotient stays within those bounds as
// well: every intermediate value is then a normal number, and scaling by powers of two would change no rounding.  The callers
// prove the ranges.  (8 instructions instead of 11 + the waits for the scaling's condition codes, and the reciprocal's part is
// shared between quotients with the same divisor.)
__device__ __forceinline__ float div_unscaled(float a, float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    float q = a * r;
    q = __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
    return __builtin_fmaf(__builtin_fmaf(-d, q, a), r, q);
}

__device__ __forceinline__ float rsqrt_rn(float x)
{
#if RT_FAST_RSQRT
    const float s = __builtin_amdgcn_sqrtf(x);
    const float below = __uint_as_float(__float_as_uint(s) - 1u), above = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_below = __builtin_fmaf(-below, s, x), r_above = __builtin_fmaf(-above, s, x);
    float d = (0.0f >= r_below) ? below : s;
    d = (0.0f < r_above) ? above : d;
    float q = rcp_unscaled(d);
    const bool general = !(x >= 0x1p-96f && x <= FLT_MAX);
    if (__ballot(general) != 0ull) {
        float y = x;
        asm volatile("" : "+v"(y));
        if (general) q = 1.0f / sqrtf(y);
    }
    return q;
#else
    return 1.0f / sqrt_rn(x);
#endif
}

// glm::normalize = v * (1 / sqrt(dot(v, v))) (func_geometric.inl:82-90, func_exponential.inl:134-139)
__device__ __forceinline__ F3 normalize3(F3 v) { return v * rsqrt_rn(dot3(v, v)); }
// glm::reflect = I - N * dot(N, I) * 2 (func_geometric.inl:104-110)
__device__ __forceinline__ F3 reflect3(F3 I, F3 N) { return I - (N * dot3(N, I)) * 2.0f; }
// glm::mix = x * (1 - a) + y * a (func_common.inl:104-112)
__device__ __forceinline__ F3 mix3(F3 x, F3 y, float a) { return x * (1.0f - a) + y * a; }
// glm::min(x, y) = (y < x) ? y : x, glm::max(x, y) = (x < y) ? y : x (func_common.inl:17-30); std::min/max alike.
__device__ __forceinline__ float sel_min(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float sel_max(float x, float y) { return (x < y) ? y : x; }

// ---- pcg32 (setseq_xsh_rr_64_32, default stream; pcg_random.hpp:158-159,413-437,484-489,845-873) ------------------
constexpr unsigned long long PCG_MULT = 6364136223846793005ULL;
constexpr unsigned long long PCG_INC = 1442695040888963407ULL;

__device__ __forceinline__ unsigned long long pixel_seed(uint32_t pixel, uint32_t frame, uint32_t seed)
{
    // splitmix64 finaliser over (frame:pixel) offset by the global seed, then the pcg32 constructor
    unsigned long long z = (((unsigned long long)frame << 32) | (unsigned long long)pixel)
                           + (unsigned long long)seed * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (z + PCG_INC) * PCG_MULT + PCG_INC;
}

// Walnut::Random::Float (Random.h:11,35-38): one engine step, (float)u32 / 2^32
__device__ __forceinline__ float rng_float(unsigned long long& state)
{
    unsigned long long old = state;
    state = old * PCG_MULT + PCG_INC;
    uint32_t rot = (uint32_t)(old >> 59);
    old ^= old >> 18;
    uint32_t x = (uint32_t)(old >> 27);
    uint32_t u = (x >> rot) | (x << ((32u - rot) & 31u));
    return (float)u / 4294967296.0f;
}

// ---- portable sincos / x^(1/2.4): + - * / and int<->float conversions only, fixed order -----------------------------
__device__ __forceinline__ void sincos_portable(float a, float& s_out, float& c_out)
{
    int k = (int)(a * 0.636619772f + 0.5f);
    float kf = (float)k;
    float r = a - kf * 1.5703125f;
    r = r - kf * 4.837512969970703125e-4f;
    r = r - kf * 7.54978995489188216e-8f;
    float z = r * r;
    float ps = -1.9515295891e-4f * z;
    ps = ps + 8.3321608736e-3f;
    ps = ps * z;
    ps = ps - 1.6666654611e-1f;
    ps = ps * z;
    ps = ps * r;
    float sn = ps + r;
    float pc = 2.443315711809948e-5f * z;
    pc = pc - 1.388731625493765e-3f;
    pc = pc * z;
    pc = pc + 4.166664568298827e-2f;
    pc = pc * z;
    pc = pc * z;
    float hz = 0.5f * z;
    float cs = pc - hz;
    cs = cs + 1.0f;
    int q = k & 3;
    float s = (q & 1) ? cs : sn;
    float c = (q & 1) ? sn : cs;
    s_out = (q & 2) ? -s : s;
    c_out = (q == 1 || q == 2) ? -c : c;
}

__device__ __forceinline__ float pow_inv24(float x)
{
    uint32_t ux = __float_as_uint(x);
    int e = (int)(ux >> 23) - 127;
    float m = __uint_as_float((ux & 0x007FFFFFu) | 0x3F800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e = e + 1; }
    float s = div_unscaled(m - 1.0f, m + 1.0f);   // m lies in (0.707, 1.4143] whatever x is: |m - 1| is 0 or at least 2^-24, m + 1 in (1.7, 2.42)
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
    float l2 = p * 2.88539008f;
    float y = ((float)e + l2) * 0.416666667f;
    int n = (int)(y + 1024.5f) - 1024;
    float g = y - (float)n;
    float t = g * 0.693147181f;
    float q = 1.98412698e-4f * t;
    q = q + 1.38888889e-3f;
    q = q * t;
    q = q + 8.33333333e-3f;
    q = q * t;
    q = q + 4.16666667e-2f;
    q = q * t;
    q = q + 1.66666667e-1f;
    q = q * t;
    q = q + 0.5f;
    q = q * t;
    q = q + 1.0f;
    q = q * t;
    q = q + 1.0f;
    if (n < -126) n = -126;
    return q * __uint_as_float((uint32_t)(n + 127) << 23);
}

// ---- tone map: Util::ACESFilm + LinearToSRGB (Renderer.h:57-98) ---------------------------------------------------
__device__ __forceinline__ float clamp01(float x) { return sel_min(sel_max(x, 0.0f), 1.0f); }

// PLAIN: x is zero or 2^-72 <= x <= 2^40 (see resolve_px): numerator and denominator of the ACES curve then lie within
// [2^-78, 2^82] and [0.14, 2^82], their quotient within [2^-75, 1.04].
template <bool PLAIN>
__device__ __forceinline__ float tonemap1(float x)
{
    float num = x * (2.51f * x + 0.03f);
    float den = x * (2.43f * x + 0.59f) + 0.14f;
    float c = clamp01(PLAIN ? div_unscaled(num, den) : num / den);          // ACESFilm
    c = clamp01(c);                        // LinearToSRGB's own clamp
    float hi = pow_inv24(c) * 1.055f - 0.055f;
    float lo = c * 12.92f;
    float a = (c < 0.0031308f) ? 1.0f : 0.0f;
    return hi * (1.0f - a) + lo * a;
}

// Sums that are zero or between 2^-40 and 2^40 (every sum of a sane scene), divided by a frame count between 1 and 2^32, take the
// short form of the division here and in the tone map; a wave that holds any other sum -- negative, huge, tiny, NaN -- takes the
// general forms for all its pixels.  Same bits either way.
__device__ __forceinline__ float4 resolve_px(F3 acc, float div, int use_ace)
{
    auto tame = [](float a) { return a == 0.0f || (a >= 0x1p-40f && a <= 0x1p40f); };
    const bool plain = tame(acc.x) && tame(acc.y) && tame(acc.z) && div >= 1.0f && div <= 0x1p32f;
    float r, g, b;
    if (__ballot(!plain) == 0ull) {
        r = div_unscaled(acc.x, div); g = div_unscaled(acc.y, div); b = div_unscaled(acc.z, div);
        if (use_ace) { r = tonemap1<true>(r); g = tonemap1<true>(g); b = tonemap1<true>(b); }
    } else {
        r = acc.x / div; g = acc.y / div; b = acc.z / div;
        if (use_ace) { r = tonemap1<false>(r); g = tonemap1<false>(g); b = tonemap1<false>(b); }
    }
    return make_float4(r, g, b, 1.0f);
}

// ---- camera ray (Camera.cpp:142-147); mat4*vec4 = (m0*v0 + m1*v1) + (m2*v2 + m3*v3) (type_mat4x4.inl:561-572) ----
__device__ __forceinline__ float mat_row(const float* m, int r, float v0, float v1, float v2, float v3)
{
    float Add0 = m[0 + r] * v0 + m[4 + r] * v1;
    float Add1 = m[8 + r] * v2 + m[12 + r] * v3;
    return Add0 + Add1;
}

__device__ __forceinline__ F3 camera_dir(const RtParams& p, uint32_t x, uint32_t y)
{
    float cx = (float)x / (float)p.width;
    float cy = (float)y / (float)p.height;
    cx = cx * 2.0f - 1.0f;
    cy = cy * 2.0f - 1.0f;
    float tx = mat_row(p.inv_proj, 0, cx, cy, 1.0f, 1.0f);
    float ty = mat_row(p.inv_proj, 1, cx, cy, 1.0f, 1.0f);
    float tz = mat_row(p.inv_proj, 2, cx, cy, 1.0f, 1.0f);
    float tw = mat_row(p.inv_proj, 3, cx, cy, 1.0f, 1.0f);
    F3 n = normalize3(f3(tx / tw, ty / tw, tz / tw));
    F3 rd = f3(mat_row(p.inv_view, 0, n.x, n.y, n.z, 0.0f), mat_row(p.inv_view, 1, n.x, n.y, n.z, 0.0f),
               mat_row(p.inv_view, 2, n.x, n.y, n.z, 0.0f));
    return normalize3(rd);
}

struct Hit { float t; int32_t id; float u, v; };
// KDTreeCPU::intersect reports a hit as soon as ANY triangle test of a visited leaf succeeds ("intersection_detected",
// KDTreeCPU.cpp:666-690), also when the test's distance never passed "tmp_t < t" -- which, with t starting at FLT_MAX, takes a
// successful test at t = +inf (its dot products overflowed) or at FLT_MAX exactly.  TraceRay then shades triangle 0 at distance
// FLT_MAX with u = v = 0, the initial values of its variables (Renderer.cpp:274-286).  Hit::id = RT_HIT_DETECTED marks that state of
// a walk (a test succeeded, no distance accepted yet); resolve_detected turns it into what TraceRay sees.
constexpr int32_t RT_HIT_DETECTED = -2;
__device__ __forceinline__ void resolve_detected(Hit& h)
{
    if (h.id == RT_HIT_DETECTED) { h.id = 0; h.t = FLT_MAX; h.u = 0.0f; h.v = 0.0f; }
}
constexpr uint32_t RT_NO_REC = 0xFFFFFFFFu;           // "no leaf record": a walk's cursor is at a node

// ---- spheres: Renderer.cpp:243-271 ------------------------------------------------------------------------------
// The reference computes t = (-b - sqrt(disc)) / (2a) for every sphere with disc >= 0 and keeps the first strictly smaller
// positive t.  Exact shortcuts skip work whose outcome is already decided:
//   * b >= 0  =>  the numerator -b - sqrt(disc) <= 0  =>  t <= 0: rejected by "t > 0";
//   * all candidates share the positive divisor 2a and IEEE division is monotone in the numerator, so a numerator that is
//     not below the best numerator so far cannot give a strictly smaller t: rejected by "t < closest";
//   * a numerator below the best one by more than a millionth of it gives a strictly smaller quotient (each quotient is
//     rounded by at most half a unit in the last place, 6e-8 of its value), so the loop only tracks the best numerator and its
//     index and divides once at the end.  Numerators closer together than that are decided by their quotients, as the
//     reference does (the second of two equal quotients loses).
// All of them leave closest / index exactly as the reference's loop would (closest is FLT_MAX on entry).
struct SphereBest { float num; int index; };

__device__ __forceinline__ void sphere_test(float4 sp, int i, F3 O, F3 D, float dbl_a, float four_a, SphereBest& best)
{
    F3 ro = O - f3(sp.x, sp.y, sp.z);
    float b = 2.0f * dot3(ro, D);
    float c = dot3(ro, ro) - sp.w;                         // sp.w = radius*radius, rounded once at upload
    float disc = b * b - four_a * c;
    // (Branches, not selects: the rays of a wave are coherent enough that whole waves skip the square root for most spheres;
    // straight-line code measured a third slower.)
    if (disc >= 0.0f && b < 0.0f) {
        float num = -b - sqrt_rn(disc);
        if (num > 0.0f && num < best.num) {
            bool take = num < best.num * 0.999999f && num > 1e-30f;
            if (!take) {                                    // about one candidate in a million
                const float t = num / dbl_a;
                take = t > 0.0f && (best.index < 0 || t < best.num / dbl_a);
            }
            if (take) { best.num = num; best.index = i; }
        }
    }
}

template <bool INLINE>
__device__ __forceinline__ void trace_spheres_from(const RtParams& p, F3 O, F3 D, float& closest, int& index)
{
    float a = dot3(D, D);
    float dbl_a = 2.0f * a;
    float four_a = 4.0f * a;
    SphereBest best; best.num = __builtin_inff(); best.index = -1;
    const uint32_t n = p.n_spheres;
    if (n == 0u) return;
    // two spheres per trip, each read (wave-uniform index: a scalar load) while the other is tested
    float4 even = INLINE ? p.sph_inline[0] : p.spheres[0], odd = even;
    for (uint32_t i = 0; i < n; i += 2u) {
        if (i + 1u < n) odd = INLINE ? p.sph_inline[i + 1u] : p.spheres[i + 1u];
        sphere_test(even, (int)i, O, D, dbl_a, four_a, best);
        if (i + 1u >= n) break;
        if (i + 2u < n) even = INLINE ? p.sph_inline[i + 2u] : p.spheres[i + 2u];
        sphere_test(odd, (int)i + 1, O, D, dbl_a, four_a, best);
    }
    if (best.index >= 0) {
        const float t = best.num / dbl_a;
        if (t > 0.0f && t < closest) { closest = t; index = best.index; }
    }
}

#ifndef RT_SPHERE_LDS_LOOP
#define RT_SPHERE_LDS_LOOP 1
#endif
// The same loop over the workgroup's copy of the sphere table in LDS: every lane reads the same address (one broadcast read
// straight into vector registers, no scalar load, no address arithmetic on the scalar unit, no copies from scalar registers).
__device__ __forceinline__ void trace_spheres_lds(const float4* spheres, uint32_t n, F3 O, F3 D, float& closest, int& index)
{
    float a = dot3(D, D);
    float dbl_a = 2.0f * a;
    float four_a = 4.0f * a;
    SphereBest best; best.num = __builtin_inff(); best.index = -1;
    for (uint32_t i = 0; i < n; ++i) sphere_test(spheres[i], (int)i, O, D, dbl_a, four_a, best);
    if (best.index >= 0) {
        const float t = best.num / dbl_a;
        if (t > 0.0f && t < closest) { closest = t; index = best.index; }
    }
}

__device__ __forceinline__ void trace_spheres(const RtParams& p, F3 O, F3 D, float& closest, int& index)
{
    if (p.n_spheres <= RT_INLINE_SPHERES) trace_spheres_from<true>(p, O, D, closest, index);   // kernel-argument copy
    else trace_spheres_from<false>(p, O, D, closest, index);
}

// ---- KD tree: KDTreeCPU.cpp:640-708 over the preorder/skip layout; Intersections.cpp:23-94 ------------------------
// Intersections::aabbIntersect (Intersections.cpp:23-47) with compare+select min/max in the reference's operand order.
// PLAIN: for a ray and a tree that cannot produce a NaN or an infinity in any of these operations (see slab_plain_ray): all twelve
// values are finite, so the hardware minimum / maximum return the operand the compare+select forms return -- up to the sign of a
// zero, which none of the comparisons below (nor the pruning test on tNear) can tell -- in 8 instructions instead of 28.
template <bool PLAIN>
__device__ __forceinline__ bool slab_test_t(float4 n0, float4 n1, F3 O, F3 invD, float& tNear)
{
    F3 l1 = (f3(n0.x, n0.y, n0.z) - O) * invD;
    F3 l2 = f3(n1.x, n1.y, n1.z) * invD;
    F3 tMin = l1 - l2;
    F3 tMax = l1 + l2;
    float tFar;
    if (PLAIN) {
        // (the instructions themselves: the compiler's fminf / fmaxf first pass every operand through a quieting v_max x, x)
        float hx, hy, hz, lx, ly, lz;
        asm("v_max_f32 %0, %1, %2" : "=v"(hx) : "v"(tMin.x), "v"(tMax.x));
        asm("v_max_f32 %0, %1, %2" : "=v"(hy) : "v"(tMin.y), "v"(tMax.y));
        asm("v_max_f32 %0, %1, %2" : "=v"(hz) : "v"(tMin.z), "v"(tMax.z));
        asm("v_min_f32 %0, %1, %2" : "=v"(lx) : "v"(tMin.x), "v"(tMax.x));
        asm("v_min_f32 %0, %1, %2" : "=v"(ly) : "v"(tMin.y), "v"(tMax.y));
        asm("v_min_f32 %0, %1, %2" : "=v"(lz) : "v"(tMin.z), "v"(tMax.z));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tFar) : "v"(hx), "v"(hy), "v"(hz));
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tNear) : "v"(lx), "v"(ly), "v"(lz));
    } else {
        float t2x = sel_max(tMin.x, tMax.x), t2y = sel_max(tMin.y, tMax.y), t2z = sel_max(tMin.z, tMax.z);
        tFar = sel_min(sel_min(t2x, t2y), t2z);
        float t1x = sel_min(tMin.x, tMax.x), t1y = sel_min(tMin.y, tMax.y), t1z = sel_min(tMin.z, tMax.z);
        tNear = sel_max(sel_max(t1x, t1y), t1z);
    }
    return !(tFar < 0.0f) && !(tNear > tFar);
}
__device__ __forceinline__ bool slab_test(float4 n0, float4 n1, F3 O, F3 invD, float& tNear) { return slab_test_t<false>(n0, n1, O, invD, tNear); }

// Can the slab test of this ray against any box of the tree meet a NaN or an infinity?  Not when every box has |centre| + extent
// <= 2^60 per axis (RtParams::slab_plain, checked at upload), the origin lies within 2^60 and every component of the inverse
// direction within 2^64: then |centre - origin| <= 2^61, both products stay below 2^125 and their sum and difference below 2^126.
// (An inverse direction is infinite for a zero AND for a denormal direction component; NaNs fail the comparisons.)
// Nor can the triangle test's determinant a = e1 . (D x e2) leave the range in which the short form of 1 / a is exact: the
// direction's components lie within 2 (inverse >= 1/2) and |e1|_1 |e2|_1 <= 2^56 for every triangle of the mesh (the same flag), so
// |a| <= 2 |D|_inf |e1|_1 |e2|_1 <= 2^58.
__device__ __forceinline__ bool slab_plain_ray(F3 O, F3 invD)
{
    return fabsf(invD.x) <= 0x1p64f && fabsf(invD.y) <= 0x1p64f && fabsf(invD.z) <= 0x1p64f
        && fabsf(invD.x) >= 0.5f && fabsf(invD.y) >= 0.5f && fabsf(invD.z) >= 0.5f
        && fabsf(O.x) <= 0x1p60f && fabsf(O.y) <= 0x1p60f && fabsf(O.z) <= 0x1p60f;
}

// Intersections::triIntersect (Intersections.cpp:56-94) on a pre-gathered record (e1, e2 formed at upload).
// PLAIN: for a ray and a mesh whose determinants stay below 2^58 (see slab_plain_ray).  The quotient 1 / a is used only where
// |a| >= 1e-5, and for 1e-5 <= |a| <= 2^58 the short form of the division gives its bits.
template <bool PLAIN>
__device__ __forceinline__ bool tri_test_t(float4 r0, float4 r1, float4 r2, F3 O, F3 D, float& tt, float& u, float& v)
{
    F3 v0 = f3(r0.x, r0.y, r0.z), e1 = f3(r0.w, r1.x, r1.y), e2 = f3(r1.z, r1.w, r2.x);
    F3 h = cross3(D, e2);
    float a = dot3(e1, h);
    bool ok = !(a > -0.00001f && a < 0.00001f);
    float f = PLAIN ? rcp_unscaled(a) : 1.0f / a;
    F3 s = O - v0;
    u = f * dot3(s, h);
    ok = ok && !(u < 0.0f || u > 1.0f);
    F3 q = cross3(s, e1);
    v = f * dot3(D, q);
    ok = ok && !(v < 0.0f || u + v > 1.0f);
    tt = f * dot3(e2, q);
    return ok && (tt > 0.00001f);
}
__device__ __forceinline__ bool tri_test(float4 r0, float4 r1, float4 r2, F3 O, F3 D, float& tt, float& u, float& v)
{
    return tri_test_t<false>(r0, r1, r2, O, D, tt, u, v);
}

// Index tests of the RT_CHECKED build (see rt_device.h): `index` if it lies below `count`, else 0 with `code` recorded.  Compiled
// out otherwise.  Codes: 1 node cursor, 2 record cursor, 3 triangle of a hit, 4 sphere of a hit, 5 material, 6 column slot,
// 7 / 9 record of a column / parked result, 8 slot of a resumed path, 10 entry node of a unit, 11 / 12 frame / pixel of a
// radiance slot, 13 start of a cooperative walk, 14-16 first record of a leaf in the three whole-wave walks.
__device__ __forceinline__ uint32_t checked(const RtParams& p, uint32_t index, uint32_t count, uint32_t code)
{
#if RT_CHECKED
    if (!(index < count)) { if (p.column_stats) atomicMax(p.column_stats + 4, (unsigned long long)code); return 0u; }
#else
    (void)p; (void)count; (void)code;
#endif
    return index;
}

// A ray with direction.x == 0 exactly has an infinite inverse there, and the slab test then yields NaN for every box on one
// side of the origin (inf - inf): such a box counts as hit with t_near = NaN and "NaN > t" never prunes it.  (A zero in y or z
// does not stick: the reference's select order drops those NaNs and the ray is pruned as usual.)  Every
// decision of the reference traversal is then independent of the running closest t, so the set of visited leaves is fixed
// and the result is the smallest t over them, the earliest-visited triangle winning ties.  That lets the whole wave walk
// the tree for one such ray: each lane takes one node of a 64-node window of the preorder array, a prefix-maximum of the
// skip links of the rejected nodes tells which nodes of the window are still reachable, reached leaves are tested lane-
// parallel, and the per-lane winners are merged by (t, record index) -- record indices grow in visit order.
// Without this a centre-column ray of the default camera (direction.x == 0) walks half the tree alone, ~10^5 dependent steps.
// Returns false (and no result) if some reached node's decision did depend on t: the caller then walks the tree the serial way.
template <bool STATS>
__device__ __forceinline__ bool trace_mesh_cooperative(const RtParams& p, F3 O, F3 D, F3 invD, uint32_t lane, Hit& hit,
                                                       uint32_t& n_nodes_visited, uint32_t& n_tris_tested)
{
    float t = FLT_MAX, hu = 0.0f, hv = 0.0f;
    uint32_t best_rec = 0xFFFFFFFFu;
    int32_t tri = -1;
    uint32_t base = 0, killed_to = 0;                       // wave-uniform: window start; nodes below killed_to are unreachable
    const uint32_t end = p.n_nodes;
    bool t_dependent = false, detected = false;
    while (base < end) {
        const uint32_t j = base + lane;
        const bool valid = j < end;
        float4 n0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), n1 = n0;
        if (valid) { n0 = p.nodes[2 * j]; n1 = p.nodes[2 * j + 1]; }
        float tNear;
        const bool box = slab_test(n0, n1, O, invD, tNear);
        const uint32_t skip = __float_as_uint(n0.w);
        const int32_t leaf_first = (int32_t)__float_as_uint(n1.w);
        uint32_t kill_end = (valid && !box) ? skip : 0u;
        // exclusive prefix maximum over lower lanes
        uint32_t pm = kill_end;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t o = __shfl_up(pm, off);
            if ((int)lane >= off) pm = pm > o ? pm : o;
        }
        uint32_t excl = __shfl_up(pm, 1);
        if (lane == 0) excl = 0u;
        const uint32_t limit = excl > killed_to ? excl : killed_to;
        const bool reached = valid && j >= limit;
        if (reached && box && tNear == tNear) t_dependent = true;        // a finite t_near could be pruned: not our case
        if (STATS && reached) n_nodes_visited++;
        if (reached && box && leaf_first >= 0) {
            uint32_t rec_index = (uint32_t)leaf_first;
            const float4* rec = p.leafrecs + 3 * (size_t)checked(p, (uint32_t)leaf_first, p.n_refs, 14);
            for (;;) {
                const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                rec += 3;
                if (STATS) n_tris_tested++;
                const uint32_t tag = __float_as_uint(r2.y);
                float tt, u, v;
                if (tri_test(r0, r1, r2, O, D, tt, u, v)) {
                    if (tt < t) { t = tt; tri = (int32_t)(tag & 0x7FFFFFFFu); hu = u; hv = v; best_rec = rec_index; }
                    else detected = true;
                }
                ++rec_index;
                if (tag & 0x80000000u) break;
            }
        }
        const uint32_t window_kill = __shfl(pm, 63);        // maximum over the whole window
        if (window_kill > killed_to) killed_to = window_kill;
        base += 64u;
        if (killed_to > base) base = killed_to;             // everything up to killed_to is inside a rejected subtree
    }
    if (__ballot(t_dependent) != 0ull) return false;
    // merge lane winners: smallest t, then smallest record index (= earliest visit)
    for (int off = 32; off > 0; off >>= 1) {
        const float ot = __shfl_xor(t, off); const uint32_t orec = __shfl_xor(best_rec, off);
        const int32_t otri = __shfl_xor(tri, off); const float ou = __shfl_xor(hu, off), ov = __shfl_xor(hv, off);
        if (ot < t || (ot == t && orec < best_rec)) { t = ot; best_rec = orec; tri = otri; hu = ou; hv = ov; }
    }
    if (tri < 0 && __ballot(detected) != 0ull) tri = RT_HIT_DETECTED;
    hit.t = t; hit.id = tri; hit.u = hu; hit.v = hv;
    return true;
}

// Wave-wide inclusive prefix maximum / minimum in six data-parallel-primitive steps (shifts inside each row of 16 lanes, then
// the last lane of a row broadcast to the following rows) -- no round trips through the LDS crossbar.  Lanes without a source
// keep the identity (0 for the unsigned maximum, +inf for the float minimum).
__device__ __forceinline__ uint32_t wave_prefix_max(uint32_t x)
{
    uint32_t o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); x = x > o ? x : o;      // row_shr:1
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false); x = x > o ? x : o;      // row_shr:2
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); x = x > o ? x : o;      // row_shr:4
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false); x = x > o ? x : o;      // row_shr:8
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); x = x > o ? x : o;      // row_bcast:15 into rows 1, 3
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false); x = x > o ? x : o;      // row_bcast:31 into rows 2, 3
    return x;
}
__device__ __forceinline__ float wave_min(float v)            // wave-uniform result; no NaN among the inputs
{
    const int inf = 0x7F800000;
    float x = v, o;
    o = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(x), 0x111, 0xf, 0xf, false)); x = o < x ? o : x;
    o = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(x), 0x112, 0xf, 0xf, false)); x = o < x ? o : x;
    o = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(x), 0x114, 0xf, 0xf, false)); x = o < x ? o : x;
    o = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(x), 0x118, 0xf, 0xf, false)); x = o < x ? o : x;
    o = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(x), 0x142, 0xa, 0xf, false)); x = o < x ? o : x;
    o = __int_as_float(__builtin_amdgcn_update_dpp(inf, __float_as_int(x), 0x143, 0xc, 0xf, false)); x = o < x ? o : x;
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
}

// The same idea for an ordinary ray whose wave has (almost) nothing else to do: at the end of a frame a wave is left with one
// or two long walks, each a chain of hundreds of dependent reads, and 62 idle lanes.  The wave then walks for one ray at a time:
// each lane tests one node of a 64-node window of the preorder array against the closest hit known when the window starts, the
// prefix maximum of the rejected nodes' skip links tells which nodes the walk reaches, reached leaves are tested lane-parallel,
// and the closest t is re-reduced for the next window.  Pruning with the t of the window start instead of the t of the moment
// visits a superset of the leaves the one-lane walk visits; a triangle the one-lane walk would have found in a leaf it visits is
// found here in the same leaf with the same arithmetic, so the closest hit is the same, with ties again resolved to the
// smallest record index.  `start`: the node cursor of the walk; hit / best_rec: its closest hit so far, in and out (all
// arguments wave-uniform).  `cap`: a distance beyond which no hit matters (see tree_body), FLT_MAX = none.
template <bool ORDERED, bool PLAIN>
__device__ __forceinline__ bool walk_cooperative(const RtParams& p, const float4* nodes, F3 O, F3 D, F3 invD, uint32_t start, uint32_t lane,
                                                 float tol, float cap, Hit& hit, uint32_t& best_rec)
{
    bool near_tie = false, detected = hit.id == RT_HIT_DETECTED;
    float t = hit.t, hu = hit.u, hv = hit.v;                // lane-local closest hit; every lane starts from the ray's
    int32_t tri = hit.id >= 0 ? hit.id : -1;
    uint32_t rec_best = best_rec;
    float t_prune = hit.t;                                  // wave-uniform
    const uint32_t end = p.n_nodes;
    start = checked(p, start, end + 1u, 13);
    uint32_t base = start, killed_to = start;
    while (base < end) {
        const uint32_t j = base + lane;
        const bool valid = j < end;
        float4 n0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), n1 = n0;
        if (valid) { n0 = nodes[2 * (size_t)j]; n1 = nodes[2 * (size_t)j + 1]; }
        float tNear;
        const bool box = slab_test_t<PLAIN>(n0, n1, O, invD, tNear);     // (lanes past the end of the array test a box of zeros: as finite)
        const bool descend = valid && box && !(tNear > fminf(ORDERED && RT_NEAR_TIES ? t_prune + (tol + p.tie_rel * t_prune) : t_prune, cap));
        const uint32_t skip = __float_as_uint(n0.w);
        const int32_t leaf_first = (int32_t)__float_as_uint(n1.w);
        // ends of the rejected subtrees up to and including this lane's node; a node that is not rejected contributes nothing,
        // so for it "up to and including" is "before"
        const uint32_t pm = wave_prefix_max((valid && !descend) ? skip : 0u);
        const uint32_t limit = pm > killed_to ? pm : killed_to;
        // The reached leaves of a window are few (a ray crosses one to three of 64 consecutive nodes) and hold about ten records
        // each: tested by their own lanes they would be ten triangle tests one after the other with sixty lanes idle.  Instead
        // up to four leaves are taken per pass, sixteen lanes each, one record per lane (a leaf's records are consecutive, so
        // a group's read is one run of memory); the record with the end mark closes its group, lanes behind it do not count.
        // Which lane tests a record does not matter: every lane keeps its own closest hit, merged by (t, record) at the end.
        unsigned long long leaves = __ballot(j >= limit && descend && leaf_first >= 0);
        while (leaves != 0ull) {
            uint32_t group_first = RT_NO_REC;
#pragma unroll
            for (uint32_t g = 0; g < 4u; ++g) {
                if (leaves != 0ull) {
                    const int src = __ffsll((long long)leaves) - 1;
                    leaves &= leaves - 1ull;
                    const uint32_t first = checked(p, (uint32_t)__builtin_amdgcn_readlane(leaf_first, src), p.n_refs, 15);
                    if ((lane >> 4) == g) group_first = first;
                }
            }
            bool open = group_first != RT_NO_REC;
            uint32_t rec_index = group_first + (lane & 15u);
            while (__ballot(open) != 0ull) {
                const uint32_t at = rec_index < p.n_refs ? rec_index : p.n_refs - 1u;
                float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0;
                if (open) { const float4* rec = p.leafrecs + 3 * (size_t)at; r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; }
                const uint32_t tag = __float_as_uint(r2.y);
                const unsigned long long ends = __ballot(open && (tag & 0x80000000u) != 0u);
                const uint32_t group_ends = (uint32_t)(ends >> (lane & 48u)) & 0xFFFFu;
                const bool counts = open && rec_index < p.n_refs && (group_ends & ((1u << (lane & 15u)) - 1u)) == 0u;
                float tt, u, v;
                if (tri_test_t<PLAIN>(r0, r1, r2, O, D, tt, u, v) && counts) {
                    const int32_t id = (int32_t)(tag & 0x7FFFFFFFu);
                    if (ORDERED && RT_NEAR_TIES && id != tri && fabsf(tt - t) <= tol + p.tie_rel * fminf(tt, t)) near_tie = true;
                    if (tt < t || (tt == t && tri >= 0 && rec_index < rec_best)) { t = tt; tri = id; hu = u; hv = v; rec_best = rec_index; }
                    else if (tri < 0) detected = true;
                }
                if (group_ends != 0u) open = false;
                rec_index += 16u;
            }
        }
        const uint32_t window_kill = (uint32_t)__builtin_amdgcn_readlane((int)pm, 63);
        if (window_kill > killed_to) killed_to = window_kill;
        base += 64u;
        if (killed_to > base) base = killed_to;
        t_prune = wave_min(t);
    }
    const float t_own = t;
    const int32_t tri_own = tri;
    for (int off = 32; off > 0; off >>= 1) {                // lane winners: smallest t, then smallest record index
        const float ot = __shfl_xor(t, off); const uint32_t orec = __shfl_xor(rec_best, off);
        const int32_t otri = __shfl_xor(tri, off); const float ou = __shfl_xor(hu, off), ov = __shfl_xor(hv, off);
        if (ot < t || (ot == t && orec < rec_best)) { t = ot; rec_best = orec; tri = otri; hu = ou; hv = ov; }
    }
    // a lane whose own closest hit is another triangle within that distance of the wave's: a near tie across lanes
    if (ORDERED && RT_NEAR_TIES && tri_own != tri && fabsf(t_own - t) <= tol + p.tie_rel * fminf(t_own, t)) near_tie = true;
    if (tri < 0 && __ballot(detected) != 0ull) tri = RT_HIT_DETECTED;
    hit.t = t; hit.id = tri; hit.u = hu; hit.v = hv;
    best_rec = rec_best;
    return __ballot(near_tie) != 0ull;
}

// KDTreeCPU::intersect's own sequence of decisions for one ray (reference order, every box pruned against the closest hit of
// the moment it is reached, a hit accepted only when strictly closer), made by the whole wave.  A window of 64 consecutive
// nodes of the reference preorder is decided with the closest t known at its start; that is exactly what the one-lane walk
// would have used up to the first leaf of the window that finds a closer hit.  The reached leaves are tested in preorder, four
// at a time with sixteen lanes each (one record per lane and pass); the first of them that holds a closer hit ends the window:
// the closest of its hits (the earliest record among equals, as the strict comparison of the sequential loop leaves it) becomes
// the closest hit, the next window starts right behind that leaf, and what the lanes behind it decided or tested is discarded
// and recomputed.  A ray accepts a closer hit a few dozen times at most, so the walk takes about (nodes spanned / 64) + (hits
// accepted) windows instead of one dependent read per node.  Used for the rays whose front-to-back walk met a near tie, for the
// rays the short forms of the tests are not for, and for the last walks of a wave in the reference's order.
// `start`, hit: the node cursor and the closest hit of a walk in the reference's order that this one continues (0 and no hit: the
// whole walk; all arguments wave-uniform).  PLAIN: the short forms of the box and triangle tests (see slab_plain_ray).
template <bool STATS, bool PLAIN>
__device__ __forceinline__ void walk_reference_cooperative(const RtParams& p, F3 O, F3 D, F3 invD, uint32_t lane, uint32_t start, Hit& hit,
                                                           uint32_t& n_nodes_visited, uint32_t& n_tris_tested)
{
    float t_cur = hit.t, hu = hit.u, hv = hit.v;            // wave-uniform
    int32_t tri = hit.id >= 0 ? hit.id : -1;
    bool detected = hit.id == RT_HIT_DETECTED;              // wave-uniform
    const uint32_t end = p.n_nodes;
    start = checked(p, start, end + 1u, 17);
    uint32_t base = start, killed_to = start;
    while (base < end) {
        const uint32_t j = base + lane;
        const bool valid = j < end;
        float4 n0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), n1 = n0;
        if (valid) { n0 = p.nodes[2 * (size_t)j]; n1 = p.nodes[2 * (size_t)j + 1]; }
        float tNear;
        const bool box = slab_test_t<PLAIN>(n0, n1, O, invD, tNear);
        const bool descend = valid && box && !(tNear > t_cur);
        const uint32_t skip = __float_as_uint(n0.w);
        const int32_t leaf_first = (int32_t)__float_as_uint(n1.w);
        const uint32_t pm = wave_prefix_max((valid && !descend) ? skip : 0u);
        const uint32_t limit = pm > killed_to ? pm : killed_to;
        const bool reached = valid && j >= limit;           // exact for the nodes that are not rejected, which is all that is asked below
        unsigned long long leaves = __ballot(reached && descend && leaf_first >= 0);
        uint32_t upto = 64u;                                // lanes of this window whose decisions stand
        while (leaves != 0ull) {
            uint32_t group_first = RT_NO_REC, group_node = 0;       // this 16-lane group's leaf: first record, lane of the window that holds it
#pragma unroll
            for (uint32_t g = 0; g < 4u; ++g) {
                if (leaves != 0ull) {
                    const int src = __ffsll((long long)leaves) - 1;
                    leaves &= leaves - 1ull;
                    const uint32_t first = checked(p, (uint32_t)__builtin_amdgcn_readlane(leaf_first, src), p.n_refs, 16);
                    if ((lane >> 4) == g) { group_first = first; group_node = (uint32_t)src; }
                }
            }
            float lt = __builtin_inff(), lu = 0.0f, lv = 0.0f;       // this lane's closest hit below t_cur
            int32_t ltri = -1;
            uint32_t lrec = RT_NO_REC, tested = 0;
            bool succeeded = false;
            bool open = group_first != RT_NO_REC;
            uint32_t rec_index = group_first + (lane & 15u);
            while (__ballot(open) != 0ull) {
                const uint32_t at = rec_index < p.n_refs ? rec_index : p.n_refs - 1u;
                float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0;
                if (open) { const float4* rec = p.leafrecs + 3 * (size_t)at; r0 = rec[0]; r1 = rec[1]; r2 = rec[2]; }
                const uint32_t tag = __float_as_uint(r2.y);
                const unsigned long long ends = __ballot(open && (tag & 0x80000000u) != 0u);
                const uint32_t group_ends = (uint32_t)(ends >> (lane & 48u)) & 0xFFFFu;
                const bool counts = open && rec_index < p.n_refs && (group_ends & ((1u << (lane & 15u)) - 1u)) == 0u;
                if (counts) ++tested;
                float tt, u, v;
                if (tri_test_t<PLAIN>(r0, r1, r2, O, D, tt, u, v) && counts) {
                    succeeded = true;
                    if (tt < t_cur && tt < lt) { lt = tt; lrec = rec_index; ltri = (int32_t)(tag & 0x7FFFFFFFu); lu = u; lv = v; }
                }
                if (group_ends != 0u) open = false;
                rec_index += 16u;
            }
            // the first leaf (lowest group) with a closer hit, and the closest hit of that leaf
            const unsigned long long closer = __ballot(ltri >= 0);
            uint32_t groups_standing = 4u;
            if (closer != 0ull) {
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) {     // within the 16-lane groups: smallest t, then earliest record
                    const float ot = __shfl_xor(lt, off); const uint32_t orec = __shfl_xor(lrec, off);
                    const int32_t otri = __shfl_xor(ltri, off); const float ou = __shfl_xor(lu, off), ov = __shfl_xor(lv, off);
                    if (ot < lt || (ot == lt && orec < lrec)) { lt = ot; lrec = orec; ltri = otri; lu = ou; lv = ov; }
                }
                const int f = ((__ffsll((long long)closer) - 1) >> 4) << 4;      // first lane of that group
                t_cur = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lt), f));
                hu = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lu), f));
                hv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lv), f));
                tri = __builtin_amdgcn_readlane(ltri, f);
                upto = (uint32_t)__builtin_amdgcn_readlane((int)group_node, f) + 1u;
                groups_standing = ((uint32_t)f >> 4) + 1u;
                leaves = 0ull;                              // what lies behind that leaf is decided again, with the new t
            }
            if (__ballot(succeeded && (lane >> 4) < groups_standing) != 0ull) detected = true;     // (only the leaves whose tests stand)
            if (STATS && (lane >> 4) < groups_standing) n_tris_tested += tested;
        }
        if (STATS) {
            uint32_t before = __shfl_up(pm, 1);             // rejected subtrees' ends over the lanes before this one
            if (lane == 0) before = 0u;
            if (lane < upto && valid && j >= (before > killed_to ? before : killed_to)) n_nodes_visited++;
        }
        const uint32_t kills = (uint32_t)__builtin_amdgcn_readlane((int)pm, (int)upto - 1);
        if (kills > killed_to) killed_to = kills;
        base += upto;
        if (killed_to > base) base = killed_to;
    }
    if (tri < 0 && detected) tri = RT_HIT_DETECTED;
    hit.t = t_cur; hit.id = tri; hit.u = hu; hit.v = hv;
}

// Util::RandomUnitVector (Renderer.h:46-54)
__device__ __forceinline__ F3 random_unit_vector(unsigned long long& rng)
{
    const float TwoPi = 2.0f * 3.14159274f;
    float z = rng_float(rng) * 2.0f - 1.0f;
    float a = rng_float(rng) * TwoPi;
    float r = sqrt_rn(1.0f - z * z);
    float sn, cs;
    sincos_portable(a, sn, cs);
    return f3(r * cs, r * sn, z);
}

constexpr float EPSILON = 0.0002f;         // Renderer.cpp:14

// ---- one lane = one path at a time ---------------------------------------------------------------------------------------
// Path state of Renderer::PerPixel (Renderer.cpp:138-229) kept across loop iterations so that a lane whose path has ended
// can start its next pixel while its neighbours are still bouncing.
struct Path {
    F3 O, D, contribution, finalColor;
    unsigned long long rng;
    uint32_t bounce;
    uint32_t column_slot;          // slot + 1 in the column-ray tables when this pixel's primary ray has direction.x == +0
};

__device__ __forceinline__ void path_begin(Path& s, const RtParams& p, uint32_t pixel, uint32_t frame)
{
    const float4 d = p.dirs[pixel];
    s.O = f3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
    s.D = f3(d.x, d.y, d.z);
    s.contribution = f3(1.0f, 1.0f, 1.0f);
    s.finalColor = f3(0.0f, 0.0f, 0.0f);
    s.rng = pixel_seed(pixel, frame, p.seed);
    s.bounce = 0;
    s.column_slot = __float_as_uint(d.w);
}

// Where the closest-hit lookups (sphere centre and material by per-lane index) read from: a copy in LDS when the scene is
// small enough (every scene of the reference application is), else HBM.
struct SceneLds {
    float4 spheres[RT_LDS_SPHERES];
    uint32_t sphere_mat[RT_LDS_SPHERES];
    float4 materials[3 * RT_LDS_MATERIALS];
};

// A lock in LDS taken by one wave at a time (lane 0 spins, the wave follows), and a look at a word other waves change.
__device__ __forceinline__ uint32_t lds_peek(uint32_t* word)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void lds_lock(uint32_t* lock, uint32_t lane)
{
    if (lane == 0) {
        while (atomicCAS(lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lds_unlock(uint32_t* lock, uint32_t lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Sphere frames, end of a workgroup's share: every wave still holds paths, but fewer each bounce, and an iteration costs a
// wave the same whether 60 lanes or 6 are in use.  A wave down to RT_SPHERE_MERGE paths hands them to the pool and ends; the
// next wave that comes by with idle lanes takes them.  Half as many waves then make the same bounces.
// count: paths waiting; alive: waves of the workgroup still in the loop (the last one never hands its paths over).
struct SpherePool {
    uint32_t lock, count, alive, pad;
    uint32_t words[RT_SPHERE_POOL_WORDS][RT_SPHERE_POOL_SLOTS];
};

// All threads of the workgroup call this once before tracing (the host picks the LDS variant only for scenes that fit).
__device__ __forceinline__ void stage_scene(const RtParams& p, SceneLds& lds)
{
    for (uint32_t i = threadIdx.x; i < p.n_spheres; i += blockDim.x) { lds.spheres[i] = p.spheres[i]; lds.sphere_mat[i] = p.sphere_mat[i]; }
    for (uint32_t i = threadIdx.x; i < 3u * p.n_materials; i += blockDim.x) lds.materials[i] = p.materials[i];
}

// What follows TraceRay in one iteration of the bounce loop (Renderer.cpp:160-226): miss, closest-hit, the next ray, emission,
// albedo and roulette.  `first` / `last`: this was the path's first bounce / the bounce budget is used up after it.  Returns
// true when the path has ended.
template <int MODE, bool in_lds>
__device__ __forceinline__ bool shade_hit(Path& s, const RtParams& p, const SceneLds& lds, int32_t* primary_id, bool first, bool last,
                                          float tS, int iS, Hit th)
{
    const F3 O = s.O, D = s.D;
    if (iS < 0 && th.id < 0) {                             // Miss(): finalColor += contribution * ambient (= 0), break
        s.finalColor = s.finalColor + s.contribution * f3(0.0f, 0.0f, 0.0f);
        if (primary_id && first) *primary_id = -1;
        return true;
    }
    F3 P, N;
    uint32_t matIndex;
    if (MODE == RT_MODE_SPHERES || tS < th.t) {             // ClosestHitSphere (Renderer.cpp:303-317); a sphere hit has t < FLT_MAX
        iS = (int)checked(p, (uint32_t)iS, p.n_spheres, 4);
        const float4 sp = in_lds ? lds.spheres[iS] : p.spheres[iS];      // per-lane index
        P = O + D * tS;
        N = normalize3(P - f3(sp.x, sp.y, sp.z));
        matIndex = in_lds ? lds.sphere_mat[iS] : p.sphere_mat[iS];
        if (primary_id && first) *primary_id = iS;
    } else {                                                // ClosestHitTriangle (Renderer.cpp:319-332)
        const float4* sh = p.trishade + 3 * (size_t)checked(p, (uint32_t)th.id, p.n_tris, 3);
        float4 s0 = sh[0], s1 = sh[1], s2 = sh[2];
        P = O + D * th.t;
        F3 nrm = (f3(s0.x, s0.y, s0.z) * ((1.0f - th.u) - th.v) + f3(s0.w, s1.x, s1.y) * th.u) + f3(s1.z, s1.w, s2.x) * th.v;
        N = normalize3(nrm);
        matIndex = __float_as_uint(s2.y);
        if (primary_id && first) *primary_id = RT_ID_TRI_BIT | th.id;
    }

    matIndex = checked(p, matIndex, p.n_materials, 5);
    const float4 m0 = in_lds ? lds.materials[3 * matIndex] : p.materials[3 * matIndex];             // albedo.rgb, roughness
    const float4 m1 = in_lds ? lds.materials[3 * matIndex + 1] : p.materials[3 * matIndex + 1];     // emission.rgb, transparency
    bool hitInside = dot3(D, N) > 0.0f;
    F3 normalSurface = hitInside ? -N : N;
    s.O = P + normalSurface * EPSILON;

    if (m1.w > 0.0f) {                                      // always transmit (Renderer.cpp:176-197)
        const float IOR = in_lds ? lds.materials[3 * matIndex + 2].x : p.materials[3 * matIndex + 2].x;
        // RefractionRay (Renderer.cpp:334-361)
        F3 ref_n = N;
        float eta_t = IOR, eta_i = 1.0f;
        float i_dot_n = dot3(D, N);
        if (i_dot_n < 0.0f) i_dot_n = -i_dot_n;
        else { ref_n = -N; eta_t = 1.0f; eta_i = IOR; }
        float eta = eta_i / eta_t;
        float k = 1.0f - (eta * eta) * (1.0f - i_dot_n * i_dot_n);
        if (!(k < 0.0f)) {
            s.O = P + ref_n * (-EPSILON);
            s.D = normalize3((D + ref_n * i_dot_n) * eta - ref_n * sqrt_rn(k));
            return last;                                    // "continue": skips emission, albedo and roulette
        }
        // total internal reflection: falls through with the direction unchanged
    } else {
        F3 diffuseRayDir = normalize3(N + random_unit_vector(s.rng));
        F3 reflectedVector = reflect3(D, N);
        s.D = normalize3(mix3(reflectedVector, diffuseRayDir, m0.w * m0.w));
    }

    s.finalColor = s.finalColor + f3(m1.x, m1.y, m1.z) * s.contribution;
    s.contribution = s.contribution * f3(m0.x, m0.y, m0.z);

    float pr = sel_max(s.contribution.x, sel_max(s.contribution.y, s.contribution.z));
    if (rng_float(s.rng) > pr) return true;                 // Russian roulette (Renderer.cpp:218-225)
    s.contribution = s.contribution * (1.0f / pr);
    return last;
}

// One iteration of the bounce loop (Renderer.cpp:149-226) in a scene of spheres.  Returns true when the path has ended (miss,
// roulette, or the bounce budget is used up).
template <bool in_lds>
__device__ __forceinline__ bool path_step(Path& s, const RtParams& p, const SceneLds& lds, int32_t* primary_id, uint32_t& n_rays)
{
    if (s.bounce >= p.bounces) return true;                // "for (i = 0; i < Bounces; i++)" with Bounces == 0
    const bool first = s.bounce == 0;
    s.bounce++;
    const bool last = s.bounce >= p.bounces;
    n_rays++;
    float tS = FLT_MAX; int iS = -1;                        // TraceRay (Renderer.cpp:231-301)
    if (RT_SPHERE_LDS_LOOP && in_lds) trace_spheres_lds(lds.spheres, p.n_spheres, s.O, s.D, tS, iS);
    else trace_spheres(p, s.O, s.D, tS, iS);
    Hit th; th.t = FLT_MAX; th.id = -1; th.u = 0.0f; th.v = 0.0f;
    return shade_hit<RT_MODE_SPHERES, in_lds>(s, p, lds, primary_id, first, last, tS, iS, th);
}

// Pixel `slot` of work unit `unit`.  Units below n_units_main are the four 8x2 strips of each 8x8 tile of the 16-row bands this
// context renders; tiles are numbered down the columns of tiles.  Units from n_units_main on (only under a tile split with the
// 3x3 filter on) are 16x1 strips of the rows just below and just above each owned band: the filter reads their sums.
__device__ __forceinline__ bool unit_pixel(const RtParams& p, uint32_t unit, uint32_t slot, uint32_t& pixel)
{
    if (unit >= p.n_units_main) {
        const uint32_t hu = unit - p.n_units_main, chunks = (p.width + 15u) / 16u;
        const uint32_t r = hu / chunks, chunk = hu - r * chunks;
        const uint32_t band = (r >> 1) * p.band_count + p.band_index;
        const uint32_t x = chunk * 16u + slot;
        const uint32_t y = (r & 1u) ? band * RT_TILE + RT_TILE : band * RT_TILE - 1u;      // band 0, lower side: wraps, rejected below
        pixel = y * p.width + x;
        return x < p.width && y < p.height;
    }
    const uint32_t tile = unit >> 2;
    const uint32_t within = ((unit & 3u) << 4) | slot;
    const uint32_t tcol = tile / p.tile_rows, trow = tile - tcol * p.tile_rows;
    const uint32_t band = (trow >> 1) * p.band_count + p.band_index;
    const uint32_t x = tcol * 8u + (within & 7u);
    const uint32_t y = band * RT_TILE + (trow & 1u) * 8u + (within >> 3);
    pixel = y * p.width + x;
    return x < p.width && y < p.height;
}

// Row y belongs to this context: one of its bands, or (with the halo on) a row bordering one.
__device__ __forceinline__ bool row_rendered(const RtParams& p, uint32_t y)
{
    if (((y / RT_TILE) % p.band_count) == p.band_index) return true;
    if (!p.halo) return false;
    const uint32_t in_band = y % RT_TILE;
    if (in_band == RT_TILE - 1u && y + 1u < p.height && (((y + 1u) / RT_TILE) % p.band_count) == p.band_index) return true;
    return in_band == 0u && y > 0u && (((y - 1u) / RT_TILE) % p.band_count) == p.band_index;
}

}  // namespace

// ---- kernels ------------------------------------------------------------------------------------------------------
// Camera::RecalculateRayDirections (Camera.cpp:134-150): one direction per pixel, cached in HBM until the camera or the
// frame size changes -- as the reference caches them in the Camera.
__global__ __launch_bounds__(256) void rt_raygen_kernel(RtParams p)
{
    const uint32_t pixel = blockIdx.x * 256u + threadIdx.x;
    if (pixel >= p.width * p.height) return;
    const uint32_t y = pixel / p.width, x = pixel - y * p.width;
    F3 d = camera_dir(p, x, y);
    p.dirs_out[pixel] = make_float4(d.x, d.y, d.z, 0.0f);
}

// ---- column rays: primary rays with direction.x == +0 (see trace_mesh_cooperative for why they cannot be pruned) ------------
// They share the origin and, because the NaN of the x slab overrides the other two axes, the same set of visited leaves: about
// half the tree, whatever the ray looks at.  Tracing them lane by lane costs ~10^5 dependent steps per ray, so they are resolved
// for the whole frame by a pre-pass instead:
//   when the directions change : rt_column_count / _scan / _assign number them in pixel order (slot + 1 goes into dirs[].w, the
//                                directions into a compact array) and rt_column_chunks boxes the directions of every 16, 64 and
//                                256 consecutive slots;
//   every frame                : rt_column_reach_* marks the visited nodes (a prefix maximum over the preorder array),
//                                rt_column_filter proves, per visited leaf record and per box of rays, that no ray of the box can
//                                pass the triangle test, and queues the (record, 16 rays) pairs it cannot decide;
//                                rt_column_exact runs the real triangle test on those and keeps min (t, record index) per ray.
// The proof is interval arithmetic over the very operations of tri_test: every floating-point operation is monotone in each
// operand and rounding is monotone, so the values tri_test would compute for ANY ray of the box lie between the values
// computed at the corners of the box.  A pair is dropped only when a rejecting comparison of tri_test holds over the whole
// interval, i.e. for every ray in it; everything else is tested exactly.  Result: bit for bit what testing every record
// against every column ray gives, at a few percent of the work.

struct Iv { float lo, hi; };
__device__ __forceinline__ Iv iv_mul_scalar(Iv a, float b)          // a * b, b one value
{
    const float x = a.lo * b, y = a.hi * b;
    Iv r; r.lo = (y < x) ? y : x; r.hi = (y < x) ? x : y; return r;
}
__device__ __forceinline__ Iv iv_mul(Iv a, Iv b)
{
    const float p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    const float l01 = (p1 < p0) ? p1 : p0, h01 = (p1 < p0) ? p0 : p1;
    const float l23 = (p3 < p2) ? p3 : p2, h23 = (p3 < p2) ? p2 : p3;
    Iv r; r.lo = (l23 < l01) ? l23 : l01; r.hi = (h01 < h23) ? h23 : h01; return r;
}
__device__ __forceinline__ Iv iv_add(Iv a, Iv b) { Iv r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi; return r; }
__device__ __forceinline__ Iv iv_sub(Iv a, Iv b) { Iv r; r.lo = a.lo - b.hi; r.hi = a.hi - b.lo; return r; }

// What a leaf record contributes to tri_test independently of the ray (all column rays start at the camera position).
struct ColumnTri { F3 e1, e2, s; float qy, qz, e2q; };

__device__ __forceinline__ ColumnTri column_tri(float4 r0, float4 r1, float4 r2, F3 O)
{
    ColumnTri c;
    const F3 v0 = f3(r0.x, r0.y, r0.z);
    c.e1 = f3(r0.w, r1.x, r1.y); c.e2 = f3(r1.z, r1.w, r2.x);
    c.s = O - v0;
    const F3 q = cross3(c.s, c.e1);
    c.qy = q.y; c.qz = q.z;
    c.e2q = dot3(c.e2, q);
    return c;
}

// True when tri_test fails for EVERY direction (+0, dy, dz) with dy in [b.x, b.y], dz in [b.z, b.w].  With direction.x == +0
// tri_test's h = cross(D, e2) is (dy*e2.z - e2.y*dz, dz*e2.x, -(e2.x*dy)) and dot(D, q) is dy*q.y + dz*q.z: the terms
// multiplied by the zero vanish exactly (x - 0 = x, 0 - x = -x, 0 + x = x for finite values).
__device__ __forceinline__ bool column_box_rejects(const ColumnTri& c, float4 b)
{
    const Iv DY = {b.x, b.y}, DZ = {b.z, b.w};
    const Iv hx = iv_sub(iv_mul_scalar(DY, c.e2.z), iv_mul_scalar(DZ, c.e2.y));
    const Iv hy = iv_mul_scalar(DZ, c.e2.x);
    const Iv t0 = iv_mul_scalar(DY, c.e2.x);
    const Iv hz = {-t0.hi, -t0.lo};
    const Iv a = iv_add(iv_add(iv_mul_scalar(hx, c.e1.x), iv_mul_scalar(hy, c.e1.y)), iv_mul_scalar(hz, c.e1.z));
    if (a.lo > -0.00001f && a.hi < 0.00001f) return true;           // "a > -EPS && a < EPS" for every ray
    if (!(a.lo > 1e-30f || a.hi < -1e-30f)) return false;           // some ray may see a (near) zero determinant
    const Iv f = {1.0f / a.hi, 1.0f / a.lo};
    const Iv du = iv_add(iv_add(iv_mul_scalar(hx, c.s.x), iv_mul_scalar(hy, c.s.y)), iv_mul_scalar(hz, c.s.z));
    const Iv u = iv_mul(f, du);
    if (u.hi < 0.0f || u.lo > 1.0f) return true;
    const Iv dv = iv_add(iv_mul_scalar(DY, c.qy), iv_mul_scalar(DZ, c.qz));
    const Iv v = iv_mul(f, dv);
    if (v.hi < 0.0f || u.lo + v.lo > 1.0f) return true;
    const Iv t = iv_mul_scalar(f, c.e2q);
    return !(t.hi > 0.00001f);
}

// -- numbering (runs when directions, bands or size change) --
__device__ __forceinline__ bool is_column_pixel(const RtParams& p, uint32_t pixel, float4& d)
{
    if (pixel >= p.width * p.height) return false;
    const uint32_t y = pixel / p.width;
    d = p.dirs_out[pixel];
    return row_rendered(p, y) && __float_as_uint(d.x) == 0u;                     // +0 exactly; -0 rays take the in-kernel path
}

__global__ __launch_bounds__(256) void rt_column_count_kernel(RtParams p)
{
    __shared__ uint32_t wave_count[4];
    float4 d;
    const bool is = is_column_pixel(p, blockIdx.x * 256u + threadIdx.x, d);
    const uint32_t n = (uint32_t)__popcll(__ballot(is));
    if ((threadIdx.x & 63u) == 0u) wave_count[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) p.column_block[blockIdx.x] = wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
}

// Exclusive prefix sum of the per-block counts, in place, by one workgroup; the total goes to *column_count.
__global__ __launch_bounds__(1024) void rt_column_scan_kernel(RtParams p, uint32_t n_blocks)
{
    __shared__ uint32_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_blocks + 1023u) / 1024u;
    const uint32_t lo = tid * per, hi = (lo + per < n_blocks) ? lo + per : n_blocks;
    // (eight reads in flight per thread: one after the other, a thread's few dozen reads are a chain of as many memory latencies)
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i += 8u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) v[k] = i + k < hi ? p.column_block[i + k] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) sum += v[k];
    }
    part[tid] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {               // Hillis-Steele inclusive scan
        const uint32_t v = tid >= off ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0u;
    for (uint32_t i = lo; i < hi; i += 8u) {
        uint32_t v[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) v[k] = i + k < hi ? p.column_block[i + k] : 0u;
#pragma unroll
        for (uint32_t k = 0; k < 8u; ++k) if (i + k < hi) { p.column_block[i + k] = run; run += v[k]; }
    }
    if (tid == 1023u) *p.column_count = part[1023];
}

__global__ __launch_bounds__(256) void rt_column_assign_kernel(RtParams p)
{
    __shared__ uint32_t wave_count[4];
    const uint32_t pixel = blockIdx.x * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    float4 d;
    const bool is = is_column_pixel(p, pixel, d);
    const unsigned long long m = __ballot(is);
    if (lane == 0u) wave_count[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t slot = p.column_block[blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    for (uint32_t w = 0; w < wv; ++w) slot += wave_count[w];
    if (pixel < p.width * p.height) {
        uint32_t slot1 = 0u;
        if (is && slot < p.column_cap) {
            p.column_pixels[slot] = pixel;
            p.column_dirs[slot] = make_float4(d.x, d.y, d.z, 0.0f);
            slot1 = slot + 1u;
        }
        d.w = __uint_as_float(slot1);
        p.dirs_out[pixel] = d;
    }
}

// Boxes of the directions of 16, 64 and 256 consecutive slots: { min dy, max dy, min dz, max dz }.  A box holding a
// non-finite or huge component must not be used for the proof: it is stored with dy_lo = NaN, which the filter checks first.
// One workgroup per 256 slots, one lane per slot: minima and maxima by butterfly exchanges over 16 and 64 lanes, the four
// waves' boxes combined through LDS.  (One thread per box, reading up to 256 directions one after the other, took 44 us.)
__global__ __launch_bounds__(256) void rt_column_chunks_kernel(RtParams p)
{
    __shared__ float4 wave_box[4];
    __shared__ int wave_ok[4];
    const uint32_t count = *p.column_count < p.column_cap ? *p.column_count : p.column_cap;
    if (blockIdx.x * 256u >= count) return;
    const uint32_t k = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const bool have = k < count;
    const float inf = __builtin_inff();
    float ylo = inf, yhi = -inf, zlo = inf, zhi = -inf;
    int ok = 1;
    if (have) {
        const float4 d = p.column_dirs[k];
        ok = (fabsf(d.y) < 1e9f && fabsf(d.z) < 1e9f) ? 1 : 0;                 // false for NaN and infinities too
        ylo = yhi = d.y; zlo = zhi = d.z;
    }
    auto combine = [&](int off) {
        const float a = __shfl_xor(ylo, off), b = __shfl_xor(yhi, off), c = __shfl_xor(zlo, off), d = __shfl_xor(zhi, off);
        ylo = (a < ylo) ? a : ylo; yhi = (b > yhi) ? b : yhi; zlo = (c < zlo) ? c : zlo; zhi = (d > zhi) ? d : zhi;
        ok &= __shfl_xor(ok, off);
    };
    auto box = [&]() { return ok ? make_float4(ylo, yhi, zlo, zhi) : make_float4(__builtin_nanf(""), 0.0f, 0.0f, 0.0f); };
    combine(1); combine(2); combine(4); combine(8);
    if ((lane & 15u) == 0u && have) p.column_box16[k >> 4] = box();
    combine(16); combine(32);
    if (lane == 0u && have) p.column_box64[k >> 6] = box();
    if (lane == 0u) { wave_box[wv] = make_float4(ylo, yhi, zlo, zhi); wave_ok[wv] = ok; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (uint32_t w = 1; w < 4u; ++w) {
            const float4 o = wave_box[w];
            ylo = (o.x < ylo) ? o.x : ylo; yhi = (o.y > yhi) ? o.y : yhi; zlo = (o.z < zlo) ? o.z : zlo; zhi = (o.w > zhi) ? o.w : zhi;
            ok &= wave_ok[w];
        }
        p.column_box256[blockIdx.x] = box();
    }
}

// -- per frame --
// Visited nodes.  In the preorder array a node is skipped exactly when some earlier node whose box the ray misses has a skip
// link beyond it (that node is an ancestor, or lies in an already skipped subtree whose range is nested in an ancestor's), so
// "visited" is an exclusive prefix maximum of (box missed ? skip link : 0) compared with the node's own index.  Three steps:
// per-block maxima, a scan of those by one workgroup, the per-node result.  The ray of slot 0 stands for all column rays.
__device__ __forceinline__ uint32_t column_kill_end(const RtParams& p, uint32_t j, F3 O, F3 invD, bool& box, float& tNear)
{
    box = false; tNear = 0.0f;
    if (j >= p.n_nodes) return 0u;
    const float4 n0 = p.nodes[2 * (size_t)j], n1 = p.nodes[2 * (size_t)j + 1];
    box = slab_test(n0, n1, O, invD, tNear);
    return box ? 0u : __float_as_uint(n0.w);
}

__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v, uint32_t lane)
{
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(v, off);
        if ((int)lane >= off) v = v > o ? v : o;
    }
    return v;
}

// The ray the reach / test kernels work for (p.ray_slot): the column rays' representative (slot 0 of the family; `count` =
// size of the family), or the representative of a family of parked bounce rays.  False when there is nothing to do.
__device__ __forceinline__ bool reach_ray(const RtParams& p, F3& O, F3& D, uint32_t& count)
{
    if (p.ray_slot < 0) {
        count = *p.column_count < p.column_cap ? *p.column_count : p.column_cap;
        if (count == 0u) return false;
        const float4 d4 = p.column_dirs[0];
        O = f3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]); D = f3(d4.x, d4.y, d4.z);
        return true;
    }
    const int32_t rep = p.park_info[p.ray_slot];
    if (rep < 0) return false;
    const float4 o4 = p.resume_paths[RT_PARK_WORDS * rep], d4 = p.resume_paths[RT_PARK_WORDS * rep + 1];
    O = f3(o4.x, o4.y, o4.z); D = f3(d4.x, d4.y, d4.z); count = (uint32_t)p.park_info[RT_PARK_FAMILIES + p.ray_slot];
    return true;
}

// Counting kernels: *dst += weight x (threads of this 256-thread workgroup with `flag`), one atomic per workgroup.  Every thread of the
// workgroup must call it.  (One atomic per wave onto the same word made the counting passes of a 10 M-triangle tree four times as
// long as the passes themselves.)
__device__ __forceinline__ void block_count_add(bool flag, unsigned long long* dst, uint32_t weight)
{
    __shared__ uint32_t wave_n[4];
    const uint32_t n = (uint32_t)__popcll(__ballot(flag));
    if ((threadIdx.x & 63u) == 0u) wave_n[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t sum = (wave_n[0] + wave_n[1]) + (wave_n[2] + wave_n[3]);
        if (sum) atomicAdd(dst, (unsigned long long)sum * weight);
    }
}

__global__ __launch_bounds__(256) void rt_column_reach_max_kernel(RtParams p)
{
    __shared__ uint32_t wave_max[4];
    F3 O, D; uint32_t count;
    if (!reach_ray(p, O, D, count)) return;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (p.ray_slot < 0 && gid == 0u) { *p.column_queue_count = 0u; *p.column_flags = 0u; }      // a new queue is about to be built
    const F3 invD = f3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
    bool box; float tNear;
    uint32_t m = column_kill_end(p, gid, O, invD, box, tNear);
    m = wave_inclusive_max(m, threadIdx.x & 63u);
    if ((threadIdx.x & 63u) == 63u) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t a = wave_max[0] > wave_max[1] ? wave_max[0] : wave_max[1];
        uint32_t b = wave_max[2] > wave_max[3] ? wave_max[2] : wave_max[3];
        p.column_block[blockIdx.x] = a > b ? a : b;
    }
}

// Exclusive prefix maximum of the per-block maxima, in place, by one workgroup.  Each of its sixteen waves owns a run of the
// array and goes through it 64 consecutive values at a time, four reads in flight (a contiguous range per THREAD would make
// every read of a wave touch 64 cache lines: with a tree of 14 M nodes this kernel then took longer than the two passes over the
// nodes on either side of it).
__global__ __launch_bounds__(1024) void rt_column_reach_scan_kernel(RtParams p, uint32_t n_blocks)
{
    __shared__ uint32_t part[16];
    F3 O, D; uint32_t count;
    if (!reach_ray(p, O, D, count)) return;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t seg = (((n_blocks + 15u) / 16u) + 63u) & ~63u;
    const uint32_t lo = wv * seg < n_blocks ? wv * seg : n_blocks, hi = lo + seg < n_blocks ? lo + seg : n_blocks;
    uint32_t* const blk = p.column_block;
    uint32_t mx = 0;
    for (uint32_t base = lo; base < hi; base += 256u) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { const uint32_t i = base + 64u * k + lane; v[k] = i < hi ? blk[i] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) mx = mx > v[k] ? mx : v[k];
    }
    mx = (uint32_t)__shfl((int)wave_inclusive_max(mx, lane), 63);
    if (lane == 0u) part[wv] = mx;
    __syncthreads();
    uint32_t run = 0;                                               // maximum over everything before this wave's run
    for (uint32_t w = 0; w < wv; ++w) run = run > part[w] ? run : part[w];
    for (uint32_t base = lo; base < hi; base += 256u) {
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) { const uint32_t i = base + 64u * k + lane; v[k] = i < hi ? blk[i] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            const uint32_t i = base + 64u * k + lane;
            const uint32_t incl = wave_inclusive_max(v[k], lane);
            uint32_t excl = __shfl_up(incl, 1);
            if (lane == 0u) excl = 0u;
            if (i < hi) blk[i] = run > excl ? run : excl;
            const uint32_t all = (uint32_t)__shfl((int)incl, 63);
            run = run > all ? run : all;
        }
    }
}

__global__ __launch_bounds__(256) void rt_column_reach_mark_kernel(RtParams p, int stats)
{
    __shared__ uint32_t wave_max[4];
    F3 O, D; uint32_t count;
    if (!reach_ray(p, O, D, count)) return;
    const uint32_t j = blockIdx.x * 256u + threadIdx.x, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const F3 invD = f3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
    bool box; float tNear;
    const uint32_t own = column_kill_end(p, j, O, invD, box, tNear);
    const uint32_t incl = wave_inclusive_max(own, lane);
    if (lane == 63u) wave_max[wv] = incl;
    __syncthreads();
    uint32_t before = p.column_block[blockIdx.x];                   // maximum over all earlier blocks
    for (uint32_t w = 0; w < wv; ++w) before = before > wave_max[w] ? before : wave_max[w];
    uint32_t excl = __shfl_up(incl, 1);
    if (lane == 0u) excl = 0u;
    const uint32_t limit = excl > before ? excl : before;
    const bool visited = j < p.n_nodes && j >= limit;
    // (one bit per node, a word per wave: a byte store per lane made this kernel ten times as long as the pass before it)
    const unsigned long long entered = __ballot(visited && box);
    if (lane == 0u && (j & ~63u) < p.n_nodes) p.reached[j >> 6] = entered;
    // a visited box with a real t_near could be pruned by the closest t so far: then the visited set is not fixed
    if (__ballot(visited && box && tNear == tNear) != 0ull && lane == 0u) atomicOr(p.ray_slot < 0 ? p.column_flags : (uint32_t*)p.park_info + 2 * RT_PARK_FAMILIES + p.ray_slot, 1u);
    if (stats) block_count_add(visited, &p.column_stats[0], count);        // every ray of the family visits these nodes
}

// Parked rays (direction.x == 0, see tree_body).  With an infinite inverse in x the slab test hits every box on one side of
// origin.x and misses every box on the other, whatever y and z are, so all parked rays with the same origin.x and the same
// sign of the zero visit the same nodes: a family.  (A glass sphere centred on the camera's x turns the whole centre column
// of primary rays into such bounce rays: same plane, hundreds per frame.)  This kernel deals the slots of the list being
// resolved into at most RT_PARK_FAMILIES families; further families get no whole-chip result and are walked by their wave.
// Each member gets a sort key (pixel, frame, slot): neighbouring pixels -- and the same pixel in the other frames of a launch,
// whose path up to a refraction is the same -- have neighbouring or equal rays, which is what the boxes below need.
__global__ __launch_bounds__(1024) void rt_park_group_kernel(RtParams p)
{
    __shared__ uint32_t key0[2], rep1, sizes[RT_PARK_FAMILIES];
    const uint32_t n = *p.resume_count < RT_PARK_SLOTS ? *p.resume_count : RT_PARK_SLOTS;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) { rep1 = 0xFFFFFFFFu; sizes[0] = sizes[1] = 0u; }
    if (tid < 3 * RT_PARK_FAMILIES) p.park_info[tid] = tid < RT_PARK_FAMILIES ? -1 : 0;
    if (tid < RT_PARK_FAMILIES) p.park_queue_count[tid] = 0u;
    if (n == 0u) return;
    if (tid == 0) { key0[0] = __float_as_uint(p.resume_paths[0].x); key0[1] = __float_as_uint(p.resume_paths[1].x); }
    __syncthreads();
    for (uint32_t sl = tid; sl < n; sl += blockDim.x) {
        const uint32_t ox = __float_as_uint(p.resume_paths[RT_PARK_WORDS * sl].x), dx = __float_as_uint(p.resume_paths[RT_PARK_WORDS * sl + 1].x);
        if (ox != key0[0] || dx != key0[1]) atomicMin(&rep1, sl);
    }
    __syncthreads();
    uint32_t k1o = 0, k1d = 0;
    if (rep1 != 0xFFFFFFFFu) { k1o = __float_as_uint(p.resume_paths[RT_PARK_WORDS * rep1].x); k1d = __float_as_uint(p.resume_paths[RT_PARK_WORDS * rep1 + 1].x); }
    for (uint32_t sl = tid; sl < n; sl += blockDim.x) {
        const uint32_t ox = __float_as_uint(p.resume_paths[RT_PARK_WORDS * sl].x), dx = __float_as_uint(p.resume_paths[RT_PARK_WORDS * sl + 1].x);
        p.park_best[sl] = ~0ull;
        int fam = -1;
        if (ox == key0[0] && dx == key0[1]) fam = 0;
        else if (rep1 != 0xFFFFFFFFu && ox == k1o && dx == k1d) fam = 1;
        p.park_flags[sl] = fam < 0 ? 1u : 0u;
        if (fam >= 0) {
            const uint32_t pixel = __float_as_uint(p.resume_paths[RT_PARK_WORDS * sl + 2].w), frame = __float_as_uint(p.resume_paths[RT_PARK_WORDS * sl + 3].w);
            static_assert(RT_PARK_SLOTS <= (1 << 14) && RT_MAX_LAUNCH_FRAMES <= (1 << 5), "sort key: pixel << 19 | frame << 14 | slot");
            p.park_keys[fam * RT_PARK_SLOTS + atomicAdd(&sizes[fam], 1u)] = ((unsigned long long)pixel << 19) | ((unsigned long long)(frame & 31u) << 14) | sl;
        }
    }
    __syncthreads();
    if (tid == 0) {
        p.park_info[0] = 0; p.park_info[RT_PARK_FAMILIES] = (int32_t)sizes[0];
        if (rep1 != 0xFFFFFFFFu) { p.park_info[1] = (int32_t)rep1; p.park_info[RT_PARK_FAMILIES + 1] = (int32_t)sizes[1]; }
    }
}

// One workgroup per family: its keys sorted (bitonic, in place), the slot list in that order, and the boxes of the origins and
// directions of every 16, 64 and 256 consecutive rays: { min oy, max oy, min oz, max oz } { min dy, max dy, min dz, max dz }
// (origin.x and the zero direction.x are the family's).  A box holding a non-finite or huge component is stored with a NaN
// first: the filter then proves nothing with it.
__global__ __launch_bounds__(1024) void rt_park_sort_kernel(RtParams p)
{
    const uint32_t fam = blockIdx.x, tid = threadIdx.x;
    if (p.park_info[fam] < 0) return;
    const uint32_t n = (uint32_t)p.park_info[RT_PARK_FAMILIES + fam];
    unsigned long long* keys = p.park_keys + fam * RT_PARK_SLOTS;
    uint32_t* list = p.park_list + fam * RT_PARK_SLOTS;
    uint32_t N = 1; while (N < n) N <<= 1;
    for (uint32_t i = n + tid; i < N; i += 1024u) keys[i] = ~0ull;
    __syncthreads();
    for (uint32_t k = 2; k <= N; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (N >> 1); t += 1024u) {
                const uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u)), l = i | j;     // i has bit j clear
                const unsigned long long a = keys[i], b = keys[l];
                const bool up = (i & k) == 0u;
                if ((a > b) == up) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    for (uint32_t i = tid; i < n; i += 1024u) list[i] = (uint32_t)(keys[i] & (RT_PARK_SLOTS - 1u));
    __syncthreads();
    const uint32_t n16 = (n + 15u) / 16u, n64 = (n + 63u) / 64u, n256 = (n + 255u) / 256u;
    float4* box16 = p.park_boxes + (size_t)fam * 2u * (RT_PARK_BOXES + 1u);
    float4* box64 = box16 + 2u * (RT_PARK_SLOTS / 16u);
    float4* box256 = box64 + 2u * (RT_PARK_SLOTS / 64u);
    for (uint32_t i = tid; i < n16 + n64 + n256; i += 1024u) {
        uint32_t first, last; float4* out;
        if (i < n16) { first = i * 16u; last = first + 16u; out = box16 + 2u * i; }
        else if (i < n16 + n64) { first = (i - n16) * 64u; last = first + 64u; out = box64 + 2u * (i - n16); }
        else { first = (i - n16 - n64) * 256u; last = first + 256u; out = box256 + 2u * (i - n16 - n64); }
        if (last > n) last = n;
        const float inf = __builtin_inff();
        float lo[4] = {inf, inf, inf, inf}, hi[4] = {-inf, -inf, -inf, -inf};
        bool usable = true;
        for (uint32_t r = first; r < last; ++r) {
            const float4 o4 = p.resume_paths[RT_PARK_WORDS * list[r]], d4 = p.resume_paths[RT_PARK_WORDS * list[r] + 1];
            const float c[4] = {o4.y, o4.z, d4.y, d4.z};
            for (int a = 0; a < 4; ++a) {
                usable = usable && fabsf(c[a]) < 1e9f;                        // false for NaN and infinities too
                lo[a] = c[a] < lo[a] ? c[a] : lo[a]; hi[a] = c[a] > hi[a] ? c[a] : hi[a];
            }
        }
        out[0] = usable ? make_float4(lo[0], hi[0], lo[1], hi[1]) : make_float4(__builtin_nanf(""), 0.0f, 0.0f, 0.0f);
        out[1] = make_float4(lo[2], hi[2], lo[3], hi[3]);
    }
    __syncthreads();
    if (tid == 0) {                                                     // the box of the whole family, from its 256-ray boxes
        float4 f0 = box256[0], f1 = box256[1];
        for (uint32_t i = 1; i < n256; ++i) {
            const float4 g0 = box256[2u * i], g1 = box256[2u * i + 1u];
            if (!(g0.x == g0.x)) f0.x = g0.x;                           // a NaN stays
            f0.x = g0.x < f0.x ? g0.x : f0.x; f0.y = g0.y > f0.y ? g0.y : f0.y; f0.z = g0.z < f0.z ? g0.z : f0.z; f0.w = g0.w > f0.w ? g0.w : f0.w;
            f1.x = g1.x < f1.x ? g1.x : f1.x; f1.y = g1.y > f1.y ? g1.y : f1.y; f1.z = g1.z < f1.z ? g1.z : f1.z; f1.w = g1.w > f1.w ? g1.w : f1.w;
        }
        box16[2u * RT_PARK_BOXES] = f0; box16[2u * RT_PARK_BOXES + 1u] = f1;
    }
}

// True when tri_test fails for EVERY ray with origin (ox, [b0.x, b0.y], [b0.z, b0.w]) and direction (+-0, [b1.x, b1.y], [b1.z, b1.w]):
// interval arithmetic over the very operations of tri_test, as column_box_rejects, now with the origin a box as well.  With
// direction.x == +-0: h.y = D.z*e2.x - e2.z*D.x = D.z*e2.x, h.z = D.x*e2.y - e2.x*D.y = -(e2.x*D.y) and D.x*q.x + D.y*q.y = D.y*q.y
// exactly (a product with the zero is a zero, and x - 0 = x, 0 - x = -x, 0 + x = x; where x itself is a zero only its sign can
// differ, which no comparison of tri_test sees).
__device__ __forceinline__ bool park_box_rejects(F3 v0, F3 e1, F3 e2, float ox, float4 b0, float4 b1)
{
    const Iv DY = {b1.x, b1.y}, DZ = {b1.z, b1.w};
    const Iv hx = iv_sub(iv_mul_scalar(DY, e2.z), iv_mul_scalar(DZ, e2.y));
    const Iv hy = iv_mul_scalar(DZ, e2.x);
    const Iv t0 = iv_mul_scalar(DY, e2.x);
    const Iv hz = {-t0.hi, -t0.lo};
    const Iv a = iv_add(iv_add(iv_mul_scalar(hx, e1.x), iv_mul_scalar(hy, e1.y)), iv_mul_scalar(hz, e1.z));
    if (a.lo > -0.00001f && a.hi < 0.00001f) return true;           // "a > -EPS && a < EPS" for every ray
    if (!(a.lo > 1e-30f || a.hi < -1e-30f)) return false;           // some ray may see a (near) zero determinant
    const Iv f = {1.0f / a.hi, 1.0f / a.lo};
    const float sx = ox - v0.x;
    const Iv sy = {b0.x - v0.y, b0.y - v0.y}, sz = {b0.z - v0.z, b0.w - v0.z};
    const Iv du = iv_add(iv_add(iv_mul_scalar(hx, sx), iv_mul(sy, hy)), iv_mul(sz, hz));
    const Iv u = iv_mul(f, du);
    if (u.hi < 0.0f || u.lo > 1.0f) return true;
    // q = cross(s, e1)
    const Iv qx = iv_sub(iv_mul_scalar(sy, e1.z), iv_mul_scalar(sz, e1.y));
    const float c1 = e1.z * sx, c2 = sx * e1.y;
    const Iv m1 = iv_mul_scalar(sz, e1.x), m2 = iv_mul_scalar(sy, e1.x);
    const Iv qy = {m1.lo - c1, m1.hi - c1};
    const Iv qz = {c2 - m2.hi, c2 - m2.lo};
    const Iv dv = iv_add(iv_mul(DY, qy), iv_mul(DZ, qz));
    const Iv v = iv_mul(f, dv);
    if (v.hi < 0.0f || u.lo + v.lo > 1.0f) return true;
    const Iv dt = iv_add(iv_add(iv_mul_scalar(qx, e2.x), iv_mul_scalar(qy, e2.y)), iv_mul_scalar(qz, e2.z));
    const Iv t = iv_mul(f, dt);
    return !(t.hi > 0.00001f);
}

// Every ray of a family (the column rays, or parked rays with one origin.x) lies in the plane x = origin.x (direction.x is a
// zero), and a hit needs u, v >= 0, u + v <= 1 with, in exact arithmetic, u e1.x + v e2.x = origin.x - v0.x: the plane has to
// cross the triangle.  The computed u and v miss the exact ones by at most (7 u_r (S_u |e1.x| + S_v |e2.x|) + 5 u_r |s.x| A) / |a|
// in that combination (u_r = 2^-24; A, S_u, S_v the sums of the absolute values of the terms of the determinant and the two
// numerators), |a| >= 1e-5 or the test rejects anyway, and A <= 2 d |e1|_1 |e2|_1, S_u, S_v <= 2 d |s|_inf |e|_1 with d = max
// |direction component|: so a triangle whose x extent stays clear of the plane by more than 2 * 38 u_r d |s|_inf |e1|_1 |e2|_1 / 1e-5
// (twice the first-order bound) fails the test for every ray of the family.  That is all but the few rows of triangles along
// the plane.  sx = origin.x - v0.x; s_inf, d: bounds of |origin - v0| and of the direction components over the family (d >= 1).
__device__ __forceinline__ bool clear_of_the_rays_plane(F3 e1, F3 e2, float sx, float s_inf, float d)
{
    const float n1 = (fabsf(e1.x) + fabsf(e1.y)) + fabsf(e1.z), n2 = (fabsf(e2.x) + fabsf(e2.y)) + fabsf(e2.z);
    const float clear = (76.0f * 5.9604645e-8f * 1e5f) * d * s_inf * n1 * n2 + 4.0f * 5.9604645e-8f * ((fabsf(e1.x) + fabsf(e2.x)) + fabsf(sx));
    const float xlo = fminf(0.0f, fminf(e1.x, e2.x)), xhi = fmaxf(0.0f, fmaxf(e1.x, e2.x));
    return sx < xlo - clear || sx > xhi + clear;
}

// One lane per leaf record of a leaf the family visits: boxes of 256, then 64, then 16 of the family's rays; the (record, 16 rays)
// pairs that cannot be proven misses go to the family's queue.
__global__ __launch_bounds__(256) void rt_park_filter_kernel(RtParams p, int stats)
{
    F3 O0, D0; uint32_t count;
    if (!reach_ray(p, O0, D0, count)) return;
    const uint32_t fam = (uint32_t)p.ray_slot;
    if (p.park_info[2 * RT_PARK_FAMILIES + fam] & 1) return;           // the visited set is not fixed: the test kernel flags the rays
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    bool use = false;
    if (r < p.n_refs) { const uint32_t leaf = p.rec_leaf[r]; use = ((p.reached[leaf >> 6] >> (leaf & 63u)) & 1ull) != 0ull; }
    if (stats) block_count_add(use, &p.column_stats[1], count);            // the tests the reference makes
    if (!use) return;
    const float4 r0 = p.leafrecs[3 * (size_t)r], r1 = p.leafrecs[3 * (size_t)r + 1], r2 = p.leafrecs[3 * (size_t)r + 2];
    const F3 v0 = f3(r0.x, r0.y, r0.z), e1 = f3(r0.w, r1.x, r1.y), e2 = f3(r1.z, r1.w, r2.x);
    const float big = 1e9f;
    const bool tame = fabsf(v0.x) < big && fabsf(v0.y) < big && fabsf(v0.z) < big && fabsf(e1.x) < big && fabsf(e1.y) < big && fabsf(e1.z) < big &&
                      fabsf(e2.x) < big && fabsf(e2.y) < big && fabsf(e2.z) < big && fabsf(O0.x) < big;
    const float4* box16 = p.park_boxes + (size_t)fam * 2u * (RT_PARK_BOXES + 1u);
    const float4* box64 = box16 + 2u * (RT_PARK_SLOTS / 16u);
    const float4* box256 = box64 + 2u * (RT_PARK_SLOTS / 64u);
    const uint32_t n16 = (count + 15u) / 16u, n64 = (count + 63u) / 64u, n256 = (count + 255u) / 256u;
    uint2* queue = p.park_queue + (size_t)fam * p.park_queue_cap;
    // the plane of the family's rays has to cross the triangle (see clear_of_the_rays_plane)
    {
        const float4 f0 = box16[2u * RT_PARK_BOXES], f1 = box16[2u * RT_PARK_BOXES + 1u];
        if (tame && f0.x == f0.x) {
            const float sx = O0.x - v0.x;
            const float sy = fmaxf(fabsf(f0.x - v0.y), fabsf(f0.y - v0.y)), sz = fmaxf(fabsf(f0.z - v0.z), fabsf(f0.w - v0.z));
            const float s_inf = fmaxf(fabsf(sx), fmaxf(sy, sz));
            const float d = fmaxf(1.0f, fmaxf(fmaxf(fabsf(f1.x), fabsf(f1.y)), fmaxf(fabsf(f1.z), fabsf(f1.w))));
            if (clear_of_the_rays_plane(e1, e2, sx, s_inf, d)) return;
        }
    }
    for (uint32_t c256 = 0; c256 < n256; ++c256) {
        const float4 a0 = box256[2u * c256], a1 = box256[2u * c256 + 1u];
        if (tame && a0.x == a0.x && park_box_rejects(v0, e1, e2, O0.x, a0, a1)) continue;
        const uint32_t e64 = (c256 * 4u + 4u < n64) ? c256 * 4u + 4u : n64;
        for (uint32_t c64 = c256 * 4u; c64 < e64; ++c64) {
            const float4 b0 = box64[2u * c64], b1 = box64[2u * c64 + 1u];
            if (tame && b0.x == b0.x && park_box_rejects(v0, e1, e2, O0.x, b0, b1)) continue;
            const uint32_t e16 = (c64 * 4u + 4u < n16) ? c64 * 4u + 4u : n16;
            for (uint32_t c16 = c64 * 4u; c16 < e16; ++c16) {
                const float4 c0 = box16[2u * c16], c1 = box16[2u * c16 + 1u];
                if (tame && c0.x == c0.x && park_box_rejects(v0, e1, e2, O0.x, c0, c1)) continue;
                const uint32_t at = atomicAdd(p.park_queue_count + fam, 1u);
                if (at < p.park_queue_cap) { queue[at] = make_uint2(r, c16); continue; }
                // queue full: this lane makes the pair's sixteen tests itself (slow, but nothing is lost)
                const uint32_t* list = p.park_list + fam * RT_PARK_SLOTS;
                const uint32_t last = (c16 * 16u + 16u < count) ? c16 * 16u + 16u : count;
                for (uint32_t k = c16 * 16u; k < last; ++k) {
                    const uint32_t sl = list[k];
                    const float4 o4 = p.resume_paths[RT_PARK_WORDS * sl], d4 = p.resume_paths[RT_PARK_WORDS * sl + 1];
                    float tt, u, v;
                    if (tri_test(r0, r1, r2, f3(o4.x, o4.y, o4.z), f3(d4.x, d4.y, d4.z), tt, u, v))
                        atomicMin(&p.park_best[sl], ((unsigned long long)__float_as_uint(tt) << 32) | (unsigned long long)r);
                }
            }
        }
    }
}

// The real triangle test for the queued (record, 16 rays) pairs of a family: 16 lanes per pair, min (t, record index) per ray.
// A family whose visited set depends on t gets no result: its rays are flagged and their waves walk.
__global__ __launch_bounds__(256) void rt_park_test_kernel(RtParams p)
{
    F3 O0, D0; uint32_t count;
    if (!reach_ray(p, O0, D0, count)) return;
    const uint32_t fam = (uint32_t)p.ray_slot;
    const uint32_t* list = p.park_list + fam * RT_PARK_SLOTS;
    if (p.park_info[2 * RT_PARK_FAMILIES + fam] & 1) {
        if (blockIdx.x == 0) for (uint32_t k = threadIdx.x; k < count; k += 256u) p.park_flags[list[k]] = 1u;
        return;
    }
    const uint32_t n = p.park_queue_count[fam] < p.park_queue_cap ? p.park_queue_count[fam] : p.park_queue_cap;
    const uint2* queue = p.park_queue + (size_t)fam * p.park_queue_cap;
    const uint32_t sub = threadIdx.x & 15u;
    for (uint32_t e = blockIdx.x * 16u + (threadIdx.x >> 4); e < n; e += gridDim.x * 16u) {
        const uint2 pr = queue[e];
        const uint32_t k = pr.y * 16u + sub;
        if (k >= count) continue;
        const uint32_t sl = list[k];
        const float4* rec = p.leafrecs + 3 * (size_t)pr.x;
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        const float4 o4 = p.resume_paths[RT_PARK_WORDS * sl], d4 = p.resume_paths[RT_PARK_WORDS * sl + 1];
        float tt, u, v;
        if (tri_test(r0, r1, r2, f3(o4.x, o4.y, o4.z), f3(d4.x, d4.y, d4.z), tt, u, v)) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(tt) << 32) | (unsigned long long)pr.x;
            atomicMin(&p.park_best[sl], key);
        }
    }
}

// One lane per leaf record of a visited leaf: boxes of 256, then 64, then 16 rays; what cannot be proven a miss is queued.
__global__ __launch_bounds__(256) void rt_column_filter_kernel(RtParams p, int stats)
{
    const uint32_t count = *p.column_count < p.column_cap ? *p.column_count : p.column_cap;
    if (count == 0u || (*p.column_flags & 1u)) return;
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    bool use = false;
    float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0, r2 = r0;
    if (r < p.n_refs) {
        const uint32_t leaf = p.rec_leaf[r];
        use = ((p.reached[leaf >> 6] >> (leaf & 63u)) & 1ull) != 0ull;
        if (use) { r0 = p.leafrecs[3 * (size_t)r]; r1 = p.leafrecs[3 * (size_t)r + 1]; r2 = p.leafrecs[3 * (size_t)r + 2]; }
    }
    if (stats) block_count_add(use, &p.column_stats[1], count);            // the tests the reference makes
    if (!use) return;
    const F3 O = f3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
    const ColumnTri c = column_tri(r0, r1, r2, O);
    // the proof needs finite, moderate numbers (no overflow, no 0 * infinity): otherwise every pair is tested exactly
    const float big = 1e9f;
    const bool tame = fabsf(c.s.x) < big && fabsf(c.s.y) < big && fabsf(c.s.z) < big && fabsf(c.e1.x) < big && fabsf(c.e1.y) < big &&
                      fabsf(c.e1.z) < big && fabsf(c.e2.x) < big && fabsf(c.e2.y) < big && fabsf(c.e2.z) < big;
    const uint32_t n16 = (count + 15u) / 16u, n64 = (count + 63u) / 64u, n256 = (count + 255u) / 256u;
    // All column rays start at the camera and have direction.x == +0: the plane x = camera.x has to cross the triangle (see
    // clear_of_the_rays_plane; the largest direction component comes from the boxes of 256 rays, a few wave-uniform reads).
    if (tame) {
        float d = 1.0f;
        bool usable = true;
        for (uint32_t c256 = 0; c256 < n256; ++c256) {
            const float4 b = p.column_box256[c256];
            usable = usable && b.x == b.x;
            d = fmaxf(d, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
        }
        if (usable && clear_of_the_rays_plane(c.e1, c.e2, c.s.x, fmaxf(fabsf(c.s.x), fmaxf(fabsf(c.s.y), fabsf(c.s.z))), d)) return;
    }
    for (uint32_t c256 = 0; c256 < n256; ++c256) {
        const float4 b256 = p.column_box256[c256];
        if (tame && b256.x == b256.x && column_box_rejects(c, b256)) continue;
        const uint32_t e64 = (c256 * 4u + 4u < n64) ? c256 * 4u + 4u : n64;
        for (uint32_t c64 = c256 * 4u; c64 < e64; ++c64) {
            const float4 b64 = p.column_box64[c64];
            if (tame && b64.x == b64.x && column_box_rejects(c, b64)) continue;
            const uint32_t e16 = (c64 * 4u + 4u < n16) ? c64 * 4u + 4u : n16;
            for (uint32_t c16 = c64 * 4u; c16 < e16; ++c16) {
                const float4 b16 = p.column_box16[c16];
                if (tame && b16.x == b16.x && column_box_rejects(c, b16)) continue;
                const uint32_t at = atomicAdd(p.column_queue_count, 1u);
                if (at < p.column_queue_cap) p.column_queue[at] = make_uint2(r, c16);
                else atomicOr(p.column_flags, 1u);                  // queue full: the trace kernel walks these rays itself
            }
        }
    }
}

// Fresh results for the column rays of this launch (the grid covers column_cap slots).
__global__ __launch_bounds__(256) void rt_column_reset_kernel(RtParams p)
{
    const uint32_t count = *p.column_count < p.column_cap ? *p.column_count : p.column_cap;
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    if (gid < count) p.column_best[gid] = ~0ull;
}

// The real triangle test for the queued (record, 16 rays) pairs: 16 lanes per pair.
__global__ __launch_bounds__(256) void rt_column_exact_kernel(RtParams p)
{
    const uint32_t count = *p.column_count < p.column_cap ? *p.column_count : p.column_cap;
    if (count == 0u || (*p.column_flags & 1u)) return;
    const uint32_t n = *p.column_queue_count < p.column_queue_cap ? *p.column_queue_count : p.column_queue_cap;
    const F3 O = f3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
    const uint32_t sub = threadIdx.x & 15u;
    for (uint32_t e = blockIdx.x * 16u + (threadIdx.x >> 4); e < n; e += gridDim.x * 16u) {
        const uint2 pr = p.column_queue[e];
        const uint32_t ray = pr.y * 16u + sub;
        if (ray >= count) continue;
        const float4* rec = p.leafrecs + 3 * (size_t)pr.x;
        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        const float4 d4 = p.column_dirs[ray];
        float tt, u, v;
        if (tri_test(r0, r1, r2, O, f3(d4.x, d4.y, d4.z), tt, u, v)) {
            const unsigned long long key = ((unsigned long long)__float_as_uint(tt) << 32) | (unsigned long long)pr.x;
            atomicMin(&p.column_best[ray], key);
        }
    }
}

// The trace kernels are persistent: the grid is sized to what the chip holds at once and every wave works through 16-pixel
// units (the four 8x2 strips of each 8x8 tile).  Each loop iteration advances every live lane by one bounce; lanes whose path
// ended take the next (pixel, frame) before the next iteration, so the wave stays full until the launch runs dry.  In sphere
// scenes a lane keeps its pixel for all n_frames frames of the launch and adds them in frame order; in tree scenes every
// (pixel, frame) is a path of its own, drawn independently, whose radiance goes to the pixel's slot of that frame, and the
// resolve pass adds the slots in frame order.  Either way the sum is that of n_frames separate Render calls, bit for bit.
// How units reach the waves depends on the scene:
//   spheres only : every region of the image costs about the same, so each workgroup owns a fixed share of the units (workgroup b
//                  takes units b, b + workgroups, ...) and its 16 waves draw them from a counter in LDS.  The waves of a SIMD do
//                  not advance at the same rate (the oldest wave is served first), so a fixed share per WAVE leaves each SIMD
//                  half empty for the last third of the frame; the LDS counter costs ~100 cycles per unit and lets all waves of a
//                  workgroup finish together.  A global ticket per unit would cost more than it balances here: vector memory
//                  operations complete in order, so every later load of the wave also waits for the atomic's return.
//   tree scenes  : a unit's cost varies by orders of magnitude (a primary ray that misses the mesh's box against eight bounces
//                  through the tree), so units are pulled from RT_WORK_HEADS ticket counters.  Each head owns every 8th unit;
//                  a wave draws from the head of its workgroup number modulo 8 (workgroups are dealt round-robin over the
//                  XCDs: one head per XCD) and moves on to the other heads when its own is exhausted.  Units are taken whole,
//                  one per 16 idle lanes and all of a refill with one atomic, so a wave never owns pixels it cannot start at
//                  once -- otherwise the last units trickle through a few lanes of one wave while the rest of the chip idles.
// A launch of n_frames Renders hands out (unit, frame) pairs: head h owns units h, h + 8, ... and its ticket t is frame t % n_frames
// of unit h + 8 (t / n_frames), so the tickets of one refill are consecutive frames of the same pixels: their primary rays are
// the same ray and walk the same nodes side by side.
// n_work: the units there are to draw (the length of unit_list, read from the device once per wave, or n_units)
__device__ __forceinline__ uint32_t head_count(const RtParams& p, uint32_t h, uint32_t n_work)
{
    return ((n_work + RT_WORK_HEADS - 1u - h) / RT_WORK_HEADS) * p.n_frames;
}
__device__ __forceinline__ uint32_t head_unit(const RtParams& p, uint32_t h, uint32_t t)
{
    const uint32_t i = h + RT_WORK_HEADS * (t / p.n_frames);
    return p.unit_list ? p.unit_list[i] : i;                // (only the units in which a primary ray can hit anything, see RtParams::unit_list)
}

template <bool STATS, bool LDS_SCENE>
__device__ __forceinline__ void sphere_body(const RtParams& p)
{
    __shared__ SceneLds lds;                                // only allocated where LDS_SCENE variants use it
    __shared__ uint32_t wg_next;                            // how many of this workgroup's units have been handed out
    __shared__ SpherePool pool;
    constexpr bool MERGE = RT_SPHERE_MERGE != 0 && !STATS;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (threadIdx.x == 0) { wg_next = 0u; pool.lock = 0u; pool.count = 0u; pool.alive = blockDim.x >> 6; }
    if (LDS_SCENE) stage_scene(p, lds);
    __syncthreads();
    // wave-uniform: current unit of the workgroup's share and its first free pixel (RT_UNIT = used up)
    uint32_t unit = 0, slot = RT_UNIT;
    bool share_done = false;
    uint32_t n_rays = 0;
    unsigned long long clock_start = 0; uint32_t n_iterations = 0;
    const bool clocked = p.wave_clock != nullptr;           // diagnostics: start / end time and trip count of every wave
    if (clocked) clock_start = __builtin_amdgcn_s_memrealtime();

    bool live = false;
    uint32_t pixel = 0, frame = 0;
    F3 acc = f3(0.0f, 0.0f, 0.0f);
    Path s;
    s.O = s.D = s.contribution = s.finalColor = f3(0.0f, 0.0f, 0.0f); s.rng = 0; s.bounce = 0; s.column_slot = 0;

    // this lane's path to / from slot `slot` of the pool
    auto path_transfer = [&](const bool save, const uint32_t slot) {
        uint32_t i = 0;
        auto U = [&](uint32_t& v) { if (save) pool.words[i][slot] = v; else v = pool.words[i][slot]; ++i; };
        auto F = [&](float& v) { if (save) pool.words[i][slot] = __float_as_uint(v); else v = __uint_as_float(pool.words[i][slot]); ++i; };
        uint32_t rng_lo = (uint32_t)s.rng, rng_hi = (uint32_t)(s.rng >> 32);
        U(pixel); U(frame); F(acc.x); F(acc.y); F(acc.z);
        F(s.O.x); F(s.O.y); F(s.O.z); F(s.D.x); F(s.D.y); F(s.D.z);
        F(s.contribution.x); F(s.contribution.y); F(s.contribution.z); F(s.finalColor.x); F(s.finalColor.y); F(s.finalColor.z);
        U(rng_lo); U(rng_hi); U(s.bounce);
        static_assert(RT_SPHERE_POOL_WORDS == 20, "path_transfer moves 20 words");
        if (!save) s.rng = ((unsigned long long)rng_hi << 32) | (unsigned long long)rng_lo;
    };

    for (;;) {
        // ---- give idle lanes pixels while there are units left
        unsigned long long need = __ballot(!live);
        while (need != 0ull && !share_done) {
            if (slot == RT_UNIT) {                          // next unit of the workgroup's share
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(&wg_next, 1u);
                k = __builtin_amdgcn_readfirstlane(k);
                unit = blockIdx.x + gridDim.x * k;
                if (unit >= p.n_units) { share_done = true; break; }
                slot = 0;
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            const uint32_t avail = RT_UNIT - slot;
            if (!live && rank < avail && unit_pixel(p, unit, slot + rank, pixel)) {
                live = true;
                frame = 0;
                acc = f3(0.0f, 0.0f, 0.0f);                 // frame index 1 restarts accumulation (Renderer.cpp:55-56)
                if (p.frame_first != 1u) { const float4 a = p.accum[pixel]; acc = f3(a.x, a.y, a.z); }
                path_begin(s, p, pixel, p.frame_first);
            }
            const uint32_t wanted = (uint32_t)__popcll(need);
            slot += wanted < avail ? wanted : avail;
            need = __ballot(!live);                         // lanes beyond this unit, and lanes whose pixel fell outside the frame
        }
        // ---- the share is handed out: waves with few paths left merge them (see SpherePool)
        if (MERGE && share_done) {
            const unsigned long long live_mask = __ballot(live);
            const uint32_t n_live = (uint32_t)__popcll(live_mask);
            const bool can_take = n_live < 64u && lds_peek(&pool.count) != 0u;
            const bool may_leave = n_live != 0u && n_live <= (uint32_t)RT_SPHERE_MERGE && lds_peek(&pool.alive) > 1u;
            if (can_take || may_leave || n_live == 0u) {
                bool leave = false;
                lds_lock(&pool.lock, lane);
                const uint32_t have = pool.count;
                if (have != 0u) {
                    // this wave's paths join the waiting ones, and the wave takes up to 64 of them back: every lane is loaded
                    // anew (lanes beyond `take` read slot 0 and stay idle), so no lane keeps old and new state side by side
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(live_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live_mask, 0u));
                    if (live) path_transfer(true, have + rank);
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    const uint32_t total = have + n_live;
                    const uint32_t take = total < 64u ? total : 64u;
                    path_transfer(false, lane < take ? total - take + lane : 0u);
                    live = lane < take;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    if (lane == 0) pool.count = total - take;
                } else if (n_live == 0u) {
                    if (lane == 0) pool.alive -= 1u;        // nothing waiting, nothing held: this wave is done
                    leave = true;
                } else if (may_leave && pool.alive > 1u) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(live_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)live_mask, 0u));
                    if (live) path_transfer(true, rank);
                    live = false;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    if (lane == 0) { pool.count = n_live; pool.alive -= 1u; }
                    leave = true;                           // another wave of the workgroup is still in the loop and will take them
                }
                lds_unlock(&pool.lock, lane);
                if (leave) break;
            }
        }
        if (__ballot(live) == 0ull) {
            if (!MERGE) break;                              // nothing in flight and nothing left to take
            continue;                                       // (merging: the block above decides when an empty wave may end)
        }
        n_iterations++;
        // The two workgroups of a CU own equal shares, but instruction issue favours the older wave, so one of them would
        // finish at half time and leave the other with half the waves per SIMD.  Issue priority goes to the younger half of
        // the grid in RT_SPHERE_PRIO of every 8 iterations and to the older half in the rest (measured: with 4 of 8 the older
        // workgroups still finish 50 us early, with 8 of 8 the younger ones do; 6 of 8 brings them within 15 us;
        // RT_SPHERE_PRIO_PERIOD, a power of two, is the 8).
        {
            const bool younger = blockIdx.x >= (gridDim.x >> 1);
            const bool favoured = ((n_iterations & (uint32_t)(RT_SPHERE_PRIO_PERIOD - 1)) < (uint32_t)RT_SPHERE_PRIO) == younger;
            if (favoured) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }

        // ---- one bounce for every lane with a path
        if (live) {
            int32_t* id_out = (p.primary_ids && frame == 0) ? p.primary_ids + pixel : nullptr;
            const bool ended = path_step<LDS_SCENE>(s, p, lds, id_out, n_rays);
            if (ended) {
                if (id_out && p.bounces == 0) *id_out = -1;
                acc = acc + s.finalColor;                   // m_AccumulationBuffer[pixelIndex] += pixelColor
                frame++;
                if (frame < p.n_frames) path_begin(s, p, pixel, p.frame_first + frame);
                else { p.accum[pixel] = make_float4(acc.x, acc.y, acc.z, 0.0f); live = false; }
            }
        }
    }

    // Frames without the 3x3 filter: the workgroup resolves its own share of the image as soon as its last wave is
    // done (Renderer.cpp:68-79).  The sums were just written by this CU (L2-resident), every lane is busy, and the work fills
    // the time in which other workgroups still trace -- instead of a separate pass over the whole frame after the launch.
    if (p.fuse_resolve) {
        __syncthreads();
        const uint32_t share = p.n_units > blockIdx.x ? (p.n_units - blockIdx.x + gridDim.x - 1u) / gridDim.x : 0u;
        const float div = (float)(p.frame_first + p.n_frames - 1u);
        for (uint32_t i = threadIdx.x; i < share * RT_UNIT; i += blockDim.x) {
            uint32_t px;
            if (unit_pixel(p, blockIdx.x + gridDim.x * (i / RT_UNIT), i % RT_UNIT, px)) {
                const float4 a = p.accum[px];
                p.image[px] = resolve_px(f3(a.x, a.y, a.z), div, p.use_ace);
            }
        }
    }

    if (clocked && lane == 0) {
        unsigned long long* wc = p.wave_clock + RT_WAVE_CLOCK_WORDS * (size_t)wave;
        wc[0] = clock_start; wc[1] = __builtin_amdgcn_s_memrealtime(); wc[2] = n_iterations;
        wc[3] = 0; wc[4] = 0; wc[5] = 0; wc[6] = 0; wc[7] = 0;
    }
    // work counters: one slot group per wave, summed on the host when read
    if (p.counters) {
        unsigned long long* slot = p.counters + 4 * (size_t)wave;
        uint32_t r = n_rays;
        for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
        if (lane == 0) {
            slot[0] += (unsigned long long)r;
            slot[3] += (unsigned long long)r * p.n_spheres;
        }
    }
}

// ---- tree scenes: traversal decoupled from the bounce loop ---------------------------------------------------------------
// The lanes of a wave walk the tree for very different numbers of steps (most primary rays miss the root box or stop after a
// few nodes; a few paths take hundreds of steps per bounce).  If every bounce ran to completion before the wave moved on, each
// wave would spend most of its trips with a handful of lanes busy.  Here a lane's walk is a resumable state (node cursor, leaf-
// record cursor, closest hit so far) that lives across the iterations of the wave's main loop, and each iteration runs the
// walk only until half of the lanes that were walking have finished.  Those are shaded, get their next ray or their next
// pixel, and join the unfinished walks in the next slice.  Every lane makes exactly the steps of KDTreeCPU::intersect for its
// ray, in the same order, so the results do not change -- only which lanes keep each other company.
// ORDERED = false: the walk of KDTreeCPU::intersect itself (node array 0 is the reference's preorder).
// ORDERED = true : node array number (direction.x < 0) + 2 (direction.y < 0) + 4 (direction.z < 0) holds the same tree in
// the preorder that puts, at every node, the child on the near side of this ray first; boxes, triangle tests and the
// pruning rule are unchanged, so once a hit is known almost everything behind it is pruned.  Equal t is resolved to the
// smaller leaf-record index, i.e. to the triangle the reference order reaches first.
constexpr uint32_t PH_IDLE = 0u, PH_BEGIN = 1u, PH_WALK = 2u, PH_SHADE = 3u, PH_REDO = 4u;

// Paths changing waves at the end of a frame.  Every unit is drawn within the first tenth of a tree frame; from then on a
// wave just finishes the paths it holds and lasts as long as the longest of them, one lane walking while 63 wait, and more
// and more waves have nothing left at all.  A wave without work therefore takes paths over from the waves of its workgroup
// that still hold several: the whole per-lane state goes through this pool in LDS.  With at most RT_COOP_LANES paths the
// adopting wave walks the tree with all 64 lanes for one ray at a time (walk_cooperative), several times faster than the
// one-lane walk, so the long paths that decide when the frame ends are shortened.  A path is the same sequence of
// operations whichever lane runs it: results do not change.
// lock: one wave at a time changes the pool; count: paths waiting in it; hungry: waves waiting for paths;
// busy: waves that hold paths or may still draw units.  A wave ends when the pool is empty and no wave is busy.
struct PathPool {
    uint32_t lock, count, hungry, busy;
    uint32_t words[RT_POOL_WORDS][RT_POOL_SLOTS];
};

template <int MODE, bool STATS, bool ORDERED>
__device__ __forceinline__ void tree_body(const RtParams& p)
{
    __shared__ SceneLds lds;                                // tree scenes read the closest-hit tables from HBM; only the sphere loop of
    __shared__ PathPool pool;                               // the mixed mode reads its table from here (trace_spheres_lds)
    // the counting kernels keep every path in the lane that started it (their per-wave clocks and trip counts describe the plain walk)
    constexpr bool SHARE = RT_SHARE != 0 && !STATS;
    // The timed kernels make their box tests in the short form (slab_test_t<true>), which is exact for rays that cannot meet a
    // NaN in them (slab_plain_ray).  The few other rays (a zero or denormal direction component in y or z, an origin beyond
    // 2^60, a tree with boxes beyond 2^60) never enter that walk: the wave makes the reference's own walk for them, with the
    // compare+select forms, as it does for the near ties of the front-to-back order (REDO: the kernel holds that route).
    constexpr bool PLAIN = RT_PLAIN_SLAB != 0 && !STATS;
    constexpr bool REDO = PLAIN || (ORDERED && RT_NEAR_TIES != 0);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const bool spheres_in_lds = RT_SPHERE_LDS_LOOP && MODE == RT_MODE_BOTH && p.n_spheres <= RT_LDS_SPHERES;
    if (spheres_in_lds)
        for (uint32_t i = threadIdx.x; i < p.n_spheres; i += blockDim.x) lds.spheres[i] = p.spheres[i];
    if (SHARE) {
        if (threadIdx.x == 0) { pool.lock = 0u; pool.count = 0u; pool.hungry = 0u; pool.busy = blockDim.x >> 6; }
    }
    if (SHARE || MODE == RT_MODE_BOTH) __syncthreads();
    const bool sharing = SHARE && p.resume == 0u;           // a resume launch only finishes the parked paths
    bool waiting = false;                                   // wave-uniform: this wave is counted in pool.hungry, not in pool.busy
    uint32_t head = blockIdx.x % RT_WORK_HEADS, heads_empty = 0;        // wave-uniform: the ticket head drawn from, heads found empty
    uint32_t n_rays = 0, n_nodes = 0, n_tris = 0, n_reused = 0;
    unsigned long long clock_start = 0, trips = 0, lane_trips = 0, walk_ticks = 0, wait_ticks = 0, shade_ticks = 0; uint32_t n_iterations = 0;
    const bool clocked = p.wave_clock != nullptr && !p.resume;
    if (clocked) clock_start = __builtin_amdgcn_s_memrealtime();

    uint32_t phase = PH_IDLE;
    uint32_t pixel = 0, frame = 0, path_steps = 0;
    Path s;
    s.O = s.D = s.contribution = s.finalColor = f3(0.0f, 0.0f, 0.0f); s.rng = 0; s.bounce = 0; s.column_slot = 0;
    const size_t plane = (size_t)p.width * p.height;        // pixels per slot plane
    // the bounce in flight: sphere hit, closest triangle so far, cursors of the walk
    float tS = FLT_MAX; int iS = -1;
    Hit th; th.t = FLT_MAX; th.id = -1; th.u = 0.0f; th.v = 0.0f;
    uint32_t wnode = 0, wrec = RT_NO_REC, wbest = RT_NO_REC, wsteps = 0;
    // front-to-back walks only: distance within which two different triangles count as a near tie, the pruning bound
    // (closest t plus that distance), whether a near tie was seen
    float wtol = 0.0f, wprune = FLT_MAX;
    bool wflag = false;
    F3 winvD = f3(0.0f, 0.0f, 0.0f);
    uint32_t woct = 0;                                      // which of the eight node arrays the walk reads (0 = the reference's preorder)
    const uint32_t end = p.n_nodes;
    const uint32_t n_work = p.unit_list ? (uint32_t)__builtin_amdgcn_readfirstlane((int)p.unit_counts[0]) : p.n_units;

    // the whole state of this lane's path to / from slot `slot` of the pool
    auto path_transfer = [&](const bool save, const uint32_t slot) {
        uint32_t i = 0;
        auto U = [&](uint32_t& v) { if (save) pool.words[i][slot] = v; else v = pool.words[i][slot]; ++i; };
        auto F = [&](float& v) { if (save) pool.words[i][slot] = __float_as_uint(v); else v = __uint_as_float(pool.words[i][slot]); ++i; };
        uint32_t rng_lo = (uint32_t)s.rng, rng_hi = (uint32_t)(s.rng >> 32), sphere = (uint32_t)iS, tri = (uint32_t)th.id, flag = wflag ? 1u : 0u;
        U(phase); U(pixel); U(frame); U(path_steps);
        F(s.O.x); F(s.O.y); F(s.O.z); F(s.D.x); F(s.D.y); F(s.D.z);
        F(s.contribution.x); F(s.contribution.y); F(s.contribution.z); F(s.finalColor.x); F(s.finalColor.y); F(s.finalColor.z);
        U(rng_lo); U(rng_hi); U(s.bounce); U(s.column_slot);
        F(tS); U(sphere); F(th.t); U(tri); F(th.u); F(th.v);
        U(wnode); U(wrec); U(wbest); U(wsteps); F(wtol); F(wprune); U(flag);
        F(winvD.x); F(winvD.y); F(winvD.z); U(woct);
        static_assert(RT_POOL_WORDS == 38, "path_transfer moves 38 words");
        if (!save) {
            s.rng = ((unsigned long long)rng_hi << 32) | (unsigned long long)rng_lo;
            iS = (int)sphere; th.id = (int32_t)tri; wflag = flag != 0u;
        }
    };

    // Spheres and mesh together, front-to-back walk: the sphere loop has run when the walk starts, and a triangle farther than the
    // sphere hit loses to it (Renderer.cpp:292-300: the sphere wins on tS < tT).  The walk therefore never needs to look beyond
    // tS plus the near-tie distance: a triangle that wins or ties (t <= tS) lies in a leaf whose chain of boxes is entered no
    // later than t plus half that distance (see RtParams::tie_*), so it is found exactly as without the cap, and anything the
    // cap hides is beaten by the sphere whatever it is.  No sphere hit: tS = FLT_MAX, no cap.
    auto sphere_cap = [&]() -> float {
        if (MODE != RT_MODE_BOTH || !ORDERED) return FLT_MAX;
        return RT_NEAR_TIES ? tS + (wtol + p.tie_rel * tS) : tS;
    };

    bool parked_hit_ready = false;                          // this lane continues a parked path whose hit the whole-chip kernels found
    uint32_t resume_slot = 0;
    if (p.resume) {
        heads_empty = RT_WORK_HEADS;                        // no units: only the parked paths are finished
        // (few paths per wave, RtParams::resume_lanes: what is left of them is a chain of walks each, and a wave that holds no more
        // than RT_COOP_LANES walks lends all its lanes to one walk at a time)
        const uint32_t n_parked = *p.resume_count < RT_PARK_SLOTS ? *p.resume_count : RT_PARK_SLOTS;
        if (wave * p.resume_lanes >= n_parked) return;
        resume_slot = wave * p.resume_lanes + lane;
        if (lane < p.resume_lanes && resume_slot < n_parked) {
            const float4* q = p.resume_paths + RT_PARK_WORDS * resume_slot;
            const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4];
            static_assert(RT_PARK_WORDS == 5, "a parked path is five float4");
            s.O = f3(q0.x, q0.y, q0.z); s.column_slot = __float_as_uint(q0.w);
            s.D = f3(q1.x, q1.y, q1.z); s.bounce = __float_as_uint(q1.w);
            s.contribution = f3(q2.x, q2.y, q2.z); pixel = __float_as_uint(q2.w);
            s.finalColor = f3(q3.x, q3.y, q3.z); frame = __float_as_uint(q3.w);
            s.rng = ((unsigned long long)__float_as_uint(q4.y) << 32) | (unsigned long long)__float_as_uint(q4.x);
            phase = PH_BEGIN; parked_hit_ready = true;
        }
    }

    for (;;) {
        // ---- give idle lanes pixels while there are units left: whole units, one per 16 idle lanes, one atomic per refill
        unsigned long long need = __ballot(phase == PH_IDLE);
        uint32_t want = (uint32_t)__popcll(need) / RT_UNIT;
        while (want != 0u && heads_empty < RT_WORK_HEADS) {
            const uint32_t count = head_count(p, head, n_work);
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(p.work_heads + head * RT_HEAD_STRIDE, want);
            t = __builtin_amdgcn_readfirstlane(t);
            const uint32_t got = t < count ? (want < count - t ? want : count - t) : 0u;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(need >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)need, 0u));
            // idle lane number `rank` takes pixel rank % 16 of the rank / 16-th ticket taken (see head_unit)
            if (phase == PH_IDLE && rank < got * RT_UNIT && unit_pixel(p, head_unit(p, head, t + rank / RT_UNIT), rank % RT_UNIT, pixel)) {
                phase = PH_BEGIN;
                frame = (t + rank / RT_UNIT) % p.n_frames;
                path_begin(s, p, pixel, p.frame_first + frame);
            }
            if (got < want) { heads_empty++; head = (head + 1u) % RT_WORK_HEADS; }
            need = __ballot(phase == PH_IDLE);              // what is still idle: lanes without a unit, and pixels outside the frame
            want = (uint32_t)__popcll(need) / RT_UNIT;
        }
        // ---- no units left: paths move from the waves that hold several to the waves that hold none (see PathPool)
        if (sharing && heads_empty >= RT_WORK_HEADS) {
            const uint32_t n_live = (uint32_t)__popcll(__ballot(phase != PH_IDLE));
            if (n_live == 0u) {
                if (!waiting) {
                    lds_lock(&pool.lock, lane);
                    if (lane == 0) { pool.busy -= 1u; pool.hungry += 1u; }
                    lds_unlock(&pool.lock, lane);
                    waiting = true;
                }
                bool leave = false;
                const unsigned long long wait_start = clocked ? __builtin_amdgcn_s_memrealtime() : 0ull;
                for (;;) {
                    if (lds_peek(&pool.count) != 0u) {
                        lds_lock(&pool.lock, lane);
                        const uint32_t have = pool.count;
                        const uint32_t take = have < (uint32_t)RT_ADOPT ? have : (uint32_t)RT_ADOPT;
                        if (lane < take) path_transfer(false, have - take + lane);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                        if (take != 0u && lane == 0) { pool.count = have - take; pool.hungry -= 1u; pool.busy += 1u; }
                        lds_unlock(&pool.lock, lane);
                        if (take != 0u) { waiting = false; break; }
                    } else if (lds_peek(&pool.busy) == 0u) {
                        lds_lock(&pool.lock, lane);
                        leave = pool.count == 0u && pool.busy == 0u;
                        lds_unlock(&pool.lock, lane);
                        if (leave) break;
                    } else __builtin_amdgcn_s_sleep(8);
                }
                if (clocked) wait_ticks += __builtin_amdgcn_s_memrealtime() - wait_start;
                if (leave) break;
            } else if (n_live > (uint32_t)RT_KEEP) {
                const uint32_t wanted = lds_peek(&pool.hungry) * (uint32_t)RT_ADOPT, have = lds_peek(&pool.count);
                if (wanted > have) {
                    // the longest walks first: they decide when the frame ends, and the adopting wave walks several times faster
                    uint32_t give = wanted - have;
                    if (give > n_live - (uint32_t)RT_KEEP) give = n_live - (uint32_t)RT_KEEP;
                    const uint32_t rank_class = phase == PH_IDLE ? 4u : (phase == PH_REDO ? 0u : (phase != PH_WALK ? 3u : (wsteps >= 64u ? 0u : (wsteps >= 16u ? 1u : 2u))));
                    uint32_t before = 0, my_rank = 0;
                    for (uint32_t k = 0; k < 4u; ++k) {
                        const unsigned long long m = __ballot(rank_class == k);
                        if (rank_class == k) my_rank = before + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                        before += (uint32_t)__popcll(m);
                    }
                    lds_lock(&pool.lock, lane);
                    const uint32_t at = pool.count;
                    if (give > RT_POOL_SLOTS - at) give = RT_POOL_SLOTS - at;
                    if (phase != PH_IDLE && my_rank < give) { path_transfer(true, at + my_rank); phase = PH_IDLE; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    if (lane == 0) pool.count = at + give;
                    lds_unlock(&pool.lock, lane);
                }
            }
        }
        if (__ballot(phase != PH_IDLE) == 0ull) break;      // nothing in flight and nothing left to take
        n_iterations++;

        // ---- lanes whose path needs its next ray traced (Renderer.cpp:149-159)
        {
            const bool begin = phase == PH_BEGIN;
            bool start = begin && s.bounce < p.bounces;     // else "for (i = 0; i < Bounces; i++)" with Bounces == 0: ends below
            bool have_hit = false;
            if (start) { th.t = FLT_MAX; th.id = -1; th.u = 0.0f; th.v = 0.0f; }
            // the primary ray's hit is known from an earlier launch (RtParams::first_hits)
            if (start && s.bounce == 0u && p.first_hit_mode == 2u) {
                const float4 h = p.first_hits[pixel];
                th.t = h.x; th.u = h.y; th.v = h.z; th.id = (int32_t)__float_as_uint(h.w);
                have_hit = true;
                n_reused++;
            }
            // rays the tree cannot prune (direction.x == 0): answered by the pre-pass, by the whole-chip kernels, or by this wave
            bool unprunable = start && !have_hit && s.D.x == 0.0f;
            if (unprunable && s.bounce == 0 && s.column_slot != 0u && p.column_best && (*p.column_flags & 1u) == 0u) {
                // primary column rays were resolved for the whole frame by the pre-pass: re-run the winning triangle test for u, v
                const unsigned long long key = p.column_best[checked(p, s.column_slot - 1u, p.column_cap, 6)];
                if (key != ~0ull) {
                    const float4* rec = p.leafrecs + 3 * (size_t)checked(p, (uint32_t)key, p.n_refs, 7);
                    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                    float tt, u, v;
                    tri_test(r0, r1, r2, s.O, s.D, tt, u, v);
                    th.t = tt; th.u = u; th.v = v; th.id = (int32_t)(__float_as_uint(r2.y) & 0x7FFFFFFFu);
                }
                have_hit = true;
                unprunable = false;
            }
            // a parked path comes back with its hit (unless a decision of its walk depended on t: then it is walked here)
            if (parked_hit_ready) {
                parked_hit_ready = false;
                if (unprunable && (p.park_flags[checked(p, resume_slot, RT_PARK_SLOTS, 8)] & 1u) == 0u) {
                    const unsigned long long key = p.park_best[checked(p, resume_slot, RT_PARK_SLOTS, 8)];
                    if (key != ~0ull) {
                        const float4* rec = p.leafrecs + 3 * (size_t)checked(p, (uint32_t)key, p.n_refs, 9);
                        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                        float tt, u, v;
                        tri_test(r0, r1, r2, s.O, s.D, tt, u, v);
                        th.t = tt; th.u = u; th.v = v; th.id = (int32_t)(__float_as_uint(r2.y) & 0x7FFFFFFFu);
                    }
                    have_hit = true;
                    unprunable = false;
                }
            }
            // in a large tree such a walk takes one wave a large part of a second while the chip idles: park the path (the
            // lane takes other pixels), let whole-chip kernels find its hit after this launch, finish it in a resume launch
            if (unprunable && p.resume != 2u && p.n_nodes >= p.park_min_nodes) {
                const uint32_t slot = atomicAdd(p.park_count, 1u);
                if (slot < RT_PARK_SLOTS) {
                    float4* q = p.park_paths + RT_PARK_WORDS * slot;
                    q[0] = make_float4(s.O.x, s.O.y, s.O.z, __uint_as_float(s.column_slot));
                    q[1] = make_float4(s.D.x, s.D.y, s.D.z, __uint_as_float(s.bounce));
                    q[2] = make_float4(s.contribution.x, s.contribution.y, s.contribution.z, __uint_as_float(pixel));
                    q[3] = make_float4(s.finalColor.x, s.finalColor.y, s.finalColor.z, __uint_as_float(frame));
                    q[4] = make_float4(__uint_as_float((uint32_t)s.rng), __uint_as_float((uint32_t)(s.rng >> 32)), 0.0f, 0.0f);
                    phase = PH_IDLE; start = false; unprunable = false;
                }
            }
            unsigned long long todo = __ballot(unprunable);
            while (todo != 0ull) {                          // the whole wave walks the tree for them, one at a time
                const int src = __ffsll((long long)todo) - 1;
                todo &= todo - 1ull;
                const F3 O = f3(__shfl(s.O.x, src), __shfl(s.O.y, src), __shfl(s.O.z, src));
                const F3 D = f3(__shfl(s.D.x, src), __shfl(s.D.y, src), __shfl(s.D.z, src));
                const F3 invD = f3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
                Hit h;
                uint32_t cn = 0, ct = 0;
                const bool ok = trace_mesh_cooperative<STATS>(p, O, D, invD, lane, h, cn, ct);
                if (ok) {
                    if ((int)lane == src) { have_hit = true; th = h; }
                    n_nodes += cn; n_tris += ct;            // counted by the lanes that did the work
                }
            }
            if (start) {
                s.bounce++;
                if (p.first_hit_mode != 1u) n_rays++;       // (the launch that only stores primary hits is not a Render: its rays are counted by the launches that use them)
                tS = FLT_MAX; iS = -1;
                if (MODE == RT_MODE_BOTH && p.first_hit_mode != 1u) {
                    if (spheres_in_lds) trace_spheres_lds(lds.spheres, p.n_spheres, s.O, s.D, tS, iS);
                    else trace_spheres(p, s.O, s.D, tS, iS);
                }
                if (have_hit || end == 0u) phase = PH_SHADE;
                else {
                    winvD = f3(1.0f / s.D.x, 1.0f / s.D.y, 1.0f / s.D.z);          // Renderer.cpp:151
                    woct = 0;
                    if (ORDERED) woct = (s.D.x < 0.0f ? 1u : 0u) | (s.D.y < 0.0f ? 2u : 0u) | (s.D.z < 0.0f ? 4u : 0u);
                    wnode = 0; wrec = RT_NO_REC; wbest = RT_NO_REC; wsteps = 0;
                    // (see RtParams::unit_entry; a ray with direction.x == 0 had no say in its unit's entry node -- it gets here only
                    // when the whole-wave walk above found a decision of its walk to depend on t -- and starts at the root)
                    if (MODE == RT_MODE_MESH && !STATS && s.bounce == 1u && p.unit_entry && s.D.x != 0.0f) wnode = checked(p, p.unit_entry[pixel], end + 1u, 10);
                    wprune = FLT_MAX; wflag = false;
                    if (ORDERED && RT_NEAR_TIES) {
                        // The part of the near-tie distance that does not grow with t (see RtParams::tie_*): the rounding of a box's
                        // entry distance, per axis a few units in the last place of (|plane - origin| + |plane|) * |1 / direction|,
                        // and the constant part of the triangle test's error bound.  The bounds assume |direction| <= 1 per
                        // component; a longer (host-supplied) direction gets an infinite distance: never pruned by a margin,
                        // any two different hits repeated in the reference's order.
                        const float bx = (fabsf(s.O.x - p.root_center[0]) + p.tie_box[0]) * fabsf(winvD.x);
                        const float by = (fabsf(s.O.y - p.root_center[1]) + p.tie_box[1]) * fabsf(winvD.y);
                        const float bz = (fabsf(s.O.z - p.root_center[2]) + p.tie_box[2]) * fabsf(winvD.z);
                        wtol = fmaxf(bx, fmaxf(by, bz)) * p.tie_ulp + p.tie_abs;
                        if (fmaxf(fabsf(s.D.x), fmaxf(fabsf(s.D.y), fabsf(s.D.z))) > 1.001f) wtol = __builtin_inff();
                    }
                    if (MODE == RT_MODE_BOTH && ORDERED) wprune = sphere_cap();
                    phase = PH_WALK;
                    // A unit whose primary rays pass the root box and then miss everything alike has the END of the node array as
                    // its entry: the walk is over before it starts (a trip would read whatever lies behind the array as a node).
                    if (wnode >= end) phase = PH_SHADE;
                    else if (PLAIN && !(p.slab_plain != 0u && slab_plain_ray(s.O, winvD))) phase = PH_REDO;     // (see PLAIN)
                    if (STATS && p.ray_dump) {              // experiments: the rays that walk, in the order the waves start them
                        const unsigned long long m = __ballot(phase == PH_WALK);
                        if (phase == PH_WALK) {
                            const int leader = __ffsll((long long)m) - 1;
                            uint32_t base = 0;
                            if ((int)lane == leader) base = atomicAdd(p.ray_dump_count, (uint32_t)__popcll(m));
                            base = (uint32_t)__shfl((int)base, leader);
                            const uint32_t at = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                            if (at < p.ray_dump_cap) {
                                p.ray_dump[2 * (size_t)at] = make_float4(s.O.x, s.O.y, s.O.z, __uint_as_float(s.bounce));
                                p.ray_dump[2 * (size_t)at + 1] = make_float4(s.D.x, s.D.y, s.D.z, __uint_as_float(pixel));
                            }
                        }
                    }
                }
            } else if (begin && phase == PH_BEGIN) {
                phase = PH_SHADE;                           // no bounces at all: the path ends with its initial colour
            }
        }

        // ---- a slice of the walks (KDTreeCPU.cpp:640-708): one node or one leaf record per lane and trip
        {
            if (REDO) {                                     // rays that met a near tie, rays the short forms are not for: the reference's own walk, by the whole wave
                unsigned long long again = __ballot(phase == PH_REDO);
                while (again != 0ull) {
                    const int src = __ffsll((long long)again) - 1;
                    again &= again - 1ull;
                    const F3 O = f3(__shfl(s.O.x, src), __shfl(s.O.y, src), __shfl(s.O.z, src));
                    const F3 D = f3(__shfl(s.D.x, src), __shfl(s.D.y, src), __shfl(s.D.z, src));
                    const F3 invD = f3(__shfl(winvD.x, src), __shfl(winvD.y, src), __shfl(winvD.z, src));
                    Hit h; h.t = FLT_MAX; h.id = -1; h.u = 0.0f; h.v = 0.0f;
                    uint32_t cn = 0, ct = 0;
                    walk_reference_cooperative<STATS, false>(p, O, D, invD, lane, 0u, h, cn, ct);
                    n_nodes += cn; n_tris += ct;
                    if ((int)lane == src) { th = h; phase = PH_SHADE; }
                }
            }
            unsigned long long walking = __ballot(phase == PH_WALK);
            const uint32_t leave_at = ((uint32_t)__popcll(walking) * p.slice_leave) >> 2;
            // a long walk is a chain of dependent loads: its wave asks for issue priority, so that the few long rays that
            // decide when the frame ends are not queued behind the bulk of short ones
            if (__ballot(phase == PH_WALK && wsteps >= RT_LONG_WALK) != 0ull) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
            const unsigned long long slice_start = clocked ? __builtin_amdgcn_s_memrealtime() : 0ull;
            while ((uint32_t)__popcll(walking) > leave_at) {
                trips++;
                if (STATS) lane_trips += (unsigned long long)__popcll(walking);
                // a wave down to its last walks lends all its lanes to one of them at a time (the counting kernels keep the
                // one-lane walk, whose steps are the ones the reference makes)
                if (!STATS && (uint32_t)__popcll(walking) <= RT_COOP_LANES) {
                    const unsigned long long at_node = __ballot(phase == PH_WALK && wrec == RT_NO_REC);
                    if (at_node != 0ull) {
                        const int src = __ffsll((long long)at_node) - 1;
                        const F3 O = f3(__shfl(s.O.x, src), __shfl(s.O.y, src), __shfl(s.O.z, src));
                        const F3 D = f3(__shfl(s.D.x, src), __shfl(s.D.y, src), __shfl(s.D.z, src));
                        const F3 invD = f3(__shfl(winvD.x, src), __shfl(winvD.y, src), __shfl(winvD.z, src));
                        Hit h;
                        h.t = __shfl(th.t, src); h.id = __shfl(th.id, src); h.u = __shfl(th.u, src); h.v = __shfl(th.v, src);
                        uint32_t best = __shfl(wbest, src);
                        const uint32_t first = __shfl(wnode, src);
                        bool near_tie = false;
                        if (ORDERED || !RT_EXACT_REF_COOP) {
                            const float4* nodes = p.nodes + 2 * ((size_t)__shfl(woct, src) * (size_t)p.n_nodes);
                            near_tie = walk_cooperative<ORDERED, PLAIN>(p, nodes, O, D, invD, first, lane, ORDERED ? __shfl(wtol, src) : 0.0f, __shfl(sphere_cap(), src), h, best);
                        } else {
                            // the reference's order: the rest of the walk with its own decisions, every box pruned against the closest hit
                            // of the moment it is reached (windows decided with an older t would visit leaves the reference prunes)
                            uint32_t cn = 0, ct = 0;
                            walk_reference_cooperative<false, PLAIN>(p, O, D, invD, lane, first, h, cn, ct);
                        }
                        if ((int)lane == src) {
                            th = h; wbest = best; wnode = end;
                            wprune = ORDERED && RT_NEAR_TIES ? h.t + (wtol + p.tie_rel * h.t) : h.t;
                            if (near_tie || wflag) { phase = PH_REDO; wflag = false; atomicAdd(p.column_stats + 2, 1ull); }
                            else phase = PH_SHADE;
                        }
                        walking = __ballot(phase == PH_WALK);
                        continue;
                    }
                }
                if (phase == PH_WALK) {
                    // One read at the cursor per trip.  Descending goes to the next node in memory (and a leaf's records are
                    // consecutive), so the read holds the next node as well: up to RT_TRIP_NODES node steps (RT_TRIP_RECS record
                    // steps) are made from it, which shortens the chain of dependent loads that a long walk is -- at the price
                    // of instructions that run for the few lanes that kept descending (see rt_device.h for the measurements).
                    const bool at_node = wrec == RT_NO_REC;
                    // (the cursor is an index into arrays named by the kernel arguments, so these are global loads, not generic ones)
                    const float4* a = at_node ? p.nodes + 2 * ((size_t)woct * (size_t)p.n_nodes + checked(p, wnode, end, 1))
                                              : p.leafrecs + 3 * (size_t)checked(p, wrec, p.n_refs, 2);
                    constexpr bool wide = RT_TRIP_NODES >= 3 || RT_TRIP_RECS >= 2, mid = RT_TRIP_NODES >= 2 || wide;
                    const float4 q0 = a[0], q1 = a[1], q2 = a[2];
                    float4 q3 = q0, q4 = q0, q5 = q0;
                    if (mid) q3 = a[3];
                    if (wide) { q4 = a[4]; q5 = a[5]; }
                    static_assert(RT_WALK_FETCH == 6, "the walk reads up to six float4 per trip");
                    // returns true when the walk went down to the next node in memory
                    auto node_step = [&](float4 n0, float4 n1) -> bool {
                        wsteps++;
                        if (STATS) n_nodes++;
                        float tNear;
                        const bool box = slab_test_t<PLAIN>(n0, n1, s.O, winvD, tNear);
                        // "if (dist_aabb_near > t) return false; if (intersects_aabb) ..." -- dist is FLT_MAX on a box miss
                        const bool descend = box && !(tNear > (ORDERED ? wprune : th.t));
                        const int32_t leaf_first = (int32_t)__float_as_uint(n1.w);
                        if (descend && leaf_first == RT_LEAF_INTERIOR) { wnode = wnode + 1; return true; }
                        wnode = __float_as_uint(n0.w);      // skip link: first node after this subtree
                        if (descend && leaf_first >= 0) wrec = (uint32_t)leaf_first;
                        return false;
                    };
                    // returns true when the leaf has more records
                    auto rec_step = [&](float4 r0, float4 r1, float4 r2) -> bool {
                        wsteps++;
                        if (STATS) n_tris++;
                        const uint32_t tag = __float_as_uint(r2.y);
                        float tt, u, v;
                        if (tri_test_t<PLAIN>(r0, r1, r2, s.O, s.D, tt, u, v)) {
                            const int32_t id = (int32_t)(tag & 0x7FFFFFFFu);
                            if (ORDERED && RT_NEAR_TIES && id != th.id && fabsf(tt - th.t) <= wtol + p.tie_rel * fminf(tt, th.t)) wflag = true;
                            if (tt < th.t || (ORDERED && tt == th.t && th.id >= 0 && wrec < wbest)) {
                                th.t = tt; th.id = id; th.u = u; th.v = v; wbest = wrec;
                                wprune = fminf(RT_NEAR_TIES ? tt + (wtol + p.tie_rel * tt) : tt, sphere_cap());
                            } else if (th.id < 0) th.id = RT_HIT_DETECTED;      // (a test that succeeded at t = +inf)
                        }
                        const bool more = (tag & 0x80000000u) == 0u;
                        wrec = more ? wrec + 1u : RT_NO_REC;
                        return more;
                    };
                    if (at_node) {
                        bool down = node_step(q0, q1);
                        if (RT_TRIP_NODES >= 2 && down) down = node_step(q2, q3);
                        if (RT_TRIP_NODES >= 3 && down) node_step(q4, q5);
                    } else {
                        const bool more = rec_step(q0, q1, q2);
                        if (RT_TRIP_RECS >= 2 && more) rec_step(q3, q4, q5);
                    }
                    if (wrec == RT_NO_REC && wnode >= end) {
                        if (ORDERED && wflag) {
                            // Two different triangles within rounding distance of each other: which of them the reference
                            // returns depends on the leaves its own order happens to prune.  The wave walks this ray again,
                            // in that order (before its next slice).
                            phase = PH_REDO; wflag = false;
                            atomicAdd(p.column_stats + 2, 1ull);
                        } else phase = PH_SHADE;
                    }
                }
                walking = __ballot(phase == PH_WALK);
            }
            if (clocked) walk_ticks += __builtin_amdgcn_s_memrealtime() - slice_start;
        }

        // ---- the lanes whose ray is traced: the rest of the bounce (Renderer.cpp:160-226)
        const unsigned long long shade_start = clocked ? __builtin_amdgcn_s_memrealtime() : 0ull;
        if (phase == PH_SHADE) resolve_detected(th);
        if (phase == PH_SHADE && p.first_hit_mode == 1u) {
            // only the primary ray's mesh hit is wanted: kept for the launches that follow
            if (p.bounces != 0u) p.first_hits[pixel] = make_float4(th.t, th.u, th.v, __uint_as_float((uint32_t)th.id));
            wsteps = 0;
            phase = PH_IDLE;
        } else if (phase == PH_SHADE) {
            int32_t* id_out = (p.primary_ids && frame == 0) ? p.primary_ids + pixel : nullptr;
            bool ended = true;
            if (p.bounces != 0u) ended = shade_hit<MODE, false>(s, p, lds, id_out, s.bounce == 1u, s.bounce >= p.bounces, tS, iS, th);
            if (STATS) path_steps += wsteps;
            wsteps = 0;
            phase = PH_BEGIN;
            if (ended) {
                if (id_out && p.bounces == 0) *id_out = -1;
                // pixelColor of this Render: the resolve pass makes "m_AccumulationBuffer[pixelIndex] += pixelColor" of it, frame by
                // frame.  The fourth component carries, from the counting kernels, the node + triangle steps of the path.
                p.slots[(size_t)checked(p, frame, p.n_frames, 11) * plane + checked(p, pixel, plane, 12)] =
                    make_float4(s.finalColor.x, s.finalColor.y, s.finalColor.z, STATS ? (float)path_steps : 0.0f);
                phase = PH_IDLE;
                path_steps = 0;
            }
        }
        if (clocked) shade_ticks += __builtin_amdgcn_s_memrealtime() - shade_start;
    }
    __builtin_amdgcn_s_setprio(0);

    if (clocked && lane == 0) {
        unsigned long long* wc = p.wave_clock + RT_WAVE_CLOCK_WORDS * (size_t)wave;
        wc[0] = clock_start; wc[1] = __builtin_amdgcn_s_memrealtime(); wc[2] = n_iterations;
        wc[3] = trips; wc[4] = lane_trips;                  // lane balance = lane_trips / (64 trips), counting kernels only
        wc[5] = walk_ticks;                                 // time inside the walk slices
        wc[6] = wait_ticks; wc[7] = shade_ticks;            // ... waiting for paths from other waves, in the shading block
    }
    // work counters: one slot group per wave, summed on the host when read
    if (p.counters) {
        unsigned long long* slot = p.counters + 4 * (size_t)wave;
        uint32_t r = n_rays, n = n_nodes, t = n_tris, u = n_reused;
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_down(r, off); n += __shfl_down(n, off); t += __shfl_down(t, off); u += __shfl_down(u, off);
        }
        if (lane == 0 && u != 0u) atomicAdd(p.column_stats + 3, (unsigned long long)u);
        if (lane == 0) {
            slot[0] += (unsigned long long)r;
            if (STATS) { slot[1] += (unsigned long long)n; slot[2] += (unsigned long long)t; }
            if (MODE != RT_MODE_MESH) slot[3] += (unsigned long long)r * p.n_spheres;
        }
    }
}

// Sphere frames: workgroups of 16 waves, two per CU, and at most 80 scalar registers so that the chip admits the full 32 waves
// per CU (above 80 it silently admits fewer than the occupancy query promises, and the surplus workgroups of a persistent grid
// only start when others have finished).
template <bool STATS, bool LDS_SCENE>
__global__ __launch_bounds__(RT_SPHERE_BLOCK) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(8, 8))) void rt_trace_spheres_kernel(RtParams p)
{
    sphere_body<STATS, LDS_SCENE>(p);
}

#ifdef RT_TREE_WAVES
#define RT_TREE_OCCUPANCY __attribute__((amdgpu_waves_per_eu(RT_TREE_WAVES, RT_TREE_WAVES)))
#else
#define RT_TREE_OCCUPANCY
#endif
template <int MODE, bool STATS, bool ORDERED>
__global__ __launch_bounds__(RT_TREE_BLOCK) RT_TREE_OCCUPANCY void rt_trace_kernel(RtParams p)
{
    // tree scenes keep the closest-hit lookups in HBM: these kernels are bound by registers (waves per SIMD), not by those loads
    tree_body<MODE, STATS, ORDERED>(p);
}

// Tree scenes, after the trace launch: "m_AccumulationBuffer[pixelIndex] += pixelColor" (Renderer.cpp:64-66) for the launch's
// n_frames Renders in frame order -- frame index 1 starts from zero (Renderer.cpp:55-56) -- and the resolve of the last one
// (Renderer.cpp:68-79).  One lane per pixel of a work unit; the rows bordering a band (halo units) only get their sums.
__global__ __launch_bounds__(256) void rt_sum_resolve_kernel(RtParams p, uint32_t frame_index, int write_image)
{
    // The trace launch in front of this one has drawn its tickets and parked what it parked: the ticket heads and the two park
    // counters start the next launch at zero, so this kernel leaves them so (the host then does without two fill launches per
    // Render; it fills them itself whenever a launch sequence did not get as far as this kernel, see rt_ctx::heads_zero).
    if (blockIdx.x == 0 && threadIdx.x < RT_WORK_HEADS) p.work_heads[threadIdx.x * RT_HEAD_STRIDE] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 2u && p.park_count) p.park_count[threadIdx.x] = 0u;
    // write_image bit 1: only the listed units (RtParams::unit_list).  The host asks for that while the units without a traced path
    // hold what the first Render of the accumulation left there -- a sum of +0, its image and ids -- which further Renders, adding
    // +0 each, would only write again unchanged.
    uint32_t unit = blockIdx.x * (256u / RT_UNIT) + threadIdx.x / RT_UNIT;
    if (write_image & 2) {
        if (unit >= p.n_list) return;
        unit = p.unit_list[unit];
    }
    write_image &= 1;
    uint32_t pixel;
    if (unit >= p.n_units || !unit_pixel(p, unit, threadIdx.x % RT_UNIT, pixel)) return;
    const size_t plane = (size_t)p.width * p.height;
    F3 acc = f3(0.0f, 0.0f, 0.0f);
    if (p.frame_first != 1u) { const float4 a = p.accum[pixel]; acc = f3(a.x, a.y, a.z); }
    float steps = 0.0f;
    const bool empty = p.unit_empty && p.unit_empty[unit] != 0;        // no path was traced: each Render adds Miss()'s colour of zero
    if (empty && p.primary_ids) p.primary_ids[pixel] = -1;
    for (uint32_t f = 0; f < p.n_frames; ++f) {
        float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (!empty) c = p.slots[(size_t)f * plane + pixel];
        acc = acc + f3(c.x, c.y, c.z);
        steps = c.w;
    }
    p.accum[pixel] = make_float4(acc.x, acc.y, acc.z, steps);
    if (write_image && unit < p.n_units_main) p.image[pixel] = resolve_px(acc, (float)frame_index, p.use_ace);
}

// Which work units are empty (RtParams::unit_list): sixteen lanes per unit, one per pixel.  A unit is empty when every pixel of
// it that lies in the frame has a primary ray with direction.x != 0 (the others take the column pre-pass) that fails the root
// node's test exactly as the walk's first step would (slab_test, "descend" in tree_body).
// A workgroup of 1024 threads classifies 64 units and leaves one word for the scan that follows: the units it found to list in
// the low byte, above it the pixels of its empty units.  (One atomic per empty unit onto a single counter, as this kernel had it,
// took a millisecond of a 4K frame's classification: 370,000 atomics on one address.)
__global__ __launch_bounds__(1024) void rt_units_classify_kernel(RtParams p)
{
    __shared__ uint32_t wave_sum[16];
    const uint32_t unit = blockIdx.x * (1024u / RT_UNIT) + threadIdx.x / RT_UNIT;
    uint32_t pixel = 0;
    const bool in_frame = unit < p.n_units && unit_pixel(p, unit, threadIdx.x % RT_UNIT, pixel);
    bool live = false;                                      // this pixel's primary ray has to be traced
    if (in_frame) {
        const float4 d = p.dirs[pixel];
        float tNear = 0.0f;
        live = d.x == 0.0f || (slab_test(p.nodes[0], p.nodes[1], f3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]),
                                         f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z), tNear) && !(tNear > FLT_MAX));
    }
    const uint32_t lane = threadIdx.x & 63u, group = lane / RT_UNIT;
    static_assert(RT_UNIT == 16, "sixteen lanes per unit");
    const unsigned long long lives = __ballot(live), frames = __ballot(in_frame);
    uint32_t word = 0;                                      // this wave's four units: listed | pixels of the empty ones << 8
    for (uint32_t g = 0; g < 4u; ++g) {
        const bool exists = blockIdx.x * (1024u / RT_UNIT) + (threadIdx.x >> 6) * 4u + g < p.n_units;
        const bool empty = ((lives >> (RT_UNIT * g)) & 0xFFFFull) == 0ull;
        if (exists) word += empty ? (uint32_t)__popcll((frames >> (RT_UNIT * g)) & 0xFFFFull) << 8 : 1u;
    }
    if (unit < p.n_units && threadIdx.x % RT_UNIT == 0u) p.unit_empty_out[unit] = ((lives >> (RT_UNIT * group)) & 0xFFFFull) == 0ull ? 1 : 0;
    if (lane == 0u) wave_sum[threadIdx.x >> 6] = word;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (uint32_t w = 0; w < 16u; ++w) sum += wave_sum[w];
        p.unit_block_out[blockIdx.x] = sum;
    }
}

// One workgroup: the listed units before each group of 64 (an exclusive prefix sum over the words of the kernel above, in
// place), their total and the pixels of all empty units into unit_counts[0 .. 1]; the two counters of the entry kernel start at zero.
__global__ __launch_bounds__(1024) void rt_units_scan_kernel(RtParams p, uint32_t n_blocks)
{
    __shared__ uint32_t part[1024], pix[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (n_blocks + 1023u) / 1024u;
    const uint32_t lo = tid * per < n_blocks ? tid * per : n_blocks, hi = lo + per < n_blocks ? lo + per : n_blocks;
    uint32_t sum = 0, pixels = 0;
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = p.unit_block_out[i]; sum += v & 0xFFu; pixels += v >> 8; }
    part[tid] = sum; pix[tid] = pixels;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {               // Hillis-Steele inclusive scan (the pixel counts only need their total)
        const uint32_t v = tid >= off ? part[tid - off] : 0u, w = tid >= off ? pix[tid - off] : 0u;
        __syncthreads();
        part[tid] += v; pix[tid] += w;
        __syncthreads();
    }
    uint32_t run = tid ? part[tid - 1] : 0u;
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = p.unit_block_out[i] & 0xFFu; p.unit_block_out[i] = run; run += v; }
    if (tid == 1023u) { p.unit_counts_out[0] = part[1023]; p.unit_counts_out[1] = pix[1023]; p.unit_counts_out[2] = 0u; p.unit_counts_out[3] = 0u; }
}

// The listed units in ascending order: one wave per group of 64 units, one lane per unit.
__global__ __launch_bounds__(256) void rt_units_assign_kernel(RtParams p)
{
    const uint32_t group64 = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t unit = group64 * 64u + lane;
    const bool listed = unit < p.n_units && p.unit_empty_out[unit] == 0;
    const unsigned long long m = __ballot(listed);
    if (listed) p.unit_list_out[p.unit_block_out[group64] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = unit;
}

// Where the walks of a listed unit's primary rays may start (RtParams::unit_entry): sixteen lanes per unit walk in step, with the
// decisions of tree_body's node_step / rec_step for a ray that has no hit yet, until two of them decide differently or one
// of them hits a triangle.  Rays with direction.x == 0 do not walk (pre-pass) and have no say.
__global__ __launch_bounds__(256) void rt_units_entry_kernel(RtParams p)
{
    __shared__ uint32_t wave_nodes[4], wave_tris[4];
    const uint32_t li = blockIdx.x * (256u / RT_UNIT) + threadIdx.x / RT_UNIT;
    const bool listed = li < p.unit_counts_out[0];           // (the grid covers every unit there could be)
    const uint32_t unit = listed ? p.unit_list_out[li] : 0u;
    uint32_t pixel = 0;
    const bool in_frame = listed && unit_pixel(p, unit, threadIdx.x % RT_UNIT, pixel);
    const uint32_t group = (threadIdx.x & 63u) / RT_UNIT;
    const unsigned long long mask = 0xFFFFull << (RT_UNIT * group);
    F3 O = f3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]), D = f3(1.0f, 1.0f, 1.0f);
    if (in_frame) { const float4 d = p.dirs[pixel]; D = f3(d.x, d.y, d.z); }
    const bool live = in_frame && D.x != 0.0f;
    const F3 invD = f3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
    uint32_t oct = 0;
    if (p.entry_ordered) oct = (D.x < 0.0f ? 1u : 0u) | (D.y < 0.0f ? 2u : 0u) | (D.z < 0.0f ? 4u : 0u);
    // one node array for the whole unit, or no common start
    const unsigned long long lives = __ballot(live) & mask;
    const uint32_t oct0 = lives ? (uint32_t)__shfl((int)oct, __ffsll((long long)lives) - 1) : 0u;
    bool done = lives == 0ull || (__ballot(live && oct != oct0) & mask) != 0ull;
    const float4* nodes = p.nodes + 2 * ((size_t)oct0 * (size_t)p.n_nodes);
    const uint32_t end = p.n_nodes;
    uint32_t cursor = 0, n_nodes = 0, n_tris = 0;           // steps the walks of this unit's live rays are spared
    while (__ballot(!done) != 0ull) {
        if (!done) {
            if (cursor >= end) done = true;                 // every ray of the unit misses everything
            else {
                const float4 n0 = nodes[2 * (size_t)cursor], n1 = nodes[2 * (size_t)cursor + 1];
                float tNear;
                const bool box = slab_test(n0, n1, O, invD, tNear);
                const bool descend = box && !(tNear > FLT_MAX);
                const unsigned long long yes = __ballot(live && descend) & mask, no = __ballot(live && !descend) & mask;
                const int32_t leaf_first = (int32_t)__float_as_uint(n1.w);
                if (yes != 0ull && no != 0ull) done = true;
                else if (no != 0ull || (leaf_first < 0 && leaf_first != RT_LEAF_INTERIOR)) { cursor = __float_as_uint(n0.w); n_nodes++; }   // all miss the box, or a leaf without records
                else if (leaf_first == RT_LEAF_INTERIOR) { cursor = cursor + 1u; n_nodes++; }
                else {
                    bool hit = false;
                    uint32_t tested = 0;
                    for (uint32_t rec = (uint32_t)leaf_first;; ++rec) {
                        const float4* r = p.leafrecs + 3 * (size_t)rec;
                        const float4 r0 = r[0], r1 = r[1], r2 = r[2];
                        float tt, u, v;
                        if (tri_test(r0, r1, r2, O, D, tt, u, v) && live) hit = true;
                        tested++;
                        if (__float_as_uint(r2.y) & 0x80000000u) break;
                    }
                    if ((__ballot(hit) & mask) != 0ull) done = true;
                    else { cursor = __float_as_uint(n0.w); n_nodes++; n_tris += tested; }
                }
            }
        }
    }
    if (in_frame) p.unit_entry_out[pixel] = cursor;
    if (!live) { n_nodes = 0; n_tris = 0; }
    for (int off = 32; off > 0; off >>= 1) { n_nodes += __shfl_down(n_nodes, off); n_tris += __shfl_down(n_tris, off); }
    if ((threadIdx.x & 63u) == 0u) { wave_nodes[threadIdx.x >> 6] = n_nodes; wave_tris[threadIdx.x >> 6] = n_tris; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nn = (wave_nodes[0] + wave_nodes[1]) + (wave_nodes[2] + wave_nodes[3]), nt = (wave_tris[0] + wave_tris[1]) + (wave_tris[2] + wave_tris[3]);
        if (nn) atomicAdd(p.unit_counts_out + 2, nn);
        if (nt) atomicAdd(p.unit_counts_out + 3, nt);
    }
}

// Resolve (Renderer.cpp:68-79): image = tonemap(accum / frame_index), over the bands this context owns.
__global__ __launch_bounds__(256) void rt_resolve_kernel(RtParams p, uint32_t frame_index)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * RT_TILE + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t band = blockIdx.y * p.band_count + p.band_index;
    const uint32_t y = band * RT_TILE + (wave >> 1) * 8u + (lane >> 3);
    if (x < p.width && y < p.height) {
        const uint32_t pixel = y * p.width + x;
        float4 a = p.accum[pixel];
        p.image[pixel] = resolve_px(f3(a.x, a.y, a.z), (float)frame_index, p.use_ace);
    }
}


// Anti-alias filter (Renderer.cpp:84-124): 3x3 window, taps with |xo + yo| > 1 skipped, weight 1 - (|xo| + |yo|) / 2
// (so the two anti-diagonal taps carry weight 0 but still take part), on the raw accumulation; overwrites the image.
__global__ __launch_bounds__(256) void rt_aa_kernel(RtParams p, uint32_t frame_index)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * RT_TILE + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t band = blockIdx.y * p.band_count + p.band_index;
    const uint32_t y = band * RT_TILE + (wave >> 1) * 8u + (lane >> 3);
    if (x >= p.width || y >= p.height) return;
    F3 acc = f3(0.0f, 0.0f, 0.0f);
    float weight = 0.0f;
    for (int yo = -1; yo <= 1; ++yo)
        for (int xo = -1; xo <= 1; ++xo) {
            const int xx = (int)x + xo, yy = (int)y + yo;
            const int sum = xo + yo;
            if ((sum < 0 ? -sum : sum) > 1 || xx < 0 || xx >= (int)p.width || yy < 0 || yy >= (int)p.height) continue;
            const float wgt = 1.0f - ((float)((yo < 0 ? -yo : yo) + (xo < 0 ? -xo : xo)) * 0.5f);
            float4 a = p.accum[(uint32_t)yy * p.width + (uint32_t)xx];
            acc = acc + f3(a.x, a.y, a.z) * wgt;
            weight += wgt;
        }
    if (weight == 0.0f) return;
    p.image[y * p.width + x] = resolve_px(acc, weight * (float)frame_index, p.use_ace);
}

// ---- experiment: a list of rays walked front to back, one lane per ray ----------------------------------------------------------
// The walk of the timed kernels (the node array of the ray's sign pattern, the short box test, up to three node steps or two
// record steps from one read of six float4, pruning at the closest hit plus the near-tie distance) applied to rays given as a
// list -- in whatever order the caller put them.  What a sorted order of scattered rays could buy is measured with this: the
// same rays in the order the waves of the trace kernel started them and sorted by direction octant and origin cell.
// refill = 0: ray i belongs to thread i; refill = 1: a persistent grid whose lanes draw the next rays of the list, sixteen at
// least at a time, as their walks end.  VARIANT only names the launch (profiles tell the two orders apart by it).
// out[i] = { t, u, v, bits(triangle) } (-1: no hit, -3: a ray the short box test is not exact for, not walked).
template <int VARIANT>
__global__ __launch_bounds__(256) void rt_probe_walk_kernel(RtParams p, const float4* __restrict__ rays, uint32_t n, uint32_t* next, int refill,
                                                            float4* __restrict__ out)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t end = p.n_nodes;
    bool walking = false, exhausted = false;
    uint32_t ray = 0;
    F3 O = f3(0.0f, 0.0f, 0.0f), D = O, invD = O;
    Hit th; th.t = FLT_MAX; th.id = -1; th.u = 0.0f; th.v = 0.0f;
    uint32_t wnode = 0, wrec = RT_NO_REC, wbest = RT_NO_REC, woct = 0;
    float wtol = 0.0f, wprune = FLT_MAX;
    auto begin = [&](uint32_t i) {
        ray = i;
        const float4 a = rays[2 * (size_t)i], b = rays[2 * (size_t)i + 1];
        O = f3(a.x, a.y, a.z); D = f3(b.x, b.y, b.z);
        invD = f3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
        woct = (D.x < 0.0f ? 1u : 0u) | (D.y < 0.0f ? 2u : 0u) | (D.z < 0.0f ? 4u : 0u);
        th.t = FLT_MAX; th.id = -1; th.u = 0.0f; th.v = 0.0f;
        wnode = 0; wrec = RT_NO_REC; wbest = RT_NO_REC; wprune = FLT_MAX;
        const float bx = (fabsf(O.x - p.root_center[0]) + p.tie_box[0]) * fabsf(invD.x);
        const float by = (fabsf(O.y - p.root_center[1]) + p.tie_box[1]) * fabsf(invD.y);
        const float bz = (fabsf(O.z - p.root_center[2]) + p.tie_box[2]) * fabsf(invD.z);
        wtol = fmaxf(bx, fmaxf(by, bz)) * p.tie_ulp + p.tie_abs;
        walking = true;
        if (!(p.slab_plain != 0u && slab_plain_ray(O, invD)) || end == 0u) {
            out[i] = make_float4(FLT_MAX, 0.0f, 0.0f, __uint_as_float((uint32_t)-3));
            walking = false;
        }
    };
    if (!refill) {
        const uint32_t i = blockIdx.x * 256u + threadIdx.x;
        if (i < n) begin(i);
        exhausted = true;
    }
    for (;;) {
        if (!exhausted) {
            const unsigned long long idle = __ballot(!walking);
            const uint32_t want = (uint32_t)__popcll(idle);
            if (want >= 16u) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(next, want);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t mine = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                if (!walking && mine < n) begin(mine);
                if (base + want >= n) exhausted = true;
            }
        }
        if (__ballot(walking) == 0ull) { if (exhausted) break; else continue; }
        if (walking) {
            const bool at_node = wrec == RT_NO_REC;
            const float4* a = at_node ? p.nodes + 2 * ((size_t)woct * (size_t)p.n_nodes + wnode) : p.leafrecs + 3 * (size_t)wrec;
            const float4 q0 = a[0], q1 = a[1], q2 = a[2], q3 = a[3], q4 = a[4], q5 = a[5];
            auto node_step = [&](float4 n0, float4 n1) -> bool {
                float tNear;
                const bool box = slab_test_t<true>(n0, n1, O, invD, tNear);
                const bool descend = box && !(tNear > wprune);
                const int32_t leaf_first = (int32_t)__float_as_uint(n1.w);
                if (descend && leaf_first == RT_LEAF_INTERIOR) { wnode = wnode + 1; return true; }
                wnode = __float_as_uint(n0.w);
                if (descend && leaf_first >= 0) wrec = (uint32_t)leaf_first;
                return false;
            };
            auto rec_step = [&](float4 r0, float4 r1, float4 r2) -> bool {
                const uint32_t tag = __float_as_uint(r2.y);
                float tt, u, v;
                if (tri_test_t<true>(r0, r1, r2, O, D, tt, u, v)) {
                    if (tt < th.t || (tt == th.t && th.id >= 0 && wrec < wbest)) {
                        th.t = tt; th.id = (int32_t)(tag & 0x7FFFFFFFu); th.u = u; th.v = v; wbest = wrec;
                        wprune = tt + (wtol + p.tie_rel * tt);
                    }
                }
                const bool more = (tag & 0x80000000u) == 0u;
                wrec = more ? wrec + 1u : RT_NO_REC;
                return more;
            };
            if (at_node) { if (node_step(q0, q1) && node_step(q2, q3)) node_step(q4, q5); }
            else { if (rec_step(q0, q1, q2)) rec_step(q3, q4, q5); }
            if (wrec == RT_NO_REC && wnode >= end) {
                out[ray] = make_float4(th.t, th.u, th.v, __uint_as_float((uint32_t)th.id));
                walking = false;
            }
        }
    }
}

extern "C" hipError_t rt_launch_probe_walk(const RtParams* p, const float4* rays, uint32_t n, uint32_t* next, int refill, int variant,
                                           unsigned grid, float4* out, hipStream_t s)
{
    if (variant) hipLaunchKernelGGL(rt_probe_walk_kernel<1>, dim3(grid), dim3(256), 0, s, *p, rays, n, next, refill, out);
    else hipLaunchKernelGGL(rt_probe_walk_kernel<0>, dim3(grid), dim3(256), 0, s, *p, rays, n, next, refill, out);
    return hipGetLastError();
}

extern "C" hipError_t rt_probe_walk_occupancy(int* blocks_per_cu)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, rt_probe_walk_kernel<0>, 256, 0);
}

// ---- launchers (called from rt_api.cpp) -----------------------------------------------------------------------------
// Dispatch over (mode, stats, ordered); sphere-only frames have no tree to order, and keep spheres + materials in LDS when
// there are few enough of them (`small_scene`).
template <class F>
static hipError_t for_variant(int mode, int stats, int ordered, int small_scene, F&& f)
{
    if (mode == RT_MODE_SPHERES) {
        if (small_scene) return stats ? f(rt_trace_spheres_kernel<true, true>) : f(rt_trace_spheres_kernel<false, true>);
        return stats ? f(rt_trace_spheres_kernel<true, false>) : f(rt_trace_spheres_kernel<false, false>);
    }
    const int key = (mode * 2 + (stats ? 1 : 0)) * 2 + ((ordered && mode != RT_MODE_SPHERES) ? 1 : 0);
    switch (key) {
    case 0:  return f(rt_trace_kernel<RT_MODE_MESH, false, false>);
    case 1:  return f(rt_trace_kernel<RT_MODE_MESH, false, true>);
    case 2:  return f(rt_trace_kernel<RT_MODE_MESH, true, false>);
    case 3:  return f(rt_trace_kernel<RT_MODE_MESH, true, true>);
    case 8:  return f(rt_trace_kernel<RT_MODE_BOTH, false, false>);
    case 9:  return f(rt_trace_kernel<RT_MODE_BOTH, false, true>);
    case 10: return f(rt_trace_kernel<RT_MODE_BOTH, true, false>);
    case 11: return f(rt_trace_kernel<RT_MODE_BOTH, true, true>);
    default: return hipErrorInvalidValue;
    }
}

// Resident workgroups per CU of the trace kernel variant: sizes the persistent grid.
// Threads per workgroup of the variant.
extern "C" unsigned rt_trace_block(int mode) { return mode == RT_MODE_SPHERES ? RT_SPHERE_BLOCK : (unsigned)RT_TREE_BLOCK; }

extern "C" hipError_t rt_trace_occupancy(int mode, int stats, int ordered, int* blocks_per_cu)
{
    return for_variant(mode, stats, ordered, 1, [&](auto kernel) {
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, kernel, (int)rt_trace_block(mode), 0);
    });
}

extern "C" hipError_t rt_launch_trace(const RtParams* p, int mode, int stats, int ordered, unsigned grid, hipStream_t s)
{
    const int small_scene = p->n_spheres <= RT_LDS_SPHERES && p->n_materials <= RT_LDS_MATERIALS;
    return for_variant(mode, stats, ordered, small_scene, [&](auto kernel) {
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(rt_trace_block(mode)), 0, s, *p);
        return hipGetLastError();
    });
}

extern "C" hipError_t rt_launch_raygen(const RtParams* p, hipStream_t s)
{
    const unsigned n = p->width * p->height;
    hipLaunchKernelGGL(rt_raygen_kernel, dim3((n + 255u) / 256u), dim3(256), 0, s, *p);
    return hipGetLastError();
}

// Number the column rays of the current directions (pixel order) and box their directions.
extern "C" hipError_t rt_launch_column_collect(const RtParams* p, hipStream_t s)
{
    const unsigned n = p->width * p->height, blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(rt_column_count_kernel, dim3(blocks), dim3(256), 0, s, *p);
    hipLaunchKernelGGL(rt_column_scan_kernel, dim3(1), dim3(1024), 0, s, *p, blocks);
    hipLaunchKernelGGL(rt_column_assign_kernel, dim3(blocks), dim3(256), 0, s, *p);
    hipLaunchKernelGGL(rt_column_chunks_kernel, dim3((p->column_cap + 255u) / 256u), dim3(256), 0, s, *p);     // (a workgroup per 256 slots; those beyond the count return at once)
    return hipGetLastError();
}

static void launch_reach(const RtParams& p, int stats, hipStream_t s)
{
    const unsigned node_blocks = (p.n_nodes + 255u) / 256u;
    hipLaunchKernelGGL(rt_column_reach_max_kernel, dim3(node_blocks), dim3(256), 0, s, p);
    hipLaunchKernelGGL(rt_column_reach_scan_kernel, dim3(1), dim3(1024), 0, s, p, node_blocks);
    hipLaunchKernelGGL(rt_column_reach_mark_kernel, dim3(node_blocks), dim3(256), 0, s, p, stats);
}

// The hits of the column rays (column_best, column_flags).  They depend on the camera, the directions, the bands and the tree
// only: the caller keeps them until one of those changes.
extern "C" hipError_t rt_launch_column_prepass(const RtParams* p, int stats, hipStream_t s)
{
    RtParams q = *p; q.ray_slot = -1;
    hipLaunchKernelGGL(rt_column_reset_kernel, dim3((q.column_cap + 255u) / 256u), dim3(256), 0, s, q);
    if (q.n_refs == 0u) return hipGetLastError();                              // a tree without a triangle: every column ray misses
    launch_reach(q, stats, s);
    hipLaunchKernelGGL(rt_column_filter_kernel, dim3((q.n_refs + 255u) / 256u), dim3(256), 0, s, q, stats);
    hipLaunchKernelGGL(rt_column_exact_kernel, dim3(2048), dim3(256), 0, s, q);
    return hipGetLastError();
}

// The hits of the parked paths of the resume list, family by family: the families sorted and boxed, per family the visited
// nodes, the interval filter over the records of the visited leaves, the real test on what is left.  The resume launch of
// the trace kernel follows.
extern "C" hipError_t rt_launch_parked_rays(const RtParams* p, int stats, hipStream_t s)
{
    RtParams q = *p;
    hipLaunchKernelGGL(rt_park_group_kernel, dim3(1), dim3(1024), 0, s, q);
    hipLaunchKernelGGL(rt_park_sort_kernel, dim3(RT_PARK_FAMILIES), dim3(1024), 0, s, q);
    for (int family = 0; family < RT_PARK_FAMILIES; ++family) {
        q.ray_slot = family;
        launch_reach(q, stats, s);
        if (q.n_refs != 0u) hipLaunchKernelGGL(rt_park_filter_kernel, dim3((q.n_refs + 255u) / 256u), dim3(256), 0, s, q, stats);
        hipLaunchKernelGGL(rt_park_test_kernel, dim3(2048), dim3(256), 0, s, q);       // (a tree without a triangle: every ray misses)
    }
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_resolve(const RtParams* p, unsigned frame_index, unsigned gx, unsigned gy, hipStream_t s)
{
    hipLaunchKernelGGL(rt_resolve_kernel, dim3(gx, gy), dim3(256), 0, s, *p, frame_index);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_sum_resolve(const RtParams* p, unsigned frame_index, int write_image, hipStream_t s)
{
    const unsigned per_block = 256u / RT_UNIT;
    const unsigned units = (write_image & 2) ? (p->n_list ? p->n_list : 1u) : p->n_units;      // (block 0 always runs: it zeroes the ticket heads)
    hipLaunchKernelGGL(rt_sum_resolve_kernel, dim3((units + per_block - 1u) / per_block), dim3(256), 0, s, *p, frame_index, write_image);
    return hipGetLastError();
}

// Words of RtParams::unit_block_out the classification needs: one per 64 units.
extern "C" unsigned rt_units_blocks(unsigned n_units) { return (n_units + 63u) / 64u; }

extern "C" hipError_t rt_launch_units_classify(const RtParams* p, hipStream_t s)
{
    const unsigned blocks = rt_units_blocks(p->n_units);
    hipLaunchKernelGGL(rt_units_classify_kernel, dim3(blocks), dim3(1024), 0, s, *p);
    hipLaunchKernelGGL(rt_units_scan_kernel, dim3(1), dim3(1024), 0, s, *p, blocks);
    hipLaunchKernelGGL(rt_units_assign_kernel, dim3((blocks + 3u) / 4u), dim3(256), 0, s, *p);
    return hipGetLastError();
}

// `max_listed`: an upper bound of the listed units (their number itself is read on the device).
extern "C" hipError_t rt_launch_units_entry(const RtParams* p, unsigned max_listed, hipStream_t s)
{
    const unsigned per_block = 256u / RT_UNIT;
    if (max_listed) hipLaunchKernelGGL(rt_units_entry_kernel, dim3((max_listed + per_block - 1u) / per_block), dim3(256), 0, s, *p);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_aa(const RtParams* p, unsigned frame_index, unsigned gx, unsigned gy, hipStream_t s)
{
    hipLaunchKernelGGL(rt_aa_kernel, dim3(gx, gy), dim3(256), 0, s, *p, frame_index);
    return hipGetLastError();
}
