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
// glm::normalize = v * (1 / sqrt(dot(v, v))) (func_geometric.inl:82-90, func_exponential.inl:134-139)
__device__ __forceinline__ F3 normalize3(F3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }
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

__device__ __forceinline__ float tonemap1(float x)
{
    float num = x * (2.51f * x + 0.03f);
    float den = x * (2.43f * x + 0.59f) + 0.14f;
    float c = clamp01(num / den);          // ACESFilm
    c = clamp01(c);                        // LinearToSRGB's own clamp
    float hi = pow_inv24(c) * 1.055f - 0.055f;
    float lo = c * 12.92f;
    float a = (c < 0.0031308f) ? 1.0f : 0.0f;
    return hi * (1.0f - a) + lo * a;
}

__device__ __forceinline__ float4 resolve_px(F3 acc, float div, int use_ace)
{
    float r = acc.x / div, g = acc.y / div, b = acc.z / div;
    if (use_ace) { r = tonemap1(r); g = tonemap1(g); b = tonemap1(b); }
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

// ---- spheres: Renderer.cpp:243-271 ------------------------------------------------------------------------------
__device__ __forceinline__ void trace_spheres(const RtParams& p, F3 O, F3 D, float& closest, int& index)
{
    float a = dot3(D, D);
    float dbl_a = 2.0f * a;
    float four_a = 4.0f * a;
    for (uint32_t i = 0; i < p.n_spheres; ++i) {
        float4 sp = p.spheres[i];                      // wave-uniform address: scalar load
        F3 ro = O - f3(sp.x, sp.y, sp.z);
        float b = 2.0f * dot3(ro, D);
        float c = dot3(ro, ro) - sp.w;                 // sp.w = radius*radius, rounded once at upload
        float disc = b * b - four_a * c;
        if (disc >= 0.0f) {
            float t = (-b - sqrtf(disc)) / dbl_a;
            if (t > 0.0f && t < closest) { closest = t; index = (int)i; }
        }
    }
}

// ---- KD tree: KDTreeCPU.cpp:640-708 over the preorder/skip layout; Intersections.cpp:23-94 ------------------------
template <bool STATS>
__device__ __forceinline__ void trace_mesh(const RtParams& p, F3 O, F3 D, F3 invD, Hit& hit, uint32_t& n_nodes_visited,
                                           uint32_t& n_tris_tested)
{
    float t = FLT_MAX;
    int32_t tri = -1;
    float hu = 0.0f, hv = 0.0f;
    uint32_t node = 0;
    const uint32_t end = p.n_nodes;
    while (node < end) {
        const float4 n0 = p.nodes[2 * node];
        const float4 n1 = p.nodes[2 * node + 1];
        if (STATS) n_nodes_visited++;
        // aabbIntersect
        F3 l1 = (f3(n0.x, n0.y, n0.z) - O) * invD;
        F3 l2 = f3(n1.x, n1.y, n1.z) * invD;
        F3 tMin = l1 - l2;
        F3 tMax = l1 + l2;
        float t2x = sel_max(tMin.x, tMax.x), t2y = sel_max(tMin.y, tMax.y), t2z = sel_max(tMin.z, tMax.z);
        float tFar = sel_min(sel_min(t2x, t2y), t2z);
        float t1x = sel_min(tMin.x, tMax.x), t1y = sel_min(tMin.y, tMax.y), t1z = sel_min(tMin.z, tMax.z);
        float tNear = sel_max(sel_max(t1x, t1y), t1z);
        bool box = !(tFar < 0.0f) && !(tNear > tFar);
        // "if (dist_aabb_near > t) return false; if (intersects_aabb) ..." -- dist is FLT_MAX on a box miss
        bool descend = box && !(tNear > t);
        const uint32_t skip = __float_as_uint(n0.w);
        const int32_t leaf_first = (int32_t)__float_as_uint(n1.w);
        if (!descend) { node = skip; continue; }
        if (leaf_first == RT_LEAF_INTERIOR) { node = node + 1; continue; }
        if (leaf_first >= 0) {
            const float4* rec = p.leafrecs + 3 * (size_t)leaf_first;
            for (;;) {
                const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                rec += 3;
                if (STATS) n_tris_tested++;
                const uint32_t tag = __float_as_uint(r2.y);
                // triIntersect (e1, e2 precomputed)
                F3 v0 = f3(r0.x, r0.y, r0.z), e1 = f3(r0.w, r1.x, r1.y), e2 = f3(r1.z, r1.w, r2.x);
                F3 h = cross3(D, e2);
                float a = dot3(e1, h);
                bool ok = !(a > -0.00001f && a < 0.00001f);
                float f = 1.0f / a;
                F3 s = O - v0;
                float u = f * dot3(s, h);
                ok = ok && !(u < 0.0f || u > 1.0f);
                F3 q = cross3(s, e1);
                float v = f * dot3(D, q);
                ok = ok && !(v < 0.0f || u + v > 1.0f);
                float tt = f * dot3(e2, q);
                ok = ok && (tt > 0.00001f);
                if (ok && tt < t) { t = tt; tri = (int32_t)(tag & 0x7FFFFFFFu); hu = u; hv = v; }
                if (tag & 0x80000000u) break;
            }
        }
        node = skip;
    }
    hit.t = t; hit.id = tri; hit.u = hu; hit.v = hv;
}

// Util::RandomUnitVector (Renderer.h:46-54)
__device__ __forceinline__ F3 random_unit_vector(unsigned long long& rng)
{
    const float TwoPi = 2.0f * 3.14159274f;
    float z = rng_float(rng) * 2.0f - 1.0f;
    float a = rng_float(rng) * TwoPi;
    float r = sqrtf(1.0f - z * z);
    float sn, cs;
    sincos_portable(a, sn, cs);
    return f3(r * cs, r * sn, z);
}

constexpr float EPSILON = 0.0002f;         // Renderer.cpp:14

// Renderer::PerPixel (Renderer.cpp:138-229) for one (pixel, frame)
template <int MODE, bool STATS>
__device__ __forceinline__ F3 per_pixel(const RtParams& p, F3 O, F3 D, unsigned long long rng, int32_t& primary_id,
                                        uint32_t& n_rays, uint32_t& n_nodes, uint32_t& n_tris)
{
    F3 finalColor = f3(0.0f, 0.0f, 0.0f);
    F3 contribution = f3(1.0f, 1.0f, 1.0f);
    const F3 ambientColor = f3(0.0f, 0.0f, 0.0f);
    primary_id = -1;

    for (uint32_t i = 0; i < p.bounces; ++i) {
        F3 invD = f3(1.0f / D.x, 1.0f / D.y, 1.0f / D.z);
        n_rays++;

        // TraceRay (Renderer.cpp:231-301)
        float tS = FLT_MAX; int iS = -1;
        Hit th; th.t = FLT_MAX; th.id = -1; th.u = 0.0f; th.v = 0.0f;
        if (MODE == RT_MODE_SPHERES || MODE == RT_MODE_BOTH) trace_spheres(p, O, D, tS, iS);
        if (MODE == RT_MODE_MESH || MODE == RT_MODE_BOTH) trace_mesh<STATS>(p, O, D, invD, th, n_nodes, n_tris);

        if (iS < 0 && th.id < 0) {                 // Miss(): Distance = -1
            finalColor = finalColor + contribution * ambientColor;
            break;
        }
        F3 P, N;
        uint32_t matIndex;
        if (tS < th.t) {                            // ClosestHitSphere (Renderer.cpp:303-317)
            float4 sp = p.spheres[iS];
            P = O + D * tS;
            N = normalize3(P - f3(sp.x, sp.y, sp.z));
            matIndex = p.sphere_mat[iS];
            if (i == 0) primary_id = iS;
        } else {                                    // ClosestHitTriangle (Renderer.cpp:319-332)
            const float4* sh = p.trishade + 3 * (size_t)th.id;
            float4 s0 = sh[0], s1 = sh[1], s2 = sh[2];
            P = O + D * th.t;
            F3 nrm = (f3(s0.x, s0.y, s0.z) * ((1.0f - th.u) - th.v) + f3(s0.w, s1.x, s1.y) * th.u)
                     + f3(s1.z, s1.w, s2.x) * th.v;
            N = normalize3(nrm);
            matIndex = __float_as_uint(s2.y);
            if (i == 0) primary_id = RT_ID_TRI_BIT | th.id;
        }

        const float4 m0 = p.materials[3 * matIndex];       // albedo.rgb, roughness
        const float4 m1 = p.materials[3 * matIndex + 1];   // emission.rgb, transparency
        bool hitInside = dot3(D, N) > 0.0f;
        F3 normalSurface = hitInside ? -N : N;
        O = P + normalSurface * EPSILON;

        if (m1.w > 0.0f) {                          // always transmit (Renderer.cpp:176-197)
            const float IOR = p.materials[3 * matIndex + 2].x;
            // RefractionRay (Renderer.cpp:334-361)
            F3 ref_n = N;
            float eta_t = IOR, eta_i = 1.0f;
            float i_dot_n = dot3(D, N);
            if (i_dot_n < 0.0f) i_dot_n = -i_dot_n;
            else { ref_n = -N; eta_t = 1.0f; eta_i = IOR; }
            float eta = eta_i / eta_t;
            float k = 1.0f - (eta * eta) * (1.0f - i_dot_n * i_dot_n);
            if (!(k < 0.0f)) {
                O = P + ref_n * (-EPSILON);
                D = normalize3((D + ref_n * i_dot_n) * eta - ref_n * sqrtf(k));
                continue;                           // skips emission, albedo and roulette
            }
            // total internal reflection: fall through with D unchanged
        } else {
            F3 diffuseRayDir = normalize3(N + random_unit_vector(rng));
            F3 reflectedVector = reflect3(D, N);
            D = normalize3(mix3(reflectedVector, diffuseRayDir, m0.w * m0.w));
        }

        finalColor = finalColor + f3(m1.x, m1.y, m1.z) * contribution;
        contribution = contribution * f3(m0.x, m0.y, m0.z);

        float pr = sel_max(contribution.x, sel_max(contribution.y, contribution.z));
        if (rng_float(rng) > pr) break;
        contribution = contribution * (1.0f / pr);
    }
    return finalColor;
}

}  // namespace

// ---- kernels ------------------------------------------------------------------------------------------------------
// grid = (ceil(w/16), bands owned by this context); 256 threads = 4 waves, each an 8x8 pixel tile, so a wave's accumulation
// and image accesses are eight 128-byte row segments.
template <int MODE, bool STATS>
__global__ __launch_bounds__(256) void rt_trace_kernel(RtParams p)
{
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t x = blockIdx.x * RT_TILE + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t band = blockIdx.y * p.band_count + p.band_index;
    const uint32_t y = band * RT_TILE + (wave >> 1) * 8u + (lane >> 3);
    uint32_t n_rays = 0, n_nodes = 0, n_tris = 0;

    if (x < p.width && y < p.height) {
        const uint32_t pixel = y * p.width + x;
        const F3 camPos = f3(p.cam_pos[0], p.cam_pos[1], p.cam_pos[2]);
        F3 dir;
        if (p.ray_dirs) dir = f3(p.ray_dirs[3 * (size_t)pixel], p.ray_dirs[3 * (size_t)pixel + 1], p.ray_dirs[3 * (size_t)pixel + 2]);
        else dir = camera_dir(p, x, y);

        F3 acc = f3(0.0f, 0.0f, 0.0f);              // frame 1 restarts accumulation (Renderer.cpp:55-56)
        if (p.frame_first != 1u) { float4 a = p.accum[pixel]; acc = f3(a.x, a.y, a.z); }

        int32_t first_id = -1;
        for (uint32_t f = 0; f < p.n_frames; ++f) {
            const uint32_t frame = p.frame_first + f;
            int32_t id;
            F3 color = per_pixel<MODE, STATS>(p, camPos, dir, pixel_seed(pixel, frame, p.seed), id, n_rays, n_nodes, n_tris);
            if (f == 0) first_id = id;
            acc = acc + color;                      // m_AccumulationBuffer[pixelIndex] += pixelColor
        }
        p.accum[pixel] = make_float4(acc.x, acc.y, acc.z, 0.0f);
        p.image[pixel] = resolve_px(acc, (float)(p.frame_first + p.n_frames - 1u), p.use_ace);
        if (p.primary_ids) p.primary_ids[pixel] = first_id;
        if (p.out_dirs) {
            p.out_dirs[3 * (size_t)pixel] = dir.x; p.out_dirs[3 * (size_t)pixel + 1] = dir.y; p.out_dirs[3 * (size_t)pixel + 2] = dir.z;
        }
    }

    // work counters: one slot group per workgroup, summed on the host when read
    if (p.counters) {
        unsigned long long* slot = p.counters + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        uint32_t r = n_rays, n = n_nodes, t = n_tris;
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_down(r, off); n += __shfl_down(n, off); t += __shfl_down(t, off);
        }
        if (lane == 0) {
            atomicAdd(&slot[0], (unsigned long long)r);
            if (STATS) { atomicAdd(&slot[1], (unsigned long long)n); atomicAdd(&slot[2], (unsigned long long)t); }
            if (MODE != RT_MODE_MESH) atomicAdd(&slot[3], (unsigned long long)r * p.n_spheres);
        }
    }
}

// Resolve only (Renderer.cpp:68-79): image = tonemap(accum / frame_index)
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
            const int s = xo + yo;
            if ((s < 0 ? -s : s) > 1 || xx < 0 || xx >= (int)p.width || yy < 0 || yy >= (int)p.height) continue;
            const float wgt = 1.0f - ((float)((yo < 0 ? -yo : yo) + (xo < 0 ? -xo : xo)) * 0.5f);
            float4 a = p.accum[(uint32_t)yy * p.width + (uint32_t)xx];
            acc = acc + f3(a.x, a.y, a.z) * wgt;
            weight += wgt;
        }
    if (weight == 0.0f) return;
    p.image[y * p.width + x] = resolve_px(acc, weight * (float)frame_index, p.use_ace);
}

// ---- launchers (called from rt_api.cpp) -----------------------------------------------------------------------------
template <int MODE>
static void launch_mode(const RtParams& p, dim3 grid, bool stats, hipStream_t s)
{
    if (stats) hipLaunchKernelGGL((rt_trace_kernel<MODE, true>), grid, dim3(256), 0, s, p);
    else       hipLaunchKernelGGL((rt_trace_kernel<MODE, false>), grid, dim3(256), 0, s, p);
}

extern "C" hipError_t rt_launch_trace(const RtParams* p, int mode, int stats, unsigned gx, unsigned gy, hipStream_t s)
{
    dim3 grid(gx, gy);
    switch (mode) {
    case RT_MODE_MESH:    launch_mode<RT_MODE_MESH>(*p, grid, stats != 0, s); break;
    case RT_MODE_SPHERES: launch_mode<RT_MODE_SPHERES>(*p, grid, stats != 0, s); break;
    default:              launch_mode<RT_MODE_BOTH>(*p, grid, stats != 0, s); break;
    }
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_resolve(const RtParams* p, unsigned frame_index, unsigned gx, unsigned gy, hipStream_t s)
{
    hipLaunchKernelGGL(rt_resolve_kernel, dim3(gx, gy), dim3(256), 0, s, *p, frame_index);
    return hipGetLastError();
}

extern "C" hipError_t rt_launch_aa(const RtParams* p, unsigned frame_index, unsigned gx, unsigned gy, hipStream_t s)
{
    hipLaunchKernelGGL(rt_aa_kernel, dim3(gx, gy), dim3(256), 0, s, *p, frame_index);
    return hipGetLastError();
}
