// ref_harness.cpp -- thin C entry points into the REFERENCE'S OWN objects, for pinning the oracle.
//
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile into oracle/_ref/libref.so together with the
// unmodified reference sources KDAccel/{KDTreeCPU,Intersections,KDTreeStructs}.cpp, Camera.cpp and
// Walnut/Random.cpp (+ header-only glm and pcg32 as vendored by the reference).  Nothing here
// re-implements the reference: every function forwards to a reference class or a glm function.
// It exists only in the build container (the reference does not travel); tests that need it skip
// when the library is absent and otherwise compare it with oracle/rt_oracle.c.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>

#include "KDAccel/KDTreeCPU.h"
#include "KDAccel/Intersections.h"
#include "Camera.h"
#include "Ray.h"
#include "Walnut/Random.h"

#define API extern "C" __attribute__((visibility("default")))

// ---- KDTreeCPU ---------------------------------------------------------------------------------
API void* ref_kd_build(int ntris, const uint32_t* tris, int nverts, const float* verts)
{
    std::vector<glm::uvec3> t(ntris);
    std::vector<glm::vec3> v(nverts);
    for (int i = 0; i < ntris; ++i) t[i] = glm::uvec3(tris[3 * i], tris[3 * i + 1], tris[3 * i + 2]);
    for (int i = 0; i < nverts; ++i) v[i] = glm::vec3(verts[3 * i], verts[3 * i + 1], verts[3 * i + 2]);
    return new KDTreeCPU(ntris, t.data(), nverts, v.data());
}

API void ref_kd_free(void* h) { delete static_cast<KDTreeCPU*>(h); }

static void walk_sizes(const KDTreeNode* n, uint32_t& nodes, uint32_t& idx, uint32_t& depth)
{
    nodes++; idx += (uint32_t)n->num_tris;
    if ((uint32_t)n->id > depth) depth = (uint32_t)n->id;
    if (n->left) walk_sizes(n->left, nodes, idx, depth);
    if (n->right) walk_sizes(n->right, nodes, idx, depth);
}

API void ref_kd_sizes(void* h, uint32_t* n_nodes, uint32_t* n_idx, uint32_t* depth)
{
    *n_nodes = 0; *n_idx = 0; *depth = 0;
    walk_sizes(static_cast<KDTreeCPU*>(h)->getRootNode(), *n_nodes, *n_idx, *depth);
}

struct Out { float* c; float* e; int32_t* l; int32_t* r; int32_t* first; int32_t* count; int32_t* idx; uint32_t nn, ni; };

static int32_t walk_export(const KDTreeNode* n, Out& o)
{
    int32_t me = (int32_t)o.nn++;
    for (int k = 0; k < 3; ++k) { o.c[3 * me + k] = n->bbox.center[k]; o.e[3 * me + k] = n->bbox.extends[k]; }
    o.first[me] = (int32_t)o.ni; o.count[me] = n->num_tris;
    for (int i = 0; i < n->num_tris; ++i) o.idx[o.ni++] = n->tri_indices[i];
    int32_t l = n->left ? walk_export(n->left, o) : -1;
    int32_t r = n->right ? walk_export(n->right, o) : -1;
    o.l[me] = l; o.r[me] = r;
    return me;
}

// Preorder (left before right) dump of the pointer tree; every node's own triangle list is included.
API void ref_kd_export(void* h, float* center3, float* extent3, int32_t* left, int32_t* right, int32_t* first,
                       int32_t* count, int32_t* leaf_idx)
{
    Out o{ center3, extent3, left, right, first, count, leaf_idx, 0, 0 };
    walk_export(static_cast<KDTreeCPU*>(h)->getRootNode(), o);
}

// KDTreeCPU::intersect for n rays sharing one origin; ray set up as Renderer.cpp:139-151 does.
API void ref_kd_intersect(void* h, const float* origin, const float* dirs3, uint32_t n, int32_t* ids, float* t,
                          float* u, float* v)
{
    const KDTreeCPU* kd = static_cast<KDTreeCPU*>(h);
    for (uint32_t i = 0; i < n; ++i) {
        Ray ray;
        ray.Origin = glm::vec3(origin[0], origin[1], origin[2]);
        ray.Direction = glm::vec3(dirs3[3 * i], dirs3[3 * i + 1], dirs3[3 * i + 2]);
        ray.DirectionInverse = glm::vec3(1.0f / ray.Direction.x, 1.0f / ray.Direction.y, 1.0f / ray.Direction.z);
        float tt = 0.0f, uu = 0.0f, vv = 0.0f; uint32_t idx = 0;
        bool hit = kd->intersect(&ray, tt, idx, uu, vv);
        ids[i] = hit ? (int32_t)(0x40000000u | idx) : -1;
        t[i] = tt; u[i] = uu; v[i] = vv;
    }
}

// Same, but each ray has its own origin (bounce rays).
API void ref_kd_intersect_rays(void* h, const float* origins3, const float* dirs3, uint32_t n, int32_t* ids,
                               float* t, float* u, float* v)
{
    for (uint32_t i = 0; i < n; ++i)
        ref_kd_intersect(h, origins3 + 3 * i, dirs3 + 3 * i, 1, ids + i, t + i, u + i, v + i);
}

// ---- Intersections -----------------------------------------------------------------------------
API int ref_aabb(const float* center, const float* extent, const float* o, const float* inv_d, float* t_near)
{
    boundingBox b; b.center = glm::vec3(center[0], center[1], center[2]); b.extends = glm::vec3(extent[0], extent[1], extent[2]);
    Ray ray; ray.Origin = glm::vec3(o[0], o[1], o[2]); ray.Direction = glm::vec3(0.0f);
    ray.DirectionInverse = glm::vec3(inv_d[0], inv_d[1], inv_d[2]);
    return Intersections::aabbIntersect(b, &ray, *t_near) ? 1 : 0;
}

API int ref_tri(const float* o, const float* d, const float* v0, const float* v1, const float* v2, float* t, float* u, float* v)
{
    Ray ray; ray.Origin = glm::vec3(o[0], o[1], o[2]); ray.Direction = glm::vec3(d[0], d[1], d[2]);
    ray.DirectionInverse = glm::vec3(0.0f);
    return Intersections::triIntersect(&ray, glm::vec3(v0[0], v0[1], v0[2]), glm::vec3(v1[0], v1[1], v1[2]),
                                       glm::vec3(v2[0], v2[1], v2[2]), *t, *u, *v) ? 1 : 0;
}

// ---- Walnut::Random / pcg32 --------------------------------------------------------------------
// Walnut::Random's engine is thread_local and private; a fresh thread sees a default-constructed one.
API void ref_random_floats(uint32_t n, float* out)
{
    std::thread th([=] { for (uint32_t i = 0; i < n; ++i) out[i] = Walnut::Random::Float(); });
    th.join();
}

API void ref_pcg32_u32(uint32_t n, uint32_t* out)
{
    pcg32 rng;
    for (uint32_t i = 0; i < n; ++i) out[i] = rng();
}

API void ref_pcg32_seeded_u32(uint64_t seed, uint32_t n, uint32_t* out)
{
    pcg32 rng(seed);
    for (uint32_t i = 0; i < n; ++i) out[i] = rng();
}

// ---- Camera ------------------------------------------------------------------------------------
// The reference's Camera with its constructor defaults (position (0,1,2), forward normalize(0,-0.1,-1)).
API void ref_camera(float fov, float znear, float zfar, uint32_t w, uint32_t h, float* pos3, float* fwd3,
                    float* proj16, float* inv_proj16, float* view16, float* inv_view16, float* dirs3)
{
    Camera cam(fov, znear, zfar);
    cam.OnResize(w, h);
    std::memcpy(pos3, &cam.GetPosition()[0], 12);
    std::memcpy(fwd3, &cam.GetDirection()[0], 12);
    std::memcpy(proj16, &cam.GetProjection()[0][0], 64);
    std::memcpy(inv_proj16, &cam.GetInverseProjection()[0][0], 64);
    std::memcpy(view16, &cam.GetView()[0][0], 64);
    std::memcpy(inv_view16, &cam.GetInverseView()[0][0], 64);
    if (dirs3) std::memcpy(dirs3, cam.GetRayDirections().data(), (size_t)w * h * 12);
}

// ---- glm primitives the Renderer calls (the real inlined glm code, for op-order pins) ------------
API void ref_glm_normalize(const float* a, float* o) { glm::vec3 r = glm::normalize(glm::vec3(a[0], a[1], a[2])); std::memcpy(o, &r[0], 12); }
API void ref_glm_reflect(const float* i, const float* n, float* o) { glm::vec3 r = glm::reflect(glm::vec3(i[0], i[1], i[2]), glm::vec3(n[0], n[1], n[2])); std::memcpy(o, &r[0], 12); }
API void ref_glm_mix(const float* x, const float* y, float a, float* o) { glm::vec3 r = glm::mix(glm::vec3(x[0], x[1], x[2]), glm::vec3(y[0], y[1], y[2]), a); std::memcpy(o, &r[0], 12); }
API float ref_glm_dot(const float* a, const float* b) { return glm::dot(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2])); }
API void ref_glm_cross(const float* a, const float* b, float* o) { glm::vec3 r = glm::cross(glm::vec3(a[0], a[1], a[2]), glm::vec3(b[0], b[1], b[2])); std::memcpy(o, &r[0], 12); }
API void ref_glm_inverse(const float* m16, float* o16) { glm::mat4 m; std::memcpy(&m[0][0], m16, 64); glm::mat4 r = glm::inverse(m); std::memcpy(o16, &r[0][0], 64); }
API void ref_glm_mat_vec(const float* m16, const float* v4, float* o4) { glm::mat4 m; std::memcpy(&m[0][0], m16, 64); glm::vec4 r = m * glm::vec4(v4[0], v4[1], v4[2], v4[3]); std::memcpy(o4, &r[0], 16); }
// glm::clamp / the element-wise forms used by ACESFilm + LinearToSRGB (Renderer.h:63-98), via glm itself.
API void ref_glm_aces_srgb(const float* c, float* o)
{
    glm::vec3 x(c[0], c[1], c[2]);
    glm::vec3 a = glm::clamp((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.0f, 1.0f);
    glm::vec3 rgb = glm::clamp(a, 0.0f, 1.0f);
    glm::vec3 lt((rgb.x < 0.0031308f) ? 1.0f : 0.0f, (rgb.y < 0.0031308f) ? 1.0f : 0.0f, (rgb.z < 0.0031308f) ? 1.0f : 0.0f);
    glm::vec3 r = glm::mix(glm::pow(rgb, glm::vec3(1.0f / 2.4f)) * 1.055f - 0.055f, rgb * 12.92f, lt);
    std::memcpy(o, &r[0], 12);
}
