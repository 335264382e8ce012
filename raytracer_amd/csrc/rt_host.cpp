// rt_host.cpp -- the parts of the path that stay on the host CPU: camera matrices and the KD build + flattening.
//
// Camera: Camera::RecalculateProjection / RecalculateView (RayTracer/src/Camera.cpp:122-132) with glm 0.9.9.9's
// perspectiveFovRH_NO, lookAtRH and compute_inverse<4,4> evaluated in glm's operation order, so the matrices handed
// to the device are the ones the reference's Camera would hold.
//
// KD build: the decisions of KDTreeCPU::constructTreeStackless (RayTracer/src/KDAccel/KDTreeCPU.cpp:424-632) -- three
// axes x the split_delta plane sweep, cost = 1.5 + areaL*nL + areaR*nR, strict '<' keeps the first minimum, children
// only when non-empty, stop at <= 10 triangles or depth 40 -- reached by binned counting: the sweep planes of one axis
// are non-decreasing, so each triangle's first plane above its min / above its max is found by binary search and the
// per-plane counts are prefix sums.  Same float comparisons against the same plane values, hence the same tree.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/rt_mi355x.h"

namespace {

// ---- glm-order helpers ------------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float dot(V3 a, V3 b) { float x = a.x * b.x, y = a.y * b.y, z = a.z * b.z; return (x + y) + z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 normalize(V3 v) { float s = 1.0f / std::sqrt(dot(v, v)); return {v.x * s, v.y * s, v.z * s}; }

inline float& at(float* m, int col, int row) { return m[col * 4 + row]; }
inline float at(const float* m, int col, int row) { return m[col * 4 + row]; }

// glm::inverse(mat4): cofactor expansion exactly as detail/func_matrix.inl:347-405 orders it
void inverse4(const float* m, float* out)
{
    const float c00 = at(m,2,2) * at(m,3,3) - at(m,3,2) * at(m,2,3);
    const float c02 = at(m,1,2) * at(m,3,3) - at(m,3,2) * at(m,1,3);
    const float c03 = at(m,1,2) * at(m,2,3) - at(m,2,2) * at(m,1,3);
    const float c04 = at(m,2,1) * at(m,3,3) - at(m,3,1) * at(m,2,3);
    const float c06 = at(m,1,1) * at(m,3,3) - at(m,3,1) * at(m,1,3);
    const float c07 = at(m,1,1) * at(m,2,3) - at(m,2,1) * at(m,1,3);
    const float c08 = at(m,2,1) * at(m,3,2) - at(m,3,1) * at(m,2,2);
    const float c10 = at(m,1,1) * at(m,3,2) - at(m,3,1) * at(m,1,2);
    const float c11 = at(m,1,1) * at(m,2,2) - at(m,2,1) * at(m,1,2);
    const float c12 = at(m,2,0) * at(m,3,3) - at(m,3,0) * at(m,2,3);
    const float c14 = at(m,1,0) * at(m,3,3) - at(m,3,0) * at(m,1,3);
    const float c15 = at(m,1,0) * at(m,2,3) - at(m,2,0) * at(m,1,3);
    const float c16 = at(m,2,0) * at(m,3,2) - at(m,3,0) * at(m,2,2);
    const float c18 = at(m,1,0) * at(m,3,2) - at(m,3,0) * at(m,1,2);
    const float c19 = at(m,1,0) * at(m,2,2) - at(m,2,0) * at(m,1,2);
    const float c20 = at(m,2,0) * at(m,3,1) - at(m,3,0) * at(m,2,1);
    const float c22 = at(m,1,0) * at(m,3,1) - at(m,3,0) * at(m,1,1);
    const float c23 = at(m,1,0) * at(m,2,1) - at(m,2,0) * at(m,1,1);
    const float fac[6][4] = {{c00, c00, c02, c03}, {c04, c04, c06, c07}, {c08, c08, c10, c11},
                             {c12, c12, c14, c15}, {c16, c16, c18, c19}, {c20, c20, c22, c23}};
    const float vec[4][4] = {{at(m,1,0), at(m,0,0), at(m,0,0), at(m,0,0)}, {at(m,1,1), at(m,0,1), at(m,0,1), at(m,0,1)},
                             {at(m,1,2), at(m,0,2), at(m,0,2), at(m,0,2)}, {at(m,1,3), at(m,0,3), at(m,0,3), at(m,0,3)}};
    float inv[16];
    for (int r = 0; r < 4; ++r) {
        const float sa = (r & 1) ? -1.0f : 1.0f, sb = -sa;
        at(inv,0,r) = ((vec[1][r] * fac[0][r] - vec[2][r] * fac[1][r]) + vec[3][r] * fac[2][r]) * sa;
        at(inv,1,r) = ((vec[0][r] * fac[0][r] - vec[2][r] * fac[3][r]) + vec[3][r] * fac[4][r]) * sb;
        at(inv,2,r) = ((vec[0][r] * fac[1][r] - vec[1][r] * fac[3][r]) + vec[3][r] * fac[5][r]) * sa;
        at(inv,3,r) = ((vec[0][r] * fac[2][r] - vec[1][r] * fac[4][r]) + vec[2][r] * fac[5][r]) * sb;
    }
    const float d0 = at(m,0,0) * at(inv,0,0), d1 = at(m,0,1) * at(inv,1,0);
    const float d2 = at(m,0,2) * at(inv,2,0), d3 = at(m,0,3) * at(inv,3,0);
    const float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
    for (int i = 0; i < 16; ++i) out[i] = inv[i] * one_over_det;
}

// ---- KD build -----------------------------------------------------------------------------------------------------
constexpr int kLeafTris = 10;      // NUM_TRIS_PER_NODE (KDTreeCPU.h:14)
constexpr int kMaxDepth = 40;      // MAX_DEPTH (KDTreeCPU.h:15)

struct BuildNode {
    float c[3], e[3];
    int32_t left = -1, right = -1;
    int32_t depth = 0;
    std::vector<int32_t> tris;     // released once the node has been split
};

struct Builder {
    const float* verts; const uint32_t* tri_idx; uint32_t ntris;
    std::vector<float> tmin[3], tmax[3];       // per-triangle axis extents (KDTreeCPU.cpp:98-132)
    std::vector<float> deltas;                 // the sweep's current_delta sequence (KDTreeCPU.cpp:478)
    std::deque<BuildNode> nodes;               // stable addresses; guarded by mu for growth
    std::vector<int32_t> work;
    size_t pending = 0;
    uint32_t max_nodes = 0;
    bool overflow = false;
    std::mutex mu;
    std::condition_variable cv;

    void split(int32_t id);
    void worker();
};

void Builder::split(int32_t id)
{
    BuildNode* node;
    { std::lock_guard<std::mutex> lk(mu); node = &nodes[(size_t)id]; }
    const int n = (int)node->tris.size();
    if (n <= kLeafTris || node->depth >= kMaxDepth) return;

    const int K = (int)deltas.size();
    // getLongestBoundingBoxSide (KDTreeCPU.cpp:93-96): the initial choice, replaced by the first finite cost
    int best_axis = (node->e[0] > node->e[1] && node->e[0] > node->e[2]) ? 0 : (node->e[1] > node->e[2] ? 1 : 2);
    float best_cost = FLT_MAX, best_plane = 0.0f;
    std::vector<float> planes((size_t)K);
    std::vector<uint32_t> histL((size_t)K + 1), histR((size_t)K + 1);

    for (int axis = 0; axis < 3; ++axis) {
        const float min_s = node->c[axis] - node->e[axis];
        const float max_s = node->c[axis] + node->e[axis];
        const float s_length = max_s - min_s;
        bool monotone = true;
        for (int k = 0; k < K; ++k) {
            planes[(size_t)k] = min_s + s_length * deltas[(size_t)k];
            if (k && planes[(size_t)k] < planes[(size_t)k - 1]) monotone = false;
        }
        std::fill(histL.begin(), histL.end(), 0u);
        std::fill(histR.begin(), histR.end(), 0u);
        const float* mn = tmin[axis].data();
        const float* mx = tmax[axis].data();
        if (monotone) {
            for (int i = 0; i < n; ++i) {
                const int32_t t = node->tris[(size_t)i];
                // first plane strictly above the value: planes from there on count the triangle on the left
                // (min < plane); planes before it count it on the right (max >= plane)
                histL[(size_t)(std::upper_bound(planes.begin(), planes.end(), mn[t]) - planes.begin())]++;
                histR[(size_t)(std::upper_bound(planes.begin(), planes.end(), mx[t]) - planes.begin())]++;
            }
        }
        uint32_t accL = 0, accR = 0;
        for (int k = 0; k < K; ++k) {
            const float plane = planes[(size_t)k];
            uint32_t count_left, count_right;
            if (monotone) {
                accL += histL[(size_t)k]; accR += histR[(size_t)k];
                count_left = accL; count_right = (uint32_t)n - accR;
            } else {                                   // never seen; keeps the result right if it ever is
                count_left = count_right = 0;
                for (int i = 0; i < n; ++i) {
                    const int32_t t = node->tris[(size_t)i];
                    if (mn[t] < plane) count_left++;
                    if (mx[t] >= plane) count_right++;
                }
            }
            float le[3] = {node->e[0], node->e[1], node->e[2]}, re[3] = {node->e[0], node->e[1], node->e[2]};
            le[axis] = (plane - min_s) * 0.5f;
            re[axis] = (max_s - plane) * 0.5f;
            const float area_left = 8.0f * ((le[0] * le[1] + le[1] * le[2]) + le[0] * le[2]);
            const float area_right = 8.0f * ((re[0] * re[1] + re[1] * re[2]) + re[0] * re[2]);
            const float cost = (1.5f + area_left * (float)count_left * 1.0f) + area_right * (float)count_right * 1.0f;
            if (cost < best_cost) { best_cost = cost; best_plane = plane; best_axis = axis; }
        }
    }

    const int ax = best_axis;
    const float min_before = node->c[ax] - node->e[ax];
    const float max_before = node->c[ax] + node->e[ax];
    const float left_extend = (best_plane - min_before) * 0.5f;
    const float left_center = best_plane - left_extend;
    const float right_extend = (max_before - best_plane) * 0.5f;
    const float right_center = best_plane + right_extend;

    std::vector<int32_t> lt, rt;
    lt.reserve((size_t)n); rt.reserve((size_t)n);
    for (int i = 0; i < n; ++i) {
        const int32_t t = node->tris[(size_t)i];
        if (tmin[ax][(size_t)t] < best_plane) lt.push_back(t);
        if (tmax[ax][(size_t)t] >= best_plane) rt.push_back(t);
    }
    lt.shrink_to_fit(); rt.shrink_to_fit();

    BuildNode proto;
    memcpy(proto.c, node->c, sizeof proto.c); memcpy(proto.e, node->e, sizeof proto.e);
    proto.depth = node->depth + 1;
    std::vector<int32_t>().swap(node->tris);

    std::lock_guard<std::mutex> lk(mu);
    if (max_nodes && nodes.size() + 2 > max_nodes) { overflow = true; return; }
    if (!lt.empty()) {
        nodes.push_back(proto);
        BuildNode& ch = nodes.back();
        ch.c[ax] = left_center; ch.e[ax] = left_extend; ch.tris = std::move(lt);
        nodes[(size_t)id].left = (int32_t)nodes.size() - 1;
        work.push_back((int32_t)nodes.size() - 1); pending++;
    }
    if (!rt.empty()) {
        nodes.push_back(proto);
        BuildNode& ch = nodes.back();
        ch.c[ax] = right_center; ch.e[ax] = right_extend; ch.tris = std::move(rt);
        nodes[(size_t)id].right = (int32_t)nodes.size() - 1;
        work.push_back((int32_t)nodes.size() - 1); pending++;
    }
}

void Builder::worker()
{
    for (;;) {
        int32_t id;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !work.empty() || pending == 0 || overflow; });
            if (overflow || (work.empty() && pending == 0)) { cv.notify_all(); return; }
            id = work.back(); work.pop_back();
        }
        split(id);
        {
            std::lock_guard<std::mutex> lk(mu);
            pending--;
        }
        cv.notify_all();
    }
}

}  // namespace

struct rt_host_kd {
    std::vector<float> center, extent;
    std::vector<int32_t> left, right, first, count, leaf_idx;
    uint32_t depth = 0;
};

extern "C" {

void rt_host_camera_default_forward(float fwd[3])
{
    // Camera.cpp:14: glm::normalize(glm::vec3(0, -0.1, -1)) -- double literals narrowed to float first
    V3 f = normalize({(float)0, (float)-0.1, (float)-1});
    fwd[0] = f.x; fwd[1] = f.y; fwd[2] = f.z;
}

void rt_host_camera_matrices(float fov_deg, float znear, float zfar, uint32_t width, uint32_t height, const float pos[3],
                             const float fwd[3], float proj[16], float inv_proj[16], float view[16], float inv_view[16])
{
    // perspectiveFovRH_NO(radians(fov), w, h, near, far) (ext/matrix_clip_space.inl:372-389)
    const float w = (float)width, h = (float)height;
    const float rad = fov_deg * (float)0.01745329251994329576923690768489;
    const float hh = std::cos(0.5f * rad) / std::sin(0.5f * rad);
    const float ww = hh * h / w;
    memset(proj, 0, 16 * sizeof(float));
    at(proj,0,0) = ww;
    at(proj,1,1) = hh;
    at(proj,2,2) = -(zfar + znear) / (zfar - znear);
    at(proj,2,3) = -1.0f;
    at(proj,3,2) = -(2.0f * zfar * znear) / (zfar - znear);
    inverse4(proj, inv_proj);

    // lookAtRH(pos, pos + fwd, (0,1,0)) (ext/matrix_transform.inl:99-119)
    const V3 eye{pos[0], pos[1], pos[2]};
    const V3 center = add(eye, {fwd[0], fwd[1], fwd[2]});
    const V3 f = normalize(sub(center, eye));
    const V3 s = normalize(cross(f, {0.0f, 1.0f, 0.0f}));
    const V3 u = cross(s, f);
    memset(view, 0, 16 * sizeof(float));
    at(view,0,0) = s.x;  at(view,1,0) = s.y;  at(view,2,0) = s.z;
    at(view,0,1) = u.x;  at(view,1,1) = u.y;  at(view,2,1) = u.z;
    at(view,0,2) = -f.x; at(view,1,2) = -f.y; at(view,2,2) = -f.z;
    at(view,3,0) = -dot(s, eye);
    at(view,3,1) = -dot(u, eye);
    at(view,3,2) = dot(f, eye);
    at(view,3,3) = 1.0f;
    inverse4(view, inv_view);
}

rt_host_kd* rt_host_kd_build(const float* verts, uint32_t nverts, const uint32_t* tri_idx, uint32_t ntris,
                             uint32_t max_nodes, int nthreads)
{
    if (!verts || !tri_idx || !nverts || !ntris) return nullptr;
    for (size_t i = 0; i < 3 * (size_t)ntris; ++i) if (tri_idx[i] >= nverts) return nullptr;

    Builder b;
    b.verts = verts; b.tri_idx = tri_idx; b.ntris = ntris; b.max_nodes = max_nodes;
    for (float d = 0.01f; d < 1.0f; d += 0.01f) b.deltas.push_back(d);

    // getMinTriValue / getMaxTriValue: the vertex indices pass through a float vector (uint -> float -> int), so an
    // index above 2^24 reads a neighbouring vertex -- kept as is (it only matters for unindexed meshes that large).
    for (int a = 0; a < 3; ++a) { b.tmin[a].resize(ntris); b.tmax[a].resize(ntris); }
    for (uint32_t t = 0; t < ntris; ++t) {
        size_t i0 = (size_t)(int)(float)tri_idx[3 * (size_t)t], i1 = (size_t)(int)(float)tri_idx[3 * (size_t)t + 1],
               i2 = (size_t)(int)(float)tri_idx[3 * (size_t)t + 2];
        if (i0 >= nverts) i0 = nverts - 1;
        if (i1 >= nverts) i1 = nverts - 1;
        if (i2 >= nverts) i2 = nverts - 1;
        for (int a = 0; a < 3; ++a) {
            const float v0 = verts[3 * i0 + a], v1 = verts[3 * i1 + a], v2 = verts[3 * i2 + a];
            b.tmin[a][t] = (v0 < v1 && v0 < v2) ? v0 : (v1 < v2 ? v1 : v2);
            b.tmax[a][t] = (v0 > v1 && v0 > v2) ? v0 : (v1 > v2 ? v1 : v2);
        }
    }

    // root box over all vertices (KDTreeCPU.cpp:159-191)
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = 0; i < nverts; ++i)
        for (int a = 0; a < 3; ++a) {
            const float v = verts[3 * (size_t)i + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
    b.nodes.emplace_back();
    BuildNode& root = b.nodes.back();
    for (int a = 0; a < 3; ++a) { root.c[a] = 0.5f * (mn[a] + mx[a]); root.e[a] = 0.5f * (mx[a] - mn[a]); }
    root.tris.resize(ntris);
    for (uint32_t t = 0; t < ntris; ++t) root.tris[t] = (int32_t)t;

    b.work.push_back(0); b.pending = 1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    std::vector<std::thread> pool;
    for (int i = 1; i < nthreads; ++i) pool.emplace_back([&b] { b.worker(); });
    b.worker();
    for (auto& th : pool) th.join();
    if (b.overflow) return nullptr;

    // flatten in preorder, left before right
    rt_host_kd* kd = new rt_host_kd();
    const size_t N = b.nodes.size();
    kd->center.resize(3 * N); kd->extent.resize(3 * N);
    kd->left.assign(N, -1); kd->right.assign(N, -1); kd->first.assign(N, 0); kd->count.assign(N, 0);
    std::vector<int32_t> order; order.reserve(N);
    std::vector<int32_t> renum(N, -1), stack{0};
    while (!stack.empty()) {
        int32_t id = stack.back(); stack.pop_back();
        renum[(size_t)id] = (int32_t)order.size(); order.push_back(id);
        const BuildNode& n = b.nodes[(size_t)id];
        if (n.right >= 0) stack.push_back(n.right);
        if (n.left >= 0) stack.push_back(n.left);
    }
    for (size_t k = 0; k < N; ++k) {
        const BuildNode& n = b.nodes[(size_t)order[k]];
        memcpy(&kd->center[3 * k], n.c, 12); memcpy(&kd->extent[3 * k], n.e, 12);
        kd->left[k] = n.left >= 0 ? renum[(size_t)n.left] : -1;
        kd->right[k] = n.right >= 0 ? renum[(size_t)n.right] : -1;
        if ((uint32_t)n.depth > kd->depth) kd->depth = (uint32_t)n.depth;
        if (n.left < 0 && n.right < 0) {
            kd->first[k] = (int32_t)kd->leaf_idx.size();
            kd->count[k] = (int32_t)n.tris.size();
            kd->leaf_idx.insert(kd->leaf_idx.end(), n.tris.begin(), n.tris.end());
        }
    }
    return kd;
}

int rt_host_kd_flat(const rt_host_kd* kd, rt_kd_flat* flat, uint32_t* depth)
{
    if (!kd || !flat) return RT_E_ARG;
    flat->n_nodes = (uint32_t)kd->left.size();
    flat->center = kd->center.data(); flat->extent = kd->extent.data();
    flat->left = kd->left.data(); flat->right = kd->right.data();
    flat->first = kd->first.data(); flat->count = kd->count.data();
    flat->n_leaf_idx = (uint32_t)kd->leaf_idx.size(); flat->leaf_idx = kd->leaf_idx.data();
    if (depth) *depth = kd->depth;
    return RT_OK;
}

void rt_host_kd_free(rt_host_kd* kd) { delete kd; }

}  // extern "C"
