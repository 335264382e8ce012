// rt_host.cpp -- the parts of the path that stay on the host CPU: camera matrices and the KD build + flattening.
//
// Camera: Camera::RecalculateProjection / RecalculateView (RayTracer/src/Camera.cpp:122-132) with glm 0.9.9.9's
// perspectiveFovRH_NO, lookAtRH and compute_inverse<4,4> evaluated in glm's operation order, so the matrices handed
// to the device are the ones the reference's Camera would hold.
//
// KD build: the decisions of KDTreeCPU::constructTreeStackless (RayTracer/src/KDAccel/KDTreeCPU.cpp:424-632) -- three
// axes x the split_delta plane sweep, cost = 1.5 + areaL*nL + areaR*nR, strict '<' keeps the first minimum, children
// only when non-empty, stop at <= 10 triangles or depth 40 -- reached by binned counting: the sweep planes of one axis
// are non-decreasing, so each triangle's first plane above its min / above its max is found by a guess from the planes'
// spacing corrected against the planes themselves, and the per-plane counts are prefix sums.  Same float comparisons against
// the same plane values, hence the same tree.  (Set RT_KD_TIMES to have the build print where its time goes.)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cfloat>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <system_error>
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
    int32_t size = 1;              // nodes of the subtree (with the node itself) and triangle references of its leaves: known for the
    int64_t refs = 0;              // nodes of a finished build_subtree, for the others after Builder::measure
    bool subtree_root = false;     // the node a build_subtree call started from
    std::vector<int32_t> tris;     // released once the node has been split
};

constexpr int kPlanes = 128;       // room for the sweep's current_delta sequence 0.01f, + 0.01f, ... while < 1 (KDTreeCPU.cpp:478): n_deltas members

// How the work is spread over the threads.  The tree's first levels hold a few nodes of millions of triangles each: those are
// split one at a time with the counting and the partition of a node cut into chunks for all threads.  Nodes below kWideNode
// triangles are tasks of a pool, and a task that finds a node of at most `local_limit` triangles builds that node's whole subtree
// on its own, in memory of its own, and hands the finished nodes over under one lock -- the lock, the queue and the wake-ups
// are then paid per subtree of thousands of nodes, not per node.  Which thread splits a node and in which order nodes are
// created does not matter: a split depends on the node alone, and the tree is renumbered in preorder at the end.
constexpr int kWideNode = 1 << 18;

struct Builder {
    const float* verts; const uint32_t* tri_idx; uint32_t ntris;
    std::vector<float> tmin[3], tmax[3];       // per-triangle axis extents (KDTreeCPU.cpp:98-132)
    float deltas[kPlanes]; int n_deltas = 0;
    std::deque<BuildNode> nodes;               // stable addresses; guarded by mu for growth
    std::vector<int32_t> work;
    size_t pending = 0;
    uint32_t max_nodes = 0;
    int local_limit = 4096;
    bool overflow = false;
    std::mutex mu;
    std::condition_variable cv;

    std::vector<int32_t> subtree_roots;        // (guarded by mu)
    bool split(const BuildNode& node, BuildNode& l, BuildNode& r, int threads) const;
    void measure(int32_t id);
    void build_subtree(int32_t id);
    void split_to_tasks(int32_t id, int threads);
    void worker();
};

// Runs f(chunk, first, last) over [0, n) in `threads` chunks (the caller's thread takes the last one).
template <typename F>
void chunked(int n, int threads, F f)
{
    if (threads <= 1) { f(0, 0, n); return; }
    std::vector<std::thread> helpers;
    const int per = (n + threads - 1) / threads;
    for (int c = 0; c + 1 < threads; ++c) {
        const int first = std::min(n, c * per), last = std::min(n, (c + 1) * per);
        try { helpers.emplace_back([=] { f(c, first, last); }); }
        catch (const std::system_error&) { f(c, first, last); }         // no thread to be had: the chunk is done here
    }
    f(threads - 1, std::min(n, (threads - 1) * per), n);
    for (auto& h : helpers) h.join();
}

// One step of KDTreeCPU::constructTreeStackless for `node`: false if it stays a leaf, else the two children (boxes, depth and
// triangle lists; a child without triangles has an empty list and is not created, KDTreeCPU.cpp:563-610).
bool Builder::split(const BuildNode& node, BuildNode& l, BuildNode& r, int threads) const
{
    const int n = (int)node.tris.size();
    if (n <= kLeafTris || node.depth >= kMaxDepth) return false;
    if (n < kWideNode) threads = 1;
    const int32_t* tris = node.tris.data();

    const int K = n_deltas;
    // getLongestBoundingBoxSide (KDTreeCPU.cpp:93-96): the initial choice, replaced by the first finite cost
    int best_axis = (node.e[0] > node.e[1] && node.e[0] > node.e[2]) ? 0 : (node.e[1] > node.e[2] ? 1 : 2);
    float best_cost = FLT_MAX, best_plane = 0.0f;
    float planes[kPlanes];
    uint32_t hist_one[2 * (kPlanes + 1)];               // per chunk: counts by first plane above the minimum, then above the maximum
    std::vector<uint32_t> hist_wide;
    if (threads > 1) hist_wide.resize((size_t)threads * 2u * (K + 1));
    uint32_t* const hist = threads > 1 ? hist_wide.data() : hist_one;
    const size_t hist_words = (size_t)threads * 2u * (K + 1);

    for (int axis = 0; axis < 3; ++axis) {
        const float min_s = node.c[axis] - node.e[axis];
        const float max_s = node.c[axis] + node.e[axis];
        const float s_length = max_s - min_s;
        bool monotone = true;
        for (int k = 0; k < K; ++k) {
            planes[k] = min_s + s_length * deltas[k];
            if (!(planes[k] >= planes[k ? k - 1 : 0])) monotone = false;        // (a NaN plane counts as out of order)
        }
        const float* mn = tmin[axis].data();
        const float* mx = tmax[axis].data();
        uint32_t* histL = hist;
        uint32_t* histR = hist + (K + 1);
        if (monotone) {
            std::fill(hist, hist + hist_words, 0u);
            // The first plane strictly above a value: planes from there on count the triangle on the left (min < plane), planes
            // before it count it on the right (max >= plane).  The planes are nearly evenly spaced, so the place is guessed by
            // arithmetic and corrected by comparing with the planes themselves: the result is the exact upper bound.
            const float per_unit = s_length > 0.0f ? 100.0f / s_length : 0.0f;      // (plane k lies near min_s + (k + 1) / 100 of the length)
            auto above = [&](float v) -> int {
                const float g = (v - min_s) * per_unit;
                int k = g > 0.0f ? (g < (float)K ? (int)g : K) : 0;         // (a NaN goes to 0 and is corrected below like any guess)
                while (k < K && !(v < planes[k])) ++k;
                while (k > 0 && v < planes[k - 1]) --k;
                return k;
            };
            chunked(n, threads, [&](int chunk, int first, int last) {
                uint32_t* hl = hist + (size_t)chunk * 2u * (K + 1);
                uint32_t* hr = hl + (K + 1);
                for (int i = first; i < last; ++i) {
                    const int32_t t = tris[i];
                    hl[above(mn[t])]++;
                    hr[above(mx[t])]++;
                }
            });
            for (int c = 1; c < threads; ++c) {
                const uint32_t* hl = hist + (size_t)c * 2u * (K + 1);
                for (int k = 0; k < 2 * (K + 1); ++k) histL[k] += hl[k];
            }
        }
        // counts per plane, then the costs of all planes (independent of each other: the compiler makes vector code of that
        // loop; every cost is the same sequence of roundings as in KDTreeCPU.cpp:500-533), then the first minimum
        int32_t count_left[kPlanes], count_right[kPlanes];
        float cost[kPlanes];
        if (monotone) {
            uint32_t accL = 0, accR = 0;
            for (int k = 0; k < K; ++k) {
                accL += histL[k]; accR += histR[k];
                count_left[k] = (int32_t)accL; count_right[k] = (int32_t)((uint32_t)n - accR);
            }
        } else {                                       // never seen; keeps the result right if it ever is
            for (int k = 0; k < K; ++k) {
                const float plane = planes[k];
                int32_t cl = 0, cr = 0;
                for (int i = 0; i < n; ++i) {
                    const int32_t t = tris[i];
                    if (mn[t] < plane) cl++;
                    if (mx[t] >= plane) cr++;
                }
                count_left[k] = cl; count_right[k] = cr;
            }
        }
        const float e0 = node.e[0], e1 = node.e[1], e2 = node.e[2];
        for (int k = 0; k < K; ++k) {
            const float plane = planes[k];
            const float lx = (plane - min_s) * 0.5f, rx = (max_s - plane) * 0.5f;
            const float l0 = axis == 0 ? lx : e0, l1 = axis == 1 ? lx : e1, l2 = axis == 2 ? lx : e2;
            const float r0 = axis == 0 ? rx : e0, r1 = axis == 1 ? rx : e1, r2 = axis == 2 ? rx : e2;
            const float area_left = 8.0f * ((l0 * l1 + l1 * l2) + l0 * l2);
            const float area_right = 8.0f * ((r0 * r1 + r1 * r2) + r0 * r2);
            cost[k] = (1.5f + area_left * (float)count_left[k] * 1.0f) + area_right * (float)count_right[k] * 1.0f;
        }
        for (int k = 0; k < K; ++k)
            if (cost[k] < best_cost) { best_cost = cost[k]; best_plane = planes[k]; best_axis = axis; }
    }

    const int ax = best_axis;
    const float min_before = node.c[ax] - node.e[ax];
    const float max_before = node.c[ax] + node.e[ax];
    const float left_extend = (best_plane - min_before) * 0.5f;
    const float left_center = best_plane - left_extend;
    const float right_extend = (max_before - best_plane) * 0.5f;
    const float right_center = best_plane + right_extend;

    memcpy(l.c, node.c, sizeof l.c); memcpy(l.e, node.e, sizeof l.e);
    memcpy(r.c, node.c, sizeof r.c); memcpy(r.e, node.e, sizeof r.e);
    l.depth = r.depth = node.depth + 1;
    l.left = l.right = r.left = r.right = -1;
    l.c[ax] = left_center; l.e[ax] = left_extend;
    r.c[ax] = right_center; r.e[ax] = right_extend;
    const float* mn = tmin[ax].data();
    const float* mx = tmax[ax].data();
    if (threads <= 1) {
        size_t nl = 0, nr = 0;
        for (int i = 0; i < n; ++i) { const int32_t t = tris[i]; nl += mn[t] < best_plane; nr += mx[t] >= best_plane; }
        l.tris.resize(nl); r.tris.resize(nr);
        int32_t* pl = l.tris.data(); int32_t* pr = r.tris.data();
        for (int i = 0; i < n; ++i) {
            const int32_t t = tris[i];
            if (mn[t] < best_plane) *pl++ = t;
            if (mx[t] >= best_plane) *pr++ = t;
        }
    } else {                                           // chunks count, then write at their offsets: the lists keep the node's order
        std::vector<size_t> cl((size_t)threads + 1, 0), cr((size_t)threads + 1, 0);
        chunked(n, threads, [&](int chunk, int first, int last) {
            size_t a = 0, b = 0;
            for (int i = first; i < last; ++i) { const int32_t t = tris[i]; a += mn[t] < best_plane; b += mx[t] >= best_plane; }
            cl[(size_t)chunk + 1] = a; cr[(size_t)chunk + 1] = b;
        });
        for (int c = 0; c < threads; ++c) { cl[(size_t)c + 1] += cl[(size_t)c]; cr[(size_t)c + 1] += cr[(size_t)c]; }
        l.tris.resize(cl[(size_t)threads]); r.tris.resize(cr[(size_t)threads]);
        chunked(n, threads, [&](int chunk, int first, int last) {
            int32_t* pl = l.tris.data() + cl[(size_t)chunk]; int32_t* pr = r.tris.data() + cr[(size_t)chunk];
            for (int i = first; i < last; ++i) {
                const int32_t t = tris[i];
                if (mn[t] < best_plane) *pl++ = t;
                if (mx[t] >= best_plane) *pr++ = t;
            }
        });
    }
    return true;
}

// A node of the shared list is split once and its children become tasks.
void Builder::split_to_tasks(int32_t id, int threads)
{
    BuildNode* node;
    { std::lock_guard<std::mutex> lk(mu); node = &nodes[(size_t)id]; }
    BuildNode l, r;
    if (!split(*node, l, r, threads)) return;
    std::vector<int32_t>().swap(node->tris);
    std::lock_guard<std::mutex> lk(mu);
    if (max_nodes && nodes.size() + 2 > max_nodes) { overflow = true; return; }
    if (!l.tris.empty()) {
        nodes.push_back(std::move(l));
        nodes[(size_t)id].left = (int32_t)nodes.size() - 1;
        work.push_back((int32_t)nodes.size() - 1); pending++;
    }
    if (!r.tris.empty()) {
        nodes.push_back(std::move(r));
        nodes[(size_t)id].right = (int32_t)nodes.size() - 1;
        work.push_back((int32_t)nodes.size() - 1); pending++;
    }
}

// The whole subtree of a node of the shared list, built by the calling thread in a list of its own (links are indices into
// that list, 0 = the node itself) and appended to the shared list in one go.
void Builder::build_subtree(int32_t id)
{
    BuildNode* top;
    { std::lock_guard<std::mutex> lk(mu); top = &nodes[(size_t)id]; }
    std::vector<BuildNode> local(1);
    memcpy(local[0].c, top->c, sizeof top->c); memcpy(local[0].e, top->e, sizeof top->e);
    local[0].depth = top->depth;
    local[0].tris.swap(top->tris);
    std::vector<int32_t> stack{0};
    BuildNode l, r;
    while (!stack.empty()) {
        const int32_t at = stack.back(); stack.pop_back();
        if (!split(local[(size_t)at], l, r, 1)) continue;
        std::vector<int32_t>().swap(local[(size_t)at].tris);
        if (max_nodes && local.size() + 2 > max_nodes) { std::lock_guard<std::mutex> lk(mu); overflow = true; return; }
        if (!l.tris.empty()) {
            local[(size_t)at].left = (int32_t)local.size(); stack.push_back((int32_t)local.size());
            local.push_back(std::move(l)); l = BuildNode();
        }
        if (!r.tris.empty()) {
            local[(size_t)at].right = (int32_t)local.size(); stack.push_back((int32_t)local.size());
            local.push_back(std::move(r)); r = BuildNode();
        }
    }
    // children were created after their parents: one sweep from the back sums the subtrees up
    for (size_t i = local.size(); i-- > 0;) {
        BuildNode& nd = local[i];
        nd.size = 1; nd.refs = (nd.left < 0 && nd.right < 0) ? (int64_t)nd.tris.size() : 0;
        if (nd.left >= 0) { nd.size += local[(size_t)nd.left].size; nd.refs += local[(size_t)nd.left].refs; }
        if (nd.right >= 0) { nd.size += local[(size_t)nd.right].size; nd.refs += local[(size_t)nd.right].refs; }
    }
    std::lock_guard<std::mutex> lk(mu);
    if (max_nodes && nodes.size() + local.size() - 1 > max_nodes) { overflow = true; return; }
    const int32_t base = (int32_t)nodes.size() - 1;       // local index i >= 1 becomes base + i
    for (size_t i = 1; i < local.size(); ++i) {
        BuildNode& nd = local[i];
        if (nd.left >= 0) nd.left += base;
        if (nd.right >= 0) nd.right += base;
        nodes.push_back(std::move(nd));
    }
    BuildNode& shared = nodes[(size_t)id];
    shared.left = local[0].left >= 0 ? local[0].left + base : -1;
    shared.right = local[0].right >= 0 ? local[0].right + base : -1;
    shared.tris.swap(local[0].tris);                      // (its own list again if it stayed a leaf)
    shared.size = local[0].size; shared.refs = local[0].refs; shared.subtree_root = true;
    subtree_roots.push_back(id);
}

// Subtree sizes of the nodes above the build_subtree roots (after the build; a few thousand nodes at most).
void Builder::measure(int32_t id)
{
    BuildNode& nd = nodes[(size_t)id];
    if (nd.subtree_root) return;
    nd.size = 1; nd.refs = (nd.left < 0 && nd.right < 0) ? (int64_t)nd.tris.size() : 0;
    if (nd.left >= 0) { measure(nd.left); nd.size += nodes[(size_t)nd.left].size; nd.refs += nodes[(size_t)nd.left].refs; }
    if (nd.right >= 0) { measure(nd.right); nd.size += nodes[(size_t)nd.right].size; nd.refs += nodes[(size_t)nd.right].refs; }
}

void Builder::worker()
{
    for (;;) {
        int32_t id;
        size_t n;
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !work.empty() || pending == 0 || overflow; });
            if (overflow || (work.empty() && pending == 0)) { cv.notify_all(); return; }
            id = work.back(); work.pop_back();
            n = nodes[(size_t)id].tris.size();
        }
        if (n <= (size_t)local_limit) build_subtree(id);
        else split_to_tasks(id, 1);
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

    const bool times = getenv("RT_KD_TIMES") != nullptr;        // prints where the build's time goes
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (times) fprintf(stderr, "rt_host_kd_build: %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    };
    Builder b;
    b.verts = verts; b.tri_idx = tri_idx; b.ntris = ntris; b.max_nodes = max_nodes;
    for (float d = 0.01f; d < 1.0f && b.n_deltas < kPlanes; d += 0.01f) b.deltas[b.n_deltas++] = d;

    // getMinTriValue / getMaxTriValue: the vertex indices pass through a float vector (uint -> float -> int), so an
    // index above 2^24 reads a neighbouring vertex -- kept as is (it only matters for unindexed meshes that large).
    for (int a = 0; a < 3; ++a) { b.tmin[a].resize(ntris); b.tmax[a].resize(ntris); }
    chunked((int)ntris, nthreads < 1 ? 1 : (nthreads > 64 ? 64 : nthreads), [&](int, int first_tri, int last_tri) {
        for (uint32_t t = (uint32_t)first_tri; t < (uint32_t)last_tri; ++t) {
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
    });

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

    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    b.local_limit = (int)std::min<uint32_t>(65536u, std::max<uint32_t>(2048u, ntris / (32u * (uint32_t)nthreads)));
    // the few nodes of the first levels, one at a time with all threads; their smaller descendants wait in b.work for the pool
    lap("triangle extents and root box");
    b.work.push_back(0); b.pending = 1;
    for (size_t next = 0; next < b.work.size() && !b.overflow;) {
        const int32_t id = b.work[next];
        if (b.nodes[(size_t)id].tris.size() < (size_t)kWideNode) { ++next; continue; }
        b.work.erase(b.work.begin() + (std::ptrdiff_t)next);
        b.split_to_tasks(id, nthreads);
        b.pending--;
    }
    lap("wide nodes split");
    std::vector<std::thread> pool;
    for (int i = 1; i < nthreads; ++i) {
        try { pool.emplace_back([&b] { b.worker(); }); }
        catch (const std::system_error&) { break; }                     // fewer workers, the same tree
    }
    b.worker();
    for (auto& th : pool) th.join();
    if (b.overflow) return nullptr;
    lap("subtrees built");

    // Flatten in preorder, left before right, the leaves' triangle lists in the same order.  With the subtree sizes known, a node's
    // number and the place of its first triangle reference follow from its parent's: every build_subtree root is a task of its own.
    b.measure(0);
    rt_host_kd* kd = new rt_host_kd();
    const size_t N = b.nodes.size();
    if ((size_t)b.nodes[0].size != N) { delete kd; return nullptr; }        // (every node is reachable from the root)
    kd->center.resize(3 * N); kd->extent.resize(3 * N);
    kd->left.resize(N); kd->right.resize(N); kd->first.resize(N); kd->count.resize(N);
    kd->leaf_idx.resize((size_t)b.nodes[0].refs);
    struct Place { int32_t id, number; int64_t ref; };
    std::vector<Place> tasks;
    std::atomic<uint32_t> depth{0};
    auto emit_subtree = [&](Place top, bool whole) {                    // whole = descend below build_subtree roots too
        std::vector<Place> stack{top};
        uint32_t deepest = 0;
        while (!stack.empty()) {
            const Place at = stack.back(); stack.pop_back();
            BuildNode& n = b.nodes[(size_t)at.id];
            if (!whole && n.subtree_root && at.id != top.id) { tasks.push_back(at); continue; }
            const size_t k = (size_t)at.number;
            memcpy(&kd->center[3 * k], n.c, 12); memcpy(&kd->extent[3 * k], n.e, 12);
            const int32_t left_size = n.left >= 0 ? b.nodes[(size_t)n.left].size : 0;
            const int64_t left_refs = n.left >= 0 ? b.nodes[(size_t)n.left].refs : 0;
            kd->left[k] = n.left >= 0 ? at.number + 1 : -1;
            kd->right[k] = n.right >= 0 ? at.number + 1 + left_size : -1;
            kd->first[k] = 0; kd->count[k] = 0;
            if ((uint32_t)n.depth > deepest) deepest = (uint32_t)n.depth;
            if (n.left < 0 && n.right < 0) {
                kd->first[k] = (int32_t)at.ref;
                kd->count[k] = (int32_t)n.tris.size();
                if (!n.tris.empty()) memcpy(&kd->leaf_idx[(size_t)at.ref], n.tris.data(), n.tris.size() * sizeof(int32_t));
                std::vector<int32_t>().swap(n.tris);
            }
            if (n.right >= 0) stack.push_back({n.right, at.number + 1 + left_size, at.ref + left_refs});
            if (n.left >= 0) stack.push_back({n.left, at.number + 1, at.ref});
        }
        uint32_t seen = depth.load();
        while (deepest > seen && !depth.compare_exchange_weak(seen, deepest)) {}
    };
    if (b.nodes[0].subtree_root) tasks.push_back({0, 0, 0});
    else emit_subtree({0, 0, 0}, false);                                // the nodes above the tasks, by this thread
    {
        std::atomic<size_t> next{0};
        auto run = [&] { for (size_t i; (i = next.fetch_add(1)) < tasks.size();) emit_subtree(tasks[i], true); };
        std::vector<std::thread> helpers;
        for (int i = 1; i < nthreads; ++i) {
            try { helpers.emplace_back(run); }
            catch (const std::system_error&) { break; }
        }
        run();
        for (auto& h : helpers) h.join();
    }
    kd->depth = depth.load();
    lap("flattened");
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
