// rt_api.cpp -- context, HBM residency and the C ABI (include/rt_mi355x.h) over the gfx950 kernels.
//
// Plays the role of the reference's Renderer object (RayTracer/src/Renderer.{h,cpp}): owns the accumulation buffer and
// the RGBA32F frame, takes the Scene/Camera state per frame, and runs one Render per call.  The KD tree arrives as the
// flat host tree and is re-laid out here for the device (preorder + skip links, leaf-ordered pre-gathered triangles).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <chrono>
#include <thread>
#include <utility>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <system_error>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_device.h"

extern "C" hipError_t rt_launch_trace(const RtParams* p, int mode, int stats, int ordered, unsigned grid, hipStream_t s);
extern "C" hipError_t rt_trace_occupancy(int mode, int stats, int ordered, int* blocks_per_cu);
extern "C" unsigned rt_trace_block(int mode);
extern "C" hipError_t rt_launch_raygen(const RtParams* p, hipStream_t s);
extern "C" hipError_t rt_launch_column_collect(const RtParams* p, hipStream_t s);
extern "C" hipError_t rt_launch_column_prepass(const RtParams* p, int stats, hipStream_t s);
extern "C" hipError_t rt_launch_units_classify(const RtParams* p, hipStream_t s);
extern "C" hipError_t rt_launch_units_entry(const RtParams* p, unsigned max_listed, hipStream_t s);
extern "C" unsigned rt_units_blocks(unsigned n_units);
extern "C" hipError_t rt_launch_sum_resolve(const RtParams* p, unsigned frame_index, int write_image, hipStream_t s);
extern "C" hipError_t rt_launch_parked_rays(const RtParams* p, int stats, hipStream_t s);
extern "C" hipError_t rt_launch_resolve(const RtParams* p, unsigned frame_index, unsigned gx, unsigned gy, hipStream_t s);
extern "C" hipError_t rt_launch_aa(const RtParams* p, unsigned frame_index, unsigned gx, unsigned gy, hipStream_t s);
extern "C" hipError_t rt_launch_probe_walk(const RtParams* p, const float4* rays, uint32_t n, uint32_t* next, int refill, int variant,
                                           unsigned grid, float4* out, hipStream_t s);
extern "C" hipError_t rt_probe_walk_occupancy(int* blocks_per_cu);
extern "C" hipError_t rt_static_setup(int device);               // rt_static.hip: per-device address of its static working memory

namespace {

// the message of the entry points that have no context (rt_create, rt_group_create, rt_render_spheres_noalloc): per thread, so that
// two threads failing at once neither race on it nor read each other's text
thread_local std::string g_create_error;

template <class T>
struct DevBuf {
    T* ptr = nullptr;
    size_t count = 0;
    void release() { if (ptr) { (void)hipFree(ptr); ptr = nullptr; } count = 0; }
    hipError_t resize(size_t n)
    {
        if (n == count && ptr) return hipSuccess;
        release();
        if (!n) return hipSuccess;
        hipError_t e = hipMalloc((void**)&ptr, n * sizeof(T));
        if (e == hipSuccess) count = n;
        return e;
    }
};

}  // namespace

struct rt_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // ring of event pairs around the trace kernel: timing never makes the host wait for the launch it just made
    static const int EV_RING = 64;
    hipEvent_t ev_start[EV_RING] = {}, ev_stop[EV_RING] = {}, ev_pre[EV_RING] = {};   // pre .. start = pre-pass, start .. stop = trace
    bool ev_has_pre[EV_RING] = {};
    uint32_t ev_frames[EV_RING] = {};                 // Renders in the launch
    unsigned ev_head = 0, ev_tail = 0;               // [tail, head) are recorded and not yet read
    std::string error;

    uint32_t width = 0, height = 0;
    rt_settings settings{1, RT_SCENE_MESH, 1, 0, 8, 1337, RT_ORDER_AUTO, 0};    // Renderer::Settings defaults (Renderer.h:103-110)
    bool have_camera = false;
    float cam_pos[3]{}, inv_proj[16]{}, inv_view[16]{};
    uint32_t band_index = 0, band_count = 1;
    bool record_ids = false, stats = false, wave_clock_on = false;

    uint32_t n_spheres = 0, n_materials = 0, n_nodes = 0, n_tris = 0, n_verts = 0;
    std::vector<float> h_verts;           // kept for the tree re-layout
    std::vector<uint32_t> h_tris;
    std::vector<uint32_t> h_sphere_mat, h_tri_mat;
    uint32_t tri_mat_max = 0, tri_mat_max_at = 0;        // largest triangle material index and a triangle that has it (checked per frame)

    DevBuf<float4> d_spheres, d_materials, d_nodes, d_leafrecs, d_trishade, d_accum, d_image, d_dirs;
    DevBuf<float4> d_slots;              // tree scenes: slot_frames planes of per-pixel radiance, one per Render of a launch
    uint32_t slot_frames = 0;
    // rows bordering this context's bands (tile split with the 3x3 filter on): whether their sums hold every frame since the
    // accumulation was last restarted, i.e. whether the filter may read them
    bool halo_complete = true;
    bool halo_in_columns = false;        // the column-ray numbering in d_dirs.w includes the halo rows
    bool column_cache = false;           // the column rays' hits on the device belong to the current camera, directions, bands and tree
    DevBuf<float4> d_first_hits;         // rt_settings.reuse_first_hits: the primary ray's mesh hit per pixel ...
    bool first_hits_valid = false;       // ... and whether it belongs to the current camera, directions, bands, tree and walk order
    // mesh-only scenes: the work units in which a primary ray can hit anything (the others are not traced, see RtParams::unit_list)
    DevBuf<unsigned char> d_unit_empty;
    DevBuf<uint32_t> d_unit_list, d_unit_counts, d_unit_entry, d_unit_block;
    bool units_valid = false;            // they belong to the current camera, directions, bands and tree
    bool entry_valid = false;            // ... and so do the entry nodes of the listed units' primary walks (found from the second Render of a view on)
    uint32_t view_age = 0;               // Renders made with the current unit list
    uint32_t entry_delay = 1;            // Renders a view must have seen before its entry nodes are worth finding (RT_ENTRY_DELAY=<n> overrides)
    uint32_t n_list = 0, empty_pixels = 0, entry_nodes = 0, entry_tris = 0;
    // What the device finds per view and the host needs for its book-keeping only (and for the size of a few grids) comes back
    // behind the kernels, into page-locked memory, and is looked at when it has arrived: no launch waits for it.
    // h_view: [0] column rays, [1] listed units, [2] pixels of empty units, [3] / [4] node steps / triangle tests the entry nodes spare one Render
    uint32_t* h_view = nullptr;
    hipEvent_t ev_columns = nullptr, ev_units = nullptr, ev_entry = nullptr;
    bool columns_pending = false, units_pending = false, entry_pending = false;     // the copy is queued and has not been seen to arrive
    bool columns_known = true, units_known = false, entry_known = false;            // the host fields above hold this view's values
    unsigned long long owed_unit_frames = 0, owed_entry_frames = 0;                 // Renders whose untraced rays / spared steps are not in the host counters yet
    // the word the main trace launch parks paths under, read back behind it (see trace_issue / trace_finish)
    uint32_t* h_parked = nullptr;
    hipEvent_t parked_ready = nullptr;
    unsigned long long host_entry_nodes = 0, host_entry_tris = 0;    // steps the entry points spared the primary walks
    unsigned long long host_rays = 0;    // primary rays of empty units: TraceRay invocations that needed no kernel
    unsigned long long view_setups = 0;  // times the unit list or the first hits were built
    DevBuf<uint32_t> d_sphere_mat, d_work_heads;
    DevBuf<int32_t> d_ids;
    DevBuf<unsigned long long> d_counters, d_wave_clock, d_column_stats;
    uint32_t last_grid_waves = 0;
    // Mesh scenes: the units without a traced path hold what the first Render of the running accumulation left in them (a sum of +0,
    // its image, ids of -1), and every launch since was one that leaves them alone: the resolve pass then visits the listed units only.
    bool empties_settled = false, settled_record_ids = false;
    // ... and, per frame buffer, whether those units' pixels hold that image: [0] is the buffer d_image names at the moment, [1] the
    // back buffer of the pipelined hand-off.  The two change places at every rt_map_image_pipelined, and a Render may leave the empty
    // units of the buffer it resolves into alone only if THAT buffer was resolved in full since the state was last dropped.
    bool image_settled[2] = {false, false};
    void unsettle() { empties_settled = false; image_settled[0] = image_settled[1] = false; }
    bool heads_zero = false;                            // the ticket heads and park counters are zero on the stream (left so by the last resolve kernel)
    uint32_t timing_every = 0, timing_count = 0;        // kernel times: 0 = nobody has asked yet; else events around every n-th launch
    unsigned resume_lanes = 1;              // parked paths per wave of a resume launch (RT_RESUME_LANES=<1..64, a power of two> overrides)
    std::vector<float4> h_spheres;       // { center.xyz, radius^2 } as uploaded
    std::vector<float4> h_materials;     // packed materials as uploaded
    // page-locked staging for the per-frame scene tables: a ring, so an upload never waits for the frame before it
    static const int UP_RING = 4;
    void* up_buf[UP_RING] = {}; size_t up_bytes[UP_RING] = {}; hipEvent_t up_done[UP_RING] = {}; unsigned up_next = 0;
    // page-locked staging for rt_map_image / rt_map_image_pipelined: two buffers, filled in turn over a copy stream of its own
    float* h_image[2] = {nullptr, nullptr};
    size_t h_image_bytes = 0;
    hipEvent_t h_image_done[2] = {};     // the copy into h_image[i] has finished
    bool h_image_pending[2] = {false, false};
    unsigned h_image_next = 0;
    hipStream_t copy_stream = nullptr;
    hipEvent_t frame_ready = nullptr;    // the frame to be copied is complete on the render stream
    DevBuf<float4> d_image_back;         // pipelined hand-off: the frame being copied out while the next one is resolved into d_image
    bool latest_in_back = false;         // ... and whether that back buffer holds the newest frame (no Render since the hand-off)
    bool use_host_dirs = false;          // d_dirs holds directions the host supplied (rt_set_ray_dirs)
    bool dirs_dirty = true;              // d_dirs must be regenerated from the camera before the next trace

    // column rays (primary rays with direction.x == +0): slots found whenever directions or bands change
    static const uint32_t kColumnCap = 1u << 16;
    static const uint32_t kColumnQueueCap = 1u << 20;
    DevBuf<uint32_t> d_column_count, d_column_pixels, d_column_flags, d_column_block, d_column_queue_count;
    DevBuf<float4> d_column_dirs, d_column_boxes;      // boxes: 16-slot, 64-slot, 256-slot arrays back to back
    DevBuf<uint2> d_column_queue;
    // bounce rays the tree cannot prune, handed to whole-chip kernels: two lists (one being resolved and resumed while the
    // resumed paths park into the other), per-slot results, the families' slot lists
    DevBuf<uint32_t> d_park_count, d_park_flags, d_park_list;
    DevBuf<int32_t> d_park_info;
    DevBuf<float4> d_park_paths;
    DevBuf<unsigned long long> d_park_best, d_park_keys;
    DevBuf<float4> d_park_boxes;
    DevBuf<uint2> d_park_queue;
    DevBuf<uint32_t> d_park_queue_count;
    static const uint32_t kParkQueueCap = 1u << 22;        // pairs per family (32 MB); beyond it the filter tests in place
    DevBuf<unsigned long long> d_column_best;
    DevBuf<unsigned long long> d_reached;
    DevBuf<uint32_t> d_rec_leaf;                        // per leaf record: the node index of its leaf (4 bytes instead of the record's 48)
    bool columns_dirty = true;
    uint32_t n_columns = 0;              // host copy of the count (read back once per change)
    uint32_t n_refs = 0;

    // experiments: rays the counting kernels append while rt_debug_dump_rays runs (RtParams::ray_dump)
    DevBuf<float4> d_ray_dump; DevBuf<uint32_t> d_ray_dump_count;
    bool dump_on = false;

    int n_cus = 0;
    int blocks_per_cu[12] = {};          // trace-kernel residency per (mode, stats, ordered), queried once
    float root_center[3] = {0.0f, 0.0f, 0.0f};                       // see RtParams
    float tie_box[3] = {0.0f, 0.0f, 0.0f}, tie_ulp = 0.0f, tie_rel = 0.0f, tie_abs = 0.0f;
    bool slab_plain = false;                                         // see RtParams
    double mesh_mag = 0.0, mesh_emax = 0.0;                          // largest |coordinate| of a node box, longest edge component of a triangle
    double host_dir_max = 1.0;                                       // largest |component| of the directions the host supplied
    bool tests_may_overflow = false;                                 // see tri_tests_stay_finite (set per launch)
    bool park_min_forced = false;
    uint32_t park_min_nodes = RT_PARK_MIN_NODES;   // RT_PARK_MIN_NODES=<n> in the environment overrides it (tests park in small trees)
    bool kd_partition = false;           // every two-child node of the uploaded tree splits its box in two along one axis
    bool kd_listing = false;             // ... and (on a sample of the triangles) every triangle is listed in each leaf whose box it overlaps

    rt_kernel_time ktime{0, 0.0, 0.0, 0.0, 0};
};

namespace {

int fail(rt_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->error = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                              \
    do { hipError_t e_ = (expr);                                                                     \
         if (e_ != hipSuccess) return fail((c), RT_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

// A kernel launch; with RT_DEBUG_SYNC=1 in the environment the stream is drained right after it, so that a fault is reported
// with the launch that caused it instead of at some later call (debugging aid: every launch then costs a host round trip).
static const bool g_debug_sync = [] { const char* e = getenv("RT_DEBUG_SYNC"); return e && *e && *e != '0'; }();
// RT_STALL_MS=<ms>: host-side steps (waits, whole entry points) that take longer than that are reported on stderr
static const double g_stall_ms = [] { const char* e = getenv("RT_STALL_MS"); return e ? atof(e) : 0.0; }();
struct StallTimer {
    const char* what; std::chrono::steady_clock::time_point t0;
    explicit StallTimer(const char* w) : what(w) { if (g_stall_ms > 0.0) t0 = std::chrono::steady_clock::now(); }
    ~StallTimer()
    {
        if (g_stall_ms <= 0.0) return;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > g_stall_ms) fprintf(stderr, "[rt stall] %s: %.2f ms\n", what, ms);
    }
};
#define LAUNCH_TRY(c, expr)                                                                           \
    do { HIP_TRY(c, expr);                                                                           \
         if (g_debug_sync) { hipError_t s_ = hipStreamSynchronize((c)->stream);                      \
             if (s_ != hipSuccess) return fail((c), RT_E_HIP, "after %s: %s", #expr, hipGetErrorString(s_)); } } while (0)

struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// 16x16-pixel workgroups over the bands this context owns (resolve / AA kernels); gy = number of owned bands.
void grid_for(const rt_ctx* c, unsigned& gx, unsigned& gy)
{
    gx = (c->width + RT_TILE - 1) / RT_TILE;
    unsigned bands = (c->height + RT_TILE - 1) / RT_TILE;
    gy = bands > c->band_index ? (bands - c->band_index + c->band_count - 1) / c->band_count : 0;
}

// Host threads for the set-up of a large mesh (leaf records, the eight node arrays): RT_HOST_THREADS, else up to eight.
unsigned host_threads()
{
    if (const char* e = getenv("RT_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) return (unsigned)(v > 64 ? 64 : v); }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw == 0 ? 1u : (hw > 8u ? 8u : hw);
}

// f(chunk, first, last) over [0, n) in up to `threads` chunks; small ranges stay on the calling thread.
template <typename F>
void in_chunks(size_t n, unsigned threads, F f)
{
    if (threads <= 1 || n < 65536) { f(0u, (size_t)0, n); return; }
    std::vector<std::thread> helpers;
    const size_t per = (n + threads - 1) / threads;
    for (unsigned k = 0; k + 1 < threads; ++k) {
        const size_t first = std::min(n, k * per), last = std::min(n, (k + 1) * per);
        try { helpers.emplace_back([=] { f(k, first, last); }); }
        catch (const std::system_error&) { f(k, first, last); }         // no thread to be had: the chunk is done here
    }
    f(threads - 1, std::min(n, (threads - 1) * per), n);
    for (auto& h : helpers) h.join();
}

constexpr float kAutoMaxTieRel = 1.0f / 512.0f;
constexpr unsigned kMaxBlocksPerCu = 8;      // 8 workgroups of 4 waves = the 32 waves a CU can hold

void fill_params(rt_ctx* c, RtParams& p, uint32_t frame_first, uint32_t n_frames)
{
    memset(&p, 0, sizeof p);
    memcpy(p.cam_pos, c->cam_pos, sizeof p.cam_pos);
    memcpy(p.inv_proj, c->inv_proj, sizeof p.inv_proj);
    memcpy(p.inv_view, c->inv_view, sizeof p.inv_view);
    p.width = c->width; p.height = c->height;
    p.frame_first = frame_first; p.n_frames = n_frames;
    p.seed = c->settings.seed; p.bounces = c->settings.bounces;
    p.n_spheres = c->n_spheres; p.n_materials = c->n_materials; p.n_nodes = c->n_nodes;
    p.band_index = c->band_index; p.band_count = c->band_count;
    unsigned gx, gy; grid_for(c, gx, gy);
    p.tile_rows = 2u * gy;                               // two 8-row tile rows per 16-row band
    p.n_units_main = ((c->width + 7) / 8) * p.tile_rows * (64u / RT_UNIT);
    // under a tile split the 3x3 filter reads the rows just outside each owned band: this context traces and sums those too
    p.halo = (c->band_count > 1 && c->settings.anti_aliasing) ? 1u : 0u;
    p.n_units = p.n_units_main + (p.halo ? gy * 2u * ((c->width + 15u) / 16u) : 0u);
    p.slots = c->d_slots.ptr;
    for (uint32_t i = 0; i < c->n_spheres && i < RT_INLINE_SPHERES; ++i) p.sph_inline[i] = c->h_spheres[i];
    p.use_ace = c->settings.use_ace;
    p.spheres = c->d_spheres.ptr; p.sphere_mat = c->d_sphere_mat.ptr; p.materials = c->d_materials.ptr;
    p.nodes = c->d_nodes.ptr; p.leafrecs = c->d_leafrecs.ptr; p.trishade = c->d_trishade.ptr;
    p.dirs = c->d_dirs.ptr; p.dirs_out = c->d_dirs.ptr;
    p.accum = c->d_accum.ptr; p.image = c->d_image.ptr;
    p.primary_ids = c->record_ids ? c->d_ids.ptr : nullptr;
    p.work_heads = c->d_work_heads.ptr;
    p.counters = c->d_counters.ptr; p.wave_clock = c->wave_clock_on ? c->d_wave_clock.ptr : nullptr; p.column_stats = c->d_column_stats.ptr;
    p.n_refs = c->n_refs; p.n_tris = c->n_tris; p.column_cap = rt_ctx::kColumnCap;
    p.column_count = c->d_column_count.ptr; p.column_pixels = c->d_column_pixels.ptr;
    p.column_best = c->d_column_best.ptr; p.column_flags = c->d_column_flags.ptr; p.reached = c->d_reached.ptr; p.rec_leaf = c->d_rec_leaf.ptr;
    p.column_dirs = c->d_column_dirs.ptr; p.column_block = c->d_column_block.ptr;
    p.column_box16 = c->d_column_boxes.ptr;
    p.column_box64 = p.column_box16 + (rt_ctx::kColumnCap + 15u) / 16u;
    p.column_box256 = p.column_box64 + (rt_ctx::kColumnCap + 63u) / 64u;
    p.column_queue = c->d_column_queue.ptr; p.column_queue_count = c->d_column_queue_count.ptr;
    p.column_queue_cap = rt_ctx::kColumnQueueCap;
    p.park_count = c->d_park_count.ptr; p.park_flags = c->d_park_flags.ptr; p.park_paths = c->d_park_paths.ptr; p.park_best = c->d_park_best.ptr;
    p.resume_count = c->d_park_count.ptr; p.resume_paths = c->d_park_paths.ptr; p.park_list = c->d_park_list.ptr; p.park_info = c->d_park_info.ptr;
    p.park_keys = c->d_park_keys.ptr; p.park_boxes = c->d_park_boxes.ptr; p.park_queue = c->d_park_queue.ptr;
    p.park_queue_count = c->d_park_queue_count.ptr; p.park_queue_cap = rt_ctx::kParkQueueCap;
    p.ray_slot = -1; p.resume = 0; p.park_min_nodes = c->park_min_nodes;
    if (c->dump_on) { p.ray_dump = c->d_ray_dump.ptr; p.ray_dump_count = c->d_ray_dump_count.ptr; p.ray_dump_cap = (uint32_t)(c->d_ray_dump.count / 2); }
    p.first_hits = nullptr; p.first_hit_mode = 0u;
    // (measured: mesh-only scenes with a large tree, and their launches of several Renders, gain 2-4 % from the shorter slices; with
    // spheres and mesh together the lanes that leave a slice run the sphere loop and the shading next, and the fuller batches of
    // longer slices are worth 3 %)
    p.slice_leave = c->settings.scene_mode != RT_SCENE_BOTH && (c->n_nodes >= RT_LARGE_TREE || n_frames > 1u) ? RT_SLICE_LEAVE_LARGE : RT_SLICE_LEAVE;
    // tuning knob: RT_SLICE_LEAVE=<1..3> (quarters of a slice's walkers that are left when it ends)
    static const int slice_leave_env = [] { const char* e = getenv("RT_SLICE_LEAVE"); return e ? atoi(e) : 0; }();
    if (slice_leave_env >= 1 && slice_leave_env <= 3) p.slice_leave = (uint32_t)slice_leave_env;
    for (int a = 0; a < 3; ++a) { p.root_center[a] = c->root_center[a]; p.tie_box[a] = c->tie_box[a]; }
    p.slab_plain = c->slab_plain ? 1u : 0u;
    p.tie_ulp = c->tie_ulp; p.tie_rel = c->tie_rel; p.tie_abs = c->tie_abs;
}

// Every material index a kernel can read must be in range: the kernels do not check.
int validate_scene(rt_ctx* c, int mode)
{
    if (mode == RT_SCENE_SPHERES || mode == RT_SCENE_BOTH) {
        for (uint32_t i = 0; i < c->n_spheres; ++i)
            if (c->h_sphere_mat[i] >= c->n_materials)
                return fail(c, RT_E_ARG, "sphere %u: material index %u out of range (%u materials)", i, c->h_sphere_mat[i], c->n_materials);
    }
    if (mode == RT_SCENE_MESH || mode == RT_SCENE_BOTH) {
        if (!c->n_nodes) return fail(c, RT_E_STATE, "scene_mode %d needs a KD tree (rt_set_mesh + rt_set_kdtree)", mode);
        if (c->n_tris && c->tri_mat_max >= c->n_materials)
            return fail(c, RT_E_ARG, "triangle %u: material index %u out of range (%u materials)", c->tri_mat_max_at, c->tri_mat_max, c->n_materials);
    }
    return RT_OK;
}

// Read back finished event pairs; with `all` (or when the ring is full) wait for the unfinished ones too.
int collect_time(rt_ctx* c, bool all)
{
    while (c->ev_tail != c->ev_head) {
        const unsigned i = c->ev_tail % rt_ctx::EV_RING;
        const bool full = c->ev_head - c->ev_tail >= (unsigned)rt_ctx::EV_RING;
        if (!all && !full && hipEventQuery(c->ev_stop[i]) == hipErrorNotReady) break;
        HIP_TRY(c, hipEventSynchronize(c->ev_stop[i]));
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_start[i], c->ev_stop[i]));
        c->ktime.launches++; c->ktime.total_ms += ms; c->ktime.last_ms = ms; c->ktime.frames += c->ev_frames[i];
        if (c->ev_has_pre[i]) {
            HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_pre[i], c->ev_start[i]));
            c->ktime.prepass_total_ms += ms;
        }
        c->ev_tail++;
    }
    return RT_OK;
}

// Host table -> device buffer on the context's stream, through the ring of page-locked staging buffers: the call returns at
// once and the copy takes its place among the frames.  (A slot is reused only after its previous copy has finished.)
int staged_upload(rt_ctx* c, void* dst, const void* src, size_t bytes)
{
    if (!bytes) return RT_OK;
    const int i = (int)(c->up_next++ % rt_ctx::UP_RING);
    if (c->up_bytes[i]) HIP_TRY(c, hipEventSynchronize(c->up_done[i]));
    if (c->up_bytes[i] < bytes) {
        if (c->up_buf[i]) { (void)hipHostFree(c->up_buf[i]); c->up_buf[i] = nullptr; c->up_bytes[i] = 0; }
        const size_t cap = bytes < 4096 ? 4096 : bytes;
        hipError_t e = hipHostMalloc(&c->up_buf[i], cap, hipHostMallocDefault);
        if (e != hipSuccess) { c->up_buf[i] = nullptr; return fail(c, RT_E_NOMEM, "hipHostMalloc(%zu): %s", cap, hipGetErrorString(e)); }
        c->up_bytes[i] = cap;
    }
    memcpy(c->up_buf[i], src, bytes);
    HIP_TRY(c, hipMemcpyAsync(dst, c->up_buf[i], bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipEventRecord(c->up_done[i], c->stream));
    return RT_OK;
}

}  // namespace

// Message for rt_last_error(NULL) from the entry points that have no context.
extern "C" __attribute__((visibility("hidden"))) void rt_internal_set_error(const char* msg) { g_create_error = msg ? msg : ""; }

extern "C" {

rt_ctx* rt_create(int device_id)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { fail(nullptr, RT_E_HIP, "no HIP device: %s", hipGetErrorString(e)); return nullptr; }
    if (device_id < 0 || device_id >= n) { fail(nullptr, RT_E_ARG, "device %d out of range (%d devices)", device_id, n); return nullptr; }
    rt_ctx* c = new rt_ctx();
    c->device = device_id;
    DeviceGuard g(device_id);
    bool ok = g.ok && hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < rt_ctx::EV_RING; ++i)
        ok = hipEventCreate(&c->ev_start[i]) == hipSuccess && hipEventCreate(&c->ev_stop[i]) == hipSuccess &&
             hipEventCreate(&c->ev_pre[i]) == hipSuccess;
    for (int i = 0; ok && i < rt_ctx::UP_RING; ++i) ok = hipEventCreateWithFlags(&c->up_done[i], hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < 2; ++i) ok = hipEventCreateWithFlags(&c->h_image_done[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&c->frame_ready, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_columns, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_units, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_entry, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->parked_ready, hipEventDisableTiming) == hipSuccess &&
         hipHostMalloc((void**)&c->h_view, 8 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess &&
         hipHostMalloc((void**)&c->h_parked, 2 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    if (ok) { memset(c->h_view, 0, 8 * sizeof(uint32_t)); memset(c->h_parked, 0, 2 * sizeof(uint32_t)); }
    if (!ok) {
        fail(nullptr, RT_E_HIP, "stream/event creation failed on device %d", device_id);
        rt_destroy(c); return nullptr;
    }
    c->stream = c->own_stream;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess || prop.multiProcessorCount <= 0 ||
        c->d_counters.resize((size_t)prop.multiProcessorCount * kMaxBlocksPerCu * 4 * 4) != hipSuccess ||
        hipMemset(c->d_counters.ptr, 0, c->d_counters.count * sizeof(unsigned long long)) != hipSuccess ||
        c->d_wave_clock.resize((size_t)prop.multiProcessorCount * kMaxBlocksPerCu * 4 * RT_WAVE_CLOCK_WORDS) != hipSuccess ||
        hipMemset(c->d_wave_clock.ptr, 0, c->d_wave_clock.count * sizeof(unsigned long long)) != hipSuccess ||
        c->d_park_count.resize(2) != hipSuccess || hipMemset(c->d_park_count.ptr, 0, 8) != hipSuccess ||
        c->d_park_flags.resize(RT_PARK_SLOTS) != hipSuccess || hipMemset(c->d_park_flags.ptr, 0, 4 * RT_PARK_SLOTS) != hipSuccess ||
        c->d_park_list.resize(RT_PARK_FAMILIES * RT_PARK_SLOTS) != hipSuccess || c->d_park_keys.resize(RT_PARK_FAMILIES * RT_PARK_SLOTS) != hipSuccess ||
        c->d_park_boxes.resize((size_t)RT_PARK_FAMILIES * 2 * (RT_PARK_BOXES + 1)) != hipSuccess ||
        c->d_park_queue.resize((size_t)RT_PARK_FAMILIES * rt_ctx::kParkQueueCap) != hipSuccess || c->d_park_queue_count.resize(RT_PARK_FAMILIES) != hipSuccess || c->d_park_info.resize(3 * RT_PARK_FAMILIES) != hipSuccess ||
        c->d_park_paths.resize(2 * (size_t)RT_PARK_SLOTS * RT_PARK_WORDS) != hipSuccess || c->d_park_best.resize(RT_PARK_SLOTS) != hipSuccess ||
        c->d_work_heads.resize(RT_WORK_HEADS * RT_HEAD_STRIDE) != hipSuccess ||
        hipMemset(c->d_work_heads.ptr, 0, RT_WORK_HEADS * RT_HEAD_STRIDE * sizeof(uint32_t)) != hipSuccess ||
        c->d_column_stats.resize(8) != hipSuccess || hipMemset(c->d_column_stats.ptr, 0, 64) != hipSuccess ||
        c->d_column_count.resize(1) != hipSuccess || c->d_column_flags.resize(1) != hipSuccess ||
        c->d_column_pixels.resize(rt_ctx::kColumnCap) != hipSuccess || c->d_column_best.resize(rt_ctx::kColumnCap) != hipSuccess ||
        c->d_column_dirs.resize(rt_ctx::kColumnCap) != hipSuccess || c->d_column_queue_count.resize(1) != hipSuccess ||
        c->d_column_boxes.resize((rt_ctx::kColumnCap + 15u) / 16u + (rt_ctx::kColumnCap + 63u) / 64u + (rt_ctx::kColumnCap + 255u) / 256u) != hipSuccess ||
        c->d_column_queue.resize(rt_ctx::kColumnQueueCap) != hipSuccess || hipMemset(c->d_column_queue_count.ptr, 0, 4) != hipSuccess ||
        hipMemset(c->d_column_count.ptr, 0, 4) != hipSuccess || hipMemset(c->d_column_flags.ptr, 0, 4) != hipSuccess) {
        fail(nullptr, RT_E_HIP, "device query / counter allocation failed on device %d", device_id);
        rt_destroy(c); return nullptr;
    }
    c->n_cus = prop.multiProcessorCount;
    (void)rt_static_setup(device_id);      // set-up of rt_render_spheres_noalloc on this device (it reports a failure itself when called)
    if (const char* e = getenv("RT_RESUME_LANES")) { const int v = atoi(e); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) c->resume_lanes = (unsigned)v; }
    if (const char* e = getenv("RT_ENTRY_DELAY")) { const int v = atoi(e); if (v >= 0) c->entry_delay = (uint32_t)v; }
    if (const char* e = getenv("RT_PARK_MIN_NODES")) { const long v = atol(e); if (v >= 1) { c->park_min_nodes = (uint32_t)v; c->park_min_forced = true; } }
    return c;
}

void rt_destroy(rt_ctx* c)
{
    if (!c) return;
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    c->d_spheres.release(); c->d_materials.release(); c->d_nodes.release(); c->d_leafrecs.release();
    c->d_trishade.release(); c->d_accum.release(); c->d_image.release(); c->d_sphere_mat.release();
    c->d_dirs.release(); c->d_ids.release(); c->d_counters.release(); c->d_wave_clock.release(); c->d_column_stats.release(); c->d_work_heads.release();
    c->d_park_count.release(); c->d_park_flags.release(); c->d_park_paths.release(); c->d_park_best.release();
    c->d_park_list.release(); c->d_park_info.release(); c->d_park_keys.release(); c->d_park_boxes.release();
    c->d_park_queue.release(); c->d_park_queue_count.release();
    c->d_column_count.release(); c->d_column_pixels.release(); c->d_column_flags.release(); c->d_column_best.release(); c->d_reached.release(); c->d_rec_leaf.release();
    c->d_column_block.release(); c->d_column_queue_count.release(); c->d_column_dirs.release(); c->d_column_boxes.release(); c->d_column_queue.release();
    for (int i = 0; i < rt_ctx::EV_RING; ++i) {
        if (c->ev_start[i]) (void)hipEventDestroy(c->ev_start[i]);
        if (c->ev_stop[i]) (void)hipEventDestroy(c->ev_stop[i]);
        if (c->ev_pre[i]) (void)hipEventDestroy(c->ev_pre[i]);
    }
    c->d_slots.release(); c->d_image_back.release(); c->d_first_hits.release(); c->d_ray_dump.release(); c->d_ray_dump_count.release();
    c->d_unit_empty.release(); c->d_unit_list.release(); c->d_unit_counts.release(); c->d_unit_entry.release(); c->d_unit_block.release();
    for (int i = 0; i < rt_ctx::UP_RING; ++i) {
        if (c->up_done[i]) (void)hipEventDestroy(c->up_done[i]);
        if (c->up_buf[i]) (void)hipHostFree(c->up_buf[i]);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->h_image_done[i]) (void)hipEventDestroy(c->h_image_done[i]);
        if (c->h_image[i]) (void)hipHostFree(c->h_image[i]);
    }
    if (c->frame_ready) (void)hipEventDestroy(c->frame_ready);
    for (hipEvent_t ev : {c->ev_columns, c->ev_units, c->ev_entry, c->parked_ready}) if (ev) (void)hipEventDestroy(ev);
    if (c->h_view) (void)hipHostFree(c->h_view);
    if (c->h_parked) (void)hipHostFree(c->h_parked);
    if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char* rt_last_error(const rt_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int rt_set_stream(rt_ctx* c, void* hip_stream)
{
    if (!c) return RT_E_ARG;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return RT_OK;
}

int rt_resize(rt_ctx* c, uint32_t w, uint32_t h)
{
    if (!c) return RT_E_ARG;
    if (!w || !h || (uint64_t)w * h > 0x7FFFFFFFull / 4) return fail(c, RT_E_ARG, "bad size %ux%u", w, h);
    if (w == c->width && h == c->height) return RT_OK;           // Renderer.cpp:23-24
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    const size_t n = (size_t)w * h;
    // all or nothing: a context whose buffers could not all be had has no frame (width = height = 0) and every call that
    // needs one returns RT_E_STATE, instead of kernels running over buffers of the wrong size
    c->width = 0; c->height = 0;
    c->d_slots.release(); c->slot_frames = 0; c->d_image_back.release(); c->latest_in_back = false;
    c->d_first_hits.release(); c->first_hits_valid = false;
    for (int i = 0; i < 2; ++i) { if (c->h_image[i]) { (void)hipHostFree(c->h_image[i]); c->h_image[i] = nullptr; } c->h_image_pending[i] = false; }
    c->h_image_bytes = 0;
    hipError_t e = c->d_accum.resize(n);
    if (e == hipSuccess) e = c->d_image.resize(n);
    if (e == hipSuccess) e = c->d_ids.resize(n);
    if (e == hipSuccess) e = c->d_dirs.resize(n);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_accum.ptr, 0, n * sizeof(float4), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_image.ptr, 0, n * sizeof(float4), c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(c->d_ids.ptr, 0xFF, n * sizeof(int32_t), c->stream);
    if (e != hipSuccess) {
        c->d_accum.release(); c->d_image.release(); c->d_ids.release(); c->d_dirs.release();
        return fail(c, e == hipErrorOutOfMemory ? RT_E_NOMEM : RT_E_HIP, "rt_resize(%u, %u): %s", w, h, hipGetErrorString(e));
    }
    c->use_host_dirs = false; c->dirs_dirty = true; c->columns_dirty = true; c->column_cache = false; c->first_hits_valid = false; c->units_valid = false; c->halo_complete = true; c->unsettle();
    c->width = w; c->height = h;
    return RT_OK;
}

int rt_set_settings(rt_ctx* c, const rt_settings* s)
{
    if (!c || !s) return RT_E_ARG;
    if (s->scene_mode < 0 || s->scene_mode > 2) return fail(c, RT_E_ARG, "scene_mode %d", s->scene_mode);
    if (s->traversal < RT_ORDER_AUTO || s->traversal > RT_ORDER_FRONT_TO_BACK) return fail(c, RT_E_ARG, "traversal %d", s->traversal);
    if (s->scene_mode != c->settings.scene_mode || s->traversal != c->settings.traversal) { c->first_hits_valid = false; c->units_valid = false; }
    if (memcmp(s, &c->settings, sizeof *s) != 0) c->unsettle();
    c->settings = *s;
    return RT_OK;
}

int rt_set_spheres(rt_ctx* c, const rt_sphere* sp, uint32_t n)
{
    if (!c || (n && !sp)) return RT_E_ARG;
    std::vector<float4> packed(n);
    std::vector<uint32_t> mat(n);
    for (uint32_t i = 0; i < n; ++i) {
        packed[i] = make_float4(sp[i].x, sp[i].y, sp[i].z, sp[i].radius * sp[i].radius);   // Renderer.cpp:257's r*r
        mat[i] = sp[i].material;
    }
    // the UI sends the tables every frame (RaytracerApp.cpp:283-331) and changes them rarely: nothing to do when they are the same
    if (n == c->n_spheres && c->d_spheres.count == n && (!n || (memcmp(packed.data(), c->h_spheres.data(), n * sizeof(float4)) == 0 &&
                                                                 memcmp(mat.data(), c->h_sphere_mat.data(), n * sizeof(uint32_t)) == 0)))
        return RT_OK;
    DeviceGuard g(c->device);
    if (c->d_spheres.count != n) {                               // another number of spheres: new buffers (rare)
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, c->d_spheres.resize(n));
        HIP_TRY(c, c->d_sphere_mat.resize(n));
    }
    c->n_spheres = 0;                                            // until both copies are queued
    if (int rc = staged_upload(c, c->d_spheres.ptr, packed.data(), n * sizeof(float4))) return rc;
    if (int rc = staged_upload(c, c->d_sphere_mat.ptr, mat.data(), n * sizeof(uint32_t))) return rc;
    c->h_spheres.swap(packed); c->h_sphere_mat.swap(mat);
    c->n_spheres = n;
    return RT_OK;
}

int rt_set_materials(rt_ctx* c, const rt_material* m, uint32_t n)
{
    if (!c || (n && !m)) return RT_E_ARG;
    std::vector<float4> packed(3 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        packed[3 * i] = make_float4(m[i].albedo[0], m[i].albedo[1], m[i].albedo[2], m[i].roughness);
        packed[3 * i + 1] = make_float4(m[i].emission[0], m[i].emission[1], m[i].emission[2], m[i].transparency);
        packed[3 * i + 2] = make_float4(m[i].ior, 0.0f, 0.0f, 0.0f);
    }
    if (n == c->n_materials && c->d_materials.count == packed.size() &&
        (!n || memcmp(packed.data(), c->h_materials.data(), packed.size() * sizeof(float4)) == 0))
        return RT_OK;
    DeviceGuard g(c->device);
    if (c->d_materials.count != packed.size()) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, c->d_materials.resize(packed.size()));
    }
    c->n_materials = 0;
    if (int rc = staged_upload(c, c->d_materials.ptr, packed.data(), packed.size() * sizeof(float4))) return rc;
    c->h_materials.swap(packed);
    c->n_materials = n;
    return RT_OK;
}

int rt_set_mesh(rt_ctx* c, const float* verts, uint32_t nverts, const uint32_t* tri_idx, const float* normals9,
                const uint32_t* tri_mat, uint32_t ntris)
{
    if (!c || !verts || !tri_idx || !normals9 || !tri_mat || !nverts || !ntris) return fail(c, RT_E_ARG, "rt_set_mesh: null/empty input");
    if (ntris >= 0x40000000u) return fail(c, RT_E_LIMIT, "rt_set_mesh: %u triangles exceed the 2^30 id space", ntris);
    for (size_t i = 0; i < 3 * (size_t)ntris; ++i)
        if (tri_idx[i] >= nverts) return fail(c, RT_E_ARG, "rt_set_mesh: vertex index %u out of range at triangle %zu", tri_idx[i], i / 3);
    DeviceGuard g(c->device);
    c->h_verts.assign(verts, verts + 3 * (size_t)nverts);
    c->h_tris.assign(tri_idx, tri_idx + 3 * (size_t)ntris);
    c->h_tri_mat.assign(tri_mat, tri_mat + ntris);
    c->tri_mat_max = 0; c->tri_mat_max_at = 0;
    for (uint32_t i = 0; i < ntris; ++i) if (tri_mat[i] > c->tri_mat_max) { c->tri_mat_max = tri_mat[i]; c->tri_mat_max_at = i; }
    std::unique_ptr<float4[]> shade_mem(new float4[3 * (size_t)ntris]);
    float4* const shade = shade_mem.get();
    in_chunks(ntris, host_threads(), [&](unsigned, size_t t_first, size_t t_last) {
        for (size_t t = t_first; t < t_last; ++t) {
            const float* n = normals9 + 9 * t;
            float mbits; memcpy(&mbits, &tri_mat[t], 4);
            shade[3 * t] = make_float4(n[0], n[1], n[2], n[3]);
            shade[3 * t + 1] = make_float4(n[4], n[5], n[6], n[7]);
            shade[3 * t + 2] = make_float4(n[8], mbits, 0.0f, 0.0f);
        }
    });
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, c->d_trishade.resize(3 * (size_t)ntris));
    HIP_TRY(c, hipMemcpy(c->d_trishade.ptr, shade, 3 * (size_t)ntris * sizeof(float4), hipMemcpyHostToDevice));
    c->n_tris = ntris; c->n_verts = nverts;
    c->n_nodes = 0; c->n_refs = 0; c->d_nodes.release(); c->d_leafrecs.release();     // a new mesh invalidates the tree
    c->column_cache = false; c->first_hits_valid = false; c->units_valid = false;
    return RT_OK;
}

int rt_set_kdtree(rt_ctx* c, const rt_kd_flat* kd)
{
    if (!c || !kd || !kd->n_nodes || !kd->center || !kd->extent || !kd->left || !kd->right || !kd->first || !kd->count ||
        (kd->n_leaf_idx && !kd->leaf_idx))
        return fail(c, RT_E_ARG, "rt_set_kdtree: null/empty input");
    if (!c->n_tris) return fail(c, RT_E_STATE, "rt_set_kdtree: call rt_set_mesh first");
    const uint32_t N = kd->n_nodes;

    // Preorder walk (left before right) with an explicit stack; every node must be reached exactly once.
    std::vector<uint32_t> order; order.reserve(N);
    std::vector<uint32_t> new_index(N, 0xFFFFFFFFu);
    std::vector<uint32_t> stack; stack.push_back(0);
    while (!stack.empty()) {
        uint32_t n = stack.back(); stack.pop_back();
        if (n >= N) return fail(c, RT_E_ARG, "rt_set_kdtree: child index %u out of range", n);
        if (new_index[n] != 0xFFFFFFFFu) return fail(c, RT_E_ARG, "rt_set_kdtree: node %u reached twice (not a tree)", n);
        new_index[n] = (uint32_t)order.size();
        order.push_back(n);
        if (kd->right[n] >= 0) stack.push_back((uint32_t)kd->right[n]);
        if (kd->left[n] >= 0) stack.push_back((uint32_t)kd->left[n]);
    }
    const uint32_t M = (uint32_t)order.size();          // unreachable nodes are dropped
    std::vector<uint32_t> subtree(M, 1);
    for (uint32_t k = M; k-- > 0;) {
        uint32_t n = order[k];
        if (kd->left[n] >= 0) subtree[k] += subtree[new_index[(uint32_t)kd->left[n]]];
        if (kd->right[n] >= 0) subtree[k] += subtree[new_index[(uint32_t)kd->right[n]]];
    }
    const unsigned threads = host_threads();
    const bool times = getenv("RT_KD_TIMES") != nullptr;        // prints where the upload's time goes
    const auto t_start = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (times) fprintf(stderr, "rt_set_kdtree: %s at %.3f s\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
    };
    size_t n_refs = 0;
    std::vector<uint32_t> ref_at(M, 0u);                 // per preorder position: the number of leaf references before it
    for (uint32_t k = 0; k < M; ++k) {
        uint32_t n = order[k];
        ref_at[k] = (uint32_t)n_refs;
        if (kd->left[n] < 0 && kd->right[n] < 0) {
            if (kd->count[n] < 0 || kd->first[n] < 0 || (uint64_t)kd->first[n] + (uint64_t)kd->count[n] > kd->n_leaf_idx)
                return fail(c, RT_E_ARG, "rt_set_kdtree: leaf %u range [%d, +%d) outside leaf_idx", n, kd->first[n], kd->count[n]);
            n_refs += (size_t)kd->count[n];
            if (n_refs >= 0x7FFFFFFFull / 3) break;
        }
    }
    if (n_refs >= 0x7FFFFFFFull / 3) return fail(c, RT_E_LIMIT, "rt_set_kdtree: %zu leaf references exceed the index space", n_refs);

    // Leaf records in the reference's visit order (preorder, left first); a leaf's first record by original node id.
    lap("preorder and leaf ranges");
    // (memory the threads below touch first: a vector would fill it with zeros on this thread)
    struct Records {
        std::unique_ptr<float4[]> mem; size_t n = 0;
        explicit Records(size_t count) : mem(count ? new float4[count] : nullptr), n(count) {}
        float4& operator[](size_t i) { return mem[i]; }
        bool empty() const { return n == 0; }
        size_t size() const { return n; }
        const float4* data() const { return mem.get(); }
    } recs(3 * n_refs);
    std::vector<int32_t> leaf_first_of(N, RT_LEAF_INTERIOR);
    std::vector<int64_t> bad_tri(threads, -1), bad_leaf(threads, -1);      // per chunk: the first triangle index out of range
    in_chunks(M, threads, [&](unsigned chunk, size_t k_first, size_t k_last) {
    for (uint32_t k = (uint32_t)k_first; k < (uint32_t)k_last; ++k) {
        uint32_t n = order[k];
        if (kd->left[n] >= 0 || kd->right[n] >= 0) continue;
        size_t next_ref = ref_at[k];
        leaf_first_of[n] = kd->count[n] > 0 ? (int32_t)next_ref : RT_LEAF_EMPTY;
        for (int32_t i = 0; i < kd->count[n]; ++i) {
            int32_t tri = kd->leaf_idx[kd->first[n] + i];
            if (tri < 0 || (uint32_t)tri >= c->n_tris) { if (bad_leaf[chunk] < 0) { bad_tri[chunk] = tri; bad_leaf[chunk] = n; } return; }
            const uint32_t* ix = &c->h_tris[3 * (size_t)tri];
            const float* v0 = &c->h_verts[3 * (size_t)ix[0]];
            const float* v1 = &c->h_verts[3 * (size_t)ix[1]];
            const float* v2 = &c->h_verts[3 * (size_t)ix[2]];
            // e1 = v1 - v0, e2 = v2 - v0: the first two operations of triIntersect (Intersections.cpp:58-59)
            volatile float e1x = v1[0] - v0[0], e1y = v1[1] - v0[1], e1z = v1[2] - v0[2];
            volatile float e2x = v2[0] - v0[0], e2y = v2[1] - v0[1], e2z = v2[2] - v0[2];
            uint32_t tag = (uint32_t)tri | ((i == kd->count[n] - 1) ? 0x80000000u : 0u);
            float tagf; memcpy(&tagf, &tag, 4);
            recs[3 * next_ref] = make_float4(v0[0], v0[1], v0[2], e1x);
            recs[3 * next_ref + 1] = make_float4(e1y, e1z, e2x, e2y);
            float nodef; memcpy(&nodef, &k, 4);
            recs[3 * next_ref + 2] = make_float4(e2z, tagf, nodef, 0.0f);
            ++next_ref;
        }
    }
    });
    for (unsigned k = 0; k < threads; ++k)
        if (bad_leaf[k] >= 0) return fail(c, RT_E_ARG, "rt_set_kdtree: triangle index %d out of range in leaf %u", (int)bad_tri[k], (unsigned)bad_leaf[k]);
    lap("leaf records");

    // Per two-child node: the axis along which its children are separated and which child is the low one; whether the
    // tree is a KD tree (each two-child node's children are its own box cut in two along that axis).
    std::vector<uint8_t> sep_axis(N, 3), low_is_left(N, 1);
    std::vector<uint8_t> chunk_partition(threads, 1);
    in_chunks(M, threads, [&](unsigned chunk, size_t k_first, size_t k_last) {
    bool partition = true;
    for (uint32_t k = (uint32_t)k_first; k < (uint32_t)k_last; ++k) {
        const uint32_t n = order[k];
        const int32_t l = kd->left[n], r = kd->right[n];
        if (l < 0 || r < 0) continue;
        const float* cn = &kd->center[3 * (size_t)n]; const float* en = &kd->extent[3 * (size_t)n];
        const float* cl = &kd->center[3 * (size_t)l]; const float* el = &kd->extent[3 * (size_t)l];
        const float* cr = &kd->center[3 * (size_t)r]; const float* er = &kd->extent[3 * (size_t)r];
        int axis = 3; float best = 0.0f;
        for (int a = 0; a < 3; ++a) {
            const float d = cl[a] < cr[a] ? cr[a] - cl[a] : cl[a] - cr[a];
            if (d > best) { best = d; axis = a; }
        }
        sep_axis[n] = (uint8_t)axis;
        if (axis == 3) continue;                                   // coincident boxes (a flat node): any order will do
        low_is_left[n] = cl[axis] <= cr[axis] ? 1 : 0;
        const float* clo = low_is_left[n] ? cl : cr; const float* elo = low_is_left[n] ? el : er;
        const float* chi = low_is_left[n] ? cr : cl; const float* ehi = low_is_left[n] ? er : el;
        for (int a = 0; a < 3 && partition; ++a) {
            if (a != axis) { partition = cl[a] == cn[a] && cr[a] == cn[a] && el[a] == en[a] && er[a] == en[a]; continue; }
            // the builder forms the child boxes with a few roundings at the magnitude of the coordinates themselves
            const float mag = (cn[a] < 0.0f ? -cn[a] : cn[a]) + en[a];
            const float tol = 1e-5f * mag + 1e-30f;
            const float lo = cn[a] - en[a], hi = cn[a] + en[a];
            auto near = [tol](float x, float y) { return (x < y ? y - x : x - y) <= tol; };
            partition = near(clo[a] - elo[a], lo) && near(chi[a] + ehi[a], hi) && near(clo[a] + elo[a], chi[a] - ehi[a]);
        }
    }
    chunk_partition[chunk] = partition ? 1 : 0;
    });
    bool partition = true;
    for (unsigned k = 0; k < threads; ++k) partition = partition && chunk_partition[k] != 0;
    lap("child order per node");

    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // the walk reads RT_WALK_FETCH consecutive float4 at its cursor (three nodes or two leaf records): pad the ends
    HIP_TRY(c, c->d_nodes.resize(8 * 2 * (size_t)M + RT_WALK_FETCH));
    HIP_TRY(c, c->d_leafrecs.resize((recs.empty() ? 3 : recs.size()) + RT_WALK_FETCH));
    HIP_TRY(c, hipMemset(c->d_nodes.ptr + 8 * 2 * (size_t)M, 0, RT_WALK_FETCH * sizeof(float4)));            // (what is read there is never used)
    HIP_TRY(c, hipMemset(c->d_leafrecs.ptr + (recs.empty() ? 3 : recs.size()), 0, RT_WALK_FETCH * sizeof(float4)));

    // Eight node arrays, one per sign pattern of a ray direction (bit a set: direction[a] < 0).  Array o is the tree in the
    // preorder that walks, at every two-child node, the child on the side a ray of that pattern comes from first.  Array 0
    // is always left-first: it is the reference's own order (RT_ORDER_REFERENCE, the column pre-pass and the cooperative
    // walk use it), and for a tree of the reference builder the left child is the low one, which is what pattern 0 wants.
    // (the arrays are independent of each other: up to `threads` of them are made at a time, each by a thread of its own)
    std::vector<uint32_t> subtree_of(N, 0);
    for (uint32_t k = 0; k < M; ++k) subtree_of[order[k]] = subtree[k];
    const unsigned at_a_time = threads < 8u ? threads : 8u;
    std::vector<std::unique_ptr<float4[]>> arrays(at_a_time);
    for (uint32_t o_first = 0; o_first < 8; o_first += at_a_time) {
        auto make_array = [&](uint32_t o) {
            if (!arrays[o - o_first]) arrays[o - o_first].reset(new float4[2 * (size_t)M]);
            float4* const nodes = arrays[o - o_first].get();
            std::vector<uint32_t> todo; todo.push_back(0);
            uint32_t k = 0;
            while (!todo.empty()) {
                const uint32_t n = todo.back(); todo.pop_back();
                const int32_t l = kd->left[n], r = kd->right[n];
                const uint32_t skip = k + subtree_of[n];
                const int32_t leaf_first = leaf_first_of[n];
                float skipf, leaff; memcpy(&skipf, &skip, 4); memcpy(&leaff, &leaf_first, 4);
                nodes[2 * (size_t)k] = make_float4(kd->center[3 * (size_t)n], kd->center[3 * (size_t)n + 1], kd->center[3 * (size_t)n + 2], skipf);
                nodes[2 * (size_t)k + 1] = make_float4(kd->extent[3 * (size_t)n], kd->extent[3 * (size_t)n + 1], kd->extent[3 * (size_t)n + 2], leaff);
                ++k;
                if (l >= 0 && r >= 0) {
                    bool left_first = true;
                    if (o != 0 && sep_axis[n] != 3) {
                        const bool from_high = (o >> sep_axis[n]) & 1u;            // the ray travels towards lower coordinates
                        left_first = from_high ? !low_is_left[n] : (bool)low_is_left[n];
                    }
                    if (left_first) { todo.push_back((uint32_t)r); todo.push_back((uint32_t)l); }
                    else            { todo.push_back((uint32_t)l); todo.push_back((uint32_t)r); }
                } else if (l >= 0) todo.push_back((uint32_t)l);
                else if (r >= 0) todo.push_back((uint32_t)r);
            }
        };
        const uint32_t o_last = o_first + at_a_time < 8u ? o_first + at_a_time : 8u;
        std::vector<std::thread> helpers;
        for (uint32_t o = o_first + 1; o < o_last; ++o) {
            if (M < 65536u) { make_array(o); continue; }
            try { helpers.emplace_back(make_array, o); }
            catch (const std::system_error&) { make_array(o); }
        }
        make_array(o_first);
        for (auto& h : helpers) h.join();
        for (uint32_t o = o_first; o < o_last; ++o)
            HIP_TRY(c, hipMemcpy(c->d_nodes.ptr + (size_t)o * 2 * M, arrays[o - o_first].get(), 2 * (size_t)M * sizeof(float4), hipMemcpyHostToDevice));
    }
    arrays.clear(); arrays.shrink_to_fit();
    lap("eight node arrays made and copied");
    if (!recs.empty()) HIP_TRY(c, hipMemcpy(c->d_leafrecs.ptr, recs.data(), recs.size() * sizeof(float4), hipMemcpyHostToDevice));
    lap("leaf records copied");
    HIP_TRY(c, c->d_reached.resize(((size_t)M + 63) / 64));
    HIP_TRY(c, hipMemset(c->d_reached.ptr, 0, c->d_reached.count * sizeof(unsigned long long)));
    {
        std::vector<uint32_t> rec_leaf(recs.size() / 3 + 1, 0u);
        in_chunks(recs.size() / 3, threads, [&](unsigned, size_t r_first, size_t r_last) {
            for (size_t r = r_first; r < r_last; ++r) memcpy(&rec_leaf[r], &recs[3 * r + 2].z, 4);
        });
        HIP_TRY(c, c->d_rec_leaf.resize(rec_leaf.size()));
        HIP_TRY(c, hipMemcpy(c->d_rec_leaf.ptr, rec_leaf.data(), rec_leaf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    {
        // the near-tie distance of the front-to-back walk (RtParams::tie_*)
        // (maxima over the nodes and over the triangles, found per chunk and combined: a maximum does not depend on the order)
        struct Bounds { double reach[3] = {0.0, 0.0, 0.0}, mag[3] = {0.0, 0.0, 0.0}, k_max = 0.0, e_max = 0.0; bool bounded = true; };
        std::vector<Bounds> part(2 * (size_t)threads);
        in_chunks(N, threads, [&](unsigned chunk, size_t n_first, size_t n_last) {
            Bounds& b = part[chunk];
            for (size_t n = n_first; n < n_last; ++n)
                for (int a = 0; a < 3; ++a) {
                    const double ce = kd->center[3 * n + a], ex = std::fabs((double)kd->extent[3 * n + a]);
                    b.reach[a] = std::max(b.reach[a], std::fabs(ce - (double)kd->center[a]) + ex);
                    b.mag[a] = std::max(b.mag[a], std::fabs(ce) + ex);
                    if (!(std::fabs(ce) + ex <= 0x1p60)) b.bounded = false;
                }
        });
        in_chunks(c->n_tris, threads, [&](unsigned chunk, size_t t_first, size_t t_last) {
            Bounds& b = part[threads + chunk];
            for (size_t t = t_first; t < t_last; ++t) {
                const uint32_t* ix = &c->h_tris[3 * t];
                const float* v0 = &c->h_verts[3 * (size_t)ix[0]]; const float* v1 = &c->h_verts[3 * (size_t)ix[1]]; const float* v2 = &c->h_verts[3 * (size_t)ix[2]];
                double n1 = 0.0, n2 = 0.0;
                for (int a = 0; a < 3; ++a) {
                    const double d1 = std::fabs((double)v1[a] - v0[a]), d2 = std::fabs((double)v2[a] - v0[a]);
                    n1 += d1; n2 += d2; b.e_max = std::max(b.e_max, std::max(d1, d2));
                }
                b.k_max = std::max(b.k_max, n1 * n2);
                if (!(n1 * n2 <= 0x1p56)) b.bounded = false;     // (also a NaN among the vertices)
            }
        });
        double reach[3] = {0.0, 0.0, 0.0}, mag[3] = {0.0, 0.0, 0.0};
        double k_max = 0.0, e_max = 0.0;                   // max |e1|_1 |e2|_1 and the longest edge component over the triangles
        bool bounded = true;                               // no box beyond 2^60, no triangle with |e1|_1 |e2|_1 beyond 2^56, no NaN: RtParams::slab_plain
        for (const Bounds& b : part) {
            for (int a = 0; a < 3; ++a) { reach[a] = std::max(reach[a], b.reach[a]); mag[a] = std::max(mag[a], b.mag[a]); }
            k_max = std::max(k_max, b.k_max); e_max = std::max(e_max, b.e_max); bounded = bounded && b.bounded;
        }
        c->slab_plain = bounded;
        c->mesh_mag = std::max(mag[0], std::max(mag[1], mag[2])); c->mesh_emax = e_max;
        if (!bounded) c->mesh_mag = HUGE_VAL;              // (a NaN among the boxes or vertices)
        const double u = 5.9604644775390625e-8, K = k_max / 1e-5;
        double scale = 1.0;                                // RT_TIE_SCALE=<x> scales the distance (experiments: below 1 the guarantee is void)
        if (const char* e = getenv("RT_TIE_SCALE")) { const double v = atof(e); if (v > 0.0) scale = v; }
        for (int a = 0; a < 3; ++a) { c->root_center[a] = kd->center[a]; c->tie_box[a] = (float)((reach[a] + mag[a]) * 1.0000002); }
        c->tie_ulp = (float)(8.0 * u * scale);
        c->tie_rel = (float)(2.0 * u * (48.0 * K + 4.0) * scale);
        c->tie_abs = (float)(112.0 * u * K * e_max * scale);
    }
    // The front-to-back walk needs every triangle in each leaf whose box it overlaps (the reference builder lists them so,
    // KDTreeCPU.cpp:548-597).  Checked on up to 65,536 evenly spaced triangles: down from the root into every child whose box,
    // shrunk by a rounding margin, the triangle's own box overlaps; the leaves reached must list it.
    bool listing = partition;
    if (listing) {
        const uint32_t step = c->n_tris > 65536u ? c->n_tris / 65536u : 1u;
        std::vector<uint32_t> todo;
        for (uint32_t t = 0; t < c->n_tris && listing; t += step) {
            const uint32_t* ix = &c->h_tris[3 * (size_t)t];
            float lo[3], hi[3];
            for (int a = 0; a < 3; ++a) {
                const float x0 = c->h_verts[3 * (size_t)ix[0] + a], x1 = c->h_verts[3 * (size_t)ix[1] + a], x2 = c->h_verts[3 * (size_t)ix[2] + a];
                lo[a] = std::min(x0, std::min(x1, x2)); hi[a] = std::max(x0, std::max(x1, x2));
            }
            todo.clear(); todo.push_back(0);
            while (!todo.empty() && listing) {
                const uint32_t n = todo.back(); todo.pop_back();
                bool inside = true;
                for (int a = 0; a < 3 && inside; ++a) {
                    const float ce = kd->center[3 * (size_t)n + a], ex = kd->extent[3 * (size_t)n + a];
                    const float margin = 1e-5f * (std::fabs(ce) + ex) + 1e-30f;
                    inside = lo[a] < ce + ex - margin && hi[a] > ce - ex + margin;
                }
                if (!inside) continue;
                const int32_t l = kd->left[n], r = kd->right[n];
                if (l < 0 && r < 0) {
                    bool listed = false;
                    for (int32_t i = 0; i < kd->count[n] && !listed; ++i) listed = (uint32_t)kd->leaf_idx[kd->first[n] + i] == t;
                    if (!listed) listing = false;
                } else {
                    if (l >= 0) todo.push_back((uint32_t)l);
                    if (r >= 0) todo.push_back((uint32_t)r);
                }
            }
        }
    }
    lap("bounds and leaf-list check");
    c->n_nodes = M; c->n_refs = (uint32_t)n_refs; c->kd_partition = partition; c->kd_listing = listing; c->column_cache = false; c->first_hits_valid = false; c->units_valid = false;
    return RT_OK;
}

int rt_traversal_in_use(const rt_ctx* c)
{
    if (!c) return RT_E_ARG;
    // AUTO walks front to back where that is both exact and cheap: a KD tree with complete leaf lists, and a mesh whose near-tie
    // distance (RtParams::tie_rel, it grows with the square of the largest triangle) is small enough for repeats to stay rare
    const bool ordered = c->settings.traversal == RT_ORDER_FRONT_TO_BACK ||
                         (c->settings.traversal == RT_ORDER_AUTO && c->kd_partition && c->kd_listing && c->tie_rel <= kAutoMaxTieRel);
    return ordered ? RT_ORDER_FRONT_TO_BACK : RT_ORDER_REFERENCE;
}

int rt_set_camera(rt_ctx* c, const float pos[3], const float inv_proj[16], const float inv_view[16])
{
    if (!c || !pos || !inv_proj || !inv_view) return RT_E_ARG;
    // directions are regenerated only when the matrices change (Camera.cpp:96-100,113-114)
    if (!c->have_camera || memcmp(c->inv_proj, inv_proj, 64) != 0 || memcmp(c->inv_view, inv_view, 64) != 0) c->dirs_dirty = true;
    if (!c->have_camera || memcmp(c->cam_pos, pos, 12) != 0) {          // the primary rays start at the camera
        c->column_cache = false; c->first_hits_valid = false; c->units_valid = false;
    }
    memcpy(c->cam_pos, pos, 12); memcpy(c->inv_proj, inv_proj, 64); memcpy(c->inv_view, inv_view, 64);
    c->have_camera = true;
    return RT_OK;
}

int rt_set_ray_dirs(rt_ctx* c, const float* xyz, size_t n)
{
    if (!c) return RT_E_ARG;
    DeviceGuard g(c->device);
    if (!xyz) { if (c->use_host_dirs) { c->use_host_dirs = false; c->dirs_dirty = true; c->columns_dirty = true; } return RT_OK; }
    if (!c->width || n != (size_t)c->width * c->height) return fail(c, RT_E_ARG, "rt_set_ray_dirs: %zu directions for a %ux%u frame", n, c->width, c->height);
    std::vector<float4> packed(n);
    double dmax = 0.0;
    for (size_t i = 0; i < n; ++i) {
        packed[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 0.0f);
        for (int a = 0; a < 3; ++a) { const double v = std::fabs((double)xyz[3 * i + a]); if (!(v <= dmax)) dmax = v == v ? v : HUGE_VAL; }
    }
    c->host_dir_max = dmax;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(c->d_dirs.ptr, packed.data(), n * sizeof(float4), hipMemcpyHostToDevice));
    c->use_host_dirs = true; c->dirs_dirty = false; c->columns_dirty = true;
    return RT_OK;
}

int rt_set_bands(rt_ctx* c, uint32_t band_index, uint32_t band_count)
{
    if (!c || !band_count || band_index >= band_count) return fail(c, RT_E_ARG, "rt_set_bands(%u, %u)", band_index, band_count);
    if (band_index != c->band_index || band_count != c->band_count) { c->columns_dirty = true; c->unsettle(); }
    c->band_index = band_index; c->band_count = band_count;
    return RT_OK;
}

namespace {

// Camera ray directions into d_dirs when the camera or the frame size changed since they were last generated.
int ensure_dirs(rt_ctx* c, const RtParams& p)
{
    if (!c->dirs_dirty || c->use_host_dirs) return RT_OK;
    LAUNCH_TRY(c, rt_launch_raygen(&p, c->stream));
    c->dirs_dirty = false; c->columns_dirty = true;
    return RT_OK;
}

// The per-view counts that have arrived from the device (see rt_ctx::h_view) move into the host's fields, and the Renders made
// before they were known are entered into the host-side counters.  `wait`: wait for the ones still on their way.
int settle_view(rt_ctx* c, bool wait)
{
    auto arrived = [&](bool& pending, hipEvent_t ev) -> int {
        if (!pending) return 0;
        if (!wait) {
            const hipError_t q = hipEventQuery(ev);
            if (q == hipErrorNotReady) return 0;
            if (q != hipSuccess) return -1;
        } else if (hipEventSynchronize(ev) != hipSuccess) return -1;
        pending = false;
        return 1;
    };
    // (the column count of the view being set up is waited for at the one place that needs it, render_issue: `wait` is for the
    // counts of a view that is being left, which arrived long ago)
    int r = 0;
    if (c->columns_pending) {
        const hipError_t q = hipEventQuery(c->ev_columns);
        if (q == hipSuccess) { c->columns_pending = false; r = 1; }
        else if (q != hipErrorNotReady) return fail(c, RT_E_HIP, "waiting for the column count");
    }
    if (r) { c->n_columns = c->h_view[0]; c->columns_known = true; }
    r = arrived(c->units_pending, c->ev_units);
    if (r < 0) return fail(c, RT_E_HIP, "waiting for the unit counts");
    if (r) { c->n_list = c->h_view[1]; c->empty_pixels = c->h_view[2]; c->units_known = true; }
    r = arrived(c->entry_pending, c->ev_entry);
    if (r < 0) return fail(c, RT_E_HIP, "waiting for the entry counts");
    if (r) { c->entry_nodes = c->h_view[3]; c->entry_tris = c->h_view[4]; c->entry_known = true; }
    if (c->units_known && c->owed_unit_frames) { c->host_rays += (unsigned long long)c->empty_pixels * c->owed_unit_frames; c->owed_unit_frames = 0; }
    if (c->entry_known && c->owed_entry_frames) {
        c->host_entry_nodes += (unsigned long long)c->entry_nodes * c->owed_entry_frames;
        c->host_entry_tris += (unsigned long long)c->entry_tris * c->owed_entry_frames;
        c->owed_entry_frames = 0;
    }
    return RT_OK;
}

// Column rays of the current directions/bands: slots into dirs[].w.  Their number comes back behind the kernels (rt_ctx::h_view)
// and is waited for in render_issue, once the unit classification has been queued as well.
int ensure_columns(rt_ctx* c, RtParams& p)
{
    // scratch for the per-block scans: one word per 256 pixels / 256 nodes / 256 slots, whichever is most
    const size_t blocks = std::max(std::max(((size_t)c->width * c->height + 255) / 256, ((size_t)c->n_nodes + 255) / 256),
                                   (size_t)(rt_ctx::kColumnCap + 255u) / 256u);
    if (c->d_column_block.count < blocks) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, c->d_column_block.resize(blocks));
        p.column_block = c->d_column_block.ptr;
    }
    if (c->halo_in_columns != (p.halo != 0u)) c->columns_dirty = true;      // the rows bordering the bands joined or left the frame
    if (!c->columns_dirty) return RT_OK;
    c->column_cache = false; c->first_hits_valid = false; c->units_valid = false; c->halo_in_columns = p.halo != 0u;
    if (c->columns_pending) HIP_TRY(c, hipEventSynchronize(c->ev_columns));     // (the copy of the view before: long done)
    LAUNCH_TRY(c, rt_launch_column_collect(&p, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_view, c->d_column_count.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(c->ev_columns, c->stream));
    c->columns_pending = true; c->columns_known = false;
    c->columns_dirty = false;
    return RT_OK;
}

// Mesh-only scenes: which work units hold a primary ray that enters the root box (RtParams::unit_list).  Three launches and no
// wait: the trace kernel reads the length of the list on the device.  From the second Render of a view on, also the node at
// which each listed unit's primary walks may start (RtParams::unit_entry): finding those costs about what they spare one Render,
// so a view that lasts a single Render (the camera is being moved) does without.
int ensure_units(rt_ctx* c, RtParams& p, bool ordered)
{
    if (!c->units_valid) {
        if (int rc = settle_view(c, true)) return rc;       // the view before: its counts have long arrived; its Renders are entered
        const size_t pixels = (size_t)c->width * c->height;
        const size_t blocks = rt_units_blocks(p.n_units);
        if (c->d_unit_entry.count != pixels || c->d_unit_empty.count < p.n_units || c->d_unit_block.count < blocks || !c->d_unit_counts.count) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, c->d_unit_entry.resize(pixels));
            if (c->d_unit_empty.count < p.n_units) { HIP_TRY(c, c->d_unit_empty.resize(p.n_units)); HIP_TRY(c, c->d_unit_list.resize(p.n_units)); }
            if (c->d_unit_block.count < blocks) HIP_TRY(c, c->d_unit_block.resize(blocks));
            if (!c->d_unit_counts.count) HIP_TRY(c, c->d_unit_counts.resize(4));
        }
        p.unit_empty_out = c->d_unit_empty.ptr; p.unit_list_out = c->d_unit_list.ptr; p.unit_counts_out = c->d_unit_counts.ptr;
        p.unit_block_out = c->d_unit_block.ptr;
        LAUNCH_TRY(c, rt_launch_units_classify(&p, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_view + 1, c->d_unit_counts.ptr, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipEventRecord(c->ev_units, c->stream));
        c->units_pending = true; c->units_known = false; c->entry_pending = false; c->entry_known = false;
        c->n_list = 0; c->empty_pixels = 0; c->entry_nodes = 0; c->entry_tris = 0;
        c->units_valid = true; c->entry_valid = false; c->view_age = 0; c->view_setups++;
        c->unsettle();
    }
    if (!c->entry_valid && c->view_age >= c->entry_delay) {
        // where the primary walks of the listed units may start, for the walk order in use
        p.unit_list_out = c->d_unit_list.ptr; p.unit_counts_out = c->d_unit_counts.ptr;
        p.unit_entry_out = c->d_unit_entry.ptr; p.entry_ordered = ordered ? 1u : 0u;
        LAUNCH_TRY(c, rt_launch_units_entry(&p, c->units_known ? c->n_list : p.n_units, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->h_view + 3, c->d_unit_counts.ptr + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipEventRecord(c->ev_entry, c->stream));
        c->entry_pending = true; c->entry_known = false;
        c->entry_valid = true;
    }
    p.unit_list = c->d_unit_list.ptr; p.unit_counts = c->d_unit_counts.ptr; p.n_list = c->units_known ? c->n_list : 0u;
    p.unit_empty = c->d_unit_empty.ptr; p.unit_entry = c->entry_valid ? c->d_unit_entry.ptr : nullptr;
    return RT_OK;
}

// Workgroups of the persistent trace grid: what the chip holds at once, but no more than there are tiles to hand out.
int trace_grid(rt_ctx* c, int mode, bool stats, bool ordered, uint32_t n_units, unsigned& grid)
{
    int& per_cu = c->blocks_per_cu[(mode * 2 + (stats ? 1 : 0)) * 2 + (ordered ? 1 : 0)];
    if (per_cu == 0) {
        int q = 0;
        HIP_TRY(c, rt_trace_occupancy(mode, stats ? 1 : 0, ordered ? 1 : 0, &q));
        per_cu = q < 1 ? 1 : (q > (int)kMaxBlocksPerCu ? (int)kMaxBlocksPerCu : q);
        // tuning knob: RT_TRACE_BLOCKS_PER_CU=<n> overrides the occupancy query (units are pulled dynamically, so any grid is correct)
        // (never beyond 32 waves per CU: the per-wave counter slots are sized for that)
        const int max_blocks = 32 / (int)(rt_trace_block(mode) / 64u);
        if (const char* e = getenv("RT_TRACE_BLOCKS_PER_CU")) { const int v = atoi(e); if (v >= 1 && v <= max_blocks) per_cu = v; }
        if (per_cu > max_blocks) per_cu = max_blocks;
    }
    grid = (unsigned)c->n_cus * (unsigned)per_cu;
    const unsigned waves_per_block = rt_trace_block(mode) / 64u;
    const unsigned by_work = (n_units * RT_UNIT / 64u + waves_per_block - 1) / waves_per_block;   // a wave per 64 pixels is the most that can help
    if (grid > by_work) grid = by_work;
    if (grid == 0) grid = 1;
    return RT_OK;
}

}  // namespace

namespace {

// Can a dot product of triIntersect overflow for some ray of this scene?  With e the longest edge component, M the largest
// coordinate, origins within S = max(|camera|, M) + M and direction components within Dm, its intermediate values stay below
// 6 Dm e^2, 6 S Dm e and 6 S e^2, and u, v, t below 1e5 times that.  Where they can overflow, a triangle test can succeed at
// t = +inf, which KDTreeCPU::intersect reports as a hit of triangle 0 (see RT_HIT_DETECTED in rt_kernels.hip): the kernels that walk
// the tree ray by ray reproduce that; the whole-chip kernels for rays with direction.x == 0 (column pre-pass, parked rays) keep
// (t, record) keys and do not, so such a scene does without them and its waves walk those rays themselves.
bool tri_tests_stay_finite(const rt_ctx* c)
{
    const double Dm = c->use_host_dirs ? std::max(1.0, c->host_dir_max) : 1.0;
    double O = c->mesh_mag;
    for (int a = 0; a < 3; ++a) O = std::max(O, std::fabs((double)c->cam_pos[a]));
    const double S = O + c->mesh_mag, e = c->mesh_emax;
    return 6e5 * S * e * std::max(e, Dm) < 1e37 && 6.0 * Dm * e * e < 1e37;     // (false for NaNs too)
}

// One launch of the trace kernel over the units of `p` and, in trees large enough for parking, the rounds that finish the paths
// it parked (bounce rays with direction.x == 0: none in most frames, the whole centre column behind a glass sphere that sits on
// the camera's x).  Their hits by whole-chip kernels, family by family, then the rest of those paths by a small launch of the
// trace kernel -- which may park again at a later bounce, into the other list: one round per bounce at most.  How many were
// parked decides what is launched, so the host needs that one word: trace_issue queues the launch and the word's copy into
// page-locked memory and returns; trace_finish waits for the word and runs the rounds.  Between the two the caller may issue
// on other contexts (rt_group_*: every GPU of a group is at work before the host waits for any of them).
// `stop`: event recorded right after the main launch.  `parking`: out, whether trace_finish has anything to wait for.
int trace_issue(rt_ctx* c, RtParams& p, int mode, bool ordered, unsigned grid, hipEvent_t stop, bool& parking)
{
    const bool tree = mode != RT_SCENE_SPHERES;
    // With spheres and mesh together such rays are systematic (the centre column behind a sphere on the camera's x), so they are
    // parked in any but the smallest trees; in a tree-only scene they are one in 10^9 and the read-back costs more than
    // the rare walk by a wave, except in large trees.
    if (mode == RT_SCENE_BOTH && !c->park_min_forced && p.park_min_nodes > RT_PARK_MIN_NODES_MIXED) p.park_min_nodes = RT_PARK_MIN_NODES_MIXED;
    if (c->tests_may_overflow) p.park_min_nodes = 0xFFFFFFFFu;          // (see tri_tests_stay_finite)
    parking = tree && c->n_nodes >= p.park_min_nodes;
    // The park counters and the unit tickets of tree scenes start at zero for every launch that draws units.  The resolve kernel
    // that ends a Render leaves them so (rt_sum_resolve_kernel); they are filled here whenever the last launch sequence did not
    // get as far as that kernel (the launch that stores first hits, an error on the way, the very first launch).
    if (parking && !c->heads_zero) HIP_TRY(c, hipMemsetAsync(c->d_park_count.ptr, 0, 2 * sizeof(uint32_t), c->stream));
    if (tree && !c->heads_zero) HIP_TRY(c, hipMemsetAsync(c->d_work_heads.ptr, 0, RT_WORK_HEADS * RT_HEAD_STRIDE * sizeof(uint32_t), c->stream));
    if (tree) c->heads_zero = false;
    LAUNCH_TRY(c, rt_launch_trace(&p, mode, c->stats ? 1 : 0, ordered ? 1 : 0, grid, c->stream));
    if (stop) HIP_TRY(c, hipEventRecord(stop, c->stream));
    if (!parking) return RT_OK;
    HIP_TRY(c, hipMemcpyAsync(c->h_parked, c->d_park_count.ptr, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipEventRecord(c->parked_ready, c->stream));
    return RT_OK;
}

int trace_finish(rt_ctx* c, RtParams& p, int mode, bool ordered, bool parking)
{
    if (!parking) return RT_OK;
    { StallTimer st("wait for the parked count"); HIP_TRY(c, hipEventSynchronize(c->parked_ready)); }
    uint32_t parked = c->h_parked[0];
    int cur = 0;
    for (uint32_t round = 0; parked != 0; ++round) {
        const bool last = round + 1 >= c->settings.bounces;
        RtParams q = p;
        q.resume_count = c->d_park_count.ptr + cur; q.resume_paths = c->d_park_paths.ptr + (size_t)cur * RT_PARK_SLOTS * RT_PARK_WORDS;
        q.park_count = c->d_park_count.ptr + (1 - cur); q.park_paths = c->d_park_paths.ptr + (size_t)(1 - cur) * RT_PARK_SLOTS * RT_PARK_WORDS;
        HIP_TRY(c, hipMemsetAsync(q.park_count, 0, sizeof(uint32_t), c->stream));
        LAUNCH_TRY(c, rt_launch_parked_rays(&q, c->stats ? 1 : 0, c->stream));
        q.resume = last ? 2u : 1u;
        const unsigned slots = parked < RT_PARK_SLOTS ? parked : RT_PARK_SLOTS;
        // few paths per wave (see RtParams::resume_lanes), as far as the per-wave counter slots allow
        const unsigned waves_per_block = rt_trace_block(mode) / 64u, max_waves = (unsigned)(c->d_counters.count / 4);
        unsigned lanes = c->resume_lanes;
        while (lanes < 64u && (slots + lanes - 1u) / lanes > max_waves / waves_per_block * waves_per_block) lanes *= 2u;
        q.resume_lanes = lanes;
        const unsigned waves = (slots + lanes - 1u) / lanes;
        LAUNCH_TRY(c, rt_launch_trace(&q, mode, c->stats ? 1 : 0, ordered ? 1 : 0, (waves + waves_per_block - 1u) / waves_per_block, c->stream));
        if (last) break;
        cur = 1 - cur;
        HIP_TRY(c, hipMemcpyAsync(c->h_parked, c->d_park_count.ptr + cur, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        parked = c->h_parked[0];
    }
    return RT_OK;
}

}  // namespace

// A launch that has been issued and not finished (render_issue / render_finish).
struct rt_pending {
    bool active = false, parking = false, tree = false, ordered = false, timed = false, skip_empty = false, reuse = false;
    bool plain_tree = false, listed_only = false, fused = false;
    int mode = 0;
    unsigned ev = 0, gx = 0, gy = 0;
    uint32_t frame_index = 0, n = 0;
    RtParams p;
};

namespace {

// Frames frame_index .. frame_index + n - 1 in one launch of the trace kernel (tree scenes: n <= RT_MAX_LAUNCH_FRAMES): everything
// up to and including that launch.  Nothing here waits for the device while a view lasts (a new frame size or a larger tree does).
int render_issue(rt_ctx* c, uint32_t frame_index, uint32_t n, rt_pending& L)
{
    L = rt_pending();
    const int mode = c->settings.scene_mode;
    const bool tree = mode != RT_SCENE_SPHERES;
    RtParams& p = L.p; fill_params(c, p, frame_index, n);
    unsigned gx, gy; grid_for(c, gx, gy);
    if (!gy) return RT_OK;                              // this context owns no band of so small a frame
    if (int rc = settle_view(c, false)) return rc;
    // The 3x3 filter under a tile split reads the sums of the rows bordering this context's bands.  They are traced and summed
    // like its own rows while the filter is on; switched on in the middle of an accumulation they would lack the earlier frames.
    if (p.halo && frame_index != 1u && !c->halo_complete)
        return fail(c, RT_E_STATE, "anti_aliasing under a tile split needs the rows bordering this context's bands in every frame of the "
                                   "accumulation: restart it (frame index 1 or rt_reset_accum) after switching the filter on");
    if (c->band_count > 1) c->halo_complete = p.halo ? (frame_index == 1u || c->halo_complete) : false;
    if (tree) {
        const size_t plane = (size_t)c->width * c->height;
        if (c->slot_frames < n) {                       // one plane of radiance slots per Render of a launch
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->slot_frames = 0;
            hipError_t e = c->d_slots.resize(plane * n);
            if (e != hipSuccess) return fail(c, RT_E_NOMEM, "%u planes of %zu pixel slots: %s", n, plane, hipGetErrorString(e));
            c->slot_frames = n;
        }
        p.slots = c->d_slots.ptr;
    }
    { StallTimer st("ensure_dirs"); if (int rc = ensure_dirs(c, p)) return rc; }
    if (tree) { StallTimer st("ensure_columns"); if (int rc = ensure_columns(c, p)) return rc; }
    const bool ordered = tree && rt_traversal_in_use(c) == RT_ORDER_FRONT_TO_BACK;
    // Units without a primary ray into the root box are not traced (a scene without spheres; the counting kernels trace all)
    const bool skip_empty = mode == RT_SCENE_MESH && c->settings.bounces != 0u && !c->stats;
    if (skip_empty) { StallTimer st("ensure_units"); if (int rc = ensure_units(c, p, ordered)) return rc; }
    // (the number of listed units is read on the device; the host sizes the grid with it once it has arrived)
    const uint32_t n_traced_units = skip_empty && c->units_known ? c->n_list : p.n_units;
    const bool plain_tree = skip_empty && !c->settings.anti_aliasing && !p.halo;         // (see rt_ctx::empties_settled)
    if (!plain_tree) c->unsettle();
    const bool listed_only = plain_tree && c->units_known && c->empties_settled && c->image_settled[0] && frame_index != 1u && c->settled_record_ids == c->record_ids;
    unsigned grid;
    if (int rc = trace_grid(c, mode, c->stats, ordered, n_traced_units * (tree ? n : 1u), grid)) return rc;
    // rt_settings.reuse_first_hits: the primary ray of a pixel and its mesh hit are the same in every Render while camera,
    // directions, bands, tree and walk order stay as they are (TraceRay draws no random number, Renderer.cpp:232-301).  A launch of
    // the trace kernel that only traces the primary rays stores the hits when any of those changed; the Renders start from them.
    // The counting kernels and the per-wave clocks describe full walks and never use it.
    const bool reuse = tree && c->settings.reuse_first_hits && c->settings.bounces != 0u && !c->stats && !c->wave_clock_on;
    const bool store_first = reuse && !c->first_hits_valid;
    if (reuse && c->d_first_hits.count != (size_t)c->width * c->height) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipError_t e = c->d_first_hits.resize((size_t)c->width * c->height);
        if (e != hipSuccess) return fail(c, RT_E_NOMEM, "first-hit buffer of %zu pixels: %s", (size_t)c->width * c->height, hipGetErrorString(e));
    }
    // events: [pre, start) = what precedes the trace kernel on mesh scenes (the column pre-pass when the camera has direction.x == +0
    // pixels, the launch that stores first hits), [start, stop) = the trace kernel
    // (only while someone asks for kernel times, and then around every timing_every-th launch: the two event records of a launch cost
    // a 1080p sphere frame 8 us of its 150, the markers keep the next launch from being set up behind the running one)
    // (launch n / 2 of every n: not the first after a reset, which starts on an idle chip and takes longer to get going)
    const bool timed = c->timing_every != 0u && c->timing_count++ % c->timing_every == c->timing_every / 2u;
    const unsigned ev = c->ev_head % rt_ctx::EV_RING;
    c->tests_may_overflow = tree && !tri_tests_stay_finite(c);
    // Whether the view has column rays decides whether the pre-pass is launched (over a large tree even kernels that return at once
    // cost more than the wait): the one word is waited for here, after the unit classification has been queued behind it, so the
    // device has work while the host wakes up.  Once per view.
    if (tree && c->columns_pending) {
        StallTimer st("wait for the column count");
        HIP_TRY(c, hipEventSynchronize(c->ev_columns));
        c->columns_pending = false; c->n_columns = c->h_view[0]; c->columns_known = true;
    }
    const bool columns = tree && (!c->columns_known || c->n_columns) && !c->tests_may_overflow && (!reuse || store_first);     // (stored hits answer the column rays too)
    // Which triangle a column ray hits depends on the camera, the directions, the bands and the tree only: the pre-pass runs when
    // one of those changed and its result (the best hit per column ray) is kept.  The counting kernels always run it: their
    // counters describe what the reference's walk costs for these rays.
    const bool prepass = columns && (!c->column_cache || c->stats);
    if (timed) c->ev_has_pre[ev] = prepass || store_first;
    if (timed && c->ev_has_pre[ev]) HIP_TRY(c, hipEventRecord(c->ev_pre[ev], c->stream));
    if (prepass) {
        LAUNCH_TRY(c, rt_launch_column_prepass(&p, c->stats ? 1 : 0, c->stream));
        c->column_cache = true;
    }
    if (!columns) p.column_best = nullptr;
    if (store_first) {
        RtParams q = p;
        q.n_frames = 1; q.first_hits = c->d_first_hits.ptr; q.first_hit_mode = 1u; q.primary_ids = nullptr;
        unsigned grid1;
        if (int rc = trace_grid(c, mode, false, ordered, n_traced_units, grid1)) return rc;
        bool parking1 = false;
        if (int rc = trace_issue(c, q, mode, ordered, grid1, nullptr, parking1)) return rc;
        if (int rc = trace_finish(c, q, mode, ordered, parking1)) return rc;
        c->first_hits_valid = true; c->view_setups++;
    }
    if (reuse) { p.first_hits = c->d_first_hits.ptr; p.first_hit_mode = 2u; }
    // sphere frames resolve inside the trace kernel; the 3x3 filter needs the neighbours' sums and keeps the separate pass
    const bool fused = !tree && !c->settings.anti_aliasing;
    p.fuse_resolve = fused ? 1u : 0u;
    if (timed) HIP_TRY(c, hipEventRecord(c->ev_start[ev], c->stream));
    c->last_grid_waves = grid * (rt_trace_block(mode) / 64u);
    bool parking = false;
    { StallTimer st("trace_issue"); if (int rc = trace_issue(c, p, mode, ordered, grid, timed ? c->ev_stop[ev] : nullptr, parking)) return rc; }
    L.active = true; L.parking = parking; L.tree = tree; L.ordered = ordered; L.timed = timed; L.skip_empty = skip_empty; L.reuse = reuse;
    L.plain_tree = plain_tree; L.listed_only = listed_only; L.fused = fused; L.mode = mode; L.ev = ev; L.gx = gx; L.gy = gy;
    L.frame_index = frame_index; L.n = n;
    return RT_OK;
}

// The rest of the launch: the paths it parked, the sums and the image.
int render_finish(rt_ctx* c, rt_pending& L)
{
    if (!L.active) return RT_OK;
    L.active = false;
    RtParams& p = L.p;
    const uint32_t n = L.n, frame_index = L.frame_index;
    if (int rc = trace_finish(c, p, L.mode, L.ordered, L.parking)) return rc;
    if (L.timed) { c->ev_frames[L.ev] = n; c->ev_head++; }
    if (L.skip_empty) {
        c->view_age++;
        c->owed_unit_frames += n;
        if (!L.reuse && p.unit_entry) c->owed_entry_frames += n;
    }
    const uint32_t frame_last = frame_index + n - 1u;
    c->latest_in_back = false;
    if (L.tree) {
        LAUNCH_TRY(c, rt_launch_sum_resolve(&p, frame_last, (c->settings.anti_aliasing ? 0 : 1) | (L.listed_only ? 2 : 0), c->stream));
        c->heads_zero = true;
        // a pass over every unit leaves the units without a path with their sum of +0, their ids and, in the buffer it wrote, their image
        if (L.plain_tree && frame_index == 1u) c->empties_settled = true;
        if (L.plain_tree && c->empties_settled && !L.listed_only) { c->image_settled[0] = true; c->settled_record_ids = c->record_ids; }
    }
    else if (!L.fused) LAUNCH_TRY(c, rt_launch_resolve(&p, frame_last, L.gx, L.gy, c->stream));
    if (c->settings.anti_aliasing) LAUNCH_TRY(c, rt_launch_aa(&p, frame_last, L.gx, L.gy, c->stream));
    return RT_OK;
}

// What every Render call checks first.
int render_checks(rt_ctx* c, uint32_t frame_index, uint32_t n)
{
    if (!frame_index || !n || frame_index + n < frame_index) return fail(c, RT_E_ARG, "rt_render: frame_index %u, n %u", frame_index, n);
    if (!c->width) return fail(c, RT_E_STATE, "rt_render before rt_resize");
    if (!c->have_camera) return fail(c, RT_E_STATE, "rt_render before rt_set_camera (the camera position is needed even with rt_set_ray_dirs)");
    return validate_scene(c, c->settings.scene_mode);
}

}  // namespace

int rt_render_frames(rt_ctx* c, uint32_t frame_index, uint32_t n)
{
    if (!c) return RT_E_ARG;
    if (int rc = render_checks(c, frame_index, n)) return rc;
    DeviceGuard g(c->device);
    StallTimer whole("rt_render_frames");
    if (int rc = collect_time(c, false)) return rc;
    // sphere scenes: a lane keeps its pixel for all n frames, whatever n; tree scenes: at most RT_MAX_LAUNCH_FRAMES planes of slots
    const uint32_t per_launch = c->settings.scene_mode == RT_SCENE_SPHERES ? n : (uint32_t)RT_MAX_LAUNCH_FRAMES;
    rt_pending L;
    for (uint32_t done = 0; done < n; done += per_launch) {
        const uint32_t k = n - done < per_launch ? n - done : per_launch;
        if (int rc = render_issue(c, frame_index + done, k, L)) return rc;
        if (int rc = render_finish(c, L)) return rc;
    }
    return RT_OK;
}

int rt_render(rt_ctx* c, uint32_t frame_index) { return rt_render_frames(c, frame_index, 1); }

int rt_resolve(rt_ctx* c, uint32_t frame_index)
{
    if (!c || !frame_index) return RT_E_ARG;
    if (!c->width) return fail(c, RT_E_STATE, "rt_resolve before rt_resize");
    DeviceGuard g(c->device);
    RtParams p; fill_params(c, p, frame_index, 1);
    unsigned gx, gy; grid_for(c, gx, gy);
    if (!gy) return RT_OK;
    c->latest_in_back = false; c->unsettle();
    LAUNCH_TRY(c, rt_launch_resolve(&p, frame_index, gx, gy, c->stream));
    if (c->settings.anti_aliasing) LAUNCH_TRY(c, rt_launch_aa(&p, frame_index, gx, gy, c->stream));
    return RT_OK;
}

int rt_reset_accum(rt_ctx* c)
{
    if (!c) return RT_E_ARG;
    if (!c->width) return fail(c, RT_E_STATE, "rt_reset_accum before rt_resize");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipMemsetAsync(c->d_accum.ptr, 0, (size_t)c->width * c->height * sizeof(float4), c->stream));
    c->halo_complete = true;                             // an empty sum is complete for every row
    c->unsettle();
    return RT_OK;
}

int rt_sync(rt_ctx* c)
{
    if (!c) return RT_E_ARG;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return collect_time(c, true);
}

int rt_read_image(rt_ctx* c, float* out)
{
    if (!c || !out) return RT_E_ARG;
    if (!c->width) return fail(c, RT_E_STATE, "no frame");
    DeviceGuard g(c->device);
    const float4* frame = c->latest_in_back ? c->d_image_back.ptr : c->d_image.ptr;
    HIP_TRY(c, hipMemcpyAsync(out, frame, (size_t)c->width * c->height * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    return rt_sync(c);
}

namespace {

// The two page-locked frame buffers, sized for the current frame.
int ensure_host_frames(rt_ctx* c, size_t bytes)
{
    if (c->h_image_bytes == bytes && c->h_image[0] && c->h_image[1]) return RT_OK;
    (void)hipStreamSynchronize(c->copy_stream);
    for (int i = 0; i < 2; ++i) {
        if (c->h_image[i]) { (void)hipHostFree(c->h_image[i]); c->h_image[i] = nullptr; }
        c->h_image_pending[i] = false;
    }
    c->h_image_bytes = 0;
    for (int i = 0; i < 2; ++i) {
        hipError_t e = hipHostMalloc((void**)&c->h_image[i], bytes, hipHostMallocDefault);
        if (e != hipSuccess) { c->h_image[i] = nullptr; return fail(c, RT_E_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
    }
    c->h_image_bytes = bytes;
    return RT_OK;
}

}  // namespace

const float* rt_map_image(rt_ctx* c)
{
    if (!c) return nullptr;
    if (!c->width) { fail(c, RT_E_STATE, "no frame"); return nullptr; }
    DeviceGuard g(c->device);
    const size_t bytes = (size_t)c->width * c->height * sizeof(float4);
    if (ensure_host_frames(c, bytes) != RT_OK) return nullptr;
    if (c->h_image_pending[0] || c->h_image_pending[1]) {          // copies of a pipelined hand-off still under way
        if (hipStreamSynchronize(c->copy_stream) != hipSuccess) { fail(c, RT_E_HIP, "hipStreamSynchronize(copy stream)"); return nullptr; }
        c->h_image_pending[0] = c->h_image_pending[1] = false;
    }
    hipError_t e = hipMemcpyAsync(c->h_image[0], c->latest_in_back ? c->d_image_back.ptr : c->d_image.ptr, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) { fail(c, RT_E_HIP, "hipMemcpyAsync: %s", hipGetErrorString(e)); return nullptr; }
    if (rt_sync(c) != RT_OK) return nullptr;
    return c->h_image[0];
}

const float* rt_map_image_pipelined(rt_ctx* c)
{
    if (!c) return nullptr;
    if (!c->width) { fail(c, RT_E_STATE, "no frame"); return nullptr; }
    DeviceGuard g(c->device);
    const size_t n = (size_t)c->width * c->height, bytes = n * sizeof(float4);
    if (ensure_host_frames(c, bytes) != RT_OK) return nullptr;
    if (c->d_image_back.count != n) {
        if (hipStreamSynchronize(c->stream) != hipSuccess || c->d_image_back.resize(n) != hipSuccess) {
            fail(c, RT_E_NOMEM, "second frame buffer of %zu bytes", bytes); return nullptr;
        }
        c->image_settled[1] = false;                     // fresh memory: no unit of it holds a frame
    }
    // The finished frame becomes the back buffer and is copied out over the copy stream; later Renders resolve into the other
    // buffer, whose own copy (two calls ago) was waited for before that call returned.
    const unsigned i = c->h_image_next & 1u;
    hipError_t e = hipEventRecord(c->frame_ready, c->stream);
    std::swap(c->d_image.ptr, c->d_image_back.ptr);
    std::swap(c->image_settled[0], c->image_settled[1]);      // (see rt_ctx::image_settled)
    c->latest_in_back = true;
    if (e == hipSuccess) e = hipStreamWaitEvent(c->copy_stream, c->frame_ready, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(c->h_image[i], c->d_image_back.ptr, bytes, hipMemcpyDeviceToHost, c->copy_stream);
    if (e == hipSuccess) e = hipEventRecord(c->h_image_done[i], c->copy_stream);
    if (e != hipSuccess) { fail(c, RT_E_HIP, "pipelined frame copy: %s", hipGetErrorString(e)); return nullptr; }
    c->h_image_pending[i] = true;
    c->h_image_next++;
    const unsigned ret = c->h_image_pending[1u - i] ? 1u - i : i;   // the frame of the call before; the very first call waits for its own
    if (hipEventSynchronize(c->h_image_done[ret]) != hipSuccess) { fail(c, RT_E_HIP, "hipEventSynchronize(frame copy)"); return nullptr; }
    if (ret != i) c->h_image_pending[ret] = false;
    if (collect_time(c, false) != RT_OK) return nullptr;
    return c->h_image[ret];
}

int rt_read_accum(rt_ctx* c, float* out_rgb)
{
    if (!c || !out_rgb) return RT_E_ARG;
    if (!c->width) return fail(c, RT_E_STATE, "no frame");
    DeviceGuard g(c->device);
    const size_t n = (size_t)c->width * c->height;
    std::vector<float4> tmp(n);
    HIP_TRY(c, hipMemcpyAsync(tmp.data(), c->d_accum.ptr, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (int rc = rt_sync(c)) return rc;
    for (size_t i = 0; i < n; ++i) { out_rgb[3 * i] = tmp[i].x; out_rgb[3 * i + 1] = tmp[i].y; out_rgb[3 * i + 2] = tmp[i].z; }
    return RT_OK;
}

int rt_set_record_ids(rt_ctx* c, int on) { if (!c) return RT_E_ARG; c->record_ids = on != 0; return RT_OK; }

int rt_read_primary_ids(rt_ctx* c, int32_t* out)
{
    if (!c || !out) return RT_E_ARG;
    if (!c->width) return fail(c, RT_E_STATE, "no frame");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipMemcpyAsync(out, c->d_ids.ptr, (size_t)c->width * c->height * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    return rt_sync(c);
}

int rt_read_ray_dirs(rt_ctx* c, float* out_xyz)
{
    if (!c || !out_xyz) return RT_E_ARG;
    if (!c->width || (!c->have_camera && !c->use_host_dirs)) return fail(c, RT_E_STATE, "rt_read_ray_dirs before rt_resize/rt_set_camera");
    DeviceGuard g(c->device);
    const size_t n = (size_t)c->width * c->height;
    RtParams p; fill_params(c, p, 1, 1);
    if (int rc = ensure_dirs(c, p)) return rc;
    std::vector<float4> tmp(n);
    HIP_TRY(c, hipMemcpyAsync(tmp.data(), c->d_dirs.ptr, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (int rc = rt_sync(c)) return rc;
    for (size_t i = 0; i < n; ++i) { out_xyz[3 * i] = tmp[i].x; out_xyz[3 * i + 1] = tmp[i].y; out_xyz[3 * i + 2] = tmp[i].z; }
    return RT_OK;
}

int rt_set_stats(rt_ctx* c, int on) { if (!c) return RT_E_ARG; c->stats = (on & 1) != 0; c->wave_clock_on = (on & 2) != 0; return RT_OK; }

int rt_read_counters(rt_ctx* c, rt_counters* out)
{
    if (!c || !out) return RT_E_ARG;
    memset(out, 0, sizeof *out);
    if (!c->d_counters.count) return RT_OK;
    DeviceGuard g(c->device);
    std::vector<unsigned long long> h(c->d_counters.count);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->d_counters.ptr, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    if (int rc = rt_sync(c)) return rc;
    if (int rc = settle_view(c, true)) return rc;
    for (size_t i = 0; i < h.size(); i += 4) {
        out->rays += h[i]; out->nodes_visited += h[i + 1]; out->tris_tested += h[i + 2]; out->sphere_tests += h[i + 3];
    }
    unsigned long long cs[5] = {0, 0, 0, 0, 0};
    HIP_TRY(c, hipMemcpy(cs, c->d_column_stats.ptr, sizeof cs, hipMemcpyDeviceToHost));
    if (cs[4]) return fail(c, RT_E_STATE, "rt_read_counters: a kernel's index test failed (code %llu, see RT_CHECKED)", cs[4]);
    out->prepass_nodes = cs[0]; out->prepass_tris = cs[1]; out->near_ties = cs[2]; out->primary_hits_reused = cs[3];
    out->nodes_visited += cs[0]; out->tris_tested += cs[1];
    out->rays += c->host_rays; out->empty_unit_rays = c->host_rays; out->view_setups = c->view_setups;
    out->entry_nodes_spared = c->host_entry_nodes; out->entry_tris_spared = c->host_entry_tris;
    return RT_OK;
}

int rt_reset_counters(rt_ctx* c)
{
    if (!c) return RT_E_ARG;
    if (!c->d_counters.count) return RT_OK;
    DeviceGuard g(c->device);
    HIP_TRY(c, hipMemsetAsync(c->d_counters.ptr, 0, c->d_counters.count * sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_column_stats.ptr, 0, 8 * sizeof(unsigned long long), c->stream));
    c->host_rays = 0; c->view_setups = 0; c->host_entry_nodes = 0; c->host_entry_tris = 0; c->owed_unit_frames = 0; c->owed_entry_frames = 0;
    return RT_OK;
}

int rt_read_wave_clock(rt_ctx* c, uint64_t* out, uint32_t max_waves, uint32_t* n_waves)
{
    if (!c || !out || !n_waves) return RT_E_ARG;
    DeviceGuard g(c->device);
    if (int rc = rt_sync(c)) return rc;
    uint32_t n = c->last_grid_waves < max_waves ? c->last_grid_waves : max_waves;
    if ((size_t)n * RT_WAVE_CLOCK_WORDS > c->d_wave_clock.count) n = (uint32_t)(c->d_wave_clock.count / RT_WAVE_CLOCK_WORDS);
    if (n) HIP_TRY(c, hipMemcpy(out, c->d_wave_clock.ptr, (size_t)n * RT_WAVE_CLOCK_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost));
    *n_waves = n;
    return RT_OK;
}

int rt_get_kernel_time(rt_ctx* c, rt_kernel_time* out)
{
    if (!c || !out) return RT_E_ARG;
    DeviceGuard g(c->device);
    if (int rc = collect_time(c, true)) return rc;
    *out = c->ktime;
    if (!c->timing_every) c->timing_every = 1;          // someone asks: from now on the launches are timed
    return RT_OK;
}

int rt_set_kernel_timing(rt_ctx* c, uint32_t every)
{
    if (!c) return RT_E_ARG;
    c->timing_every = every; c->timing_count = 0;
    return RT_OK;
}

int rt_reset_kernel_time(rt_ctx* c)
{
    if (!c) return RT_E_ARG;
    DeviceGuard g(c->device);
    if (int rc = collect_time(c, true)) return rc;
    c->ktime = rt_kernel_time{0, 0.0, 0.0, 0.0, 0};
    if (!c->timing_every) c->timing_every = 1;
    c->timing_count = 0;
    return RT_OK;
}

// ---- for rt_group.hip (not part of the C ABI) ----
// One launch of rt_render_frames in two halves, so that a group can issue on every context before it finishes any.
// rt_internal_launch_frames: the Renders one launch of this context holds at most (0 = any number).
__attribute__((visibility("hidden"))) uint32_t rt_internal_launch_frames(const rt_ctx* c) { return c->settings.scene_mode == RT_SCENE_SPHERES ? 0u : (uint32_t)RT_MAX_LAUNCH_FRAMES; }
__attribute__((visibility("hidden"))) rt_pending* rt_internal_pending_new() { return new rt_pending(); }
__attribute__((visibility("hidden"))) void rt_internal_pending_free(rt_pending* L) { delete L; }
__attribute__((visibility("hidden"))) int rt_internal_render_issue(rt_ctx* c, uint32_t frame_index, uint32_t n, rt_pending* L)
{
    if (int rc = render_checks(c, frame_index, n)) return rc;
    DeviceGuard g(c->device);
    StallTimer whole("rt_render_frames");
    if (int rc = collect_time(c, false)) return rc;
    return render_issue(c, frame_index, n, *L);
}
__attribute__((visibility("hidden"))) int rt_internal_render_finish(rt_ctx* c, rt_pending* L)
{
    DeviceGuard g(c->device);
    return render_finish(c, *L);
}
// 1: the issued launch parks paths and its main trace kernel has not finished yet; 0: it has, or the launch needs no word back.
__attribute__((visibility("hidden"))) int rt_internal_issue_in_flight(rt_ctx* c, const rt_pending* L)
{
    if (!L->active || !L->parking) return 0;
    DeviceGuard g(c->device);
    return hipEventQuery(c->parked_ready) == hipErrorNotReady ? 1 : 0;
}
__attribute__((visibility("hidden"))) hipStream_t rt_internal_stream(rt_ctx* c) { return c->stream; }
__attribute__((visibility("hidden"))) int rt_internal_device(const rt_ctx* c) { return c->device; }
__attribute__((visibility("hidden"))) void rt_internal_size(const rt_ctx* c, uint32_t* w, uint32_t* h) { *w = c->width; *h = c->height; }
// Resolve (and filter) the frame from sums that are not this context's own buffer: the sum over a group's contexts.
__attribute__((visibility("hidden"))) int rt_internal_resolve_from(rt_ctx* c, const void* accum_float4, uint32_t frame_index)
{
    if (!c->width || !frame_index) return fail(c, RT_E_STATE, "resolve before rt_resize");
    DeviceGuard g(c->device);
    RtParams p; fill_params(c, p, frame_index, 1);
    p.accum = const_cast<float4*>(static_cast<const float4*>(accum_float4));
    unsigned gx, gy; grid_for(c, gx, gy);
    if (!gy) return RT_OK;
    c->latest_in_back = false; c->unsettle();
    LAUNCH_TRY(c, rt_launch_resolve(&p, frame_index, gx, gy, c->stream));
    if (c->settings.anti_aliasing) LAUNCH_TRY(c, rt_launch_aa(&p, frame_index, gx, gy, c->stream));
    return RT_OK;
}

// ---- experiments (see include/rt_mi355x.h) ----
int rt_debug_dump_rays(rt_ctx* c, uint32_t frame_index, float* out_rays8, uint32_t cap, uint32_t* n_out)
{
    if (!c || !out_rays8 || !cap || !n_out) return RT_E_ARG;
    if (c->settings.scene_mode == RT_SCENE_SPHERES) return fail(c, RT_E_STATE, "rt_debug_dump_rays: a scene with a tree");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, c->d_ray_dump.resize(2 * (size_t)cap));
    HIP_TRY(c, c->d_ray_dump_count.resize(1));
    HIP_TRY(c, hipMemsetAsync(c->d_ray_dump_count.ptr, 0, sizeof(uint32_t), c->stream));
    const bool stats = c->stats;
    c->stats = true; c->dump_on = true;
    const int rc = rt_render(c, frame_index);
    c->stats = stats; c->dump_on = false;
    if (rc) return rc;
    uint32_t n = 0;
    HIP_TRY(c, hipMemcpyAsync(&n, c->d_ray_dump_count.ptr, sizeof n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_out = n;
    if (n > cap) n = cap;
    HIP_TRY(c, hipMemcpy(out_rays8, c->d_ray_dump.ptr, (size_t)n * 8 * sizeof(float), hipMemcpyDeviceToHost));
    c->d_ray_dump.release();
    return RT_OK;
}

int rt_debug_walk_rays(rt_ctx* c, const float* rays8, uint32_t n, int variant, int refill, uint32_t repeats, float* out_hits4, float* out_ms)
{
    if (!c || !rays8 || !n || !repeats || !out_ms) return RT_E_ARG;
    if (!c->n_nodes) return fail(c, RT_E_STATE, "rt_debug_walk_rays: no tree");
    DeviceGuard g(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    DevBuf<float4> d_rays, d_out; DevBuf<uint32_t> d_next;
    struct Free { DevBuf<float4>&a, &b; DevBuf<uint32_t>& n; ~Free() { a.release(); b.release(); n.release(); } } freer{d_rays, d_out, d_next};
    HIP_TRY(c, d_rays.resize(2 * (size_t)n)); HIP_TRY(c, d_out.resize(n)); HIP_TRY(c, d_next.resize(1));
    HIP_TRY(c, hipMemcpy(d_rays.ptr, rays8, (size_t)n * 8 * sizeof(float), hipMemcpyHostToDevice));
    RtParams p; fill_params(c, p, 1, 1);
    unsigned grid = (n + 255u) / 256u;
    if (refill) {
        int per_cu = 0;
        HIP_TRY(c, rt_probe_walk_occupancy(&per_cu));
        const unsigned full = (unsigned)c->n_cus * (unsigned)(per_cu < 1 ? 1 : per_cu);
        if (grid > full) grid = full;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(c, hipEventCreate(&e0)); HIP_TRY(c, hipEventCreate(&e1));
    double total = 0.0;
    int rc = RT_OK;
    for (uint32_t r = 0; r <= repeats && rc == RT_OK; ++r) {             // (the first launch warms the caches and is not counted)
        hipError_t e = hipMemsetAsync(d_next.ptr, 0, sizeof(uint32_t), c->stream);
        if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
        if (e == hipSuccess) e = rt_launch_probe_walk(&p, d_rays.ptr, n, d_next.ptr, refill, variant, grid, d_out.ptr, c->stream);
        if (e == hipSuccess) e = hipEventRecord(e1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = fail(c, RT_E_HIP, "rt_debug_walk_rays: %s", hipGetErrorString(e));
        if (r) total += ms;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc) return rc;
    *out_ms = (float)(total / repeats);
    if (out_hits4) HIP_TRY(c, hipMemcpy(out_hits4, d_out.ptr, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost));
    return RT_OK;
}

void* rt_device_accum(rt_ctx* c) { return c ? (void*)c->d_accum.ptr : nullptr; }
void* rt_device_image(rt_ctx* c) { return c ? (void*)(c->latest_in_back ? c->d_image_back.ptr : c->d_image.ptr) : nullptr; }

}  // extern "C"
