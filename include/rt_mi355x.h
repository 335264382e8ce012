/*
 * rt_mi355x.h -- C ABI of the MI355X (gfx950) implementation of the Soulztorm/Raytracer per-pixel tracing path.
 *
 * The reference has no plugin/FFI layer: its seam is the C++ class `Renderer`
 * (RayTracer/src/Renderer.h:101-161) driven by one caller (RayTracer/src/RaytracerApp.cpp:259-366).  The entry
 * points below are what a drop-in `Renderer` backed by the GPU binds; each one names the reference member or
 * data it replaces.  Plain pointers and sizes only; the caller owns every host pointer (it is copied before
 * the call returns).  A context is not thread-safe and belongs to one GPU.
 *
 * Return value: 0 = ok, negative = error (RT_E_*); rt_last_error() gives the message.
 * Citations are file:line in the reference tree.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum {
    RT_OK = 0,
    RT_E_ARG = -1,          /* bad argument (null, out-of-range index, size mismatch) */
    RT_E_STATE = -2,        /* call order (render before resize/camera/scene) */
    RT_E_HIP = -3,          /* a HIP runtime call failed */
    RT_E_NOMEM = -4,
    RT_E_LIMIT = -5         /* beyond a stated fixed limit */
};

/* Sphere (RayTracer/src/Scene.h:10-15). */
typedef struct { float x, y, z, radius; uint32_t material; } rt_sphere;

/* Material without the display name (RayTracer/src/Scene.h:26-35). */
typedef struct { float albedo[3]; float emission[3]; float roughness, transparency, ior; } rt_material;

/* Renderer::Settings (RayTracer/src/Renderer.h:103-110) as plain ints.  `accumulate` is carried for completeness only: the caller
   decides the next frame index from it (Renderer.cpp:128-131) and passes that to rt_render.  `scene_mode` carries UseSphereScene
   (0 = mesh through the KD tree, 1 = spheres) and adds 2 = both, nearest wins by the strict sphere-first compare
   of Renderer.cpp:292-300.  `seed` seeds the per-(pixel, frame) pcg32 streams (the value Walnut::Random::Init
   intended, Random.h:22, is 1337).  `traversal` (no counterpart in the reference) selects the order in which the KD
   tree is walked, see RT_ORDER_*.  `reuse_first_hits` (no counterpart in the reference; 0 = off, the default): mesh scenes keep
   the primary ray's closest triangle hit per pixel for as long as camera, directions, bands, tree and walk order stay as they
   are, and every Render starts its paths from it instead of walking the tree for bounce 0 -- the ray and the tree are the
   same and TraceRay draws no random number (Renderer.cpp:141-154,232-301), so every frame keeps its bits; the hits are found by
   one extra launch (primary rays only) after any of those changed.  rt_counters.rays still counts those TraceRay invocations,
   rt_counters.primary_hits_reused says how many of them were answered from the kept hits. */
typedef struct {
    int32_t  accumulate;
    int32_t  scene_mode;
    int32_t  use_ace;
    int32_t  anti_aliasing;
    uint32_t bounces;
    uint32_t seed;
    int32_t  traversal;
    int32_t  reuse_first_hits;
} rt_settings;

enum { RT_SCENE_MESH = 0, RT_SCENE_SPHERES = 1, RT_SCENE_BOTH = 2 };

/* Tree walk order.
   RT_ORDER_REFERENCE: exactly KDTreeCPU::intersect (KDAccel/KDTreeCPU.cpp:649-708): box test, left subtree, right
     subtree, a subtree skipped when its box's t_near exceeds the closest t so far.  Same visited nodes and tested
     triangles as the reference, ray by ray (the nodes_visited / tris_tested counters equal the reference's).
   RT_ORDER_FRONT_TO_BACK: the same box and triangle tests and the same pruning rule, but at every node the child on
     the ray's near side is walked first (eight node arrays in HBM, one per sign pattern of the ray direction, so the
     walk stays a stackless "next node or skip link" loop).  Result = the smallest t over the leaves whose chain of
     boxes the ray hits, equal t resolved to the triangle the reference order reaches first.  That is the reference's
     result whenever every triangle is listed in each leaf whose box it overlaps (the reference builder,
     KDTreeCPU.cpp:548-597, guarantees it), except when two different triangles are hit within rounding distance of
     each other: then the reference's answer depends on which leaves its own order happened to prune.  The walk notices
     such near ties (it prunes with the closest t plus that distance, so it sees both triangles) and repeats the ray
     with KDTreeCPU::intersect's own sequence of decisions; rt_counters.near_ties counts them.  Far fewer nodes and
     triangles per ray.
   RT_ORDER_AUTO (default): front to back when the uploaded tree's two-child nodes split their box in two along one
     axis (a KD tree, as the reference builds), otherwise the reference order. */
enum { RT_ORDER_AUTO = 0, RT_ORDER_REFERENCE = 1, RT_ORDER_FRONT_TO_BACK = 2 };

/* Host KD tree flattened for upload: what a walk of KDTreeCPU::getRootNode() (KDAccel/KDTreeCPU.cpp:73-76,
   node fields KDAccel/KDTreeStructs.h:55-76) emits.  Children are node indices, -1 = none (a node may have
   one child, KDTreeCPU.cpp:600-620); a node with no child is a leaf and owns leaf_idx[first .. first+count).
   first/count of interior nodes are ignored.  Node 0 is the root. */
typedef struct {
    uint32_t       n_nodes;
    const float*   center;      /* n_nodes * 3, boundingBox::center  */
    const float*   extent;      /* n_nodes * 3, boundingBox::extends */
    const int32_t* left;
    const int32_t* right;
    const int32_t* first;
    const int32_t* count;
    uint32_t       n_leaf_idx;
    const int32_t* leaf_idx;    /* triangle indices */
} rt_kd_flat;

/* Work counters; a ray is one TraceRay invocation (Renderer.cpp:154). nodes_visited / tris_tested are only
   maintained while rt_set_stats(ctx, 1); in RT_ORDER_REFERENCE they equal what KDTreeCPU::intersect visits and tests
   for the same rays.  prepass_nodes / prepass_tris are the part of those two sums that belongs to primary rays with
   direction.x == +0, which a per-frame pre-pass resolves without walking the tree ray by ray: they are what the
   reference's walk costs for those rays, not operations the GPU performed.  near_ties (always maintained): rays of the
   front-to-back walk that hit two different triangles within rounding distance of each other and were therefore walked
   again in the reference's order (see rt_settings.traversal).  primary_hits_reused (always maintained): primary rays, part of
   `rays`, whose hit came from the kept first hits (rt_settings.reuse_first_hits) instead of a walk.  empty_unit_rays (always
   maintained): primary rays, part of `rays`, of 16-pixel work units none of whose primary rays enters the tree's root box -- in a
   scene without spheres those paths end in Miss() with a colour of zero (Renderer.cpp:160-165) and are not traced at all (found
   once per camera, directions, bands and tree; the counting mode of rt_set_stats traces them like any other).  view_setups:
   how often that per-view data (the list of units to trace, the kept first hits) was built since the last reset -- it should
   stay put while the same camera and scene are re-sent frame after frame.  entry_nodes_spared / entry_tris_spared: node steps
   and triangle tests of primary walks that were not made because the sixteen primary rays of a work unit share the first
   decisions of their walks (same origin, neighbouring directions) and start at the node where they part -- up to there
   none of them has hit anything (mesh-only scenes; found with the unit list; not in counting mode). */
typedef struct { uint64_t rays, nodes_visited, tris_tested, sphere_tests, prepass_nodes, prepass_tris, near_ties, primary_hits_reused,
                 empty_unit_rays, view_setups, entry_nodes_spared, entry_tris_spared; } rt_counters;

/* Device time measured with HIP events on the stream the kernels were launched on: total_ms / last_ms over the
   trace kernel (the dominant kernel) alone, prepass_total_ms over the pre-pass kernels that precede it on mesh
   scenes whose camera has direction.x == +0 pixels; frames = Renders those launches held (rt_render_frames puts several in one). */
typedef struct { uint64_t launches; double total_ms; double last_ms; double prepass_total_ms; uint64_t frames; } rt_kernel_time;

/* ---- lifetime ------------------------------------------------------------------------------- */
/* Renderer::Renderer() (Renderer.cpp:16-18) on HIP device `device_id`. Returns NULL on failure. */
rt_ctx*     rt_create(int device_id);
void        rt_destroy(rt_ctx*);
const char* rt_last_error(const rt_ctx*);          /* ctx may be NULL: this thread's last error from a call without a context */
/* Run the context's work on `hip_stream` (a hipStream_t) instead of its own stream; NULL restores its own. */
int         rt_set_stream(rt_ctx*, void* hip_stream);

/* ---- per-frame inputs ------------------------------------------------------------------------ */
/* Renderer::OnResize (Renderer.cpp:20-43): (re)allocates accumulation + image, no-op when unchanged. */
int rt_resize(rt_ctx*, uint32_t width, uint32_t height);
/* Renderer::GetSettings() fields (Renderer.h:103-111). */
int rt_set_settings(rt_ctx*, const rt_settings*);
/* Scene::spheres / Scene::materials (Scene.h:37-40); re-sent every frame by the UI (RaytracerApp.cpp:283-331). */
int rt_set_spheres(rt_ctx*, const rt_sphere*, uint32_t n);
int rt_set_materials(rt_ctx*, const rt_material*, uint32_t n);
/* Scene::triangles + the KDTreeCPU vertex/index arrays (Scene.h:17-24,41; KDTreeCPU.cpp:26-34): verts nverts*3,
   tri_idx ntris*3, normals9 ntris*9 (Triangle::Normals[0..2]), tri_mat ntris.  Uploaded once. */
int rt_set_mesh(rt_ctx*, const float* verts, uint32_t nverts, const uint32_t* tri_idx,
                const float* normals9, const uint32_t* tri_mat, uint32_t ntris);
/* Scene::kd_tree (Scene.h:43): the flattened host tree; requires rt_set_mesh first.
   RT_ORDER_AUTO / RT_ORDER_FRONT_TO_BACK give the reference's hits only for a tree that lists every triangle in each leaf whose
   box it overlaps, as the reference builder does (KDTreeCPU.cpp:548-597); the upload checks that on a sample of the triangles
   and AUTO falls back to the reference order when the check fails. */
int rt_set_kdtree(rt_ctx*, const rt_kd_flat*);
/* The order the next mesh frame will be walked in under the current settings and tree: RT_ORDER_REFERENCE or
   RT_ORDER_FRONT_TO_BACK (resolves RT_ORDER_AUTO). */
int rt_traversal_in_use(const rt_ctx*);
/* Camera::GetPosition / GetInverseProjection / GetInverseView (Camera.h:15-19), column-major like glm::mat4;
   ray directions are generated on the device exactly as Camera::RecalculateRayDirections (Camera.cpp:134-150). */
int rt_set_camera(rt_ctx*, const float pos[3], const float inv_proj[16], const float inv_view[16]);
/* Alternative: Camera::GetRayDirections() (Camera.h:22) computed by the host, n = width*height xyz triples.
   Pass NULL to return to device-side generation. */
int rt_set_ray_dirs(rt_ctx*, const float* xyz, size_t n);

/* ---- multi-GPU partition (no counterpart in the reference) ---------------------------------------- */
/* This context renders only the 16-row bands b with b % band_count == band_index (screen-tile split);
   (0, 1) = the whole frame.  With rt_settings.anti_aliasing on it also traces and sums the single rows bordering its bands,
   because the 3x3 filter (Renderer.cpp:84-124) reads them; they must be part of every frame of an accumulation, so switching
   the filter on under a tile split takes effect from the next restart (frame index 1 or rt_reset_accum): until then
   rt_render returns RT_E_STATE. */
int rt_set_bands(rt_ctx*, uint32_t band_index, uint32_t band_count);

/* ---- several GPUs from one process (no counterpart in the reference) ------------------------------ */
/* A group = one context per listed device (a device may be listed more than once), holding the same scene: the caller sets
   size, settings, scene and camera on every rt_group_ctx(g, i) alike.  Context 0 assembles and hands off the frame
   (rt_map_image(rt_group_ctx(g, 0))).  The other GPUs write their part straight into GPU 0's memory (peer access over xGMI);
   nothing is exchanged while frames are traced.  Returns NULL on failure (rt_last_error(NULL)). */
typedef struct rt_group rt_group;
rt_group*   rt_group_create(const int* device_ids, int n_devices);
void        rt_group_destroy(rt_group*);
int         rt_group_size(const rt_group*);
rt_ctx*     rt_group_ctx(rt_group*, int i);
const char* rt_group_last_error(const rt_group*);
/* Screen-tile split, for single frames: Renders frame_index .. frame_index + n_frames - 1 with context i tracing the 16-row
   bands b % n == i (rt_set_bands); the rows of the last frame are gathered in context 0's image.  Every pixel has one owner for
   the whole accumulation, so the frame equals the one-GPU frame bit for bit. */
int rt_group_render_tiles(rt_group*, uint32_t frame_index, uint32_t n_frames);
/* Sample split, for progressive accumulation: the n * frames_per_ctx Renders frame_first .. of this call are dealt out by
   frame index, context i taking the block frame_first + i * frames_per_ctx ..., each into its own accumulation buffer
   (frame_first == 1 starts all of them from zero).  Then the contexts' sums are added in context order on GPU 0 --
   ((a0 + a1) + a2) + ..., 12 B per pixel and context over xGMI -- and context 0's image is resolved from the total as frame
   frame_first + n * frames_per_ctx - 1.  The total differs from the one-GPU sum only by the order of the additions
   (|d| <= 1e-4 + 1e-4 |ref|).  rt_group_read_accum: that total, width*height*3 floats. */
int rt_group_render_samples(rt_group*, uint32_t frame_first, uint32_t frames_per_ctx);
int rt_group_read_accum(rt_group*, float* out_rgb);
/* Diagnostics.  Both calls above issue the trace launch on every context before the host waits for any of them (in a tree
   large enough for bounce rays with direction.x == 0 to be handed to whole-chip kernels, every launch is followed by a wait for
   one word: how many paths it parked).  This returns how many of those launches were still running when the last call had
   issued on all its contexts -- with several GPUs and frames of more than a few hundred microseconds, all of them; 0 when the
   scene needs no such word; -1 before the first call. */
int rt_group_issue_overlap(const rt_group*);

/* ---- the hot path ---------------------------------------------------------------------------- */
/* Renderer::Render (Renderer.cpp:45-135) for 1-based frame index `frame_index` (m_frameindex): trace, accumulate,
   resolve, optional AA filter.  frame_index == 1 restarts accumulation (Renderer.cpp:55-56).  The work is queued on the
   context's stream and the call returns without waiting for it, with one exception: in a tree of 2^18 nodes or more (2^12 with
   spheres and mesh together) the call waits for its trace kernel to finish, because what follows it depends on one word the
   kernel leaves behind -- the number of bounce rays with direction.x == 0 it handed to whole-chip kernels; the passes after
   that (those rays' hits, the sums, the image) are again only queued.  Once frame size, scene and tree are in place, the only
   other wait is for one word per NEW view of a tree scene (how many primary rays have direction.x == +0: it decides whether the
   pre-pass for them is launched), taken while the kernels that classify the view's work units are still queued; the other
   per-view counts, which the host keeps for rt_read_counters only, come back behind the kernels and are never waited for. */
int rt_render(rt_ctx*, uint32_t frame_index);
/* n consecutive Render calls (frame_index .. frame_index+n-1) with the scene, camera and settings of the moment; identical
   accumulation (the per-frame values are added in frame order), only the last frame's image is resolved.  Sphere scenes: one
   launch.  Tree scenes: launches of up to 16 frames whose (pixel, frame) paths are drawn independently, so the lanes of the
   chip stay busy across the frames -- several times the rate of single Renders. */
int rt_render_frames(rt_ctx*, uint32_t frame_index, uint32_t n);
/* No counterpart in the reference.  One Render of a SPHERE scene (rt_render with scene_mode RT_SCENE_SPHERES) for callers that
   cannot allocate where they render: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  It
   needs no context and calls no hipMalloc / hipFree / hipHostMalloc: the ray directions and the device copies of the scene
   tables, which rt_render keeps in buffers of the context, live in one static device array of the library (32 MiB + 2 KiB on
   each GPU, held until the process ends), and the caller passes the rest: d_accum and d_image are DEVICE buffers of
   width*height float4 (rgb + unused w; RGBA32F) on `device`; frame_index == 1 starts the sum from zero, any other index adds to
   what d_accum holds.  Same kernels as rt_render, so the same bits.  Asynchronous on `hip_stream` (a hipStream_t, NULL = the
   default stream).  Fixed limits, RT_E_LIMIT beyond them: width*height <= 2,097,152 (1920x1080 fits), 32 spheres, 32 materials.
   Every call on a GPU uses that GPU's one array: calls that overlap in time on one GPU race; calls ordered by a stream or by
   events do not, and nothing has to survive in the array from one call to the next.  Errors: rt_last_error(NULL), kept per
   calling thread.  Set-up: the first use of a device looks up that device's address of the array and the size of the persistent
   grid (hipGetSymbolAddress, a device and an occupancy query: no allocation, but not calls to make inside a stream capture).
   rt_render_spheres_noalloc_setup(device) does exactly that and nothing else; rt_create(device) does it too.  Call one of them, or
   make one ordinary call, on every device BEFORE the first call under graph capture there. */
int rt_render_spheres_noalloc_setup(int device);
int rt_render_spheres_noalloc(int device, void* hip_stream, const rt_sphere* spheres, uint32_t n_spheres,
                              const rt_material* materials, uint32_t n_materials, const rt_settings* settings,
                              const float cam_pos[3], const float inv_proj[16], const float inv_view[16],
                              uint32_t width, uint32_t height, uint32_t frame_index, float* d_accum, float* d_image);
/* Resolve only (Renderer.cpp:68-79, 84-124): rebuild the image from the accumulation buffer as if
   `frame_index` frames had been summed (used after a cross-GPU sum of accumulation buffers). */
int rt_resolve(rt_ctx*, uint32_t frame_index);
/* Zero the accumulation buffer (the memset of Renderer.cpp:55-56) without rendering: lets a context whose first frame
   index is not 1 (sample-split across GPUs) start from an empty sum. */
int rt_reset_accum(rt_ctx*);
int rt_sync(rt_ctx*);

/* ---- outputs --------------------------------------------------------------------------------- */
/* m_ImageData -> Walnut::Image::SetData (Renderer.cpp:134, Walnut/Image.h:23): width*height*4 floats RGBA32F,
   row 0 first.  Synchronises. */
int rt_read_image(rt_ctx*, float* out_rgba32f);
/* The same frame without the caller's buffer: copied into page-locked host memory the library owns (twice the PCIe rate
   of a copy into pageable memory, and no second host copy) and returned by address -- hand it straight to
   Walnut::Image::SetData.  Valid until the next rt_map_image / rt_resize / rt_destroy.  NULL on failure.  Synchronises. */
const float* rt_map_image(rt_ctx*);
/* The same hand-off without waiting for the newest frame: starts its copy into one of two page-locked buffers on a copy stream
   of the library and returns the frame of the PREVIOUS call (the very first call returns its own), so the copy of frame k runs
   beside the trace of frame k + 1 and a Render costs max(trace, copy) instead of their sum -- at the price of showing each frame
   one call late.  Valid until the next rt_map_image* / rt_resize / rt_destroy. */
const float* rt_map_image_pipelined(rt_ctx*);
/* m_AccumulationBuffer (Renderer.h:157): width*height*3 floats. */
int rt_read_accum(rt_ctx*, float* out_rgb);
/* closestSphereIndex / closestTriangleIndex of the primary ray (Renderer.cpp:233-237): -1 miss, sphere index,
   or 0x40000000 | triangle index.  Recording is off until rt_set_record_ids(ctx, 1). */
int rt_set_record_ids(rt_ctx*, int on);
int rt_read_primary_ids(rt_ctx*, int32_t* out);
/* Camera::GetRayDirections() as generated on the device: width*height*3 floats. */
int rt_read_ray_dirs(rt_ctx*, float* out_xyz);
int rt_set_stats(rt_ctx*, int on);
int rt_read_counters(rt_ctx*, rt_counters* out);
int rt_reset_counters(rt_ctx*);
/* Diagnostics.  rt_set_stats(ctx, flags): bit 0 = count nodes / triangles (the counting kernels), bit 1 = record, per wave
   of the trace launch, 8 values: the 100 MHz device clock at its start and end, the main-loop iterations it made and, in
   tree scenes, the trips of its walk loop (one read at each walking lane's cursor per trip), the sum over those trips of
   the lanes that were walking (counting kernels only; / (64 x trips) = how full the trips were), the clock ticks spent
   inside the walk loop, those spent without a path of its own, waiting for paths of the workgroup's other waves, and those
   spent in the shading block.  out holds 8 * max_waves values; the number of waves written is returned through *n_waves. */
int rt_read_wave_clock(rt_ctx*, uint64_t* out, uint32_t max_waves, uint32_t* n_waves);
int rt_get_kernel_time(rt_ctx*, rt_kernel_time* out);
int rt_reset_kernel_time(rt_ctx*);
/* Which launches are timed.  A context times nothing until rt_get_kernel_time or rt_reset_kernel_time is called for the first
   time (the event records around a launch cost a 1080p sphere frame about 5 % of its time: a caller that never asks never
   pays); from then on every launch, or with rt_set_kernel_timing(ctx, n) every n-th launch: launches n / 2, n / 2 + n, ...
   counted from the call or from the last rt_reset_kernel_time (n = 0: none).  rt_kernel_time::launches / frames count the
   timed launches.  No counterpart in the reference. */
int rt_set_kernel_timing(rt_ctx*, uint32_t every);
/* Device addresses (for collectives over the buffers): accumulation is width*height float4 (rgb + unused w),
   image is width*height float4 RGBA. */
void* rt_device_accum(rt_ctx*);
void* rt_device_image(rt_ctx*);

/* ---- experiments (no counterpart in the reference; used by tools/sorted_rays_probe.py) ------------------ */
/* One Render of a tree scene with the counting kernels that also writes down every ray that starts a walk, in the order the
   waves start them: 8 floats per ray { origin.xyz, bits(bounce, 1 = primary) } { direction.xyz, bits(pixel) }.  *n_out = rays
   there were (the first `cap` of them are returned).  Synchronises. */
int rt_debug_dump_rays(rt_ctx*, uint32_t frame_index, float* out_rays8, uint32_t cap, uint32_t* n_out);
/* The front-to-back walk of the timed kernels over a list of rays (same layout; the two words of bits are ignored), one lane
   per ray in list order (refill 0) or a persistent grid whose lanes draw the next rays as their walks end (refill 1).
   out_hits4 (may be NULL): { t, u, v, bits(triangle, -1 = none) } per ray; *out_ms: mean kernel time of `repeats` launches (HIP
   events, after one launch that is not counted).  variant 0 / 1 only names the kernel in profiles.  Synchronises. */
int rt_debug_walk_rays(rt_ctx*, const float* rays8, uint32_t n, int variant, int refill, uint32_t repeats, float* out_hits4, float* out_ms);

/* ---- host-side helpers (CPU; mirror the reference's host code that stays on the host) ------------ */
/* Camera (Camera.cpp:11-19,122-132): glm::perspectiveFov / lookAt / inverse in glm's evaluation order. */
void rt_host_camera_default_forward(float fwd[3]);
void rt_host_camera_matrices(float fov_deg, float znear, float zfar, uint32_t width, uint32_t height,
                             const float pos[3], const float fwd[3],
                             float proj[16], float inv_proj[16], float view[16], float inv_view[16]);

/* KD build with the semantics of KDTreeCPU::constructTreeStackless (KDTreeCPU.cpp:424-632): same split decisions
   and triangle order, produced by binned counting instead of the 297-plane sweep.  The returned tree is owned
   by the library; `flat` points into it until rt_host_kd_free.  max_nodes guards against the builder's
   divergence on meshes where more than 10 triangle boxes overlap (0 = no guard). */
typedef struct rt_host_kd rt_host_kd;
rt_host_kd* rt_host_kd_build(const float* verts, uint32_t nverts, const uint32_t* tri_idx, uint32_t ntris,
                             uint32_t max_nodes, int nthreads);
int         rt_host_kd_flat(const rt_host_kd*, rt_kd_flat* flat, uint32_t* depth);
void        rt_host_kd_free(rt_host_kd*);

#ifdef __cplusplus
}
#endif
#endif /* RT_MI355X_H */
