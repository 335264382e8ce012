// rt_static.hip -- rt_render_spheres_noalloc: one Render of a sphere scene for callers that cannot allocate.
//
// Some callers may not call hipMalloc where they call the renderer: code under HIP graph capture, tight serving loops,
// harnesses that forbid it.  rt_render (rt_api.cpp) keeps the ray directions and the device copies of the scene tables in
// buffers the context allocated; here the same kernels get that memory from one file-scope device array of fixed size, and
// the accumulation buffer and the frame from the caller.  Nothing in this file allocates or frees device or page-locked
// memory after set-up (rt_static_setup: once per device, from rt_create or from the first call on that device).
//
// Every call on a GPU uses that GPU's one array: calls that overlap in time on the same GPU race.  Calls that follow each other
// on one stream, or are ordered by events, are fine; nothing in the array has to survive from one call to the next (each call
// rewrites what it reads: the scene tables and the directions of its own width x height pixels).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "../../include/rt_mi355x.h"
#include "rt_device.h"

extern "C" hipError_t rt_launch_trace(const RtParams* p, int mode, int stats, int ordered, unsigned grid, hipStream_t s);
extern "C" hipError_t rt_trace_occupancy(int mode, int stats, int ordered, int* blocks_per_cu);
extern "C" unsigned rt_trace_block(int mode);
extern "C" hipError_t rt_launch_raygen(const RtParams* p, hipStream_t s);
extern "C" hipError_t rt_launch_resolve(const RtParams* p, unsigned frame_index, unsigned gx, unsigned gy, hipStream_t s);
extern "C" hipError_t rt_launch_aa(const RtParams* p, unsigned frame_index, unsigned gx, unsigned gy, hipStream_t s);
extern "C" void rt_internal_set_error(const char* msg);          // what rt_last_error(NULL) returns (rt_api.cpp)

namespace {

// Layout of the array, in float4: ray directions of up to RT_STATIC_MAX_PIXELS pixels, then the scene tables.
constexpr size_t kDirs = 0;
constexpr size_t kSpheres = kDirs + RT_STATIC_MAX_PIXELS;                    // RT_STATIC_MAX_SPHERES x { centre, radius^2 }
constexpr size_t kSphereMat = kSpheres + RT_STATIC_MAX_SPHERES;              // RT_STATIC_MAX_SPHERES x uint32 (4 per float4)
constexpr size_t kMaterials = kSphereMat + RT_STATIC_MAX_SPHERES / 4;        // RT_STATIC_MAX_MATERIALS x 3 float4
constexpr size_t kScratchFloat4 = kMaterials + 3 * RT_STATIC_MAX_MATERIALS;

// 32 MiB + 2.1 KiB on every GPU the library is loaded on, until the process ends.
static __device__ __align__(256) float4 scratch[kScratchFloat4];

// The scene tables travel in the kernel arguments (2.1 KB) and are written to the array by one small launch: a host-to-device
// copy from the caller's pageable memory would need staging memory of the runtime's.
struct StaticTables {
    float4 spheres[RT_STATIC_MAX_SPHERES];
    uint32_t sphere_mat[RT_STATIC_MAX_SPHERES];
    float4 materials[3 * RT_STATIC_MAX_MATERIALS];
    uint32_t n_spheres, n_materials;
};

__global__ __launch_bounds__(128) void rt_static_stage_kernel(StaticTables t, float4* base)
{
    const uint32_t i = threadIdx.x;
    if (i < t.n_spheres) {
        base[kSpheres + i] = t.spheres[i];
        reinterpret_cast<uint32_t*>(base + kSphereMat)[i] = t.sphere_mat[i];
    }
    if (i < 3u * t.n_materials) base[kMaterials + i] = t.materials[i];
}

constexpr int kMaxDevices = 64;
struct PerDevice { std::once_flag once; hipError_t status = hipErrorNotInitialized; float4* ptr = nullptr; int n_cus = 0, blocks_per_cu = 0; };
PerDevice g_dev[kMaxDevices];

int fail(int code, const char* fmt, ...)
{
    char buf[400];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    rt_internal_set_error(buf);
    return code;
}

struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

// Once per device, with that device current: the address of this device's copy of the array (each GPU holds its own, at its
// own address; in host code the array's name is only a handle for this query), and what sizes the persistent grid.
extern "C" hipError_t rt_static_setup(int device)
{
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    PerDevice& d = g_dev[device];
    std::call_once(d.once, [&] {
        DeviceGuard g(device);
        if (!g.ok) { d.status = hipErrorInvalidDevice; return; }
        void* sym = nullptr;
        hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
        if (e != hipSuccess) { d.status = e; return; }
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess) { d.status = e; return; }
        int per_cu = 0;
        e = rt_trace_occupancy(RT_MODE_SPHERES, 0, 0, &per_cu);
        if (e != hipSuccess) { d.status = e; return; }
        const int max_blocks = 32 / (int)(rt_trace_block(RT_MODE_SPHERES) / 64u);
        d.blocks_per_cu = per_cu < 1 ? 1 : (per_cu > max_blocks ? max_blocks : per_cu);
        d.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
        d.ptr = static_cast<float4*>(sym);
        d.status = hipSuccess;
    });
    return d.status;
}

extern "C" int rt_render_spheres_noalloc_setup(int device)
{
    if (device < 0 || device >= kMaxDevices) return fail(RT_E_ARG, "rt_render_spheres_noalloc_setup: device %d", device);
    if (hipError_t e = rt_static_setup(device)) return fail(RT_E_HIP, "rt_render_spheres_noalloc_setup: device %d: %s", device, hipGetErrorString(e));
    return RT_OK;
}

extern "C" int rt_render_spheres_noalloc(int device, void* hip_stream, const rt_sphere* spheres, uint32_t n_spheres,
                                         const rt_material* materials, uint32_t n_materials, const rt_settings* settings,
                                         const float cam_pos[3], const float inv_proj[16], const float inv_view[16],
                                         uint32_t width, uint32_t height, uint32_t frame_index, float* d_accum, float* d_image)
{
    if (!settings || !cam_pos || !inv_proj || !inv_view || !d_accum || !d_image || (n_spheres && !spheres) || (n_materials && !materials))
        return fail(RT_E_ARG, "rt_render_spheres_noalloc: null argument");
    if (settings->scene_mode != RT_SCENE_SPHERES) return fail(RT_E_ARG, "rt_render_spheres_noalloc: scene_mode %d (sphere scenes only)", settings->scene_mode);
    if (!frame_index) return fail(RT_E_ARG, "rt_render_spheres_noalloc: frame indices are 1-based");
    if (!width || !height) return fail(RT_E_ARG, "rt_render_spheres_noalloc: bad size %ux%u", width, height);
    if ((uint64_t)width * height > RT_STATIC_MAX_PIXELS)
        return fail(RT_E_LIMIT, "rt_render_spheres_noalloc: %ux%u = %llu pixels exceed the fixed maximum of %u", width, height,
                    (unsigned long long)width * height, (unsigned)RT_STATIC_MAX_PIXELS);
    if (n_spheres > RT_STATIC_MAX_SPHERES) return fail(RT_E_LIMIT, "rt_render_spheres_noalloc: %u spheres exceed the fixed maximum of %d", n_spheres, RT_STATIC_MAX_SPHERES);
    if (n_materials > RT_STATIC_MAX_MATERIALS) return fail(RT_E_LIMIT, "rt_render_spheres_noalloc: %u materials exceed the fixed maximum of %d", n_materials, RT_STATIC_MAX_MATERIALS);
    for (uint32_t i = 0; i < n_spheres; ++i)                     // the kernels do not check
        if (spheres[i].material >= n_materials)
            return fail(RT_E_ARG, "rt_render_spheres_noalloc: sphere %u: material index %u out of range (%u materials)", i, spheres[i].material, n_materials);
    if (device < 0 || device >= kMaxDevices) return fail(RT_E_ARG, "rt_render_spheres_noalloc: device %d", device);
    if (hipError_t e = rt_static_setup(device)) return fail(RT_E_HIP, "rt_render_spheres_noalloc: set-up on device %d: %s", device, hipGetErrorString(e));
    const PerDevice& d = g_dev[device];
    DeviceGuard g(device);
    if (!g.ok) return fail(RT_E_HIP, "rt_render_spheres_noalloc: hipSetDevice(%d)", device);
    hipStream_t stream = (hipStream_t)hip_stream;
    float4* const base = d.ptr;                                  // never the array's name: that is not a device address here

    StaticTables t;
    t.n_spheres = n_spheres; t.n_materials = n_materials;
    for (uint32_t i = 0; i < n_spheres; ++i) {
        t.spheres[i] = make_float4(spheres[i].x, spheres[i].y, spheres[i].z, spheres[i].radius * spheres[i].radius);   // Renderer.cpp:257's r*r
        t.sphere_mat[i] = spheres[i].material;
    }
    for (uint32_t i = 0; i < n_materials; ++i) {
        t.materials[3 * i] = make_float4(materials[i].albedo[0], materials[i].albedo[1], materials[i].albedo[2], materials[i].roughness);
        t.materials[3 * i + 1] = make_float4(materials[i].emission[0], materials[i].emission[1], materials[i].emission[2], materials[i].transparency);
        t.materials[3 * i + 2] = make_float4(materials[i].ior, 0.0f, 0.0f, 0.0f);
    }

    RtParams p;
    memset(&p, 0, sizeof p);
    memcpy(p.cam_pos, cam_pos, sizeof p.cam_pos);
    memcpy(p.inv_proj, inv_proj, sizeof p.inv_proj);
    memcpy(p.inv_view, inv_view, sizeof p.inv_view);
    p.width = width; p.height = height;
    p.frame_first = frame_index; p.n_frames = 1;
    p.seed = settings->seed; p.bounces = settings->bounces;
    p.n_spheres = n_spheres; p.n_materials = n_materials;
    p.band_index = 0; p.band_count = 1;
    const unsigned gx = (width + RT_TILE - 1) / RT_TILE, gy = (height + RT_TILE - 1) / RT_TILE;
    p.tile_rows = 2u * gy;
    p.n_units_main = ((width + 7) / 8) * p.tile_rows * (64u / RT_UNIT);
    p.n_units = p.n_units_main;
    for (uint32_t i = 0; i < n_spheres; ++i) p.sph_inline[i] = t.spheres[i];
    p.use_ace = settings->use_ace;
    p.spheres = base + kSpheres; p.sphere_mat = reinterpret_cast<const uint32_t*>(base + kSphereMat); p.materials = base + kMaterials;
    p.dirs = base + kDirs; p.dirs_out = base + kDirs;
    p.accum = reinterpret_cast<float4*>(d_accum); p.image = reinterpret_cast<float4*>(d_image);
    p.ray_slot = -1;
    const bool fused = !settings->anti_aliasing;
    p.fuse_resolve = fused ? 1u : 0u;

    hipLaunchKernelGGL(rt_static_stage_kernel, dim3(1), dim3(128), 0, stream, t, base);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = rt_launch_raygen(&p, stream);       // Camera::RecalculateRayDirections for this call's camera and size
    if (e == hipSuccess) {
        const unsigned waves_per_block = rt_trace_block(RT_MODE_SPHERES) / 64u;
        unsigned grid = (unsigned)d.n_cus * (unsigned)d.blocks_per_cu;
        const unsigned by_work = (p.n_units * RT_UNIT / 64u + waves_per_block - 1) / waves_per_block;
        if (grid > by_work) grid = by_work;
        if (grid == 0) grid = 1;
        e = rt_launch_trace(&p, RT_MODE_SPHERES, 0, 0, grid, stream);
    }
    if (e == hipSuccess && !fused) e = rt_launch_resolve(&p, frame_index, gx, gy, stream);
    if (e == hipSuccess && settings->anti_aliasing) e = rt_launch_aa(&p, frame_index, gx, gy, stream);
    if (e != hipSuccess) return fail(RT_E_HIP, "rt_render_spheres_noalloc: launch failed: %s", hipGetErrorString(e));
    return RT_OK;
}
