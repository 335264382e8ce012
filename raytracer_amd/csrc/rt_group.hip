// rt_group.hip -- one frame over several GPUs from one process: the rt_group_* entry points of include/rt_mi355x.h.
//
// A group is a list of contexts, one per listed device (a device may be listed more than once), that hold the same scene and
// camera.  Pixels and samples are independent and the scene is read-only, so nothing is exchanged while frames are traced:
//   tile split   : context i renders the 16-row bands b with b % n == i (rt_set_bands) and pushes the finished rows of its
//                  frame straight into context 0's frame buffer -- a kernel on GPU i writing over xGMI into GPU 0's memory
//                  (peer access), one launch per displayed frame, each GPU over its own link to GPU 0;
//   sample split : context i renders its own block of frame indices into its own accumulation buffer; for display the
//                  three sums per pixel (12 B, not the padded 16) are pushed the same way into staging memory on GPU 0 and
//                  added there in context order -- ((a0 + a1) + a2) + ... -- so the result does not depend on timing.
// Ordering is by events: GPU 0's stream waits for every push before it reads.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/rt_mi355x.h"
#include "rt_device.h"

// context internals the group needs (rt_api.cpp)
extern "C" hipStream_t rt_internal_stream(rt_ctx*);
extern "C" int rt_internal_device(const rt_ctx*);
extern "C" void rt_internal_size(const rt_ctx*, uint32_t* w, uint32_t* h);
extern "C" int rt_internal_resolve_from(rt_ctx*, const void* accum_float4, uint32_t frame_index);
extern "C" void rt_internal_set_error(const char* msg);
struct rt_pending;
extern "C" uint32_t rt_internal_launch_frames(const rt_ctx*);
extern "C" rt_pending* rt_internal_pending_new();
extern "C" void rt_internal_pending_free(rt_pending*);
extern "C" int rt_internal_render_issue(rt_ctx*, uint32_t frame_index, uint32_t n, rt_pending*);
extern "C" int rt_internal_render_finish(rt_ctx*, rt_pending*);
extern "C" int rt_internal_issue_in_flight(rt_ctx*, const rt_pending*);

namespace {

// Rows of the bands `band_index` (mod band_count) of a width x height float4 frame, copied from src to dst (same layout).
__global__ __launch_bounds__(256) void rt_push_bands_kernel(const float4* __restrict__ src, float4* __restrict__ dst, uint32_t width,
                                                            uint32_t height, uint32_t band_index, uint32_t band_count)
{
    const uint32_t band = blockIdx.y * band_count + band_index;
    const uint32_t row0 = band * RT_TILE;
    const uint32_t rows = row0 + RT_TILE <= height ? RT_TILE : height - row0;
    const size_t first = (size_t)row0 * width, n = (size_t)rows * width;
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) dst[first + i] = src[first + i];
}

// The three sums of every pixel of an accumulation buffer (float4, fourth component unused), packed.
__global__ __launch_bounds__(256) void rt_push_rgb_kernel(const float4* __restrict__ accum, float* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const float4 a = accum[i];
        dst[3 * i] = a.x; dst[3 * i + 1] = a.y; dst[3 * i + 2] = a.z;
    }
}

// sum = ((own + part[0]) + part[1]) + ...: the contexts' sums added in context order.
__global__ __launch_bounds__(256) void rt_add_parts_kernel(const float4* __restrict__ own, const float* __restrict__ parts, uint32_t n_parts,
                                                           float4* __restrict__ sum, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) {
        const float4 a = own[i];
        float r = a.x, g = a.y, b = a.z;
        for (uint32_t k = 0; k < n_parts; ++k) {
            const float* q = parts + (size_t)k * 3 * n + 3 * i;
            r = r + q[0]; g = g + q[1]; b = b + q[2];
        }
        sum[i] = make_float4(r, g, b, 0.0f);
    }
}

struct DeviceGuard {
    int prev = -1; bool ok = false;
    explicit DeviceGuard(int dev) { ok = hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct rt_group {
    std::vector<rt_ctx*> ctx;
    std::vector<hipEvent_t> pushed;          // per context: its push into GPU 0's memory has finished
    std::vector<rt_pending*> pending;        // per context: the launch that has been issued and not finished
    int issue_overlap = -1;                  // see rt_group_issue_overlap
    std::string error;
    float* d_parts = nullptr;                // on GPU 0: (n - 1) x width*height x 3 sums pushed by the other contexts
    float4* d_sum = nullptr;                 // on GPU 0: the sum over all contexts (what the displayed frame is resolved from)
    size_t pixels = 0;                       // what d_parts / d_sum are sized for
    bool banded = false;                     // rt_set_bands(i, n) is in force
};

namespace {

int gfail(rt_group* g, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (g) g->error = buf; else rt_internal_set_error(buf);
    return code;
}

int ctx_fail(rt_group* g, int i, int rc, const char* what)
{
    return gfail(g, rc, "%s on context %d: %s", what, i, rt_last_error(g->ctx[(size_t)i]));
}

#define G_TRY(g, expr)                                                                                \
    do { hipError_t e_ = (expr);                                                                     \
         if (e_ != hipSuccess) return gfail((g), RT_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

int set_banding(rt_group* g, bool on)
{
    if (g->banded == on) return RT_OK;
    const uint32_t n = (uint32_t)g->ctx.size();
    for (uint32_t i = 0; i < n; ++i)
        if (int rc = rt_set_bands(g->ctx[i], on ? i : 0u, on ? n : 1u)) return ctx_fail(g, (int)i, rc, "rt_set_bands");
    g->banded = on;
    return RT_OK;
}

// Renders first .. first + n - 1 (context i: first_of(i)) on every context: the trace launch of each is issued before the host
// waits for any of them -- for the word a launch parks paths under (trees large enough for that), which decides what follows
// it -- so the GPUs of a group trace side by side, as the rows of Renderer::Render run side by side and are joined at the end
// (Renderer.cpp:59-81).  Tree scenes take launches of up to 16 Renders: each such round is issued everywhere, then finished.
template <class FirstOf>
int render_everywhere(rt_group* g, uint32_t n, FirstOf first_of, const char* what)
{
    const uint32_t ctxs = (uint32_t)g->ctx.size();
    uint32_t per_launch = rt_internal_launch_frames(g->ctx[0]);
    if (!per_launch || per_launch > n) per_launch = n;
    g->issue_overlap = 0;
    for (uint32_t done = 0; done < n; done += per_launch) {
        const uint32_t k = n - done < per_launch ? n - done : per_launch;
        int rc = RT_OK, failed = -1;
        for (uint32_t i = 0; i < ctxs && rc == RT_OK; ++i) {
            rc = rt_internal_render_issue(g->ctx[i], first_of(i) + done, k, g->pending[i]);
            if (rc != RT_OK) failed = (int)i;
        }
        if (rc == RT_OK && done == 0) {
            int in_flight = 0;
            for (uint32_t i = 0; i < ctxs; ++i) in_flight += rt_internal_issue_in_flight(g->ctx[i], g->pending[i]);
            g->issue_overlap = in_flight;
        }
        // (what was issued is finished even after a failure elsewhere: a launch must not be left without its resolve pass)
        for (uint32_t i = 0; i < ctxs; ++i) {
            const int rf = rt_internal_render_finish(g->ctx[i], g->pending[i]);
            if (rf != RT_OK && rc == RT_OK) { rc = rf; failed = (int)i; }
        }
        if (rc != RT_OK) return ctx_fail(g, failed, rc, what);
    }
    return RT_OK;
}

unsigned copy_grid(size_t n) { const size_t b = (n + 255) / 256; return (unsigned)(b < 8192 ? (b ? b : 1) : 8192); }

}  // namespace

extern "C" {

rt_group* rt_group_create(const int* device_ids, int n_devices)
{
    if (!device_ids || n_devices < 1 || n_devices > 64) { gfail(nullptr, RT_E_ARG, "rt_group_create: %d devices", n_devices); return nullptr; }
    rt_group* g = new rt_group();
    for (int i = 0; i < n_devices; ++i) {
        rt_ctx* c = rt_create(device_ids[i]);
        if (!c) { rt_group_destroy(g); return nullptr; }                     // rt_last_error(NULL) holds rt_create's message
        g->ctx.push_back(c);
        g->pending.push_back(rt_internal_pending_new());
        hipEvent_t ev = nullptr;
        DeviceGuard dg(device_ids[i]);
        if (!dg.ok || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            gfail(nullptr, RT_E_HIP, "rt_group_create: event on device %d", device_ids[i]); rt_group_destroy(g); return nullptr;
        }
        g->pushed.push_back(ev);
    }
    // the other GPUs write into GPU 0's memory
    const int dev0 = device_ids[0];
    for (int i = 1; i < n_devices; ++i) {
        if (device_ids[i] == dev0) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device_ids[i], dev0) != hipSuccess || !can) {
            gfail(nullptr, RT_E_HIP, "rt_group_create: device %d cannot access the memory of device %d", device_ids[i], dev0);
            rt_group_destroy(g); return nullptr;
        }
        DeviceGuard dg(device_ids[i]);
        const hipError_t e = hipDeviceEnablePeerAccess(dev0, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
            gfail(nullptr, RT_E_HIP, "rt_group_create: hipDeviceEnablePeerAccess(%d) from device %d: %s", dev0, device_ids[i], hipGetErrorString(e));
            rt_group_destroy(g); return nullptr;
        }
        (void)hipGetLastError();
    }
    return g;
}

void rt_group_destroy(rt_group* g)
{
    if (!g) return;
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        if (g->ctx[i]) (void)rt_sync(g->ctx[i]);
    }
    if (!g->ctx.empty()) {
        DeviceGuard dg(rt_internal_device(g->ctx[0]));
        if (g->d_parts) (void)hipFree(g->d_parts);
        if (g->d_sum) (void)hipFree(g->d_sum);
    }
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        if (i < g->pushed.size() && g->pushed[i]) { DeviceGuard dg(rt_internal_device(g->ctx[i])); (void)hipEventDestroy(g->pushed[i]); }
        rt_destroy(g->ctx[i]);
    }
    for (rt_pending* L : g->pending) rt_internal_pending_free(L);
    delete g;
}

int rt_group_size(const rt_group* g) { return g ? (int)g->ctx.size() : 0; }
rt_ctx* rt_group_ctx(rt_group* g, int i) { return (g && i >= 0 && (size_t)i < g->ctx.size()) ? g->ctx[(size_t)i] : nullptr; }
const char* rt_group_last_error(const rt_group* g) { return g ? g->error.c_str() : rt_last_error(nullptr); }

int rt_group_render_tiles(rt_group* g, uint32_t frame_index, uint32_t n_frames)
{
    if (!g) return RT_E_ARG;
    const uint32_t n = (uint32_t)g->ctx.size();
    if (int rc = set_banding(g, true)) return rc;
    uint32_t w = 0, h = 0;
    rt_internal_size(g->ctx[0], &w, &h);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t wi = 0, hi = 0;
        rt_internal_size(g->ctx[i], &wi, &hi);
        if (wi != w || hi != h || !w) return gfail(g, RT_E_STATE, "rt_group_render_tiles: context %u is %ux%u, context 0 %ux%u (rt_resize all of them alike)", i, wi, hi, w, h);
    }
    // every GPU is at work on its bands before the host waits for any of them, and before any push is queued
    if (!n_frames) return gfail(g, RT_E_ARG, "rt_group_render_tiles: no frames");
    if (int rc = render_everywhere(g, n_frames, [&](uint32_t) { return frame_index; }, "rt_render_frames")) return rc;
    float4* const frame0 = static_cast<float4*>(rt_device_image(g->ctx[0]));
    const uint32_t bands = (h + RT_TILE - 1) / RT_TILE;
    for (uint32_t i = 1; i < n; ++i) {
        const uint32_t own = bands > i ? (bands - i + n - 1) / n : 0;
        DeviceGuard dg(rt_internal_device(g->ctx[i]));
        hipStream_t s = rt_internal_stream(g->ctx[i]);
        if (own) {
            hipLaunchKernelGGL(rt_push_bands_kernel, dim3(copy_grid((size_t)RT_TILE * w) > 64 ? 64 : copy_grid((size_t)RT_TILE * w), own), dim3(256), 0, s,
                               static_cast<const float4*>(rt_device_image(g->ctx[i])), frame0, w, h, i, n);
            G_TRY(g, hipGetLastError());
        }
        G_TRY(g, hipEventRecord(g->pushed[i], s));
    }
    {
        DeviceGuard dg(rt_internal_device(g->ctx[0]));
        for (uint32_t i = 1; i < n; ++i) G_TRY(g, hipStreamWaitEvent(rt_internal_stream(g->ctx[0]), g->pushed[i], 0));
    }
    return RT_OK;
}

int rt_group_render_samples(rt_group* g, uint32_t frame_first, uint32_t frames_per_ctx)
{
    if (!g) return RT_E_ARG;
    const uint32_t n = (uint32_t)g->ctx.size();
    if (!frame_first || !frames_per_ctx || (uint64_t)frame_first + (uint64_t)n * frames_per_ctx > 0xFFFFFFFFull)
        return gfail(g, RT_E_ARG, "rt_group_render_samples: frame_first %u, %u frames per context", frame_first, frames_per_ctx);
    if (int rc = set_banding(g, false)) return rc;
    uint32_t w = 0, h = 0;
    rt_internal_size(g->ctx[0], &w, &h);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t wi = 0, hi = 0;
        rt_internal_size(g->ctx[i], &wi, &hi);
        if (wi != w || hi != h || !w) return gfail(g, RT_E_STATE, "rt_group_render_samples: context %u is %ux%u, context 0 %ux%u (rt_resize all of them alike)", i, wi, hi, w, h);
    }
    const size_t px = (size_t)w * h;
    const int dev0 = rt_internal_device(g->ctx[0]);
    if (g->pixels != px) {
        DeviceGuard dg(dev0);
        for (uint32_t i = 0; i < n; ++i) if (int rc = rt_sync(g->ctx[i])) return ctx_fail(g, (int)i, rc, "rt_sync");
        if (g->d_parts) { (void)hipFree(g->d_parts); g->d_parts = nullptr; }
        if (g->d_sum) { (void)hipFree(g->d_sum); g->d_sum = nullptr; }
        g->pixels = 0;
        if (n > 1) G_TRY(g, hipMalloc((void**)&g->d_parts, (size_t)(n - 1) * px * 3 * sizeof(float)));
        G_TRY(g, hipMalloc((void**)&g->d_sum, px * sizeof(float4)));
        g->pixels = px;
    }
    // context i renders the frame indices frame_first + i * frames_per_ctx ... of this call; a new accumulation (frame_first
    // == 1) starts every context from zero (context 0 by its frame index 1, Renderer.cpp:55-56, the others explicitly)
    for (uint32_t i = 1; i < n; ++i)
        if (frame_first == 1u) { if (int rc = rt_reset_accum(g->ctx[i])) return ctx_fail(g, (int)i, rc, "rt_reset_accum"); }
    if (int rc = render_everywhere(g, frames_per_ctx, [&](uint32_t i) { return frame_first + i * frames_per_ctx; }, "rt_render_frames")) return rc;
    for (uint32_t i = 1; i < n; ++i) {
        DeviceGuard dg(rt_internal_device(g->ctx[i]));
        hipStream_t s = rt_internal_stream(g->ctx[i]);
        hipLaunchKernelGGL(rt_push_rgb_kernel, dim3(copy_grid(px)), dim3(256), 0, s, static_cast<const float4*>(rt_device_accum(g->ctx[i])),
                           g->d_parts + (size_t)(i - 1) * 3 * px, px);
        G_TRY(g, hipGetLastError());
        G_TRY(g, hipEventRecord(g->pushed[i], s));
    }
    DeviceGuard dg(dev0);
    hipStream_t s0 = rt_internal_stream(g->ctx[0]);
    for (uint32_t i = 1; i < n; ++i) G_TRY(g, hipStreamWaitEvent(s0, g->pushed[i], 0));
    hipLaunchKernelGGL(rt_add_parts_kernel, dim3(copy_grid(px)), dim3(256), 0, s0, static_cast<const float4*>(rt_device_accum(g->ctx[0])),
                       g->d_parts, n - 1u, g->d_sum, px);
    G_TRY(g, hipGetLastError());
    // the pushes of the next call may not overwrite the parts before this sum has read them
    G_TRY(g, hipEventRecord(g->pushed[0], s0));
    for (uint32_t i = 1; i < n; ++i) {
        DeviceGuard di(rt_internal_device(g->ctx[i]));
        G_TRY(g, hipStreamWaitEvent(rt_internal_stream(g->ctx[i]), g->pushed[0], 0));
    }
    if (int rc = rt_internal_resolve_from(g->ctx[0], g->d_sum, frame_first + n * frames_per_ctx - 1u)) return ctx_fail(g, 0, rc, "resolve");
    return RT_OK;
}

int rt_group_issue_overlap(const rt_group* g) { return g ? g->issue_overlap : -1; }

int rt_group_read_accum(rt_group* g, float* out_rgb)
{
    if (!g || !out_rgb) return RT_E_ARG;
    if (!g->d_sum || !g->pixels) return gfail(g, RT_E_STATE, "rt_group_read_accum before rt_group_render_samples");
    DeviceGuard dg(rt_internal_device(g->ctx[0]));
    std::vector<float4> tmp(g->pixels);
    G_TRY(g, hipMemcpyAsync(tmp.data(), g->d_sum, g->pixels * sizeof(float4), hipMemcpyDeviceToHost, rt_internal_stream(g->ctx[0])));
    if (int rc = rt_sync(g->ctx[0])) return ctx_fail(g, 0, rc, "rt_sync");
    for (size_t i = 0; i < g->pixels; ++i) { out_rgb[3 * i] = tmp[i].x; out_rgb[3 * i + 1] = tmp[i].y; out_rgb[3 * i + 2] = tmp[i].z; }
    return RT_OK;
}

}  // extern "C"
