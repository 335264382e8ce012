// rt_bench.cpp -- headless bench mode of the C++ host: the reference application's per-frame sequence
// (RaytracerLayer::Render, RayTracer/src/RaytracerApp.cpp:355-366: Timer, OnResize, Render, "Last render: %.3fms") without the
// window, over the C ABI of include/rt_mi355x.h.  Plain C++17, no HIP in this file; links librt_mi355x.so.
//
//   rt_bench [--config 1|2|3|4|5] [--width W --height H --bounces B --mode 0|1|2 --grid N --scale S]
//            [--frames K] [--warmup W] [--frames-per-launch F] [--d2h 0|1|2] [--devices 0,1,...] [--split tiles|samples]
//            [--reuse-first-hits 0|1] [--moving 0|1]
//
// --config picks one of the five benchmark configurations (default 2: the default sphere scene at 1920x1080, 5 bounces);
// the other options override it.  --d2h 1 ends every Render with the frame in host memory (rt_map_image), as
// Renderer::Render does (Renderer.cpp:134); 2 = the pipelined hand-off; 0 (default) leaves frames in HBM.
// --devices lists the GPUs of a group (a device may be repeated); --split tiles (one frame over all GPUs, default) or
// samples (every GPU its own frame indices).  --reuse-first-hits 1 sets rt_settings.reuse_first_hits (mesh scenes keep each
// pixel's primary hit while the camera stands still).  --moving 1: the camera turns by 2 mrad about the vertical axis before
// every Render and the accumulation restarts, as RaytracerLayer::OnUpdate does whenever Camera::OnUpdate reports motion
// (RaytracerApp.cpp:259-263): every Render is the first frame of a new view.  Prints one JSON line: ms per frame (wall clock around the Render calls,
// what the reference shows as "Last render"), Mrays/s from the device's ray counter, trace-kernel ms from HIP events.
#include <chrono>
#include <unistd.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rt_mi355x.h"

namespace {

struct Options {
    int config = 2;
    uint32_t width = 1920, height = 1080, bounces = 5;
    int mode = RT_SCENE_SPHERES;
    int grid = 0;                 // quads per side of the displaced grid (0 = no mesh)
    float scale = 1.0f;           // the grid covers 3 x 3 units times this
    uint32_t frames = 64, warmup = 5, per_launch = 1;
    int d2h = 0;
    std::vector<int> devices{0};
    bool samples = false;
    int reuse_first_hits = 0;
    int moving = 0;
};

void preset(Options& o, int config)
{
    o.config = config;
    switch (config) {
    case 1: o.width = 1280; o.height = 720; o.bounces = 5; o.mode = RT_SCENE_SPHERES; o.grid = 0; o.frames = 100; break;
    case 2: o.width = 1920; o.height = 1080; o.bounces = 5; o.mode = RT_SCENE_SPHERES; o.grid = 0; o.frames = 64; break;
    case 3: o.width = 1920; o.height = 1080; o.bounces = 8; o.mode = RT_SCENE_MESH; o.grid = 187; o.scale = 1.0f; o.frames = 64; break;
    case 4: o.width = 3840; o.height = 2160; o.bounces = 8; o.mode = RT_SCENE_MESH; o.grid = 2236; o.scale = 10.0f; o.frames = 16; break;
    case 5: o.width = 3840; o.height = 2160; o.bounces = 8; o.mode = RT_SCENE_BOTH; o.grid = 2236; o.scale = 10.0f; o.frames = 16; o.samples = true; break;
    default: fprintf(stderr, "rt_bench: --config 1..5\n"); exit(2);
    }
}

// The seven materials and nine spheres the reference application hard-codes (RaytracerApp.cpp:136-254).
void default_scene(std::vector<rt_sphere>& s, std::vector<rt_material>& m)
{
    auto mat = [](float r, float g, float b, float rough, float transp = 0.0f, float ior = 1.0f, float e = 0.0f) {
        return rt_material{{r, g, b}, {e, e, e}, rough, transp, ior};
    };
    m = {mat(0.8f, 0.8f, 0.8f, 1.0f), mat(0.35f, 1.0f, 0.17f, 1.0f), mat(1.0f, 0.0f, 0.0f, 1.0f), mat(0.0f, 0.5f, 1.0f, 0.9f),
         mat(1.0f, 0.8f, 0.0f, 0.0f, 1.0f, 1.1f), mat(1.0f, 1.0f, 1.0f, 1.0f, 0.0f, 1.0f, 5.0f), mat(1.0f, 1.0f, 1.0f, 0.0f)};
    s = {{0.0f, 1002.0f, 0.0f, 1000.0f, 0}, {0.0f, -1000.0f, 0.0f, 1000.0f, 0}, {0.0f, 0.0f, -1002.0f, 1000.0f, 0},
         {-1002.0f, 0.0f, 0.0f, 1000.0f, 1}, {1002.0f, 0.0f, 0.0f, 1000.0f, 2}, {0.5f, 0.3f, -0.3f, 0.3f, 3},
         {-0.5f, 0.3f, -0.1f, 0.3f, 4}, {0.0f, 0.5f, -1.5f, 0.5f, 6}, {0.0f, 2.9f, 0.0f, 1.0f, 5}};
}

// Displaced height field of n x n quads: y = 0.3 + 0.15 sin(4x) cos(5z), shared vertices, analytic normals; every ninth quad row
// emissive (the reference has no ambient light).  Valence <= 6, which the reference's KD builder needs to terminate.
struct Mesh { std::vector<float> verts, normals9; std::vector<uint32_t> tris, tri_mat; };

Mesh displaced_grid(int n, float k, uint32_t mat_offset)
{
    Mesh m;
    const int nv = n + 1;
    const double x0 = -1.5 * k, x1 = 1.5 * k, z0 = -3.0 * k, z1 = 0.0;
    m.verts.resize((size_t)3 * nv * nv);
    std::vector<float> nrm((size_t)3 * nv * nv);
    for (int j = 0; j < nv; ++j)
        for (int i = 0; i < nv; ++i) {
            const double x = x0 + (x1 - x0) * i / n, z = z0 + (z1 - z0) * j / n;
            const double y = 0.3 + 0.15 * std::sin(4.0 * x) * std::cos(5.0 * z);
            const double dx = 0.6 * std::cos(4.0 * x) * std::cos(5.0 * z), dz = -0.75 * std::sin(4.0 * x) * std::sin(5.0 * z);
            const double len = std::sqrt(dx * dx + 1.0 + dz * dz);
            const size_t v = (size_t)j * nv + i;
            m.verts[3 * v] = (float)x; m.verts[3 * v + 1] = (float)y; m.verts[3 * v + 2] = (float)z;
            nrm[3 * v] = (float)(-dx / len); nrm[3 * v + 1] = (float)(1.0 / len); nrm[3 * v + 2] = (float)(-dz / len);
        }
    const size_t nt = (size_t)2 * n * n;
    m.tris.resize(3 * nt); m.tri_mat.resize(nt); m.normals9.resize(9 * nt);
    size_t t = 0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            const uint32_t v00 = (uint32_t)(j * nv + i), v10 = v00 + 1, v01 = v00 + (uint32_t)nv, v11 = v01 + 1;
            const uint32_t quad[2][3] = {{v00, v01, v10}, {v10, v01, v11}};
            for (int h = 0; h < 2; ++h, ++t) {
                for (int c = 0; c < 3; ++c) {
                    m.tris[3 * t + c] = quad[h][c];
                    for (int a = 0; a < 3; ++a) m.normals9[9 * t + 3 * c + a] = nrm[3 * (size_t)quad[h][c] + a];
                }
                m.tri_mat[t] = mat_offset + ((j % 9) == 4 ? 1u : 0u);
            }
        }
    return m;
}

[[noreturn]] void die(const char* what, const char* msg) { fprintf(stderr, "rt_bench: %s: %s\n", what, msg ? msg : ""); exit(1); }

}  // namespace

int main(int argc, char** argv)
{
    Options o;
    preset(o, 2);
    for (int i = 1; i < argc; ++i) if (!strcmp(argv[i], "--config") && i + 1 < argc) preset(o, atoi(argv[i + 1]));
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) die(a.c_str(), "needs a value"); return argv[++i]; };
        if (a == "--config") (void)val();
        else if (a == "--width") o.width = (uint32_t)atoi(val());
        else if (a == "--height") o.height = (uint32_t)atoi(val());
        else if (a == "--bounces") o.bounces = (uint32_t)atoi(val());
        else if (a == "--mode") o.mode = atoi(val());
        else if (a == "--grid") o.grid = atoi(val());
        else if (a == "--scale") o.scale = (float)atof(val());
        else if (a == "--frames") o.frames = (uint32_t)atoi(val());
        else if (a == "--warmup") o.warmup = (uint32_t)atoi(val());
        else if (a == "--frames-per-launch") o.per_launch = (uint32_t)atoi(val());
        else if (a == "--d2h") o.d2h = atoi(val());
        else if (a == "--split") o.samples = !strcmp(val(), "samples");
        else if (a == "--reuse-first-hits") o.reuse_first_hits = atoi(val()) != 0;
        else if (a == "--moving") o.moving = atoi(val()) != 0;
        else if (a == "--devices") {
            o.devices.clear();
            for (const char* p = val(); *p;) { o.devices.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
        } else die(a.c_str(), "unknown option");
    }
    if (!o.frames || !o.per_launch || o.devices.empty() || (o.mode != RT_SCENE_SPHERES && o.grid <= 0)) die("options", "nothing to render");
    const uint32_t n_dev = (uint32_t)o.devices.size();

    // ---- scene (RaytracerLayer's constructor, RaytracerApp.cpp:27-254, with a synthetic mesh in place of the OBJ file)
    std::vector<rt_sphere> spheres; std::vector<rt_material> materials;
    default_scene(spheres, materials);
    Mesh mesh; rt_host_kd* kd = nullptr; rt_kd_flat flat{}; uint32_t depth = 0;
    double build_s = 0.0;
    if (o.mode != RT_SCENE_SPHERES) {
        mesh = displaced_grid(o.grid, o.scale, (uint32_t)materials.size());
        materials.push_back(rt_material{{0.75f, 0.7f, 0.6f}, {0.0f, 0.0f, 0.0f}, 0.9f, 0.0f, 1.0f});
        materials.push_back(rt_material{{1.0f, 0.95f, 0.9f}, {6.0f, 5.5f, 5.0f}, 1.0f, 0.0f, 1.0f});
        const auto t0 = std::chrono::steady_clock::now();
        unsigned th = std::thread::hardware_concurrency(); if (th > 16) th = 16; if (!th) th = 1;
        kd = rt_host_kd_build(mesh.verts.data(), (uint32_t)(mesh.verts.size() / 3), mesh.tris.data(), (uint32_t)(mesh.tris.size() / 3), 0, (int)th);
        if (!kd || rt_host_kd_flat(kd, &flat, &depth) != RT_OK) die("rt_host_kd_build", "failed");
        build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    // ---- camera (Camera(70, 0.05, 100), RaytracerApp.cpp:24; Camera.cpp:11-19,122-132)
    float pos[3] = {0.0f, 1.0f, 2.0f}, fwd0[3], fwd[3], proj[16], inv_proj[16], view[16], inv_view[16];
    rt_host_camera_default_forward(fwd0);
    memcpy(fwd, fwd0, sizeof fwd);
    rt_host_camera_matrices(70.0f, 0.05f, 100.0f, o.width, o.height, pos, fwd, proj, inv_proj, view, inv_view);
    uint32_t pose = 0;
    auto turn = [&]() {                            // Camera::OnUpdate on mouse motion (Camera.cpp:80-100): a new forward direction, new matrices
        const float a = 0.002f * (float)(++pose), c = std::cos(a), s = std::sin(a);
        fwd[0] = fwd0[0] * c + fwd0[2] * s; fwd[1] = fwd0[1]; fwd[2] = -fwd0[0] * s + fwd0[2] * c;
        rt_host_camera_matrices(70.0f, 0.05f, 100.0f, o.width, o.height, pos, fwd, proj, inv_proj, view, inv_view);
    };

    rt_group* g = rt_group_create(o.devices.data(), (int)n_dev);
    if (!g) die("rt_group_create", rt_last_error(nullptr));
    rt_settings st{1, o.mode, 1, 0, o.bounces, 1337, RT_ORDER_AUTO, o.reuse_first_hits};
    auto upload = [&]() {                          // what Renderer::Render reads from Scene and Camera, every frame
        for (uint32_t i = 0; i < n_dev; ++i) {
            rt_ctx* c = rt_group_ctx(g, (int)i);
            if (rt_resize(c, o.width, o.height) || rt_set_settings(c, &st) || rt_set_materials(c, materials.data(), (uint32_t)materials.size()) ||
                rt_set_spheres(c, spheres.data(), (uint32_t)spheres.size()) || rt_set_camera(c, pos, inv_proj, inv_view))
                die("per-frame upload", rt_last_error(c));
        }
    };
    for (uint32_t i = 0; i < n_dev && kd; ++i) {
        rt_ctx* c = rt_group_ctx(g, (int)i);
        if (rt_set_mesh(c, mesh.verts.data(), (uint32_t)(mesh.verts.size() / 3), mesh.tris.data(), mesh.normals9.data(), mesh.tri_mat.data(),
                        (uint32_t)(mesh.tris.size() / 3)) || rt_set_kdtree(c, &flat))
            die("mesh upload", rt_last_error(c));
    }
    rt_ctx* c0 = rt_group_ctx(g, 0);

    uint32_t frame_index = 1;                      // m_frameindex (Renderer.h:121)
    auto render = [&]() {                          // one pass of RaytracerLayer::Render; returns the Renders it made
        if (o.moving) { turn(); frame_index = 1; }  // RaytracerApp.cpp:259-263
        upload();
        uint32_t made;
        int rc;
        if (n_dev == 1) { rc = rt_render_frames(c0, frame_index, o.per_launch); made = o.per_launch; }
        else if (o.samples) { rc = rt_group_render_samples(g, frame_index, o.per_launch); made = n_dev * o.per_launch; }
        else { rc = rt_group_render_tiles(g, frame_index, o.per_launch); made = o.per_launch; }
        if (rc) die("render", n_dev == 1 ? rt_last_error(c0) : rt_group_last_error(g));
        frame_index += made;
        if (o.d2h && !(o.d2h == 2 ? rt_map_image_pipelined(c0) : rt_map_image(c0))) die("rt_map_image", rt_last_error(c0));
        return made;
    };
    auto sync_all = [&]() { for (uint32_t i = 0; i < n_dev; ++i) if (rt_sync(rt_group_ctx(g, (int)i))) die("rt_sync", rt_last_error(rt_group_ctx(g, (int)i))); };

    for (uint32_t f = 0; f < o.warmup; ++f) render();
    sync_all();
    for (uint32_t i = 0; i < n_dev; ++i) {
        rt_reset_counters(rt_group_ctx(g, (int)i)); rt_reset_kernel_time(rt_group_ctx(g, (int)i));
        // sphere frames one per launch: events around every 8th launch only (timing each one slows what is timed); else every launch
        rt_set_kernel_timing(rt_group_ctx(g, (int)i), (o.per_launch > 1 || o.mode != 1) ? 1u : 8u);
    }
    frame_index = 1;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t made = 0;
    while (made < o.frames) made += render();
    sync_all();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    uint64_t rays = 0, reused = 0, skipped = 0; double kernel_ms = 0.0; uint64_t kframes = 0;
    for (uint32_t i = 0; i < n_dev; ++i) {
        rt_counters cnt; rt_kernel_time kt;
        if (rt_read_counters(rt_group_ctx(g, (int)i), &cnt) || rt_get_kernel_time(rt_group_ctx(g, (int)i), &kt)) die("counters", rt_last_error(rt_group_ctx(g, (int)i)));
        rays += cnt.rays; reused += cnt.primary_hits_reused; skipped += cnt.empty_unit_rays; if (i == 0) { kernel_ms = kt.total_ms; kframes = kt.frames; }
    }
    printf("{\"bench\": \"rt_bench (C++ host, headless)\", \"config\": %d, \"width\": %u, \"height\": %u, \"bounces\": %u, \"scene_mode\": %d, "
           "\"triangles\": %zu, \"kd_nodes\": %u, \"kd_depth\": %u, \"kd_build_s\": %.2f, \"devices\": %u, \"split\": \"%s\", "
           "\"frames\": %u, \"frames_per_launch\": %u, \"d2h\": %d, \"ms_per_frame\": %.4f, \"mrays_per_s\": %.1f, \"rays_per_frame\": %.0f, "
           "\"trace_kernel_ms_per_frame_gpu0\": %.4f, \"reuse_first_hits\": %d, \"primary_hits_reused_per_frame\": %.0f, \"empty_unit_rays_per_frame\": %.0f, \"moving\": %d}\n",
           o.config, o.width, o.height, o.bounces, o.mode, mesh.tris.size() / 3, flat.n_nodes, depth, build_s, n_dev,
           n_dev == 1 ? "none" : (o.samples ? "samples" : "tiles"), made, o.per_launch, o.d2h, dt / made * 1e3, (double)rays / dt / 1e6,
           (double)rays / made, kframes ? kernel_ms / (double)kframes : 0.0, o.reuse_first_hits, (double)reused / made, (double)skipped / made, o.moving);
    rt_group_destroy(g);
    if (kd) rt_host_kd_free(kd);
    // RT_BENCH_FAST_EXIT=1: leave without the runtime's exit handlers (the sanitizer build, tools/asan_bench.sh: the ROCm
    // AddressSanitizer run time fails a check of its own inside them after everything of this program has been freed)
    if (getenv("RT_BENCH_FAST_EXIT")) { fflush(stdout); fflush(stderr); _exit(0); }
    return 0;
}
