// renderer_hip_demo -- the drop-in exercised with the REFERENCE'S OWN host types: Scene / Sphere / Material / Triangle
// (Scene.h), Camera (Camera.h/.cpp) and KDTreeCPU (the reference builder, KDAccel/*.cpp) are compiled unmodified from the
// reference tree; only Renderer is replaced by RendererHIP (include/RendererHIP.h) and Walnut::Image (Vulkan) by an in-memory
// image.  Built by oracle/Makefile into oracle/_ref/ (where the reference tree exists) and run by the GPU tests, which compare
// what it writes with the oracle.
//
//   renderer_hip_demo <scene.bin> <out.bin>
// scene.bin: u32 w, h, frames, bounces, use_sphere_scene, n_spheres, n_materials, n_verts, n_tris;
//            spheres {x y z r, u32 mat}; materials {albedo3 emission3 rough transp ior}; verts xyz; tris u32x3; normals 9/tri; u32 mat/tri
// out.bin  : u32 frame_index_after, n_kd_nodes; image w*h*4 f32 (what Image::SetData received); accumulation w*h*3 f32
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "Scene.h"
#include "Camera.h"
#include "RendererHIP.h"

enum class MemImageFormat { None = 0, RGBA, RGBA32F };
class MemImage {
public:
    MemImage(uint32_t w, uint32_t h, MemImageFormat) : m_w(w), m_h(h), data((size_t)4 * w * h) {}
    void SetData(const void* d) { std::memcpy(data.data(), d, data.size() * sizeof(float)); }
    void Resize(uint32_t w, uint32_t h) { m_w = w; m_h = h; data.assign((size_t)4 * w * h, 0.0f); }
    uint32_t GetWidth() const { return m_w; }
    uint32_t GetHeight() const { return m_h; }
    std::vector<float> data;
private:
    uint32_t m_w, m_h;
};
using Renderer = RendererHIP<Scene, Camera, MemImage, MemImageFormat::RGBA32F>;

template <class T> static bool rd(FILE* f, T* p, size_t n) { return fread(p, sizeof(T), n, f) == n; }

int main(int argc, char** argv)
{
    if (argc != 3 && argc != 4) { fprintf(stderr, "usage: %s scene.bin out.bin [contexts on GPU 0, tile split]\n", argv[0]); return 2; }
    const int n_ctx = argc == 4 ? atoi(argv[3]) : 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    uint32_t hd[9];
    if (!rd(f, hd, 9)) return 2;
    const uint32_t w = hd[0], h = hd[1], frames = hd[2], bounces = hd[3], use_spheres = hd[4], ns = hd[5], nm = hd[6], nv = hd[7], nt = hd[8];

    Scene scene;
    for (uint32_t i = 0; i < ns; ++i) {
        float v[4]; uint32_t m;
        if (!rd(f, v, 4) || !rd(f, &m, 1)) return 2;
        Sphere s; s.Position = glm::vec3(v[0], v[1], v[2]); s.Radius = v[3]; s.MaterialIndex = m;
        scene.spheres.push_back(s);
    }
    for (uint32_t i = 0; i < nm; ++i) {
        float v[9];
        if (!rd(f, v, 9)) return 2;
        Material m; m.Albedo = glm::vec3(v[0], v[1], v[2]); m.Emission = glm::vec3(v[3], v[4], v[5]);
        m.Roughness = v[6]; m.Transparency = v[7]; m.IOR = v[8];
        scene.materials.push_back(m);
    }
    std::vector<glm::vec3> verts(nv); std::vector<glm::uvec3> tris(nt);
    std::vector<float> normals((size_t)9 * nt); std::vector<uint32_t> tmat(nt);
    if (nv && !rd(f, &verts[0][0], (size_t)3 * nv)) return 2;
    if (nt && (!rd(f, &tris[0][0], (size_t)3 * nt) || !rd(f, normals.data(), normals.size()) || !rd(f, tmat.data(), nt))) return 2;
    fclose(f);
    for (uint32_t t = 0; t < nt; ++t) {                     // as RaytracerApp.cpp:60-127 fills Scene::triangles
        Triangle tri;
        for (int c = 0; c < 3; ++c) {
            tri.Vertices.push_back(verts[tris[t][c]]);
            tri.Normals.push_back(glm::vec3(normals[9 * (size_t)t + 3 * c], normals[9 * (size_t)t + 3 * c + 1], normals[9 * (size_t)t + 3 * c + 2]));
        }
        tri.MaterialIndex = tmat[t];
        scene.triangles.push_back(tri);
    }
    if (nt) scene.kd_tree = std::make_shared<KDTreeCPU>((int)nt, tris.data(), (int)nv, verts.data());   // the reference builder

    Camera camera(70.0f, 0.05f, 100.0f);                   // RaytracerApp.cpp:24
    Renderer renderer(std::vector<int>((size_t)(n_ctx < 1 ? 1 : n_ctx), 0), Renderer::RT_SPLIT_TILES);   // one context: as Renderer renderer(0)
    renderer.GetSettings().UseSphereScene = use_spheres != 0;
    renderer.GetSettings().Bounces = bounces;
    for (uint32_t i = 0; i < frames; ++i) {                 // RaytracerLayer::Render, RaytracerApp.cpp:355-366
        renderer.OnResize(w, h);
        camera.OnResize(w, h);
        renderer.Render(scene, camera);
    }
    // the accumulation buffer: every context holds the sums of its own 16-row bands (one context: all of them)
    std::vector<float> accum((size_t)3 * w * h), part((size_t)3 * w * h);
    const int n_in_group = rt_group_size(renderer.group());
    for (int i = 0; i < n_in_group; ++i) {
        rt_ctx* c = rt_group_ctx(renderer.group(), i);
        if (rt_read_accum(c, part.data()) != RT_OK) { fprintf(stderr, "%s\n", rt_last_error(c)); return 1; }
        for (uint32_t y = 0; y < h; ++y)
            if ((int)((y / 16u) % (uint32_t)n_in_group) == i) memcpy(&accum[(size_t)3 * w * y], &part[(size_t)3 * w * y], (size_t)3 * w * sizeof(float));
    }
    uint32_t n_nodes = 0;
    if (nt) { RtFlatTree flat; rt_flatten_kdtree(scene.kd_tree->getRootNode(), flat); n_nodes = (uint32_t)flat.left.size(); }

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    const uint32_t tail[2] = {renderer.GetFrameIndex(), n_nodes};
    fwrite(tail, 4, 2, o);
    fwrite(renderer.GetImage()->data.data(), 4, renderer.GetImage()->data.size(), o);
    fwrite(accum.data(), 4, accum.size(), o);
    fclose(o);
    printf("\nrendered %u frame(s) %ux%u, next frame index %u, kd nodes %u\n", frames, w, h, tail[0], n_nodes);
    return 0;
}
