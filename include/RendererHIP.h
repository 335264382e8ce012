// RendererHIP.h -- drop-in for the reference's `Renderer` (RayTracer/src/Renderer.h:101-161) on top of the C ABI in
// rt_mi355x.h.  Header-only C++17; link librt_mi355x.so.
//
// Same public members, same meaning:
//     Settings& GetSettings();  void OnResize(w, h);  void Render(const Scene&, const Camera&);
//     std::shared_ptr<Image> GetImage();  void ResetFrameIndex();  uint32_t GetFrameIndex();
// so RaytracerLayer (RayTracer/src/RaytracerApp.cpp:259-366) compiles against it unchanged:
//     using Renderer = RendererHIP<Scene, Camera, Walnut::Image, Walnut::ImageFormat::RGBA32F>;
// Default-constructed it renders on GPU 0.  RendererHIP(devices, RT_SPLIT_TILES | RT_SPLIT_SAMPLES) spreads every Render over
// several GPUs of the node from this one process (rt_group_*): by 16-row bands of the frame (single frames: the result is the
// one-GPU frame bit for bit), or by frame index (progressive accumulation: one Render then adds one frame per GPU and
// GetFrameIndex advances by their number).
//
// The template parameters are the reference's own types (Scene.h:37-44, Camera.h:6-48, Walnut/Image.h:16-52); they are
// parameters only so that this header can also be instantiated with light stand-ins where Vulkan is absent.
// What is read from them:
//   SceneT : spheres[i].{Position,Radius,MaterialIndex}, materials[i].{Albedo,Emission,Roughness,Transparency,IOR},
//            triangles[i].{Normals[0..2],MaterialIndex}, kd_tree (smart pointer to KDTreeCPU: getRootNode, getMeshVerts,
//            getMeshTris, getMeshNumVerts, getMeshNumTris)
//   CameraT: GetPosition(), GetInverseProjection(), GetInverseView() (glm types: contiguous floats, column-major)
//   ImageT : ImageT(w, h, kFormat), GetWidth(), GetHeight(), Resize(w, h), SetData(const void*); kFormat is the image
//            type's RGBA32F enumerator (Walnut/Image.h:9-14)
#pragma once

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_mi355x.h"

// Flattens the pointer tree under KDTreeCPU::getRootNode() (KDAccel/KDTreeCPU.cpp:73-76; node fields
// KDAccel/KDTreeStructs.h:55-76) into the arrays rt_set_kdtree takes.  Preorder, left before right; only leaves keep
// their triangle lists.
struct RtFlatTree {
    std::vector<float> center, extent;
    std::vector<int32_t> left, right, first, count, leaf_idx;
    rt_kd_flat view() const
    {
        rt_kd_flat f;
        f.n_nodes = (uint32_t)left.size(); f.center = center.data(); f.extent = extent.data();
        f.left = left.data(); f.right = right.data(); f.first = first.data(); f.count = count.data();
        f.n_leaf_idx = (uint32_t)leaf_idx.size(); f.leaf_idx = leaf_idx.data();
        return f;
    }
};

template <class NodeT>
inline void rt_flatten_kdtree(const NodeT* root, RtFlatTree& out)
{
    out = RtFlatTree();
    struct Item { const NodeT* node; int32_t parent; bool is_left; };
    std::vector<Item> stack{{root, -1, false}};
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        const NodeT* n = it.node;
        const int32_t me = (int32_t)out.left.size();
        if (it.parent >= 0) (it.is_left ? out.left : out.right)[(size_t)it.parent] = me;
        for (int k = 0; k < 3; ++k) { out.center.push_back(n->bbox.center[k]); out.extent.push_back(n->bbox.extends[k]); }
        out.left.push_back(-1); out.right.push_back(-1);
        if (!n->left && !n->right) {
            out.first.push_back((int32_t)out.leaf_idx.size()); out.count.push_back(n->num_tris);
            out.leaf_idx.insert(out.leaf_idx.end(), n->tri_indices, n->tri_indices + n->num_tris);
        } else { out.first.push_back(0); out.count.push_back(0); }
        if (n->right) stack.push_back({n->right, me, false});      // popped after the left child: preorder
        if (n->left) stack.push_back({n->left, me, true});
    }
}

template <class SceneT, class CameraT, class ImageT, auto kFormat>
class RendererHIP {
public:
    struct Settings {                      // Renderer::Settings (Renderer.h:103-110)
        bool Render = true;
        bool Accumulate = true;
        bool UseSphereScene = false;
        bool UseACE_Color = true;
        bool AntiAliasing = false;
        uint32_t Bounces = 8;
        int TraversalOrder = RT_ORDER_AUTO;    // not in the reference: KD tree walk order, see rt_mi355x.h
        bool ReuseFirstHits = false;           // not in the reference: mesh scenes keep each pixel's primary hit while the camera stands still
                                               // (same frames, bit for bit; rt_settings.reuse_first_hits)
        bool PipelinedHandoff = false;         // not in the reference: Render hands the frame over one call late while its
                                               // copy to the host runs beside the next trace (rt_map_image_pipelined)
    };
    Settings& GetSettings() { return m_settings; }

    enum Split { RT_SPLIT_TILES = 0, RT_SPLIT_SAMPLES = 1 };

    explicit RendererHIP(int device = 0) : RendererHIP(std::vector<int>{device}, RT_SPLIT_TILES) {}
    RendererHIP(const std::vector<int>& devices, Split split)
        : m_group(rt_group_create(devices.data(), (int)devices.size())), m_split(split)
    {
        if (!m_group) throw std::runtime_error(std::string("rt_group_create: ") + rt_last_error(nullptr));
        m_ctx = rt_group_ctx(m_group, 0);
    }
    ~RendererHIP() { rt_group_destroy(m_group); }
    RendererHIP(const RendererHIP&) = delete;
    RendererHIP& operator=(const RendererHIP&) = delete;

    // Renderer::OnResize (Renderer.cpp:20-43)
    void OnResize(uint32_t width, uint32_t height)
    {
        if (m_Image && m_Image->GetWidth() == width && m_Image->GetHeight() == height) return;
        for (int i = 0; i < rt_group_size(m_group); ++i)          // first: if it throws, image and contexts still agree
            check(rt_group_ctx(m_group, i), rt_resize(rt_group_ctx(m_group, i), width, height), "rt_resize");
        if (m_Image) m_Image->Resize(width, height);
        else m_Image = std::make_shared<ImageT>(width, height, kFormat);
        ResetFrameIndex();
    }

    // Renderer::Render (Renderer.cpp:45-135): on return the image object holds the new frame.
    void Render(const SceneT& scene, const CameraT& camera)
    {
        rt_settings st;
        st.accumulate = m_settings.Accumulate; st.scene_mode = m_settings.UseSphereScene ? RT_SCENE_SPHERES : RT_SCENE_MESH;
        st.use_ace = m_settings.UseACE_Color; st.anti_aliasing = m_settings.AntiAliasing;
        st.bounces = m_settings.Bounces; st.seed = 1337; st.traversal = m_settings.TraversalOrder;
        st.reuse_first_hits = m_settings.ReuseFirstHits;

        // spheres and materials are edited live by the UI (RaytracerApp.cpp:283-331): sent every frame (a few hundred bytes)
        m_spheres.resize(scene.spheres.size());
        for (size_t i = 0; i < scene.spheres.size(); ++i) {
            const auto& s = scene.spheres[i];
            m_spheres[i] = rt_sphere{s.Position[0], s.Position[1], s.Position[2], s.Radius, (uint32_t)s.MaterialIndex};
        }
        m_materials.resize(scene.materials.size());
        for (size_t i = 0; i < scene.materials.size(); ++i) {
            const auto& m = scene.materials[i];
            m_materials[i] = rt_material{{m.Albedo[0], m.Albedo[1], m.Albedo[2]}, {m.Emission[0], m.Emission[1], m.Emission[2]},
                                         m.Roughness, m.Transparency, m.IOR};
        }
        // mesh + KD tree are immutable after the layer's constructor (RaytracerApp.cpp:130): uploaded once per tree object
        if (scene.kd_tree && (const void*)&*scene.kd_tree != m_tree_key) upload_mesh(scene);

        const auto& pos = camera.GetPosition();
        const float p3[3] = {pos[0], pos[1], pos[2]};
        const int n = rt_group_size(m_group);
        for (int i = 0; i < n; ++i) {                       // every GPU of the group holds the same scene and camera
            rt_ctx* c = rt_group_ctx(m_group, i);
            check(c, rt_set_settings(c, &st), "rt_set_settings");
            check(c, rt_set_materials(c, m_materials.data(), (uint32_t)m_materials.size()), "rt_set_materials");
            check(c, rt_set_spheres(c, m_spheres.data(), (uint32_t)m_spheres.size()), "rt_set_spheres");
            check(c, rt_set_camera(c, p3, &camera.GetInverseProjection()[0][0], &camera.GetInverseView()[0][0]), "rt_set_camera");
        }
        uint32_t frames = 1;
        if (n == 1) check(rt_render(m_ctx, m_frameindex), "rt_render");
        else if (m_split == RT_SPLIT_TILES) gcheck(rt_group_render_tiles(m_group, m_frameindex, 1), "rt_group_render_tiles");
        else { gcheck(rt_group_render_samples(m_group, m_frameindex, 1), "rt_group_render_samples"); frames = (uint32_t)n; }
        if (m_settings.Accumulate) m_frameindex += frames; else m_frameindex = 1;      // Renderer.cpp:128-131
        // page-locked staging owned by the library
        const float* frame = m_settings.PipelinedHandoff ? rt_map_image_pipelined(m_ctx) : rt_map_image(m_ctx);
        if (!frame) check(RT_E_HIP, "rt_map_image");
        m_Image->SetData(frame);                                                // Renderer.cpp:134
    }

    std::shared_ptr<ImageT> GetImage() { return m_Image; }
    void ResetFrameIndex() { m_frameindex = 1; }
    uint32_t GetFrameIndex() { return m_frameindex; }

    rt_ctx* context() { return m_ctx; }          // the context that assembles and hands off the frame (GPU devices[0])
    rt_group* group() { return m_group; }

private:
    void check(int rc, const char* what) { check(m_ctx, rc, what); }
    void check(rt_ctx* c, int rc, const char* what)
    {
        if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_last_error(c));
    }
    void gcheck(int rc, const char* what)
    {
        if (rc != RT_OK) throw std::runtime_error(std::string(what) + ": " + rt_group_last_error(m_group));
    }

    void upload_mesh(const SceneT& scene)
    {
        const auto& kd = *scene.kd_tree;
        const uint32_t nverts = (uint32_t)kd.getMeshNumVerts(), ntris = (uint32_t)kd.getMeshNumTris();
        std::vector<float> verts((size_t)3 * nverts), normals((size_t)9 * ntris);
        std::vector<uint32_t> idx((size_t)3 * ntris), mats(ntris);
        for (uint32_t i = 0; i < nverts; ++i) for (int k = 0; k < 3; ++k) verts[3 * (size_t)i + k] = kd.getMeshVerts()[i][k];
        for (uint32_t t = 0; t < ntris; ++t) {
            for (int k = 0; k < 3; ++k) idx[3 * (size_t)t + k] = kd.getMeshTris()[t][k];
            const auto& tri = scene.triangles[t];
            for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) normals[9 * (size_t)t + 3 * c + k] = tri.Normals[c][k];
            mats[t] = (uint32_t)tri.MaterialIndex;
        }
        RtFlatTree flat;
        rt_flatten_kdtree(kd.getRootNode(), flat);
        const rt_kd_flat view = flat.view();
        for (int i = 0; i < rt_group_size(m_group); ++i) {
            rt_ctx* c = rt_group_ctx(m_group, i);
            check(c, rt_set_mesh(c, verts.data(), nverts, idx.data(), normals.data(), mats.data(), ntris), "rt_set_mesh");
            check(c, rt_set_kdtree(c, &view), "rt_set_kdtree");
        }
        m_tree_key = (const void*)&kd;
    }

    rt_group* m_group;
    rt_ctx* m_ctx = nullptr;
    Split m_split;
    Settings m_settings;
    std::shared_ptr<ImageT> m_Image;
    std::vector<rt_sphere> m_spheres;
    std::vector<rt_material> m_materials;
    const void* m_tree_key = nullptr;
    uint32_t m_frameindex = 1;
};
