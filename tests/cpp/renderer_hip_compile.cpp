// Compile-and-link check of include/RendererHIP.h against the reference's OWN Scene.h / Camera.h / KDTreeCPU.h (taken from
// the reference tree on the include path; only built where that tree exists) with an in-memory image class in place of
// Walnut::Image, whose header needs Vulkan.  Never run on a GPU-less box: it only has to compile and link.
#include <cstdint>
#include <vector>
#include "Scene.h"
#include "Camera.h"
#include "RendererHIP.h"

enum class TestImageFormat { None = 0, RGBA, RGBA32F };
class TestImage {
public:
    TestImage(uint32_t w, uint32_t h, TestImageFormat) : m_w(w), m_h(h), m_data((size_t)4 * w * h) {}
    void SetData(const void* d) { std::memcpy(m_data.data(), d, m_data.size() * sizeof(float)); }
    void Resize(uint32_t w, uint32_t h) { m_w = w; m_h = h; m_data.assign((size_t)4 * w * h, 0.0f); }
    uint32_t GetWidth() const { return m_w; }
    uint32_t GetHeight() const { return m_h; }
private:
    uint32_t m_w, m_h; std::vector<float> m_data;
};

using Renderer = RendererHIP<Scene, Camera, TestImage, TestImageFormat::RGBA32F>;

int render_once(Scene& scene, Camera& camera)
{
    Renderer r(0);
    r.GetSettings().UseSphereScene = true;
    r.GetSettings().Bounces = 5;
    r.OnResize(64, 48);
    camera.OnResize(64, 48);
    r.Render(scene, camera);
    return (int)r.GetFrameIndex() + (r.GetImage() ? 1 : 0);
}
