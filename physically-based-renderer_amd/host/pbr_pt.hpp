// pbr_pt.hpp — C++17 host-side mirror of the reference's scene interface over the C-ABI (include/ptc.h).
//
// Names and argument meaning follow the reference so that code written against it reads the same:
//   pbr::MeshVertex       src/pbr_engine/engine/pbr/MeshVertex.hpp:14-19
//   pbr::MeshBuilder      src/pbr_engine/engine/pbr/MeshBuilder.hpp:12-37   (indices widened to u32)
//   pbr::Transform        src/pbr_engine/engine/pbr/Scene.hpp:19-23         (rotation as w,x,y,z)
//   pbr::MaterialData     src/pbr_engine/engine/pbr/Material.hpp:14-16      (+ metal-rough, emissive)
//   pbr::PathTraceRenderSystem::render  replaces  pbr::PbrRenderSystem::render (PbrRenderSystem.hpp:46-47)
// Errors become std::runtime_error, like the reference's own failure sites (gltf_viewer/App.cpp:80,83).
#pragma once
#include <ptc.h>

#include <array>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace pbr {

struct MeshVertex {
  std::array<float, 3> position{};
  std::array<float, 3> normal{};
  std::array<float, 4> tangent{};
  std::array<float, 2> texCoords{};
};
static_assert(sizeof(MeshVertex) == sizeof(ptc_vertex), "R1: 48-byte vertex record");

struct Transform {
  std::array<float, 3> position{0, 0, 0};
  std::array<float, 4> rotation{1, 0, 0, 0};   // w, x, y, z
  std::array<float, 3> scale{1, 1, 1};
};

struct MaterialData {
  std::array<float, 4> color{1, 1, 1, 1};
  float metallic = 0.0f, roughness = 1.0f;
  std::array<float, 3> emissive{0, 0, 0};
};

// The viewer's start-up camera: app::CameraController's defaults (src/gltf_viewer/CameraController.hpp:25-40: position 0, pitch 0,
// yaw -pi/2, vertical fov pi/2) and its getDirection() / getCameraData() (CameraController.hpp:128-136), in the same float arithmetic.
struct ViewerCamera {
  std::array<float, 3> position{0.0f, 0.0f, 0.0f};
  float pitch = 0.0f, yaw = -1.57079632679489661923f, fov = 1.57079632679489661923f;
  [[nodiscard]] auto direction() const -> std::array<float, 3> {
    const float cp = std::cos(pitch);
    const float x = cp * std::cos(yaw), y = std::sin(pitch), z = cp * std::sin(yaw);
    const float il = 1.0f / std::sqrt(x * x + y * y + z * z);
    return {x * il, y * il, z * il};
  }
  [[nodiscard]] auto target() const -> std::array<float, 3> { const auto d = direction(); return {position[0] + d[0], position[1] + d[1], position[2] + d[2]}; }
};

struct PrimitiveSpan { int material; std::uint32_t firstVertex, vertexCount, firstIndex, indexCount; };

class MeshBuilder {
public:
  struct Primitive { int material = 0; std::vector<MeshVertex> vertices; std::vector<std::uint32_t> indices; };
  struct BuiltMesh { std::vector<MeshVertex> vertices; std::vector<std::uint32_t> indices; std::vector<PrimitiveSpan> primitives; };
  auto addPrimitive(Primitive p) -> MeshBuilder& { _primitives.emplace_back(std::move(p)); return *this; }
  [[nodiscard]] auto build() const -> BuiltMesh {
    BuiltMesh b;
    std::uint32_t cv = 0, ci = 0;
    for (auto const& p : _primitives) {
      b.vertices.insert(b.vertices.end(), p.vertices.begin(), p.vertices.end());
      b.indices.insert(b.indices.end(), p.indices.begin(), p.indices.end());
      b.primitives.push_back({p.material, cv, (std::uint32_t)p.vertices.size(), ci, (std::uint32_t)p.indices.size()});
      cv += (std::uint32_t)p.vertices.size();
      ci += (std::uint32_t)p.indices.size();
    }
    return b;
  }
private:
  std::vector<Primitive> _primitives;
};

class PathTraceRenderSystem {
public:
  explicit PathTraceRenderSystem(int device) : _ctx(ptc_create(device)) {
    if (!_ctx) throw std::runtime_error(ptc_last_error(nullptr));
  }
  // a view of a context owned elsewhere (one device of a DeviceGroup)
  explicit PathTraceRenderSystem(ptc_ctx* borrowed) : _ctx(borrowed), _owned(false) {
    if (!_ctx) throw std::runtime_error("PathTraceRenderSystem: null context");
  }
  ~PathTraceRenderSystem() { if (_owned) ptc_destroy(_ctx); }
  PathTraceRenderSystem(PathTraceRenderSystem const&) = delete;
  auto operator=(PathTraceRenderSystem const&) -> PathTraceRenderSystem& = delete;

  auto beginScene() -> void { ck(ptc_scene_begin(_ctx)); }
  auto addMaterial(MaterialData const& m) -> int {
    return ck(ptc_add_material(_ctx, m.color.data(), m.metallic, m.roughness, m.emissive.data(), -1, -1, -1));
  }
  // one ptc mesh per PrimitiveSpan, firstVertex applied like drawIndexed's vertexOffset (PbrRenderSystem.cpp:454-460)
  auto addMesh(MeshBuilder::BuiltMesh const& b) -> std::vector<int> {
    std::vector<int> ids;
    for (auto const& s : b.primitives)
      ids.push_back(ck(ptc_add_mesh(_ctx, reinterpret_cast<ptc_vertex const*>(b.vertices.data() + s.firstVertex), s.vertexCount,
                                    b.indices.data() + s.firstIndex, s.indexCount, s.material)));
    return ids;
  }
  auto addInstance(int mesh, Transform const& t) -> void { ck(ptc_add_instance(_ctx, mesh, t.position.data(), t.rotation.data(), t.scale.data())); }
  // pbr::makeCameraData(position, target, fov, aspect), CameraData.hpp:22-32
  auto setCamera(std::array<float, 3> position, std::array<float, 3> target, float fov, float aspect) -> void {
    ck(ptc_set_camera(_ctx, position.data(), target.data(), fov, aspect));
  }
  // PTC_BVH_SAH (default) or PTC_BVH_LBVH, for the scene being described
  auto setBvhBuilder(int builder) -> void { ck(ptc_set_bvh_builder(_ctx, builder)); }
  auto commitScene() -> void { ck(ptc_scene_commit(_ctx)); }

  // replaces PbrRenderSystem::render: fills an fp32 RGBA radiance buffer (w*h*4, y-down)
  auto render(int w, int h, int spp, std::uint64_t seed, int maxBounces, int integrator = PTC_INTEGRATOR_PATH) -> std::vector<float> {
    ck(ptc_render(_ctx, w, h, spp, seed, maxBounces, integrator));
    std::vector<float> out((std::size_t)w * h * 4);
    ck(ptc_read_radiance_rgba32f(_ctx, out.data()));
    _w = w; _h = h;
    return out;
  }
  // TonemapperSystem::run (TonemapperSystem.cpp:97-134)
  auto tonemap() -> std::vector<std::uint8_t> {
    std::vector<std::uint8_t> out((std::size_t)_w * _h * 4);
    ck(ptc_tonemap_rgba8(_ctx, out.data()));
    return out;
  }
  // the same image in the reference's HdrImage format, RGBA16F (PbrRenderSystem.hpp:21, HdrImage.cpp:20): what a viewer shim
  // copies into the image the tonemapper samples
  auto radianceHalf() -> std::vector<std::uint16_t> {
    std::vector<std::uint16_t> out((std::size_t)_w * _h * 4);
    ck(ptc_read_radiance_rgba16f(_ctx, out.data()));
    return out;
  }
  auto readRadiance(int w, int h) -> std::vector<float> {
    std::vector<float> out((std::size_t)w * h * 4);
    ck(ptc_read_radiance_rgba32f(_ctx, out.data()));
    _w = w; _h = h;
    return out;
  }
  auto stats() -> ptc_stats { ptc_stats s; ck(ptc_get_stats(_ctx, &s)); return s; }
  auto handle() -> ptc_ctx* { return _ctx; }

private:
  auto ck(int rc) -> int { if (rc < 0) throw std::runtime_error(ptc_last_error(_ctx)); return rc; }
  ptc_ctx* _ctx;
  bool _owned = true;
  int _w = 0, _h = 0;
};

// Several GPUs driven by one process (ptc_group: one context per device + an RCCL communicator): the scene described on device(0) is
// committed to every device by commitScene(), render() traces device i's 32x32-pixel tiles on device i and reduces the framebuffer onto device 0 (ncclReduce).
// The reference has a single vk::Device (core/GpuHandle.cpp:94-101); this is the build's multi-GPU addition.
class DeviceGroup {
public:
  explicit DeviceGroup(std::vector<int> const& devices) : _g(ptc_group_create(devices.data(), (int)devices.size())) {
    if (!_g) throw std::runtime_error(ptc_group_last_error(nullptr));
    for (int i = 0; i < ptc_group_size(_g); ++i) _views.emplace_back(new PathTraceRenderSystem(ptc_group_ctx(_g, i)));
  }
  ~DeviceGroup() { _views.clear(); ptc_group_destroy(_g); }
  DeviceGroup(DeviceGroup const&) = delete;
  auto operator=(DeviceGroup const&) -> DeviceGroup& = delete;
  [[nodiscard]] auto size() const -> int { return (int)_views.size(); }
  auto device(int i) -> PathTraceRenderSystem& { return *_views[(std::size_t)i]; }
  // the scene described (and possibly committed) on device(0) goes to every device with one host build
  void commitScene() { if (ptc_group_scene_commit(_g) < 0) throw std::runtime_error(ptc_group_last_error(_g)); }
  auto render(int w, int h, int spp, std::uint64_t seed, int maxBounces, int integrator = PTC_INTEGRATOR_PATH) -> std::vector<float> {
    if (ptc_group_render(_g, w, h, spp, seed, maxBounces, integrator) < 0) throw std::runtime_error(ptc_group_last_error(_g));
    return _views[0]->readRadiance(w, h);
  }

private:
  ptc_group* _g;
  std::vector<std::unique_ptr<PathTraceRenderSystem>> _views;
};

}  // namespace pbr
