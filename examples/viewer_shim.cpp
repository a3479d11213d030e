// viewer_shim.cpp — INTEGRATION.md §1–2 as a translation unit that compiles: the three touch points a maintainer adds to the
// reference's gltf_viewer so that libptc.so renders behind its seam.
//
// The reference cannot be built here (C++26, Vulkan, glm, fastgltf: SURVEY §8c), so the reference's types appear through their mirrors in
// host/pbr_pt.hpp (same names, same fields: pbr::MeshBuilder::BuiltMesh, pbr::PrimitiveSpan, pbr::Transform, pbr::MaterialData) and its
// camera controller through pbr::ViewerCamera (CameraController.hpp:25-40,128-136).  What each function stands in for:
//   shim::uploadMaterial   Asset::loadMaterial                      src/pbr_engine/gltf/pbr/gltf/Asset.cpp:135-160
//   shim::uploadMesh       Asset::loadMesh after builder.build()    src/pbr_engine/gltf/pbr/gltf/Asset.cpp:221-231
//   shim::uploadNode       Asset::loadNode (TRS -> pbr::Transform)  src/pbr_engine/gltf/pbr/gltf/Asset.cpp:239-244
//   shim::Frame::record    App::recordCommands' render call         src/gltf_viewer/App.cpp:384-393
//   shim::Frame::rotate    App::update's per-frame node rotation    src/gltf_viewer/App.cpp:306-313   (refit; a rebuild on the device when the refitted tree has become too costly)
// main() drives them the way App::run does: load, then a loop of displayed frames, one sample per frame while the camera is still.
//
// usage: viewer_shim DEVICE [frames [rebuild_ratio]]    DEVICE -1 = description only (PTC_DEVICE_NONE): the scene half runs, the render half reports
//                                       "no device" and the program still exits 0 — that is what the CPU test runs.
#include <pbr_pt.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

namespace shim {

inline void ck(ptc_ctx* ctx, int rc) { if (rc < 0) throw std::runtime_error(ptc_last_error(ctx)); }   // App.cpp:80,83 throw std::runtime_error too

// Asset::loadMaterial: baseColorFactor (+ what the reference does not read yet: metallic, roughness, emissive); texture ids from ptc_add_texture_rgba8
inline int uploadMaterial(ptc_ctx* ctx, pbr::MaterialData const& m, int texColor = -1, int texNormal = -1, int texMetalRough = -1) {
  const int id = ptc_add_material(ctx, m.color.data(), m.metallic, m.roughness, m.emissive.data(), texColor, texNormal, texMetalRough);
  ck(ctx, id);
  return id;
}

// Asset::loadMesh: one ptc mesh per PrimitiveSpan; the reference's u16 indices arrive widened (MeshBuilder here already holds u32)
inline std::vector<int> uploadMesh(ptc_ctx* ctx, pbr::MeshBuilder::BuiltMesh const& built, std::map<int, int> const& materialIdOf) {
  static_assert(sizeof(pbr::MeshVertex) == sizeof(ptc_vertex), "R1: 48-byte record, same field order");
  std::vector<int> ids;
  for (auto const& span : built.primitives) {
    const int mesh = ptc_add_mesh(ctx, reinterpret_cast<ptc_vertex const*>(built.vertices.data() + span.firstVertex), span.vertexCount,
                                  built.indices.data() + span.firstIndex, span.indexCount, materialIdOf.at(span.material));
    ck(ctx, mesh);
    ids.push_back(mesh);
  }
  return ids;
}

// Asset::loadNode: the node's TRS as the reference's pbr::Transform (rotation w,x,y,z like glm::quat's constructor); returns the instance ids
inline std::vector<int> uploadNode(ptc_ctx* ctx, std::vector<int> const& meshIds, pbr::Transform const& T) {
  std::vector<int> inst;
  for (int mesh : meshIds) { const int id = ptc_add_instance(ctx, mesh, T.position.data(), T.rotation.data(), T.scale.data()); ck(ctx, id); inst.push_back(id); }
  return inst;
}

// App::recordCommands: was `_pbrSystem.render(cmdBuffer, _scene, _gBuffer, _hdrImage.getImage(), extent);`
class Frame {
public:
  Frame(ptc_ctx* ctx, int w, int h, int sppBudget, std::uint64_t seed) : _ctx(ctx), _w(w), _h(h), _budget(sppBudget), _seed(seed), _staging16((std::size_t)w * h * 4) {}
  // one displayed frame; returns false when the context has no device (description-only run)
  bool record(pbr::ViewerCamera const& cam, bool sceneOrCameraChanged) {
    const auto target = cam.target();
    ck(_ctx, ptc_set_camera(_ctx, cam.position.data(), target.data(), cam.fov, (float)_w / (float)_h));
    int rc = PTC_OK;
    if (sceneOrCameraChanged || !_begun) rc = ptc_frame_begin(_ctx, _w, _h, _budget, _seed, 8, PTC_INTEGRATOR_PATH, 0, 1);
    if (rc == PTC_E_DEVICE) return false;
    ck(_ctx, rc);
    _begun = true;
    ck(_ctx, ptc_frame_add_samples(_ctx, 1));                 // progressive: one sample per displayed frame
    ck(_ctx, ptc_frame_resolve(_ctx));                        // sum / samples so far
    ck(_ctx, ptc_read_radiance_rgba16f(_ctx, _staging16.data()));   // the HdrImage's own format (RGBA16F); the TransferStager copies it into _hdrImage
    return true;
  }
  // App::update rotates every node a little each frame, without bound (App.cpp:306-313): new TRS for the instances, then a REFIT (0.5 ms at 250 k triangles) —
  // and, because a refit keeps the topology and the octant slots of the geometry the tree was built for, a REBUILD on the device (2 ms) once the refitted
  // tree's surface-area cost has grown past `rebuildRatio` times what it was when the tree was built (ptc_stats.bvh_sa_cost / bvh_sa_cost_built: on the atrium
  // a SAH tree refitted to ratio 1.2 — pi/16 of the viewer's rotation — costs what a freshly built LBVH costs: 14.6 node visits per ray against 12.0 at the commit; after a half turn it is 55 — profiles/r04_refit_curve.txt).  The image is the same either way.
  void rotate(std::vector<int> const& instances, pbr::Transform const& T, double rebuildRatio = 1.2) {
    for (int id : instances) ck(_ctx, ptc_update_instance(_ctx, id, T.position.data(), T.rotation.data(), T.scale.data()));
    ck(_ctx, ptc_scene_refit(_ctx));
    ptc_stats st;
    ck(_ctx, ptc_get_stats(_ctx, &st));
    if (st.bvh_sa_cost_built > 0.0 && st.bvh_sa_cost > rebuildRatio * st.bvh_sa_cost_built) {
      const int rc = ptc_scene_rebuild(_ctx);
      if (rc == PTC_E_DEVICE) return;                         // a description-only context has nothing to rebuild on
      ck(_ctx, rc);
      ++_rebuilds;
    }
  }
  [[nodiscard]] auto staging() const -> std::vector<std::uint16_t> const& { return _staging16; }
  [[nodiscard]] auto rebuilds() const -> int { return _rebuilds; }

private:
  ptc_ctx* _ctx;
  int _w, _h, _budget;
  std::uint64_t _seed;
  bool _begun = false;
  int _rebuilds = 0;
  std::vector<std::uint16_t> _staging16;
};

}  // namespace shim

int main(int argc, char** argv) {
  const int device = argc > 1 ? std::atoi(argv[1]) : PTC_DEVICE_NONE;
  const int frames = argc > 2 ? std::atoi(argv[2]) : 4;
  const double rebuildRatio = argc > 3 ? std::atof(argv[3]) : 1.2;
  ptc_ctx* ctx = ptc_create(device);
  if (!ctx) { std::fprintf(stderr, "viewer_shim: %s\n", ptc_last_error(nullptr)); return 1; }
  try {
    // ---- App::loadAsset: ptc_scene_begin before Asset::loadScene, ptc_scene_commit after it (App.cpp:161-175) ----
    shim::ck(ctx, ptc_scene_begin(ctx));
    pbr::MaterialData wall; wall.color = {0.8f, 0.3f, 0.2f, 1.0f};
    pbr::MaterialData lamp; lamp.color = {0.0f, 0.0f, 0.0f, 1.0f}; lamp.emissive = {8.0f, 8.0f, 8.0f};
    std::map<int, int> materialIdOf{{0, shim::uploadMaterial(ctx, wall)}, {1, shim::uploadMaterial(ctx, lamp)}};
    auto quad = [](float z, float half, int material) {       // a quad facing +z, in front of the viewer's start-up camera
      pbr::MeshBuilder::Primitive p;
      p.material = material;
      const float xy[4][2] = {{-half, -half}, {half, -half}, {half, half}, {-half, half}};
      for (auto const& c : xy) { pbr::MeshVertex v; v.position = {c[0], c[1], z}; v.normal = {0, 0, 1}; v.tangent = {1, 0, 0, 1}; v.texCoords = {c[0], c[1]}; p.vertices.push_back(v); }
      p.indices = {0, 1, 2, 0, 2, 3};
      return p;
    };
    const pbr::MeshBuilder::BuiltMesh built = pbr::MeshBuilder().addPrimitive(quad(-4.0f, 2.0f, 0)).addPrimitive(quad(-3.0f, 0.4f, 1)).build();
    const std::vector<int> meshIds = shim::uploadMesh(ctx, built, materialIdOf);
    pbr::Transform T;                                          // the node's transform (identity to start with)
    const std::vector<int> instances = shim::uploadNode(ctx, meshIds, T);
    const pbr::ViewerCamera cam;                               // CameraController's defaults
    const auto tg = cam.target();
    shim::ck(ctx, ptc_set_camera(ctx, cam.position.data(), tg.data(), cam.fov, 16.0f / 9.0f));
    shim::ck(ctx, ptc_scene_commit(ctx));
    ptc_stats st;
    shim::ck(ctx, ptc_get_stats(ctx, &st));
    // ---- App::run: displayed frames ----
    shim::Frame frame(ctx, 160, 90, 64, 7);
    bool rendered = true;
    double sum = 0.0;
    for (int f = 0; f < frames && rendered; ++f) {
      bool changed = f == 0;
      if (f == frames / 2 && f > 0) {                          // half way: the node turns, as App::update does every frame
        const float a = 0.25f;
        T.rotation = {std::cos(a / 2), 0.0f, 0.0f, std::sin(a / 2)};
        frame.rotate(instances, T, rebuildRatio);
        changed = true;
      }
      rendered = frame.record(cam, changed);
    }
    if (rendered) for (std::uint16_t h : frame.staging()) sum += (double)h;
    std::printf("{\"device\": %d, \"triangles\": %u, \"rendered\": %s, \"frames\": %d, \"staging_sum\": %.0f, \"rebuilds\": %d}\n", device, st.n_triangles, rendered ? "true" : "false", frames, sum,
                frame.rebuilds());
    if (!rendered) std::printf("viewer_shim: no device (PTC_DEVICE_NONE): scene described and committed, render calls answered PTC_E_DEVICE\n");
  } catch (std::exception const& e) {
    std::fprintf(stderr, "viewer_shim: %s\n", e.what());
    ptc_destroy(ctx);
    return 1;
  }
  ptc_destroy(ctx);
  return 0;
}
