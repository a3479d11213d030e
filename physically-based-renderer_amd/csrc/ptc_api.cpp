// ptc_api.cpp — the C-ABI of include/ptc.h over the HIP wavefront path tracer.
//
// One context = one HIP device + one stream.  Everything a frame needs is enqueued on that stream
// without host synchronisation: queue sizes live in device memory and the persistent kernels read
// them there, so a whole batch (raygen → [trace, shade, shadow, advance] × bounces → accumulate) is
// a single asynchronous burst.  The host blocks only in ptc_sync / read-backs / ptc_get_stats.
//
// There is no CPU path in this library: without a usable HIP device ptc_create fails.
#include "../../include/ptc.h"
#include "ptc_internal.h"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {
std::string g_create_error;

struct Span { hipEvent_t a, b; int kind; };   // kind: 0 trace_closest, 1 trace_any, 2 shade, 3 whole batch

template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0;
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};
}  // namespace

struct ptc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  LaunchCfg cfg{};
  uint32_t nodelet_budget = 85;   // wide nodes staged in LDS: the top four levels (1+4+16+64) of the tree, 48 B each = 4 KB
  size_t max_batch_paths = (size_t)1 << 28;   // paths in flight over all lanes: large batches amortise launch tails (sized for 288 GB of HBM:
                                              // 176 B per path -> 47 GB of queues at 1080p x 64 spp x 2 lanes; 2^27 is 2 % slower, 2^29 1 % faster)
  bool timing = true;
  // description
  std::vector<HostMaterial> mats;
  std::vector<HostMesh> meshes;
  std::vector<HostInstance> insts;
  std::vector<HostTexture> texs;
  HostEnv env;
  float cam_pos[3]{}, cam_target[3]{}, cam_fov = 0, cam_aspect = 1;
  bool have_cam = false;
  int tex_linear = 0;                    // PTC_FILTER_*: texture filter of the scene being described
  // committed scene
  bool committed = false;
  HostBuilt built;
  DevScene dsc{};
  DevCamera cam{};
  std::vector<void*> scene_allocs;
  // queues (lane 0); further lanes run independent batches on their own streams so that one batch's
  // launch tails overlap another batch's full-occupancy phases
  DevQueues q{};
  std::vector<void*> queue_allocs;
  struct Lane { hipStream_t stream = nullptr; DevQueues q{}; std::vector<void*> allocs; uint2* stack_ovf = nullptr; hipEvent_t acc_done = nullptr; };
  std::vector<Lane> extra;          // lanes 1..n_lanes-1
  int n_lanes = 2;
  hipEvent_t acc_done0 = nullptr;   // lane 0's "accumulate finished" event
  uint64_t batches_issued = 0;
  // frame
  bool in_frame = false;
  DevFrame fr{};
  int spp_total = 0, integrator = 0;
  uint32_t samples_done = 0;
  DevBuf<uint32_t> owned;
  DevBuf<float4> accum, radiance;
  DevBuf<uint32_t> ldr;
  int rad_w = 0, rad_h = 0;
  // stats
  ptc_stats stats{};
  std::vector<Span> spans;
  std::vector<hipEvent_t> event_pool;
  size_t events_used = 0;
};

namespace {

int fail(ptc_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }

#define HIP_TRY(c, expr)                                                                                 \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return fail((c), PTC_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

template <class T> int dev_alloc(ptc_ctx* c, std::vector<void*>& owner, T** out, size_t count) {
  void* p = nullptr;
  HIP_TRY(c, hipMalloc(&p, (count ? count : 1) * sizeof(T)));
  owner.push_back(p);
  *out = (T*)p;
  return PTC_OK;
}
template <class T> int dev_upload(ptc_ctx* c, std::vector<void*>& owner, const T** out, const std::vector<T>& v) {
  T* p = nullptr;
  int rc = dev_alloc(c, owner, &p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIP_TRY(c, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return PTC_OK;
}
void free_all(std::vector<void*>& v) { for (void* p : v) (void)hipFree(p); v.clear(); }

int ensure_buf_u32(ptc_ctx* c, DevBuf<uint32_t>& b, size_t n) {
  if (b.n >= n && b.p) return PTC_OK;
  b.release();
  HIP_TRY(c, hipMalloc((void**)&b.p, (n ? n : 1) * sizeof(uint32_t)));
  b.n = n;
  return PTC_OK;
}
int ensure_buf_f4(ptc_ctx* c, DevBuf<float4>& b, size_t n) {
  if (b.n >= n && b.p) return PTC_OK;
  b.release();
  HIP_TRY(c, hipMalloc((void**)&b.p, (n ? n : 1) * sizeof(float4)));
  b.n = n;
  return PTC_OK;
}

int ensure_queues_in(ptc_ctx* c, DevQueues& dst, std::vector<void*>& allocs, uint32_t cap) {
  if (dst.cap >= cap && dst.cnt) return PTC_OK;
  free_all(allocs);
  DevQueues q{};
  int rc = 0;
  for (int k = 0; k < 2; ++k) {
    rc |= dev_alloc(c, allocs, &q.ray[k].A, cap); rc |= dev_alloc(c, allocs, &q.ray[k].B, cap);
    rc |= dev_alloc(c, allocs, &q.ray[k].C, cap);
  }
  rc |= dev_alloc(c, allocs, &q.shadow.A, cap); rc |= dev_alloc(c, allocs, &q.shadow.B, cap);
  rc |= dev_alloc(c, allocs, &q.shadow.C, cap);
  rc |= dev_alloc(c, allocs, &q.hit, cap); rc |= dev_alloc(c, allocs, &q.lpath, cap);
  rc |= dev_alloc(c, allocs, &q.cnt, (size_t)CNT_N);
  rc |= dev_alloc(c, allocs, &q.stats, (size_t)ST_N);
  if (rc) { free_all(allocs); dst = DevQueues{}; return rc < 0 ? PTC_E_DEVICE : rc; }
  HIP_TRY(c, hipMemset(q.cnt, 0, CNT_N * sizeof(uint32_t)));
  HIP_TRY(c, hipMemset(q.stats, 0, ST_N * sizeof(unsigned long long)));
  q.cap = cap;
  dst = q;
  return PTC_OK;
}
int ensure_queues(ptc_ctx* c, uint32_t cap) { return ensure_queues_in(c, c->q, c->queue_allocs, cap); }

// lane accessors: lane 0 is the context's primary stream / queues
hipStream_t lane_stream(ptc_ctx* c, int l) { return l == 0 ? c->stream : c->extra[(size_t)l - 1].stream; }
DevQueues& lane_q(ptc_ctx* c, int l) { return l == 0 ? c->q : c->extra[(size_t)l - 1].q; }
hipEvent_t lane_acc_event(ptc_ctx* c, int l) { return l == 0 ? c->acc_done0 : c->extra[(size_t)l - 1].acc_done; }
DevScene lane_scene(ptc_ctx* c, int l) { DevScene d = c->dsc; if (l > 0) d.stack_ovf = c->extra[(size_t)l - 1].stack_ovf; return d; }

hipEvent_t next_event(ptc_ctx* c) {
  if (c->events_used == c->event_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    c->event_pool.push_back(e);
  }
  return c->event_pool[c->events_used++];
}
struct ScopedSpan {   // records a start/stop event pair around launches on one of the context's streams
  ptc_ctx* c; hipStream_t st; Span s{}; bool on;
  ScopedSpan(ptc_ctx* c_, hipStream_t st_, int kind) : c(c_), st(st_), on(c_->timing) {
    if (!on) return;
    s.kind = kind; s.a = next_event(c); s.b = next_event(c);
    if (!s.a || !s.b) { on = false; return; }
    (void)hipEventRecord(s.a, st);
  }
  ~ScopedSpan() { if (on) { (void)hipEventRecord(s.b, st); c->spans.push_back(s); } }
};

int configure_launch(ptc_ctx* c) {
  // Traversal stack: a 4-wide node defers up to 3 children per level, so a ray needs at most
  // 3·(depth+1) entries.  `stack_lds` of them live in LDS (8 B each, 512 B per level and wave), the rest in
  // a global overflow slab.  LDS per block = nodelets·48 B + waves·stack_lds·512 B.
  const int need = 3 * ((int)c->built.max_depth + 1);
  int l = 8;    // stack entries per lane kept in LDS (2 KB per entry and block): with 85 nodelets 8 entries leave room for 7 blocks (28 waves) per CU, +3.5 % over 10 entries / 6 blocks
  if (const char* e = std::getenv("PTC_STACK_LDS")) { int v = std::atoi(e); if (v >= 1 && v <= 64) l = v; }
  if (l > need) l = need;
  c->cfg.stack_lds = l;
  const int waves = pt_trace_block_threads() / 64;
  const size_t lds = (size_t)c->built.n_nodelets * 48 + (size_t)waves * l * 512;
  int per_cu = (int)((160u * 1024u) / lds);
  const int max_per_cu = 32 / waves;                             // 32 waves per CU
  if (per_cu > max_per_cu) per_cu = max_per_cu;
  if (per_cu < 1) return fail(c, PTC_E_ARG, "configure_launch: nodelets + stack exceed the 160 KiB of LDS");
  if (const char* e = std::getenv("PTC_TRACE_BLOCKS_PER_CU")) { int v = std::atoi(e); if (v >= 1 && v <= per_cu) per_cu = v; }
  c->cfg.trace_blocks_per_cu = per_cu;
  const uint32_t ovf = (uint32_t)(need - l > 0 ? need - l : 1);
  const size_t total_waves = (size_t)c->cfg.n_cu * 32;           // upper bound on resident trace waves
  uint2* p = nullptr;
  int rc = dev_alloc(c, c->scene_allocs, &p, total_waves * ovf * 64);
  if (rc) return rc;
  c->dsc.stack_ovf = p; c->dsc.ovf_depth = ovf;
  for (auto& ln : c->extra) {                                    // concurrent kernels must not share a slab
    uint2* pl = nullptr;
    if ((rc = dev_alloc(c, c->scene_allocs, &pl, total_waves * ovf * 64))) return rc;
    ln.stack_ovf = pl;
  }
  return PTC_OK;
}

// One wavefront batch of n samples per owned pixel on lane `l`, fully asynchronous.  Batches on different
// lanes overlap; only the per-pixel accumulation is ordered (sample order), through the acc_done events.
int run_batch(ptc_ctx* c, int l, uint32_t first_sample, uint32_t n_samples) {
  const uint32_t n_paths = c->fr.n_owned * n_samples;
  hipStream_t st = lane_stream(c, l);
  const DevQueues& q = lane_q(c, l);
  const DevScene sc = lane_scene(c, l);
  ScopedSpan whole(c, st, 3);
  pt_launch_set_counts(st, q, n_paths, 0);
  if (c->integrator == PTC_INTEGRATOR_RASTER_COMPAT) {
    pt_launch_raygen(st, c->cam, c->fr, q, 0, 1, true);
    { ScopedSpan t(c, st, 0); pt_launch_trace_closest(st, c->cfg, sc, q, 0, true); c->stats.launches_trace_closest++; }
    pt_launch_shade_raster(st, sc, c->cam, c->fr, q, c->accum.p);
  } else {
    pt_launch_raygen(st, c->cam, c->fr, q, first_sample, n_samples, false);
    for (int b = 0; b <= c->fr.max_bounces; ++b) {
      { ScopedSpan t(c, st, 0); pt_launch_trace_closest(st, c->cfg, sc, q, b & 1, false); c->stats.launches_trace_closest++; }
      { ScopedSpan t(c, st, 2); pt_launch_shade(st, c->cfg, sc, c->fr, q, b & 1, (uint32_t)b); }
      if (b < c->fr.max_bounces && (sc.n_lights > 0 || sc.env_ok)) {
        ScopedSpan t(c, st, 1); pt_launch_trace_any(st, c->cfg, sc, q, nullptr); c->stats.launches_trace_any++;
      }
      pt_launch_advance(st, q);
    }
    // sample-order accumulation: wait for the previous batch's accumulate (it ran on the previous lane)
    if (c->n_lanes > 1 && c->batches_issued > 0) {
      const int prev = (int)((c->batches_issued - 1) % (uint64_t)c->n_lanes);
      if (prev != l) HIP_TRY(c, hipStreamWaitEvent(st, lane_acc_event(c, prev), 0));
    }
    pt_launch_accumulate(st, c->fr, q, c->accum.p, n_samples);
    if (c->n_lanes > 1) HIP_TRY(c, hipEventRecord(lane_acc_event(c, l), st));
  }
  c->batches_issued++;
  HIP_TRY(c, hipGetLastError());
  return PTC_OK;
}

int sync_all_lanes(ptc_ctx* c) {
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (auto& ln : c->extra) HIP_TRY(c, hipStreamSynchronize(ln.stream));
  return PTC_OK;
}

void collect_times(ptc_ctx* c) {
  for (const Span& s : c->spans) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) continue;
    const double sec = 1e-3 * (double)ms;
    if (s.kind == 0) c->stats.seconds_trace_closest += sec;
    else if (s.kind == 1) c->stats.seconds_trace_any += sec;
    else if (s.kind == 2) c->stats.seconds_shade += sec;
    else c->stats.seconds_render += sec;
  }
  c->spans.clear();
  c->events_used = 0;
}

}  // namespace

// =================================================================================================
extern "C" {

int ptc_abi_version(void) { return PTC_ABI_VERSION; }

ptc_ctx* ptc_create(int device_id) {
  if (device_id == PTC_DEVICE_NONE) {   // description-only context: host flatten + BVH build, no rendering
    ptc_ctx* c = new ptc_ctx();
    c->device = PTC_DEVICE_NONE;
    if (const char* s = std::getenv("PTC_NODELETS")) c->nodelet_budget = (uint32_t)std::strtoul(s, nullptr, 10);
    return c;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) { g_create_error = std::string("ptc_create: no HIP device (") + hipGetErrorString(e) + "); this library has no CPU path"; return nullptr; }
  if (device_id < 0 || device_id >= n) { g_create_error = "ptc_create: device id out of range"; return nullptr; }
  if ((e = hipSetDevice(device_id)) != hipSuccess) { g_create_error = std::string("ptc_create: hipSetDevice: ") + hipGetErrorString(e); return nullptr; }
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess) { g_create_error = std::string("ptc_create: ") + hipGetErrorString(e); return nullptr; }
  if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) { g_create_error = std::string("ptc_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only"; return nullptr; }
  ptc_ctx* c = new ptc_ctx();
  c->device = device_id;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { g_create_error = std::string("ptc_create: ") + hipGetErrorString(e); delete c; return nullptr; }
  c->cfg.n_cu = prop.multiProcessorCount;
  c->cfg.trace_blocks_per_cu = 4;
  c->cfg.stack_lds = 12;
  if (const char* s = std::getenv("PTC_NODELETS")) c->nodelet_budget = (uint32_t)std::strtoul(s, nullptr, 10);
  if (const char* s = std::getenv("PTC_BATCH_PATHS")) { size_t v = std::strtoull(s, nullptr, 10); if (v >= 1024) c->max_batch_paths = v; }
  if (const char* s = std::getenv("PTC_TIMING")) c->timing = std::atoi(s) != 0;
  if (const char* s = std::getenv("PTC_LANES")) { int v = std::atoi(s); if (v >= 1 && v <= 8) c->n_lanes = v; }
  bool ok = hipEventCreateWithFlags(&c->acc_done0, hipEventDisableTiming) == hipSuccess;
  c->extra.resize((size_t)c->n_lanes - 1);
  for (auto& ln : c->extra)
    ok = ok && hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ln.acc_done, hipEventDisableTiming) == hipSuccess;
  if (!ok) { g_create_error = "ptc_create: could not create the lane streams / events"; ptc_destroy(c); return nullptr; }
  return c;
}

void ptc_destroy(ptc_ctx* c) {
  if (!c) return;
  if (c->device < 0) { delete c; return; }
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto& ln : c->extra) {
    if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); }
    if (ln.acc_done) (void)hipEventDestroy(ln.acc_done);
    free_all(ln.allocs);
  }
  if (c->acc_done0) (void)hipEventDestroy(c->acc_done0);
  free_all(c->scene_allocs); free_all(c->queue_allocs);
  c->owned.release(); c->accum.release(); c->radiance.release(); c->ldr.release();
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char* ptc_last_error(const ptc_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int ptc_scene_begin(ptc_ctx* c) {
  if (!c) return PTC_E_ARG;
  if (c->device >= 0) {
    HIP_TRY(c, hipSetDevice(c->device));
    { int rs = sync_all_lanes(c); if (rs) return rs; }
  }
  c->mats.clear(); c->meshes.clear(); c->insts.clear(); c->texs.clear(); c->env = HostEnv{}; c->tex_linear = 0;
  c->have_cam = false; c->committed = false; c->in_frame = false;
  free_all(c->scene_allocs);
  return PTC_OK;
}

int ptc_add_material(ptc_ctx* c, const float base_color[4], float metallic, float roughness, const float emissive[3],
                     int tex_color, int tex_normal, int tex_mr) {
  if (!c) return PTC_E_ARG;
  if (!base_color || !emissive) return fail(c, PTC_E_ARG, "add_material: null pointer");
  const int nt = (int)c->texs.size();
  if (tex_color >= nt || tex_normal >= nt || tex_mr >= nt) return fail(c, PTC_E_ARG, "add_material: texture id out of range");
  HostMaterial m;
  std::memcpy(m.base, base_color, 16); m.metallic = metallic; m.roughness = roughness; std::memcpy(m.emissive, emissive, 12);
  m.tex_color = tex_color; m.tex_normal = tex_normal; m.tex_mr = tex_mr;
  c->mats.push_back(m);
  return (int)c->mats.size() - 1;
}

int ptc_add_texture_rgba8(ptc_ctx* c, const uint8_t* px, int w, int h) {
  if (!c) return PTC_E_ARG;
  if (!px || w <= 0 || h <= 0) return fail(c, PTC_E_ARG, "add_texture: bad argument");
  HostTexture t;
  t.px.assign(px, px + (size_t)w * h * 4); t.w = w; t.h = h;
  c->texs.push_back(std::move(t));
  return (int)c->texs.size() - 1;
}

int ptc_add_mesh(ptc_ctx* c, const ptc_vertex* verts, uint32_t n_verts, const uint32_t* indices, uint32_t n_indices, int material) {
  if (!c) return PTC_E_ARG;
  if (!verts || !indices || n_verts == 0 || n_indices == 0 || (n_indices % 3u)) return fail(c, PTC_E_ARG, "add_mesh: bad argument");
  if (material < 0 || material >= (int)c->mats.size()) return fail(c, PTC_E_ARG, "add_mesh: material out of range");
  for (uint32_t i = 0; i < n_indices; ++i) if (indices[i] >= n_verts) return fail(c, PTC_E_ARG, "add_mesh: index out of range");
  HostMesh m;
  m.v.resize(n_verts);
  static_assert(sizeof(HostVertex) == sizeof(ptc_vertex) && sizeof(ptc_vertex) == 48, "R1 vertex record is 48 bytes");
  std::memcpy(m.v.data(), verts, (size_t)n_verts * sizeof(ptc_vertex));
  m.idx.assign(indices, indices + n_indices);
  m.material = material;
  c->meshes.push_back(std::move(m));
  return (int)c->meshes.size() - 1;
}

int ptc_add_instance(ptc_ctx* c, int mesh, const float t[3], const float q_wxyz[4], const float s[3]) {
  if (!c) return PTC_E_ARG;
  if (!t || !q_wxyz || !s) return fail(c, PTC_E_ARG, "add_instance: null pointer");
  if (mesh < 0 || mesh >= (int)c->meshes.size()) return fail(c, PTC_E_ARG, "add_instance: mesh out of range");
  HostInstance in;
  in.mesh = mesh;
  ptc_trs_to_matrix(t, q_wxyz, s, in.m);
  c->insts.push_back(in);
  return (int)c->insts.size() - 1;
}

int ptc_add_instance_matrix(ptc_ctx* c, int mesh, const float model[16]) {
  if (!c) return PTC_E_ARG;
  if (!model) return fail(c, PTC_E_ARG, "add_instance_matrix: null pointer");
  if (mesh < 0 || mesh >= (int)c->meshes.size()) return fail(c, PTC_E_ARG, "add_instance_matrix: mesh out of range");
  HostInstance in;
  in.mesh = mesh;
  std::memcpy(in.m, model, 64);
  c->insts.push_back(in);
  return (int)c->insts.size() - 1;
}

int ptc_set_camera(ptc_ctx* c, const float pos[3], const float target[3], float fov_y, float aspect) {
  if (!c) return PTC_E_ARG;
  if (!pos || !target) return fail(c, PTC_E_ARG, "set_camera: null pointer");
  std::memcpy(c->cam_pos, pos, 12); std::memcpy(c->cam_target, target, 12); c->cam_fov = fov_y; c->cam_aspect = aspect;
  c->have_cam = true;
  if (c->committed) ptc_make_camera(c->cam_pos, c->cam_target, c->cam_fov, c->cam_aspect, c->cam);
  return PTC_OK;
}

int ptc_set_texture_filter(ptc_ctx* c, int filter) {
  if (!c) return PTC_E_ARG;
  if (filter != PTC_FILTER_NEAREST && filter != PTC_FILTER_LINEAR) return fail(c, PTC_E_ARG, "set_texture_filter: unknown filter");
  c->tex_linear = filter;
  return PTC_OK;
}

int ptc_set_env_latlong_rgb32f(ptc_ctx* c, const float* rgb, int w, int h) {
  if (!c) return PTC_E_ARG;
  if (!rgb) { c->env = HostEnv{}; return PTC_OK; }
  if (w <= 0 || h <= 0 || (uint64_t)w * (uint64_t)h > (1u << 28)) return fail(c, PTC_E_ARG, "set_env: bad size");
  c->env.rgb.assign(rgb, rgb + (size_t)w * h * 3); c->env.w = w; c->env.h = h;
  return PTC_OK;
}

int ptc_scene_commit(ptc_ctx* c) {
  if (!c) return PTC_E_ARG;
  if (!c->have_cam) return fail(c, PTC_E_STATE, "scene_commit: no camera");
  if (c->device >= 0) HIP_TRY(c, hipSetDevice(c->device));
  const auto t0 = std::chrono::steady_clock::now();
  const std::string e = ptc_build_scene(c->mats, c->meshes, c->insts, c->texs, c->env, c->nodelet_budget, c->built);
  if (!e.empty()) return fail(c, PTC_E_STATE, e);
  ptc_make_camera(c->cam_pos, c->cam_target, c->cam_fov, c->cam_aspect, c->cam);
  if (c->device < 0) {   // description-only context: nothing to upload
    c->committed = true; c->in_frame = false;
    std::memset(&c->stats, 0, sizeof c->stats);
    c->stats.seconds_commit = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->stats.n_triangles = c->built.n_tris; c->stats.n_bvh_nodes = c->built.n_nodes; c->stats.n_emitters = c->built.n_lights;
    c->stats.bvh_max_depth = c->built.max_depth;
    return PTC_OK;
  }
  free_all(c->scene_allocs);
  const HostBuilt& B = c->built;
  DevScene d{};
  int rc = 0;
  {
    const float* p = nullptr;
    rc |= dev_upload(c, c->scene_allocs, &p, B.nodes); d.nodes = (const float4*)p;
    rc |= dev_upload(c, c->scene_allocs, &p, B.tris); d.tris = (const float4*)p;
    rc |= dev_upload(c, c->scene_allocs, &p, B.mats); d.mats = (const float4*)p;
    rc |= dev_upload(c, c->scene_allocs, &p, B.lights); d.lights = (const float4*)p;
    rc |= dev_upload(c, c->scene_allocs, &d.cdf, B.cdf);
    rc |= dev_upload(c, c->scene_allocs, &p, B.shade); d.shade = (const float4*)p;
    rc |= dev_upload(c, c->scene_allocs, &p, B.shade_tex); d.shade_tex = (const float4*)p;
    rc |= dev_upload(c, c->scene_allocs, &d.texels, B.texels);
    { const int32_t* ti = nullptr; rc |= dev_upload(c, c->scene_allocs, &ti, B.tex_info); d.tex_info = (const int4*)ti; }
    rc |= dev_upload(c, c->scene_allocs, &p, B.env); d.env = (const float4*)p;
    rc |= dev_upload(c, c->scene_allocs, &d.env_marg, B.env_marg);
    rc |= dev_upload(c, c->scene_allocs, &d.env_cond, B.env_cond);
  }
  if (rc) { free_all(c->scene_allocs); return PTC_E_DEVICE; }
  d.env_w = B.env_w; d.env_h = B.env_h; d.env_ok = B.env_ok;
  d.tex_linear = c->tex_linear;
  d.n_lights = B.n_lights; d.n_mats = (uint32_t)c->mats.size(); d.n_nodelets = B.n_nodelets; d.ray_eps = B.ray_eps;
  c->dsc = d;
  { int rc2 = configure_launch(c); if (rc2) { free_all(c->scene_allocs); return rc2; } }
  c->committed = true; c->in_frame = false;
  std::memset(&c->stats, 0, sizeof c->stats);
  c->stats.seconds_commit = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  c->stats.n_triangles = B.n_tris; c->stats.n_bvh_nodes = B.n_nodes; c->stats.n_emitters = B.n_lights; c->stats.bvh_max_depth = B.max_depth;
  return PTC_OK;
}

int ptc_frame_begin(ptc_ctx* c, int w, int h, int spp_total, uint64_t seed, int max_bounces, int integrator, int tile_rank, int tile_count) {
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  if (!c->committed) return fail(c, PTC_E_STATE, "frame_begin: scene not committed");
  if (w <= 0 || h <= 0 || spp_total <= 0 || max_bounces < 0 || (uint64_t)w * (uint64_t)h > 0x7fffffffull) return fail(c, PTC_E_ARG, "frame_begin: bad size");
  if (integrator != PTC_INTEGRATOR_PATH && integrator != PTC_INTEGRATOR_RASTER_COMPAT) return fail(c, PTC_E_ARG, "frame_begin: unknown integrator");
  if (tile_count < 1 || tile_rank < 0 || tile_rank >= tile_count) return fail(c, PTC_E_ARG, "frame_begin: bad tile rank/count");
  HIP_TRY(c, hipSetDevice(c->device));
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  std::vector<uint32_t> owned;
  ptc_owned_pixels(w, h, tile_rank, tile_count, owned);
  int rc;
  if ((rc = ensure_buf_u32(c, c->owned, owned.size()))) return rc;
  if (!owned.empty()) HIP_TRY(c, hipMemcpy(c->owned.p, owned.data(), owned.size() * 4, hipMemcpyHostToDevice));
  if ((rc = ensure_buf_f4(c, c->accum, owned.size()))) return rc;
  if ((rc = ensure_buf_f4(c, c->radiance, (size_t)w * h))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->accum.p, 0, (owned.size() ? owned.size() : 1) * sizeof(float4), c->stream));
  HIP_TRY(c, hipMemsetAsync(c->radiance.p, 0, (size_t)w * h * sizeof(float4), c->stream));
  c->rad_w = w; c->rad_h = h;
  c->fr.w = w; c->fr.h = h; c->fr.max_bounces = max_bounces; c->fr.n_owned = (uint32_t)owned.size(); c->fr.owned = c->owned.p;
  {  // seed_hash = pcg(seed_lo + pcg(seed_hi)), same hash as pt_device.h
    auto pcg = [](uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t x = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (x >> 22) ^ x; };
    c->fr.seed_hash = pcg((uint32_t)seed + pcg((uint32_t)(seed >> 32)));
  }
  c->spp_total = integrator == PTC_INTEGRATOR_RASTER_COMPAT ? 1 : spp_total;
  c->integrator = integrator; c->samples_done = 0;
  // queue capacity per lane: as many samples per batch as fit max_batch_paths split over the lanes
  size_t per = owned.empty() ? 1 : c->max_batch_paths / owned.size() / (size_t)c->n_lanes;
  if (per < 1) per = 1;
  if (per > (size_t)c->spp_total) per = (size_t)c->spp_total;
  const size_t cap = (owned.empty() ? 1 : owned.size()) * per;
  if (cap > 0xfffffff0ull) return fail(c, PTC_E_ARG, "frame_begin: batch too large");
  if ((rc = ensure_queues(c, (uint32_t)cap))) return rc;
  for (auto& ln : c->extra) if ((rc = ensure_queues_in(c, ln.q, ln.allocs, (uint32_t)cap))) return rc;
  for (int l = 0; l < c->n_lanes; ++l) HIP_TRY(c, hipMemsetAsync(lane_q(c, l).stats, 0, ST_N * sizeof(unsigned long long), lane_stream(c, l)));
  { int rs = sync_all_lanes(c); if (rs) return rs; }     // accum/radiance/statistics are cleared before any lane starts
  c->batches_issued = 0;
  ptc_stats keep = c->stats;
  std::memset(&c->stats, 0, sizeof c->stats);
  c->stats.seconds_commit = keep.seconds_commit; c->stats.n_triangles = keep.n_triangles; c->stats.n_bvh_nodes = keep.n_bvh_nodes;
  c->stats.n_emitters = keep.n_emitters; c->stats.bvh_max_depth = keep.bvh_max_depth;
  c->spans.clear(); c->events_used = 0;
  c->in_frame = true;
  return PTC_OK;
}

int ptc_frame_add_samples(ptc_ctx* c, int n_samples) {
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  if (!c->in_frame) return fail(c, PTC_E_STATE, "frame_add_samples: no frame");
  if (n_samples <= 0) return fail(c, PTC_E_ARG, "frame_add_samples: n_samples <= 0");
  HIP_TRY(c, hipSetDevice(c->device));
  if (c->fr.n_owned == 0) { c->samples_done += (uint32_t)n_samples; return PTC_OK; }
  if (c->integrator == PTC_INTEGRATOR_RASTER_COMPAT) {
    if (c->samples_done == 0) { int rc = run_batch(c, 0, 0, 1); if (rc) return rc; }
    c->samples_done += (uint32_t)n_samples;
    c->stats.paths = c->fr.n_owned;
    return PTC_OK;
  }
  uint32_t left = (uint32_t)n_samples;
  const uint32_t per = c->q.cap / c->fr.n_owned;
  while (left) {
    const uint32_t k = left < per ? left : per;
    int rc = run_batch(c, (int)(c->batches_issued % (uint64_t)c->n_lanes), c->samples_done, k);
    if (rc) return rc;
    c->samples_done += k; left -= k;
    c->stats.paths += (uint64_t)c->fr.n_owned * k;
  }
  return PTC_OK;
}

int ptc_frame_resolve(ptc_ctx* c) {
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  if (!c->in_frame) return fail(c, PTC_E_STATE, "frame_resolve: no frame");
  HIP_TRY(c, hipSetDevice(c->device));
  // the last accumulate may have run on another lane: stream 0 waits for every lane's accumulate event
  if (c->n_lanes > 1 && c->integrator == PTC_INTEGRATOR_PATH && c->batches_issued > 0)
    for (int l = 1; l < c->n_lanes && (uint64_t)l < c->batches_issued + 1; ++l)
      if ((uint64_t)l < c->batches_issued) HIP_TRY(c, hipStreamWaitEvent(c->stream, lane_acc_event(c, l), 0));
  if (c->fr.n_owned) pt_launch_resolve(c->stream, c->fr, c->accum.p, c->radiance.p, (float)c->spp_total, c->integrator == PTC_INTEGRATOR_RASTER_COMPAT);
  HIP_TRY(c, hipGetLastError());
  return PTC_OK;
}

int ptc_sync(ptc_ctx* c) {
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  HIP_TRY(c, hipSetDevice(c->device));
  return sync_all_lanes(c);
}

int ptc_render(ptc_ctx* c, int w, int h, int spp, uint64_t seed, int max_bounces, int integrator) {
  int rc = ptc_frame_begin(c, w, h, spp, seed, max_bounces, integrator, 0, 1);
  if (rc) return rc;
  if ((rc = ptc_frame_add_samples(c, spp))) return rc;
  if ((rc = ptc_frame_resolve(c))) return rc;
  return ptc_sync(c);
}

int ptc_read_radiance_rgba32f(ptc_ctx* c, float* out) {
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  if (!out) return fail(c, PTC_E_ARG, "read_radiance: null pointer");
  if (!c->radiance.p || c->rad_w == 0) return fail(c, PTC_E_STATE, "read_radiance: nothing rendered");
  HIP_TRY(c, hipSetDevice(c->device));
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  HIP_TRY(c, hipMemcpy(out, c->radiance.p, (size_t)c->rad_w * c->rad_h * sizeof(float4), hipMemcpyDeviceToHost));
  return PTC_OK;
}

void* ptc_radiance_device_ptr(ptc_ctx* c) { return (c && c->device >= 0) ? (void*)c->radiance.p : nullptr; }

int ptc_write_radiance_rgba32f(ptc_ctx* c, const float* in) {
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  if (!in) return fail(c, PTC_E_ARG, "write_radiance: null pointer");
  if (!c->radiance.p || c->rad_w == 0) return fail(c, PTC_E_STATE, "write_radiance: no frame");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(c->radiance.p, in, (size_t)c->rad_w * c->rad_h * sizeof(float4), hipMemcpyHostToDevice));
  return PTC_OK;
}

int ptc_tonemap_rgba8(ptc_ctx* c, uint8_t* out) {
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  if (!out) return fail(c, PTC_E_ARG, "tonemap: null pointer");
  if (!c->radiance.p || c->rad_w == 0) return fail(c, PTC_E_STATE, "tonemap: nothing rendered");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure_buf_u32(c, c->ldr, (size_t)c->rad_w * c->rad_h))) return rc;
  pt_launch_tonemap(c->stream, c->radiance.p, c->ldr.p, c->rad_w, c->rad_h);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(out, c->ldr.p, (size_t)c->rad_w * c->rad_h * 4, hipMemcpyDeviceToHost));
  return PTC_OK;
}

int ptc_get_stats(ptc_ctx* c, ptc_stats* out) {
  if (!c) return PTC_E_ARG;
  if (!out) return fail(c, PTC_E_ARG, "get_stats: null pointer");
  if (c->device < 0) { *out = c->stats; return PTC_OK; }
  HIP_TRY(c, hipSetDevice(c->device));
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  if (c->q.stats) {
    unsigned long long st[ST_N] = {0};
    for (int l = 0; l < c->n_lanes; ++l) {
      if (!lane_q(c, l).stats) continue;
      unsigned long long one[ST_N];
      HIP_TRY(c, hipMemcpy(one, lane_q(c, l).stats, sizeof one, hipMemcpyDeviceToHost));
      for (int i = 0; i < ST_N; ++i) st[i] += one[i];
    }
    ptc_stats& s = c->stats;
    s.segments = st[ST_SEGMENTS]; s.shadow_rays = st[ST_SHADOW]; s.hits = st[ST_HITS];
    s.node_visits_closest = st[ST_NODES_C]; s.tri_tests_closest = st[ST_TRIS_C];
    s.node_visits_any = st[ST_NODES_A]; s.tri_tests_any = st[ST_TRIS_A];
    // SURVEY §8d byte model with this build's record sizes (DESIGN.md §"Algorithmic bytes")
    s.algorithmic_bytes = s.segments * (2u * 56u + 2u * 16u) + s.node_visits_closest * 48u + s.tri_tests_closest * 48u + s.hits * 176u +
                          s.shadow_rays * (2u * 44u) + s.node_visits_any * 48u + s.tri_tests_any * 48u + s.paths * (2u * 16u);
  }
  collect_times(c);
  *out = c->stats;
  return PTC_OK;
}

// ---- test hooks -----------------------------------------------------------------------------------
int ptc_debug_trace_closest(ptc_ctx* c, const float* origins, const float* dirs, uint32_t n, float* out_t, int32_t* out_prim, float* out_uv) {
  if (c && c->device >= 0) { int rs = sync_all_lanes(c); if (rs) return rs; }
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  if (!c->committed) return fail(c, PTC_E_STATE, "debug_trace_closest: scene not committed");
  if (!origins || !dirs || !out_t || !out_prim || !out_uv || n == 0) return fail(c, PTC_E_ARG, "debug_trace_closest: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->in_frame = false;
  int rc;
  if ((rc = ensure_queues(c, n))) return rc;
  std::vector<float4> A(n), B(n);
  for (uint32_t i = 0; i < n; ++i) {
    A[i] = make_float4(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2], dirs[i * 3]);
    B[i] = make_float4(dirs[i * 3 + 1], dirs[i * 3 + 2], 0.0f, 0.0f);
  }
  HIP_TRY(c, hipMemcpy(c->q.ray[0].A, A.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->q.ray[0].B, B.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  for (int l = 0; l < c->n_lanes; ++l) if (lane_q(c, l).stats) HIP_TRY(c, hipMemset(lane_q(c, l).stats, 0, ST_N * sizeof(unsigned long long)));
  pt_launch_set_counts(c->stream, c->q, n, 0);
  pt_launch_trace_closest(c->stream, c->cfg, c->dsc, c->q, 0, false);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  std::vector<float4> H(n);
  HIP_TRY(c, hipMemcpy(H.data(), c->q.hit, n * sizeof(float4), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; ++i) {
    int32_t pc; std::memcpy(&pc, &H[i].y, 4);   // prim | class<<28, or -1
    out_t[i] = H[i].x; out_prim[i] = pc < 0 ? -1 : (pc & 0x0fffffff); out_uv[i * 2] = H[i].z; out_uv[i * 2 + 1] = H[i].w;
  }
  return PTC_OK;
}

int ptc_debug_trace_any(ptc_ctx* c, const float* origins, const float* dirs, const float* tmax, uint32_t n, uint8_t* out_occluded) {
  if (c && c->device >= 0) { int rs = sync_all_lanes(c); if (rs) return rs; }
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  if (!c->committed) return fail(c, PTC_E_STATE, "debug_trace_any: scene not committed");
  if (!origins || !dirs || !tmax || !out_occluded || n == 0) return fail(c, PTC_E_ARG, "debug_trace_any: bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->in_frame = false;
  int rc;
  if ((rc = ensure_queues(c, n))) return rc;
  std::vector<float4> A(n), B(n);
  for (uint32_t i = 0; i < n; ++i) {
    A[i] = make_float4(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2], dirs[i * 3]);
    B[i] = make_float4(dirs[i * 3 + 1], dirs[i * 3 + 2], tmax[i], 0.0f);
  }
  HIP_TRY(c, hipMemcpy(c->q.shadow.A, A.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->q.shadow.B, B.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  uint8_t* d_out = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d_out, n));
  for (int l = 0; l < c->n_lanes; ++l) if (lane_q(c, l).stats) HIP_TRY(c, hipMemset(lane_q(c, l).stats, 0, ST_N * sizeof(unsigned long long)));
  pt_launch_set_counts(c->stream, c->q, 0, n);
  pt_launch_trace_any(c->stream, c->cfg, c->dsc, c->q, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(out_occluded, d_out, n, hipMemcpyDeviceToHost);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail(c, PTC_E_DEVICE, std::string("debug_trace_any: ") + hipGetErrorString(e));
  return PTC_OK;
}

int ptc_debug_get_flat_scene(ptc_ctx* c, uint32_t* n_verts, uint32_t* n_tris, ptc_vertex* verts, uint32_t* indices, int32_t* tri_material) {
  if (!c) return PTC_E_ARG;
  if (!c->committed) return fail(c, PTC_E_STATE, "debug_get_flat_scene: scene not committed");
  const HostBuilt& B = c->built;
  if (n_verts) *n_verts = (uint32_t)B.wverts.size();
  if (n_tris) *n_tris = B.n_tris;
  if (verts) std::memcpy(verts, B.wverts.data(), B.wverts.size() * sizeof(ptc_vertex));
  if (indices) std::memcpy(indices, B.widx.data(), B.widx.size() * 4);
  if (tri_material) std::memcpy(tri_material, B.tri_mat.data(), B.tri_mat.size() * 4);
  return PTC_OK;
}

int ptc_debug_get_description(ptc_ctx* c, int* n_materials, int* n_textures) {
  if (!c) return PTC_E_ARG;
  if (n_materials) *n_materials = (int)c->mats.size();
  if (n_textures) *n_textures = (int)c->texs.size();
  return PTC_OK;
}

int ptc_debug_get_material(ptc_ctx* c, int index, float out_factors[9], int out_textures[3]) {
  if (!c) return PTC_E_ARG;
  if (index < 0 || (size_t)index >= c->mats.size() || !out_factors || !out_textures) return fail(c, PTC_E_ARG, "debug_get_material: bad argument");
  const HostMaterial& m = c->mats[(size_t)index];
  std::memcpy(out_factors, m.base, 16); out_factors[4] = m.metallic; out_factors[5] = m.roughness; std::memcpy(out_factors + 6, m.emissive, 12);
  out_textures[0] = m.tex_color; out_textures[1] = m.tex_normal; out_textures[2] = m.tex_mr;
  return PTC_OK;
}

int ptc_debug_get_texture(ptc_ctx* c, int index, int* w, int* h, uint8_t* rgba) {
  if (!c) return PTC_E_ARG;
  if (index < 0 || (size_t)index >= c->texs.size()) return fail(c, PTC_E_ARG, "debug_get_texture: bad argument");
  const HostTexture& t = c->texs[(size_t)index];
  if (w) *w = t.w;
  if (h) *h = t.h;
  if (rgba) std::memcpy(rgba, t.px.data(), t.px.size());
  return PTC_OK;
}

int ptc_debug_get_counters(ptc_ctx* c, uint64_t* out, int n) {
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path");
  if (!out || n <= 0) return fail(c, PTC_E_ARG, "debug_get_counters: bad argument");
  if (!c->q.stats) return fail(c, PTC_E_STATE, "debug_get_counters: nothing rendered");
  HIP_TRY(c, hipSetDevice(c->device));
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  unsigned long long st[ST_N] = {0};
  for (int l = 0; l < c->n_lanes; ++l) {
    if (!lane_q(c, l).stats) continue;
    unsigned long long one[ST_N];
    HIP_TRY(c, hipMemcpy(one, lane_q(c, l).stats, sizeof one, hipMemcpyDeviceToHost));
    for (int i = 0; i < ST_N; ++i) st[i] += one[i];
  }
  for (int i = 0; i < n; ++i) out[i] = i < ST_N ? st[i] : 0;
  return ST_N;
}

int ptc_debug_get_bvh(ptc_ctx* c, uint32_t* n_nodes, uint32_t* n_tris, float* nodes, float* tris) {
  if (!c) return PTC_E_ARG;
  if (!c->committed) return fail(c, PTC_E_STATE, "debug_get_bvh: scene not committed");
  const HostBuilt& B = c->built;
  if (n_nodes) *n_nodes = B.n_nodes;
  if (n_tris) *n_tris = B.n_tri_records;
  if (nodes) std::memcpy(nodes, B.nodes.data(), B.nodes.size() * 4);
  if (tris) std::memcpy(tris, B.tris.data(), B.tris.size() * 4);
  return PTC_OK;
}

}  // extern "C"
