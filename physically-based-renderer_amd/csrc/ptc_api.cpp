// ptc_api.cpp — the C-ABI of include/ptc.h over the HIP wavefront path tracer.
//
// One context = one HIP device + `n_lanes` streams ("lanes").  Everything a frame needs is enqueued without host
// synchronisation: queue sizes live in device memory and the persistent kernels read them there, so a whole batch
// (set_counts → raygen → [closest, shade, scan, any] × bounces → accumulate) is a single asynchronous burst.  With PTC_LANES > 1
// successive batches of a frame alternate between the lanes, so one batch's launch tails overlap another batch's
// full-occupancy phases (only the per-pixel accumulation is ordered, in sample order, by events); the default is one lane,
// which the round-2 kernels make the faster arrangement.  The host blocks only in ptc_sync / read-backs / ptc_get_stats.
//
// There is no CPU path in this library: without a usable HIP device ptc_create fails.
#include "../../include/ptc.h"
#include "ptc_internal.h"
#include "pt_refit.h"
#include "pt_build.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <memory>
#include <string>
#include <vector>

#define PTC_STR2(x) #x
#define PTC_STR(x) PTC_STR2(x)
namespace {
std::string g_create_error;

struct Span { hipEvent_t a, b; int kind; };   // kind: 0 trace_closest, 1 trace_any, 2 shade, 3 whole batch, 4 the RCCL reduce

template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0;
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

// One lane: a stream with its own wavefront queues.  cnt/stats are allocated once; the large arrays grow on demand.
struct Lane {
  hipStream_t stream = nullptr;
  DevQueues q{};
  std::vector<void*> allocs;        // the large queue arrays (sized q.cap)
  uint2* stack_ovf = nullptr;       // traversal-stack overflow slab (belongs to the committed scene)
  DevScene* d_scene = nullptr;      // this lane's DevScene in device memory (k_shade reads it through a pointer instead of ~200 B of kernel arguments)
  hipEvent_t acc_done = nullptr;    // "this lane's last accumulate finished"
  // PTC_TRACE_OVERLAP=1: the shadow rays of bounce b are traced on a second stream beside the closest-hit launch of bounce b + 1
  hipStream_t stream2 = nullptr;
  uint2* stack_ovf2 = nullptr;      // the any-hit launches' own overflow slab (concurrent kernels must not share one)
  std::vector<hipEvent_t> ev_scan, ev_any;
};

// ---- RCCL, loaded on first use (a renderer that never reduces does not need librccl at load time) -------------------
struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
Rccl g_rccl;
bool rccl_load() {
  if (g_rccl.so) return true;
  void* so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!so) so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!so) so = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!so) { const char* e = dlerror(); g_rccl.err = std::string("librccl.so not loadable: ") + (e ? e : "?"); return false; }
  bool ok = true;
  auto sym = [&](const char* name) { void* p = dlsym(so, name); if (!p) { ok = false; g_rccl.err = std::string("librccl.so lacks ") + name; } return p; };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.Reduce = (decltype(g_rccl.Reduce))sym("ncclReduce");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  if (!ok) { dlclose(so); return false; }
  g_rccl.so = so;
  return true;
}
}  // namespace

struct ptc_ctx {
  int device = 0;
  std::string err;
  LaunchCfg cfg{};
  uint32_t toplet_budget = 73;   // 64-byte records staged in LDS: the top three levels (1+8+64 nodes) of the tree = 4.6 KB
  size_t max_batch_paths = (size_t)7 << 27;   // paths in flight over all lanes (939,524,096).  Large batches amortise what a launch costs regardless of its size
                                              // (drain of the persistent waves, small late-bounce launches): round 2 measured 2^29 1.5 % faster than 2^28, 2^27 3 % and
                                              // 2^25 24 % slower; round 4 7 x 2^27 another 0.8 % faster than 2^29 (profiles/r04_trace_variants.txt).  176 B per path = 165 GB of
                                              // queues when a 1080p frame is rendered at >= 453 spp — sized for 288 GB of HBM; frame_begin lowers it to what 60 % of the free memory holds.
  int timing = 1;                     // PTC_TIMING: 0 no events at all; 1 (default) a span per batch, and a span per kernel where the kernels of a batch run one after the other
                                      // (a small batch runs its trace kernels beside each other: their spans would include each other, and 54 event records are 0.2 ms of a 3-ms frame); 2 a span per kernel always
  // description
  std::vector<HostMaterial> mats;
  std::vector<HostMesh> meshes;
  std::vector<HostInstance> insts;
  std::vector<HostTexture> texs;
  HostEnv env;
  float cam_pos[3]{}, cam_target[3]{}, cam_fov = 0, cam_aspect = 1;
  bool have_cam = false;
  int tex_linear = 0;                    // PTC_FILTER_*: texture filter of the scene being described
  int bvh_default = PTC_BVH_SAH;         // PTC_BVH_*: builder a new scene description starts with (PTC_BVH=lbvh in the environment changes it)
  int bvh_builder = PTC_BVH_SAH;         // builder of the scene being described
  // committed scene
  bool committed = false;
  size_t committed_insts = 0;       // instances the committed scene was built from (ptc_scene_refit refuses a description that has grown since)
  std::shared_ptr<HostBuilt> built = std::make_shared<HostBuilt>();   // the host build; the contexts of a ptc_group share one (ptc_group_scene_commit)
  DevScene dsc{};
  DevCamera cam{};
  std::vector<void*> scene_allocs;
  // refit on the device (pt_refit.h): the plan is built and uploaded by the first ptc_scene_refit after a commit
  int refit_on_device = 1;          // PTC_REFIT=host: ptc_scene_refit recomputes on the host and uploads (the round-3a path, kept as the cross-check)
  bool refit_ready = false;
  RefitPlan plan;
  DevRefit drf{};
  bool host_stale = false;          // the device refitted in place: built's vertex-dependent arrays are those of an earlier state until refresh_host_copy
  bool last_refit_on_device = false;
  bool commit_on_device = false;      // the last ptc_scene_commit flattened and built on the device (device_commit)
  int trace_rays_per_lane = 8;      // PTC_TRACE_RAYS_PER_LANE: rays per lane of the trace kernels' grid a batch should offer before the grid is made smaller (run_batch)
  int trace_overlap = 1;            // PTC_TRACE_OVERLAP: the shadow rays of bounce b are traced on the lane's second stream beside the closest-hit launch of bounce b + 1 (they are
                                    // independent; k_shade(b + 1) waits for both).  1 (default) = batches of up to 2^26 paths, whose launches do not keep the chip full for long:
                                    // -5 % .. -17 % frame time from 16 spp down to 1 spp at 1080p (profiles/r03_viewer_loop.txt); 2 = every batch (+0.4 % at the benchmark's
                                    // batch size, but the two kernels' launch durations then include each other: not the default, so that what bench.py and rocprofv3 time
                                    // per kernel stays a kernel's own time); 0 = never
  BuildScratch bscratch;            // device scratch of ptc_scene_rebuild (pt_build.hip), grow-only
  struct RebuildBufs { float4* recs = nullptr; size_t recs_cap = 0; uint32_t* levels = nullptr; size_t levels_cap = 0; float* nbox = nullptr; size_t nbox_cap = 0; };
  RebuildBufs rb_spare, rb_live;    // a rebuild writes the new tree into the spare unit array / level list / box array and the ones it replaces become the spare: no
                                    // allocation in a viewer's steady state (rb_live: capacities of the arrays in use; the spare set is not in scene_allocs)
  std::vector<float> xf_live;       // instance transforms of the last refit the device completed (a refused one re-flattens its scratch vertices from these)
  // lanes: lane 0 is the context's primary stream (resolve, tonemap, conversions, the reduce)
  std::vector<Lane> lanes;
  int n_lanes = 1;                  // PTC_LANES: >1 runs successive batches on separate streams.  With the round-2 kernels one lane
                                    // is 3.7 % faster than two (co-scheduled launches slow each other down by more than the tails they fill)
  uint64_t batches_issued = 0;
  // frame
  bool in_frame = false;
  DevFrame fr{};
  int spp_total = 0, integrator = 0;
  uint32_t samples_done = 0;        // samples issued to the device
  uint32_t sample_base = 0;         // index of the frame's first sample (ptc_frame_set_sample_range / ptc_frame_restore): sample k of the frame has index sample_base + k
  uint32_t resolve_divisor = 0;     // 0: the resolve divides by the samples accumulated; else by this (sample-range sharding: partial means that sum to the mean)
  uint32_t pending = 0;             // samples accepted by frame_add_samples and not yet issued (deferred batching)
  uint32_t per_batch = 1;           // samples of one full batch = max_batch_paths / owned pixels / lanes
  DevBuf<uint32_t> owned;
  bool owned_key_valid = false;     // c->owned holds the list for (owned_w, owned_h, owned_rank, owned_count)
  int owned_w = 0, owned_h = 0, owned_rank = 0, owned_count = 0;
  uint32_t owned_n = 0;
  DevBuf<float4> accum, radiance;
  DevBuf<uint32_t> ldr;
  DevBuf<uint2> half;               // RGBA16F copy of the radiance buffer
  int rad_w = 0, rad_h = 0;
  // multi-GPU
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_size = 0;
  bool comm_owned = true;           // false: the communicator belongs to a ptc_group
  // stats
  ptc_stats stats{};
  std::vector<Span> spans;
  std::vector<hipEvent_t> free_events;
  size_t events_created = 0;
};

struct ptc_group {
  std::vector<ptc_ctx*> ctx;
  std::vector<ncclComm_t> comms;
  std::string err;
};

namespace {

constexpr size_t kQueueBytesPerPath = 176;   // ensure_lane_queues: 2 x 48 (ray ping-pong) + 48 (shadow) + 16 (hit) + 16 (path radiance)
constexpr size_t kMaxSpans = 1024;   // timing spans (event pairs) kept at most; see run_batch
int fail(ptc_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
const char* const kNoDevice = "this context has no device (PTC_DEVICE_NONE): the call needs a gfx950 GPU; there is no CPU path";

#define HIP_TRY(c, expr)                                                                                 \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess)                                                                                \
      return fail((c), e_ == hipErrorOutOfMemory ? PTC_E_NOMEM : PTC_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define NCCL_TRY(c, expr)                                                                                \
  do {                                                                                                   \
    ncclResult_t r_ = (expr);                                                                            \
    if (r_ != ncclSuccess) return fail((c), PTC_E_DEVICE, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
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
void free_rebuild_spare(ptc_ctx* c) {      // the arrays a rebuild replaced (kept for the next one); the live set is in scene_allocs
  if (c->rb_spare.recs) (void)hipFree(c->rb_spare.recs);
  if (c->rb_spare.levels) (void)hipFree(c->rb_spare.levels);
  if (c->rb_spare.nbox) (void)hipFree(c->rb_spare.nbox);
  c->rb_spare = ptc_ctx::RebuildBufs(); c->rb_live = ptc_ctx::RebuildBufs();
}

template <class T> int ensure_buf(ptc_ctx* c, DevBuf<T>& b, size_t n) {
  if (b.n >= n && b.p) return PTC_OK;
  b.release();
  HIP_TRY(c, hipMalloc((void**)&b.p, (n ? n : 1) * sizeof(T)));
  b.n = n;
  return PTC_OK;
}

int sync_all_lanes(ptc_ctx* c) {
  for (auto& ln : c->lanes) HIP_TRY(c, hipStreamSynchronize(ln.stream));
  return PTC_OK;
}

// Every lane gets queues for at least `cap` paths (grow only; all lanes keep the same capacity, so a batch fits whichever
// lane it is routed to).  Growing waits for the work in flight first: the old arrays may still be in use.
int ensure_lane_queues(ptc_ctx* c, uint32_t cap) {
  bool grow = false;
  for (auto& ln : c->lanes) grow = grow || ln.q.cap < cap;
  if (!grow) return PTC_OK;
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  for (auto& ln : c->lanes) {
    if (ln.q.cap >= cap) continue;
    free_all(ln.allocs);
    ln.q.cap = 0;
    DevQueues q = ln.q;   // keeps cnt / stats / the per-segment arrays
    int rc = 0;
    const size_t slots = ptc_seg_slots(cap, (uint32_t)c->cfg.shade_waves);   // segments are padded to multiples of 64 slots
    for (int k = 0; k < 2 && !rc; ++k) {
      rc = dev_alloc(c, ln.allocs, &q.ray[k].A, slots); if (!rc) rc = dev_alloc(c, ln.allocs, &q.ray[k].B, slots);
      if (!rc) rc = dev_alloc(c, ln.allocs, &q.ray[k].C, slots);
    }
    if (!rc) rc = dev_alloc(c, ln.allocs, &q.shadow.A, slots);
    if (!rc) rc = dev_alloc(c, ln.allocs, &q.shadow.B, slots);
    if (!rc) rc = dev_alloc(c, ln.allocs, &q.shadow.C, slots);
    if (!rc) rc = dev_alloc(c, ln.allocs, &q.hit, slots);
    if (!rc) rc = dev_alloc(c, ln.allocs, &q.lpath, cap);
    if (rc) { free_all(ln.allocs); return rc; }
    q.cap = cap;
    ln.q = q;
  }
  return PTC_OK;
}

// ---- timing spans: event pairs around the kernels of a batch; events are recycled as soon as they have completed, so a
// progressive loop that never asks for statistics does not grow the pool -----------------------------------------------
hipEvent_t next_event(ptc_ctx* c) {
  if (!c->free_events.empty()) { hipEvent_t e = c->free_events.back(); c->free_events.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  c->events_created++;
  return e;
}
void collect_times(ptc_ctx* c, bool all_done) {
  size_t keep = 0;
  for (size_t i = 0; i < c->spans.size(); ++i) {
    const Span s = c->spans[i];
    if (!all_done && hipEventQuery(s.b) != hipSuccess) { c->spans[keep++] = s; continue; }
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      const double sec = 1e-3 * (double)ms;
      if (s.kind == 0) c->stats.seconds_trace_closest += sec;
      else if (s.kind == 1) c->stats.seconds_trace_any += sec;
      else if (s.kind == 2) c->stats.seconds_shade += sec;
      else if (s.kind == 4) c->stats.seconds_reduce += sec;
      else c->stats.seconds_render += sec;
    }
    c->free_events.push_back(s.a); c->free_events.push_back(s.b);
  }
  c->spans.resize(keep);
}
struct ScopedSpan {   // records a start/stop event pair around launches on one of the context's streams
  ptc_ctx* c; hipStream_t st; Span s{}; bool on;
  ScopedSpan(ptc_ctx* c_, hipStream_t st_, int kind, bool enable = true) : c(c_), st(st_), on(c_->timing != 0 && enable) {
    if (!on) return;
    s.kind = kind; s.a = next_event(c); s.b = next_event(c);
    if (!s.a || !s.b) { if (s.a) c->free_events.push_back(s.a); on = false; return; }
    (void)hipEventRecord(s.a, st);
  }
  ~ScopedSpan() { if (on) { (void)hipEventRecord(s.b, st); c->spans.push_back(s); } }
};

// pt_trace_blocks_per_cu asks the runtime four occupancy questions; the answer depends on the kernels and the LDS size alone (every device is a gfx950), so it is asked once per
// size and process (sizes above 64 KiB also set a function attribute per device and are asked every time)
int trace_blocks_per_cu_cached(size_t lds) {
  static std::mutex mu;
  static std::map<size_t, int> known;
  if (lds > 64u * 1024u) return pt_trace_blocks_per_cu(lds);
  { std::lock_guard<std::mutex> lock(mu); const auto it = known.find(lds); if (it != known.end()) return it->second; }
  const int v = pt_trace_blocks_per_cu(lds);
  if (v > 0) { std::lock_guard<std::mutex> lock(mu); known[lds] = v; }
  return v;
}

int configure_launch(ptc_ctx* c) {
  // Traversal stack: at most one group of pending children per tree level, so a ray needs at most depth+1 entries.
  // `stack_lds` of them live in LDS (8 B each, 512 B per level and wave), the rest in a global overflow slab.
  // LDS per block = staged top of the tree (1.2 KB) + waves·stack_lds·512 B + the 2-KiB slot-order table + the waves' prepared rays (5 KB) + 512 B static.
  // Default: the most stack entries (at most 6) with which the register limit of 8 blocks (32 waves) per CU still fits the 160 KiB of LDS — 5 since the
  // prepared rays of round 4 (19.1 KB per block).
  const int need = (int)c->built->max_depth + 2;
  int l = 6, per_cu = 0;
  bool l_forced = false;
  if (const char* e = std::getenv("PTC_STACK_LDS")) { int v = std::atoi(e); if (v >= 1 && v <= 64) { l = v; l_forced = true; } }
  if (l > need) l = need;
  size_t lds = 0;
  for (;; --l) {
    c->cfg.stack_lds = l;
    lds = pt_trace_lds_bytes(c->cfg, c->dsc);
    if (lds > 160u * 1024u) { if (l > 1 && !l_forced) continue; return fail(c, PTC_E_ARG, "configure_launch: staged tree top + stack exceed the 160 KiB of LDS"); }
    per_cu = trace_blocks_per_cu_cached(lds);     // registers, static LDS and launch bounds included
    if (per_cu >= 8 || l <= 2 || l_forced) break;
  }
  if (per_cu < 1) return fail(c, PTC_E_DEVICE, "configure_launch: the trace kernels do not fit a CU with this LDS size");
  if (const char* e = std::getenv("PTC_TRACE_BLOCKS_PER_CU")) { int v = std::atoi(e); if (v >= 1 && v <= per_cu) per_cu = v; }
  c->cfg.trace_blocks_per_cu = per_cu;
  const uint32_t ovf = (uint32_t)(need - l > 0 ? need - l : 1);
  const size_t total_waves = (size_t)c->cfg.n_cu * (size_t)per_cu * (size_t)(pt_trace_block_threads() / 64);   // the persistent grid
  c->dsc.ovf_depth = ovf;
  for (auto& ln : c->lanes) {                                    // concurrent kernels must not share a slab
    uint2* pl = nullptr;
    int rc = dev_alloc(c, c->scene_allocs, &pl, total_waves * ovf * 64);
    if (rc) return rc;
    ln.stack_ovf = pl;
    ln.stack_ovf2 = nullptr;
    if (c->trace_overlap) {
      if ((rc = dev_alloc(c, c->scene_allocs, &pl, total_waves * ovf * 64))) return rc;
      ln.stack_ovf2 = pl;
      if (!ln.stream2 && hipStreamCreateWithFlags(&ln.stream2, hipStreamNonBlocking) != hipSuccess) return fail(c, PTC_E_DEVICE, "configure_launch: hipStreamCreate failed");
    }
  }
  return PTC_OK;
}

DevScene lane_scene(ptc_ctx* c, int l) { DevScene d = c->dsc; d.stack_ovf = c->lanes[(size_t)l].stack_ovf; return d; }
// the lane's queues with the segment layout of a batch of n slots (ptc_internal.h, "SEGMENTED queues")
DevQueues batch_queues(ptc_ctx* c, int l, uint32_t n) {
  DevQueues q = c->lanes[(size_t)l].q;
  ptc_seg_layout(n, (uint32_t)c->cfg.shade_waves, q.n_seg, q.seg_len);
  return q;
}
bool is_raster(int integrator) { return integrator == PTC_INTEGRATOR_RASTER_COMPAT || integrator == PTC_INTEGRATOR_RASTER_GBUFFER16; }

// One wavefront batch of n samples per owned pixel on lane `l`, fully asynchronous.
int run_batch(ptc_ctx* c, int l, uint32_t first_sample, uint32_t n_samples) {
  const uint32_t n_paths = c->fr.n_owned * n_samples;
  Lane& ln = c->lanes[(size_t)l];
  hipStream_t st = ln.stream;
  const DevQueues q = batch_queues(c, l, n_paths);
  const DevScene sc = lane_scene(c, l);
  // The persistent grid of the trace kernels follows the batch: a wave wants several refills' worth of rays (c->trace_rays_per_lane per lane) to run in its
  // steady state; 8192 waves over the 2 M rays of a 1080p x 1 spp frame are 4 refills each, most of the launch is start-up and drain (0.8 ms for bounce 0,
  // 0.25 ms for the last bounces: tools/viewer_loop.py).  Batches of the benchmark's size keep the full grid.
  LaunchCfg cfg = c->cfg;
  {
    const uint64_t per_block = (uint64_t)pt_trace_block_threads() * (uint64_t)c->trace_rays_per_lane;
    uint64_t per_cu = ((uint64_t)n_paths + per_block * (uint64_t)cfg.n_cu - 1u) / (per_block * (uint64_t)cfg.n_cu);
    if (per_cu < 1) per_cu = 1;
    if (per_cu < (uint64_t)cfg.trace_blocks_per_cu) cfg.trace_blocks_per_cu = (int)per_cu;
  }
  if (c->spans.size() > kMaxSpans) {      // bounded event pool: harvest what has completed; if the host runs far ahead of
    collect_times(c, false);              // the device, wait for the oldest batch (back-pressure) instead of growing
    if (c->spans.size() > kMaxSpans) { (void)hipEventSynchronize(c->spans[c->spans.size() - kMaxSpans].b); collect_times(c, false); }
  }
  ScopedSpan whole(c, st, 3);
  pt_launch_set_counts(st, cfg, q, n_paths, 0);
  if (is_raster(c->integrator)) {
    pt_launch_raygen(st, c->cam, c->fr, q, 0, 1, true);
    { ScopedSpan t(c, st, 0); pt_launch_trace_closest(st, cfg, sc, q, 0, true); c->stats.launches_trace_closest++; }
    pt_launch_shade_raster(st, sc, c->cam, c->fr, q, c->accum.p, c->integrator == PTC_INTEGRATOR_RASTER_GBUFFER16);
  } else {
    pt_launch_raygen(st, c->cam, c->fr, q, first_sample, n_samples, false);
    const bool shadows = sc.n_lights > 0 || sc.env_ok;
    const bool small_batch = n_paths <= (1u << 26);
    const bool overlap = (c->trace_overlap == 2 || (c->trace_overlap == 1 && small_batch)) && shadows && ln.stream2 && ln.stack_ovf2;
    if (overlap) {
      const size_t need = (size_t)c->fr.max_bounces + 1;
      while (ln.ev_scan.size() < need) { hipEvent_t e = nullptr; HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); ln.ev_scan.push_back(e); }
      while (ln.ev_any.size() < need) { hipEvent_t e = nullptr; HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming)); ln.ev_any.push_back(e); }
    }
    DevScene sc_any = sc;
    if (overlap) sc_any.stack_ovf = ln.stack_ovf2;
    const bool per_kernel = !overlap || c->timing >= 2;       // overlapped launches: the batch's span only (ptc_stats.seconds_render); seconds_trace_* / seconds_shade stay 0
    for (int b = 0; b <= c->fr.max_bounces; ++b) {
      { ScopedSpan t(c, st, 0, per_kernel); pt_launch_trace_closest(st, cfg, sc, q, b & 1, false); c->stats.launches_trace_closest++; }
      // k_shade(b) overwrites the shadow queue any(b - 1) reads and adds to the path radiance it adds to
      if (overlap && b > 0) HIP_TRY(c, hipStreamWaitEvent(st, ln.ev_any[(size_t)b - 1], 0));
      { ScopedSpan t(c, st, 2, per_kernel); pt_launch_shade(st, cfg, ln.d_scene, c->fr, q, b & 1, (uint32_t)b); }
      if (b == c->fr.max_bounces) break;                         // the last bounce's shade produces no rays
      pt_launch_scan(st, cfg, q, (b + 1) & 1);
      if (overlap) {      // any(b) on the second stream, beside closest(b + 1)
        HIP_TRY(c, hipEventRecord(ln.ev_scan[(size_t)b], st));
        HIP_TRY(c, hipStreamWaitEvent(ln.stream2, ln.ev_scan[(size_t)b], 0));
        { ScopedSpan t(c, ln.stream2, 1, per_kernel); pt_launch_trace_any(ln.stream2, cfg, sc_any, q, nullptr); c->stats.launches_trace_any++; }
        HIP_TRY(c, hipEventRecord(ln.ev_any[(size_t)b], ln.stream2));
      } else if (shadows) {
        ScopedSpan t(c, st, 1); pt_launch_trace_any(st, cfg, sc, q, nullptr); c->stats.launches_trace_any++;
      }
    }
    // sample-order accumulation: wait for the previous batch's accumulate (it ran on the previous lane)
    if (c->n_lanes > 1 && c->batches_issued > 0) {
      const int prev = (int)((c->batches_issued - 1) % (uint64_t)c->n_lanes);
      if (prev != l) HIP_TRY(c, hipStreamWaitEvent(st, c->lanes[(size_t)prev].acc_done, 0));
    }
    pt_launch_accumulate(st, c->fr, q, c->accum.p, n_samples);
    if (c->n_lanes > 1) HIP_TRY(c, hipEventRecord(ln.acc_done, st));
  }
  c->batches_issued++;
  HIP_TRY(c, hipGetLastError());
  return PTC_OK;
}

// Issue `k` samples (k <= per_batch) as one batch on the next lane.
int issue(ptc_ctx* c, uint32_t k) {
  if (is_raster(c->integrator)) {
    if (c->samples_done == 0) {
      int rc = ensure_lane_queues(c, c->fr.n_owned);
      if (rc) return rc;
      if ((rc = run_batch(c, 0, 0, 1))) return rc;
      c->stats.paths = c->fr.n_owned;
    }
    c->samples_done += k;
    return PTC_OK;
  }
  const uint64_t cap = (uint64_t)c->fr.n_owned * k;
  int rc = ensure_lane_queues(c, (uint32_t)cap);
  if (rc) return rc;
  if ((rc = run_batch(c, (int)(c->batches_issued % (uint64_t)c->n_lanes), c->sample_base + c->samples_done, k))) return rc;
  c->samples_done += k;
  c->stats.paths += cap;
  return PTC_OK;
}
// Issue everything frame_add_samples has accepted so far.
int flush(ptc_ctx* c) {
  if (!c->in_frame) return PTC_OK;
  while (c->pending) {
    const uint32_t k = c->pending < c->per_batch ? c->pending : c->per_batch;
    c->pending -= k;
    if (c->fr.n_owned == 0) { c->samples_done += k; continue; }
    int rc = issue(c, k);
    if (rc) return rc;
  }
  return PTC_OK;
}

// stream 0 waits for the accumulates of all lanes (the last batches may have run elsewhere)
int join_lanes_on_stream0(ptc_ctx* c) {
  if (c->n_lanes > 1 && !is_raster(c->integrator))
    for (int l = 1; l < c->n_lanes; ++l)
      if ((uint64_t)l < c->batches_issued) HIP_TRY(c, hipStreamWaitEvent(c->lanes[0].stream, c->lanes[(size_t)l].acc_done, 0));
  return PTC_OK;
}

int need_device(ptc_ctx* c) {
  if (!c) return PTC_E_ARG;
  if (c->device < 0) return fail(c, PTC_E_DEVICE, kNoDevice);
  HIP_TRY(c, hipSetDevice(c->device));
  return PTC_OK;
}

int sum_lane_stats(ptc_ctx* c, unsigned long long st[ST_N]) {
  for (int i = 0; i < ST_N; ++i) st[i] = 0;
  for (auto& ln : c->lanes) {
    if (!ln.q.stats) continue;
    unsigned long long one[ST_N * ST_STRIDE];
    HIP_TRY(c, hipMemcpy(one, ln.q.stats, sizeof one, hipMemcpyDeviceToHost));
    for (int i = 0; i < ST_N; ++i) st[i] += one[i * ST_STRIDE];
  }
  return PTC_OK;
}

// RGBA16F view of the radiance buffer, converted on stream 0 (after everything queued there: resolve, reduce, write).
int convert_half(ptc_ctx* c) {
  if (!c->radiance.p || c->rad_w == 0) return fail(c, PTC_E_STATE, "radiance_rgba16f: nothing rendered");
  const size_t n = (size_t)c->rad_w * c->rad_h;
  int rc = ensure_buf(c, c->half, n);
  if (rc) return rc;
  pt_launch_to_half(c->lanes[0].stream, c->radiance.p, c->half.p, (uint32_t)n);
  HIP_TRY(c, hipGetLastError());
  return PTC_OK;
}

int debug_prepare(ptc_ctx* c, uint32_t n, const char* who) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->committed) return fail(c, PTC_E_STATE, std::string(who) + ": scene not committed");
  { int rf = flush(c); if (rf) return rf; }
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  c->in_frame = false; c->pending = 0;
  int rc = ensure_lane_queues(c, n);
  if (rc) return rc;
  for (auto& ln : c->lanes) HIP_TRY(c, hipMemset(ln.q.stats, 0, ST_N * ST_STRIDE * sizeof(unsigned long long)));
  return PTC_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int ptc_abi_version(void) { return PTC_ABI_VERSION; }

#ifndef PTC_KERNEL_SHA
#define PTC_KERNEL_SHA "unknown"
#endif
// How the library launches: the kernels' compile-time constants, then what the context (or, without one, a fresh context with an empty environment) uses.
// A roofline figure belongs to a launch policy as much as to a kernel source: tools/make_kernel_model.py records this string, bench.py compares.
const char* ptc_launch_policy(const ptc_ctx* c) {
  static thread_local std::string out;
  static const ptc_ctx defaults;                     // member initialisers = the built-in defaults (ptc_create then reads the environment)
  const ptc_ctx& x = c ? *c : defaults;
  char buf[512];
  std::snprintf(buf, sizeof buf, " | stack_lds_max=6 segments_per_cu_default=64 nodelets=%u lanes=%d batch_paths=%zu trace_overlap=%d(<=2^26 paths) rays_per_lane=%d shade_sort=%d refit=%s bvh=%s",
                x.toplet_budget, x.n_lanes, x.max_batch_paths, x.trace_overlap, x.trace_rays_per_lane, x.cfg.shade_sort, x.refit_on_device ? "device" : "host",
                x.bvh_builder == PTC_BVH_LBVH ? "lbvh" : "sah");
  out = std::string(pt_kernel_policy()) + buf;
  if (c && c->committed && c->device >= 0) {
    std::snprintf(buf, sizeof buf, " | trace_blocks_per_cu=%d stack_lds=%d lds_units=%u ovf_depth=%u shade_segments=%d shade_tables_lds=%d", c->cfg.trace_blocks_per_cu, c->cfg.stack_lds,
                  c->dsc.n_lds_units, c->dsc.ovf_depth, c->cfg.shade_waves, c->cfg.shade_tables_lds);
    out += buf;
  }
  return out.c_str();
}

const char* ptc_build_info(void) { return "ptc abi " PTC_STR(PTC_ABI_VERSION) " gfx950 kernels-sha256 " PTC_KERNEL_SHA; }

ptc_ctx* ptc_create(int device_id) {
  if (device_id == PTC_DEVICE_NONE) {   // description-only context: host flatten + BVH build, no rendering
    ptc_ctx* c = new ptc_ctx();
    c->device = PTC_DEVICE_NONE;
    if (const char* s = std::getenv("PTC_NODELETS")) c->toplet_budget = (uint32_t)std::strtoul(s, nullptr, 10);
    if (const char* s = std::getenv("PTC_BVH")) { if (std::strcmp(s, "lbvh") == 0) c->bvh_default = c->bvh_builder = PTC_BVH_LBVH; }
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
  c->cfg.n_cu = prop.multiProcessorCount;
  c->cfg.trace_blocks_per_cu = 4;
  c->cfg.stack_lds = 6;
  {   // segments of a queue = waves of k_shade's grid; a CU holds 16 of them at a time (4 waves per SIMD).  Whole multiples of 16 only: 8 / 12 / 20 / 24 leave a
      // partial round and cost 4-20 %.  With the round-3b kernel (global instead of FLAT gathers) 16 / 32 / 48 / 64 per CU give k_shade 0.1296 / 0.1296 / 0.1245 / 0.1236 s
      // per 6 steps on the atrium and 0.1768 / 0.1973 / 0.1902 / 0.1851 on the textured atrium (profiles/r03_shade_segments.txt): four short rounds beat two,
      // one round is best where every segment costs the same and worst where they do not.  64 = PTC_MAX_SEGMENTS / 256 CUs.
    int per_cu = 64;
    if (const char* s = std::getenv("PTC_SEGMENTS_PER_CU")) { int v = std::atoi(s); if (v >= 1 && v <= 64) per_cu = v; }
    uint32_t n = (uint32_t)(c->cfg.n_cu * per_cu);
    c->cfg.shade_waves = (int)(n > PTC_MAX_SEGMENTS ? PTC_MAX_SEGMENTS : n);
  }
  // P9's material sort is built and bit-exact either way, and OFF by default: k_shade runs at the rate of the CUs' memory path, so class-uniform
  // waves buy nothing, while the sort reads the hit words a second time and turns the ray loads into gathers: -11 % k_shade time without it on the
  // atrium, -5 % on the textured atrium (profiles/r03_shade_variants.txt).  Output compaction (ballot + mbcnt prefix) is always on.
  if (const char* s = std::getenv("PTC_SHADE_SORT")) c->cfg.shade_sort = std::atoi(s) != 0 ? 1 : 0;
  if (const char* s = std::getenv("PTC_TRACE_RAYS_PER_LANE")) { int v = std::atoi(s); if (v >= 1 && v <= 4096) c->trace_rays_per_lane = v; }
  if (const char* s = std::getenv("PTC_TRACE_OVERLAP")) { int v = std::atoi(s); if (v >= 0 && v <= 2) c->trace_overlap = v; }
  if (const char* s = std::getenv("PTC_NODELETS")) c->toplet_budget = (uint32_t)std::strtoul(s, nullptr, 10);
  if (const char* s = std::getenv("PTC_BATCH_PATHS")) { size_t v = std::strtoull(s, nullptr, 10); if (v >= 1024) c->max_batch_paths = v; }
  if (const char* s = std::getenv("PTC_TIMING")) { const int v = std::atoi(s); c->timing = v < 0 ? 0 : (v > 2 ? 2 : v); }
  if (const char* s = std::getenv("PTC_BVH")) { if (std::strcmp(s, "lbvh") == 0) c->bvh_default = c->bvh_builder = PTC_BVH_LBVH; }
  if (const char* s = std::getenv("PTC_LANES")) { int v = std::atoi(s); if (v >= 1 && v <= 8) c->n_lanes = v; }
  c->lanes.resize((size_t)c->n_lanes);
  bool ok = true;
  for (auto& ln : c->lanes) {
    ok = ok && hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&ln.acc_done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&ln.q.cnt, CNT_N * sizeof(uint32_t)) == hipSuccess && hipMalloc((void**)&ln.q.stats, ST_N * ST_STRIDE * sizeof(unsigned long long)) == hipSuccess;
    ok = ok && hipMemset(ln.q.cnt, 0, CNT_N * sizeof(uint32_t)) == hipSuccess && hipMemset(ln.q.stats, 0, ST_N * ST_STRIDE * sizeof(unsigned long long)) == hipSuccess;
    uint32_t* segs = nullptr;                                     // seg_ray[2], seg_sh, pre_ray, pre_sh: one allocation
    const size_t per = PTC_MAX_SEGMENTS + 64;
    ok = ok && hipMalloc((void**)&segs, 5 * per * sizeof(uint32_t)) == hipSuccess && hipMemset(segs, 0, 5 * per * sizeof(uint32_t)) == hipSuccess;
    if (ok) { ln.q.seg_ray[0] = segs; ln.q.seg_ray[1] = segs + per; ln.q.seg_sh = segs + 2 * per; ln.q.pre_ray = segs + 3 * per; ln.q.pre_sh = segs + 4 * per; }
    ok = ok && hipMalloc((void**)&ln.d_scene, sizeof(DevScene)) == hipSuccess;
  }
  if (!ok) { g_create_error = "ptc_create: could not create the lane streams / events / counters"; ptc_destroy(c); return nullptr; }
  return c;
}

void ptc_destroy(ptc_ctx* c) {
  if (!c) return;
  if (c->device < 0) { delete c; return; }
  (void)hipSetDevice(c->device);
  for (auto& ln : c->lanes) if (ln.stream) (void)hipStreamSynchronize(ln.stream);
  if (c->comm && c->comm_owned && g_rccl.so) (void)g_rccl.CommDestroy(c->comm);
  for (const Span& s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  for (hipEvent_t e : c->free_events) (void)hipEventDestroy(e);
  for (auto& ln : c->lanes) {
    if (ln.acc_done) (void)hipEventDestroy(ln.acc_done);
    free_all(ln.allocs);
    if (ln.q.cnt) (void)hipFree(ln.q.cnt);
    if (ln.q.stats) (void)hipFree(ln.q.stats);
    if (ln.q.seg_ray[0]) (void)hipFree(ln.q.seg_ray[0]);
    if (ln.d_scene) (void)hipFree(ln.d_scene);
    if (c->bscratch.p) { (void)hipFree(c->bscratch.p); c->bscratch = BuildScratch(); }
    free_rebuild_spare(c);
    if (ln.stream) (void)hipStreamDestroy(ln.stream);
    if (ln.stream2) (void)hipStreamDestroy(ln.stream2);
    for (hipEvent_t e : ln.ev_scan) (void)hipEventDestroy(e);
    for (hipEvent_t e : ln.ev_any) (void)hipEventDestroy(e);
  }
  free_all(c->scene_allocs);
  c->owned.release(); c->accum.release(); c->radiance.release(); c->ldr.release(); c->half.release();
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
  c->bvh_builder = c->bvh_default;
  c->have_cam = false; c->committed = false; c->in_frame = false; c->pending = 0;
  free_all(c->scene_allocs);
  for (auto& ln : c->lanes) ln.stack_ovf = nullptr;
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

namespace {
int commit_upload(ptc_ctx* c, std::chrono::steady_clock::time_point t0, bool skeleton = false);
// Device half of a refit: c->built holds the refitted arrays.  same_sizes: overwrite in place what depends on the vertex positions (textures,
// environment and materials stay where they are); else (an emitter appeared or vanished under a degenerate scale) upload everything.
int refit_upload(ptc_ctx* c, bool same_sizes, std::chrono::steady_clock::time_point t0) {
  const HostBuilt& B = *c->built;
  c->host_stale = false; c->last_refit_on_device = false;
  if (!same_sizes) return commit_upload(c, t0);
  HIP_TRY(c, hipMemcpy((void*)c->dsc.recs, B.recs.data(), B.recs.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy((void*)c->dsc.shade, B.shade.data(), B.shade.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy((void*)c->dsc.lights, B.lights.data(), B.lights.size() * 4, hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy((void*)c->dsc.cdf, B.cdf.data(), B.cdf.size() * 4, hipMemcpyHostToDevice));
  c->dsc.ray_eps = B.ray_eps; c->dsc.n_lights = B.n_lights;
  for (int k = 0; k < 3; ++k) { c->dsc.grid_lo[k] = B.grid_lo[k]; c->dsc.grid_step[k] = B.grid_step[k]; }
  for (int l = 0; l < c->n_lanes; ++l) {
    const DevScene ds = lane_scene(c, l);
    HIP_TRY(c, hipMemcpy(c->lanes[(size_t)l].d_scene, &ds, sizeof ds, hipMemcpyHostToDevice));
  }
  return PTC_OK;
}

bool refit_on_device(ptc_ctx* c) {
  if (const char* s = std::getenv("PTC_REFIT")) return std::strcmp(s, "host") != 0;
  return c->refit_on_device != 0;
}

// The plan of the committed scene in HBM + the scratch arrays of the refit kernels; once per commit.
int ensure_refit_plan(ptc_ctx* c) {
  if (c->refit_ready) return PTC_OK;
  const HostBuilt& B = *c->built;
  ptc_refit_plan(c->mats, c->meshes, c->insts, B, c->plan);
  const RefitPlan& P = c->plan;
  DevRefit d{};
  int rc = dev_upload(c, c->scene_allocs, &d.mesh_verts, P.mesh_verts);
  if (!rc) rc = dev_upload(c, c->scene_allocs, &d.vert_inst, P.vert_inst);
  if (!rc) rc = dev_upload(c, c->scene_allocs, &d.inst_first, P.inst_first);
  if (!rc) rc = dev_upload(c, c->scene_allocs, &d.inst_src, P.inst_src);
  if (!rc) rc = dev_upload(c, c->scene_allocs, &d.widx, B.widx);
  if (!rc) rc = dev_upload(c, c->scene_allocs, &d.level_nodes, P.level_nodes);
  if (!rc) rc = dev_alloc(c, c->scene_allocs, &d.inst_xf, c->insts.size() * 21);
  if (!rc) rc = dev_alloc(c, c->scene_allocs, &d.wverts, (size_t)P.n_verts);
  if (!rc) rc = dev_alloc(c, c->scene_allocs, &d.wbt, (size_t)P.n_verts * 3);
  if (!rc) rc = dev_alloc(c, c->scene_allocs, &d.nbox, (size_t)(B.n_units / 4u + 1u) * 6);
  if (!rc) rc = dev_alloc(c, c->scene_allocs, &d.bounds, 8);
  if (!rc) rc = dev_alloc(c, c->scene_allocs, &d.cost, 1);
  if (!rc) { std::vector<uint32_t> cls; ptc_prim_classes(c->mats, B.tri_mat, cls); rc = dev_upload(c, c->scene_allocs, &d.prim_cls, cls); }
  if (rc) return rc;
  d.recs = const_cast<float4*>(c->dsc.recs); d.shade = const_cast<float4*>(c->dsc.shade);
  d.n_verts = P.n_verts; d.n_tris = P.n_tris; d.shade_stride = B.shade_stride;
  c->drf = d;
  c->refit_ready = true;
  return PTC_OK;
}

// A refit moves the committed scene: ptc_add_mesh / ptc_add_instance* are accepted after a commit (they describe the NEXT commit), and a refit of a
// description that has grown since would index the committed arrays out of bounds — on the device without anybody noticing.  Same test, same
// error as the host path (build_or_refit), made before anything is uploaded or launched.
bool description_matches_commit(const ptc_ctx* c) {
  if (c->insts.size() != c->committed_insts) return false;
  uint64_t nv = 0, nt = 0;
  for (const HostInstance& in : c->insts) {
    if (in.mesh < 0 || (size_t)in.mesh >= c->meshes.size()) return false;
    nv += c->meshes[(size_t)in.mesh].v.size(); nt += c->meshes[(size_t)in.mesh].idx.size() / 3;
  }
  return nv == c->built->n_wverts && nt == c->built->n_tris;
}
const char* const kDescriptionChanged = "scene_refit: the scene's meshes or instances changed since the commit (only transforms may)";

float scene_half_area(const float lo[3], const float hi[3]) {      // the host's box_half_area of the scene box
  const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
  return ex * ey + ey * ez + ez * ex;
}

// Refit on the device.  Returns PTC_OK, an error, or +1: "not this way" (the set of emitters changed) — the caller refits on the host.
// may_write_built: c->built is this context's own, or a group's fresh copy every member writes the same values to.
int device_refit(ptc_ctx* c) {
  { int rc = ensure_refit_plan(c); if (rc) return rc; }
  std::vector<float> xf, lights, cdf;
  if (!ptc_refit_instance_transforms(c->insts, xf)) return fail(c, PTC_E_STATE, "scene_commit: non-finite vertex position after the instance transform");
  if (!ptc_refit_emitters(c->mats, c->meshes, c->insts, c->plan, *c->built, lights, cdf)) return 1;
  hipStream_t st = c->lanes[0].stream;
  const DevRefit& d = c->drf;
  HIP_TRY(c, hipMemcpyAsync(d.inst_xf, xf.data(), xf.size() * 4, hipMemcpyHostToDevice, st));
  pt_launch_refit_geometry(st, d);
  uint32_t raw[8];
  HIP_TRY(c, hipMemcpyAsync(raw, d.bounds, sizeof raw, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  float lo[3], hi[3]; bool bad = false;
  pt_refit_decode_bounds(raw, lo, hi, &bad);
  if (bad) {     // nothing of the scene was written (k_refit_prims saw the flag); the scratch vertices go back to the state the scene in HBM was made from
    if (!c->xf_live.empty()) {
      HIP_TRY(c, hipMemcpyAsync(d.inst_xf, c->xf_live.data(), c->xf_live.size() * 4, hipMemcpyHostToDevice, st));
      pt_launch_refit_geometry(st, d);
      HIP_TRY(c, hipStreamSynchronize(st));
    }
    return fail(c, PTC_E_STATE, "scene_commit: non-finite vertex position after the instance transform");
  }
  HostBuilt& B = *c->built;
  ptc_refit_grid(lo, hi, B.grid_lo, B.grid_step, &B.ray_eps);
  pt_launch_refit_nodes(st, d, c->plan.level_first, B.grid_lo, B.grid_step, B.sa_unit);      // the cost in the unit of the build: comparable with bvh_sa_cost_built
  HIP_TRY(c, hipGetLastError());
  unsigned long long cost_fixed = 0;
  HIP_TRY(c, hipMemcpyAsync(&cost_fixed, d.cost, sizeof cost_fixed, hipMemcpyDeviceToHost, st));
  B.lights = lights; B.cdf = cdf;
  HIP_TRY(c, hipMemcpyAsync((void*)c->dsc.lights, B.lights.data(), B.lights.size() * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync((void*)c->dsc.cdf, B.cdf.data(), B.cdf.size() * 4, hipMemcpyHostToDevice, st));
  c->dsc.ray_eps = B.ray_eps;
  for (int k = 0; k < 3; ++k) { c->dsc.grid_lo[k] = B.grid_lo[k]; c->dsc.grid_step[k] = B.grid_step[k]; }
  for (int l = 0; l < c->n_lanes; ++l) {
    const DevScene ds = lane_scene(c, l);
    HIP_TRY(c, hipMemcpyAsync(c->lanes[(size_t)l].d_scene, &ds, sizeof ds, hipMemcpyHostToDevice, st));
  }
  HIP_TRY(c, hipStreamSynchronize(st));
  B.sa_cost_fixed = cost_fixed;
  c->stats.bvh_sa_cost = (double)cost_fixed / (double)PTC_SA_COST_ONE;
  c->host_stale = true; c->last_refit_on_device = true;
  c->xf_live.swap(xf);
  return PTC_OK;
}

// scene_allocs bookkeeping of ptc_scene_rebuild: an array of the committed scene is replaced
void scene_free(ptc_ctx* c, const void* p) {
  if (!p) return;
  auto it = std::find(c->scene_allocs.begin(), c->scene_allocs.end(), const_cast<void*>(p));
  if (it != c->scene_allocs.end()) { (void)hipFree(*it); c->scene_allocs.erase(it); }
}

// ptc_scene_rebuild on the device: a refit's geometry pass, then a NEW tree for the vertices as they now lie in HBM (pt_build.hip), then the refit's node pass over
// it.  Returns PTC_OK, an error, or +1: "not this way" (the set of emitters changed, fewer than two triangles): the caller builds on the host.
// fresh: the device half of a COMMIT on the device (device_commit): there is no tree yet, the launch configuration follows the tree and is the caller's.
int device_rebuild(ptc_ctx* c, bool fresh = false) {
  { int rc = ensure_refit_plan(c); if (rc) return rc; }
  if (c->built->n_tris < 2u) return 1;
  std::vector<float> xf, lights, cdf;
  if (!ptc_refit_instance_transforms(c->insts, xf)) return fail(c, PTC_E_STATE, "scene_commit: non-finite vertex position after the instance transform");
  if (!ptc_refit_emitters(c->mats, c->meshes, c->insts, c->plan, *c->built, lights, cdf)) return 1;
  hipStream_t st = c->lanes[0].stream;
  DevRefit& d = c->drf;
  HIP_TRY(c, hipMemcpyAsync(d.inst_xf, xf.data(), xf.size() * 4, hipMemcpyHostToDevice, st));
  pt_launch_refit_geometry(st, d);
  uint32_t raw[8];
  HIP_TRY(c, hipMemcpyAsync(raw, d.bounds, sizeof raw, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  float lo[3], hi[3]; bool bad = false;
  pt_refit_decode_bounds(raw, lo, hi, &bad);
  if (bad) {
    if (!c->xf_live.empty()) {
      HIP_TRY(c, hipMemcpyAsync(d.inst_xf, c->xf_live.data(), c->xf_live.size() * 4, hipMemcpyHostToDevice, st));
      pt_launch_refit_geometry(st, d);
      HIP_TRY(c, hipStreamSynchronize(st));
    }
    return fail(c, PTC_E_STATE, "scene_commit: non-finite vertex position after the instance transform");
  }
  if (!c->rb_live.recs) {       // first rebuild since the commit: the arrays in use are the commit's, exactly as large as their content
    c->rb_live.recs = const_cast<float4*>(c->dsc.recs); c->rb_live.recs_cap = c->built->n_units;
    c->rb_live.levels = const_cast<uint32_t*>(d.level_nodes); c->rb_live.levels_cap = c->built->n_nodes;
    c->rb_live.nbox = d.nbox; c->rb_live.nbox_cap = (size_t)(c->built->n_units / 4u + 1u) * 6;
  }
  BuildOut out;
  out.recs = c->rb_spare.recs; out.recs_cap = c->rb_spare.recs_cap; out.level_nodes = c->rb_spare.levels; out.level_cap = c->rb_spare.levels_cap;
  c->rb_spare.recs = nullptr; c->rb_spare.levels = nullptr; c->rb_spare.recs_cap = c->rb_spare.levels_cap = 0;      // the build owns them now (it may free them)
  const std::string e = pt_build_lbvh(st, d.wverts, d.widx, d.prim_cls, d.n_tris, c->toplet_budget, c->bscratch, out);
  if (!e.empty()) { if (out.recs) (void)hipFree(out.recs); if (out.level_nodes) (void)hipFree(out.level_nodes); return fail(c, PTC_E_DEVICE, e); }
  // the new tree replaces the old one: unit array, the refit's level lists, the per-record boxes; the replaced arrays are the next rebuild's spare set
  const size_t nbox_need = (size_t)(out.n_units / 4u + 1u) * 6;
  float* nbox = c->rb_spare.nbox; size_t nbox_cap = c->rb_spare.nbox_cap;
  c->rb_spare.nbox = nullptr; c->rb_spare.nbox_cap = 0;
  if (!nbox || nbox_cap < nbox_need) {
    if (nbox) (void)hipFree(nbox);
    nbox_cap = nbox_need + nbox_need / 8u;
    if (hipMalloc((void**)&nbox, nbox_cap * sizeof(float)) != hipSuccess) { (void)hipFree(out.recs); (void)hipFree(out.level_nodes); return fail(c, PTC_E_NOMEM, "scene_rebuild: out of device memory"); }
  }
  for (void* old : {(void*)c->rb_live.recs, (void*)c->rb_live.levels, (void*)c->rb_live.nbox}) {       // out of the scene's list, not freed
    auto it = std::find(c->scene_allocs.begin(), c->scene_allocs.end(), old);
    if (it != c->scene_allocs.end()) c->scene_allocs.erase(it);
  }
  c->rb_spare = c->rb_live;
  c->rb_live.recs = out.recs; c->rb_live.recs_cap = out.recs_cap; c->rb_live.levels = out.level_nodes; c->rb_live.levels_cap = out.level_cap; c->rb_live.nbox = nbox; c->rb_live.nbox_cap = nbox_cap;
  c->scene_allocs.push_back(out.recs); c->scene_allocs.push_back(out.level_nodes); c->scene_allocs.push_back(nbox);
  d.recs = out.recs; d.level_nodes = out.level_nodes; d.nbox = nbox;
  c->plan.level_first = out.level_first; c->plan.level_nodes.clear();
  HostBuilt& B = *c->built;
  ptc_refit_grid(lo, hi, B.grid_lo, B.grid_step, &B.ray_eps);
  B.sa_unit = scene_half_area(lo, hi);                      // a new topology: a new unit of its cost
  pt_launch_refit_nodes(st, d, c->plan.level_first, B.grid_lo, B.grid_step, B.sa_unit);
  HIP_TRY(c, hipGetLastError());
  unsigned long long cost_fixed = 0;
  HIP_TRY(c, hipMemcpyAsync(&cost_fixed, d.cost, sizeof cost_fixed, hipMemcpyDeviceToHost, st));
  B.lights = lights; B.cdf = cdf;
  HIP_TRY(c, hipMemcpyAsync((void*)c->dsc.lights, B.lights.data(), B.lights.size() * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync((void*)c->dsc.cdf, B.cdf.data(), B.cdf.size() * 4, hipMemcpyHostToDevice, st));
  // the host's picture of the build: sizes follow the new tree, the arrays come back from HBM when somebody asks (refresh_host_copy), and the topology the host
  // refit needs is gone — the next host-path refit builds from scratch
  B.n_nodes = out.n_nodes; B.n_units = out.n_units; B.max_depth = out.max_depth; B.n_tri_records = out.n_tri_records;
  B.n_lds_units = B.n_units < c->toplet_budget * 4u ? B.n_units : c->toplet_budget * 4u;
  B.recs.clear();                       // refresh_host_copy sizes and fills them when somebody asks
  B.topology.reset();
  c->dsc.recs = out.recs; c->dsc.n_lds_units = B.n_lds_units; c->dsc.ray_eps = B.ray_eps;
  for (int k = 0; k < 3; ++k) { c->dsc.grid_lo[k] = B.grid_lo[k]; c->dsc.grid_step[k] = B.grid_step[k]; }
  if (!fresh) {   // a deeper tree needs a deeper overflow slab behind the stack entries kept in LDS
    const int need = (int)B.max_depth + 2;
    const uint32_t ovf = (uint32_t)(need - c->cfg.stack_lds > 0 ? need - c->cfg.stack_lds : 1);
    if (ovf > c->dsc.ovf_depth) {
      const size_t total_waves = (size_t)c->cfg.n_cu * (size_t)c->cfg.trace_blocks_per_cu * (size_t)(pt_trace_block_threads() / 64);
      for (auto& ln : c->lanes) {
        uint2* pl = nullptr;
        scene_free(c, ln.stack_ovf); ln.stack_ovf = nullptr;
        int rc = dev_alloc(c, c->scene_allocs, &pl, total_waves * ovf * 64);
        if (rc) return rc;
        ln.stack_ovf = pl;
        if (ln.stack_ovf2) {
          scene_free(c, ln.stack_ovf2); ln.stack_ovf2 = nullptr;
          if ((rc = dev_alloc(c, c->scene_allocs, &pl, total_waves * ovf * 64))) return rc;
          ln.stack_ovf2 = pl;
        }
      }
      c->dsc.ovf_depth = ovf;
    }
  }
  if (!fresh)
    for (int l = 0; l < c->n_lanes; ++l) {
      const DevScene ds = lane_scene(c, l);
      HIP_TRY(c, hipMemcpyAsync(c->lanes[(size_t)l].d_scene, &ds, sizeof ds, hipMemcpyHostToDevice, st));
    }
  HIP_TRY(c, hipStreamSynchronize(st));
  B.sa_cost_fixed = cost_fixed;
  c->stats.bvh_sa_cost = c->stats.bvh_sa_cost_built = (double)cost_fixed / (double)PTC_SA_COST_ONE;
  c->stats.n_bvh_nodes = B.n_nodes; c->stats.bvh_max_depth = B.max_depth;
  c->host_stale = true; c->last_refit_on_device = true;
  c->xf_live.swap(xf);
  return PTC_OK;
}

// The debug getters read the host build: after a refit on the device its vertex-dependent arrays come back from HBM first.
int refresh_host_copy(ptc_ctx* c) {
  if (!c->host_stale || c->device < 0) return PTC_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  HostBuilt& B = *c->built;
  B.recs.resize((size_t)B.n_units * 4);                                     // a tree or a commit made on the device left the host arrays unsized
  B.shade.resize((size_t)B.n_tris * B.shade_stride * 4);
  B.wverts.resize(B.n_wverts);
  HIP_TRY(c, hipMemcpy(B.recs.data(), c->dsc.recs, B.recs.size() * 4, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(B.shade.data(), c->dsc.shade, B.shade.size() * 4, hipMemcpyDeviceToHost));
  HIP_TRY(c, hipMemcpy(B.wverts.data(), c->drf.wverts, B.wverts.size() * sizeof(HostVertex), hipMemcpyDeviceToHost));
  c->host_stale = false;
  return PTC_OK;
}
// A full host build of the description as it stands + upload (what ptc_scene_commit does), keeping what a refit / rebuild keeps of the statistics.
int host_build_and_upload(ptc_ctx* c, std::chrono::steady_clock::time_point t0, bool as_refit) {
  auto built = std::make_shared<HostBuilt>();
  const std::string e = ptc_build_scene(c->mats, c->meshes, c->insts, c->texs, c->env, c->toplet_budget, as_refit ? c->bvh_builder : PTC_BVH_LBVH, *built);
  if (!e.empty()) return fail(c, PTC_E_STATE, e);
  const ptc_stats keep = c->stats;
  c->built = built;
  const int rc = commit_upload(c, t0);
  if (rc) return rc;
  c->last_refit_on_device = false;
  const double dt = c->stats.seconds_commit;
  c->stats.seconds_commit = keep.seconds_commit; c->stats.seconds_refit = keep.seconds_refit; c->stats.seconds_rebuild = keep.seconds_rebuild;
  (as_refit ? c->stats.seconds_refit : c->stats.seconds_rebuild) = dt;
  return PTC_OK;
}
}  // namespace

int ptc_update_instance_matrix(ptc_ctx* c, int instance, const float model[16]) {
  if (!c) return PTC_E_ARG;
  if (!model) return fail(c, PTC_E_ARG, "update_instance_matrix: null pointer");
  if (instance < 0 || instance >= (int)c->insts.size()) return fail(c, PTC_E_ARG, "update_instance: instance out of range");
  std::memcpy(c->insts[(size_t)instance].m, model, 64);
  return PTC_OK;
}

int ptc_update_instance(ptc_ctx* c, int instance, const float t[3], const float q_wxyz[4], const float s[3]) {
  if (!c) return PTC_E_ARG;
  if (!t || !q_wxyz || !s) return fail(c, PTC_E_ARG, "update_instance: null pointer");
  if (instance < 0 || instance >= (int)c->insts.size()) return fail(c, PTC_E_ARG, "update_instance: instance out of range");
  ptc_trs_to_matrix(t, q_wxyz, s, c->insts[(size_t)instance].m);
  return PTC_OK;
}

int ptc_scene_refit(ptc_ctx* c) {
  if (!c) return PTC_E_ARG;
  if (!c->committed) return fail(c, PTC_E_STATE, "scene_refit: scene not committed");
  if (!description_matches_commit(c)) return fail(c, PTC_E_STATE, kDescriptionChanged);
  if (c->device >= 0) {
    HIP_TRY(c, hipSetDevice(c->device));
    { int rf = flush(c); if (rf) return rf; }
    { int rs = sync_all_lanes(c); if (rs) return rs; }
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (c->built.use_count() > 1) c->built = std::make_shared<HostBuilt>(*c->built);      // a group shares one build: this context now gets its own
  if (c->device >= 0 && refit_on_device(c)) {
    const int rd = device_refit(c);
    if (rd <= 0) {
      if (rd == PTC_OK) { c->in_frame = false; c->pending = 0; c->stats.seconds_refit = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
      return rd;
    }
  }
  if (!c->built->topology) return host_build_and_upload(c, t0, /*as_refit=*/true);      // the tree in HBM was built on the device (ptc_scene_rebuild): the host has no topology to refit
  HostBuilt& B = *c->built;
  const size_t n_recs = B.recs.size(), n_shade = B.shade.size(), n_lights = B.lights.size(), n_cdf = B.cdf.size();
  const std::string e = ptc_refit_scene(c->mats, c->meshes, c->insts, c->texs, c->env, B);
  if (!e.empty()) return fail(c, PTC_E_STATE, e);
  c->in_frame = false; c->pending = 0;
  c->stats.n_emitters = B.n_lights;
  c->stats.bvh_sa_cost = (double)B.sa_cost_fixed / (double)PTC_SA_COST_ONE;
  if (c->device >= 0) {
    int rc = refit_upload(c, B.recs.size() == n_recs && B.shade.size() == n_shade && B.lights.size() == n_lights && B.cdf.size() == n_cdf, t0);
    if (rc) return rc;
  }
  c->stats.seconds_refit = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return PTC_OK;
}

int ptc_scene_rebuild(ptc_ctx* c) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->committed) return fail(c, PTC_E_STATE, "scene_rebuild: scene not committed");
  if (!description_matches_commit(c)) return fail(c, PTC_E_STATE, kDescriptionChanged);
  { int rf = flush(c); if (rf) return rf; }
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  const auto t0 = std::chrono::steady_clock::now();
  if (c->built.use_count() > 1) c->built = std::make_shared<HostBuilt>(*c->built);      // a group shares one build: this context now gets its own
  const char* how = std::getenv("PTC_REBUILD");
  int rd = (how && std::strcmp(how, "host") == 0) ? 1 : device_rebuild(c);
  if (rd < 0) return rd;
  if (rd > 0) { if ((rd = host_build_and_upload(c, t0, /*as_refit=*/false))) return rd; }      // PTC_REBUILD=host, an emitter appeared or vanished, a single triangle
  else c->stats.seconds_rebuild = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  c->in_frame = false; c->pending = 0;
  return PTC_OK;
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

int ptc_set_bvh_builder(ptc_ctx* c, int builder) {
  if (!c) return PTC_E_ARG;
  if (builder != PTC_BVH_SAH && builder != PTC_BVH_LBVH) return fail(c, PTC_E_ARG, "set_bvh_builder: unknown builder");
  c->bvh_builder = builder;
  return PTC_OK;
}

int ptc_set_env_latlong_rgb32f(ptc_ctx* c, const float* rgb, int w, int h) {
  if (!c) return PTC_E_ARG;
  if (!rgb) { c->env = HostEnv{}; return PTC_OK; }
  if (w <= 0 || h <= 0 || w > 65536 || h > 65536 || (uint64_t)w * (uint64_t)h > (1u << 28)) return fail(c, PTC_E_ARG, "set_env: bad size");
  c->env.rgb.assign(rgb, rgb + (size_t)w * h * 3); c->env.w = w; c->env.h = h;
  return PTC_OK;
}

namespace {
int commit_finish(ptc_ctx* c, std::chrono::steady_clock::time_point t0);
// Device half of a commit: upload c->built, size the launches.  The caller has set c->built, the camera and seconds_commit's start.
// skeleton: c->built is ptc_build_skeleton's — the tables are uploaded, the shading records allocated and zeroed, there is no tree yet: device_commit goes on from here.
int commit_upload(ptc_ctx* c, std::chrono::steady_clock::time_point t0, bool skeleton) {
  ptc_make_camera(c->cam_pos, c->cam_target, c->cam_fov, c->cam_aspect, c->cam);
  c->in_frame = false; c->pending = 0;
  c->committed_insts = c->insts.size();
  if (c->device < 0) {   // description-only context: nothing to upload
    c->committed = true;
    std::memset(&c->stats, 0, sizeof c->stats);
    c->stats.seconds_commit = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->stats.n_triangles = c->built->n_tris; c->stats.n_bvh_nodes = c->built->n_nodes; c->stats.n_emitters = c->built->n_lights;
    c->stats.bvh_max_depth = c->built->max_depth;
    c->stats.bvh_sa_cost = c->stats.bvh_sa_cost_built = (double)c->built->sa_cost_fixed / (double)PTC_SA_COST_ONE;
    return PTC_OK;
  }
  c->committed = false; c->commit_on_device = false;
  free_all(c->scene_allocs);
  free_rebuild_spare(c);
  c->refit_ready = false; c->host_stale = false; c->plan = RefitPlan(); c->drf = DevRefit{}; c->xf_live.clear();
  for (auto& ln : c->lanes) ln.stack_ovf = nullptr;
  const HostBuilt& B = *c->built;
  DevScene d{};
  int rc = 0;
  {
    const float* p = nullptr;
    auto up = [&](const std::vector<float>& v, const float4** out) { if (!rc) { rc = dev_upload(c, c->scene_allocs, &p, v); *out = (const float4*)p; } };
    if (!skeleton) up(B.recs, &d.recs);
    up(B.mats, &d.mats); up(B.lights, &d.lights);
    if (!rc) rc = dev_upload(c, c->scene_allocs, &d.cdf, B.cdf);
    if (!skeleton) up(B.shade, &d.shade);
    else if (!rc) {
      float4* sh = nullptr;
      const size_t units = (size_t)B.n_tris * B.shade_stride;
      rc = dev_alloc(c, c->scene_allocs, &sh, units);
      if (!rc && hipMemsetAsync(sh, 0, units * sizeof(float4), c->lanes[0].stream) != hipSuccess) rc = fail(c, PTC_E_DEVICE, "scene_commit: hipMemset failed");
      d.shade = sh;
    }
    if (!rc) rc = dev_upload(c, c->scene_allocs, &d.texels, B.texels);
    if (!rc) { const int32_t* ti = nullptr; rc = dev_upload(c, c->scene_allocs, &ti, B.tex_info); d.tex_info = (const int4*)ti; }
    if (!rc) { const uint32_t* st = nullptr; rc = dev_upload(c, c->scene_allocs, &st, B.set_texels); d.set_texels = (const uint4*)st; }
    if (!rc) { const int32_t* si = nullptr; rc = dev_upload(c, c->scene_allocs, &si, B.set_info); d.set_info = (const int4*)si; }
    if (!rc) rc = dev_upload(c, c->scene_allocs, &d.env_marg_guide, B.env_marg_guide);
    if (!rc) rc = dev_upload(c, c->scene_allocs, &d.env_cond_guide, B.env_cond_guide);
    up(B.env, &d.env);
    if (!rc) rc = dev_upload(c, c->scene_allocs, &d.env_marg, B.env_marg);
    if (!rc) rc = dev_upload(c, c->scene_allocs, &d.env_cond, B.env_cond);
  }
  if (rc) { free_all(c->scene_allocs); return rc; }
  d.env_w = B.env_w; d.env_h = B.env_h; d.env_ok = B.env_ok;
  d.tex_linear = c->tex_linear;
  d.shade_stride = B.shade_stride;
  d.n_lights = B.n_lights; d.n_mats = (uint32_t)c->mats.size(); d.n_lds_units = B.n_lds_units; d.ray_eps = B.ray_eps;
  for (int k = 0; k < 3; ++k) { d.grid_lo[k] = B.grid_lo[k]; d.grid_step[k] = B.grid_step[k]; }
  c->dsc = d;
  if (skeleton) return PTC_OK;
  return commit_finish(c, t0);
}
// the launches follow the tree (its depth, the staged top): configuration, the lanes' copies of the scene, the statistics of a commit
int commit_finish(ptc_ctx* c, std::chrono::steady_clock::time_point t0) {
  const HostBuilt& B = *c->built;
  const bool timing = std::getenv("PTC_BUILD_TIMING") != nullptr;
  const auto tc0 = std::chrono::steady_clock::now();
  { int rc2 = configure_launch(c); if (rc2) { free_all(c->scene_allocs); return rc2; } }
  if (timing) std::fprintf(stderr, "    configure_launch            %7.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
  c->cfg.shade_tables_lds = pt_shade_tables_fit(c->dsc) ? 1 : 0;
  for (int l = 0; l < c->n_lanes; ++l) {
    const DevScene ds = lane_scene(c, l);
    HIP_TRY(c, hipMemcpy(c->lanes[(size_t)l].d_scene, &ds, sizeof ds, hipMemcpyHostToDevice));
  }
  c->committed = true;
  std::memset(&c->stats, 0, sizeof c->stats);
  c->stats.seconds_commit = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  c->stats.n_triangles = B.n_tris; c->stats.n_bvh_nodes = B.n_nodes; c->stats.n_emitters = B.n_lights; c->stats.bvh_max_depth = B.max_depth;
  c->stats.bvh_sa_cost = c->stats.bvh_sa_cost_built = (double)B.sa_cost_fixed / (double)PTC_SA_COST_ONE;
  return PTC_OK;
}

// ptc_scene_commit with the LBVH builder on a device context: the host describes (ptc_build_skeleton: indices, materials, emitters, textures), the DEVICE flattens the
// vertices, writes the shading records and builds the tree (pt_refit.hip, pt_build.hip) — the arrays in HBM are byte for byte those of the host's LBVH commit
// (tests/test_gpu_parity.py).  Returns PTC_OK, an error, or +1: "not this way" (fewer than two triangles): the caller commits on the host.
int device_commit(ptc_ctx* c, std::chrono::steady_clock::time_point t0) {
  const bool timing = std::getenv("PTC_BUILD_TIMING") != nullptr;      // phase times on stderr, as the host build prints them
  auto tprev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "  %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tprev).count());
    tprev = now;
  };
  auto built = std::make_shared<HostBuilt>();
  const std::string e = ptc_build_skeleton(c->mats, c->meshes, c->insts, c->texs, c->env, c->toplet_budget, *built);
  if (!e.empty()) return fail(c, PTC_E_STATE, e);
  if (built->n_tris < 2u) return 1;
  c->built = built;
  lap("describe (skeleton)");
  { int rc = commit_upload(c, t0, /*skeleton=*/true); if (rc) return rc; }
  lap("free + tables upload");
  { int rc = ensure_refit_plan(c); if (rc) { free_all(c->scene_allocs); return rc; } }
  lap("plan + its upload");
  {
    const int32_t *d_mat = nullptr, *d_light = nullptr;
    int rc = dev_upload(c, c->scene_allocs, &d_mat, built->tri_mat);
    if (!rc) rc = dev_upload(c, c->scene_allocs, &d_light, built->prim_light);
    if (rc) { free_all(c->scene_allocs); return rc; }
    pt_launch_refit_seed(c->lanes[0].stream, c->drf, d_mat, d_light);
  }
  const int rd = device_rebuild(c, /*fresh=*/true);
  if (rd) { free_all(c->scene_allocs); c->refit_ready = false; return rd; }
  lap("flatten + build on the device");
  const int rf = commit_finish(c, t0);
  lap("launch configuration");
  c->commit_on_device = rf == PTC_OK;
  return rf;
}
}  // namespace

int ptc_scene_commit(ptc_ctx* c) {
  if (!c) return PTC_E_ARG;
  if (!c->have_cam) return fail(c, PTC_E_STATE, "scene_commit: no camera");
  if (c->device >= 0) {
    HIP_TRY(c, hipSetDevice(c->device));
    { int rs = sync_all_lanes(c); if (rs) return rs; }
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (c->device >= 0 && c->bvh_builder == PTC_BVH_LBVH) {      // north_star's tree is the one that builds on the device: PTC_COMMIT=host keeps the host's build of it (the cross-check path)
    const char* how = std::getenv("PTC_COMMIT");
    if (!(how && std::strcmp(how, "host") == 0)) {
      const int rd = device_commit(c, t0);
      if (rd <= 0) return rd;
    }
  }
  auto built = std::make_shared<HostBuilt>();
  const std::string e = ptc_build_scene(c->mats, c->meshes, c->insts, c->texs, c->env, c->toplet_budget, c->bvh_builder, *built);
  if (!e.empty()) return fail(c, PTC_E_STATE, e);
  c->built = built;
  return commit_upload(c, t0);
}

int ptc_frame_begin(ptc_ctx* c, int w, int h, int spp_total, uint64_t seed, int max_bounces, int integrator, int tile_rank, int tile_count) {
  { int rd = need_device(c); if (rd) return rd; }
  c->in_frame = false; c->pending = 0;      // whatever happens below, the previous frame is over
  if (!c->committed) return fail(c, PTC_E_STATE, "frame_begin: scene not committed");
  if (w <= 0 || h <= 0 || spp_total <= 0 || max_bounces < 0 || (uint64_t)w * (uint64_t)h > 0x7fffffffull) return fail(c, PTC_E_ARG, "frame_begin: bad size");
  if (integrator != PTC_INTEGRATOR_PATH && !is_raster(integrator)) return fail(c, PTC_E_ARG, "frame_begin: unknown integrator");
  if (tile_count < 1 || tile_rank < 0 || tile_rank >= tile_count) return fail(c, PTC_E_ARG, "frame_begin: bad tile rank/count");
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  int rc;
  // the list of owned pixels (tile-Morton order) depends on the image size and the tile assignment only: a viewer that renders frame after frame at
  // one size keeps the list it has on the device (2 M entries: 10 ms of host time and an 8 MB upload per frame otherwise — tools/viewer_loop.py)
  if (!(c->owned_key_valid && c->owned_w == w && c->owned_h == h && c->owned_rank == tile_rank && c->owned_count == tile_count && c->owned.p)) {
    std::vector<uint32_t> owned;
    ptc_owned_pixels(w, h, tile_rank, tile_count, owned);
    c->owned_key_valid = false;
    if ((rc = ensure_buf(c, c->owned, owned.size()))) return rc;
    if (!owned.empty()) HIP_TRY(c, hipMemcpy(c->owned.p, owned.data(), owned.size() * 4, hipMemcpyHostToDevice));
    c->owned_n = (uint32_t)owned.size();
    c->owned_w = w; c->owned_h = h; c->owned_rank = tile_rank; c->owned_count = tile_count; c->owned_key_valid = true;
  }
  const size_t n_owned = c->owned_n;
  if ((rc = ensure_buf(c, c->accum, n_owned))) return rc;
  if ((rc = ensure_buf(c, c->radiance, (size_t)w * h))) return rc;
  hipStream_t s0 = c->lanes[0].stream;
  HIP_TRY(c, hipMemsetAsync(c->accum.p, 0, (n_owned ? n_owned : 1) * sizeof(float4), s0));
  HIP_TRY(c, hipMemsetAsync(c->radiance.p, 0, (size_t)w * h * sizeof(float4), s0));
  c->rad_w = w; c->rad_h = h;
  c->fr.w = w; c->fr.h = h; c->fr.max_bounces = max_bounces; c->fr.n_owned = (uint32_t)n_owned; c->fr.owned = c->owned.p;
  {  // seed_hash = pcg(seed_lo + pcg(seed_hi)), same hash as pt_device.h
    auto pcg = [](uint32_t v) { uint32_t s = v * 747796405u + 2891336453u; uint32_t x = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u; return (x >> 22) ^ x; };
    c->fr.seed_hash = pcg((uint32_t)seed + pcg((uint32_t)(seed >> 32)));
  }
  c->spp_total = is_raster(integrator) ? 1 : spp_total;
  c->integrator = integrator; c->samples_done = 0; c->sample_base = 0; c->resolve_divisor = 0;
  // samples of one full batch: as many as fit max_batch_paths split over the lanes.  The queues themselves are sized by the
  // batches actually issued (frame_add_samples), not by spp_total: a progressive loop adding one sample at a time needs
  // queues for one sample per pixel only.
  size_t batch_paths = c->max_batch_paths;
  size_t min_cap = (size_t)-1;
  for (const auto& ln : c->lanes) min_cap = ln.q.cap < min_cap ? ln.q.cap : min_cap;
  if (n_owned && (uint64_t)n_owned * (uint64_t)c->spp_total <= (uint64_t)min_cap) {
    // the whole frame fits the queues every lane already has (frame_add_samples refuses more than spp_total): nothing will be allocated,
    // no need to ask the driver how much memory is free (a call of 0.1-0.2 ms, every frame of a viewer's loop)
  } else {   // no more than 60 % of the device memory that is free now (plus what the lanes' queues already hold) goes into queues
    size_t free_b = 0, total_b = 0, held = 0;
    for (const auto& ln : c->lanes) held += (size_t)ln.q.cap * kQueueBytesPerPath;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
      const size_t fit = (size_t)(0.6 * (double)(free_b + held)) / kQueueBytesPerPath;
      if (fit < batch_paths) batch_paths = fit;
    }
  }
  size_t per = !n_owned ? 1 : batch_paths / n_owned / (size_t)c->n_lanes;
  if (per < 1) per = 1;
  if (per > 0x7fffffffu) per = 0x7fffffffu;
  {   // slot indices are 32 bits, and the segmented layout pads a batch by up to 64 slots per segment
    const uint64_t max_slots = 0xfffffff0ull - 64ull * (uint64_t)c->cfg.shade_waves - 64ull;
    if (n_owned && (uint64_t)n_owned * per > max_slots) per = max_slots / n_owned;
    if (per < 1) per = 1;
  }
  c->per_batch = (uint32_t)per;
  for (auto& ln : c->lanes) HIP_TRY(c, hipMemsetAsync(ln.q.stats, 0, ST_N * ST_STRIDE * sizeof(unsigned long long), ln.stream));
  { int rs = sync_all_lanes(c); if (rs) return rs; }     // accum/radiance/statistics are cleared before any lane starts
  c->batches_issued = 0;
  collect_times(c, true);      // all lanes are idle: every span is complete; the previous frame's times are dropped below
  ptc_stats keep = c->stats;
  std::memset(&c->stats, 0, sizeof c->stats);
  c->stats.seconds_commit = keep.seconds_commit; c->stats.seconds_refit = keep.seconds_refit; c->stats.n_triangles = keep.n_triangles; c->stats.n_bvh_nodes = keep.n_bvh_nodes;
  c->stats.n_emitters = keep.n_emitters; c->stats.bvh_max_depth = keep.bvh_max_depth;
  c->stats.bvh_sa_cost = keep.bvh_sa_cost; c->stats.bvh_sa_cost_built = keep.bvh_sa_cost_built; c->stats.seconds_rebuild = keep.seconds_rebuild;
  c->in_frame = true;
  return PTC_OK;
}

int ptc_frame_reserve(ptc_ctx* c) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->in_frame) return fail(c, PTC_E_STATE, "frame_reserve: no frame");
  if (c->fr.n_owned == 0) return PTC_OK;
  const uint32_t k = is_raster(c->integrator) ? 1u : (c->per_batch < (uint32_t)c->spp_total ? c->per_batch : (uint32_t)c->spp_total);
  return ensure_lane_queues(c, c->fr.n_owned * k);     // n_owned * per_batch fits 32 bits by construction (frame_begin)
}

int ptc_frame_add_samples(ptc_ctx* c, int n_samples) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->in_frame) return fail(c, PTC_E_STATE, "frame_add_samples: no frame");
  if (n_samples <= 0) return fail(c, PTC_E_ARG, "frame_add_samples: n_samples <= 0");
  if (!is_raster(c->integrator) && (uint64_t)c->samples_done + c->pending + (uint64_t)n_samples > (uint64_t)c->spp_total)
    return fail(c, PTC_E_ARG, "frame_add_samples: more samples than the spp_total given to frame_begin");
  c->pending += (uint32_t)n_samples;
  // full batches go out at once; a remainder waits for more samples (or for resolve / sync / a read-back), so that many
  // small calls still produce full-width launches
  while (c->pending >= c->per_batch) {
    c->pending -= c->per_batch;
    if (c->fr.n_owned == 0) { c->samples_done += c->per_batch; continue; }
    int rc = issue(c, c->per_batch);
    if (rc) return rc;
  }
  return PTC_OK;
}

int ptc_frame_resolve(ptc_ctx* c) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->in_frame) return fail(c, PTC_E_STATE, "frame_resolve: no frame");
  { int rf = flush(c); if (rf) return rf; }
  { int rj = join_lanes_on_stream0(c); if (rj) return rj; }
  // divisor: the samples accumulated so far, so a progressive viewer sees a correctly exposed image after every call
  if (c->fr.n_owned && c->samples_done)
    pt_launch_resolve(c->lanes[0].stream, c->fr, c->accum.p, c->radiance.p, (float)(c->resolve_divisor ? c->resolve_divisor : c->samples_done), is_raster(c->integrator));
  HIP_TRY(c, hipGetLastError());
  return PTC_OK;
}

int ptc_frame_set_sample_range(ptc_ctx* c, uint32_t first_sample, uint32_t resolve_divisor) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->in_frame) return fail(c, PTC_E_STATE, "frame_set_sample_range: no frame");
  if (c->samples_done || c->pending) return fail(c, PTC_E_STATE, "frame_set_sample_range: call it right after ptc_frame_begin, before any sample");
  if (is_raster(c->integrator)) return fail(c, PTC_E_ARG, "frame_set_sample_range: the raster integrators have one sample");
  if ((uint64_t)first_sample + (uint64_t)c->spp_total > 0xffffffffull) return fail(c, PTC_E_ARG, "frame_set_sample_range: sample indices exceed 32 bits");
  c->sample_base = first_sample; c->resolve_divisor = resolve_divisor;
  return PTC_OK;
}

int ptc_frame_checkpoint(ptc_ctx* c, float* accum_rgba, uint64_t* n_owned_pixels, uint32_t* samples_done) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->in_frame) return fail(c, PTC_E_STATE, "frame_checkpoint: no frame");
  if (is_raster(c->integrator)) return fail(c, PTC_E_ARG, "frame_checkpoint: the raster integrators have nothing to resume");
  { int rf = flush(c); if (rf) return rf; }
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  if (n_owned_pixels) *n_owned_pixels = c->fr.n_owned;
  if (samples_done) *samples_done = c->samples_done;
  if (accum_rgba && c->fr.n_owned) HIP_TRY(c, hipMemcpy(accum_rgba, c->accum.p, (size_t)c->fr.n_owned * sizeof(float4), hipMemcpyDeviceToHost));
  return PTC_OK;
}

int ptc_frame_restore(ptc_ctx* c, const float* accum_rgba, uint64_t n_owned_pixels, uint32_t samples_done) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->in_frame) return fail(c, PTC_E_STATE, "frame_restore: no frame (ptc_frame_begin with the checkpointed frame's parameters first)");
  if (!accum_rgba) return fail(c, PTC_E_ARG, "frame_restore: null pointer");
  if (c->samples_done || c->pending) return fail(c, PTC_E_STATE, "frame_restore: call it right after ptc_frame_begin, before any sample");
  if (is_raster(c->integrator)) return fail(c, PTC_E_ARG, "frame_restore: the raster integrators have nothing to resume");
  if (n_owned_pixels != c->fr.n_owned) return fail(c, PTC_E_ARG, "frame_restore: the checkpoint is of another frame (size or tile share differ)");
  if (samples_done > (uint32_t)c->spp_total) return fail(c, PTC_E_ARG, "frame_restore: more samples than this frame's spp_total");
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  if (c->fr.n_owned) HIP_TRY(c, hipMemcpy(c->accum.p, accum_rgba, (size_t)c->fr.n_owned * sizeof(float4), hipMemcpyHostToDevice));
  c->samples_done = samples_done;
  return PTC_OK;
}

int ptc_sync(ptc_ctx* c) {
  { int rd = need_device(c); if (rd) return rd; }
  { int rf = flush(c); if (rf) return rf; }
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
  { int rd = need_device(c); if (rd) return rd; }
  if (!out) return fail(c, PTC_E_ARG, "read_radiance: null pointer");
  if (!c->radiance.p || c->rad_w == 0) return fail(c, PTC_E_STATE, "read_radiance: nothing rendered");
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  HIP_TRY(c, hipMemcpy(out, c->radiance.p, (size_t)c->rad_w * c->rad_h * sizeof(float4), hipMemcpyDeviceToHost));
  return PTC_OK;
}

void* ptc_radiance_device_ptr(ptc_ctx* c) { return (c && c->device >= 0) ? (void*)c->radiance.p : nullptr; }

int ptc_read_radiance_rgba16f(ptc_ctx* c, uint16_t* out) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!out) return fail(c, PTC_E_ARG, "read_radiance_rgba16f: null pointer");
  { int rc = convert_half(c); if (rc) return rc; }
  HIP_TRY(c, hipStreamSynchronize(c->lanes[0].stream));
  HIP_TRY(c, hipMemcpy(out, c->half.p, (size_t)c->rad_w * c->rad_h * sizeof(uint2), hipMemcpyDeviceToHost));
  return PTC_OK;
}
void* ptc_radiance_rgba16f_device_ptr(ptc_ctx* c) {
  if (need_device(c)) return nullptr;
  if (convert_half(c)) return nullptr;
  if (hipStreamSynchronize(c->lanes[0].stream) != hipSuccess) return nullptr;
  return (void*)c->half.p;
}

int ptc_write_radiance_rgba32f(ptc_ctx* c, const float* in) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!in) return fail(c, PTC_E_ARG, "write_radiance: null pointer");
  if (!c->radiance.p || c->rad_w == 0) return fail(c, PTC_E_STATE, "write_radiance: no frame");
  { int rs = sync_all_lanes(c); if (rs) return rs; }     // every lane: a resolve or a reduce may still be writing the buffer
  HIP_TRY(c, hipMemcpy(c->radiance.p, in, (size_t)c->rad_w * c->rad_h * sizeof(float4), hipMemcpyHostToDevice));
  return PTC_OK;
}

int ptc_tonemap_rgba8(ptc_ctx* c, uint8_t* out) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!out) return fail(c, PTC_E_ARG, "tonemap: null pointer");
  if (!c->radiance.p || c->rad_w == 0) return fail(c, PTC_E_STATE, "tonemap: nothing rendered");
  int rc;
  if ((rc = ensure_buf(c, c->ldr, (size_t)c->rad_w * c->rad_h))) return rc;
  hipStream_t s0 = c->lanes[0].stream;
  pt_launch_tonemap(s0, c->radiance.p, c->ldr.p, c->rad_w, c->rad_h);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(s0));
  HIP_TRY(c, hipMemcpy(out, c->ldr.p, (size_t)c->rad_w * c->rad_h * 4, hipMemcpyDeviceToHost));
  return PTC_OK;
}

int ptc_get_stats(ptc_ctx* c, ptc_stats* out) {
  if (!c) return PTC_E_ARG;
  if (!out) return fail(c, PTC_E_ARG, "get_stats: null pointer");
  if (c->device < 0) { *out = c->stats; return PTC_OK; }
  HIP_TRY(c, hipSetDevice(c->device));
  { int rf = flush(c); if (rf) return rf; }
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  unsigned long long st[ST_N];
  { int rc = sum_lane_stats(c, st); if (rc) return rc; }
  ptc_stats& s = c->stats;
  s.segments = st[ST_SEGMENTS]; s.shadow_rays = st[ST_SHADOW]; s.hits = st[ST_HITS];
  s.node_visits_closest = st[ST_NODES_C]; s.tri_tests_closest = st[ST_TRIS_C];
  s.node_visits_any = st[ST_NODES_A]; s.tri_tests_any = st[ST_TRIS_A];
  // SURVEY §8d byte model with this build's record sizes (DESIGN.md §"Algorithmic bytes")
  s.algorithmic_bytes = s.segments * (2u * 56u + 2u * 16u) + s.node_visits_closest * 64u + s.tri_tests_closest * 48u + s.hits * 176u +
                        s.shadow_rays * (2u * 44u) + s.node_visits_any * 64u + s.tri_tests_any * 48u + s.paths * (2u * 16u);
  collect_times(c, true);
  *out = c->stats;
  return PTC_OK;
}

// ---- multi-GPU: RCCL reduce of the framebuffer ----------------------------------------------------------
int ptc_comm_unique_id(uint8_t out[PTC_COMM_ID_BYTES]) {
  static_assert(PTC_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ptc.h mirrors NCCL_UNIQUE_ID_BYTES");
  if (!out) return PTC_E_ARG;
  if (!rccl_load()) { g_create_error = g_rccl.err; return PTC_E_DEVICE; }
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return PTC_E_DEVICE; }
  std::memcpy(out, id.internal, PTC_COMM_ID_BYTES);
  return PTC_OK;
}

int ptc_comm_init(ptc_ctx* c, const uint8_t id[PTC_COMM_ID_BYTES], int rank, int n_ranks) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(c, PTC_E_ARG, "comm_init: bad argument");
  if (c->comm) return fail(c, PTC_E_STATE, "comm_init: this context already has a communicator");
  if (!rccl_load()) return fail(c, PTC_E_DEVICE, g_rccl.err);
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, PTC_COMM_ID_BYTES);
  NCCL_TRY(c, g_rccl.CommInitRank(&c->comm, n_ranks, uid, rank));
  c->comm_rank = rank; c->comm_size = n_ranks; c->comm_owned = true;
  return PTC_OK;
}

int ptc_comm_reduce_radiance(ptc_ctx* c, int root) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->comm) return fail(c, PTC_E_STATE, "comm_reduce_radiance: no communicator (ptc_comm_init / ptc_group_create)");
  if (root < 0 || root >= c->comm_size) return fail(c, PTC_E_ARG, "comm_reduce_radiance: bad root");
  if (!c->radiance.p || c->rad_w == 0) return fail(c, PTC_E_STATE, "comm_reduce_radiance: nothing rendered");
  // in place on stream 0, behind the resolve: ranks own disjoint tiles and hold zeros elsewhere, so the fp32 sum is x + 0
  ScopedSpan t(c, c->lanes[0].stream, 4);            // seconds_reduce: the collective as this rank's stream sees it (it includes waiting for the slowest rank)
  NCCL_TRY(c, g_rccl.Reduce(c->radiance.p, c->radiance.p, (size_t)c->rad_w * c->rad_h * 4, ncclFloat32, ncclSum, root, c->comm, c->lanes[0].stream));
  return PTC_OK;
}

int ptc_comm_destroy(ptc_ctx* c) {
  { int rd = need_device(c); if (rd) return rd; }
  if (!c->comm) return PTC_OK;
  if (!c->comm_owned) return fail(c, PTC_E_STATE, "comm_destroy: the communicator belongs to a ptc_group");
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  NCCL_TRY(c, g_rccl.CommDestroy(c->comm));
  c->comm = nullptr; c->comm_size = 0;
  return PTC_OK;
}

ptc_group* ptc_group_create(const int* device_ids, int n_devices) {
  if (!device_ids || n_devices < 1 || n_devices > 64) { g_create_error = "ptc_group_create: bad argument"; return nullptr; }
  bool none = true;
  for (int i = 0; i < n_devices; ++i) none = none && device_ids[i] == PTC_DEVICE_NONE;
  if (none) {      // a description-only group (every id PTC_DEVICE_NONE): the host half of the group calls — one build shared by all contexts — without GPUs or RCCL
    ptc_group* g = new ptc_group();
    for (int i = 0; i < n_devices; ++i) g->ctx.push_back(ptc_create(PTC_DEVICE_NONE));
    return g;
  }
  if (!rccl_load()) { g_create_error = g_rccl.err; return nullptr; }
  ptc_group* g = new ptc_group();
  for (int i = 0; i < n_devices; ++i) {
    ptc_ctx* c = ptc_create(device_ids[i]);
    if (!c) { ptc_group_destroy(g); return nullptr; }
    g->ctx.push_back(c);
  }
  g->comms.resize((size_t)n_devices, nullptr);
  const ncclResult_t r = g_rccl.CommInitAll(g->comms.data(), n_devices, device_ids);
  if (r != ncclSuccess) { g_create_error = std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r); g->comms.clear(); ptc_group_destroy(g); return nullptr; }
  for (int i = 0; i < n_devices; ++i) {
    ptc_ctx* c = g->ctx[(size_t)i];
    c->comm = g->comms[(size_t)i]; c->comm_rank = i; c->comm_size = n_devices; c->comm_owned = false;
  }
  return g;
}

int ptc_group_size(const ptc_group* g) { return g ? (int)g->ctx.size() : 0; }

int ptc_group_scene_commit(ptc_group* g) {
  if (!g || g->ctx.empty()) return PTC_E_ARG;
  ptc_ctx* c0 = g->ctx[0];
  int rc = c0->committed ? PTC_OK : ptc_scene_commit(c0);     // flatten + BVH build, once, on the host (a scene device 0 has committed already is taken as it is)
  if (rc) { g->err = std::string("device 0: ") + ptc_last_error(c0); return rc; }
  for (size_t i = 1; i < g->ctx.size(); ++i) {
    ptc_ctx* c = g->ctx[i];
    if (c->device >= 0) {
      if (hipSetDevice(c->device) != hipSuccess) { g->err = "ptc_group_scene_commit: hipSetDevice failed"; return PTC_E_DEVICE; }
      if ((rc = sync_all_lanes(c))) { g->err = "device " + std::to_string(i) + ": " + ptc_last_error(c); return rc; }
    }
    const auto t0 = std::chrono::steady_clock::now();
    // the description travels too (materials are counted from it, a later ptc_scene_commit on this context rebuilds from it)
    c->mats = c0->mats; c->meshes = c0->meshes; c->insts = c0->insts; c->texs = c0->texs; c->env = c0->env;
    std::memcpy(c->cam_pos, c0->cam_pos, 12); std::memcpy(c->cam_target, c0->cam_target, 12); c->cam_fov = c0->cam_fov; c->cam_aspect = c0->cam_aspect;
    c->have_cam = true; c->tex_linear = c0->tex_linear; c->bvh_builder = c0->bvh_builder; c->toplet_budget = c0->toplet_budget;
    c->built = c0->built;                             // shared, read-only from here on
    if ((rc = commit_upload(c, t0))) { g->err = "device " + std::to_string(i) + ": " + ptc_last_error(c); return rc; }
  }
  return PTC_OK;
}
ptc_ctx* ptc_group_ctx(ptc_group* g, int i) { return (g && i >= 0 && (size_t)i < g->ctx.size()) ? g->ctx[(size_t)i] : nullptr; }
const char* ptc_group_last_error(const ptc_group* g) { return g ? g->err.c_str() : g_create_error.c_str(); }

int ptc_group_scene_refit(ptc_group* g) {
  if (!g || g->ctx.empty()) return PTC_E_ARG;
  ptc_ctx* c0 = g->ctx[0];
  if (!c0->committed) { g->err = "ptc_group_scene_refit: the group's scene is not committed"; return PTC_E_STATE; }
  if (!description_matches_commit(c0)) { g->err = kDescriptionChanged; return PTC_E_STATE; }
  // device 0's instances carry the new transforms (ptc_update_instance* on ptc_group_ctx(g, 0)): one refit on the host, the arrays go to every device
  for (ptc_ctx* c : g->ctx) {
    if (c->device < 0) continue;
    if (hipSetDevice(c->device) != hipSuccess) { g->err = "ptc_group_scene_refit: hipSetDevice failed"; return PTC_E_DEVICE; }
    int rc = flush(c); if (!rc) rc = sync_all_lanes(c);
    if (rc) { g->err = std::string("ptc_group_scene_refit: ") + ptc_last_error(c); return rc; }
  }
  const auto t0 = std::chrono::steady_clock::now();
  if (c0->device >= 0 && refit_on_device(c0)) {       // every device refits its own copy in place: nothing but the 84 bytes per instance and the emitter table cross the bus
    bool host_way = false;
    auto mine = std::make_shared<HostBuilt>(*c0->built);
    for (size_t i = 0; i < g->ctx.size() && !host_way; ++i) {
      ptc_ctx* c = g->ctx[i];
      if (hipSetDevice(c->device) != hipSuccess) { g->err = "ptc_group_scene_refit: hipSetDevice failed"; return PTC_E_DEVICE; }
      if (i) c->insts = c0->insts;
      c->built = mine;
      const int rc = device_refit(c);
      if (rc > 0) { host_way = true; break; }      // decided from the description alone, before any kernel ran: all devices take the host path together
      if (rc) {
        g->err = "device " + std::to_string(i) + ": " + ptc_last_error(c);
        if (i == 0) return rc;                     // nothing has been refitted yet (a refused refit leaves the device's scene as it was)
        host_way = true; break;                    // devices 0..i-1 hold the new state: the host path below brings ALL of them to one state, or fails as a whole
      }
      c->in_frame = false; c->pending = 0;
      c->stats.seconds_refit = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    if (!host_way) return PTC_OK;
  }
  auto built = std::make_shared<HostBuilt>(*c0->built);                 // the devices keep rendering from the old arrays until theirs are overwritten
  const size_t n_recs = built->recs.size(), n_shade = built->shade.size(), n_lights = built->lights.size(), n_cdf = built->cdf.size();
  const std::string e = ptc_refit_scene(c0->mats, c0->meshes, c0->insts, c0->texs, c0->env, *built);
  if (!e.empty()) { g->err = e; return PTC_E_STATE; }
  const bool same = built->recs.size() == n_recs && built->shade.size() == n_shade && built->lights.size() == n_lights && built->cdf.size() == n_cdf;
  for (size_t i = 0; i < g->ctx.size(); ++i) {
    ptc_ctx* c = g->ctx[i];
    if (c->device >= 0 && hipSetDevice(c->device) != hipSuccess) { g->err = "ptc_group_scene_refit: hipSetDevice failed"; return PTC_E_DEVICE; }
    if (i) c->insts = c0->insts;
    c->built = built;
    c->in_frame = false; c->pending = 0;
    const int rc = c->device >= 0 ? refit_upload(c, same, t0) : PTC_OK;
    if (rc) { g->err = "device " + std::to_string(i) + ": " + ptc_last_error(c); return rc; }
    c->stats.seconds_refit = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  return PTC_OK;
}

int ptc_group_render(ptc_group* g, int w, int h, int spp, uint64_t seed, int max_bounces, int integrator) {
  if (!g || g->ctx.empty()) return PTC_E_ARG;
  const int n = (int)g->ctx.size();
  auto bail = [&](int i, int rc) { g->err = std::string("device ") + std::to_string(i) + ": " + ptc_last_error(g->ctx[(size_t)i]); return rc; };
  // every device traces all samples of its tiles; all of it is queued before anything is waited for
  for (int i = 0; i < n; ++i) { int rc = ptc_frame_begin(g->ctx[(size_t)i], w, h, spp, seed, max_bounces, integrator, i, n); if (rc) return bail(i, rc); }
  for (int i = 0; i < n; ++i) { int rc = ptc_frame_add_samples(g->ctx[(size_t)i], spp); if (rc) return bail(i, rc); }
  for (int i = 0; i < n; ++i) { int rc = ptc_frame_resolve(g->ctx[(size_t)i]); if (rc) return bail(i, rc); }
  if (n > 1) {
    ncclResult_t r = g_rccl.GroupStart();
    for (int i = 0; i < n && r == ncclSuccess; ++i) {
      ptc_ctx* c = g->ctx[(size_t)i];
      if (hipSetDevice(c->device) != hipSuccess) { g->err = "ptc_group_render: hipSetDevice failed"; (void)g_rccl.GroupEnd(); return PTC_E_DEVICE; }
      r = g_rccl.Reduce(c->radiance.p, c->radiance.p, (size_t)w * h * 4, ncclFloat32, ncclSum, 0, c->comm, c->lanes[0].stream);
    }
    const ncclResult_t r2 = g_rccl.GroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess) { g->err = std::string("ptc_group_render: ncclReduce: ") + g_rccl.GetErrorString(r != ncclSuccess ? r : r2); return PTC_E_DEVICE; }
  }
  for (int i = 0; i < n; ++i) { int rc = ptc_sync(g->ctx[(size_t)i]); if (rc) return bail(i, rc); }
  return PTC_OK;
}

void ptc_group_destroy(ptc_group* g) {
  if (!g) return;
  for (ptc_ctx* c : g->ctx)
    if (c && c->device >= 0) { (void)hipSetDevice(c->device); for (auto& ln : c->lanes) if (ln.stream) (void)hipStreamSynchronize(ln.stream); }
  for (ncclComm_t cm : g->comms) if (cm && g_rccl.so) (void)g_rccl.CommDestroy(cm);
  for (ptc_ctx* c : g->ctx) { if (c) { c->comm = nullptr; ptc_destroy(c); } }
  delete g;
}

// ---- test hooks -----------------------------------------------------------------------------------
int ptc_debug_trace_closest(ptc_ctx* c, const float* origins, const float* dirs, uint32_t n, float* out_t, int32_t* out_prim, float* out_uv) {
  if (!c) return PTC_E_ARG;
  if (c->device >= 0 && (!origins || !dirs || !out_t || !out_prim || !out_uv || n == 0)) return fail(c, PTC_E_ARG, "debug_trace_closest: bad argument");
  { int rc = debug_prepare(c, n, "debug_trace_closest"); if (rc) return rc; }
  const Lane& ln = c->lanes[0];
  std::vector<float4> A(n), B(n);
  for (uint32_t i = 0; i < n; ++i) {
    A[i] = make_float4(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2], dirs[i * 3]);
    B[i] = make_float4(dirs[i * 3 + 1], dirs[i * 3 + 2], 0.0f, 0.0f);
  }
  HIP_TRY(c, hipMemcpy(ln.q.ray[0].A, A.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(ln.q.ray[0].B, B.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  const DevQueues q = batch_queues(c, 0, n);
  pt_launch_set_counts(ln.stream, c->cfg, q, n, 0);
  pt_launch_trace_closest(ln.stream, c->cfg, lane_scene(c, 0), q, 0, false);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(ln.stream));
  std::vector<float4> H(n);
  HIP_TRY(c, hipMemcpy(H.data(), ln.q.hit, n * sizeof(float4), hipMemcpyDeviceToHost));
  for (uint32_t i = 0; i < n; ++i) {
    int32_t pc; std::memcpy(&pc, &H[i].y, 4);   // prim | class<<28, or -1
    out_t[i] = H[i].x; out_prim[i] = pc < 0 ? -1 : (pc & 0x0fffffff); out_uv[i * 2] = H[i].z; out_uv[i * 2 + 1] = H[i].w;
  }
  return PTC_OK;
}

int ptc_debug_trace_any(ptc_ctx* c, const float* origins, const float* dirs, const float* tmax, uint32_t n, uint8_t* out_occluded) {
  if (!c) return PTC_E_ARG;
  if (c->device >= 0 && (!origins || !dirs || !tmax || !out_occluded || n == 0)) return fail(c, PTC_E_ARG, "debug_trace_any: bad argument");
  { int rc = debug_prepare(c, n, "debug_trace_any"); if (rc) return rc; }
  const Lane& ln = c->lanes[0];
  std::vector<float4> A(n), B(n);
  for (uint32_t i = 0; i < n; ++i) {
    A[i] = make_float4(origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2], dirs[i * 3]);
    B[i] = make_float4(dirs[i * 3 + 1], dirs[i * 3 + 2], tmax[i], 0.0f);
  }
  HIP_TRY(c, hipMemcpy(ln.q.shadow.A, A.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(ln.q.shadow.B, B.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  uint8_t* d_out = nullptr;
  HIP_TRY(c, hipMalloc((void**)&d_out, n));
  const DevQueues q = batch_queues(c, 0, n);
  pt_launch_set_counts(ln.stream, c->cfg, q, 0, n);
  pt_launch_trace_any(ln.stream, c->cfg, lane_scene(c, 0), q, d_out);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipStreamSynchronize(ln.stream);
  if (e == hipSuccess) e = hipMemcpy(out_occluded, d_out, n, hipMemcpyDeviceToHost);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail(c, PTC_E_DEVICE, std::string("debug_trace_any: ") + hipGetErrorString(e));
  return PTC_OK;
}

int ptc_debug_get_flat_scene(ptc_ctx* c, uint32_t* n_verts, uint32_t* n_tris, ptc_vertex* verts, uint32_t* indices, int32_t* tri_material) {
  if (!c) return PTC_E_ARG;
  if (!c->committed) return fail(c, PTC_E_STATE, "debug_get_flat_scene: scene not committed");
  { int rr = refresh_host_copy(c); if (rr) return rr; }
  const HostBuilt& B = *c->built;
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
  { int rd = need_device(c); if (rd) return rd; }
  if (!out || n <= 0) return fail(c, PTC_E_ARG, "debug_get_counters: bad argument");
  { int rf = flush(c); if (rf) return rf; }
  { int rs = sync_all_lanes(c); if (rs) return rs; }
  unsigned long long st[ST_N];
  { int rc = sum_lane_stats(c, st); if (rc) return rc; }
  for (int i = 0; i < n; ++i) out[i] = i < ST_N ? st[i] : 0;
  return ST_N;
}

int ptc_debug_get_bvh(ptc_ctx* c, uint32_t* n_nodes, uint32_t* n_tris, uint32_t* n_units, float* units, float grid[6]) {
  if (!c) return PTC_E_ARG;
  if (!c->committed) return fail(c, PTC_E_STATE, "debug_get_bvh: scene not committed");
  { int rr = refresh_host_copy(c); if (rr) return rr; }
  const HostBuilt& B = *c->built;
  if (n_nodes) *n_nodes = B.n_nodes;
  if (n_tris) *n_tris = B.n_tri_records;
  if (n_units) *n_units = B.n_units;
  if (units) std::memcpy(units, B.recs.data(), B.recs.size() * 4);
  if (grid) for (int k = 0; k < 3; ++k) { grid[k] = B.grid_lo[k]; grid[3 + k] = B.grid_step[k]; }
  return PTC_OK;
}

// Context internals for tests of the host logic: [0] HIP events created so far, [1] timing spans waiting to be collected,
// [2] queue capacity (paths) of lane 0, [3] samples of one full batch, [4] samples accepted but not yet issued,
// [5] trace blocks per CU, [6] stack entries per lane kept in LDS.
// identity of the host build a context renders from (the contexts of a group share one: ptc_group_scene_commit): tests compare the values
uint64_t ptc_debug_host_build_id(const ptc_ctx* c) { return c ? (uint64_t)(uintptr_t)c->built.get() : 0u; }

int ptc_debug_get_internals(ptc_ctx* c, uint64_t out[8]) {
  if (!c || !out) return PTC_E_ARG;
  for (int i = 0; i < 8; ++i) out[i] = 0;
  out[0] = c->events_created; out[1] = c->spans.size(); out[2] = c->lanes.empty() ? 0 : c->lanes[0].q.cap; out[3] = c->per_batch; out[4] = c->pending;
  out[5] = (uint64_t)c->cfg.trace_blocks_per_cu; out[6] = (uint64_t)c->cfg.stack_lds; out[7] = (c->last_refit_on_device ? 1u : 0u) | (c->commit_on_device ? 2u : 0u);
  return PTC_OK;
}

// The host's share of a refit on the device, run without a device (CPU tests, sanitizer builds): builds the plan of the committed scene and the
// emitter table of the CURRENT transforms from the emissive primitives alone, and checks them against the host build — call it after
// ptc_scene_refit on a description-only context.  out: [0] world vertices, [1] primitives, [2] 8-wide nodes in the level lists, [3] levels,
// [4] emissive-material primitives, [5] 1 if the level lists hold every node address of the tree exactly once with children after parents,
// [6] 1 if the emitter table and cdf equal the host refit's bit for bit (0 also when the set of emitters changed), [7] 1 if all transforms are finite.
int ptc_debug_refit_host_parts(ptc_ctx* c, uint64_t out[8]) {
  if (!c || !out) return PTC_E_ARG;
  if (!c->committed) return fail(c, PTC_E_STATE, "debug_refit_host_parts: scene not committed");
  { int rr = refresh_host_copy(c); if (rr) return rr; }
  const HostBuilt& B = *c->built;
  RefitPlan P;
  ptc_refit_plan(c->mats, c->meshes, c->insts, B, P);
  std::vector<float> xf, lights, cdf;
  const bool finite = ptc_refit_instance_transforms(c->insts, xf);
  const bool same_set = ptc_refit_emitters(c->mats, c->meshes, c->insts, P, B, lights, cdf);
  out[0] = P.n_verts; out[1] = P.n_tris; out[2] = P.level_nodes.size(); out[3] = P.level_first.empty() ? 0 : P.level_first.size() - 1; out[4] = P.emit_prims.size() / 5;
  // every node once, and a node's children (its block's interior records) in an earlier level than the node itself
  bool ok = P.level_nodes.size() == B.n_nodes && !P.level_first.empty() && P.level_first.back() == P.level_nodes.size() && P.vert_inst.size() == B.wverts.size();
  std::vector<int32_t> level_of((size_t)B.n_units / 4 + 1, -1);
  for (size_t l = 0; ok && l + 1 < P.level_first.size(); ++l)
    for (uint32_t i = P.level_first[l]; i < P.level_first[l + 1]; ++i) {
      const uint32_t a = P.level_nodes[i];
      if ((a & 3u) || a >= B.n_units || level_of[a >> 2] >= 0) { ok = false; break; }
      level_of[a >> 2] = (int32_t)l;
    }
  for (size_t i = 0; ok && i < P.level_nodes.size(); ++i) {
    const uint32_t a = P.level_nodes[i];
    uint32_t w2, w3; std::memcpy(&w2, &B.recs[(size_t)a * 4 + 2], 4); std::memcpy(&w3, &B.recs[(size_t)a * 4 + 3], 4);
    const uint32_t imask = (w2 >> 8) & 255u;
    for (uint32_t k = 0; k < (uint32_t)__builtin_popcount(imask); ++k) {
      const uint32_t ch = w3 + 4u * k;
      if (ch >= B.n_units || level_of[ch >> 2] < 0 || level_of[ch >> 2] >= level_of[a >> 2]) { ok = false; break; }
    }
  }
  out[5] = ok ? 1u : 0u;
  out[6] = (same_set && lights.size() == B.lights.size() && cdf.size() == B.cdf.size() && std::memcmp(lights.data(), B.lights.data(), lights.size() * 4) == 0 &&
            std::memcmp(cdf.data(), B.cdf.data(), cdf.size() * 4) == 0) ? 1u : 0u;
  out[7] = finite ? 1u : 0u;
  return PTC_OK;
}

// The host's share of a COMMIT on the device (ptc_build_skeleton), run without a device (CPU tests, sanitizer builds): describes the committed description again the way
// device_commit does — no flatten, the emitter table from the emissive primitives alone — and holds it against the host build the context was committed with.
// out: [0] primitives, [1] emitters, [2] 1 if world vertex indices and material per primitive agree, [3] 1 if the emitter index per primitive, the emitter table and its cdf
// agree bit for bit, [4] 1 if the material table agrees, [5] 1 if textures, texture sets and environment tables agree, [6] 1 if shading-record stride and vertex count agree.
int ptc_debug_commit_host_parts(ptc_ctx* c, uint64_t out[8]) {
  if (!c || !out) return PTC_E_ARG;
  if (!c->committed) return fail(c, PTC_E_STATE, "debug_commit_host_parts: scene not committed");
  { int rr = refresh_host_copy(c); if (rr) return rr; }
  if (!description_matches_commit(c)) return fail(c, PTC_E_STATE, kDescriptionChanged);
  const HostBuilt& B = *c->built;
  HostBuilt S;
  const std::string e = ptc_build_skeleton(c->mats, c->meshes, c->insts, c->texs, c->env, c->toplet_budget, S);
  if (!e.empty()) return fail(c, PTC_E_STATE, e);
  for (int i = 0; i < 8; ++i) out[i] = 0;
  auto same = [](const auto& a, const auto& b) { return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(a[0])) == 0); };
  out[0] = S.n_tris; out[1] = S.n_lights;
  out[2] = (S.n_tris == B.n_tris && same(S.widx, B.widx) && same(S.tri_mat, B.tri_mat)) ? 1u : 0u;
  out[3] = (S.n_lights == B.n_lights && same(S.prim_light, B.prim_light) && same(S.lights, B.lights) && same(S.cdf, B.cdf)) ? 1u : 0u;
  out[4] = same(S.mats, B.mats) ? 1u : 0u;
  out[5] = (same(S.texels, B.texels) && same(S.tex_info, B.tex_info) && same(S.set_texels, B.set_texels) && same(S.set_info, B.set_info) && same(S.env, B.env) && same(S.env_marg, B.env_marg) &&
            same(S.env_cond, B.env_cond) && same(S.env_marg_guide, B.env_marg_guide) && same(S.env_cond_guide, B.env_cond_guide) && S.env_w == B.env_w && S.env_h == B.env_h && S.env_ok == B.env_ok) ? 1u : 0u;
  out[6] = (S.shade_stride == B.shade_stride && S.n_wverts == B.n_wverts && B.n_wverts == B.wverts.size()) ? 1u : 0u;
  return PTC_OK;
}

// The tables k_shade reads besides the BVH: shading records (4 * stride floats per primitive), emitters (20 floats each), their power cdf.
// Sizes come back through the pointers; arrays may be null.
int ptc_debug_get_shading_tables(ptc_ctx* c, uint32_t* stride, float* shade, uint32_t* n_lights, float* lights, float* cdf) {
  if (!c) return PTC_E_ARG;
  if (!c->committed) return fail(c, PTC_E_STATE, "debug_get_shading_tables: scene not committed");
  { int rr = refresh_host_copy(c); if (rr) return rr; }
  const HostBuilt& B = *c->built;
  if (stride) *stride = B.shade_stride;
  if (n_lights) *n_lights = B.n_lights;
  if (shade) std::memcpy(shade, B.shade.data(), B.shade.size() * 4);
  if (lights) std::memcpy(lights, B.lights.data(), B.lights.size() * 4);
  if (cdf) std::memcpy(cdf, B.cdf.data(), B.cdf.size() * 4);
  return PTC_OK;
}

}  // extern "C"
