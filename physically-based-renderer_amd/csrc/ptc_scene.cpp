// ptc_scene.cpp — host side of ptc_scene_commit: flatten instances to world space, build the 8-wide SAH BVH,
// lay it out for the trace kernels, build the emitter table.
//
// Reference conventions implemented here (file:line under the reference checkout):
//   R2  primitive concatenation        src/pbr_engine/engine/pbr/MeshBuilder.cpp:16-55
//   R3  model = T·R·S, normal matrix   src/pbr_engine/engine/pbr/ModelPushConstant.hpp:33-46
//   R4  world-space N/T                assets/shaders/geometry_pass/vertex.glsl:25-36
//   R5  camera basis                   src/pbr_engine/engine/pbr/CameraData.hpp:22-32
// Everything is plain IEEE binary32 without contraction (this file is compiled -ffp-contract=off),
// so the flattened scene is reproducible bit for bit.
#include "ptc_internal.h"
#include "pt_refit.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <chrono>
#include <pthread.h>
#include <sched.h>
#include <unistd.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <new>
#include <thread>

namespace {

// A parallel loop over [0, n) in chunks for the per-vertex / per-triangle / per-node passes of the flatten, of the emission and of a refit: independent
// element by element, so the results do not depend on the split.  The workers are a small pool that sleeps between loops (PTC_HOST_THREADS or up to 16
// threads, the caller included): threads created per loop would not do — a new thread starts on its parent's CPU and is moved to an idle one only at the
// next scheduler tick, and a chunk of these passes (3 ms) is over before that: eight fresh threads ran one after the other.
class HostPool {
public:
  static HostPool& get() {
    static HostPool p;
    static const int once = pthread_atfork(nullptr, nullptr, [] {      // a forked child has none of the workers: it forgets them (never joins them) and loops alone
      HostPool& q = get();
      new (&q.workers_) std::vector<std::thread>();
      new (&q.cv_) std::condition_variable();          // the parent's workers wait on these: pthread_cond_destroy would wait for them to leave, forever
      new (&q.finished_) std::condition_variable();
      new (&q.m_) std::mutex();
      new (&q.run_mutex_) std::mutex();
      q.n_threads_ = 1;
      q.cur_ = nullptr;
    });
    (void)once;
    return p;
  }
  unsigned size() const { return n_threads_; }
  // fn(chunk) for chunk in [0, n_chunks), on the pool's threads and the caller
  template <class F> void run(size_t n_chunks, F&& fn) {
    if (n_chunks == 0) return;
    if (n_threads_ <= 1 || n_chunks == 1) { for (size_t i = 0; i < n_chunks; ++i) fn(i); return; }
    std::unique_lock<std::mutex> one(run_mutex_);          // one loop at a time (contexts on several host threads may commit concurrently)
    std::function<void(size_t)> job = [&fn](size_t i) { fn(i); };
    Run r;                                                  // the loop's own tickets and counts: a worker that wakes late for an EARLIER loop holds that loop's Run, or none
    r.job = &job; r.n_chunks = n_chunks;
    {
      std::lock_guard<std::mutex> g(m_);
      cur_ = &r; ++epoch_;
    }
    cv_.notify_all();
    work(r, false);
    std::unique_lock<std::mutex> g(m_);
    finished_.wait(g, [&] { return r.done == r.n_chunks && r.inside == 0; });     // every chunk done AND every worker that took this Run has left it: `r` and `job` may go
    cur_ = nullptr;
  }
private:
  HostPool() {
    unsigned t = std::thread::hardware_concurrency();
    if (const char* e = std::getenv("PTC_HOST_THREADS")) { const int v = std::atoi(e); if (v >= 1) t = (unsigned)v; }
    n_threads_ = t < 1u ? 1u : (t > 16u ? 16u : t);
    // A worker first visits a CPU of its own (the i-th of the process's allowed set, rotated by the process id) and then gets the whole set back:
    // the scheduler wakes a sleeping thread on the CPU it last ran on when that CPU is idle, so the workers start their loops spread over the machine
    // instead of queued behind the caller (where they sat for a whole 30 ms loop on a VM whose CPUs share no cache domain) — and they are not pinned.
    cpu_set_t all;
    CPU_ZERO(&all);
    const bool have_set = sched_getaffinity(0, sizeof all, &all) == 0;
    std::vector<int> allowed;
    if (have_set) for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &all)) allowed.push_back(c);
    const unsigned rot = (unsigned)getpid();
    for (unsigned i = 1; i < n_threads_; ++i)
      workers_.emplace_back([this, i, all, allowed, rot] {
        if (allowed.size() > 1) {
          cpu_set_t one;
          CPU_ZERO(&one);
          CPU_SET(allowed[(i + rot) % allowed.size()], &one);
          if (pthread_setaffinity_np(pthread_self(), sizeof one, &one) == 0) { sched_yield(); (void)pthread_setaffinity_np(pthread_self(), sizeof all, &all); }
        }
        loop();
      });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> g(m_); stop_ = true; }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  struct Run { const std::function<void(size_t)>* job = nullptr; size_t n_chunks = 0, done = 0; unsigned inside = 0; std::atomic<size_t> next{0}; };
  void work(Run& r, bool worker) {
    size_t mine = 0;
    for (;;) {
      const size_t i = r.next.fetch_add(1);
      if (i >= r.n_chunks) break;
      (*r.job)(i);
      ++mine;
    }
    std::lock_guard<std::mutex> g(m_);
    r.done += mine;
    if (worker) --r.inside;
    if (r.done == r.n_chunks && r.inside == 0) finished_.notify_all();
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      Run* r;
      {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
        r = cur_;                       // under the lock: either the loop that is running (it cannot return before `inside` is back to 0) or none
        if (r) ++r->inside;
      }
      if (r) work(*r, true);
    }
  }
  unsigned n_threads_ = 1;
  std::vector<std::thread> workers_;
  std::mutex m_, run_mutex_;
  std::condition_variable cv_, finished_;
  Run* cur_ = nullptr;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};
template <class F> void parallel_for(size_t n, size_t min_chunk, F&& fn) {
  HostPool& pool = HostPool::get();
  size_t nc = min_chunk ? n / min_chunk : 1;                  // chunks: a few per thread, so that uneven ones even out
  if (nc > (size_t)pool.size() * 4) nc = (size_t)pool.size() * 4;
  if (nc <= 1 || pool.size() <= 1) { fn((size_t)0, n); return; }
  const size_t per = (n + nc - 1) / nc;
  pool.run(nc, [&](size_t c) {
    const size_t lo = c * per, hi = lo + per < n ? lo + per : n;
    if (lo < hi) fn(lo, hi);
  });
}

constexpr int kLeafMax = 2;
constexpr int kWide = 8;      // child slots per BVH node
constexpr int kNodeWords = 16; // 64-byte node = 4 units of 16 bytes
constexpr float kCostNode = 1.0f, kCostPrim = 1.0f;   // surface-area cost of one node visit / one triangle test (collapse)
constexpr int kTile = 32;

struct Mat34 { float m[16]; float n[9]; };  // column-major model (glm layout) + normal matrix

// mat3(transpose(inverse(model))) of a column-major 4x4 model matrix, via the cofactors of its upper 3x3
// (makeModelPushConstant(glm::mat4x4), ModelPushConstant.hpp:33-38).
Mat34 from_matrix(const float m[16]) {
  Mat34 r;
  std::memcpy(r.m, m, 64);
  float A[9];
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 3; ++k) A[c * 3 + k] = m[c * 4 + k];
  auto a = [&](int row, int col) { return A[col * 3 + row]; };
  const float c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
  const float c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
  const float c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
  const float c10 = a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2);
  const float c11 = a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0);
  const float c12 = a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1);
  const float c20 = a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1);
  const float c21 = a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2);
  const float c22 = a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0);
  const float det = a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02;
  const float id = 1.0f / det;
  r.n[0] = c00 * id; r.n[1] = c10 * id; r.n[2] = c20 * id;
  r.n[3] = c01 * id; r.n[4] = c11 * id; r.n[5] = c21 * id;
  r.n[6] = c02 * id; r.n[7] = c12 * id; r.n[8] = c22 * id;
  return r;
}

inline float fdot(const float a[3], const float b[3]) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
inline void fcross(const float a[3], const float b[3], float o[3]) {
  o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
  o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
  o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
inline void fnormalize(float v[3]) {
  const float inv = 1.0f / sqrtf(fdot(v, v));
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
}
inline void mul_n(const float N[9], const float v[3], float o[3]) {
  o[0] = N[0] * v[0] + N[3] * v[1] + N[6] * v[2];
  o[1] = N[1] * v[0] + N[4] * v[1] + N[7] * v[2];
  o[2] = N[2] * v[0] + N[5] * v[1] + N[8] * v[2];
}

struct Box { float lo[3], hi[3]; };
inline Box empty_box() {
  Box b;
  for (int k = 0; k < 3; ++k) { b.lo[k] = std::numeric_limits<float>::infinity(); b.hi[k] = -std::numeric_limits<float>::infinity(); }
  return b;
}
inline void grow(Box& b, const Box& o) {
  for (int k = 0; k < 3; ++k) { b.lo[k] = o.lo[k] < b.lo[k] ? o.lo[k] : b.lo[k]; b.hi[k] = o.hi[k] > b.hi[k] ? o.hi[k] : b.hi[k]; }
}

// ---- binary tree by top-down binned surface-area splits -------------------------------------------------
// SplitNode covers positions [lo,hi] of `ord`; child >= 0: node index; < 0: ~position of a single triangle.
// Per range: kSahBins equal bins over the centroid bounds on each axis, cost = half_area(L)·nL + half_area(R)·nR,
// minimum over (axis, boundary) with ties to the lowest axis then boundary, STABLE partition; coincident
// centroids → cut at the middle index.  The binary tree goes down to single triangles; which subtrees become the
// wide tree's leaves is decided by the collapse (ptc_build_scene).
struct SplitNode { uint32_t lo, hi; int32_t left, right; };
constexpr int kSahBins = 32;

inline float box_half_area(const Box& b) {
  const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
  return ex * ey + ey * ez + ez * ex;
}

// ---- LBVH option (BASELINE north_star's "flattened LBVH"): the binary tree is the radix tree of 63-bit Morton codes ------------
// Per axis q = (uint32)((centre - cl) * (2^21 / (ch - cl))) clamped to 2^21 - 1 over the bounds cl..ch of all box centres (0 on a
// degenerate axis); x in bit 0, y in bit 1, z in bit 2 of each triple.  Positions are ordered by (code, primitive id); the tree is the radix
// tree of the keys (code, sorted position): a range splits where the highest bit in which its end KEYS differ turns 1 — a bit of the code or,
// in a range of one code, of the position — so that every internal node can be found on its own (the device build, pt_build.hip).
inline uint64_t spread_bits_3(uint32_t v) {
  uint64_t r = 0;
  for (int b = 0; b < 21; ++b) r |= (uint64_t)((v >> b) & 1u) << (3 * b);
  return r;
}
std::vector<uint64_t> sort_by_morton_code(const std::vector<float>& ctr, std::vector<uint32_t>& ord) {
  const uint32_t n = (uint32_t)ord.size();
  float cl[3], ch[3], scale[3];
  for (int k = 0; k < 3; ++k) { cl[k] = std::numeric_limits<float>::infinity(); ch[k] = -cl[k]; }
  for (uint32_t p = 0; p < n; ++p)
    for (int k = 0; k < 3; ++k) { const float c = ctr[(size_t)p * 3 + k]; cl[k] = cl[k] < c ? cl[k] : c; ch[k] = ch[k] > c ? ch[k] : c; }
  for (int k = 0; k < 3; ++k) scale[k] = ch[k] > cl[k] ? 2097152.0f / (ch[k] - cl[k]) : 0.0f;
  std::vector<std::pair<uint64_t, uint32_t>> keyed(n);
  for (uint32_t p = 0; p < n; ++p) {
    uint64_t code = 0;
    for (int k = 0; k < 3; ++k) {
      const float f = (ctr[(size_t)p * 3 + k] - cl[k]) * scale[k];
      const uint32_t q = f >= 2097151.0f ? 2097151u : (uint32_t)f;
      code |= spread_bits_3(q) << k;
    }
    keyed[p] = {code, p};
  }
  std::sort(keyed.begin(), keyed.end());
  std::vector<uint64_t> codes(n);
  for (uint32_t i = 0; i < n; ++i) { codes[i] = keyed[i].first; ord[i] = keyed[i].second; }
  return codes;
}
inline uint32_t morton_last_left(const std::vector<uint64_t>& codes, uint32_t lo, uint32_t hi) {
  const uint64_t diff = codes[lo] ^ codes[hi];
  if (!diff) {                             // one code: the key goes on with the sorted position (Karras 2012, sec. 4)
    const uint32_t pbit = 1u << (31 - __builtin_clz(lo ^ hi));
    return (hi & ~(pbit - 1u)) - 1u;
  }
  const uint64_t top = (uint64_t)1 << (63 - __builtin_clzll(diff));
  // the bit is clear at lo and set at hi, and the codes are sorted: find the last position where it is clear
  const auto first_set = std::partition_point(codes.begin() + lo, codes.begin() + hi + 1, [top](uint64_t c) { return (c & top) == 0; });
  return (uint32_t)(first_set - codes.begin()) - 1u;
}

// Where the range [lo, hi] of `ord` is cut: returns the last position of the left part (the SAH path partitions ord[lo..hi] in place, stably; `scratch`
// holds hi - lo + 1 entries).  A pure function of the range's content, so disjoint ranges can be cut by different threads.
uint32_t split_range(const std::vector<Box>& tbox, const std::vector<float>& ctr, const std::vector<uint64_t>& codes, bool lbvh, std::vector<uint32_t>& ord,
                     uint32_t* scratch, uint32_t lo, uint32_t hi) {
  if (lbvh) return morton_last_left(codes, lo, hi);
  struct Bin { Box box; uint32_t count; };
  float cl[3], ch[3];
  for (int k = 0; k < 3; ++k) { cl[k] = std::numeric_limits<float>::infinity(); ch[k] = -cl[k]; }
  for (uint32_t i = lo; i <= hi; ++i) {
    const float* c = &ctr[(size_t)ord[i] * 3];
    for (int k = 0; k < 3; ++k) { cl[k] = cl[k] < c[k] ? cl[k] : c[k]; ch[k] = ch[k] > c[k] ? ch[k] : c[k]; }
  }
  float best_cost = std::numeric_limits<float>::infinity();
  int best_axis = -1, best_bin = 0;
  for (int k = 0; k < 3; ++k) {
    if (!(ch[k] > cl[k])) continue;
    const float scale = (float)kSahBins / (ch[k] - cl[k]);
    Bin bins[kSahBins];
    for (Bin& bn : bins) { bn.box = empty_box(); bn.count = 0; }
    for (uint32_t i = lo; i <= hi; ++i) {
      const uint32_t p = ord[i];
      int j = (int)((ctr[(size_t)p * 3 + k] - cl[k]) * scale);
      j = j > kSahBins - 1 ? kSahBins - 1 : j;
      bins[j].count++;
      grow(bins[j].box, tbox[p]);
    }
    float suffix_area[kSahBins];
    uint32_t suffix_count[kSahBins];
    Box acc = empty_box();
    uint32_t cnt = 0;
    for (int j = kSahBins - 1; j >= 1; --j) {
      cnt += bins[j].count;
      grow(acc, bins[j].box);
      suffix_count[j] = cnt;
      suffix_area[j] = cnt ? box_half_area(acc) : 0.0f;
    }
    acc = empty_box();
    cnt = 0;
    for (int j = 0; j + 1 < kSahBins; ++j) {
      cnt += bins[j].count;
      grow(acc, bins[j].box);
      if (cnt == 0 || suffix_count[j + 1] == 0) continue;
      const float cost = box_half_area(acc) * (float)cnt + suffix_area[j + 1] * (float)suffix_count[j + 1];
      if (cost < best_cost) { best_cost = cost; best_axis = k; best_bin = j; }
    }
  }
  if (best_axis < 0) return lo + (hi - lo) / 2;
  const float scale = (float)kSahBins / (ch[best_axis] - cl[best_axis]);
  uint32_t nl = 0, nr = 0;
  for (uint32_t i = lo; i <= hi; ++i) {
    const uint32_t p = ord[i];
    int j = (int)((ctr[(size_t)p * 3 + best_axis] - cl[best_axis]) * scale);
    j = j > kSahBins - 1 ? kSahBins - 1 : j;
    if (j <= best_bin) ord[lo + nl++] = p; else scratch[nr++] = p;
  }
  std::copy(scratch, scratch + nr, ord.begin() + lo + nl);
  return lo + nl - 1;
}

// split_range for a LARGE range, its passes spread over the pool (the top of the tree is a handful of such ranges and everything else waits for them).
// Same cut: bounds and bins are minima / maxima / counts — the same in any order (only the sign of a zero can differ, and areas and comparisons do not read
// it) — and the partition keeps the order of both parts (per-chunk counts, exclusive prefix, every chunk writes its own stretch of the scratch buffer).
uint32_t split_range_parallel(const std::vector<Box>& tbox, const std::vector<float>& ctr, std::vector<uint32_t>& ord, uint32_t* scratch, uint32_t lo, uint32_t hi) {
  struct Bin { Box box; uint32_t count; };
  const size_t n = (size_t)hi - lo + 1;
  const size_t nc = std::min<size_t>((size_t)HostPool::get().size() * 2, n / 8192);
  const size_t per = (n + nc - 1) / nc;
  auto chunk = [&](size_t c, size_t& a, size_t& b) { a = lo + c * per; b = std::min<size_t>(a + per, (size_t)hi + 1); };
  struct Bounds { float cl[3], ch[3]; };
  std::vector<Bounds> cb(nc);
  HostPool::get().run(nc, [&](size_t c) {
    size_t a, b; chunk(c, a, b);
    Bounds r;
    for (int k = 0; k < 3; ++k) { r.cl[k] = std::numeric_limits<float>::infinity(); r.ch[k] = -r.cl[k]; }
    for (size_t i = a; i < b; ++i) {
      const float* q = &ctr[(size_t)ord[i] * 3];
      for (int k = 0; k < 3; ++k) { r.cl[k] = r.cl[k] < q[k] ? r.cl[k] : q[k]; r.ch[k] = r.ch[k] > q[k] ? r.ch[k] : q[k]; }
    }
    cb[c] = r;
  });
  float cl[3], ch[3];
  for (int k = 0; k < 3; ++k) { cl[k] = std::numeric_limits<float>::infinity(); ch[k] = -cl[k]; }
  for (const Bounds& r : cb)
    for (int k = 0; k < 3; ++k) { cl[k] = cl[k] < r.cl[k] ? cl[k] : r.cl[k]; ch[k] = ch[k] > r.ch[k] ? ch[k] : r.ch[k]; }
  float scale[3];
  for (int k = 0; k < 3; ++k) scale[k] = ch[k] > cl[k] ? (float)kSahBins / (ch[k] - cl[k]) : 0.0f;
  std::vector<Bin> part(nc * 3 * kSahBins);
  HostPool::get().run(nc, [&](size_t c) {
    size_t a, b; chunk(c, a, b);
    Bin* bins = &part[c * 3 * kSahBins];
    for (int j = 0; j < 3 * kSahBins; ++j) { bins[j].box = empty_box(); bins[j].count = 0; }
    for (size_t i = a; i < b; ++i) {
      const uint32_t p = ord[i];
      for (int k = 0; k < 3; ++k) {
        if (!(ch[k] > cl[k])) continue;
        int j = (int)((ctr[(size_t)p * 3 + (size_t)k] - cl[k]) * scale[k]);
        j = j > kSahBins - 1 ? kSahBins - 1 : j;
        bins[k * kSahBins + j].count++;
        grow(bins[k * kSahBins + j].box, tbox[p]);
      }
    }
  });
  float best_cost = std::numeric_limits<float>::infinity();
  int best_axis = -1, best_bin = 0;
  for (int k = 0; k < 3; ++k) {
    if (!(ch[k] > cl[k])) continue;
    Bin bins[kSahBins];
    for (Bin& bn : bins) { bn.box = empty_box(); bn.count = 0; }
    for (size_t c = 0; c < nc; ++c)
      for (int j = 0; j < kSahBins; ++j) { const Bin& o = part[(c * 3 + (size_t)k) * kSahBins + (size_t)j]; if (o.count) { bins[j].count += o.count; grow(bins[j].box, o.box); } }
    float suffix_area[kSahBins];
    uint32_t suffix_count[kSahBins];
    Box acc = empty_box();
    uint32_t cnt = 0;
    for (int j = kSahBins - 1; j >= 1; --j) {
      cnt += bins[j].count;
      grow(acc, bins[j].box);
      suffix_count[j] = cnt;
      suffix_area[j] = cnt ? box_half_area(acc) : 0.0f;
    }
    acc = empty_box();
    cnt = 0;
    for (int j = 0; j + 1 < kSahBins; ++j) {
      cnt += bins[j].count;
      grow(acc, bins[j].box);
      if (cnt == 0 || suffix_count[j + 1] == 0) continue;
      const float cost = box_half_area(acc) * (float)cnt + suffix_area[j + 1] * (float)suffix_count[j + 1];
      if (cost < best_cost) { best_cost = cost; best_axis = k; best_bin = j; }
    }
  }
  if (best_axis < 0) return lo + (hi - lo) / 2;
  const float sc = scale[best_axis], c0 = cl[best_axis];
  auto goes_left = [&](uint32_t p) { int j = (int)((ctr[(size_t)p * 3 + (size_t)best_axis] - c0) * sc); j = j > kSahBins - 1 ? kSahBins - 1 : j; return j <= best_bin; };
  std::vector<uint32_t> nleft(nc + 1, 0u);
  HostPool::get().run(nc, [&](size_t c) {
    size_t a, b; chunk(c, a, b);
    uint32_t k = 0;
    for (size_t i = a; i < b; ++i) k += goes_left(ord[i]) ? 1u : 0u;
    nleft[c + 1] = k;
  });
  for (size_t c = 0; c < nc; ++c) nleft[c + 1] += nleft[c];
  const uint32_t nl = nleft[nc];
  HostPool::get().run(nc, [&](size_t c) {
    size_t a, b; chunk(c, a, b);
    uint32_t l = nleft[c], r = nl + (uint32_t)(a - lo) - nleft[c];
    for (size_t i = a; i < b; ++i) { const uint32_t p = ord[i]; if (goes_left(p)) scratch[l++] = p; else scratch[r++] = p; }
  });
  HostPool::get().run(nc, [&](size_t c) {
    size_t a, b; chunk(c, a, b);
    std::copy(scratch + (a - lo), scratch + (b - lo), ord.begin() + (ptrdiff_t)a);
  });
  return lo + nl - 1;
}

// The subtree over [lo, hi] (hi > lo), appended to `out` with indices local to `out` (its root is out[first]); parents before children.
void build_subtree(const std::vector<Box>& tbox, const std::vector<float>& ctr, const std::vector<uint64_t>& codes, bool lbvh, std::vector<uint32_t>& ord,
                   uint32_t* scratch, uint32_t lo0, uint32_t hi0, std::vector<SplitNode>& out) {
  struct Work { uint32_t lo, hi; int32_t node; };
  std::vector<Work> todo;
  const int32_t root = (int32_t)out.size();
  out.push_back({lo0, hi0, 0, 0});
  todo.push_back({lo0, hi0, root});
  while (!todo.empty()) {
    const Work w = todo.back();
    todo.pop_back();
    const uint32_t last_left = split_range(tbox, ctr, codes, lbvh, ord, scratch, w.lo, w.hi);
    auto link = [&](uint32_t lo, uint32_t hi) -> int32_t {
      if (lo == hi) return ~(int32_t)lo;
      const int32_t id = (int32_t)out.size();
      out.push_back({lo, hi, 0, 0});
      todo.push_back({lo, hi, id});
      return id;
    };
    const int32_t l = link(w.lo, last_left), r = link(last_left + 1, w.hi);
    out[(size_t)w.node].left = l;
    out[(size_t)w.node].right = r;
  }
}

// The binary tree over all triangles.  The top is cut on the calling thread until the ranges are small (the cuts of a large range are what
// everything below waits for, and they are few); every range at or below that size is a task for the pool, built into its own array and
// appended afterwards — its nodes after all nodes of the top, so that parents still come before children in `out` (what the bottom-up passes rely
// on).  The tree is the one the sequential build makes: a cut depends on its range alone.
void build_split_tree(const std::vector<Box>& tbox, std::vector<uint32_t>& ord, std::vector<SplitNode>& out, bool lbvh, std::vector<uint32_t>& sub_first) {
  const uint32_t n = (uint32_t)ord.size();
  sub_first.clear();
  std::vector<float> ctr((size_t)n * 3);
  parallel_for(n, 16384, [&](size_t p0, size_t p1) {
    for (size_t p = p0; p < p1; ++p)
      for (int k = 0; k < 3; ++k) ctr[p * 3 + (size_t)k] = 0.5f * (tbox[p].lo[k] + tbox[p].hi[k]);
  });
  std::vector<uint64_t> codes;
  if (lbvh) codes = sort_by_morton_code(ctr, ord);
  std::vector<uint32_t> right_part(n);
  out.clear();
  out.reserve(n);
  const unsigned threads = HostPool::get().size();
  const uint32_t task_size = threads > 1 ? std::max<uint32_t>(2048u, n / (threads * 8u)) : n;      // one thread: the whole tree is "the top"
  struct Work { uint32_t lo, hi; int32_t node; };
  struct Task { uint32_t lo, hi; int32_t parent; bool right; };
  std::vector<Work> todo;
  std::vector<Task> tasks;
  out.push_back({0, n - 1, 0, 0});
  todo.push_back({0, n - 1, 0});
  while (!todo.empty()) {
    const Work w = todo.back();
    todo.pop_back();
    const uint32_t last_left = (!lbvh && threads > 1 && w.hi - w.lo + 1u >= 32768u) ? split_range_parallel(tbox, ctr, ord, right_part.data() + w.lo, w.lo, w.hi)
                                                                                     : split_range(tbox, ctr, codes, lbvh, ord, right_part.data() + w.lo, w.lo, w.hi);
    auto link = [&](uint32_t lo, uint32_t hi, bool right) -> int32_t {
      if (lo == hi) return ~(int32_t)lo;
      if (threads > 1 && hi - lo + 1u <= task_size) { tasks.push_back({lo, hi, w.node, right}); return 0; }     // patched when the task's nodes have their places
      const int32_t id = (int32_t)out.size();
      out.push_back({lo, hi, 0, 0});
      todo.push_back({lo, hi, id});
      return id;
    };
    const int32_t l = link(w.lo, last_left, false), r = link(last_left + 1, w.hi, true);
    out[(size_t)w.node].left = l;
    out[(size_t)w.node].right = r;
  }
  if (tasks.empty()) return;
  sub_first.push_back((uint32_t)out.size());           // node indices [sub_first[t], sub_first[t + 1]) are task t's subtree; below sub_first[0]: the top
  std::vector<std::vector<SplitNode>> sub(tasks.size());
  parallel_for(tasks.size(), 1, [&](size_t t0, size_t t1) {
    for (size_t t = t0; t < t1; ++t) {
      sub[t].reserve(tasks[t].hi - tasks[t].lo);
      build_subtree(tbox, ctr, codes, lbvh, ord, right_part.data() + tasks[t].lo, tasks[t].lo, tasks[t].hi, sub[t]);
    }
  });
  for (size_t t = 0; t < tasks.size(); ++t) {
    const int32_t base = (int32_t)out.size();
    for (const SplitNode& nd : sub[t]) out.push_back({nd.lo, nd.hi, nd.left >= 0 ? nd.left + base : nd.left, nd.right >= 0 ? nd.right + base : nd.right});
    (tasks[t].right ? out[(size_t)tasks[t].parent].right : out[(size_t)tasks[t].parent].left) = base;
    sub_first.push_back((uint32_t)out.size());
  }
}

inline int32_t leaf_code(uint32_t first, uint32_t count) { return (int32_t) ~(first | ((count - 1u) << 28)); }

// ---- what a refit keeps (HostBuilt::topology): the order of the triangles, the binary tree they were split into, which of its
// subtrees the 8-wide nodes took as children and in which slots, and where every node and children block lies in the unit array.
// ptc_refit_scene recomputes everything that depends on vertex positions — boxes, quantised planes, triangle and shading records, emitters —
// and leaves these alone: the tree keeps its shape and its layout, so the new unit array has the size of the old one.
struct WChild { bool leaf; uint32_t lo, hi; int32_t radix; Box box; };
struct Wide { WChild slot[kWide]; bool used[kWide]; int n; };
struct Slot { int32_t radix; uint32_t depth; };
struct Topology {
  std::vector<uint32_t> ord;
  std::vector<SplitNode> radix;
  std::vector<Slot> order;
  std::vector<Wide> wide;
  std::vector<uint32_t> child_base, block_order, node_addr, block_addr;
  uint64_t next_unit = 4;
  uint32_t n_tri_records = 0, toplet_budget = 0;
  std::vector<float> wbt;            // world bitangent per vertex
  std::vector<uint32_t> sub_first;   // binary-tree node ranges of the subtrees the build handed to the pool (bottom-up passes run over them in parallel, then over the top)
};

}  // namespace

// =================================================================================================
// translate(t) * toMat4(q) * scale(s)  (makeModelPushConstant(position, rotation, scale), ModelPushConstant.hpp:40-46);
// column-major like glm, quaternion order (w,x,y,z).
void ptc_trs_to_matrix(const float t[3], const float q[4], const float s[3], float m[16]) {
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  const float xx = x * x, yy = y * y, zz = z * z, xz = x * z, xy = x * y, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
  float R[9];
  R[0] = 1.0f - 2.0f * (yy + zz); R[1] = 2.0f * (xy + wz);        R[2] = 2.0f * (xz - wy);
  R[3] = 2.0f * (xy - wz);        R[4] = 1.0f - 2.0f * (xx + zz); R[5] = 2.0f * (yz + wx);
  R[6] = 2.0f * (xz + wy);        R[7] = 2.0f * (yz - wx);        R[8] = 1.0f - 2.0f * (xx + yy);
  for (int c = 0; c < 3; ++c) {
    for (int k = 0; k < 3; ++k) m[c * 4 + k] = R[c * 3 + k] * s[c];
    m[c * 4 + 3] = 0.0f;
  }
  m[12] = t[0]; m[13] = t[1]; m[14] = t[2]; m[15] = 1.0f;
}

void ptc_make_camera(const float pos[3], const float target[3], float fov, float aspect, DevCamera& cam) {
  // glm::lookAtRH(pos, target, up = (0,-1,0)): f = normalize(target-pos), s = normalize(f × up), u = s × f
  float f[3] = {target[0] - pos[0], target[1] - pos[1], target[2] - pos[2]};
  fnormalize(f);
  const float up[3] = {0.0f, -1.0f, 0.0f};
  float s[3], u[3];
  fcross(f, up, s);
  fnormalize(s);
  fcross(s, f, u);
  for (int k = 0; k < 3; ++k) { cam.pos[k] = pos[k]; cam.f[k] = f[k]; cam.s[k] = s[k]; cam.u[k] = u[k]; }
  const float th = (float)std::tan((double)fov * 0.5);   // 1/proj[1][1] of glm::perspectiveRH_NO
  cam.sy = th;
  cam.sx = aspect * th;
}

void ptc_owned_pixels(int w, int h, int tile_rank, int tile_count, std::vector<uint32_t>& out) {
  auto compact = [](uint32_t x) {
    x &= 0x55555555u; x = (x ^ (x >> 1)) & 0x33333333u; x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu; x = (x ^ (x >> 8)) & 0x0000ffffu; return x;
  };
  const uint32_t tx = (uint32_t)(w + kTile - 1) / kTile, ty = (uint32_t)(h + kTile - 1) / kTile;
  uint32_t side = 1;
  while (side < tx || side < ty) side <<= 1;
  out.clear();
  uint32_t t = 0;
  for (uint32_t m = 0; m < side * side; ++m) {       // Morton walk over the tile grid
    const uint32_t cx = compact(m), cy = compact(m >> 1);
    if (cx >= tx || cy >= ty) continue;
    const bool mine = (t % (uint32_t)tile_count) == (uint32_t)tile_rank;
    ++t;
    if (!mine) continue;
    for (uint32_t p = 0; p < (uint32_t)(kTile * kTile); ++p) {   // Morton inside the tile: 64 consecutive = 8×8 block
      const uint32_t x = cx * kTile + compact(p), y = cy * kTile + compact(p >> 1);
      if (x < (uint32_t)w && y < (uint32_t)h) out.push_back(y * (uint32_t)w + x);
    }
  }
}

namespace {
// A texture SET is a distinct (colour, normal, metal-rough) triple of texture ids among the materials that have a texture, numbered in material order.
void material_sets(const std::vector<HostMaterial>& mats, std::vector<int32_t>& mat_set, std::vector<int32_t>& set_tex) {
  mat_set.assign(mats.size(), -1); set_tex.clear();
  for (size_t i = 0; i < mats.size(); ++i) {
    const HostMaterial& m = mats[i];
    if (m.tex_color < 0 && m.tex_normal < 0 && m.tex_mr < 0) continue;
    int32_t found = -1;
    for (size_t k = 0; k * 3 < set_tex.size(); ++k)
      if (set_tex[k * 3] == m.tex_color && set_tex[k * 3 + 1] == m.tex_normal && set_tex[k * 3 + 2] == m.tex_mr) { found = (int32_t)k; break; }
    if (found < 0) { found = (int32_t)(set_tex.size() / 3); set_tex.push_back(m.tex_color); set_tex.push_back(m.tex_normal); set_tex.push_back(m.tex_mr); }
    mat_set[i] = found;
  }
}
void fill_materials(const std::vector<HostMaterial>& mats, const std::vector<int32_t>& mat_set, HostBuilt& B) {
  B.mats.assign(mats.size() * 16, 0.0f);
  for (size_t i = 0; i < mats.size(); ++i) {
    float* o = &B.mats[i * 16];
    std::memcpy(&o[12], &mat_set[i], 4);
    const HostMaterial& m = mats[i];
    o[0] = m.base[0]; o[1] = m.base[1]; o[2] = m.base[2]; o[3] = m.metallic;
    o[4] = m.emissive[0]; o[5] = m.emissive[1]; o[6] = m.emissive[2]; o[7] = m.roughness;
    o[8] = m.base[3]; std::memcpy(&o[9], &m.tex_color, 4); std::memcpy(&o[10], &m.tex_normal, 4); std::memcpy(&o[11], &m.tex_mr, 4);
  }
}
// textures, texture sets and the environment light of a commit (they do not depend on the instances' transforms)
void fill_textures_env(const std::vector<HostTexture>& texs, const HostEnv& env, const std::vector<int32_t>& set_tex, HostBuilt& B) {
  for (const auto& t : texs) {
    const int32_t info[4] = {(int32_t)B.texels.size(), t.w, t.h, 0};
    B.tex_info.insert(B.tex_info.end(), info, info + 4);
    const size_t np = (size_t)t.w * (size_t)t.h;
    for (size_t i = 0; i < np; ++i) {
      uint32_t u; std::memcpy(&u, &t.px[i * 4], 4);
      B.texels.push_back(u);
    }
  }
  if (B.texels.empty()) B.texels.push_back(0u);
  if (B.tex_info.empty()) B.tex_info.assign(4, 0);
  // texture sets: the set's textures interleaved per texel (absent texture: 0), 8x8 tiles row-major over the image (width and height
  // padded to multiples of 8), Morton order inside a tile
  for (size_t k = 0; k * 3 < set_tex.size(); ++k) {
    int w = -1, h = -1; bool same = true;
    for (int j = 0; j < 3; ++j) {
      const int32_t t = set_tex[k * 3 + (size_t)j];
      if (t < 0) continue;
      if (w < 0) { w = texs[(size_t)t].w; h = texs[(size_t)t].h; }
      else if (texs[(size_t)t].w != w || texs[(size_t)t].h != h) same = false;
    }
    if (!same || w <= 0) { const int32_t info[4] = {-1, 0, 0, 0}; B.set_info.insert(B.set_info.end(), info, info + 4); continue; }
    const uint32_t tw = ((uint32_t)w + 7u) / 8u, th = ((uint32_t)h + 7u) / 8u;
    const size_t off = B.set_texels.size() / 4;
    if (off + (size_t)tw * th * 64 > 0x7fffffffull) { const int32_t info[4] = {-1, 0, 0, 0}; B.set_info.insert(B.set_info.end(), info, info + 4); continue; }
    const int32_t info[4] = {(int32_t)off, w, h, (int32_t)tw};
    B.set_info.insert(B.set_info.end(), info, info + 4);
    B.set_texels.resize((off + (size_t)tw * th * 64) * 4, 0u);
    for (int y = 0; y < h; ++y)
      for (int x = 0; x < w; ++x) {
        const uint32_t lx = (uint32_t)x & 7u, ly = (uint32_t)y & 7u;
        const uint32_t mo = (lx & 1u) | ((ly & 1u) << 1) | ((lx & 2u) << 1) | ((ly & 2u) << 2) | ((lx & 4u) << 2) | ((ly & 4u) << 3);
        const size_t at = off + ((size_t)((uint32_t)y >> 3) * tw + ((uint32_t)x >> 3)) * 64 + mo;
        for (int j = 0; j < 3; ++j) {
          const int32_t t = set_tex[k * 3 + (size_t)j];
          if (t < 0) continue;
          uint32_t u; std::memcpy(&u, &texs[(size_t)t].px[((size_t)y * (size_t)w + (size_t)x) * 4], 4);
          B.set_texels[at * 4 + (size_t)j] = u;
        }
      }
  }
  if (B.set_texels.empty()) B.set_texels.assign(4, 0u);
  if (B.set_info.empty()) { const int32_t info[4] = {-1, 0, 0, 0}; B.set_info.assign(info, info + 4); }
  // ---- environment light: radiance + pmf per texel, row-marginal and per-row conditional cdfs ---------------
  B.env_w = env.w; B.env_h = env.h; B.env_ok = 0;
  if (env.w > 0 && env.h > 0) {
    const int w = env.w, hgt = env.h;
    const size_t np = (size_t)w * (size_t)hgt;
    B.env.resize(np * 4); B.env_cond.resize(np); B.env_marg.resize((size_t)hgt);
    std::vector<float> rowsum((size_t)hgt);
    float totalw = 0.0f;
    for (int y = 0; y < hgt; ++y) {
      const float sr = (float)std::sin(3.14159265358979323846 * ((double)y + 0.5) / (double)hgt);
      float runw = 0.0f;
      for (int x = 0; x < w; ++x) {
        const float* t = &env.rgb[((size_t)y * w + x) * 3];
        float f = fmaf(t[2], 0.0722f, fmaf(t[1], 0.7152f, t[0] * 0.2126f)) * sr;
        if (!(f > 0.0f)) f = 0.0f;
        float* o = &B.env[((size_t)y * w + x) * 4];
        o[0] = t[0]; o[1] = t[1]; o[2] = t[2]; o[3] = f;
        runw += f;
        B.env_cond[(size_t)y * w + x] = runw;
      }
      rowsum[(size_t)y] = runw; totalw += runw;
      for (int x = 0; x < w; ++x) B.env_cond[(size_t)y * w + x] = runw > 0.0f ? B.env_cond[(size_t)y * w + x] / runw : (float)(x + 1) / (float)w;
      B.env_cond[(size_t)y * w + (size_t)(w - 1)] = 1.0f;
    }
    if (totalw > 0.0f) {
      float runw = 0.0f;
      for (int y = 0; y < hgt; ++y) { runw += rowsum[(size_t)y]; B.env_marg[(size_t)y] = runw / totalw; }
      B.env_marg[(size_t)hgt - 1] = 1.0f;
      for (size_t i = 0; i < np; ++i) B.env[i * 4 + 3] = B.env[i * 4 + 3] / totalw;
      B.env_ok = 1;
    } else {
      for (size_t i = 0; i < np; ++i) B.env[i * 4 + 3] = 0.0f;
    }
  }
  // guide tables of the two cdf searches of env_sample: guide[b] = the index the search returns for r = b / PTC_ENV_GUIDE (16 bits: maps up to 65536 texels wide and high)
  {
    auto search = [](const float* cdf, uint32_t n, float r) { uint32_t lo = 0, hi = n - 1u; while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cdf[mid] > r) hi = mid; else lo = mid + 1u; } return lo; };
    if (B.env_ok) {
      const uint32_t G = PTC_ENV_GUIDE;
      B.env_marg_guide.resize(G + 1);
      for (uint32_t b = 0; b <= G; ++b) B.env_marg_guide[b] = (uint16_t)search(B.env_marg.data(), (uint32_t)B.env_h, (float)b / (float)G);
      B.env_cond_guide.resize((size_t)B.env_h * (G + 1));
      for (int y = 0; y < B.env_h; ++y)
        for (uint32_t b = 0; b <= G; ++b) B.env_cond_guide[(size_t)y * (G + 1) + b] = (uint16_t)search(&B.env_cond[(size_t)y * (size_t)B.env_w], (uint32_t)B.env_w, (float)b / (float)G);
    }
    if (B.env_marg_guide.empty()) B.env_marg_guide.assign(PTC_ENV_GUIDE + 1, 0);
    if (B.env_cond_guide.empty()) B.env_cond_guide.assign(PTC_ENV_GUIDE + 1, 0);
  }
}

std::string build_or_refit(const std::vector<HostMaterial>& mats, const std::vector<HostMesh>& meshes,
                           const std::vector<HostInstance>& insts, const std::vector<HostTexture>& texs, const HostEnv& env,
                           uint32_t toplet_budget, int bvh_builder, HostBuilt& B, bool refit) {
  if (insts.empty()) return "scene_commit: no instances";
  const bool timing = std::getenv("PTC_BUILD_TIMING") != nullptr;      // phase times of the host build / refit on stderr
  auto tprev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "  %-28s %7.2f ms\n", what, std::chrono::duration<double, std::milli>(now - tprev).count());
    tprev = now;
  };
  uint64_t nv = 0, nt = 0;
  for (const auto& in : insts) { nv += meshes[(size_t)in.mesh].v.size(); nt += meshes[(size_t)in.mesh].idx.size() / 3; }
  if (nt >= (1u << 28)) return "scene_commit: too many triangles";
  std::shared_ptr<Topology> topo;
  if (refit) {
    topo = std::static_pointer_cast<Topology>(B.topology);
    if (!topo || B.n_tris != nt || B.n_wverts != nv) return "scene_refit: the scene's meshes or instances changed since the commit (only transforms may)";
    toplet_budget = topo->toplet_budget;
  } else {
    B = HostBuilt();
    topo = std::make_shared<Topology>();
    topo->toplet_budget = toplet_budget;
  }
  // A refit flattens into temporaries and takes them over only when every position is finite: a refused refit leaves the host build as it was (the device
  // path restores its scratch vertices likewise), so the debug getters keep describing what is being rendered.
  std::vector<HostVertex> wverts_new;
  std::vector<float> wbt_new;
  std::vector<HostVertex>& wverts = refit ? wverts_new : B.wverts;
  std::vector<float>& wbt = refit ? wbt_new : topo->wbt;   // world bitangent per vertex (vertex.glsl:35)
  wverts.resize(nv);
  B.n_wverts = (uint32_t)nv;
  wbt.resize(nv * 3);
  B.widx.resize(nt * 3);
  B.tri_mat.resize(nt);
  // ---- flatten ----------------------------------------------------------------------------------
  std::vector<uint32_t> inst_vb(insts.size()), inst_tb(insts.size());
  {
    uint32_t vb = 0, tb = 0;
    for (size_t i = 0; i < insts.size(); ++i) {
      inst_vb[i] = vb; inst_tb[i] = tb;
      vb += (uint32_t)meshes[(size_t)insts[i].mesh].v.size(); tb += (uint32_t)(meshes[(size_t)insts[i].mesh].idx.size() / 3);
    }
  }
  // work items: (instance, slice of 4096 of its vertices) and (instance, slice of its triangles), so that one large mesh does not serialise the pass
  struct Slice { uint32_t inst, lo, hi; };
  std::vector<Slice> vslices, tslices;
  for (size_t i = 0; i < insts.size(); ++i) {
    const HostMesh& m = meshes[(size_t)insts[i].mesh];
    for (size_t lo = 0; lo < m.v.size(); lo += 4096) vslices.push_back({(uint32_t)i, (uint32_t)lo, (uint32_t)(lo + 4096 < m.v.size() ? lo + 4096 : m.v.size())});
    const size_t ntm = m.idx.size() / 3;
    for (size_t lo = 0; lo < ntm; lo += 8192) tslices.push_back({(uint32_t)i, (uint32_t)lo, (uint32_t)(lo + 8192 < ntm ? lo + 8192 : ntm)});
  }
  parallel_for(vslices.size(), 1, [&](size_t s0, size_t s1) {
  for (size_t si = s0; si < s1; ++si) {
    const HostInstance& in = insts[vslices[si].inst];
    const HostMesh& m = meshes[(size_t)in.mesh];
    const Mat34 M = from_matrix(in.m);
    const uint32_t vb = inst_vb[vslices[si].inst];
    for (size_t k = vslices[si].lo; k < vslices[si].hi; ++k) {
      const HostVertex& s = m.v[k];
      HostVertex& d = wverts[vb + k];
      for (int r = 0; r < 3; ++r)   // m[0]*x + m[1]*y + m[2]*z + m[3]
        d.position[r] = M.m[0 + r] * s.position[0] + M.m[4 + r] * s.position[1] + M.m[8 + r] * s.position[2] + M.m[12 + r];
      mul_n(M.n, s.normal, d.normal);
      fnormalize(d.normal);
      mul_n(M.n, s.tangent, d.tangent);
      fnormalize(d.tangent);
      d.tangent[3] = s.tangent[3];
      {
        float cr[3], sc3[3], bt[3];
        fcross(s.normal, s.tangent, cr);
        sc3[0] = cr[0] * s.tangent[3]; sc3[1] = cr[1] * s.tangent[3]; sc3[2] = cr[2] * s.tangent[3];
        mul_n(M.n, sc3, bt);
        fnormalize(bt);
        wbt[(size_t)(vb + k) * 3 + 0] = bt[0]; wbt[(size_t)(vb + k) * 3 + 1] = bt[1]; wbt[(size_t)(vb + k) * 3 + 2] = bt[2];
      }
      d.texcoord[0] = s.texcoord[0];
      d.texcoord[1] = s.texcoord[1];
    }
  }
  });
  parallel_for(tslices.size(), 1, [&](size_t s0, size_t s1) {
  for (size_t si = s0; si < s1; ++si) {
    const HostInstance& in = insts[tslices[si].inst];
    const HostMesh& m = meshes[(size_t)in.mesh];
    const uint32_t vb = inst_vb[tslices[si].inst], tb = inst_tb[tslices[si].inst];
    for (size_t k = tslices[si].lo; k < tslices[si].hi; ++k) {
      for (int c = 0; c < 3; ++c) B.widx[(tb + k) * 3 + c] = vb + m.idx[k * 3 + c];
      B.tri_mat[tb + k] = m.material;
    }
  }
  });
  lap("flatten");
  const uint32_t n = (uint32_t)nt;
  B.n_tris = n;
  // NaN / Inf anywhere in the flattened positions (bad vertices or a bad instance matrix) is an error: the builder
  // computes bin indices from them
  for (const auto& v : wverts)
    if (!(std::isfinite(v.position[0]) && std::isfinite(v.position[1]) && std::isfinite(v.position[2]))) return "scene_commit: non-finite vertex position after the instance transform";
  if (refit) { B.wverts.swap(wverts_new); topo->wbt.swap(wbt_new); }
  // ---- triangle boxes, scene bounds ------------------------------------------------------------------
  std::vector<Box> tbox(n);
  Box sb = empty_box();
  {
    std::vector<Box> part(64, empty_box());          // min / max are exact and order-independent: the merge gives the sequential scene box
    std::atomic<unsigned> next_part{0};
    parallel_for(n, 16384, [&](size_t p0, size_t p1) {
      Box mine = empty_box();
      for (size_t p = p0; p < p1; ++p) {
        Box b = empty_box();
        for (int c = 0; c < 3; ++c) {
          const float* P = B.wverts[B.widx[p * 3 + c]].position;
          for (int k = 0; k < 3; ++k) { b.lo[k] = P[k] < b.lo[k] ? P[k] : b.lo[k]; b.hi[k] = P[k] > b.hi[k] ? P[k] : b.hi[k]; }
        }
        tbox[p] = b;
        grow(mine, b);
      }
      part[next_part.fetch_add(1) % 64u] = mine;
    });
    for (const Box& b : part) if (b.lo[0] <= b.hi[0]) grow(sb, b);
  }
  lap("triangle boxes");
  float diag = sb.hi[0] - sb.lo[0];
  if (sb.hi[1] - sb.lo[1] > diag) diag = sb.hi[1] - sb.lo[1];
  if (sb.hi[2] - sb.lo[2] > diag) diag = sb.hi[2] - sb.lo[2];
  B.ray_eps = 1e-4f * (diag > 1e-6f ? diag : 1e-6f);
  std::vector<uint32_t>& ord = topo->ord;        // position in BVH order → original primitive id (partitioned by the build)
  if (!refit) { ord.resize(n); for (uint32_t p = 0; p < n; ++p) ord[p] = p; }
  // ---- texture sets and material classes ------------------------------------------------------------------------------------
  // A texture SET is a distinct (colour, normal, metal-rough) triple of texture ids among the materials that have a texture, numbered in
  // material order.  The material CLASS travels in the triangle record and the hit word and is what k_shade sorts by (a wave shades 64
  // hits of one class): 0 = untextured Lambert, 1 = untextured GGX, 2..6 = textured, 2 + set % 5; 7 is reserved for environment misses.
  std::vector<int32_t> mat_set, set_tex;     // set of every material (-1: untextured); 3 texture ids per set
  material_sets(mats, mat_set, set_tex);
  auto material_class = [&](int32_t mi) -> uint32_t {
    const HostMaterial& hm = mats[(size_t)mi];
    if (mat_set[(size_t)mi] >= 0) return 2u + (uint32_t)mat_set[(size_t)mi] % 5u;
    return (hm.metallic == 0.0f && hm.roughness >= 1.0f) ? 0u : 1u;
  };
  // triangle record of sorted position i: (v0, prim id) (e1, class) (e2, 0); emitted below in node order
  auto tri_record = [&](uint32_t i, float* o) {
    const uint32_t p = ord[i];
    const float* a = B.wverts[B.widx[p * 3 + 0]].position;
    const float* b = B.wverts[B.widx[p * 3 + 1]].position;
    const float* c = B.wverts[B.widx[p * 3 + 2]].position;
    const uint32_t cls = material_class(B.tri_mat[p]);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; std::memcpy(&o[3], &p, 4);
    o[4] = b[0] - a[0]; o[5] = b[1] - a[1]; o[6] = b[2] - a[2]; std::memcpy(&o[7], &cls, 4);
    o[8] = c[0] - a[0]; o[9] = c[1] - a[1]; o[10] = c[2] - a[2]; o[11] = 0.0f;
  };
  // ---- hierarchy --------------------------------------------------------------------------------------
  // Binary tree by binned surface-area splits (down to single triangles) → 8-wide collapse by dynamic programming over
  // the surface-area cost (Ylitie, Karras, Laine 2017, sec. 3.1): for a binary node n with box half-area A_n and P_n
  // triangles, C(n,i) = least cost of representing its subtree by at most i roots (a root = a leaf or an 8-wide node):
  //   C(n,1) = min(C_leaf, C_int)   C(n,i) = min(C_dist(n,i), C(n,i-1))   C_dist(n,j) = min_k C(left,k) + C(right,j-k)
  //   C_leaf = A_n·P_n·kCostPrim if P_n <= kLeafMax else inf            C_int = C_dist(n,8) + A_n·kCostNode
  // ties: leaf over interior, fewer roots over more, smallest k.  A node's children are the roots of C_dist(n,8), left to
  // right.  The (<= 8) children are then placed into the node's 8 SLOTS so that a
  // child in slot s lies in octant s of the node (bit k of s set = towards +axis k): repeatedly the (child, free
  // slot) pair with the largest  (+-)dx + (+-)dy + (+-)dz,  d = child centre - node centre, ties to the lowest child
  // then the lowest slot.  A ray then orders the slots by its direction signs alone (descending slot ^ octant), with
  // no per-child distance sort.
  //
  // Storage: ONE array of 16-byte units holding 64-byte nodes and 48-byte triangle records; the unit index is the address the
  // kernels use.  A node's origin is kept to 16 bits per axis on a grid over the scene box (org = scene_lo + oq·step, rounded
  // down), child boxes are quantised to 8 bits against that origin:
  //   w0      oq.x | oq.y<<16            w1   oq.z | ex<<16 | ey<<24       w2   ez | imask<<8 | lmask<<16 | two<<24
  //   w3      block: unit address of this node's children block            plane = org + q · 2^(e-127)
  //   w4,5    qlo.x slots 0-3 / 4-7      w6,7  qlo.y     w8,9   qlo.z
  //   w10,11  qhi.x                      w12,13 qhi.y    w14,15 qhi.z               (empty slot: qlo 255, qhi 0)
  // imask / lmask: slots holding an interior / a leaf child; two: leaf slots with 2 triangles.  A children block starts on a
  // 64-byte boundary and holds the interior children (nodes, 4 units each, slot order) followed by the triangles of the leaf
  // children (3 units each, slot order): child node = block + 4·popcount(imask below its slot), triangle = block +
  // 4·popcount(imask) + 3·(number of triangles in lower leaf slots).  The root is the node at unit 0.  64-byte alignment of the
  // nodes is what lets four adjacent lanes fetch one node as one contiguous 64-byte piece (a quarter of the address-processing
  // cost of four unrelated 16-byte gathers: tools/gather_bench.hip).
  // Layout: blocks breadth-first until `toplet_budget` 64-byte records exist (the trace kernels stage that prefix in LDS),
  // then depth-first.
  std::vector<SplitNode>& radix = topo->radix;
  std::vector<Box> radix_box;
  struct Dp { float c[8]; uint8_t leaf1, same[8], k[9]; };   // c[i], same[i]: i = 1..7; k[j]: j = 2..8
  std::vector<Dp> dp;
  auto link_box = [&](int32_t link) { return link < 0 ? tbox[ord[(size_t)~link]] : radix_box[(size_t)link]; };
  auto dp_cost = [&](int32_t link, int i) { return link < 0 ? box_half_area(link_box(link)) * 1.0f * kCostPrim : dp[(size_t)link].c[i]; };
  auto min7 = [](int k) { return k > 7 ? 7 : k; };
  auto assign_slots = [&](const WChild* kid, int n, int* slot_of) {
    Box nb = empty_box();
    for (int i = 0; i < n; ++i) grow(nb, kid[i].box);
    float d[kWide][3];
    for (int i = 0; i < n; ++i)
      for (int k = 0; k < 3; ++k) d[i][k] = 0.5f * (kid[i].box.lo[k] + kid[i].box.hi[k]) - 0.5f * (nb.lo[k] + nb.hi[k]);
    bool child_done[kWide] = {false}, slot_used[kWide] = {false};
    for (int round = 0; round < n; ++round) {
      int bi = -1, bs = -1; float bsc = 0.0f;
      for (int i = 0; i < n; ++i) {
        if (child_done[i]) continue;
        for (int sl = 0; sl < kWide; ++sl) {
          if (slot_used[sl]) continue;
          const float sc = ((sl & 1) ? d[i][0] : -d[i][0]) + ((sl & 2) ? d[i][1] : -d[i][1]) + ((sl & 4) ? d[i][2] : -d[i][2]);
          if (bi < 0 || sc > bsc) { bi = i; bs = sl; bsc = sc; }
        }
      }
      child_done[bi] = true; slot_used[bs] = true; slot_of[bi] = bs;
    }
  };
  // the roots of the best forest of at most i roots below `link`, appended left to right
  struct ForestJob { int32_t link; int i; };
  auto forest = [&](int32_t link0, int i0, WChild* kid, int& n) {
    ForestJob stk[64]; int sp = 0;
    stk[sp++] = {link0, i0};
    while (sp > 0) {
      const ForestJob j = stk[--sp];
      WChild c;
      c.box = link_box(j.link);
      if (j.link < 0) { c.leaf = true; c.lo = c.hi = (uint32_t)~j.link; c.radix = -1; kid[n++] = c; continue; }
      const SplitNode& r = radix[(size_t)j.link]; const Dp& d = dp[(size_t)j.link];
      int i = j.i;
      while (i > 1 && d.same[i]) --i;
      if (i == 1) { c.leaf = d.leaf1 != 0; c.lo = r.lo; c.hi = r.hi; c.radix = j.link; kid[n++] = c; continue; }
      const int kk = d.k[i];
      stk[sp++] = {r.right, min7(i - kk)};      // left first
      stk[sp++] = {r.left, min7(kk)};
    }
  };
  auto make_wide = [&](const WChild* kid, int n) {
    int slot_of[kWide];
    assign_slots(kid, n, slot_of);
    Wide w;
    w.n = n;
    for (int sl = 0; sl < kWide; ++sl) w.used[sl] = false;
    for (int i = 0; i < n; ++i) { w.slot[slot_of[i]] = kid[i]; w.used[slot_of[i]] = true; }
    return w;
  };
  auto expand = [&](int32_t r) {
    WChild kid[kWide];
    int n = 0;
    const int kk = dp[(size_t)r].k[8];
    forest(radix[(size_t)r].left, min7(kk), kid, n);
    forest(radix[(size_t)r].right, min7(8 - kk), kid, n);
    return make_wide(kid, n);
  };
  // 8-bit quantisation of the children's [lo,hi] on one axis against the node's [org, nhi]: scale 2^(e-127)
  // is the smallest power of two with (nhi-org)/scale <= 255; lower planes floor, upper planes ceil, each
  // nudged until the float expression org + q*scale brackets the exact plane.
  auto pow2_biased = [](uint32_t e) { const uint32_t u = e << 23; float f; std::memcpy(&f, &u, 4); return f; };
  auto quantize_axis = [&](const float* clo, const float* chi, int nk, float org, float nhi, uint32_t& e_out, uint32_t* qlo, uint32_t* qhi) {
    const float f = (nhi - org) / 255.0f;
    uint32_t u; std::memcpy(&u, &f, 4);
    uint32_t e = (u >> 23) & 255u;
    if (u & 0x007fffffu) e += 1u;
    if (e < 1u) e = 1u;
    for (;; ++e) {
      const float sc = pow2_biased(e);
      bool ok = true;
      for (int i = 0; i < nk && ok; ++i) {
        float fl = std::floor((clo[i] - org) / sc);
        if (fl < 0.0f) fl = 0.0f;
        if (fl > 255.0f) fl = 255.0f;
        int q = (int)fl;
        while (q > 0 && org + (float)q * sc > clo[i]) --q;
        qlo[i] = (uint32_t)q;
        float ce = std::ceil((chi[i] - org) / sc);
        if (ce < 0.0f) ce = 0.0f;
        if (ce > 255.0f) { ok = false; break; }
        int q2 = (int)ce;
        while (q2 < 255 && org + (float)q2 * sc < chi[i]) ++q2;
        if (org + (float)q2 * sc < chi[i]) { ok = false; break; }
        qhi[i] = (uint32_t)q2;
      }
      if (ok) break;
    }
    e_out = e;
  };
  std::vector<Slot>& order = topo->order;              // node index → radix node (or -1 for the single-triangle special case)
  std::vector<Wide>& wide = topo->wide;                // node index → its children
  std::vector<uint32_t>& child_base = topo->child_base; // node index → index of its first interior child
  std::vector<uint32_t>& block_order = topo->block_order; // node indices in the order their children blocks were allocated
  // boxes of all radix nodes, bottom-up (iterative post-order)
  // radix nodes are numbered parents-first, so descending index order is a valid post-order; the subtrees the build handed to the pool are disjoint index
  // ranges behind the top's, so they go first, in parallel, and the top last
  auto bottom_up = [&](auto&& visit) {
    const std::vector<uint32_t>& sf = topo->sub_first;
    size_t top_end = radix.size();
    if (sf.size() >= 2 && sf.back() == radix.size()) {
      parallel_for(sf.size() - 1, 1, [&](size_t t0, size_t t1) {
        for (size_t t = t0; t < t1; ++t)
          for (size_t idx = sf[t + 1]; idx-- > sf[t];) visit(idx);
      });
      top_end = sf[0];
    }
    for (size_t idx = top_end; idx-- > 0;) visit(idx);
  };
  auto compute_radix_boxes = [&]() {
    radix_box.assign(radix.size(), empty_box());
    bottom_up([&](size_t idx) {
      const SplitNode& r = radix[idx];
      Box b = r.left < 0 ? tbox[ord[(size_t)~r.left]] : radix_box[(size_t)r.left];
      grow(b, r.right < 0 ? tbox[ord[(size_t)~r.right]] : radix_box[(size_t)r.right]);
      radix_box[idx] = b;
    });
  };
  if (refit) {    // same tree, same slots: only the boxes of the children follow the moved triangles
    compute_radix_boxes();
    parallel_for(wide.size(), 2048, [&](size_t i0, size_t i1) {
      for (size_t i = i0; i < i1; ++i)
        for (int sl = 0; sl < kWide; ++sl) {
          if (!wide[i].used[sl]) continue;
          WChild& ch = wide[i].slot[sl];
          ch.box = ch.radix >= 0 ? radix_box[(size_t)ch.radix] : tbox[ord[ch.lo]];
        }
    });
  } else if (n == 1) {   // a single triangle: two identical leaf children (mirrors the binary special case)
    WChild kid[2];
    for (int k = 0; k < 2; ++k) { kid[k].leaf = true; kid[k].lo = kid[k].hi = 0; kid[k].radix = -1; kid[k].box = tbox[0]; }
    const Wide w = make_wide(kid, 2);
    order.push_back({-1, 0}); wide.push_back(w); child_base.push_back(1); block_order.push_back(0);
  } else {
    build_split_tree(tbox, ord, radix, bvh_builder == 1, topo->sub_first);
    lap("  binary tree");
    compute_radix_boxes();
    lap("  binary boxes");
    // cost tables, bottom-up: radix nodes are numbered parents-first, so descending index order is a valid post-order
    dp.assign(radix.size(), Dp());
    bottom_up([&](size_t idx) {
      const SplitNode& r = radix[idx];
      Dp& d = dp[idx];
      float dist[9];
      for (int j = 2; j <= 8; ++j) {
        float best = std::numeric_limits<float>::infinity(); int bk = 1;
        for (int k = 1; k < j; ++k) {
          const float v = dp_cost(r.left, min7(k)) + dp_cost(r.right, min7(j - k));
          if (v < best) { best = v; bk = k; }
        }
        dist[j] = best; d.k[j] = (uint8_t)bk;
      }
      const uint32_t P = r.hi - r.lo + 1u;
      const float area = box_half_area(radix_box[idx]);
      const float cleaf = P <= (uint32_t)kLeafMax ? area * (float)P * kCostPrim : std::numeric_limits<float>::infinity();
      const float cint = dist[8] + area * kCostNode;
      d.leaf1 = cleaf <= cint; d.c[1] = d.leaf1 ? cleaf : cint;
      for (int i = 2; i <= 7; ++i) {
        if (dist[i] < d.c[i - 1]) { d.c[i] = dist[i]; d.same[i] = 0; } else { d.c[i] = d.c[i - 1]; d.same[i] = 1; }
      }
    });
    lap("  collapse cost tables");
    auto number = [&](int32_t r, uint32_t depth) {
      order.push_back({r, depth});
      wide.push_back(expand(r));
      child_base.push_back(0xffffffffu);
    };
    // allocate the interior children of node `idx` as one consecutive block, in slot order
    auto alloc_children = [&](uint32_t idx) {
      child_base[idx] = (uint32_t)order.size();
      block_order.push_back(idx);
      const Wide w = wide[idx];
      for (int sl = 0; sl < kWide; ++sl)
        if (w.used[sl] && !w.slot[sl].leaf) number(w.slot[sl].radix, order[idx].depth + 1);
    };
    number(0, 0);
    size_t head = 0;
    for (; head < order.size() && order.size() < toplet_budget; ++head) alloc_children((uint32_t)head);   // breadth-first top
    {
      std::vector<uint32_t> stack;                                                   // depth-first remainder
      const size_t n_top = order.size();
      for (size_t i = head; i < n_top; ++i) {
        stack.push_back((uint32_t)i);
        while (!stack.empty()) {
          const uint32_t idx = stack.back();
          stack.pop_back();
          alloc_children(idx);
          int k = 0;
          for (int sl = 0; sl < kWide; ++sl) k += (wide[idx].used[sl] && !wide[idx].slot[sl].leaf) ? 1 : 0;
          for (int j = k - 1; j >= 0; --j) stack.push_back(child_base[idx] + (uint32_t)j);   // first child on top
        }
      }
    }
  }
  lap(refit ? "refit boxes" : "tree build + collapse");
  // ---- unit addresses: root at 0, then the children blocks in allocation order, each on a 64-byte boundary ---------------
  B.n_nodes = (uint32_t)order.size();
  std::vector<uint32_t>& node_addr = topo->node_addr;
  std::vector<uint32_t>& block_addr = topo->block_addr;
  uint64_t& next_unit = topo->next_unit;
  uint32_t& n_tri_records = topo->n_tri_records;
  if (!refit) { node_addr.assign(B.n_nodes, 0u); block_addr.assign(B.n_nodes, 0u); next_unit = 4; n_tri_records = 0; }
  for (const uint32_t idx : block_order) {
    if (refit) break;
    const Wide& w = wide[idx];
    uint32_t ni = 0, nt = 0;
    for (int sl = 0; sl < kWide; ++sl) {
      if (!w.used[sl]) continue;
      if (!w.slot[sl].leaf) ++ni; else nt += w.slot[sl].hi - w.slot[sl].lo + 1u;
    }
    block_addr[idx] = (uint32_t)next_unit;
    for (uint32_t i = 0; i < ni; ++i) node_addr[child_base[idx] + i] = (uint32_t)next_unit + 4u * i;
    next_unit += ((uint64_t)4 * ni + (uint64_t)3 * nt + 3u) & ~(uint64_t)3;
    n_tri_records += nt;
    if (next_unit >= (1ull << 31)) return "scene_commit: BVH too large";
  }
  // ---- emit nodes and triangles -----------------------------------------------------------------------------------------
  if (!refit) B.recs.assign((size_t)next_unit * 4, 0.0f);      // a refit rewrites every node and triangle record in place; the padding between blocks stays zero
  B.n_units = (uint32_t)next_unit;
  float grid_step[3];
  for (int k = 0; k < 3; ++k) { const float st = (sb.hi[k] - sb.lo[k]) / 65535.0f; grid_step[k] = st > 0.0f ? st : 1.0f; B.grid_lo[k] = sb.lo[k]; B.grid_step[k] = grid_step[k]; }
  uint32_t maxd = 0;
  for (uint32_t idx = 0; idx < B.n_nodes; ++idx) if (order[idx].depth > maxd) maxd = order[idx].depth;
  // surface-area cost of the tree (ptc_stats.bvh_sa_cost): per used child slot half_area(child) / half_area(scene box at build time), a two-triangle leaf twice, each term
  // truncated to 2^-20 and summed as an integer — the terms and the sum k_refit_nodes forms on the device (pt_refit.hip)
  std::atomic<uint64_t> sa_cost{0};
  if (!refit) B.sa_unit = box_half_area(sb);
  const float scene_area = B.sa_unit;
  parallel_for(B.n_nodes, 1024, [&](size_t n0, size_t n1) {
  uint64_t cost_part = 0;
  for (uint32_t idx = (uint32_t)n0; idx < (uint32_t)n1; ++idx) {
    const Wide& w = wide[idx];
    if (scene_area > 0.0f)
      for (int sl = 0; sl < kWide; ++sl) {
        if (!w.used[sl]) continue;
        const uint64_t term = (uint64_t)((box_half_area(w.slot[sl].box) / scene_area) * PTC_SA_COST_ONE);
        cost_part += (w.slot[sl].leaf && w.slot[sl].hi - w.slot[sl].lo + 1u == 2u) ? 2u * term : term;
      }
    uint32_t word[kNodeWords] = {0};
    uint32_t e[3], oq[3], qlo[3][kWide], qhi[3][kWide];
    int sl_of[kWide], nk = 0;                      // used slots in ascending order
    for (int sl = 0; sl < kWide; ++sl) if (w.used[sl]) sl_of[nk++] = sl;
    for (int k = 0; k < 3; ++k) {
      float clo[kWide], chi[kWide], nlo = w.slot[sl_of[0]].box.lo[k], nhi = w.slot[sl_of[0]].box.hi[k];
      uint32_t ql[kWide], qh[kWide];
      for (int i = 0; i < nk; ++i) {
        clo[i] = w.slot[sl_of[i]].box.lo[k]; chi[i] = w.slot[sl_of[i]].box.hi[k];
        nlo = clo[i] < nlo ? clo[i] : nlo; nhi = chi[i] > nhi ? chi[i] : nhi;
      }
      // origin on the 16-bit scene grid, rounded down: the largest q with fmaf(q, step, lo) <= the node's lower bound
      float fq = std::floor((nlo - sb.lo[k]) / grid_step[k]);
      if (fq < 0.0f) fq = 0.0f;
      if (fq > 65535.0f) fq = 65535.0f;
      uint32_t q16 = (uint32_t)fq;
      while (q16 > 0u && fmaf((float)q16, grid_step[k], sb.lo[k]) > nlo) --q16;
      oq[k] = q16;
      const float org = fmaf((float)q16, grid_step[k], sb.lo[k]);
      for (int sl = 0; sl < kWide; ++sl) { qlo[k][sl] = 255; qhi[k][sl] = 0; }
      quantize_axis(clo, chi, nk, org, nhi, e[k], ql, qh);
      for (int i = 0; i < nk; ++i) { qlo[k][sl_of[i]] = ql[i]; qhi[k][sl_of[i]] = qh[i]; }
    }
    uint32_t imask = 0, lmask = 0, two = 0, ni = 0;
    for (int sl = 0; sl < kWide; ++sl) if (w.used[sl] && !w.slot[sl].leaf) { imask |= 1u << sl; ++ni; }
    float* tri_out = &B.recs[((size_t)block_addr[idx] + 4u * ni) * 4];
    for (int sl = 0; sl < kWide; ++sl) {
      if (!w.used[sl] || !w.slot[sl].leaf) continue;
      const WChild& ch = w.slot[sl];
      lmask |= 1u << sl;
      if (ch.hi - ch.lo + 1u == 2u) two |= 1u << sl;
      for (uint32_t t = ch.lo; t <= ch.hi; ++t) { tri_record(t, tri_out); tri_out += 12; }
    }
    word[0] = oq[0] | (oq[1] << 16);
    word[1] = oq[2] | (e[0] << 16) | (e[1] << 24);
    word[2] = e[2] | (imask << 8) | (lmask << 16) | (two << 24);
    word[3] = block_addr[idx];
    auto pack4 = [](const uint32_t* q) { return q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24); };
    for (int k = 0; k < 3; ++k) {
      word[4 + 2 * k] = pack4(&qlo[k][0]); word[5 + 2 * k] = pack4(&qlo[k][4]);
      word[10 + 2 * k] = pack4(&qhi[k][0]); word[11 + 2 * k] = pack4(&qhi[k][4]);
    }
    std::memcpy(&B.recs[(size_t)node_addr[idx] * 4], word, sizeof word);
  }
  sa_cost.fetch_add(cost_part);
  });
  B.sa_cost_fixed = sa_cost.load();
  lap("emit nodes + triangles");
  B.max_depth = maxd;
  B.n_tri_records = n_tri_records;
  B.n_lds_units = B.n_units < toplet_budget * 4u ? B.n_units : toplet_budget * 4u;
  // ---- materials --------------------------------------------------------------------------------------
  fill_materials(mats, mat_set, B);
  // ---- emitters (original primitive order), power pmf / cdf ----------------------------------------------
  B.prim_light.assign(n, -1);
  B.lights.clear(); B.cdf.clear();
  std::vector<float> weight;
  for (uint32_t p = 0; p < n; ++p) {
    const HostMaterial& m = mats[(size_t)B.tri_mat[p]];
    if (!(m.emissive[0] > 0.0f || m.emissive[1] > 0.0f || m.emissive[2] > 0.0f)) continue;
    const float* a = B.wverts[B.widx[p * 3 + 0]].position;
    const float* b = B.wverts[B.widx[p * 3 + 1]].position;
    const float* c = B.wverts[B.widx[p * 3 + 2]].position;
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    float cr[3];
    fcross(e1, e2, cr);
    const float len = sqrtf(fdot(cr, cr));
    const float area = 0.5f * len;
    const float lum = fmaf(m.emissive[2], 0.0722f, fmaf(m.emissive[1], 0.7152f, m.emissive[0] * 0.2126f));
    const float wgt = area * lum;
    if (!(wgt > 0.0f)) continue;
    const float il = 1.0f / len;
    B.prim_light[p] = (int32_t)weight.size();
    weight.push_back(wgt);
    const float rec[20] = {a[0], a[1], a[2], area, e1[0], e1[1], e1[2], 0.0f /*pmf*/, e2[0], e2[1], e2[2], 0.0f,
                           cr[0] * il, cr[1] * il, cr[2] * il, 0.0f, m.emissive[0], m.emissive[1], m.emissive[2], 0.0f};
    B.lights.insert(B.lights.end(), rec, rec + 20);
  }
  B.n_lights = (uint32_t)weight.size();
  float total = 0.0f;
  for (float wv : weight) total += wv;
  float run = 0.0f;
  B.cdf.resize(weight.size());
  for (size_t i = 0; i < weight.size(); ++i) {
    run += weight[i];
    B.cdf[i] = run / total;
    B.lights[i * 20 + 7] = weight[i] / total;
  }
  lap("materials + emitters");
  // ---- per-primitive shading records: positions + normals of the three vertices, material, emitter index; with a textured material in
  // the scene also uv / tangent / bitangent of the three vertices (units 5..10) and a unit of padding
  bool any_tex = false;
  for (const auto& m : mats) any_tex = any_tex || m.tex_color >= 0 || m.tex_normal >= 0 || m.tex_mr >= 0;
  B.shade_stride = any_tex ? 12u : 5u;
  const size_t fl = (size_t)B.shade_stride * 4;
  B.shade.assign((size_t)n * fl, 0.0f);
  parallel_for(n, 8192, [&](size_t p0, size_t p1) {
  for (uint32_t p = (uint32_t)p0; p < (uint32_t)p1; ++p) {
    const HostVertex& a = B.wverts[B.widx[p * 3 + 0]];
    const HostVertex& b = B.wverts[B.widx[p * 3 + 1]];
    const HostVertex& c = B.wverts[B.widx[p * 3 + 2]];
    float* o = &B.shade[(size_t)p * fl];
    o[0] = a.position[0]; o[1] = a.position[1]; o[2] = a.position[2]; std::memcpy(&o[3], &B.tri_mat[p], 4);
    o[4] = b.position[0]; o[5] = b.position[1]; o[6] = b.position[2]; std::memcpy(&o[7], &B.prim_light[p], 4);
    o[8] = c.position[0]; o[9] = c.position[1]; o[10] = c.position[2]; o[11] = a.normal[0];
    o[12] = a.normal[1]; o[13] = a.normal[2]; o[14] = b.normal[0]; o[15] = b.normal[1];
    o[16] = b.normal[2]; o[17] = c.normal[0]; o[18] = c.normal[1]; o[19] = c.normal[2];
    if (any_tex) {
      float* t = o + 20;
      for (int k = 0; k < 3; ++k) {
        const uint32_t vi = B.widx[p * 3 + (uint32_t)k];
        const HostVertex& v = B.wverts[vi];
        t[k * 2 + 0] = v.texcoord[0]; t[k * 2 + 1] = v.texcoord[1];
        for (int j = 0; j < 3; ++j) { t[6 + k * 3 + j] = v.tangent[j]; t[15 + k * 3 + j] = topo->wbt[(size_t)vi * 3 + (size_t)j]; }
      }
    }
  }
  });
  lap("shading records");
  if (!refit) fill_textures_env(texs, env, set_tex, B);
  if (!B.cdf.empty()) B.cdf.back() = 1.0f;
  if (B.cdf.empty()) B.cdf.push_back(1.0f);
  if (B.lights.empty()) B.lights.assign(20, 0.0f);
  lap("textures + environment");
  B.topology = topo;
  return std::string();
}
}  // namespace

std::string ptc_build_scene(const std::vector<HostMaterial>& mats, const std::vector<HostMesh>& meshes,
                            const std::vector<HostInstance>& insts, const std::vector<HostTexture>& texs, const HostEnv& env,
                            uint32_t toplet_budget, int bvh_builder, HostBuilt& B) {
  return build_or_refit(mats, meshes, insts, texs, env, toplet_budget, bvh_builder, B, false);
}

namespace {
// The emitter table from the emissive primitives alone (emit_prims: prim, instance, its 3 vertex indices inside the instance's mesh), each vertex taken through its instance's
// matrix with the flatten's expression.  committed != nullptr: a refit — false when the set of emitters is not the committed one.  committed == nullptr: a commit without a host
// flatten (ptc_build_skeleton) — the emitter index of every primitive is written to *prim_light (sized by the caller, -1 everywhere).
bool emitters_from_prims(const std::vector<HostMaterial>& mats, const std::vector<HostMesh>& meshes, const std::vector<HostInstance>& insts, const std::vector<int32_t>& emit_prims,
                         const HostBuilt* committed, std::vector<int32_t>* prim_light, std::vector<float>& lights, std::vector<float>& cdf);
}
// The host's share of a commit whose tree is built ON THE DEVICE (ptc_scene_commit with the LBVH builder on a device context): everything that does not need a flattened
// vertex — world vertex indices and material per primitive, materials, texture sets, environment tables, and the emitter table (the emissive primitives alone, each vertex
// through its instance's matrix with the flatten's expression).  The flatten, the shading records and the tree are the device's (pt_refit.hip, pt_build.hip): `out` keeps
// their SIZES, its wverts / shade / recs come back from HBM when somebody asks, and it has no topology (as after ptc_scene_rebuild).
std::string ptc_build_skeleton(const std::vector<HostMaterial>& mats, const std::vector<HostMesh>& meshes, const std::vector<HostInstance>& insts,
                               const std::vector<HostTexture>& texs, const HostEnv& env, uint32_t toplet_budget, HostBuilt& B) {
  if (insts.empty()) return "scene_commit: no instances";
  uint64_t nv = 0, nt = 0;
  for (const auto& in : insts) { nv += meshes[(size_t)in.mesh].v.size(); nt += meshes[(size_t)in.mesh].idx.size() / 3; }
  if (nt >= (1u << 28)) return "scene_commit: too many triangles";
  B = HostBuilt();
  B.n_wverts = (uint32_t)nv;       // the flatten is the device's: refresh_host_copy sizes and fills wverts when somebody asks
  B.widx.resize(nt * 3);
  B.tri_mat.resize(nt);
  B.n_tris = (uint32_t)nt;
  std::vector<int32_t> emit_prims;
  {
    uint32_t vb = 0, tb = 0;
    for (size_t i = 0; i < insts.size(); ++i) {
      const HostMesh& m = meshes[(size_t)insts[i].mesh];
      const size_t ntm = m.idx.size() / 3;
      for (size_t k = 0; k < ntm; ++k) {
        for (int c = 0; c < 3; ++c) B.widx[(tb + k) * 3 + c] = vb + m.idx[k * 3 + c];
        B.tri_mat[tb + k] = m.material;
      }
      const HostMaterial& hm = mats[(size_t)m.material];
      if (hm.emissive[0] > 0.0f || hm.emissive[1] > 0.0f || hm.emissive[2] > 0.0f)
        for (size_t k = 0; k < ntm; ++k) {
          const int32_t e[5] = {(int32_t)(tb + k), (int32_t)i, (int32_t)m.idx[k * 3], (int32_t)m.idx[k * 3 + 1], (int32_t)m.idx[k * 3 + 2]};
          emit_prims.insert(emit_prims.end(), e, e + 5);
        }
      vb += (uint32_t)m.v.size(); tb += (uint32_t)ntm;
    }
  }
  std::vector<int32_t> mat_set, set_tex;
  material_sets(mats, mat_set, set_tex);
  fill_materials(mats, mat_set, B);
  B.prim_light.assign(nt, -1);
  (void)emitters_from_prims(mats, meshes, insts, emit_prims, nullptr, &B.prim_light, B.lights, B.cdf);
  B.n_lights = 0;
  for (int32_t l : B.prim_light) B.n_lights += l >= 0 ? 1u : 0u;
  bool any_tex = false;
  for (const auto& m : mats) any_tex = any_tex || m.tex_color >= 0 || m.tex_normal >= 0 || m.tex_mr >= 0;
  B.shade_stride = any_tex ? 12u : 5u;
  fill_textures_env(texs, env, set_tex, B);
  B.n_lds_units = toplet_budget * 4u;      // upper bound until the tree exists (it sizes the trace kernels' LDS)
  return std::string();
}

// Refit: the instances' transforms changed (and nothing else).  Vertices are flattened again, every box of the committed tree is
// recomputed bottom-up and re-quantised (the scene box, hence the origin grid and the ray offset, follow the geometry), triangle records,
// shading records and the emitter table are rewritten; the tree keeps its topology, its slots and its layout.
std::string ptc_refit_scene(const std::vector<HostMaterial>& mats, const std::vector<HostMesh>& meshes, const std::vector<HostInstance>& insts,
                            const std::vector<HostTexture>& texs, const HostEnv& env, HostBuilt& B) {
  return build_or_refit(mats, meshes, insts, texs, env, 0, 0, B, true);
}

// ---- the host's share of a refit on the device (pt_refit.h) -----------------------------------------------------------------------
void ptc_prim_classes(const std::vector<HostMaterial>& mats, const std::vector<int32_t>& tri_mat, std::vector<uint32_t>& out) {
  // as build_or_refit numbers them: a texture SET is a distinct (colour, normal, metal-rough) triple among the textured materials, in material order
  std::vector<int32_t> mat_set(mats.size(), -1), set_tex;
  for (size_t i = 0; i < mats.size(); ++i) {
    const HostMaterial& m = mats[i];
    if (m.tex_color < 0 && m.tex_normal < 0 && m.tex_mr < 0) continue;
    int32_t found = -1;
    for (size_t k = 0; k * 3 < set_tex.size(); ++k)
      if (set_tex[k * 3] == m.tex_color && set_tex[k * 3 + 1] == m.tex_normal && set_tex[k * 3 + 2] == m.tex_mr) { found = (int32_t)k; break; }
    if (found < 0) { found = (int32_t)(set_tex.size() / 3); set_tex.push_back(m.tex_color); set_tex.push_back(m.tex_normal); set_tex.push_back(m.tex_mr); }
    mat_set[i] = found;
  }
  out.resize(tri_mat.size());
  for (size_t p = 0; p < tri_mat.size(); ++p) {
    const size_t mi = (size_t)tri_mat[p];
    const HostMaterial& hm = mats[mi];
    out[p] = mat_set[mi] >= 0 ? 2u + (uint32_t)mat_set[mi] % 5u : ((hm.metallic == 0.0f && hm.roughness >= 1.0f) ? 0u : 1u);
  }
}

void ptc_refit_plan(const std::vector<HostMaterial>& mats, const std::vector<HostMesh>& meshes, const std::vector<HostInstance>& insts, const HostBuilt& B, RefitPlan& P) {
  P = RefitPlan();
  std::vector<uint32_t> mesh_first(meshes.size());
  for (size_t m = 0; m < meshes.size(); ++m) { mesh_first[m] = (uint32_t)P.mesh_verts.size(); P.mesh_verts.insert(P.mesh_verts.end(), meshes[m].v.begin(), meshes[m].v.end()); }
  uint32_t vb = 0, tb = 0;
  for (size_t i = 0; i < insts.size(); ++i) {
    const HostMesh& m = meshes[(size_t)insts[i].mesh];
    P.inst_first.push_back(vb);
    P.inst_src.push_back(mesh_first[(size_t)insts[i].mesh]);
    P.vert_inst.insert(P.vert_inst.end(), m.v.size(), (uint32_t)i);
    const HostMaterial& hm = mats[(size_t)m.material];
    if (hm.emissive[0] > 0.0f || hm.emissive[1] > 0.0f || hm.emissive[2] > 0.0f)
      for (size_t k = 0; k * 3 < m.idx.size(); ++k) {
        const int32_t e[5] = {(int32_t)(tb + k), (int32_t)i, (int32_t)m.idx[k * 3], (int32_t)m.idx[k * 3 + 1], (int32_t)m.idx[k * 3 + 2]};
        P.emit_prims.insert(P.emit_prims.end(), e, e + 5);
      }
    vb += (uint32_t)m.v.size(); tb += (uint32_t)(m.idx.size() / 3);
  }
  P.n_verts = vb; P.n_tris = tb;
  if (!B.topology) { P.level_first.assign(1, 0u); return; }      // a commit on the device (ptc_build_skeleton): the level lists come from the device build
  const Topology& topo = *std::static_pointer_cast<Topology>(B.topology);
  uint32_t maxd = 0;
  for (const Slot& s : topo.order) maxd = s.depth > maxd ? s.depth : maxd;
  std::vector<uint32_t> count(maxd + 2, 0u);
  for (const Slot& s : topo.order) ++count[maxd - s.depth + 1];            // level 0 = the deepest nodes
  for (size_t l = 1; l < count.size(); ++l) count[l] += count[l - 1];
  P.level_first = count;
  P.level_nodes.resize(topo.order.size());
  std::vector<uint32_t> at(count.begin(), count.end() - 1);
  for (size_t idx = 0; idx < topo.order.size(); ++idx) P.level_nodes[at[maxd - topo.order[idx].depth]++] = topo.node_addr[idx];
}

bool ptc_refit_instance_transforms(const std::vector<HostInstance>& insts, std::vector<float>& out) {
  out.resize(insts.size() * 21);
  bool finite = true;
  for (size_t i = 0; i < insts.size(); ++i) {
    const Mat34 M = from_matrix(insts[i].m);
    float* o = &out[i * 21];
    for (int c = 0; c < 4; ++c)
      for (int r = 0; r < 3; ++r) { o[c * 3 + r] = M.m[c * 4 + r]; finite = finite && std::isfinite(M.m[c * 4 + r]); }
    for (int k = 0; k < 9; ++k) o[12 + k] = M.n[k];
  }
  return finite;
}

void ptc_refit_grid(const float lo[3], const float hi[3], float grid_lo[3], float grid_step[3], float* ray_eps) {
  float diag = hi[0] - lo[0];
  if (hi[1] - lo[1] > diag) diag = hi[1] - lo[1];
  if (hi[2] - lo[2] > diag) diag = hi[2] - lo[2];
  *ray_eps = 1e-4f * (diag > 1e-6f ? diag : 1e-6f);
  for (int k = 0; k < 3; ++k) { const float st = (hi[k] - lo[k]) / 65535.0f; grid_step[k] = st > 0.0f ? st : 1.0f; grid_lo[k] = lo[k]; }
}

bool ptc_refit_emitters(const std::vector<HostMaterial>& mats, const std::vector<HostMesh>& meshes, const std::vector<HostInstance>& insts, const RefitPlan& P,
                        const HostBuilt& B, std::vector<float>& lights, std::vector<float>& cdf) {
  return emitters_from_prims(mats, meshes, insts, P.emit_prims, &B, nullptr, lights, cdf);
}
namespace {
bool emitters_from_prims(const std::vector<HostMaterial>& mats, const std::vector<HostMesh>& meshes, const std::vector<HostInstance>& insts, const std::vector<int32_t>& emit_prims,
                         const HostBuilt* committed, std::vector<int32_t>* prim_light, std::vector<float>& lights, std::vector<float>& cdf) {
  lights.clear(); cdf.clear();
  std::vector<float> weight;
  int32_t cached_inst = -1;
  Mat34 M{};
  for (size_t j = 0; j * 5 < emit_prims.size(); ++j) {
    const int32_t* e = &emit_prims[j * 5];
    const uint32_t p = (uint32_t)e[0];
    const HostMesh& mesh = meshes[(size_t)insts[(size_t)e[1]].mesh];
    const HostMaterial& m = mats[(size_t)mesh.material];
    if (e[1] != cached_inst) { M = from_matrix(insts[(size_t)e[1]].m); cached_inst = e[1]; }
    float w[3][3];
    for (int c = 0; c < 3; ++c) {
      const float* s = mesh.v[(size_t)e[2 + c]].position;
      for (int r = 0; r < 3; ++r) w[c][r] = M.m[0 + r] * s[0] + M.m[4 + r] * s[1] + M.m[8 + r] * s[2] + M.m[12 + r];
    }
    const float *a = w[0], *b = w[1], *c = w[2];
    const float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    float cr[3];
    fcross(e1, e2, cr);
    const float len = sqrtf(fdot(cr, cr));
    const float area = 0.5f * len;
    const float lum = fmaf(m.emissive[2], 0.0722f, fmaf(m.emissive[1], 0.7152f, m.emissive[0] * 0.2126f));
    const float wgt = area * lum;
    const bool is_light = wgt > 0.0f;
    if (committed && is_light != (committed->prim_light[p] >= 0)) return false;
    if (!is_light) continue;
    if (committed && committed->prim_light[p] != (int32_t)weight.size()) return false;
    if (prim_light) (*prim_light)[p] = (int32_t)weight.size();
    const float il = 1.0f / len;
    weight.push_back(wgt);
    const float rec[20] = {a[0], a[1], a[2], area, e1[0], e1[1], e1[2], 0.0f, e2[0], e2[1], e2[2], 0.0f,
                           cr[0] * il, cr[1] * il, cr[2] * il, 0.0f, m.emissive[0], m.emissive[1], m.emissive[2], 0.0f};
    lights.insert(lights.end(), rec, rec + 20);
  }
  if (committed && weight.size() != committed->n_lights) return false;
  float total = 0.0f;
  for (float wv : weight) total += wv;
  float run = 0.0f;
  cdf.resize(weight.size());
  for (size_t i = 0; i < weight.size(); ++i) {
    run += weight[i];
    cdf[i] = run / total;
    lights[i * 20 + 7] = weight[i] / total;
  }
  if (!cdf.empty()) cdf.back() = 1.0f;
  if (cdf.empty()) cdf.push_back(1.0f);
  if (lights.empty()) lights.assign(20, 0.0f);
  return true;
}
}  // namespace
