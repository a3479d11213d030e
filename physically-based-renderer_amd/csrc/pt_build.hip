// pt_build.hip — the LBVH build on the device (see pt_build.h; the definition of every value is the host build in ptc_scene.cpp, bvh_builder = LBVH).
//
// Everything here is integer / byte work with a few float comparisons per node: HBM- and latency-bound, nothing for MFMA.  The arithmetic that decides
// the tree — centres, Morton quantisation, box unions, half areas, the collapse-cost sums, the slot scores — is written with the host's expressions in
// the host's order (-ffp-contract=off on both sides), because the contract is byte equality of the result, not a tree "as good as" the host's.
#include "pt_build.h"

#include <string>
#include <vector>

namespace {
constexpr int kBlock = 256;
constexpr int kWideB = 8;
constexpr uint32_t kLeafMaxB = 2;
constexpr uint32_t kEmptyLink = 0x80000000u;
constexpr uint32_t kUnset = 0xffffffffu;
constexpr uint32_t kSortTile = 512;       // keys per wave of a sort pass: 8 steps of 64, in order (the sort is stable)

struct Box6 { float lo[3], hi[3]; };
struct DpT { uint32_t bits; float c[7]; };   // c[i - 1] = C(n, i), i = 1..7; bits: 0 leaf1, i (2..7) same[i], 8 + 3 (j - 2) .. : k[j], j = 2..8

struct Bld {
  const HostVertex* wverts; const uint32_t* widx; const uint32_t* prim_cls; uint32_t n, budget;
  Box6* tbox; uint32_t* cb;
  unsigned long long *key_a, *key_b; uint32_t *val_a, *val_b; uint32_t* hist; uint32_t* digit_total;
  int32_t *r_left, *r_right; uint32_t *r_lo, *r_hi, *r_parent, *leaf_parent, *r_flag;
  Box6* r_box; DpT* dp;
  int32_t* w_radix; uint32_t* w_parent; int32_t* w_link; uint32_t* w_child; uint32_t *w_own, *w_sub, *w_baddr, *w_naddr;
  uint32_t* counters;    // [0] 8-wide nodes so far, [1] units of the array, [2] entries of top_order
  uint32_t* top_order;
};

__device__ inline uint32_t ordered_u(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ inline float unordered_f(uint32_t enc) { return __uint_as_float((enc & 0x80000000u) ? (enc ^ 0x80000000u) : ~enc); }
__device__ inline void grow6(Box6& b, const Box6& o) {
  for (int k = 0; k < 3; ++k) { b.lo[k] = o.lo[k] < b.lo[k] ? o.lo[k] : b.lo[k]; b.hi[k] = o.hi[k] > b.hi[k] ? o.hi[k] : b.hi[k]; }
}
__device__ inline float half_area6(const Box6& b) {
  const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
  return ex * ey + ey * ez + ez * ex;
}

__global__ void k_bld_init(Bld b) {
  const uint32_t i = threadIdx.x;
  if (i < 6) b.cb[i] = i < 3 ? 0xffffffffu : 0u;
  if (i == 0) { b.counters[0] = 1u; b.counters[1] = 0u; b.counters[2] = 0u; b.w_radix[0] = 0; b.w_parent[0] = kUnset; }
}

// ---- triangle boxes (the host's tbox: < and > from +-inf over the vertices in order) and the bounds of their centres -------------------------
__global__ __launch_bounds__(kBlock) void k_bld_prims(Bld b) {
  float cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < b.n; p += gridDim.x * kBlock) {
    Box6 t;
    for (int k = 0; k < 3; ++k) { t.lo[k] = INFINITY; t.hi[k] = -INFINITY; }
    for (int c = 0; c < 3; ++c) {
      const float* P = b.wverts[b.widx[(size_t)p * 3 + c]].position;
      for (int k = 0; k < 3; ++k) { t.lo[k] = P[k] < t.lo[k] ? P[k] : t.lo[k]; t.hi[k] = P[k] > t.hi[k] ? P[k] : t.hi[k]; }
    }
    b.tbox[p] = t;
    for (int k = 0; k < 3; ++k) { const float c = 0.5f * (t.lo[k] + t.hi[k]); cl[k] = fminf(cl[k], c); ch[k] = fmaxf(ch[k], c); }
  }
  __shared__ float red[6][kBlock / 64];
  for (int k = 0; k < 3; ++k) {
    for (int d = 32; d >= 1; d >>= 1) { cl[k] = fminf(cl[k], __shfl_xor(cl[k], d)); ch[k] = fmaxf(ch[k], __shfl_xor(ch[k], d)); }
    if ((threadIdx.x & 63) == 0) { red[k][threadIdx.x >> 6] = cl[k]; red[3 + k][threadIdx.x >> 6] = ch[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[threadIdx.x][0];
    for (int w = 1; w < kBlock / 64; ++w) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
    if (threadIdx.x < 3) { if (v != INFINITY) atomicMin(&b.cb[threadIdx.x], ordered_u(v)); }
    else if (v != -INFINITY) atomicMax(&b.cb[threadIdx.x], ordered_u(v));
  }
}

__device__ inline unsigned long long spread21(uint32_t v) {       // bit i of v -> bit 3 i
  unsigned long long x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
__global__ __launch_bounds__(kBlock) void k_bld_codes(Bld b) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= b.n) return;
  const Box6 t = b.tbox[p];
  unsigned long long code = 0;
  for (int k = 0; k < 3; ++k) {
    const float cl = unordered_f(b.cb[k]), ch = unordered_f(b.cb[3 + k]);
    const float scale = ch > cl ? 2097152.0f / (ch - cl) : 0.0f;
    const float f = (0.5f * (t.lo[k] + t.hi[k]) - cl) * scale;
    const uint32_t q = f >= 2097151.0f ? 2097151u : (uint32_t)f;
    code |= spread21(q) << k;
  }
  b.key_a[p] = code; b.val_a[p] = p;
}

// ---- LSD radix sort, 8 bits per pass.  A wave owns a tile of kSortTile consecutive keys and takes them 64 at a time, in order; the rank of a key among
// the earlier keys of its tile with the same digit is (the tile's running count of that digit) + (lanes below it with that digit: 8 ballots intersected,
// then mbcnt) — stable, no atomics on the output side, no sort inside the tile.
__device__ inline uint32_t lane_of() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ inline uint32_t mbcnt_u64(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ inline void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ __launch_bounds__(kBlock) void k_sort_hist(const unsigned long long* keys, uint32_t n, int shift, uint32_t* hist) {
  __shared__ uint32_t h[kBlock / 64][256];
  const uint32_t wave = threadIdx.x >> 6, lane = lane_of();
  for (uint32_t i = lane; i < 256u; i += 64u) h[wave][i] = 0u;
  wave_sync_lds();
  const uint32_t tile = blockIdx.x * (kBlock / 64) + wave;
  const size_t base = (size_t)tile * kSortTile;
  for (uint32_t s = 0; s < kSortTile; s += 64u) {
    const size_t i = base + s + lane;
    if (i < n) atomicAdd(&h[wave][(uint32_t)(keys[i] >> shift) & 255u], 1u);
  }
  wave_sync_lds();
  if (base < n) for (uint32_t i = lane; i < 256u; i += 64u) hist[(size_t)tile * 256u + i] = h[wave][i];
}
// hist[tile][digit] -> the first output position of that digit of that tile: digits ascending, tiles ascending inside a digit.  k_sort_scan_tiles: block d
// turns digit d's column into its exclusive prefix over the tiles (256 tiles per round, wave shuffles + one LDS exchange) and leaves the column's total;
// k_sort_scan_digits: exclusive prefix of the 256 totals — the scatter adds the two.
__global__ __launch_bounds__(256) void k_sort_scan_tiles(uint32_t* hist, uint32_t tiles, uint32_t* total) {
  __shared__ uint32_t wsum[4];
  __shared__ uint32_t carry_s;
  const uint32_t d = blockIdx.x, lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0u;
  __syncthreads();
  for (uint32_t t0 = 0; t0 < tiles; t0 += 256u) {
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t v = t < tiles ? hist[(size_t)t * 256u + d] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if ((int)lane >= o) inc += u; }
    if (lane == 63u) wsum[wave] = inc;
    __syncthreads();
    uint32_t before = carry_s;
    for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
    if (t < tiles) hist[(size_t)t * 256u + d] = before + inc - v;
    __syncthreads();
    if (threadIdx.x == 255u) carry_s = before + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) total[d] = carry_s;
}
__global__ __launch_bounds__(256) void k_sort_scan_digits(uint32_t* total) {
  __shared__ uint32_t wsum[4];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t v = total[threadIdx.x];
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if ((int)lane >= o) inc += u; }
  if (lane == 63u) wsum[wave] = inc;
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
  total[threadIdx.x] = before + inc - v;
}
__global__ __launch_bounds__(kBlock) void k_sort_scatter(const unsigned long long* keys, const uint32_t* vals, uint32_t n, int shift, const uint32_t* offs, const uint32_t* digit_base,
                                                         unsigned long long* keys_out, uint32_t* vals_out) {
  __shared__ uint32_t cnt[kBlock / 64][256];
  const uint32_t wave = threadIdx.x >> 6, lane = lane_of();
  const uint32_t tile = blockIdx.x * (kBlock / 64) + wave;
  const size_t base = (size_t)tile * kSortTile;
  if (base >= n) return;
  for (uint32_t i = lane; i < 256u; i += 64u) cnt[wave][i] = digit_base[i] + offs[(size_t)tile * 256u + i];
  wave_sync_lds();
  for (uint32_t s = 0; s < kSortTile; s += 64u) {
    const size_t i = base + s + lane;
    const bool valid = i < n;
    const unsigned long long key = valid ? keys[i] : 0ull;
    const uint32_t val = valid ? vals[i] : 0u;
    const uint32_t dg = (uint32_t)(key >> shift) & 255u;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool on = ((dg >> bit) & 1u) != 0u;
      const unsigned long long m = __ballot(on);
      peers &= on ? m : ~m;
    }
    const uint32_t rank = mbcnt_u64(peers);
    uint32_t pos = 0;
    if (valid) pos = cnt[wave][dg] + rank;
    wave_sync_lds();
    if (valid && rank == 0u) cnt[wave][dg] += (uint32_t)__popcll(peers);
    wave_sync_lds();
    if (valid) { keys_out[pos] = key; vals_out[pos] = val; }
  }
}

// ---- the radix tree of the keys (code, position): every internal node from the sorted codes alone (Karras 2012).  Internal node i covers [min(i, j), max(i, j)];
// node 0 is the root.  A child link >= 0 is an internal node, < 0 is ~(position of a single triangle) — the host's SplitNode.
__device__ inline int delta_keys(const unsigned long long* codes, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const unsigned long long a = codes[i], c = codes[j];
  if (a != c) return (int)__clzll((long long)(a ^ c));
  return 64 + (int)__clz((int)((uint32_t)i ^ (uint32_t)j));
}
__global__ __launch_bounds__(kBlock) void k_bld_radix(Bld b) {
  const int i = (int)(blockIdx.x * kBlock + threadIdx.x), n = (int)b.n;
  if (i >= n - 1) return;
  const unsigned long long* codes = b.key_a;
  const int d = delta_keys(codes, n, i, i + 1) - delta_keys(codes, n, i, i - 1) > 0 ? 1 : -1;
  const int dmin = delta_keys(codes, n, i, i - d);
  int lmax = 2;
  while (delta_keys(codes, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax / 2; t >= 1; t /= 2)
    if (delta_keys(codes, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = delta_keys(codes, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) / 2;
    if (delta_keys(codes, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + (d < 0 ? -1 : 0);
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const int32_t left = lo == gamma ? ~gamma : gamma, right = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
  b.r_left[i] = left; b.r_right[i] = right; b.r_lo[i] = (uint32_t)lo; b.r_hi[i] = (uint32_t)hi;
  if (left < 0) b.leaf_parent[gamma] = (uint32_t)i; else b.r_parent[gamma] = (uint32_t)i;
  if (right < 0) b.leaf_parent[gamma + 1] = (uint32_t)i; else b.r_parent[gamma + 1] = (uint32_t)i;
  if (i == 0) b.r_parent[0] = kUnset;
}

// ---- bottom-up: box and collapse-cost table of every internal node (ptc_scene.cpp: compute_radix_boxes + "cost tables, bottom-up").  One ROUND per launch: a
// node whose children were complete before this launch computes and stamps itself with the round's number; a child stamped in the same round does not count, so
// everything a node reads was written by an earlier launch and no fence is needed.  (The textbook form — one thread per triangle walks up, the second arrival at
// a node proceeds — needs a release / acquire pair of agent-scope fences per step, and on a chip of 8 XCDs with an L2 each those are L2 write-backs and
// invalidations: 1.6 ms for the 249,936-triangle atrium against 0.4 ms for the ~60 rounds of this one.)
__device__ inline Box6 link_box(const Bld& b, int32_t link) { return link < 0 ? b.tbox[b.val_a[(size_t)~link]] : b.r_box[(size_t)link]; }
__device__ inline int min7(int k) { return k > 7 ? 7 : k; }
__global__ __launch_bounds__(kBlock) void k_bld_up(Bld b, uint32_t round) {
  const uint32_t cur = blockIdx.x * kBlock + threadIdx.x;
  if (cur + 1u >= b.n || b.r_flag[cur]) return;
  const int32_t left = b.r_left[cur], right = b.r_right[cur];
  if (left >= 0) { const uint32_t f = b.r_flag[left]; if (f == 0u || f >= round) return; }
  if (right >= 0) { const uint32_t f = b.r_flag[right]; if (f == 0u || f >= round) return; }
  {
    const Box6 bl = link_box(b, left), br = link_box(b, right);
    Box6 nb = bl;
    grow6(nb, br);
    float cl[8], cr[8];
    if (left < 0) { const float a = half_area6(bl) * 1.0f * 1.0f; for (int i = 1; i <= 7; ++i) cl[i] = a; }
    else { const DpT dl = b.dp[(size_t)left]; for (int i = 1; i <= 7; ++i) cl[i] = dl.c[i - 1]; }
    if (right < 0) { const float a = half_area6(br) * 1.0f * 1.0f; for (int i = 1; i <= 7; ++i) cr[i] = a; }
    else { const DpT dr = b.dp[(size_t)right]; for (int i = 1; i <= 7; ++i) cr[i] = dr.c[i - 1]; }
    DpT d;
    d.bits = 0u;
    float dist[9];
    for (int j = 2; j <= 8; ++j) {
      float best = INFINITY; int bk = 1;
      for (int k = 1; k < j; ++k) {
        const float v = cl[min7(k)] + cr[min7(j - k)];
        if (v < best) { best = v; bk = k; }
      }
      dist[j] = best; d.bits |= (uint32_t)bk << (8 + 3 * (j - 2));
    }
    const uint32_t P = b.r_hi[cur] - b.r_lo[cur] + 1u;
    const float area = half_area6(nb);
    const float cleaf = P <= kLeafMaxB ? area * (float)P * 1.0f : INFINITY;
    const float cint = dist[8] + area * 1.0f;
    const bool leaf1 = cleaf <= cint;
    if (leaf1) d.bits |= 1u;
    d.c[0] = leaf1 ? cleaf : cint;
    for (int i = 2; i <= 7; ++i) {
      if (dist[i] < d.c[i - 2]) d.c[i - 1] = dist[i]; else { d.c[i - 1] = d.c[i - 2]; d.bits |= 1u << i; }
    }
    b.r_box[cur] = nb;
    b.dp[cur] = d;
    b.r_flag[cur] = round;
  }
}

// ---- 8-wide collapse, one level of the wide tree per launch (ptc_scene.cpp: expand = forest + make_wide / assign_slots).  A link of a slot: >= 0 the radix node
// of an interior child, kEmptyLink an unused slot, else ~(first sorted position | (triangles - 1) << 28) of a leaf.
struct Kid { int32_t link; bool leaf; uint32_t lo, hi; Box6 box; };
__device__ inline void forest(const Bld& b, int32_t link0, int i0, Kid* kid, int& n) {
  int32_t sl[16]; int si[16]; int sp = 0;
  sl[sp] = link0; si[sp] = i0; ++sp;
  while (sp > 0) {
    --sp;
    const int32_t link = sl[sp]; int i = si[sp];
    Kid c;
    c.box = link_box(b, link);
    if (link < 0) { c.leaf = true; c.lo = c.hi = (uint32_t)~link; c.link = -1; kid[n++] = c; continue; }
    const uint32_t bits = b.dp[(size_t)link].bits;
    while (i > 1 && ((bits >> i) & 1u)) --i;
    if (i == 1) { c.leaf = (bits & 1u) != 0u; c.lo = b.r_lo[link]; c.hi = b.r_hi[link]; c.link = link; kid[n++] = c; continue; }
    const int kk = (int)((bits >> (8 + 3 * (i - 2))) & 7u);
    sl[sp] = b.r_right[link]; si[sp] = min7(i - kk); ++sp;       // left first
    sl[sp] = b.r_left[link]; si[sp] = min7(kk); ++sp;
  }
}
__global__ __launch_bounds__(64) void k_bld_widen(Bld b, uint32_t first, uint32_t count) {
  const uint32_t t = blockIdx.x * 64u + threadIdx.x;
  if (t >= count) return;
  const uint32_t idx = first + t;
  const int32_t r = b.w_radix[idx];
  Kid kid[kWideB];
  int n = 0;
  {
    const int kk = (int)((b.dp[(size_t)r].bits >> (8 + 3 * 6)) & 7u);       // k[8]
    forest(b, b.r_left[r], min7(kk), kid, n);
    forest(b, b.r_right[r], min7(8 - kk), kid, n);
  }
  // slots: repeatedly the (child, free slot) pair with the largest +-dx +-dy +-dz, d = child centre - node centre; ties: lowest child, then lowest slot
  Box6 nb;
  for (int k = 0; k < 3; ++k) { nb.lo[k] = INFINITY; nb.hi[k] = -INFINITY; }
  for (int i = 0; i < n; ++i) grow6(nb, kid[i].box);
  float d[kWideB][3];
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) d[i][k] = 0.5f * (kid[i].box.lo[k] + kid[i].box.hi[k]) - 0.5f * (nb.lo[k] + nb.hi[k]);
  uint32_t child_done = 0, slot_used = 0;
  int slot_of[kWideB];
  for (int round = 0; round < n; ++round) {
    int bi = -1, bs = -1; float bsc = 0.0f;
    for (int i = 0; i < n; ++i) {
      if ((child_done >> i) & 1u) continue;
      for (int sl = 0; sl < kWideB; ++sl) {
        if ((slot_used >> sl) & 1u) continue;
        const float sc = ((sl & 1) ? d[i][0] : -d[i][0]) + ((sl & 2) ? d[i][1] : -d[i][1]) + ((sl & 4) ? d[i][2] : -d[i][2]);
        if (bi < 0 || sc > bsc) { bi = i; bs = sl; bsc = sc; }
      }
    }
    child_done |= 1u << bi; slot_used |= 1u << bs; slot_of[bi] = bs;
  }
  int32_t links[kWideB]; uint32_t childs[kWideB];
  for (int sl = 0; sl < kWideB; ++sl) { links[sl] = (int32_t)kEmptyLink; childs[sl] = kUnset; }
  uint32_t ni = 0, nt = 0;
  for (int i = 0; i < n; ++i) {
    const int sl = slot_of[i];
    if (kid[i].leaf) { links[sl] = (int32_t)~(kid[i].lo | ((kid[i].hi - kid[i].lo) << 28)); nt += kid[i].hi - kid[i].lo + 1u; }
    else {
      links[sl] = kid[i].link;
      const uint32_t c = atomicAdd(&b.counters[0], 1u);
      b.w_radix[c] = kid[i].link; b.w_parent[c] = idx | ((uint32_t)sl << 28);
      childs[sl] = c; ++ni;
    }
  }
  for (int sl = 0; sl < kWideB; ++sl) { b.w_link[(size_t)idx * 8 + sl] = links[sl]; b.w_child[(size_t)idx * 8 + sl] = childs[sl]; }
  b.w_own[idx] = (4u * ni + 3u * nt + 3u) & ~3u;
  b.w_baddr[idx] = kUnset;
}

// units of a subtree's blocks (its own children block + those of all nodes below), one level per launch from the deepest up
__global__ __launch_bounds__(kBlock) void k_bld_sizes(Bld b, uint32_t first, uint32_t count) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= count) return;
  const uint32_t idx = first + t;
  uint32_t s = b.w_own[idx];
  for (int sl = 0; sl < kWideB; ++sl) { const uint32_t c = b.w_child[(size_t)idx * 8 + sl]; if (c != kUnset) s += b.w_sub[c]; }
  b.w_sub[idx] = s;
}

// The top of the layout, by one thread (it is at most `budget` + 7 nodes): children blocks breadth-first while fewer than `budget` nodes are numbered, then every
// numbered node that has no block yet gets the start of its subtree's depth-first stretch (ptc_scene.cpp: alloc_children, "breadth-first top", "depth-first remainder").
__global__ void k_bld_top(Bld b) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  uint32_t* order = b.top_order;
  uint32_t cnt = 1, head = 0, next_unit = 4;
  order[0] = 0u;
  b.w_naddr[0] = 0u;
  for (; head < cnt && cnt < b.budget; ++head) {
    const uint32_t idx = order[head];
    b.w_baddr[idx] = next_unit;
    next_unit += b.w_own[idx];
    for (int sl = 0; sl < kWideB; ++sl) { const uint32_t c = b.w_child[(size_t)idx * 8 + sl]; if (c != kUnset) order[cnt++] = c; }
  }
  for (uint32_t i = head; i < cnt; ++i) { const uint32_t idx = order[i]; b.w_baddr[idx] = next_unit; next_unit += b.w_sub[idx]; }
  b.counters[1] = next_unit; b.counters[2] = cnt;
}
// the rest, one level per launch from the root down: a node's block follows its parent's block and the subtrees of its lower interior siblings
__global__ __launch_bounds__(kBlock) void k_bld_addr(Bld b, uint32_t first, uint32_t count) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= count) return;
  const uint32_t idx = first + t;
  const uint32_t pw = b.w_parent[idx];
  if (pw == kUnset) return;                                  // the root
  const uint32_t p = pw & 0x0fffffffu, slot = pw >> 28;
  uint32_t below_units = 0, below_nodes = 0;
  for (uint32_t sl = 0; sl < slot; ++sl) { const uint32_t c = b.w_child[(size_t)p * 8 + sl]; if (c != kUnset) { below_units += b.w_sub[c]; ++below_nodes; } }
  const uint32_t pb = b.w_baddr[p];
  b.w_naddr[idx] = pb + 4u * below_nodes;
  if (b.w_baddr[idx] == kUnset) b.w_baddr[idx] = pb + b.w_own[p] + below_units;
}

// node headers, triangle records (ids only: the refit kernels write the geometry), and the refit's list of node addresses by level
__global__ __launch_bounds__(kBlock) void k_bld_emit(Bld b, uint32_t first, uint32_t count, float4* recs, uint32_t* level_nodes, uint32_t out_first) {
  const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= count) return;
  const uint32_t idx = first + t;
  uint32_t imask = 0, lmask = 0, two = 0, ni = 0;
  for (int sl = 0; sl < kWideB; ++sl) {
    const uint32_t l = (uint32_t)b.w_link[(size_t)idx * 8 + sl];
    if (l == kEmptyLink) continue;
    if ((int32_t)l >= 0) { imask |= 1u << sl; ++ni; }
    else { lmask |= 1u << sl; if (((~l) >> 28) == 1u) two |= 1u << sl; }
  }
  const uint32_t block = b.w_baddr[idx], at = b.w_naddr[idx];
  reinterpret_cast<uint4*>(recs)[at] = make_uint4(0u, 0u, (imask << 8) | (lmask << 16) | (two << 24), block);
  uint32_t tri = block + 4u * ni;
  for (int sl = 0; sl < kWideB; ++sl) {
    if (!((lmask >> sl) & 1u)) continue;
    const uint32_t code = ~(uint32_t)b.w_link[(size_t)idx * 8 + sl];
    const uint32_t lo = code & 0x0fffffffu, cntt = (code >> 28) + 1u;
    for (uint32_t k = 0; k < cntt; ++k) {
      const uint32_t prim = b.val_a[lo + k];
      recs[tri] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(prim));
      recs[tri + 1] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(b.prim_cls[prim]));
      recs[tri + 2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      tri += 3u;
    }
  }
  level_nodes[out_first + t] = at;
}

template <class T> T* carve(char*& p, size_t count) {
  T* r = reinterpret_cast<T*>(p);
  p += (count * sizeof(T) + 255u) & ~(size_t)255u;
  return r;
}
#define BLD_TRY(x) do { const hipError_t e_ = (x); if (e_ != hipSuccess) return std::string("device build: ") + hipGetErrorString(e_); } while (0)
}  // namespace

std::string pt_build_lbvh(hipStream_t st, const HostVertex* wverts, const uint32_t* widx, const uint32_t* prim_cls, uint32_t n, uint32_t toplet_budget,
                          BuildScratch& S, BuildOut& out) {
  if (n < 2u) return "device build: fewer than two triangles (the host build handles the special case)";
  if (n > (1u << 24)) return "device build: more than 2^24 triangles";
  const uint32_t tiles = (n + kSortTile - 1u) / kSortTile;
  const size_t budget = toplet_budget;
  // ---- scratch: one allocation, carved ----
  Bld b{};
  b.wverts = wverts; b.widx = widx; b.prim_cls = prim_cls; b.n = n; b.budget = toplet_budget;
  auto layout = [&](char* base) {
    char* p = base;
    b.tbox = carve<Box6>(p, n); b.cb = carve<uint32_t>(p, 8);
    b.key_a = carve<unsigned long long>(p, n); b.key_b = carve<unsigned long long>(p, n); b.val_a = carve<uint32_t>(p, n); b.val_b = carve<uint32_t>(p, n);
    b.hist = carve<uint32_t>(p, (size_t)(tiles + 4u) * 256u); b.digit_total = carve<uint32_t>(p, 256);
    b.r_left = carve<int32_t>(p, n); b.r_right = carve<int32_t>(p, n); b.r_lo = carve<uint32_t>(p, n); b.r_hi = carve<uint32_t>(p, n);
    b.r_parent = carve<uint32_t>(p, n); b.leaf_parent = carve<uint32_t>(p, n); b.r_flag = carve<uint32_t>(p, n);
    b.r_box = carve<Box6>(p, n); b.dp = carve<DpT>(p, n);
    b.w_radix = carve<int32_t>(p, n); b.w_parent = carve<uint32_t>(p, n); b.w_link = carve<int32_t>(p, (size_t)n * 8); b.w_child = carve<uint32_t>(p, (size_t)n * 8);
    b.w_own = carve<uint32_t>(p, n); b.w_sub = carve<uint32_t>(p, n); b.w_baddr = carve<uint32_t>(p, n); b.w_naddr = carve<uint32_t>(p, n);
    b.counters = carve<uint32_t>(p, 8); b.top_order = carve<uint32_t>(p, budget + 16);
    return (size_t)(p - base);
  };
  const size_t need = layout(nullptr);
  if (S.bytes < need) {
    if (S.p) (void)hipFree(S.p);
    S.p = nullptr; S.bytes = 0;
    BLD_TRY(hipMalloc(&S.p, need));
    S.bytes = need;
  }
  layout(static_cast<char*>(S.p));
  const auto blocks = [](uint32_t count, uint32_t per) { return dim3((count + per - 1u) / per); };
  // ---- boxes, keys, sort ----
  hipLaunchKernelGGL(k_bld_init, dim3(1), dim3(64), 0, st, b);
  BLD_TRY(hipMemsetAsync(b.r_flag, 0, (size_t)n * 4, st));
  { uint32_t g = (n + kBlock - 1u) / kBlock; if (g > 1024u) g = 1024u; hipLaunchKernelGGL(k_bld_prims, dim3(g), dim3(kBlock), 0, st, b); }
  hipLaunchKernelGGL(k_bld_codes, blocks(n, kBlock), dim3(kBlock), 0, st, b);
  {
    const dim3 g((tiles + kBlock / 64 - 1u) / (kBlock / 64));
    unsigned long long *ki = b.key_a, *ko = b.key_b; uint32_t *vi = b.val_a, *vo = b.val_b;
    for (int pass = 0; pass < 8; ++pass) {          // 63 bits of code: 8 passes of 8; an even number of passes leaves the result in key_a / val_a
      hipLaunchKernelGGL(k_sort_hist, g, dim3(kBlock), 0, st, ki, n, pass * 8, b.hist);
      hipLaunchKernelGGL(k_sort_scan_tiles, dim3(256), dim3(256), 0, st, b.hist, tiles, b.digit_total);
      hipLaunchKernelGGL(k_sort_scan_digits, dim3(1), dim3(256), 0, st, b.digit_total);
      hipLaunchKernelGGL(k_sort_scatter, g, dim3(kBlock), 0, st, ki, vi, n, pass * 8, b.hist, b.digit_total, ko, vo);
      std::swap(ki, ko); std::swap(vi, vo);
    }
  }
  // ---- binary tree, boxes, cost tables ----
  hipLaunchKernelGGL(k_bld_radix, blocks(n - 1u, kBlock), dim3(kBlock), 0, st, b);
  for (uint32_t round = 1;;) {            // rounds = the height of the binary tree (40-70 for a scene's Morton codes); the root's stamp is read back every 16
    for (int k = 0; k < 16; ++k, ++round) hipLaunchKernelGGL(k_bld_up, blocks(n - 1u, kBlock), dim3(kBlock), 0, st, b, round);
    uint32_t root_done = 0;
    BLD_TRY(hipMemcpyAsync(&root_done, &b.r_flag[0], 4, hipMemcpyDeviceToHost, st));
    BLD_TRY(hipStreamSynchronize(st));
    if (root_done) break;
    if (round > 8192u) return "device build: the binary tree is deeper than 8192 levels";
  }
  // ---- 8-wide collapse, level by level ----
  std::vector<uint32_t> lvl_first{0u};
  uint32_t total = 1;
  for (uint32_t first = 0, count = 1; count > 0;) {
    hipLaunchKernelGGL(k_bld_widen, blocks(count, 64), dim3(64), 0, st, b, first, count);
    uint32_t now = 0;
    BLD_TRY(hipMemcpyAsync(&now, &b.counters[0], 4, hipMemcpyDeviceToHost, st));
    BLD_TRY(hipStreamSynchronize(st));
    first += count; count = now - first; total = now;
    lvl_first.push_back(first);
    if (lvl_first.size() > 4096) return "device build: the 8-wide tree is deeper than 4096 levels";
  }
  lvl_first.pop_back();                                      // the last push was the end of the last non-empty level...
  lvl_first.push_back(total);                                // ... which is `total`
  const size_t levels = lvl_first.size() - 1;
  for (size_t l = levels; l-- > 0;) hipLaunchKernelGGL(k_bld_sizes, blocks(lvl_first[l + 1] - lvl_first[l], kBlock), dim3(kBlock), 0, st, b, lvl_first[l], lvl_first[l + 1] - lvl_first[l]);
  hipLaunchKernelGGL(k_bld_top, dim3(1), dim3(64), 0, st, b);
  for (size_t l = 1; l < levels; ++l) hipLaunchKernelGGL(k_bld_addr, blocks(lvl_first[l + 1] - lvl_first[l], kBlock), dim3(kBlock), 0, st, b, lvl_first[l], lvl_first[l + 1] - lvl_first[l]);
  uint32_t ctr[4] = {0, 0, 0, 0};
  BLD_TRY(hipMemcpyAsync(ctr, b.counters, sizeof ctr, hipMemcpyDeviceToHost, st));
  BLD_TRY(hipStreamSynchronize(st));
  const uint32_t n_units = ctr[1];
  if (n_units >= (1u << 31) || n_units < 4u) return "device build: BVH too large";
  // ---- emit ----
  if (!out.recs || out.recs_cap < n_units) {
    if (out.recs) (void)hipFree(out.recs);
    out.recs = nullptr; out.recs_cap = (size_t)n_units + n_units / 8u;
    if (hipMalloc((void**)&out.recs, out.recs_cap * 16) != hipSuccess) { out.recs = nullptr; out.recs_cap = 0; return "device build: out of device memory"; }
  }
  if (!out.level_nodes || out.level_cap < total) {
    if (out.level_nodes) (void)hipFree(out.level_nodes);
    out.level_nodes = nullptr; out.level_cap = (size_t)total + total / 8u;
    if (hipMalloc((void**)&out.level_nodes, out.level_cap * 4) != hipSuccess) { out.level_nodes = nullptr; out.level_cap = 0; return "device build: out of device memory"; }
  }
  float4* recs = out.recs; uint32_t* level_nodes = out.level_nodes;
  BLD_TRY(hipMemsetAsync(recs, 0, (size_t)n_units * 16, st));
  out.level_first.assign(1, 0u);
  uint32_t pos = 0;
  for (size_t l = levels; l-- > 0;) {                        // deepest level first
    const uint32_t count = lvl_first[l + 1] - lvl_first[l];
    hipLaunchKernelGGL(k_bld_emit, blocks(count, kBlock), dim3(kBlock), 0, st, b, lvl_first[l], count, recs, level_nodes, pos);
    pos += count;
    out.level_first.push_back(pos);
  }
  BLD_TRY(hipGetLastError());
  out.n_nodes = total; out.n_units = n_units; out.max_depth = (uint32_t)levels - 1u; out.n_tri_records = n;
  return std::string();
}
