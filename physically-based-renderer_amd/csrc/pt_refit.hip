// pt_refit.hip — the refit of a committed scene on the device (see pt_refit.h; the definition of every value is ptc_refit_scene in ptc_scene.cpp).
//
//   k_refit_flatten   one thread per world vertex: instance transform of position / normal / tangent / bitangent           (R1, vertex.glsl:28-40)
//   k_refit_prims     grid-stride over the primitives: the shading record of each, and the scene box by block reduction + 6 atomics per block
//   k_refit_nodes     one launch per level of the 8-wide tree, deepest first, one thread per node: child boxes (triangles of the leaf slots,
//                     the stored boxes of the interior children), node origin on the scene grid, 8-bit child planes, the triangle records of
//                     the leaf slots; the node keeps its slot masks and children block, a triangle record its primitive id and class
// All of it is HBM-bound byte shuffling with a few divisions per node: 48 B read + 60 B written per vertex, 3 gathers + 80 / 192 B written per
// primitive, 64 B + its triangles read and written per node.  No LDS tiling applies (every record is touched once), so the only layout rule is
// the one the arrays already obey: whole records per thread, threads in record order.
#include "pt_refit.h"

namespace {
constexpr int kBlock = 256;
constexpr int kNodeBlock = 64;

__device__ inline float dot3(const float a[3], const float b[3]) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
__device__ inline void cross3(const float a[3], const float b[3], float o[3]) {
  o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
  o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
  o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}
__device__ inline void normalize3(float v[3]) {
  const float inv = 1.0f / sqrtf(dot3(v, v));
  v[0] *= inv; v[1] *= inv; v[2] *= inv;
}
__device__ inline void mul_n(const float* N, const float v[3], float o[3]) {
  o[0] = N[0] * v[0] + N[3] * v[1] + N[6] * v[2];
  o[1] = N[1] * v[0] + N[4] * v[1] + N[7] * v[2];
  o[2] = N[2] * v[0] + N[5] * v[1] + N[8] * v[2];
}
// order-preserving map float → uint32 (negative values below positive ones), for atomicMin / atomicMax
__device__ inline uint32_t ordered(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }

__global__ void k_refit_init(uint32_t* bounds) {
  const uint32_t i = threadIdx.x;
  if (i < 8) bounds[i] = i < 3 ? 0xffffffffu : 0u;
}

__global__ __launch_bounds__(kBlock) void k_refit_flatten(const DevRefit r) {
  const uint32_t v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= r.n_verts) return;
  const uint32_t inst = r.vert_inst[v];
  const float* m = r.inst_xf + (size_t)inst * 21;      // m[c*3 + row] = model[c*4 + row]
  const float* N = m + 12;
  const HostVertex s = r.mesh_verts[r.inst_src[inst] + (v - r.inst_first[inst])];
  HostVertex d;
  for (int k = 0; k < 3; ++k) d.position[k] = m[0 + k] * s.position[0] + m[3 + k] * s.position[1] + m[6 + k] * s.position[2] + m[9 + k];
  mul_n(N, s.normal, d.normal);
  normalize3(d.normal);
  float t3[3];
  mul_n(N, s.tangent, t3);
  normalize3(t3);
  d.tangent[0] = t3[0]; d.tangent[1] = t3[1]; d.tangent[2] = t3[2]; d.tangent[3] = s.tangent[3];
  float cr[3], sc[3], bt[3];
  cross3(s.normal, s.tangent, cr);
  sc[0] = cr[0] * s.tangent[3]; sc[1] = cr[1] * s.tangent[3]; sc[2] = cr[2] * s.tangent[3];
  mul_n(N, sc, bt);
  normalize3(bt);
  d.texcoord[0] = s.texcoord[0]; d.texcoord[1] = s.texcoord[1];
  r.wverts[v] = d;
  r.wbt[(size_t)v * 3 + 0] = bt[0]; r.wbt[(size_t)v * 3 + 1] = bt[1]; r.wbt[(size_t)v * 3 + 2] = bt[2];
  if (!(isfinite(d.position[0]) && isfinite(d.position[1]) && isfinite(d.position[2]))) atomicOr(&r.bounds[6], 1u);
}

__global__ __launch_bounds__(kBlock) void k_refit_prims(const DevRefit r) {
  if (r.bounds[6]) return;                 // a non-finite position: the host reports it, the scene in HBM stays as it was
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  const bool tex = r.shade_stride == 12u;
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < r.n_tris; p += gridDim.x * kBlock) {
    const uint32_t vi[3] = {r.widx[(size_t)p * 3], r.widx[(size_t)p * 3 + 1], r.widx[(size_t)p * 3 + 2]};
    const HostVertex a = r.wverts[vi[0]], b = r.wverts[vi[1]], c = r.wverts[vi[2]];
    for (int k = 0; k < 3; ++k) {
      lo[k] = fminf(lo[k], fminf(a.position[k], fminf(b.position[k], c.position[k])));
      hi[k] = fmaxf(hi[k], fmaxf(a.position[k], fmaxf(b.position[k], c.position[k])));
    }
    float4* o = r.shade + (size_t)p * r.shade_stride;
    const float mat = o[0].w, light = o[1].w;        // material and emitter index of the primitive: not a refit's business
    o[0] = make_float4(a.position[0], a.position[1], a.position[2], mat);
    o[1] = make_float4(b.position[0], b.position[1], b.position[2], light);
    o[2] = make_float4(c.position[0], c.position[1], c.position[2], a.normal[0]);
    o[3] = make_float4(a.normal[1], a.normal[2], b.normal[0], b.normal[1]);
    o[4] = make_float4(b.normal[2], c.normal[0], c.normal[1], c.normal[2]);
    if (tex) {
      const float* ba = r.wbt + (size_t)vi[0] * 3; const float* bb = r.wbt + (size_t)vi[1] * 3; const float* bc = r.wbt + (size_t)vi[2] * 3;
      o[5] = make_float4(a.texcoord[0], a.texcoord[1], b.texcoord[0], b.texcoord[1]);
      o[6] = make_float4(c.texcoord[0], c.texcoord[1], a.tangent[0], a.tangent[1]);
      o[7] = make_float4(a.tangent[2], b.tangent[0], b.tangent[1], b.tangent[2]);
      o[8] = make_float4(c.tangent[0], c.tangent[1], c.tangent[2], ba[0]);
      o[9] = make_float4(ba[1], ba[2], bb[0], bb[1]);
      o[10] = make_float4(bb[2], bc[0], bc[1], bc[2]);
    }
  }
  // scene box: min / max over finite values is the same number in any order (only the sign of a zero can differ, which nothing downstream reads)
  __shared__ float red[6][kBlock / 64];
  for (int k = 0; k < 3; ++k) {
    for (int d = 32; d >= 1; d >>= 1) { lo[k] = fminf(lo[k], __shfl_xor(lo[k], d)); hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], d)); }
    if ((threadIdx.x & 63) == 0) { red[k][threadIdx.x >> 6] = lo[k]; red[3 + k][threadIdx.x >> 6] = hi[k]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = red[threadIdx.x][0];
    for (int w = 1; w < kBlock / 64; ++w) v = threadIdx.x < 3 ? fminf(v, red[threadIdx.x][w]) : fmaxf(v, red[threadIdx.x][w]);
    if (threadIdx.x < 3) { if (v != INFINITY) atomicMin(&r.bounds[threadIdx.x], ordered(v)); }
    else if (v != -INFINITY) atomicMax(&r.bounds[threadIdx.x], ordered(v));
  }
}

struct Box3 { float lo[3], hi[3]; };

// box of the triangle whose record starts at unit `at` (the record's word 3 is its primitive id), as the host's tbox: < and > from +-inf over the vertices in order
__device__ inline Box3 triangle_box(const DevRefit& r, uint32_t at, const float*& pa, const float*& pb, const float*& pc) {
  const uint32_t prim = __float_as_uint(r.recs[at].w);
  pa = r.wverts[r.widx[(size_t)prim * 3 + 0]].position;
  pb = r.wverts[r.widx[(size_t)prim * 3 + 1]].position;
  pc = r.wverts[r.widx[(size_t)prim * 3 + 2]].position;
  Box3 b;
  for (int k = 0; k < 3; ++k) {
    float l = INFINITY, h = -INFINITY;
    l = pa[k] < l ? pa[k] : l; h = pa[k] > h ? pa[k] : h;
    l = pb[k] < l ? pb[k] : l; h = pb[k] > h ? pb[k] : h;
    l = pc[k] < l ? pc[k] : l; h = pc[k] > h ? pc[k] : h;
    b.lo[k] = l; b.hi[k] = h;
  }
  return b;
}
__device__ inline void write_triangle(const DevRefit& r, uint32_t at, const float* a, const float* b, const float* c) {
  const float w0 = r.recs[at].w, w1 = r.recs[at + 1].w;      // primitive id, material class
  r.recs[at] = make_float4(a[0], a[1], a[2], w0);
  r.recs[at + 1] = make_float4(b[0] - a[0], b[1] - a[1], b[2] - a[2], w1);
  r.recs[at + 2] = make_float4(c[0] - a[0], c[1] - a[1], c[2] - a[2], 0.0f);
}

// EIGHT lanes per node, one per child slot (round 4, second session).  One thread per node walked its eight slots one after the other: up to 16 triangles behind three dependent
// loads each (record -> vertex index -> vertex) and ~60 correctly rounded divisions, 20-50 us per LEVEL whatever its size — 0.35 of the refit's 0.5 ms.  Now lane s of an octet
// loads the box of slot s (an interior child's stored box, or its one or two triangles, whose records it rewrites), the octet exchanges the eight boxes (48 shuffles), every lane
// derives the node's box, origin and first exponent from them — the arithmetic and the order of the one-thread version, so the bytes are the same —, and in the exponent search lane
// s quantises slot s; the octet's verdict is a byte of a ballot.  Lane 0 packs and writes the node.
__global__ __launch_bounds__(kNodeBlock) void k_refit_nodes(const DevRefit r, uint32_t first, uint32_t count, float glx, float gly, float glz, float gsx, float gsy, float gsz, float scene_area) {
  const uint32_t t = blockIdx.x * kNodeBlock + threadIdx.x;
  const uint32_t i = t >> 3, sl = t & 7u;
  const int oct0 = (int)(threadIdx.x & 63u & ~7u);          // first lane of this octet inside the wave
  const bool live = i < count;                             // uniform over an octet
  unsigned long long my_cost = 0ull;
  uint32_t addr = 0, imask = 0, lmask = 0, two = 0, block = 0, headz = 0;
  if (live) {
    addr = r.level_nodes[first + i];
    const uint4 head = reinterpret_cast<const uint4*>(r.recs)[addr];
    imask = (head.z >> 8) & 255u; lmask = (head.z >> 16) & 255u; two = (head.z >> 24) & 255u; block = head.w; headz = head.z;
  }
  const uint32_t used = imask | lmask;
  const bool mine = ((used >> sl) & 1u) != 0u;
  Box3 b;
  for (int k = 0; k < 3; ++k) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; }
  if ((imask >> sl) & 1u) {
    const uint32_t child = block + 4u * (uint32_t)__popc(imask & ((1u << sl) - 1u));
    const float* nb = r.nbox + (size_t)(child >> 2) * 6;
    for (int k = 0; k < 3; ++k) { b.lo[k] = nb[k]; b.hi[k] = nb[3 + k]; }
  } else if ((lmask >> sl) & 1u) {
    const uint32_t below = (1u << sl) - 1u;
    const uint32_t tris_below = (uint32_t)__popc(lmask & below) + (uint32_t)__popc(two & lmask & below);
    const uint32_t at = block + 4u * (uint32_t)__popc(imask) + 3u * tris_below;
    const float *pa, *pb, *pc;
    b = triangle_box(r, at, pa, pb, pc);
    write_triangle(r, at, pa, pb, pc);
    if ((two >> sl) & 1u) {
      const Box3 b2 = triangle_box(r, at + 3u, pa, pb, pc);
      write_triangle(r, at + 3u, pa, pb, pc);
      for (int k = 0; k < 3; ++k) { b.lo[k] = b2.lo[k] < b.lo[k] ? b2.lo[k] : b.lo[k]; b.hi[k] = b2.hi[k] > b.hi[k] ? b2.hi[k] : b.hi[k]; }
    }
  }
  // the eight boxes of the node, in every lane of its octet
  float clo[3][8], chi[3][8];
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) { clo[k][j] = __shfl(b.lo[k], oct0 + j); chi[k][j] = __shfl(b.hi[k], oct0 + j); }
  const float glo[3] = {glx, gly, glz}, gstep[3] = {gsx, gsy, gsz};
  uint32_t oq[3], e3[3], wlo[3][2], whi[3][2];
  float nbx[6];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float nlo = 0.0f, nhi = 0.0f;
    bool have = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (!((used >> j) & 1u)) continue;
      if (!have) { nlo = clo[k][j]; nhi = chi[k][j]; have = true; }
      else { nlo = clo[k][j] < nlo ? clo[k][j] : nlo; nhi = chi[k][j] > nhi ? chi[k][j] : nhi; }
    }
    nbx[k] = nlo; nbx[3 + k] = nhi;
    // origin on the 16-bit scene grid, rounded down
    float fq = floorf((nlo - glo[k]) / gstep[k]);
    if (fq < 0.0f) fq = 0.0f;
    if (fq > 65535.0f) fq = 65535.0f;
    uint32_t q16 = (uint32_t)fq;
    while (q16 > 0u && fmaf((float)q16, gstep[k], glo[k]) > nlo) --q16;
    oq[k] = q16;
    const float org = fmaf((float)q16, gstep[k], glo[k]);
    // 8-bit planes: the smallest power-of-two scale that covers the node, lower planes floored, upper planes ceiled, nudged until they bracket
    const float f = (nhi - org) / 255.0f;
    const uint32_t u = __float_as_uint(f);
    uint32_t e = (u >> 23) & 255u;
    if (u & 0x007fffffu) e += 1u;
    if (e < 1u) e = 1u;
    bool done = !live;
    uint32_t ql = 255u, qh = 0u;           // an empty slot
    for (;;) {
      bool ok = true;
      if (!done && mine) {
        const float sc = __uint_as_float(e << 23);
        float fl = floorf((b.lo[k] - org) / sc);
        if (fl < 0.0f) fl = 0.0f;
        if (fl > 255.0f) fl = 255.0f;
        int q = (int)fl;
        while (q > 0 && org + (float)q * sc > b.lo[k]) --q;
        ql = (uint32_t)q;
        float ce = ceilf((b.hi[k] - org) / sc);
        if (ce < 0.0f) ce = 0.0f;
        if (ce > 255.0f) { ok = false; ce = 255.0f; }
        int q2 = (int)ce;
        while (q2 < 255 && org + (float)q2 * sc < b.hi[k]) ++q2;
        if (org + (float)q2 * sc < b.hi[k]) ok = false;
        qh = (uint32_t)q2;
      }
      const unsigned long long verdict = __ballot(done || ok);
      if (!done) { if (((verdict >> oct0) & 0xffull) == 0xffull) done = true; else ++e; }
      if (!__ballot(!done)) break;
    }
    e3[k] = e;
    uint32_t pl[2] = {0u, 0u}, ph[2] = {0u, 0u};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t a = (uint32_t)__shfl((int)ql, oct0 + j), c = (uint32_t)__shfl((int)qh, oct0 + j);
      pl[j >> 2] |= a << (8 * (j & 3));
      ph[j >> 2] |= c << (8 * (j & 3));
    }
    wlo[k][0] = pl[0]; wlo[k][1] = pl[1]; whi[k][0] = ph[0]; whi[k][1] = ph[1];
  }
  if (live && sl == 0u) {
    uint4* out = reinterpret_cast<uint4*>(r.recs) + addr;
    out[0] = make_uint4(oq[0] | (oq[1] << 16), oq[2] | (e3[0] << 16) | (e3[1] << 24), e3[2] | (headz & 0xffffff00u), block);
    out[1] = make_uint4(wlo[0][0], wlo[0][1], wlo[1][0], wlo[1][1]);
    out[2] = make_uint4(wlo[2][0], wlo[2][1], whi[0][0], whi[0][1]);
    out[3] = make_uint4(whi[1][0], whi[1][1], whi[2][0], whi[2][1]);
    float* nb = r.nbox + (size_t)(addr >> 2) * 6;
    for (int k = 0; k < 6; ++k) nb[k] = nbx[k];
  }
  // surface-area cost of this node's children (ptc_stats.bvh_sa_cost): half_area(child) / half_area(scene) per used slot, a two-triangle leaf twice
  if (live && mine && scene_area > 0.0f) {
    const float ex = b.hi[0] - b.lo[0], ey = b.hi[1] - b.lo[1], ez = b.hi[2] - b.lo[2];
    const unsigned long long term = (unsigned long long)(((ex * ey + ey * ez + ez * ex) / scene_area) * PTC_SA_COST_ONE);
    my_cost = ((two >> sl) & 1u) ? 2ull * term : term;
  }
  // one atomic per wave (a block is one wave): integer sum, the same bits in any order
  for (int d = 32; d >= 1; d >>= 1) my_cost += __shfl_xor(my_cost, d);
  if ((threadIdx.x & 63) == 0 && my_cost) atomicAdd(r.cost, my_cost);
}
}  // namespace

// A commit on the device starts from shading records that hold nothing but their two indices (k_refit_prims writes the rest around them).
namespace {
__global__ __launch_bounds__(kBlock) void k_refit_seed(float4* shade, uint32_t stride, const int32_t* tri_mat, const int32_t* prim_light, uint32_t n) {
  const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= n) return;
  float4* o = shade + (size_t)p * stride;
  o[0].w = __int_as_float(tri_mat[p]);
  o[1].w = __int_as_float(prim_light[p]);
}
}  // namespace
void pt_launch_refit_seed(hipStream_t st, const DevRefit& r, const int32_t* tri_mat, const int32_t* prim_light) {
  if (r.n_tris) hipLaunchKernelGGL(k_refit_seed, dim3((r.n_tris + kBlock - 1) / kBlock), dim3(kBlock), 0, st, r.shade, r.shade_stride, tri_mat, prim_light, r.n_tris);
}

void pt_launch_refit_geometry(hipStream_t st, const DevRefit& r) {
  hipLaunchKernelGGL(k_refit_init, dim3(1), dim3(64), 0, st, r.bounds);
  if (r.n_verts) hipLaunchKernelGGL(k_refit_flatten, dim3((r.n_verts + kBlock - 1) / kBlock), dim3(kBlock), 0, st, r);
  uint32_t blocks = (r.n_tris + kBlock - 1) / kBlock;
  if (blocks > 1024u) blocks = 1024u;
  if (blocks) hipLaunchKernelGGL(k_refit_prims, dim3(blocks), dim3(kBlock), 0, st, r);
}

void pt_refit_decode_bounds(const uint32_t raw[8], float lo[3], float hi[3], bool* non_finite) {
  for (int k = 0; k < 6; ++k) {
    const uint32_t enc = raw[k];
    const uint32_t u = (enc & 0x80000000u) ? (enc ^ 0x80000000u) : ~enc;
    float f;
    __builtin_memcpy(&f, &u, 4);
    (k < 3 ? lo[k] : hi[k - 3]) = f;
  }
  *non_finite = raw[6] != 0u;
}

void pt_launch_refit_nodes(hipStream_t st, const DevRefit& r, const std::vector<uint32_t>& level_first, const float gl[3], const float gs[3], float scene_area) {
  (void)hipMemsetAsync(r.cost, 0, sizeof(unsigned long long), st);
  for (size_t l = 0; l + 1 < level_first.size(); ++l) {
    const uint32_t first = level_first[l], count = level_first[l + 1] - first;
    if (!count) continue;
    hipLaunchKernelGGL(k_refit_nodes, dim3((count * 8u + kNodeBlock - 1) / kNodeBlock), dim3(kNodeBlock), 0, st, r, first, count, gl[0], gl[1], gl[2], gs[0], gs[1], gs[2], scene_area);      // eight lanes per node
  }
}
