// pt_kernels.hip — the wavefront path tracer's kernels for gfx950 (wave64).
//
// Stages:  P1 k_raygen · P3 k_trace_closest · P9 per-wave class sort (epilogue of
// k_trace_closest) · P5–P8 k_shade · P4 k_trace_any · P10 k_accumulate / k_resolve · R8 k_shade_raster
// · R9 k_tonemap.  All queue traffic is SoA of 16-byte lanes (one dwordx4 per lane, 1 KiB per
// wave-instruction); compaction is ballot + mbcnt prefix + ONE atomic per wave and class; no float
// atomics anywhere (every per-path / per-pixel word has a single owner), so results do not depend on
// scheduling.  MFMA is unused: there is no dense contraction on this path.
#include "pt_device.h"
#include "ptc_internal.h"

#define TRACE_BLOCK 256
#define TRACE_WAVES (TRACE_BLOCK / 64)
#define LDS_STACK_DEPTH 32
#define SCRATCH_STACK_DEPTH 64
#define CUR_DONE ((int)0x80000000)

// ---- small helpers --------------------------------------------------------------------------------
PT_DEV float4 ld4(const float4* p) { return *p; }

// One wave-wide allocation of popc(mask) consecutive slots; lanes in `mask` get base+rank.
PT_DEV uint32_t wave_alloc(uint32_t* ctr, uint64_t mask, uint32_t lane) {
  uint32_t base = 0;
  const int leader = __ffsll((unsigned long long)mask) - 1;
  if ((int)lane == leader) base = atomicAdd(ctr, (uint32_t)__popcll(mask));
  base = __shfl((int)base, leader);
  return base + mbcnt64(mask);
}
PT_DEV uint32_t wave_fetch(uint32_t* ctr, uint32_t amount, uint32_t lane) {
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(ctr, amount);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
}
PT_DEV unsigned long long wave_sum(unsigned long long v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// =================================================================================================
// bookkeeping kernels
__global__ void k_set_counts(uint32_t* cnt, uint32_t n_rays, uint32_t n_shadow) {
  if (threadIdx.x < CNT_N) cnt[threadIdx.x] = (threadIdx.x == CNT_RAYS) ? n_rays : (threadIdx.x == CNT_SHADOW ? n_shadow : 0u);
}
__global__ void k_advance(uint32_t* cnt) {
  if (threadIdx.x == 0) {
    const uint32_t next = cnt[CNT_NEXT];
    cnt[CNT_RAYS] = next; cnt[CNT_SORT0] = 0; cnt[CNT_SORT1] = 0; cnt[CNT_NEXT] = 0; cnt[CNT_SHADOW] = 0;
    cnt[CNT_WORK_TRACE] = 0; cnt[CNT_WORK_SHADE] = 0; cnt[CNT_WORK_SHADOW] = 0;
  }
}

// =================================================================================================
// P1 ray generation.  path id p → owned pixel j = p % n_owned, sample = first + p / n_owned.
// ndc = 2·((px+ξ)/W, (py+η)/H) − 1, y-down, no flip (PbrRenderSystem.cpp:425-430);
// view-space dir (ndc.x·aspect·tan(fov/2), ndc.y·tan(fov/2), −1) taken to world by the lookAtRH basis.
template <bool RASTER>
__global__ __launch_bounds__(256) void k_raygen(DevCamera cam, DevFrame fr, DevQueues q, uint32_t first_sample, uint32_t n_paths) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= n_paths) return;
  const uint32_t j = p % fr.n_owned, sl = p / fr.n_owned;
  const uint32_t pixel = fr.owned[j];
  const uint32_t px = pixel % (uint32_t)fr.w, py = pixel / (uint32_t)fr.w;
  const uint32_t sample = first_sample + sl;
  const uint32_t key = path_key(fr.seed_hash, pixel, sample);
  float jx = 0.5f, jy = 0.5f;
  if (!RASTER) { jx = rng_f(key, 0, 0); jy = rng_f(key, 0, 1); }
  const float fx = ((float)px + jx) / (float)fr.w, fy = ((float)py + jy) / (float)fr.h;
  const float dvx = (2.0f * fx - 1.0f) * cam.sx, dvy = (2.0f * fy - 1.0f) * cam.sy;
  const v3 cs = V3(cam.s[0], cam.s[1], cam.s[2]), cu = V3(cam.u[0], cam.u[1], cam.u[2]), cf = V3(cam.f[0], cam.f[1], cam.f[2]);
  const v3 d = normalize3(vfma(cs, dvx, vfma(cu, dvy, cf)));
  float bz = 1.0f, bw = 1.0f;   // throughput.xy
  if (RASTER) {                 // Vulkan clips NDC z to [0,1] under a −1..1 projection: near = 2fn/(f+n)
    const float len = pt_sqrt(pt_fma(dvy, dvy, pt_fma(dvx, dvx, 1.0f)));
    const float dnear = (2.0f * PT_ZFAR * PT_ZNEAR) / (PT_ZFAR + PT_ZNEAR);
    bz = dnear * len; bw = PT_ZFAR * len;
  }
  q.ray.A[p] = make_float4(cam.pos[0], cam.pos[1], cam.pos[2], d.x);
  q.ray.B[p] = make_float4(d.y, d.z, bz, bw);
  q.ray.C[p] = make_float4(1.0f, 0.0f, __uint_as_float(p), __uint_as_float(key));
  q.ray.D[p] = 0u;
  q.lpath[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// =================================================================================================
// P3 closest-hit traversal + triangle intersection, persistent waves.
//
// Each wave pulls 64 consecutive rays with one atomic, every lane walks the 64-byte-node LBVH with
// its own stack (LDS, stride 64 dwords: bank = lane for every depth, so no conflicts; or scratch when
// the tree is deeper than the LDS stack), near child first, ties to child 0.  Closest hit is the
// lexicographic minimum of (t, original primitive id).  The first `n_nodelets` nodes (breadth-first
// top of the tree) are read from an LDS copy ("nodelets") instead of L1/L2.
// MODE 0: epilogue sorts surviving paths by material class into the two-ended queue (P9).
// MODE 1: epilogue writes (t, prim, u, v) in place (debug / raster).  CULL: R6 back-face culling +
//         per-ray [tmin,tmax] from B.zw.
struct TravStack {
  int* lds;           // &stack[wave][0][lane]
};

template <bool LDS_STACK> struct Stack;
template <> struct Stack<true> {
  int* base; int sp;
  PT_DEV void init(int* lds_base) { base = lds_base; sp = 0; }
  PT_DEV void push(int v) { base[sp * 64] = v; ++sp; }
  PT_DEV int pop() { --sp; return base[sp * 64]; }
  PT_DEV bool empty() const { return sp == 0; }
};
template <> struct Stack<false> {
  int st[SCRATCH_STACK_DEPTH]; int sp;
  PT_DEV void init(int*) { sp = 0; }
  PT_DEV void push(int v) { st[sp] = v; ++sp; }
  PT_DEV int pop() { --sp; return st[sp]; }
  PT_DEV bool empty() const { return sp == 0; }
};

struct Node16 { float4 q0, q1, q2, q3; };

template <bool NODELETS>
PT_DEV Node16 load_node(const DevScene& sc, const float4* lds_nodes, int cur) {
  Node16 n;
  if (NODELETS && (uint32_t)cur < sc.n_nodelets) {
    const float4* p = lds_nodes + (size_t)cur * 4;
    n.q0 = p[0]; n.q1 = p[1]; n.q2 = p[2]; n.q3 = p[3];
  } else {
    const float4* p = sc.nodes + (size_t)cur * 4;
    n.q0 = p[0]; n.q1 = p[1]; n.q2 = p[2]; n.q3 = p[3];
  }
  return n;
}

template <int MODE, bool CULL, bool LDS_STACK, bool NODELETS>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_closest(DevScene sc, DevQueues q) {
  extern __shared__ float4 lds_raw[];
  // LDS carve: [nodelets: n_nodelets × 64 B][stacks: TRACE_WAVES × DEPTH × 64 × 4 B]
  float4* lds_nodes = lds_raw;
  int* lds_stack = reinterpret_cast<int*>(lds_raw + (NODELETS ? (size_t)sc.n_nodelets * 4 : 0));
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  if (NODELETS) {
    for (uint32_t i = threadIdx.x; i < sc.n_nodelets * 4u; i += TRACE_BLOCK) lds_nodes[i] = sc.nodes[i];
    __syncthreads();
  }
  const uint32_t n = q.cnt[CNT_RAYS];
  unsigned long long c_nodes = 0, c_tris = 0, c_rays = 0, c_hits = 0;
  for (;;) {
    const uint32_t base = wave_fetch(&q.cnt[CNT_WORK_TRACE], 64u, lane);
    if (base >= n) break;
    const uint32_t i = base + lane;
    const bool active = i < n;
    float4 A = make_float4(0, 0, 0, 0), Bq = make_float4(0, 0, 0, 0);
    if (active) { A = q.ray.A[i]; Bq = q.ray.B[i]; }
    const ray_t r = make_ray(V3(A.x, A.y, A.z), V3(A.w, Bq.x, Bq.y));
    const float tmin = CULL ? Bq.z : 0.0f;
    float best_t = CULL ? Bq.w : PT_T_INF, best_u = 0.0f, best_v = 0.0f;
    int best_prim = 0x7fffffff, best_cls = 0;
    bool found = false;
    uint32_t nv = 0, nt = 0;
    Stack<LDS_STACK> st;
    st.init(lds_stack + wave * (LDS_STACK_DEPTH * 64) + lane);
    int cur = active ? 0 : CUR_DONE;
    while (cur != CUR_DONE) {
      while (cur >= 0) {
        const Node16 nd = load_node<NODELETS>(sc, lds_nodes, cur);
        ++nv;
        float t0, t1;
        const bool h0 = box_hit(r, nd.q0.x, nd.q0.y, nd.q0.z, nd.q0.w, nd.q1.x, nd.q1.y, tmin, best_t, t0);
        const bool h1 = box_hit(r, nd.q1.z, nd.q1.w, nd.q2.x, nd.q2.y, nd.q2.z, nd.q2.w, tmin, best_t, t1);
        const int c0 = __float_as_int(nd.q3.x), c1 = __float_as_int(nd.q3.y);
        if (h0 && h1) {
          int first = c0, second = c1;
          if (t1 < t0) { first = c1; second = c0; }
          st.push(second);
          cur = first;
        } else if (h0) cur = c0;
        else if (h1) cur = c1;
        else cur = st.empty() ? CUR_DONE : st.pop();
      }
      if (cur != CUR_DONE) {
        const uint32_t code = (uint32_t)~cur;
        const uint32_t first = code & 0x0fffffffu, count = (code >> 28) + 1u;
        for (uint32_t k = first; k < first + count; ++k) {
          const float4 a = sc.tris[(size_t)k * 3 + 0], b = sc.tris[(size_t)k * 3 + 1], c = sc.tris[(size_t)k * 3 + 2];
          ++nt;
          float t, u, v;
          if (!tri_test<CULL>(r, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), t, u, v)) continue;
          const int pid = __float_as_int(a.w);
          if (t > tmin && (t < best_t || (t == best_t && pid < best_prim))) {
            best_t = t; best_u = u; best_v = v; best_prim = pid; best_cls = __float_as_int(b.w); found = true;
          }
        }
        cur = st.empty() ? CUR_DONE : st.pop();
      }
    }
    c_nodes += nv; c_tris += nt; c_rays += active ? 1u : 0u; c_hits += found ? 1u : 0u;
    // ---- epilogue ------------------------------------------------------------------------------
    if (MODE == 1) {
      if (active) q.hit[i] = make_float4(found ? best_t : -1.0f, __int_as_float(found ? best_prim : -1), best_u, best_v);
    } else {
      // P9: per-wave material-class sort.  Class 0 (Lambert) grows up from slot 0, class 1 (GGX) grows
      // down from slot cap-1; misses leave the wavefront here (background radiance is 0).
      float4 Cq = make_float4(0, 0, 0, 0); uint32_t Dq = 0;
      if (found) { Cq = q.ray.C[i]; Dq = q.ray.D[i]; }
      const uint64_t m0 = __ballot(found && best_cls == 0);
      const uint64_t m1 = __ballot(found && best_cls != 0);
      uint32_t pos = 0;
      if (m0) { const uint32_t s = wave_alloc(&q.cnt[CNT_SORT0], m0, lane); if (found && best_cls == 0) pos = s; }
      if (m1) { const uint32_t s = wave_alloc(&q.cnt[CNT_SORT1], m1, lane); if (found && best_cls != 0) pos = q.cap - 1u - s; }
      if (found) {
        q.sorted.A[pos] = A; q.sorted.B[pos] = Bq; q.sorted.C[pos] = Cq; q.sorted.D[pos] = Dq;
        q.sorted.H[pos] = make_float4(best_t, __int_as_float(best_prim), best_u, best_v);
      }
    }
  }
  c_nodes = wave_sum(c_nodes); c_tris = wave_sum(c_tris); c_rays = wave_sum(c_rays); c_hits = wave_sum(c_hits);
  if (lane == 0 && c_rays) {
    atomicAdd(&q.stats[ST_NODES_C], c_nodes); atomicAdd(&q.stats[ST_TRIS_C], c_tris);
    atomicAdd(&q.stats[ST_SEGMENTS], c_rays); atomicAdd(&q.stats[ST_HITS], c_hits);
  }
}

// =================================================================================================
// P4 any-hit traversal for the NEE shadow rays; unoccluded rays add their contribution to the
// path's radiance word (single owner: one shadow ray per path per bounce).
template <bool LDS_STACK, bool NODELETS, bool DEBUG_OUT>
__global__ __launch_bounds__(TRACE_BLOCK) void k_trace_any(DevScene sc, DevQueues q, uint8_t* debug_out) {
  extern __shared__ float4 lds_raw[];
  float4* lds_nodes = lds_raw;
  int* lds_stack = reinterpret_cast<int*>(lds_raw + (NODELETS ? (size_t)sc.n_nodelets * 4 : 0));
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  if (NODELETS) {
    for (uint32_t i = threadIdx.x; i < sc.n_nodelets * 4u; i += TRACE_BLOCK) lds_nodes[i] = sc.nodes[i];
    __syncthreads();
  }
  const uint32_t n = q.cnt[CNT_SHADOW];
  unsigned long long c_nodes = 0, c_tris = 0, c_rays = 0;
  for (;;) {
    const uint32_t base = wave_fetch(&q.cnt[CNT_WORK_SHADOW], 64u, lane);
    if (base >= n) break;
    const uint32_t i = base + lane;
    const bool active = i < n;
    float4 A = make_float4(0, 0, 0, 0), Bq = make_float4(0, 0, 0, 0);
    if (active) { A = q.shadow.A[i]; Bq = q.shadow.B[i]; }
    const ray_t r = make_ray(V3(A.x, A.y, A.z), V3(A.w, Bq.x, Bq.y));
    const float tmax = Bq.z;
    bool occluded = false;
    uint32_t nv = 0, nt = 0;
    Stack<LDS_STACK> st;
    st.init(lds_stack + wave * (LDS_STACK_DEPTH * 64) + lane);
    int cur = active ? 0 : CUR_DONE;
    while (cur != CUR_DONE) {
      while (cur >= 0) {
        const Node16 nd = load_node<NODELETS>(sc, lds_nodes, cur);
        ++nv;
        float t0, t1;
        const bool h0 = box_hit(r, nd.q0.x, nd.q0.y, nd.q0.z, nd.q0.w, nd.q1.x, nd.q1.y, 0.0f, tmax, t0);
        const bool h1 = box_hit(r, nd.q1.z, nd.q1.w, nd.q2.x, nd.q2.y, nd.q2.z, nd.q2.w, 0.0f, tmax, t1);
        const int c0 = __float_as_int(nd.q3.x), c1 = __float_as_int(nd.q3.y);
        if (h0 && h1) {
          int first = c0, second = c1;
          if (t1 < t0) { first = c1; second = c0; }
          st.push(second);
          cur = first;
        } else if (h0) cur = c0;
        else if (h1) cur = c1;
        else cur = st.empty() ? CUR_DONE : st.pop();
      }
      if (cur != CUR_DONE) {
        const uint32_t code = (uint32_t)~cur;
        const uint32_t first = code & 0x0fffffffu, count = (code >> 28) + 1u;
        for (uint32_t k = first; k < first + count; ++k) {
          const float4 a = sc.tris[(size_t)k * 3 + 0], b = sc.tris[(size_t)k * 3 + 1], c = sc.tris[(size_t)k * 3 + 2];
          ++nt;
          float t, u, v;
          if (tri_test<false>(r, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), t, u, v) && t > 0.0f && t < tmax) {
            occluded = true;
            break;
          }
        }
        cur = (occluded || st.empty()) ? CUR_DONE : st.pop();
      }
    }
    c_nodes += nv; c_tris += nt; c_rays += active ? 1u : 0u;
    if (DEBUG_OUT) {
      if (active) debug_out[i] = occluded ? 1 : 0;
    } else if (active && !occluded) {
      const float4 Cq = q.shadow.C[i];
      const uint32_t path = __float_as_uint(Bq.w);
      float4 L = q.lpath[path];
      L.x = L.x + Cq.x; L.y = L.y + Cq.y; L.z = L.z + Cq.z;
      q.lpath[path] = L;
    }
  }
  c_nodes = wave_sum(c_nodes); c_tris = wave_sum(c_tris); c_rays = wave_sum(c_rays);
  if (lane == 0 && c_rays) {
    atomicAdd(&q.stats[ST_NODES_A], c_nodes); atomicAdd(&q.stats[ST_TRIS_A], c_tris); atomicAdd(&q.stats[ST_SHADOW], c_rays);
  }
}

// =================================================================================================
// P5–P8 shading: surface reconstruction, emission with MIS, next-event estimation, BSDF sampling,
// Russian roulette.  Input is the class-sorted queue, so a wave is (boundary waves aside) uniform
// in material class and the GGX / Lambert branches below do not diverge.
__global__ __launch_bounds__(256) void k_shade(DevScene sc, DevFrame fr, DevQueues q) {
  const uint32_t lane = lane_id();
  const uint32_t n0 = q.cnt[CNT_SORT0], n1 = q.cnt[CNT_SORT1];
  const uint32_t w0 = (n0 + 63u) >> 6, w1 = (n1 + 63u) >> 6;
  for (;;) {
    const uint32_t wv = wave_fetch(&q.cnt[CNT_WORK_SHADE], 1u, lane);
    if (wv >= w0 + w1) break;
    bool valid; uint32_t slot;
    if (wv < w0) { const uint32_t k = wv * 64u + lane; valid = k < n0; slot = k; }
    else { const uint32_t k = (wv - w0) * 64u + lane; valid = k < n1; slot = q.cap - 1u - k; }
    bool alive = false, has_shadow = false;
    float4 oA, oB, oC; uint32_t oD = 0;              // continuation ray
    float4 sA, sB, sC;                               // shadow ray
    oA = oB = oC = sA = sB = sC = make_float4(0, 0, 0, 0);
    if (valid) {
      const float4 A = q.sorted.A[slot], Bq = q.sorted.B[slot], Cq = q.sorted.C[slot], H = q.sorted.H[slot];
      const uint32_t b = q.sorted.D[slot];
      const v3 d = V3(A.w, Bq.x, Bq.y);
      v3 T = V3(Bq.z, Bq.w, Cq.x);
      const float prev_pdf = Cq.y;
      const uint32_t path = __float_as_uint(Cq.z), key = __float_as_uint(Cq.w);
      const float ht = H.x, hu = H.z, hv = H.w;
      const uint32_t prim = (uint32_t)__float_as_int(H.y);
      // ---- P5 surface reconstruction: gather the three R1 vertex records ----
      const uint32_t i0 = sc.widx[prim * 3 + 0], i1 = sc.widx[prim * 3 + 1], i2 = sc.widx[prim * 3 + 2];
      const float* va = sc.wverts + (size_t)i0 * 12; const float* vb = sc.wverts + (size_t)i1 * 12; const float* vc = sc.wverts + (size_t)i2 * 12;
      const v3 Pa = V3(va[0], va[1], va[2]), Pb = V3(vb[0], vb[1], vb[2]), Pc = V3(vc[0], vc[1], vc[2]);
      const v3 Na = V3(va[3], va[4], va[5]), Nb = V3(vb[3], vb[4], vb[5]), Nc = V3(vc[3], vc[4], vc[5]);
      const float hw = 1.0f - hu - hv;
      const v3 P = V3(pt_fma(Pc.x, hv, pt_fma(Pb.x, hu, Pa.x * hw)), pt_fma(Pc.y, hv, pt_fma(Pb.y, hu, Pa.y * hw)), pt_fma(Pc.z, hv, pt_fma(Pb.z, hu, Pa.z * hw)));
      v3 ng = normalize3(cross3(Pb - Pa, Pc - Pa));
      v3 ns = normalize3(V3(pt_fma(Nc.x, hv, pt_fma(Nb.x, hu, Na.x * hw)), pt_fma(Nc.y, hv, pt_fma(Nb.y, hu, Na.y * hw)), pt_fma(Nc.z, hv, pt_fma(Nb.z, hu, Na.z * hw))));
      const v3 wo = -d;
      const bool front = dot3(ng, wo) > 0.0f;
      if (dot3(ns, ng) < 0.0f) ns = -ns;
      if (!front) { ng = -ng; ns = -ns; }
      if (!(dot3(ns, wo) > 0.0f)) ns = ng;
      const int mat = sc.tri_mat[prim];
      const float4 M0 = sc.mats[mat * 3 + 0], M1 = sc.mats[mat * 3 + 1];
      float4 L = q.lpath[path];
      bool Ldirty = false;
      // ---- emission (one-sided), MIS against next-event estimation ----
      const int li = sc.prim_light[prim];
      if (li >= 0 && front) {
        const float4 l0 = sc.lights[li * 5 + 0], l1 = sc.lights[li * 5 + 1], l4 = sc.lights[li * 5 + 4];
        float wgt = 1.0f;
        if (b > 0u) {
          const float cosl = dot3(ng, wo);
          const float pl = (l1.w * (ht * ht)) / (l0.w * cosl);
          const float pb2 = prev_pdf * prev_pdf;
          wgt = pb2 / pt_fma(pl, pl, pb2);
        }
        L.x = pt_fma(T.x * l4.x, wgt, L.x); L.y = pt_fma(T.y * l4.y, wgt, L.y); L.z = pt_fma(T.z * l4.z, wgt, L.z);
        Ldirty = true;
      }
      if (Ldirty) q.lpath[path] = L;
      if ((int)b < fr.max_bounces) {
        const bsdf_t bs = make_bsdf(V3(M0.x, M0.y, M0.z), M0.w, M1.w);
        v3 tx, ty; onb(ns, tx, ty);
        const v3 wol = V3(dot3(tx, wo), dot3(ty, wo), dot3(ns, wo));
        const float ps = spec_prob(bs, fmax2(wol.z, 1e-4f));
        const v3 porg = vfma(ng, sc.ray_eps, P);
        const uint32_t rb = b + 1u;
        // ---- P7 next-event estimation ----
        if (sc.n_lights > 0u) {
          const float u0 = rng_f(key, rb, 0), r1 = rng_f(key, rb, 1), r2 = rng_f(key, rb, 2);
          uint32_t lo = 0, hi = sc.n_lights - 1u;
          while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (sc.cdf[mid] > u0) hi = mid; else lo = mid + 1u; }
          const float4 l0 = sc.lights[lo * 5 + 0], l1 = sc.lights[lo * 5 + 1], l2 = sc.lights[lo * 5 + 2], l3 = sc.lights[lo * 5 + 3], l4 = sc.lights[lo * 5 + 4];
          const float su = pt_sqrt(r1);
          const float bu = su * (1.0f - r2), bv = su * r2;
          const v3 y = vfma(V3(l2.x, l2.y, l2.z), bv, vfma(V3(l1.x, l1.y, l1.z), bu, V3(l0.x, l0.y, l0.z)));
          const v3 dv = y - P;
          const float dist2 = dot3(dv, dv);
          if (dist2 > 0.0f) {
            const float dist = pt_sqrt(dist2);
            const v3 wi = dv * (1.0f / dist);
            const float cosl = -dot3(V3(l3.x, l3.y, l3.z), wi);
            const v3 wil = V3(dot3(tx, wi), dot3(ty, wi), dot3(ns, wi));
            if (cosl > 0.0f && wil.z > 0.0f && dot3(ng, wi) > 0.0f) {
              const float pl = (l1.w * dist2) / (l0.w * cosl);
              v3 f; float pb; bsdf_eval(bs, wol, wil, ps, f, pb);
              const float pl2 = pl * pl;
              const float wgt = pl2 / pt_fma(pb, pb, pl2);
              const float k = (wil.z * wgt) / pl;
              has_shadow = true;
              sA = make_float4(porg.x, porg.y, porg.z, wi.x);
              sB = make_float4(wi.y, wi.z, dist * 0.999f, __uint_as_float(path));
              sC = make_float4(T.x * f.x * l4.x * k, T.y * f.y * l4.y * k, T.z * f.z * l4.z * k, 0.0f);
            }
          }
        }
        // ---- P6 continuation + P8 Russian roulette ----
        const float ul = rng_f(key, rb, 3), s1 = rng_f(key, rb, 4), s2 = rng_f(key, rb, 5);
        v3 wil;
        bool ok = bsdf_sample(bs, wol, ps, ul, s1, s2, wil);
        v3 wi = V3(0, 0, 0); float pdf = 0.0f;
        if (ok) {
          wi = vfma(tx, wil.x, vfma(ty, wil.y, ns * wil.z));
          ok = dot3(ng, wi) > 0.0f;
        }
        if (ok) {
          v3 f; bsdf_eval(bs, wol, wil, ps, f, pdf);
          ok = pdf > 0.0f;
          if (ok) {
            const float k = wil.z / pdf;
            T = V3(T.x * f.x * k, T.y * f.y * k, T.z * f.z * k);
          }
        }
        if (ok && rb >= PT_RR_START) {
          const float qq = max3c(T);
          ok = qq > 0.0f;
          if (ok) {
            const float pr = fmin2(fmax2(qq, PT_RR_PMIN), 1.0f);
            const float ur = rng_f(key, rb, 6);
            ok = !(ur >= pr);
            if (ok) T = V3(T.x / pr, T.y / pr, T.z / pr);
          }
        }
        if (ok) {
          alive = true;
          oA = make_float4(porg.x, porg.y, porg.z, wi.x);
          oB = make_float4(wi.y, wi.z, T.x, T.y);
          oC = make_float4(T.z, pdf, __uint_as_float(path), __uint_as_float(key));
          oD = rb;
        }
      }
    }
    // ---- compaction: ballot + mbcnt prefix + one atomic per wave and queue ----
    const uint64_t ms = __ballot(has_shadow);
    if (ms) {
      const uint32_t s = wave_alloc(&q.cnt[CNT_SHADOW], ms, lane);
      if (has_shadow) { q.shadow.A[s] = sA; q.shadow.B[s] = sB; q.shadow.C[s] = sC; }
    }
    const uint64_t ma = __ballot(alive);
    if (ma) {
      const uint32_t s = wave_alloc(&q.cnt[CNT_NEXT], ma, lane);
      if (alive) { q.ray.A[s] = oA; q.ray.B[s] = oB; q.ray.C[s] = oC; q.ray.D[s] = oD; }
    }
  }
}

// =================================================================================================
// P10 accumulate: per owned pixel, add this batch's per-path radiance in sample-index order.
__global__ __launch_bounds__(256) void k_accumulate(DevFrame fr, const float4* lpath, float4* accum, uint32_t n_samples) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= fr.n_owned) return;
  float4 a = accum[j];
  for (uint32_t s = 0; s < n_samples; ++s) {
    const float4 L = lpath[(size_t)s * fr.n_owned + j];
    a.x = a.x + L.x; a.y = a.y + L.y; a.z = a.z + L.z;
  }
  accum[j] = a;
}

// sum / spp → full-frame RGBA32F, alpha 1 (path) or the shaded alpha (raster-compat)
__global__ __launch_bounds__(256) void k_resolve(DevFrame fr, const float4* accum, float4* radiance, float spp, int raster) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= fr.n_owned) return;
  const float4 a = accum[j];
  float4 o;
  if (raster) o = a;
  else { o.x = a.x / spp; o.y = a.y / spp; o.z = a.z / spp; o.w = 1.0f; }
  radiance[fr.owned[j]] = o;
}

// =================================================================================================
// Raster-compat shading (R6–R8): the reference's deferred Blinn-Phong result at the primary hit.
//   fragment.glsl:24-27 with the flat normal texel (0.5,0.5,1): N = normalize(interpolated normal)
//   lighting.glsl:25-28: V = normalize(cam − P), L = V;  BlinnPhong.lib.glsl:4-10
__global__ __launch_bounds__(256) void k_shade_raster(DevScene sc, DevCamera cam, DevFrame fr, DevQueues q, float4* accum) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= fr.n_owned) return;
  const float4 H = q.hit[j];
  float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // G-buffer clear → colour 0
  const int prim = __float_as_int(H.y);
  if (prim >= 0) {
    const float hu = H.z, hv = H.w, hw = 1.0f - hu - hv;
    const uint32_t i0 = sc.widx[prim * 3 + 0], i1 = sc.widx[prim * 3 + 1], i2 = sc.widx[prim * 3 + 2];
    const float* va = sc.wverts + (size_t)i0 * 12; const float* vb = sc.wverts + (size_t)i1 * 12; const float* vc = sc.wverts + (size_t)i2 * 12;
    const v3 P = V3(pt_fma(vc[0], hv, pt_fma(vb[0], hu, va[0] * hw)), pt_fma(vc[1], hv, pt_fma(vb[1], hu, va[1] * hw)), pt_fma(vc[2], hv, pt_fma(vb[2], hu, va[2] * hw)));
    const v3 N = normalize3(V3(pt_fma(vc[3], hv, pt_fma(vb[3], hu, va[3] * hw)), pt_fma(vc[4], hv, pt_fma(vb[4], hu, va[4] * hw)), pt_fma(vc[5], hv, pt_fma(vb[5], hu, va[5] * hw))));
    const int mat = sc.tri_mat[prim];
    const float4 M0 = sc.mats[mat * 3 + 0], M2 = sc.mats[mat * 3 + 2];
    const v3 V = normalize3(V3(cam.pos[0], cam.pos[1], cam.pos[2]) - P);
    const v3 Hh = normalize3(V + V);
    const float ndv = fmax2(dot3(N, V), 0.0f), ndh = fmax2(dot3(N, Hh), 0.0f);
    const float s2 = ndh * ndh, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8, s32 = s16 * s16, spec = s32 * s32;
    o = make_float4(pt_fma(M0.x, ndv, spec), pt_fma(M0.y, ndv, spec), pt_fma(M0.z, ndv, spec), pt_fma(M2.x, ndv, spec));
  }
  accum[j] = o;
}

// =================================================================================================
// R9 tonemap: ACES fit (matrices as GLSL reads them: column-major, i.e. transposed — SURVEY §3.4),
// gamma 2.2, clamp, UNORM8.  16×16 workgroups like TonemapperSystem.cpp:18,131-133.
PT_DEV float rrt_odt(float c) {
  const float num = c * (c + 0.0245786f) - 0.000090537f;
  const float den = c * (0.983729f * c + 0.4329510f) + 0.238081f;
  return num / den;
}
__global__ __launch_bounds__(256) void k_tonemap(const float4* radiance, uint32_t* out, int w, int h) {
  const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
  if (x >= w || y >= h) return;
  const float4 c = radiance[(size_t)y * w + x];
  const float ir = 0.59719f * c.x + 0.07600f * c.y + 0.02840f * c.z;
  const float ig = 0.35458f * c.x + 0.90834f * c.y + 0.13383f * c.z;
  const float ib = 0.04823f * c.x + 0.01566f * c.y + 0.83777f * c.z;
  const float fr = rrt_odt(ir), fg = rrt_odt(ig), fb = rrt_odt(ib);
  const float orr = 1.60475f * fr + -0.10208f * fg + -0.00327f * fb;
  const float og = -0.53108f * fr + 1.10813f * fg + -0.07276f * fb;
  const float ob = -0.07367f * fr + -0.00605f * fg + 1.07602f * fb;
  const float v4[4] = {orr, og, ob, c.w};
  uint32_t packed = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v = v4[k];
    if (k < 3) v = pt_pow(fmax2(v, 0.0f), 1.0f / 2.2f);
    v = fmin2(fmax2(v, 0.0f), 1.0f);
    packed |= ((uint32_t)(int)(v * 255.0f + 0.5f)) << (8 * k);
  }
  out[(size_t)y * w + x] = packed;
}

// =================================================================================================
// launchers
static size_t trace_lds_bytes(const LaunchCfg& cfg, const DevScene& sc, bool nodelets) {
  size_t b = nodelets ? (size_t)sc.n_nodelets * 64 : 0;
  if (cfg.lds_stack) b += (size_t)TRACE_WAVES * LDS_STACK_DEPTH * 64 * 4;
  return b;
}

void pt_launch_set_counts(hipStream_t s, const DevQueues& q, uint32_t n_rays, uint32_t n_shadow) { hipLaunchKernelGGL(k_set_counts, dim3(1), dim3(64), 0, s, q.cnt, n_rays, n_shadow); }
void pt_launch_advance(hipStream_t s, const DevQueues& q) { hipLaunchKernelGGL(k_advance, dim3(1), dim3(64), 0, s, q.cnt); }

void pt_launch_raygen(hipStream_t s, const DevCamera& cam, const DevFrame& fr, const DevQueues& q, uint32_t first_sample, uint32_t n_samples, bool raster) {
  const uint32_t n_paths = fr.n_owned * n_samples;
  const dim3 grid((n_paths + 255u) / 256u);
  if (raster) hipLaunchKernelGGL(k_raygen<true>, grid, dim3(256), 0, s, cam, fr, q, first_sample, n_paths);
  else hipLaunchKernelGGL(k_raygen<false>, grid, dim3(256), 0, s, cam, fr, q, first_sample, n_paths);
}

template <int MODE, bool CULL>
static void launch_tc(hipStream_t s, const LaunchCfg& cfg, const DevScene& sc, const DevQueues& q) {
  const bool nodelets = sc.n_nodelets > 0;
  const dim3 grid((unsigned)(cfg.n_cu * cfg.trace_blocks_per_cu));
  const size_t lds = trace_lds_bytes(cfg, sc, nodelets);
  if (cfg.lds_stack) {
    if (nodelets) hipLaunchKernelGGL((k_trace_closest<MODE, CULL, true, true>), grid, dim3(TRACE_BLOCK), lds, s, sc, q);
    else hipLaunchKernelGGL((k_trace_closest<MODE, CULL, true, false>), grid, dim3(TRACE_BLOCK), lds, s, sc, q);
  } else {
    if (nodelets) hipLaunchKernelGGL((k_trace_closest<MODE, CULL, false, true>), grid, dim3(TRACE_BLOCK), lds, s, sc, q);
    else hipLaunchKernelGGL((k_trace_closest<MODE, CULL, false, false>), grid, dim3(TRACE_BLOCK), lds, s, sc, q);
  }
}
void pt_launch_trace_closest(hipStream_t s, const LaunchCfg& cfg, const DevScene& sc, const DevQueues& q, int mode) {
  if (mode == 0) launch_tc<0, false>(s, cfg, sc, q);
  else if (mode == 1) launch_tc<1, false>(s, cfg, sc, q);
  else launch_tc<1, true>(s, cfg, sc, q);
}

void pt_launch_trace_any(hipStream_t s, const LaunchCfg& cfg, const DevScene& sc, const DevQueues& q, uint8_t* debug_out) {
  const bool nodelets = sc.n_nodelets > 0;
  const dim3 grid((unsigned)(cfg.n_cu * cfg.trace_blocks_per_cu));
  const size_t lds = trace_lds_bytes(cfg, sc, nodelets);
#define TA(L, N, D) hipLaunchKernelGGL((k_trace_any<L, N, D>), grid, dim3(TRACE_BLOCK), lds, s, sc, q, debug_out)
  if (debug_out) {
    if (cfg.lds_stack) { if (nodelets) TA(true, true, true); else TA(true, false, true); }
    else { if (nodelets) TA(false, true, true); else TA(false, false, true); }
  } else {
    if (cfg.lds_stack) { if (nodelets) TA(true, true, false); else TA(true, false, false); }
    else { if (nodelets) TA(false, true, false); else TA(false, false, false); }
  }
#undef TA
}

void pt_launch_shade(hipStream_t s, const LaunchCfg& cfg, const DevScene& sc, const DevFrame& fr, const DevQueues& q) {
  hipLaunchKernelGGL(k_shade, dim3((unsigned)(cfg.n_cu * 8)), dim3(256), 0, s, sc, fr, q);
}
void pt_launch_accumulate(hipStream_t s, const DevFrame& fr, const DevQueues& q, float4* accum, uint32_t n_samples) {
  hipLaunchKernelGGL(k_accumulate, dim3((fr.n_owned + 255u) / 256u), dim3(256), 0, s, fr, (const float4*)q.lpath, accum, n_samples);
}
void pt_launch_shade_raster(hipStream_t s, const DevScene& sc, const DevCamera& cam, const DevFrame& fr, const DevQueues& q, float4* accum) {
  hipLaunchKernelGGL(k_shade_raster, dim3((fr.n_owned + 255u) / 256u), dim3(256), 0, s, sc, cam, fr, q, accum);
}
void pt_launch_resolve(hipStream_t s, const DevFrame& fr, const float4* accum, float4* radiance, float spp, bool raster) {
  hipLaunchKernelGGL(k_resolve, dim3((fr.n_owned + 255u) / 256u), dim3(256), 0, s, fr, accum, radiance, spp, raster ? 1 : 0);
}
void pt_launch_tonemap(hipStream_t s, const float4* radiance, uint32_t* rgba8, int w, int h) {
  hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((w + 15) / 16), (unsigned)((h + 15) / 16)), dim3(256), 0, s, radiance, rgba8, w, h);
}
