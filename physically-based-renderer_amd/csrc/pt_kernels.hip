// pt_kernels.hip — the wavefront path tracer's kernels for gfx950 (wave64).
//
// Stages:  P1 k_raygen · P3 k_trace_closest · P9 compaction (and, optionally, material sort) inside
// k_shade · P5–P8 k_shade · P4 k_trace_any · P10 k_accumulate / k_resolve · R8 k_shade_raster · R9 k_tonemap;
// queue bookkeeping k_set_counts / k_scan.
//
// Design rules that came out of the profiles (profiles/r01_v1_* ... profiles/r03_*):
//  * a single device word takes ≈88 atomics/µs, so NOTHING does one atomic per wave-iteration: trace
//    waves pull chunks of <= 512 rays (their first chunk is static, later ones cost one atomic) and refill idle lanes
//    from their private chunk; the shade kernel needs no atomic at all: a queue is cut into segments, one wave owns a
//    segment and compacts its outputs into the same segment of the output arrays (ptc_internal.h, "SEGMENTED queues");
//  * k_shade runs at the rate of the CUs' vector-memory path: whatever adds loads to it loses, the material sort included
//    (it stays as k_shade<true>, off by default);
//  * traversal has poor lane utilisation when a wave waits for its slowest ray, so finished lanes are
//    refilled in place (persistent while-while with dynamic fetch);
//  * every queue access is a 16-byte lane (dwordx4); no float atomics anywhere — each per-path /
//    per-pixel word has one owner, so results do not depend on scheduling.  MFMA is unused: there is
//    no dense contraction on this path.
#include "pt_device.h"
#include "ptc_internal.h"

#ifndef TRACE_BLOCK
#define TRACE_BLOCK 256
#endif
#ifndef TRACE_MIN_WAVES
#define TRACE_MIN_WAVES 8         // waves per SIMD the register allocator must leave room for (launch bounds): 64 VGPRs; 8 instead of 7 blocks per CU: +4 %
#endif
#define TRACE_WAVES (TRACE_BLOCK / 64)
#ifndef TRACE_CHUNK
#define TRACE_CHUNK 512u          // rays per work-fetch atomic
#endif
#ifndef TRACE_CHUNK_SHORT
#define TRACE_CHUNK_SHORT 64u     // rays per chunk of a SHORT queue (less than TRACE_CHUNK rays per wave of the grid): dealt round-robin, no atomics (struct Reservoir)
#endif
#ifndef TRACE_REFILL_IDLE
#define TRACE_REFILL_IDLE 1       // idle lanes that make the wave hand out prepared rays (round 4: a hand-out is ~25 vector instructions; the round-3 refill cost ~85 at 0.27 lane utilisation and waited for 12)
#endif
#ifndef TRACE_RING
#define TRACE_RING 64             // prepared rays a wave holds in LDS (struct RayRing)
#endif
#define RING_FIELDS 5             // slot, inv.x, inv.y, inv.z, octant
#ifndef TRACE_NODE_MIN
#define TRACE_NODE_MIN 40         // leave the node loop when fewer lanes than this are still at interior nodes
#endif                            // while others wait at a leaf (keeps both phases well populated; 24: -6 %, 32: -1.5 %, 48: -3 %, 56: -18 %)
#ifndef TRACE_NODE_MIN_ANY
#define TRACE_NODE_MIN_ANY TRACE_NODE_MIN      // the same threshold in k_trace_any (swept separately in round 4: profiles/r04_trace_variants.txt)
#endif
#ifndef SHADE_BLOCK
#define SHADE_BLOCK 256           // a block only shares the staged light / material / row-cdf tables; its waves never synchronise after that.  Round 3a (FLAT gathers): 64: +19 % kernel
                                  // time, 128: +1 %, 256: 0, 512: -2 %; round 3b (global gathers, 64 segments per CU): 128 / 256 / 512 = 0.1208 / 0.1180 / 0.1178 s per 6 steps on the atrium,
                                  // 0.1677 / 0.1662 / 0.1831 on the textured atrium — smaller blocks free their wave slots sooner (profiles/r03_shade_segments.txt)
#endif
#ifndef SHADE_MIN_WAVES
#define SHADE_MIN_WAVES 4         // <= 128 VGPRs: four independent waves per SIMD
#endif
#define SHADE_WAVES (SHADE_BLOCK / 64)
#define SHADE_LDS_LIGHTS 64       // emitter table and material table are staged in LDS when they fit
#define SHADE_LDS_MATS 64
#define SHADE_LDS_ENV_ROWS 2048   // rows of an environment map whose row cdf is staged in LDS (8 KB)
#define HIT_CLASS_SHIFT 28

// ---- small helpers --------------------------------------------------------------------------------
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
// bookkeeping kernels: per-segment counts -> chunk prefix (ptc_internal.h, "SEGMENTED queues")
//
// One block of SCAN_BLOCK threads.  The trace kernels hand out rays in chunks that never straddle a segment; the chunk size follows the
// queue's length: TRACE_CHUNK rays when the queue is long, down to one wave's worth when it is short (about one chunk, a static one, per wave of
// the persistent grid), so that a small late-bounce queue still spreads over the whole chip.
#define SCAN_BLOCK 1024
PT_DEV uint32_t block_sum_1024(uint32_t v, uint32_t* s_w /*[16]*/, uint32_t& excl) {   // returns the block total, excl = exclusive prefix of this thread
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o); if ((int)lane >= o) inc += t; }
  __syncthreads();                       // s_w may still be read from a previous call
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (uint32_t w = 0; w < SCAN_BLOCK / 64; ++w) { const uint32_t t = s_w[w]; if (w < wave) before += t; total += t; }
  excl = before + inc - v;
  return total;
}
// chunk prefix of one queue: pre[s] = chunks in the segments before s, pre[n_seg] = all chunks; hdr = (total records, chunk size, chunks)
PT_DEV void scan_chunks(const uint32_t* seg, uint32_t n_seg, uint32_t* pre, uint32_t* hdr, uint32_t trace_waves, uint32_t* s_w) {
  const uint32_t per = (n_seg + SCAN_BLOCK - 1u) / SCAN_BLOCK;       // consecutive segments per thread (<= 16)
  const uint32_t s0 = threadIdx.x * per, s1 = s0 + per < n_seg ? s0 + per : n_seg;
  uint32_t mine = 0, dummy;
  for (uint32_t sg = s0; sg < s1; ++sg) mine += seg[sg];
  const uint32_t total = block_sum_1024(mine, s_w, dummy);
  // TRACE_CHUNK rays per chunk when the queue is long (a wave's first chunk is static, the rest go through the work counter), TRACE_CHUNK_SHORT when it holds
  // less than TRACE_CHUNK rays per wave of the grid: those are dealt round-robin (wave w owns chunks w, w + W, w + 2W, ...: no atomic at all).  Round 3 cut a
  // short queue into ONE chunk per wave — 512 consecutive rays of a 1080p x 1 spp frame are one 32x16-pixel patch, patches differ by 3x in cost and nothing
  // balanced them: a launch took as long as its most expensive patch (0.74 ms for 2 M camera rays, 0.38 ms for the 100 k rays of the last bounce).  Eight chunks
  // of 64 from eight places of the image average out.  (Round 3a's two chunks per wave THROUGH the counter cost 32 k atomics at 88 per us.)
  const uint32_t per_wave = (total + trace_waves - 1u) / trace_waves;
  const uint32_t chunk = per_wave >= TRACE_CHUNK ? TRACE_CHUNK : TRACE_CHUNK_SHORT;
  uint32_t chunks = 0;
  for (uint32_t sg = s0; sg < s1; ++sg) chunks += (seg[sg] + chunk - 1u) / chunk;
  uint32_t run;
  const uint32_t all = block_sum_1024(chunks, s_w, run);
  for (uint32_t sg = s0; sg < s1; ++sg) { pre[sg] = run; run += (seg[sg] + chunk - 1u) / chunk; }
  if (threadIdx.x == 0) { pre[n_seg] = all; hdr[0] = total; hdr[1] = chunk; hdr[2] = all; }
}
// identity layout: n_rays rays at the slots [0, n_rays) of ray[0], n_shadow shadow rays at [0, n_shadow) (k_raygen, test hooks)
__global__ __launch_bounds__(SCAN_BLOCK) void k_set_counts(DevQueues q, uint32_t n_rays, uint32_t n_shadow, uint32_t trace_waves) {
  __shared__ uint32_t s_w[SCAN_BLOCK / 64];
  for (uint32_t sg = threadIdx.x; sg < q.n_seg; sg += SCAN_BLOCK) {
    const uint64_t lo = (uint64_t)sg * q.seg_len;
    q.seg_ray[0][sg] = lo >= n_rays ? 0u : (n_rays - lo > q.seg_len ? q.seg_len : (uint32_t)(n_rays - lo));
    q.seg_ray[1][sg] = 0u;
    q.seg_sh[sg] = lo >= n_shadow ? 0u : (n_shadow - lo > q.seg_len ? q.seg_len : (uint32_t)(n_shadow - lo));
  }
  __syncthreads();
  scan_chunks(q.seg_ray[0], q.n_seg, q.pre_ray, &q.cnt[CNT_RAY_TOTAL], trace_waves, s_w);
  scan_chunks(q.seg_sh, q.n_seg, q.pre_sh, &q.cnt[CNT_SH_TOTAL], trace_waves, s_w);
  if (threadIdx.x == 0) { q.cnt[CNT_WORK_TRACE] = 0; q.cnt[CNT_WORK_SHADOW] = 0; }
}
// behind k_shade: the rays it wrote to ray[qi_next] are the next bounce's queue, its shadow rays this bounce's shadow queue
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan(DevQueues q, int qi_next, uint32_t trace_waves) {
  __shared__ uint32_t s_w[SCAN_BLOCK / 64];
  scan_chunks(q.seg_ray[qi_next], q.n_seg, q.pre_ray, &q.cnt[CNT_RAY_TOTAL], trace_waves, s_w);
  scan_chunks(q.seg_sh, q.n_seg, q.pre_sh, &q.cnt[CNT_SH_TOTAL], trace_waves, s_w);
  if (threadIdx.x == 0) { q.cnt[CNT_WORK_TRACE] = 0; q.cnt[CNT_WORK_SHADOW] = 0; }
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
  const RayQ& r = q.ray[0];
  r.A[p] = make_float4(cam.pos[0], cam.pos[1], cam.pos[2], d.x);
  r.B[p] = make_float4(d.y, d.z, bz, bw);
  r.C[p] = make_float4(1.0f, 0.0f, __uint_as_float(p), __uint_as_float(key));
  q.lpath[p] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// =================================================================================================
// traversal machinery shared by the closest-hit and any-hit kernels
//
// 8-wide BVH in one array of 16-byte units: 64-byte nodes, 48-byte triangle records (layout: ptc_scene.cpp); addresses are
// unit indices.  Both kernels are bound by vector-instruction issue (0.81 / 0.73 of the measured ceiling; 42 % of the
// wave-cycles wait for an issue slot), so the design minimises instructions per ray: fat nodes (12 visits per ray instead of 16),
// no sort, one stack entry per node instead of one per child, and everything that raises lane utilisation (lane refill, phase
// thresholds, helper lanes in the leaf phase).  Every lane gathers its own node with four 16-byte loads; the top of the tree
// (breadth-first prefix of the unit array) is staged in LDS once per block.  The alternative fetch — a wave loads its 64 nodes
// cooperatively, four adjacent lanes per node, straight into LDS (global_load_lds_dwordx4) — needs a quarter of the address lookups
// (tools/gather_bench.hip: 64 instead of 45 GB/s per CU) but was 13 % slower here (commit 8dfd8eb has it): its 4-KiB stage per wave
// costs occupancy, and the gather rate is not what binds.
// A ray keeps ONE group of pending interior children in
// registers — (block address, hits<<8 | imask): the slots of one node that were hit and not yet entered — and the
// older groups on a per-lane stack of 8-byte entries.  The first `L` entries live in LDS (stride 64 lanes:
// ds_read/write_b64 is conflict-free at every mix of depths), deeper entries — rare: at most one group per tree level
// is ever pending — go to a global overflow slab laid out [wave][depth][lane].  No per-child entry distance is kept and
// nothing is sorted: a child sits in the slot of the octant it lies in (builder), so the order in which a ray enters
// the hit slots follows from its direction signs alone (descending slot ^ octant; a 2-KiB LDS table gives the next slot).
struct WStack {
  typedef unsigned int ux2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) ux2 lds_u2;
  typedef __attribute__((address_space(1))) ux2 glb_u2;
  lds_u2* lds; glb_u2* ovf; int sp; int L;     // ovf: the WAVE's slab (a uniform pointer: it lives in scalar registers; the lane's column is added where it is used, on the rare deep pushes)
  PT_DEV void init(uint2* lds_base, uint2* ovf_wave, int l) { lds = (lds_u2*)lds_base; ovf = (glb_u2*)ovf_wave; sp = 0; L = l; }
  PT_DEV void reset() { sp = 0; }
  PT_DEV void push(uint32_t base, uint32_t mask) {
    const ux2 e = {base, mask};
    if (sp < L) lds[sp * 64] = e; else ovf[(sp - L) * 64 + (int)lane_id()] = e;      // explicit address spaces: ds_write_b64 / global_store_dwordx2
    ++sp;
  }
  PT_DEV uint2 pop() {
    --sp;
    ux2 e;
    if (sp < L) e = lds[sp * 64]; else e = ovf[(sp - L) * 64 + (int)lane_id()];      // ds_read_b64 / global_load_dwordx2, never FLAT
    return make_uint2(e.x, e.y);
  }
  PT_DEV bool empty() const { return sp == 0; }
};

typedef float float2v __attribute__((ext_vector_type(2)));
PT_DEV float hw_max(float a, float b) { return __builtin_fmaxf(a, b); }   // v_max_f32 / v_max3_f32: same value as
PT_DEV float hw_min(float a, float b) { return __builtin_fminf(a, b); }   // the ternary for non-NaN operands (±0 aside)
PT_DEV float ubyte_f(uint32_t w, int i) { return (float)((w >> (8 * i)) & 255u); }   // v_cvt_f32_ubyte{i}

// lane state of a traversal
#define CUR_DONE ((int)0x80000000)        // idle lane: wants a new ray
#define CUR_FINISHED (CUR_DONE + 1)       // ray finished, result not yet published
#define CUR_LEAF (CUR_DONE + 2)           // triangles of the last visited node pending

// Direction octant of a ray: bit k set when component k of the direction is >= 0 (the oracle's ray_octant).
PT_DEV uint32_t ray_octant(const ray_t& r) { return (r.inv.x >= 0.0f ? 1u : 0u) | (r.inv.y >= 0.0f ? 2u : 0u) | (r.inv.z >= 0.0f ? 4u : 0u); }

// Slot order table: s_order[oct][hits] = the set bit s of `hits` with the largest (s ^ oct).
#define ORDER_TABLE_BYTES 2048
PT_DEV void build_order_table(uint8_t* tab) {
  for (uint32_t i = threadIdx.x; i < ORDER_TABLE_BYTES; i += TRACE_BLOCK) {
    const uint32_t oct = i >> 8, hits = i & 255u;
    uint32_t best = 0, bk = 0;
    for (uint32_t sl = 0; sl < 8u; ++sl)
      if ((hits >> sl) & 1u) { const uint32_t k = (sl ^ oct) + 1u; if (k > bk) { bk = k; best = sl; } }
    tab[i] = (uint8_t)best;
  }
}

// One node visit: the 8 slots of node `cur` (read from the staged top of the tree at LDS byte address `node_addr` when it lies there,
// else gathered from global memory) against [tmin, tlimit].  Returns the hit mask (bit s = slot s, empty slots masked off), the node's children-block address
// and its slot masks (imask | lmask<<8 | two<<16).
// Per-ray constants of the slab test: the sign of each direction component says which of a child's two planes on that
// axis is entered first, so near/far need no min/max (identical values to min(t0,t1) / max(t0,t1): fma is monotonic in
// the plane coordinate).  plane distance = fma(q, 2^(e-127)·inv, fma(org, inv, -ood)), org = fma(oq, grid_step, grid_lo).
//
// Instruction economy (round 3, tools/valu_mix_bench.hip -> profiles/r03_valu_mix.json): on gfx950 a stream of v_fma / v_mul / v_add /
// v_mov or plain integer and / or / add issues one wave64 instruction per 2 cycles and SIMD, a stream of anything else (conversions, min /
// max, compares, v_cndmask, shifts, SDWA forms, v_fma_mix) one per 4; mixed, every instruction costs about one 2-cycle slot and the
// 4-cycle kinds bind only where they run back to back.  The slot test below — 6 v_cvt_f32_ubyte, 6 fma, 4 min/max, v_cmp + v_addc — takes
// the time of its 12 four-cycle instructions and of its 18 issue slots alike, so trading slow instructions for more fast ones loses: the
// hit mask from the sign of tf - tn by and / or / fma (4 fast for 2 slow) -1 %, bytes 0 and 1 converted by and / or / sub (3 fast for 1 slow)
// -4 %, both -5 % (commit c436d48, profiles/r03_node_visit_variants.txt).
PT_DEV uint32_t node_visit(const DevScene& sc, uint32_t node_addr, int cur, const ray_t& r, uint32_t oct, float tmin, float tlimit,
                           uint32_t& block, uint32_t& masks) {
  typedef float fx4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const fx4 lds_f4;
  typedef __attribute__((address_space(1))) const fx4 glb_f4;
  fx4 f0, f1, f2, f3;
  if ((uint32_t)cur < sc.n_lds_units) {
    lds_f4* p = (lds_f4*)(uintptr_t)node_addr;                    // ds_read_b128 × 4
    f0 = p[0]; f1 = p[1]; f2 = p[2]; f3 = p[3];
  } else {
    glb_f4* p = (glb_f4*)sc.recs + (uint32_t)cur;                 // global_load_dwordx4 × 4
    f0 = p[0]; f1 = p[1]; f2 = p[2]; f3 = p[3];
  }
  const uint32_t w0 = __float_as_uint(f0.x), w1 = __float_as_uint(f0.y), w2 = __float_as_uint(f0.z);
  const float ox = pt_fma((float)(w0 & 0xffffu), sc.grid_step[0], sc.grid_lo[0]), oy = pt_fma((float)(w0 >> 16), sc.grid_step[1], sc.grid_lo[1]),
              oz = pt_fma((float)(w1 & 0xffffu), sc.grid_step[2], sc.grid_lo[2]);
  const float ax = __uint_as_float(((w1 >> 16) & 255u) << 23) * r.inv.x, ay = __uint_as_float((w1 >> 24) << 23) * r.inv.y,
              az = __uint_as_float((w2 & 255u) << 23) * r.inv.z;
  // o·inv is recomputed here (the same product make_ray forms) instead of living in three registers for the ray's lifetime:
  // that is what keeps the kernel within the 64 VGPRs of 8 waves per SIMD without spilling
  const float bx = pt_fma(ox, r.inv.x, -(r.o.x * r.inv.x)), by = pt_fma(oy, r.inv.y, -(r.o.y * r.inv.y)), bz = pt_fma(oz, r.inv.z, -(r.o.z * r.inv.z));
  const bool px = (oct & 1u) != 0u, py = (oct & 2u) != 0u, pz = (oct & 4u) != 0u;
  // [0]: slots 0-3, [1]: slots 4-7
  const uint32_t lx[2] = {__float_as_uint(f1.x), __float_as_uint(f1.y)}, ly[2] = {__float_as_uint(f1.z), __float_as_uint(f1.w)},
                 lz[2] = {__float_as_uint(f2.x), __float_as_uint(f2.y)}, hx[2] = {__float_as_uint(f2.z), __float_as_uint(f2.w)},
                 hy[2] = {__float_as_uint(f3.x), __float_as_uint(f3.y)}, hz[2] = {__float_as_uint(f3.z), __float_as_uint(f3.w)};
  uint32_t hits = 0;
#pragma unroll
  for (int h = 1; h >= 0; --h) {
    const uint32_t nqx = px ? lx[h] : hx[h], fqx = px ? hx[h] : lx[h];
    const uint32_t nqy = py ? ly[h] : hy[h], fqy = py ? hy[h] : ly[h];
    const uint32_t nqz = pz ? lz[h] : hz[h], fqz = pz ? hz[h] : lz[h];
#pragma unroll
    for (int i = 3; i >= 0; --i) {      // slot 4h+i; descending, so that shifting the results in leaves slot s in bit s
      const float tn = hw_max(hw_max(hw_max(pt_fma(ubyte_f(nqx, i), ax, bx), pt_fma(ubyte_f(nqy, i), ay, by)), pt_fma(ubyte_f(nqz, i), az, bz)), tmin);
      const float tf = hw_min(hw_min(hw_min(pt_fma(ubyte_f(fqx, i), ax, bx), pt_fma(ubyte_f(fqy, i), ay, by)), pt_fma(ubyte_f(fqz, i), az, bz)), tlimit);
#ifndef PT_NO_ADDC
      asm("v_cmp_le_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hits) : "v"(tn), "v"(tf) : "vcc");   // hits = 2·hits + (tn <= tf)
#else
      hits = hits + hits + (tn <= tf ? 1u : 0u);
#endif
    }
  }
  masks = w2 >> 8;
  block = __float_as_uint(f0.w);
  return hits & ((masks | (masks >> 8)) & 255u);
}

// Pending-group bookkeeping.  G = (gbase = children-block address, gmask = hits<<8 | imask) is the group in registers; `nh` are the interior hits of
// the node just visited.  A non-empty new group goes on top (the old one is saved), then the next child is taken from the
// top group: slot by the order table (closest hit) or the lowest set bit (any hit).  Returns the next node or CUR_FINISHED.
template <bool ORDERED>
PT_DEV int advance(uint32_t& gbase, uint32_t& gmask, WStack& st, const uint8_t* order_tab, uint32_t oct) {
  typedef __attribute__((address_space(3))) const uint8_t lds_u8c;
  if ((gmask >> 8) == 0u) {
    if (st.empty()) return CUR_FINISHED;
    const uint2 e = st.pop();
    gbase = e.x; gmask = e.y;
  }
  const uint32_t hits = gmask >> 8;
  const uint32_t sl = ORDERED ? (uint32_t)((lds_u8c*)order_tab)[(oct << 8) | hits] : (uint32_t)__builtin_ctz(hits);
  gmask &= ~(256u << sl);
  return (int)(gbase + 4u * (uint32_t)__builtin_popcount(gmask & 255u & ((1u << sl) - 1u)));
}
PT_DEV void enter_group(uint32_t& gbase, uint32_t& gmask, WStack& st, uint32_t nbase, uint32_t nhits, uint32_t nimask) {
  if (nhits == 0u) return;
  if ((gmask >> 8) != 0u) st.push(gbase, gmask);
  gbase = nbase; gmask = (nhits << 8) | nimask;
}

// Pending triangles of the last visited node: tbase = unit address of the node's first triangle record,
// tmask = lhits | lmask<<8 | two<<16 | second<<24.  Returns the unit address of the next triangle to test and removes it
// from the set: the hit leaf slots in ascending order, a 2-triangle leaf's second triangle right after its first.
PT_DEV uint32_t next_triangle(uint32_t tbase, uint32_t& tmask) {
  const uint32_t sl = (uint32_t)__builtin_ctz(tmask & 255u);
  const uint32_t below = (1u << sl) - 1u;
  const uint32_t second = tmask >> 24;
  const uint32_t k = tbase + 3u * ((uint32_t)__builtin_popcount((tmask >> 8) & 255u & below) + (uint32_t)__builtin_popcount((tmask >> 16) & 255u & below) + second);
  const bool more = second == 0u && ((tmask >> (16u + sl)) & 1u) != 0u;
  tmask = more ? (tmask | (1u << 24)) : ((tmask & 0x00ffffffu) & ~(1u << sl));
  return k;
}

// triangles still pending in a lane's set (after next_triangle took some)
PT_DEV uint32_t pending_triangles(uint32_t tmask) {
  const uint32_t lh = tmask & 255u;
  return (uint32_t)__builtin_popcount(lh) + (uint32_t)__builtin_popcount((tmask >> 16) & lh) - (tmask >> 24);
}
#ifndef LEAF_EXTRA
#define LEAF_EXTRA 1     // triangles a lane may hand to free lanes per leaf pass
#endif
// Leaf pass, work distribution: every lane at a leaf tests one triangle itself and hands up to LEAF_EXTRA more to lanes that are not at a leaf.
// `extra` (0..LEAF_EXTRA) is what the lane would hand out; tasks are numbered by the exclusive prefix of `extra` over the lanes (ballot per bit of
// `extra` + mbcnt) and the first `ntask` = min(total, free lanes) of them find a helper: task i goes to the i-th free lane.
struct LeafDeal { uint32_t off, take, ntask, rank_free; bool helper; };
PT_DEV LeafDeal leaf_deal(uint32_t extra, bool leaf, uint64_t m_leaf) {
  LeafDeal d;
  uint32_t off = 0, total = 0;
#pragma unroll
  for (int b = 0; (1 << b) <= LEAF_EXTRA; ++b) {
    const uint64_t m = __ballot(((extra >> b) & 1u) != 0u);
    off += mbcnt64(m) << b; total += (uint32_t)__popcll(m) << b;
  }
  const uint64_t m_free = ~m_leaf;
  const uint32_t n_free = (uint32_t)__popcll(m_free);
  d.rank_free = mbcnt64(m_free);
  d.ntask = total < n_free ? total : n_free;
  d.helper = !leaf && d.rank_free < d.ntask;
  d.off = off;
  d.take = off >= d.ntask ? 0u : (d.ntask - off < extra ? d.ntask - off : extra);
  return d;
}

// Wave-private reservoir of input slots: idle lanes are refilled from a chunk of consecutive rays of one queue segment (ballot +
// mbcnt).  Chunks are numbered over the whole queue (k_scan's prefix of chunks per segment); the first chunk of every wave is static
// (wave w of the grid owns chunk w), the rest are handed out by one atomic per chunk.  Atomics on one word are served one at a time
// (~88 per us): when every wave of the persistent grid (8192) opened with one, an all but empty launch took 140 us; a queue the static
// round covers costs none.  (Looking at the counter with an agent-scope load before asking for a chunk, to spare the failing atomics
// at the end, made the kernels 53 % slower.)
struct Reservoir {
  uint32_t next, end, n_chunks, chunk, my_static; bool exhausted, static_left;
  PT_DEV void init(const uint32_t* hdr, uint32_t wave_in_grid) {
    chunk = hdr[1]; n_chunks = hdr[2];
    my_static = wave_in_grid; static_left = wave_in_grid < n_chunks;
    next = end = 0; exhausted = false;
  }
  // nothing for any wave of this block: its static chunks lie behind the queue's end, and then so do all dynamic ones
  static PT_DEV bool block_has_no_work(const uint32_t* hdr) { return hdr[2] <= blockIdx.x * TRACE_WAVES; }
  // chunk number -> slots [next, end): 64-ary search of the chunk prefix for the segment, then the chunk's place inside it
  PT_DEV void open_chunk(const DevQueues& q, const uint32_t* pre, const uint32_t* seg, uint32_t c, uint32_t lane) {
    uint32_t lo = 0, span = q.n_seg;                       // the answer (largest s with pre[s] <= c) lies in [lo, lo + span)
    while (span > 1u) {
      const uint32_t step = (span + 63u) >> 6, idx = lo + lane * step;
      const uint32_t v = lane * step < span ? pre[idx] : 0xffffffffu;
      const uint32_t k = (uint32_t)__popcll(__ballot(v <= c)) - 1u;
      lo += k * step;
      span = span - k * step < step ? span - k * step : step;
    }
    const uint32_t first = pre[lo], cnt = seg[lo];
    const uint32_t j = c - (uint32_t)__builtin_amdgcn_readfirstlane((int)first), live = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
    lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
    next = lo * q.seg_len + j * chunk;
    end = lo * q.seg_len + ((j + 1u) * chunk < live ? (j + 1u) * chunk : live);
  }
  // the next (at most `want`) consecutive slots of the wave's current chunk: [first, first + n); n = 0: the queue is exhausted
  PT_DEV uint32_t take(const DevQueues& q, const uint32_t* pre, const uint32_t* seg, uint32_t* ctr, uint32_t lane, uint32_t want, uint32_t& first) {
    if (next >= end && !exhausted) {
      if (chunk < TRACE_CHUNK) {                                      // a short queue: this wave's next chunk of the round-robin deal
        if (my_static < n_chunks) { open_chunk(q, pre, seg, my_static, lane); my_static += gridDim.x * TRACE_WAVES; }
        else exhausted = true;
      } else if (static_left) { static_left = false; open_chunk(q, pre, seg, my_static, lane); }
      else {
        const uint32_t waves = gridDim.x * TRACE_WAVES;
        if (n_chunks <= waves) exhausted = true;                     // the static round covered the queue: no atomic at all
        else {
          const uint64_t c = (uint64_t)waves + wave_fetch(ctr, 1u, lane);
          if (c >= n_chunks) exhausted = true; else open_chunk(q, pre, seg, (uint32_t)c, lane);
        }
      }
    }
    const uint32_t avail = end - next;
    const uint32_t n = avail < want ? avail : want;
    first = next; next += n;
    return n;
  }
};

// Prepared rays of a wave (round 4).  The round-3 refill ran make_ray — three IEEE divisions, ~85 vector instructions with the queue loads — for
// the ~17 idle lanes of a round, i.e. at 0.27 lane utilisation, and was therefore put off until 12 lanes had nothing to do: on average 13 % of a
// wave's lanes sat idle through its node and leaf iterations.  Now the wave PREPARES the next TRACE_RING rays of its chunk with all 64 lanes
// (queue loads, safe_dir, the divisions, the octant) and parks (slot, inv, octant) in LDS; handing one to an idle lane is five ds_reads and the
// two queue loads of (o, d) — second readers of lines the staging brought to L2 —, cheap enough to do as soon as a lane is free.  Which lane
// gets which ray changes neither a hit (written to the ray's own slot) nor a counter (sums over rays).
struct RayRing {
  typedef __attribute__((address_space(3))) uint32_t lds_u32;
  lds_u32* f;      // this wave's [RING_FIELDS][TRACE_RING] words
  uint32_t n;      // prepared rays left (wave-uniform); they are handed out from the top
  PT_DEV void init(uint32_t* base) { f = (lds_u32*)base; n = 0; }
  PT_DEV void stage(Reservoir& res, const DevQueues& q, const uint32_t* pre, const uint32_t* seg, uint32_t* ctr, const float4* QA, const float4* QB, uint32_t lane) {
    uint32_t first;
    const uint32_t k = res.take(q, pre, seg, ctr, lane, TRACE_RING, first);
    if (lane < k) {
      const uint32_t ri = first + lane;
      const float4 A = QA[ri], B = QB[ri];
      const ray_t r = make_ray(V3(A.x, A.y, A.z), V3(A.w, B.x, B.y));
      f[lane] = ri;
      f[TRACE_RING + lane] = __float_as_uint(r.inv.x); f[2 * TRACE_RING + lane] = __float_as_uint(r.inv.y); f[3 * TRACE_RING + lane] = __float_as_uint(r.inv.z);
      f[4 * TRACE_RING + lane] = (r.inv.x >= 0.0f ? 1u : 0u) | (r.inv.y >= 0.0f ? 2u : 0u) | (r.inv.z >= 0.0f ? 4u : 0u);
    }
    n = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  PT_DEV bool pop(bool idle, uint32_t& ri, v3& inv, uint32_t& oct) {
    const uint64_t m = __ballot(idle);
    const uint32_t rank = mbcnt64(m), need = (uint32_t)__popcll(m);
    const bool got = idle && rank < n;
    if (got) {
      const uint32_t i = n - 1u - rank;
      ri = f[i];
      inv = V3(__uint_as_float(f[TRACE_RING + i]), __uint_as_float(f[2 * TRACE_RING + i]), __uint_as_float(f[3 * TRACE_RING + i]));
      oct = f[4 * TRACE_RING + i];
    }
    n -= need < n ? need : n;
    return got;
  }
};

#ifdef PT_DIAG
#define DIAG_ITER(var) do { const uint64_t m_ = __ballot(true); if ((int)lane == __ffsll((unsigned long long)m_) - 1) ++(var); } while (0)
#else
#define DIAG_ITER(var) do { } while (0)
#endif

// LDS of a trace block: [n_lds_units × 16 B: the top of the tree][waves × L × 64 stack entries of 8 B][2 KiB slot-order table (closest hit only)][waves × RING_FIELDS × TRACE_RING words: the prepared rays]
struct TraceLds { uint32_t top_addr; uint2* stack; uint8_t* order_tab; uint32_t* ring; };
PT_DEV TraceLds trace_lds(float4* lds_raw, const DevScene& sc, int stack_lds, bool closest) {
  TraceLds t;
  t.top_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds_raw;
  t.stack = reinterpret_cast<uint2*>(lds_raw + sc.n_lds_units);
  t.order_tab = reinterpret_cast<uint8_t*>(t.stack + (size_t)TRACE_WAVES * (size_t)stack_lds * 64u);
  t.ring = reinterpret_cast<uint32_t*>(t.order_tab + (closest ? ORDER_TABLE_BYTES : 0));
  return t;
}
// triangle record at unit address k: from the staged top of the tree or from global memory (two explicit address spaces: a
// pointer that may be either compiles to FLAT loads, which occupy both the LDS and the vector-memory pipe)
PT_DEV void load_triangle(const DevScene& sc, uint32_t top_addr, uint32_t k, float4& a, float4& b, float4& c) {
  typedef float fx4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) const fx4 lds_f4;
  typedef __attribute__((address_space(1))) const fx4 glb_f4;
  fx4 x, y, z;
  if (k + 3u <= sc.n_lds_units) { lds_f4* p = (lds_f4*)(uintptr_t)(top_addr + 16u * k); x = p[0]; y = p[1]; z = p[2]; }
  else { glb_f4* p = (glb_f4*)sc.recs + k; x = p[0]; y = p[1]; z = p[2]; }
  a = make_float4(x.x, x.y, x.z, x.w); b = make_float4(y.x, y.y, y.z, y.w); c = make_float4(z.x, z.y, z.z, z.w);
}

// =================================================================================================
// P3 closest-hit traversal + triangle intersection: persistent waves, dynamic lane refill, 8-wide BVH.
// Per node: the 8 slots are tested against [tmin, best_t]; the triangles of the hit leaf slots are intersected (slot
// order) before the ray descends into the hit interior children, nearest octant first.
// Closest hit = lexicographic minimum of (t, original primitive id).
// The hit record (t, prim | class<<28, u, v) is written IN PLACE at the ray's slot (miss: prim = -1).
// CULL: R6 back-face culling + per-ray [tmin,tmax] from B.zw (raster-compat primary rays).
template <bool CULL>
__global__ __launch_bounds__(TRACE_BLOCK, TRACE_MIN_WAVES) void k_trace_closest(DevScene sc, DevQueues q, int qi, int stack_lds) {
  extern __shared__ float4 lds_raw[];
  __shared__ uint8_t s_pair[2][TRACE_WAVES][64];   // leaf phase: k-th owner with a second triangle <-> k-th free lane
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  if (Reservoir::block_has_no_work(&q.cnt[CNT_RAY_TOTAL])) return;     // a short queue: most blocks of the persistent grid leave before staging anything
  const TraceLds L = trace_lds(lds_raw, sc, stack_lds, true);
  uint8_t* order_tab = L.order_tab;
  for (uint32_t i = threadIdx.x; i < sc.n_lds_units; i += TRACE_BLOCK) lds_raw[i] = sc.recs[i];
  build_order_table(order_tab);
  __syncthreads();
  const RayQ rq = q.ray[qi];
  unsigned long long nv = 0, nr = 0, nh = 0;   // wave totals, only updated at wave-uniform points: they live in SGPRs
  uint32_t nt = 0;                               // per lane (updated inside the divergent leaf phase)
  uint32_t d_node = 0, d_tri = 0, d_round = 0, d_leftpass = 0, d_left = 0;
  (void)d_node; (void)d_tri; (void)d_round; (void)d_leftpass; (void)d_left;
#ifdef PT_STAMP
  unsigned long long t_refill = 0, t_node = 0, t_leaf = 0, t_fin = 0, t_mark = __builtin_amdgcn_s_memtime();
#define STAMP(acc) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - t_mark; t_mark = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(acc) do { } while (0)
#endif
  Reservoir res; res.init(&q.cnt[CNT_RAY_TOTAL], (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * TRACE_WAVES + wave)));
  const uint32_t* seg_cnt = q.seg_ray[qi];
  WStack st;
  st.init(L.stack + (size_t)wave * (size_t)stack_lds * 64u + lane,
          sc.stack_ovf + ((size_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * TRACE_WAVES + wave)) * sc.ovf_depth) * 64u, stack_lds);
  int cur = CUR_DONE;
  uint32_t ri = 0, oct = 0, gbase = 0, gmask = 0, tbase = 0, tmask = 0;
  ray_t r = make_ray(V3(0, 0, 0), V3(0, 0, 1));
  float tmin = 0.0f, best_t = PT_T_INF, best_u = 0.0f, best_v = 0.0f;
  int best_prim = 0x7fffffff, best_cls = 0;
  bool found = false;
  RayRing ring; ring.init(L.ring + (size_t)wave * RING_FIELDS * TRACE_RING);
  for (;;) {
    // ---- hand prepared rays to the idle lanes (RayRing) ----
    {
      bool idle = cur == CUR_DONE;
      uint64_t mi = __ballot(idle);
      if ((uint32_t)__popcll(mi) >= TRACE_REFILL_IDLE || !__ballot(cur != CUR_DONE)) {
        while (mi) {
          if (ring.n == 0u) {
            if (res.exhausted) break;
            ring.stage(res, q, q.pre_ray, seg_cnt, &q.cnt[CNT_WORK_TRACE], rq.A, rq.B, lane);
            if (ring.n == 0u) break;
          }
          v3 inv = V3(0, 0, 0); uint32_t oc = 0;
          const bool got = ring.pop(idle, ri, inv, oc);
          nr += (unsigned long long)__popcll(__ballot(got));
          if (got) {
            const float4 A = rq.A[ri], Bq = rq.B[ri];
            r.o = V3(A.x, A.y, A.z); r.d = V3(A.w, Bq.x, Bq.y); r.inv = inv;
            oct = oc;
            tmin = CULL ? Bq.z : 0.0f;
            best_t = CULL ? Bq.w : PT_T_INF; best_u = 0.0f; best_v = 0.0f; best_prim = 0x7fffffff; best_cls = 0; found = false;
            st.reset();
            gmask = 0; tmask = 0;
            cur = 0;
          }
          idle = cur == CUR_DONE; mi = __ballot(idle);
        }
      }
    }
    STAMP(t_refill);
    if (!__ballot(cur != CUR_DONE)) break;        // nothing in flight, nothing prepared, the queue exhausted
    // ---- one cycle: node visits while enough lanes walk, one leaf pass, publish ----
    DIAG_ITER(d_round);
    {
      for (;;) {
        const uint64_t mn = __ballot(cur >= 0);
        if (!mn) break;
        if (__popcll(mn) < TRACE_NODE_MIN && __ballot(cur == CUR_LEAF)) break;   // few walkers, triangles waiting
        nv += (unsigned long long)__popcll(mn);
        if (cur >= 0) {
          DIAG_ITER(d_node);
          uint32_t nb, masks;
          const uint32_t node_addr = L.top_addr + 16u * (uint32_t)cur;
          const uint32_t hits = node_visit(sc, node_addr, cur, r, oct, tmin, best_t, nb, masks);
          const uint32_t imask = masks & 255u, lhits = hits & (masks >> 8);
          enter_group(gbase, gmask, st, nb, hits & imask, imask);
          tbase = nb + 4u * (uint32_t)__builtin_popcount(imask);
          tmask = lhits | (masks & 0x00ffff00u);
          cur = lhits ? CUR_LEAF : advance<true>(gbase, gmask, st, order_tab, oct);
        }
      }
      STAMP(t_node);
      // ---- leaf phase: every lane with pending triangles tests one; a lane that then still has triangles pending hands its next
      // one to a lane that is not at a leaf (the k-th such owner to the k-th free lane, through two 64-byte LDS tables; the helper
      // fetches the owner's ray through ds_bpermute and returns its result the same way).  Same tests, same results and counters
      // as one triangle per pass — closest hit is an order-independent minimum — in fewer passes.
#if LEAF_EXTRA == 0    // no hand-outs: every lane at a leaf tests its own next triangle (no cross-lane traffic at all)
      {
        const bool leaf = cur == CUR_LEAF;
        if (__ballot(leaf)) {
          DIAG_ITER(d_tri);
          if (leaf) {
            const uint32_t k = next_triangle(tbase, tmask);
            float4 a, b, c;
            load_triangle(sc, L.top_addr, k, a, b, c);
            float t, u, v;
            const bool hit = tri_test<CULL>(r, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), t, u, v);
            const int pid = __float_as_int(a.w);
            ++nt;
            if (hit && t > tmin && (t < best_t || (t == best_t && pid < best_prim))) {
              best_t = t; best_u = u; best_v = v; best_prim = pid; best_cls = __float_as_int(b.w); found = true;
            }
            if ((tmask & 255u) == 0u) cur = advance<true>(gbase, gmask, st, order_tab, oct);
          }
        }
      }
#elif LEAF_EXTRA > 1
      {
        const bool leaf = cur == CUR_LEAF;
        const uint64_t m_leaf = __ballot(leaf);
        if (m_leaf) {
          DIAG_ITER(d_tri);
          uint32_t k1 = 0, pend = 0;
          if (leaf) { k1 = next_triangle(tbase, tmask); pend = pending_triangles(tmask); }
          const LeafDeal dl = leaf_deal(pend < LEAF_EXTRA ? pend : LEAF_EXTRA, leaf, m_leaf);
          typedef __attribute__((address_space(3))) volatile uint8_t lds_u8;
          lds_u8* tab_task = (lds_u8*)s_pair[0][wave];      // task -> owner lane | which of its extras << 6
          lds_u8* tab_helper = (lds_u8*)s_pair[1][wave];    // task -> helper lane
          uint32_t kx[LEAF_EXTRA];
#pragma unroll
          for (uint32_t j = 0; j < LEAF_EXTRA; ++j) {
            kx[j] = 0;
            if (j < dl.take) { kx[j] = next_triangle(tbase, tmask); tab_task[dl.off + j] = (uint8_t)(lane | (j << 6)); }
          }
          if (dl.helper) tab_helper[dl.rank_free] = (uint8_t)lane;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const uint32_t task = dl.helper ? (uint32_t)tab_task[dl.rank_free] : lane;
          const int src = (int)(task & 63u);
          ray_t rr;
          rr.o = V3(__shfl(r.o.x, src), __shfl(r.o.y, src), __shfl(r.o.z, src));
          rr.d = V3(__shfl(r.d.x, src), __shfl(r.d.y, src), __shfl(r.d.z, src));
          uint32_t k = k1;
#pragma unroll
          for (uint32_t j = 0; j < LEAF_EXTRA; ++j) {
            const uint32_t kj = (uint32_t)__shfl((int)kx[j], src);       // unconditional: a shuffle inside ?: would run with the owners masked off
            if (dl.helper && (task >> 6) == j) k = kj;
          }
          bool hit = false; float t = 0.0f, u = 0.0f, v = 0.0f; int pid = 0, cls = 0;
          if (leaf || dl.helper) {
            float4 a, b, c;
            load_triangle(sc, L.top_addr, k, a, b, c);
            hit = tri_test<CULL>(rr, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), t, u, v);
            pid = __float_as_int(a.w); cls = __float_as_int(b.w);
          }
          if (leaf) {
            ++nt;
            if (hit && t > tmin && (t < best_t || (t == best_t && pid < best_prim))) {
              best_t = t; best_u = u; best_v = v; best_prim = pid; best_cls = cls; found = true;
            }
          }
          // results of the handed-out triangles: (t, prim | class) of each, then (u, v) of the best of them only
          const float tt = hit ? t : -INFINITY;
          const int pw = pid | (cls << HIT_CLASS_SHIFT);
          float bt = 0.0f; int bp = 0, bl = (int)lane; bool have = false;
#pragma unroll
          for (uint32_t j = 0; j < LEAF_EXTRA; ++j) {
            const int hl = j < dl.take ? (int)tab_helper[dl.off + j] : (int)lane;
            const float tj = __shfl(tt, hl);
            const int pj = __shfl(pw, hl);
            if (j < dl.take) {
              ++nt;
              const int prim_j = pj & ((1 << HIT_CLASS_SHIFT) - 1);
              if (tj > tmin && (!have || tj < bt || (tj == bt && prim_j < (bp & ((1 << HIT_CLASS_SHIFT) - 1))))) { bt = tj; bp = pj; bl = hl; have = true; }
            }
          }
          const float ub = __shfl(u, bl), vb = __shfl(v, bl);
          if (have) {
            const int prim_b = bp & ((1 << HIT_CLASS_SHIFT) - 1);
            if (bt < best_t || (bt == best_t && prim_b < best_prim)) {
              best_t = bt; best_u = ub; best_v = vb; best_prim = prim_b; best_cls = (int)((uint32_t)bp >> HIT_CLASS_SHIFT); found = true;
            }
          }
          if (leaf && (tmask & 255u) == 0u) cur = advance<true>(gbase, gmask, st, order_tab, oct);
#ifdef PT_DIAG
          {
            const uint32_t left = (cur == CUR_LEAF) ? pending_triangles(tmask) : 0u;
            const uint64_t m_left = __ballot(left != 0u);
            if (m_left) { if ((int)lane == __ffsll((unsigned long long)m_left) - 1) ++d_leftpass; d_left += left; }
          }
#endif
        }
      }
#else
      {
        const bool leaf = cur == CUR_LEAF;
        const uint64_t m_leaf = __ballot(leaf);
        if (m_leaf) {
          DIAG_ITER(d_tri);
          uint32_t k1 = 0, k2 = 0;
          bool more = false;
          if (leaf) { k1 = next_triangle(tbase, tmask); more = (tmask & 255u) != 0u; }
          const uint64_t m_more = __ballot(more), m_free = ~m_leaf;
          const uint32_t n_more = (uint32_t)__popcll(m_more), n_free = (uint32_t)__popcll(m_free);
          const uint32_t rank_more = mbcnt64(m_more), rank_free = mbcnt64(m_free);
          const bool paired = more && rank_more < n_free, helper = !leaf && rank_free < n_more;
          typedef __attribute__((address_space(3))) volatile uint8_t lds_u8;
          lds_u8* tab_owner = (lds_u8*)s_pair[0][wave];
          lds_u8* tab_helper = (lds_u8*)s_pair[1][wave];
          if (paired) { k2 = next_triangle(tbase, tmask); tab_owner[rank_more] = (uint8_t)lane; }
          if (helper) tab_helper[rank_free] = (uint8_t)lane;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int src = helper ? (int)tab_owner[rank_free] : (int)lane;        // whose ray this lane tests with
          ray_t rr;
          rr.o = V3(__shfl(r.o.x, src), __shfl(r.o.y, src), __shfl(r.o.z, src));
          rr.d = V3(__shfl(r.d.x, src), __shfl(r.d.y, src), __shfl(r.d.z, src));
          const uint32_t k2_src = (uint32_t)__shfl((int)k2, src);                 // unconditional: a shuffle inside ?: would run with the owners masked off
          const uint32_t k = helper ? k2_src : k1;
          bool hit = false; float t = 0.0f, u = 0.0f, v = 0.0f; int pid = 0, cls = 0;
          if (leaf || helper) {
            float4 a, b, c;
            load_triangle(sc, L.top_addr, k, a, b, c);
            hit = tri_test<CULL>(rr, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), t, u, v);
            pid = __float_as_int(a.w); cls = __float_as_int(b.w);
          }
          if (leaf) {
            ++nt;
            if (hit && t > tmin && (t < best_t || (t == best_t && pid < best_prim))) {
              best_t = t; best_u = u; best_v = v; best_prim = pid; best_cls = cls; found = true;
            }
          }
          const int hsrc = paired ? (int)tab_helper[rank_more] : (int)lane;      // the lane that tested this owner's second triangle
          const int h2 = __shfl(hit ? 1 : 0, hsrc);
          const float t2 = __shfl(t, hsrc), u2 = __shfl(u, hsrc), v2 = __shfl(v, hsrc);
          const int pid2 = __shfl(pid, hsrc), cls2 = __shfl(cls, hsrc);
          if (paired) {
            ++nt;
            if (h2 && t2 > tmin && (t2 < best_t || (t2 == best_t && pid2 < best_prim))) {
              best_t = t2; best_u = u2; best_v = v2; best_prim = pid2; best_cls = cls2; found = true;
            }
          }
          if (leaf && (tmask & 255u) == 0u) cur = advance<true>(gbase, gmask, st, order_tab, oct);
#ifdef PT_DIAG     // what a pass leaves behind: passes after which some lane still has triangles pending, and how many triangles those are
          {
            const uint32_t left = (cur == CUR_LEAF) ? (uint32_t)__builtin_popcount(tmask & 255u) + (uint32_t)__builtin_popcount((tmask >> 16) & tmask & 255u) - (tmask >> 24) : 0u;
            const uint64_t m_left = __ballot(left != 0u);
            if (m_left) { if ((int)lane == __ffsll((unsigned long long)m_left) - 1) ++d_leftpass; d_left += left; }
          }
#endif
        }
      }
#endif
      STAMP(t_leaf);
      nh += (unsigned long long)__popcll(__ballot(cur == CUR_FINISHED && found));
      if (cur == CUR_FINISHED) {                                       // this lane's ray is finished: publish in place
        q.hit[ri] = make_float4(found ? best_t : -1.0f, __int_as_float(found ? (best_prim | (best_cls << HIT_CLASS_SHIFT)) : -1), best_u, best_v);
        cur = CUR_DONE;
      }
      STAMP(t_fin);
    }
  }
  {   // counters: summed over the block first, then one atomic per counter, each counter on its own line (ptc_internal.h, ST_STRIDE)
    __shared__ unsigned long long s_stat[TRACE_WAVES][4];
    const unsigned long long c_tris = wave_sum(nt);
    if (lane == 0) { s_stat[wave][0] = nv; s_stat[wave][1] = c_tris; s_stat[wave][2] = nr; s_stat[wave][3] = nh; }
    __syncthreads();
    if (threadIdx.x < 4u) {
      unsigned long long v = 0;
      for (int w = 0; w < TRACE_WAVES; ++w) v += s_stat[w][threadIdx.x];
      const int at = threadIdx.x == 0u ? ST_NODES_C : threadIdx.x == 1u ? ST_TRIS_C : threadIdx.x == 2u ? ST_SEGMENTS : ST_HITS;
      if (v) atomicAdd(&q.stats[at * ST_STRIDE], v);
    }
  }
#ifdef PT_STAMP
  if (lane == 0) { atomicAdd(&q.stats[ST_DIAG_NODE_ITERS * ST_STRIDE], t_node); atomicAdd(&q.stats[ST_DIAG_TRI_ITERS * ST_STRIDE], t_leaf); atomicAdd(&q.stats[ST_DIAG_LEAF_VISITS * ST_STRIDE], t_fin); atomicAdd(&q.stats[ST_DIAG_ROUNDS * ST_STRIDE], t_refill); }
#endif
#ifdef PT_DIAG
  {
    unsigned long long a0 = wave_sum(d_node), a1 = wave_sum(d_tri), a3 = wave_sum(d_round), a4 = wave_sum(d_leftpass), a5 = wave_sum(d_left);
    if (lane == 0) { atomicAdd(&q.stats[ST_DIAG_NODE_ITERS * ST_STRIDE], a0); atomicAdd(&q.stats[ST_DIAG_TRI_ITERS * ST_STRIDE], a1); atomicAdd(&q.stats[ST_DIAG_ROUNDS * ST_STRIDE], a3);
                     atomicAdd(&q.stats[ST_DIAG_LEAF_VISITS * ST_STRIDE], a4); atomicAdd(&q.stats[ST_DIAG_REFILLED * ST_STRIDE], a5); }
  }
#endif
}

// =================================================================================================
// P4 any-hit traversal for the NEE shadow rays: same nodes and machinery, the interval is fixed, hit slots are taken in
// ascending slot order (occlusion needs no front-to-back order) and the first triangle hit inside (0, tmax) ends the ray.
// An unoccluded ray adds its contribution to the path's radiance word (single owner: one shadow ray per path per bounce).
template <bool DEBUG_OUT>
__global__ __launch_bounds__(TRACE_BLOCK, TRACE_MIN_WAVES) void k_trace_any(DevScene sc, DevQueues q, int stack_lds, uint8_t* debug_out) {
  extern __shared__ float4 lds_raw[];
  __shared__ uint8_t s_pair[2][TRACE_WAVES][64];   // leaf phase: k-th owner with a second triangle <-> k-th free lane
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  if (Reservoir::block_has_no_work(&q.cnt[CNT_SH_TOTAL])) return;
  const TraceLds L = trace_lds(lds_raw, sc, stack_lds, false);
  for (uint32_t i = threadIdx.x; i < sc.n_lds_units; i += TRACE_BLOCK) lds_raw[i] = sc.recs[i];
  __syncthreads();
  unsigned long long nv = 0, nr = 0;
  uint32_t nt = 0;
  Reservoir res; res.init(&q.cnt[CNT_SH_TOTAL], (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * TRACE_WAVES + wave)));
  WStack st;
  st.init(L.stack + (size_t)wave * (size_t)stack_lds * 64u + lane,
          sc.stack_ovf + ((size_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * TRACE_WAVES + wave)) * sc.ovf_depth) * 64u, stack_lds);
  int cur = CUR_DONE;
  uint32_t ri = 0, path = 0, oct = 0, gbase = 0, gmask = 0, tbase = 0, tmask = 0;
  ray_t r = make_ray(V3(0, 0, 0), V3(0, 0, 1));
  float tmax = 0.0f;
  bool occluded = false;
  RayRing ring; ring.init(L.ring + (size_t)wave * RING_FIELDS * TRACE_RING);
  for (;;) {
    {
      bool idle = cur == CUR_DONE;
      uint64_t mi = __ballot(idle);
      if ((uint32_t)__popcll(mi) >= TRACE_REFILL_IDLE || !__ballot(cur != CUR_DONE)) {
        while (mi) {
          if (ring.n == 0u) {
            if (res.exhausted) break;
            ring.stage(res, q, q.pre_sh, q.seg_sh, &q.cnt[CNT_WORK_SHADOW], q.shadow.A, q.shadow.B, lane);
            if (ring.n == 0u) break;
          }
          v3 inv = V3(0, 0, 0); uint32_t oc = 0;
          const bool got = ring.pop(idle, ri, inv, oc);
          nr += (unsigned long long)__popcll(__ballot(got));
          if (got) {
            const float4 A = q.shadow.A[ri], Bq = q.shadow.B[ri];
            r.o = V3(A.x, A.y, A.z); r.d = V3(A.w, Bq.x, Bq.y); r.inv = inv;
            oct = oc;
            tmax = Bq.z; path = __float_as_uint(Bq.w);
            occluded = false;
            st.reset();
            gmask = 0; tmask = 0;
            cur = 0;
          }
          idle = cur == CUR_DONE; mi = __ballot(idle);
        }
      }
    }
    if (!__ballot(cur != CUR_DONE)) break;
    {
      for (;;) {
        const uint64_t mn = __ballot(cur >= 0);
        if (!mn) break;
        if (__popcll(mn) < TRACE_NODE_MIN_ANY && __ballot(cur == CUR_LEAF)) break;
        nv += (unsigned long long)__popcll(mn);
        if (cur >= 0) {
          uint32_t nb, masks;
          const uint32_t node_addr = L.top_addr + 16u * (uint32_t)cur;
          const uint32_t hits = node_visit(sc, node_addr, cur, r, oct, 0.0f, tmax, nb, masks);
          const uint32_t imask = masks & 255u, lhits = hits & (masks >> 8);
          enter_group(gbase, gmask, st, nb, hits & imask, imask);
          tbase = nb + 4u * (uint32_t)__builtin_popcount(imask);
          tmask = lhits | (masks & 0x00ffff00u);
          cur = lhits ? CUR_LEAF : advance<false>(gbase, gmask, st, nullptr, 0u);
        }
      }
#if LEAF_EXTRA == 0
      {
        const bool leaf = cur == CUR_LEAF;
        if (__ballot(leaf)) {
          if (leaf) {
            const uint32_t k = next_triangle(tbase, tmask);
            float4 a, b, c;
            load_triangle(sc, L.top_addr, k, a, b, c);
            float t, u, v;
            const bool hit = tri_test<false>(r, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), t, u, v) && t > 0.0f && t < tmax;
            ++nt;
            if (hit) { occluded = true; cur = CUR_FINISHED; }
            else if ((tmask & 255u) == 0u) cur = advance<false>(gbase, gmask, st, nullptr, 0u);
          }
        }
      }
#elif LEAF_EXTRA > 1
      {
        // ---- leaf phase: as in k_trace_closest, up to LEAF_EXTRA more of a lane's pending triangles are tested in the same pass by free lanes.
        // The counter stays the sequential one: a test is counted (and used) only when all tests before it missed.
        const bool leaf = cur == CUR_LEAF;
        const uint64_t m_leaf = __ballot(leaf);
        if (m_leaf) {
          uint32_t k1 = 0, pend = 0;
          if (leaf) { k1 = next_triangle(tbase, tmask); pend = pending_triangles(tmask); }
          const LeafDeal dl = leaf_deal(pend < LEAF_EXTRA ? pend : LEAF_EXTRA, leaf, m_leaf);
          typedef __attribute__((address_space(3))) volatile uint8_t lds_u8;
          lds_u8* tab_task = (lds_u8*)s_pair[0][wave];
          lds_u8* tab_helper = (lds_u8*)s_pair[1][wave];
          uint32_t kx[LEAF_EXTRA];
#pragma unroll
          for (uint32_t j = 0; j < LEAF_EXTRA; ++j) {
            kx[j] = 0;
            if (j < dl.take) { kx[j] = next_triangle(tbase, tmask); tab_task[dl.off + j] = (uint8_t)(lane | (j << 6)); }
          }
          if (dl.helper) tab_helper[dl.rank_free] = (uint8_t)lane;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const uint32_t task = dl.helper ? (uint32_t)tab_task[dl.rank_free] : lane;
          const int src = (int)(task & 63u);
          ray_t rr;
          rr.o = V3(__shfl(r.o.x, src), __shfl(r.o.y, src), __shfl(r.o.z, src));
          rr.d = V3(__shfl(r.d.x, src), __shfl(r.d.y, src), __shfl(r.d.z, src));
          const float tmax_src = __shfl(tmax, src);
          uint32_t k = k1;
#pragma unroll
          for (uint32_t j = 0; j < LEAF_EXTRA; ++j) {
            const uint32_t kj = (uint32_t)__shfl((int)kx[j], src);
            if (dl.helper && (task >> 6) == j) k = kj;
          }
          bool hit = false;
          if (leaf || dl.helper) {
            float4 a, b, c;
            load_triangle(sc, L.top_addr, k, a, b, c);
            float t, u, v;
            hit = tri_test<false>(rr, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), t, u, v) && t > 0.0f && t < tmax_src;
          }
          bool done = false;
          if (leaf) { ++nt; if (hit) { occluded = true; cur = CUR_FINISHED; done = true; } }
#pragma unroll
          for (uint32_t j = 0; j < LEAF_EXTRA; ++j) {
            const int hl = j < dl.take ? (int)tab_helper[dl.off + j] : (int)lane;
            const int hj = __shfl(hit ? 1 : 0, hl);
            if (j < dl.take && !done) { ++nt; if (hj) { occluded = true; cur = CUR_FINISHED; done = true; } }
          }
          if (cur == CUR_LEAF && (tmask & 255u) == 0u) cur = advance<false>(gbase, gmask, st, nullptr, 0u);
        }
      }
#else
      {
        // ---- leaf phase: as in k_trace_closest, a lane's second pending triangle is tested in the same pass by a free lane.  The
        // counter stays the sequential one: the second test is counted (and used) only when the first one missed.
        const bool leaf = cur == CUR_LEAF;
        const uint64_t m_leaf = __ballot(leaf);
        if (m_leaf) {
          uint32_t k1 = 0, k2 = 0;
          bool more = false;
          if (leaf) { k1 = next_triangle(tbase, tmask); more = (tmask & 255u) != 0u; }
          const uint64_t m_more = __ballot(more), m_free = ~m_leaf;
          const uint32_t n_more = (uint32_t)__popcll(m_more), n_free = (uint32_t)__popcll(m_free);
          const uint32_t rank_more = mbcnt64(m_more), rank_free = mbcnt64(m_free);
          const bool paired = more && rank_more < n_free, helper = !leaf && rank_free < n_more;
          typedef __attribute__((address_space(3))) volatile uint8_t lds_u8;
          lds_u8* tab_owner = (lds_u8*)s_pair[0][wave];
          lds_u8* tab_helper = (lds_u8*)s_pair[1][wave];
          uint32_t tmask_after = tmask;
          if (paired) { k2 = next_triangle(tbase, tmask_after); tab_owner[rank_more] = (uint8_t)lane; }
          if (helper) tab_helper[rank_free] = (uint8_t)lane;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int src = helper ? (int)tab_owner[rank_free] : (int)lane;
          ray_t rr;
          rr.o = V3(__shfl(r.o.x, src), __shfl(r.o.y, src), __shfl(r.o.z, src));
          rr.d = V3(__shfl(r.d.x, src), __shfl(r.d.y, src), __shfl(r.d.z, src));
          const float tmax_src = __shfl(tmax, src);
          const uint32_t k2_src = (uint32_t)__shfl((int)k2, src);
          const uint32_t k = helper ? k2_src : k1;
          bool hit = false;
          if (leaf || helper) {
            float4 a, b, c;
            load_triangle(sc, L.top_addr, k, a, b, c);
            float t, u, v;
            hit = tri_test<false>(rr, V3(a.x, a.y, a.z), V3(b.x, b.y, b.z), V3(c.x, c.y, c.z), t, u, v) && t > 0.0f && t < tmax_src;
          }
          const int hsrc = paired ? (int)tab_helper[rank_more] : (int)lane;
          const int h2 = __shfl(hit ? 1 : 0, hsrc);
          if (leaf) {
            ++nt;
            if (hit) { occluded = true; cur = CUR_FINISHED; }
            else if (paired) {
              ++nt; tmask = tmask_after;
              if (h2) { occluded = true; cur = CUR_FINISHED; }
            }
            if (cur == CUR_LEAF && (tmask & 255u) == 0u) cur = advance<false>(gbase, gmask, st, nullptr, 0u);
          }
        }
      }
#endif
      if (cur == CUR_FINISHED) {
        if (DEBUG_OUT) debug_out[ri] = occluded ? 1 : 0;
        else if (!occluded) {
          const float4 Cq = q.shadow.C[ri];
          float4 L = q.lpath[path];
          L.x = L.x + Cq.x; L.y = L.y + Cq.y; L.z = L.z + Cq.z;
          q.lpath[path] = L;
        }
        cur = CUR_DONE;
      }
    }
  }
  {
    __shared__ unsigned long long s_stat[TRACE_WAVES][3];
    const unsigned long long c_tris = wave_sum(nt);
    if (lane == 0) { s_stat[wave][0] = nv; s_stat[wave][1] = c_tris; s_stat[wave][2] = nr; }
    __syncthreads();
    if (threadIdx.x < 3u) {
      unsigned long long v = 0;
      for (int w = 0; w < TRACE_WAVES; ++w) v += s_stat[w][threadIdx.x];
      const int at = threadIdx.x == 0u ? ST_NODES_A : threadIdx.x == 1u ? ST_TRIS_A : ST_SHADOW;
      if (v) atomicAdd(&q.stats[at * ST_STRIDE], v);
    }
  }
}

// =================================================================================================
// R7 sampler: NEAREST, REPEAT, no mips (the reference creates its samplers with default create-info, gltf/Asset.cpp:116-117);
// RGBA8 UNORM texel → float / 255.
// ---- address spaces of k_shade's scene accesses ---------------------------------------------------------------------------------------
// k_shade receives the scene through a pointer (DevScene lies in device memory), so the array pointers inside it are values LOADED from memory and
// the compiler has to treat them as generic: every gather became a FLAT load, which takes a slot in the LDS queue as well as in the vector-memory
// queue and, because the two return out of order, can only be waited for with vmcnt(0) lgkmcnt(0) — no partial waits, no two chains of loads in
// flight.  GScene is the same scene with its arrays typed as what they are (global memory); Dual is a table that lies in LDS when it fits and in
// global memory when it does not, each behind its own address space.  The helpers below are templates over the scene type, so that
// k_shade_raster (DevScene by value: kernel-argument pointers, known to be global) uses them unchanged.
typedef float fx4 __attribute__((ext_vector_type(4)));
typedef uint32_t ux4 __attribute__((ext_vector_type(4)));
typedef int ix4 __attribute__((ext_vector_type(4)));
#define AS_GLOBAL __attribute__((address_space(1)))
#define AS_LDS __attribute__((address_space(3)))
PT_DEV float4 ld4(const float4* p, size_t i) { return p[i]; }
PT_DEV float4 ld4(AS_GLOBAL const fx4* p, size_t i) { const fx4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
PT_DEV float4 ld4(AS_LDS const fx4* p, size_t i) { const fx4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
PT_DEV uint4 ldu4(const uint4* p, size_t i) { return p[i]; }
PT_DEV uint4 ldu4(AS_GLOBAL const ux4* p, size_t i) { const ux4 v = p[i]; return make_uint4(v.x, v.y, v.z, v.w); }
PT_DEV int4 ldi4(const int4* p, size_t i) { return p[i]; }
PT_DEV int4 ldi4(AS_GLOBAL const ix4* p, size_t i) { const ix4 v = p[i]; return make_int4(v.x, v.y, v.z, v.w); }
// LDS_ONLY: the table is known (on the host, at launch) to lie in LDS — no fallback pointer to keep in scalar registers, no uniform branch per access
template <class T, bool LDS_ONLY> struct Dual {
  AS_LDS const T* l; AS_GLOBAL const T* g; bool in_lds;
  PT_DEV T operator[](size_t i) const { return (LDS_ONLY || in_lds) ? l[i] : g[i]; }
};
template <bool LDS_ONLY> PT_DEV float4 ld4(const Dual<fx4, LDS_ONLY>& p, size_t i) { const fx4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
struct GScene {
  AS_GLOBAL const fx4* shade; uint32_t shade_stride;
  AS_GLOBAL const uint32_t* texels; AS_GLOBAL const ix4* tex_info; AS_GLOBAL const ux4* set_texels; AS_GLOBAL const ix4* set_info; int tex_linear;
  AS_GLOBAL const fx4* env; AS_GLOBAL const float* env_cond; AS_GLOBAL const uint16_t* env_cond_guide; int env_w, env_h, env_ok;
  uint32_t n_lights, n_mats; float ray_eps;
};
PT_DEV GScene global_view(const DevScene& d) {
  GScene g;
  g.shade = (AS_GLOBAL const fx4*)d.shade; g.shade_stride = d.shade_stride;
  g.texels = (AS_GLOBAL const uint32_t*)d.texels; g.tex_info = (AS_GLOBAL const ix4*)d.tex_info;
  g.set_texels = (AS_GLOBAL const ux4*)d.set_texels; g.set_info = (AS_GLOBAL const ix4*)d.set_info; g.tex_linear = d.tex_linear;
  g.env = (AS_GLOBAL const fx4*)d.env; g.env_cond = (AS_GLOBAL const float*)d.env_cond; g.env_cond_guide = (AS_GLOBAL const uint16_t*)d.env_cond_guide;
  g.env_w = d.env_w; g.env_h = d.env_h; g.env_ok = d.env_ok;
  g.n_lights = d.n_lights; g.n_mats = d.n_mats; g.ray_eps = d.ray_eps;
  return g;
}

template <class S> PT_DEV float4 texel_rgba(const S& sc, size_t at) {
  const uint32_t p = sc.texels[at];
  return make_float4((float)(p & 255u) / 255.0f, (float)((p >> 8) & 255u) / 255.0f, (float)((p >> 16) & 255u) / 255.0f, (float)(p >> 24) / 255.0f);
}
template <class S> PT_DEV float4 tex_fetch(const S& sc, int tex, float u, float v) {
  const int4 ti = ldi4(sc.tex_info, (size_t)tex);
  const float fu = u - __builtin_floorf(u), fv = v - __builtin_floorf(v);
  if (!sc.tex_linear) {
    int x = (int)(fu * (float)ti.y), y = (int)(fv * (float)ti.z);
    if (x > ti.y - 1) x = ti.y - 1;
    if (y > ti.z - 1) y = ti.z - 1;
    return texel_rgba(sc, (size_t)ti.x + (size_t)y * (size_t)ti.y + (size_t)x);
  }
  // PTC_FILTER_LINEAR: texel centres at i + 0.5, REPEAT wrap, lerp(a, b, t) = fma(t, b - a, a), x then y
  const float x = pt_fma(fu, (float)ti.y, -0.5f), y = pt_fma(fv, (float)ti.z, -0.5f);
  const float x0f = __builtin_floorf(x), y0f = __builtin_floorf(y);
  const float tx = x - x0f, ty = y - y0f;
  int x0 = (int)x0f, y0 = (int)y0f;
  int x1 = x0 + 1, y1 = y0 + 1;
  if (x0 < 0) x0 += ti.y;
  if (y0 < 0) y0 += ti.z;
  if (x1 > ti.y - 1) x1 -= ti.y;
  if (y1 > ti.z - 1) y1 -= ti.z;
  const size_t r0 = (size_t)ti.x + (size_t)y0 * (size_t)ti.y, r1 = (size_t)ti.x + (size_t)y1 * (size_t)ti.y;
  const float4 c00 = texel_rgba(sc, r0 + (size_t)x0), c10 = texel_rgba(sc, r0 + (size_t)x1), c01 = texel_rgba(sc, r1 + (size_t)x0), c11 = texel_rgba(sc, r1 + (size_t)x1);
  const float ax = pt_fma(tx, c10.x - c00.x, c00.x), ay = pt_fma(tx, c10.y - c00.y, c00.y), az = pt_fma(tx, c10.z - c00.z, c00.z), aw = pt_fma(tx, c10.w - c00.w, c00.w);
  const float bx = pt_fma(tx, c11.x - c01.x, c01.x), by = pt_fma(tx, c11.y - c01.y, c01.y), bz = pt_fma(tx, c11.z - c01.z, c01.z), bw = pt_fma(tx, c11.w - c01.w, c01.w);
  return make_float4(pt_fma(ty, bx - ax, ax), pt_fma(ty, by - ay, ay), pt_fma(ty, bz - az, az), pt_fma(ty, bw - aw, aw));
}

// Texel address inside a texture set: 8x8 tiles row-major over the image, Morton order inside a tile (ptc_scene.cpp).
PT_DEV size_t set_texel_at(int4 si, int x, int y) {
  const uint32_t lx = (uint32_t)x & 7u, ly = (uint32_t)y & 7u;
  const uint32_t mo = (lx & 1u) | ((ly & 1u) << 1) | ((lx & 2u) << 1) | ((ly & 2u) << 2) | ((lx & 4u) << 2) | ((ly & 4u) << 3);
  return (size_t)si.x + ((size_t)((uint32_t)y >> 3) * (size_t)si.w + ((uint32_t)x >> 3)) * 64u + mo;
}
PT_DEV float4 unorm8x4(uint32_t p) {
  return make_float4((float)(p & 255u) / 255.0f, (float)((p >> 8) & 255u) / 255.0f, (float)((p >> 16) & 255u) / 255.0f, (float)(p >> 24) / 255.0f);
}
// The three texels (colour, normal, metal-rough) of a texture set at (u, v): ONE 16-byte gather per tap instead of three 4-byte gathers from
// three textures.  Same arithmetic per texture as tex_fetch (the set's textures share one size, so the tap coordinates are the same).
template <class S> PT_DEV void set_fetch(const S& sc, int4 si, float u, float v, float4& c, float4& n, float4& m) {
  const float fu = u - __builtin_floorf(u), fv = v - __builtin_floorf(v);
  if (!sc.tex_linear) {
    int x = (int)(fu * (float)si.y), y = (int)(fv * (float)si.z);
    if (x > si.y - 1) x = si.y - 1;
    if (y > si.z - 1) y = si.z - 1;
    const uint4 t = ldu4(sc.set_texels, set_texel_at(si, x, y));
    c = unorm8x4(t.x); n = unorm8x4(t.y); m = unorm8x4(t.z);
    return;
  }
  const float x = pt_fma(fu, (float)si.y, -0.5f), y = pt_fma(fv, (float)si.z, -0.5f);
  const float x0f = __builtin_floorf(x), y0f = __builtin_floorf(y);
  const float tx = x - x0f, ty = y - y0f;
  int x0 = (int)x0f, y0 = (int)y0f;
  int x1 = x0 + 1, y1 = y0 + 1;
  if (x0 < 0) x0 += si.y;
  if (y0 < 0) y0 += si.z;
  if (x1 > si.y - 1) x1 -= si.y;
  if (y1 > si.z - 1) y1 -= si.z;
  const uint4 t00 = ldu4(sc.set_texels, set_texel_at(si, x0, y0)), t10 = ldu4(sc.set_texels, set_texel_at(si, x1, y0)),
              t01 = ldu4(sc.set_texels, set_texel_at(si, x0, y1)), t11 = ldu4(sc.set_texels, set_texel_at(si, x1, y1));
  auto bil = [&](uint32_t p00, uint32_t p10, uint32_t p01, uint32_t p11) {
    const float4 c00 = unorm8x4(p00), c10 = unorm8x4(p10), c01 = unorm8x4(p01), c11 = unorm8x4(p11);
    const float ax = pt_fma(tx, c10.x - c00.x, c00.x), ay = pt_fma(tx, c10.y - c00.y, c00.y), az = pt_fma(tx, c10.z - c00.z, c00.z), aw = pt_fma(tx, c10.w - c00.w, c00.w);
    const float bx = pt_fma(tx, c11.x - c01.x, c01.x), by = pt_fma(tx, c11.y - c01.y, c01.y), bz = pt_fma(tx, c11.z - c01.z, c01.z), bw = pt_fma(tx, c11.w - c01.w, c01.w);
    return make_float4(pt_fma(ty, bx - ax, ax), pt_fma(ty, by - ay, ay), pt_fma(ty, bz - az, az), pt_fma(ty, bw - aw, aw));
  };
  c = bil(t00.x, t10.x, t01.x, t11.x); n = bil(t00.y, t10.y, t01.y, t11.y); m = bil(t00.z, t10.z, t01.z, t11.z);
}

// Textured surface attributes of primitive `prim` at barycentrics (hu, hv): base colour × texel, metallic-roughness texel
// (glTF: G = roughness, B = metallic), tangent-space normal map through the interpolated TBN (fragment.glsl:24-30).
// ni = interpolated (un-normalised) vertex normal.  Updates base/metallic/roughness/ns in place.  `set` is the material's texture
// set (M3.x): when the set has an interleaved copy the three texels come with one gather, else one fetch per texture.
template <class S> PT_DEV void apply_textures(const S& sc, float4 t0, float4 t1, float4 t2, float4 t3, float4 t4, float4 t5, float hu, float hv, float hw, float4 M2, int set, v3 ni, float base[4],
                           float& metallic, float& roughness, v3& ns) {
  const int tex_color = __float_as_int(M2.y), tex_normal = __float_as_int(M2.z), tex_mr = __float_as_int(M2.w);
  const float tu = pt_fma(t1.x, hv, pt_fma(t0.z, hu, t0.x * hw)), tv = pt_fma(t1.y, hv, pt_fma(t0.w, hu, t0.y * hw));
  float4 cc = make_float4(1, 1, 1, 1), cn = make_float4(0.5f, 0.5f, 1.0f, 1.0f), cm = make_float4(1, 1, 1, 1);
  const int4 si = ldi4(sc.set_info, (size_t)(set < 0 ? 0 : set));
  if (set >= 0 && si.x >= 0) set_fetch(sc, si, tu, tv, cc, cn, cm);
  else {
    if (tex_color >= 0) cc = tex_fetch(sc, tex_color, tu, tv);
    if (tex_mr >= 0) cm = tex_fetch(sc, tex_mr, tu, tv);
    if (tex_normal >= 0) cn = tex_fetch(sc, tex_normal, tu, tv);
  }
  if (tex_color >= 0) { base[0] = base[0] * cc.x; base[1] = base[1] * cc.y; base[2] = base[2] * cc.z; base[3] = base[3] * cc.w; }
  if (tex_mr >= 0) { roughness = roughness * cm.y; metallic = metallic * cm.z; }
  if (tex_normal >= 0) {
    const float nx = 2.0f * cn.x - 1.0f, ny = 2.0f * cn.y - 1.0f, nz = 2.0f * cn.z - 1.0f;
    // tangents: a = (t1.z, t1.w, t2.x) b = (t2.y, t2.z, t2.w) c = (t3.x, t3.y, t3.z); bitangents: a = (t3.w, t4.x, t4.y) b = (t4.z, t4.w, t5.x) c = (t5.y, t5.z, t5.w)
    const v3 ti = V3(pt_fma(t3.x, hv, pt_fma(t2.y, hu, t1.z * hw)), pt_fma(t3.y, hv, pt_fma(t2.z, hu, t1.w * hw)), pt_fma(t3.z, hv, pt_fma(t2.w, hu, t2.x * hw)));
    const v3 bi = V3(pt_fma(t5.y, hv, pt_fma(t4.z, hu, t3.w * hw)), pt_fma(t5.z, hv, pt_fma(t4.w, hu, t4.x * hw)), pt_fma(t5.w, hv, pt_fma(t5.x, hu, t4.y * hw)));
    ns = normalize3(vfma(ti, nx, vfma(bi, ny, ni * nz)));
  }
}

// Lat-long environment: radiance and solid-angle pdf of a unit direction (piecewise-constant texels).
template <class S> PT_DEV void env_lookup(const S& sc, v3 d, v3& Le, float& pdf) {
  const float u = pt_atan2(d.z, d.x) * (0.5f * PT_INV_PI) + 0.5f;
  const float dy = fmin2(fmax2(d.y, -1.0f), 1.0f);
  const float st = pt_sqrt(fmax2(0.0f, 1.0f - dy * dy));
  const float vv = pt_atan2(st, dy) * PT_INV_PI;
  int x = (int)(u * (float)sc.env_w), y = (int)(vv * (float)sc.env_h);
  if (x > sc.env_w - 1) x = sc.env_w - 1;
  if (x < 0) x = 0;
  if (y > sc.env_h - 1) y = sc.env_h - 1;
  if (y < 0) y = 0;
  const float4 t = ld4(sc.env, (size_t)y * (size_t)sc.env_w + (size_t)x);
  Le = V3(t.x, t.y, t.z);
  pdf = sc.env_ok ? (t.w * (float)sc.env_w * (float)sc.env_h) / (2.0f * PT_PI * PT_PI * fmax2(st, 1e-6f)) : 0.0f;
}
template <class P> PT_DEV uint32_t cdf_search(const P& cdf, uint32_t n, float r) {
  uint32_t lo = 0, hi = n - 1u;
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cdf[mid] > r) hi = mid; else lo = mid + 1u; }
  return lo;
}
// Importance-sample the environment: row by the marginal cdf, column by the row's conditional cdf, uniform inside the texel.
// cdf_search over [guide[b], guide[b + 1]], b = floor(PTC_ENV_GUIDE r): the same index as the search over the whole table (the answer is monotonic in r
// and guide[b] is the answer for r = b/64), after 2 + log2(range) instead of log2(n) dependent loads
template <class P, class G> PT_DEV uint32_t cdf_search_guided(const P& cdf, const G& guide, float r) {
  const uint32_t b = (uint32_t)(r * (float)PTC_ENV_GUIDE);
  uint32_t lo = guide[b], hi = guide[b + 1u];
  while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (cdf[mid] > r) hi = mid; else lo = mid + 1u; }
  return lo;
}
// marg / marg_guide: the row cdf and its guide — the scene's arrays, or k_shade's copies in LDS
template <class S, class P, class G> PT_DEV v3 env_sample(const S& sc, const P& marg, const G& marg_guide, float r1, float r2) {
  const uint32_t y = cdf_search_guided(marg, marg_guide, r1);
  const float m0 = y ? marg[y - 1u] : 0.0f, m1 = marg[y];
  float xi_v = m1 > m0 ? (r1 - m0) / (m1 - m0) : 0.5f;
  const auto cc = sc.env_cond + (size_t)y * (size_t)sc.env_w;
  const uint32_t x = cdf_search_guided(cc, sc.env_cond_guide + (size_t)y * (PTC_ENV_GUIDE + 1u), r2);
  const float c0 = x ? cc[x - 1u] : 0.0f, c1 = cc[x];
  float xi_u = c1 > c0 ? (r2 - c0) / (c1 - c0) : 0.5f;
  xi_u = fmin2(fmax2(xi_u, 0.0f), 0.999999f); xi_v = fmin2(fmax2(xi_v, 0.0f), 0.999999f);
  const float u = ((float)x + xi_u) / (float)sc.env_w, vv = ((float)y + xi_v) / (float)sc.env_h;
  float s2, c2, st, ct;
  sincos2pi(u, s2, c2);             // phi = 2 pi u - pi: cos phi = -cos(2 pi u), sin phi = -sin(2 pi u)
  sincos2pi(0.5f * vv, st, ct);     // theta = pi v
  return V3(st * -c2, ct, st * -s2);
}

// =================================================================================================
// P9 + P5–P8 shading.  One WAVE owns one queue segment (ptc_internal.h, "SEGMENTED queues"): no block barrier and no atomic anywhere
// in the loop, so the four waves a SIMD holds are four independent streams of loads and arithmetic (the round-2 kernel kept the
// eight waves of a 512-thread block in lockstep at three barriers per window).  That alone changed nothing: the kernel runs at the rate of
// the CUs' vector-memory path (two waves per SIMD are 93 % as fast as four), so what pays is fewer bytes and fewer gathers.
// SORT = false (default): a batch is simply the next 64 slots of the segment, coalesced 64-byte ray + hit records.
// SORT = true (P9's material sort per wave): the wave reads the class words of the next 64 slots of its segment, and appends each slot to the ring
// of its class in LDS (ballot + mbcnt prefix per class present); as soon as a ring holds 64 slots they are shaded as ONE batch, so every
// batch but a segment's last few is uniform in class (Lambert / GGX / environment miss; 8 classes fit the hit word).  Misses without an
// environment are dropped here.
// P5–P8: surface reconstruction, emission with MIS, next-event estimation, BSDF sampling, Russian roulette.
// P9, compaction: continuation rays and shadow rays of a batch go to the front of the wave's OWN segment of the output arrays (ballot +
// mbcnt prefix behind a wave-private cursor): a wave never writes more records than it has read, so the segment cannot overflow.
#define SHADE_RING 128            // slots a class ring holds: a batch is taken as soon as 64 are there, so at most 63 + 64 wait
// (Emptying the rings every 128 / 256 / 512 / 1024 slots, so that a slot's record is read soon after its neighbours', was slower: k_shade +6 % / +3.5 % /
// +1 % / +1 % on the atrium and +44 % / +17 % / +11 % on the textured atrium — partial batches cost more than the re-read lines: profiles/r03_shade_variants.txt.)
PT_DEV void wave_lds_sync() {     // LDS written by some lanes of this wave is read by others
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool SORT, bool TABLES_LDS>
__global__ __launch_bounds__(SHADE_BLOCK, SHADE_MIN_WAVES) void k_shade(const DevScene* __restrict__ scp, DevFrame fr, DevQueues q, int qi, uint32_t b) {
  // small scene-wide tables staged once per block: emitter records + power cdf, materials
  __shared__ float4 s_light[SHADE_LDS_LIGHTS * 5];
  __shared__ float s_cdf[SHADE_LDS_LIGHTS];
  __shared__ float4 s_mat[SHADE_LDS_MATS * 4];
  __shared__ uint32_t s_ring[SORT ? SHADE_WAVES : 1][PTC_MATERIAL_CLASSES][SORT ? SHADE_RING : 1];
  __shared__ float s_marg[SHADE_LDS_ENV_ROWS];                   // the environment's row cdf and its guide, when the map is at most this high
  __shared__ uint16_t s_marg_guide[PTC_ENV_GUIDE + 2];
  const DevScene& dsc = *scp;
  const GScene sc = global_view(dsc);
  const uint32_t lane = lane_id();
  const uint32_t wave = threadIdx.x >> 6;
  const bool lds_lights = TABLES_LDS || sc.n_lights <= SHADE_LDS_LIGHTS, lds_mats = TABLES_LDS || sc.n_mats <= SHADE_LDS_MATS;
  if (lds_lights) {
    for (uint32_t i = threadIdx.x; i < sc.n_lights * 5u; i += SHADE_BLOCK) s_light[i] = ld4((AS_GLOBAL const fx4*)dsc.lights, i);
    for (uint32_t i = threadIdx.x; i < sc.n_lights; i += SHADE_BLOCK) s_cdf[i] = ((AS_GLOBAL const float*)dsc.cdf)[i];
  }
  if (lds_mats)
    for (uint32_t i = threadIdx.x; i < sc.n_mats * 4u; i += SHADE_BLOCK) s_mat[i] = ld4((AS_GLOBAL const fx4*)dsc.mats, i);
  const bool lds_marg = sc.env_ok != 0 && (TABLES_LDS || sc.env_h <= SHADE_LDS_ENV_ROWS);
  if (lds_marg) {
    for (uint32_t i = threadIdx.x; i < (uint32_t)sc.env_h; i += SHADE_BLOCK) s_marg[i] = ((AS_GLOBAL const float*)dsc.env_marg)[i];
    for (uint32_t i = threadIdx.x; i <= PTC_ENV_GUIDE; i += SHADE_BLOCK) s_marg_guide[i] = ((AS_GLOBAL const uint16_t*)dsc.env_marg_guide)[i];
  }
  __syncthreads();
  const Dual<float, TABLES_LDS> env_marg = {(AS_LDS const float*)s_marg, (AS_GLOBAL const float*)dsc.env_marg, lds_marg};
  const Dual<uint16_t, TABLES_LDS> env_marg_guide = {(AS_LDS const uint16_t*)s_marg_guide, (AS_GLOBAL const uint16_t*)dsc.env_marg_guide, lds_marg};
  const uint32_t seg = blockIdx.x * SHADE_WAVES + wave;
  if (seg >= q.n_seg) return;
  const Dual<fx4, TABLES_LDS> lights = {(AS_LDS const fx4*)s_light, (AS_GLOBAL const fx4*)dsc.lights, lds_lights};
  const Dual<float, TABLES_LDS> cdf = {(AS_LDS const float*)s_cdf, (AS_GLOBAL const float*)dsc.cdf, lds_lights};
  const Dual<fx4, TABLES_LDS> mats = {(AS_LDS const fx4*)s_mat, (AS_GLOBAL const fx4*)dsc.mats, lds_mats};
  // light-kind selection probabilities for NEE: environment vs emissive triangles
  const bool has_env = sc.env_w > 0, env_nee = has_env && sc.env_ok != 0;
  const float p_env = env_nee ? (sc.n_lights > 0u ? 0.5f : 1.0f) : 0.0f, p_area = 1.0f - p_env;
  const RayQ rin = q.ray[qi], rout = q.ray[qi ^ 1];
  const uint32_t base = seg * q.seg_len;                       // first slot of the segment, in every queue array
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.seg_ray[qi][seg]);
  typedef __attribute__((address_space(3))) volatile uint32_t lds_u32;
  lds_u32* ring = (lds_u32*)&s_ring[SORT ? wave : 0][0][0];
#ifdef PT_STAMP_SHADE   // wave-cycles per phase (tools/stamp_shade.py)
  unsigned long long t_front = 0, t_load = 0, t_math = 0, t_back = 0, t_mark = __builtin_amdgcn_s_memtime();
#define SSTAMP(acc) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - t_mark; t_mark = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define SSTAMP_LOADS(acc) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SSTAMP(acc); } while (0)
#else
#define SSTAMP(acc) do { } while (0)
#define SSTAMP_LOADS(acc) do { } while (0)
#endif
  uint32_t out_a = 0, out_s = 0;          // wave cursors: continuation / shadow rays written so far
  uint32_t fillv = 0;                     // lane c holds the number of slots waiting in ring c
  uint32_t ready = 0, nonempty = 0;       // class masks: ring holds >= 64 / > 0 slots
  uint32_t i0 = 0;
  int w_pref = SORT && lane < n ? __float_as_int(q.hit[base + lane].y) : -1;    // class word of the group to be sorted next
  for (;;) {
    bool valid; uint32_t item = 0u;
    if (!SORT) {          // a batch is the next 64 slots of the segment, whatever their materials
      if (i0 >= n) break;
      valid = i0 + lane < n; item = i0 + lane;
      i0 += 64u;
    } else {
    if (!ready) {
      if (i0 < n) {
        // ---- P9 front end: sort the next 64 slots into the class rings ----
        const uint32_t i = i0 + lane;
        int cls = -1;
        if (i < n) cls = w_pref >= 0 ? ((w_pref >> HIT_CLASS_SHIFT) & (PTC_MATERIAL_CLASSES - 1)) : (has_env ? PTC_MATERIAL_CLASSES - 1 : -1);
#ifdef SHADE_TWO_RING     // experiment (profiles/r04_shade_variants.txt): two rings only — textured hits / everything else — the split the hit word's class >= 2 gives
        if (cls >= 0) cls = (cls >= 2 && cls != PTC_MATERIAL_CLASSES - 1) ? 1 : 0;
#endif
        i0 += 64u;
        w_pref = i0 + lane < n ? __float_as_int(q.hit[base + i0 + lane].y) : -1;
        uint64_t rem = __ballot(cls >= 0);
        while (rem) {
          const int c = __builtin_amdgcn_readlane(cls, __ffsll((unsigned long long)rem) - 1);
          const uint64_t m = __ballot(cls == c);
          uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)fillv, c);
          if (cls == c) ring[(uint32_t)c * SHADE_RING + f + mbcnt64(m)] = i;
          f += (uint32_t)__popcll(m);
          if (lane == (uint32_t)c) fillv = f;
          nonempty |= 1u << c;
          if (f >= 64u) ready |= 1u << c;
          rem &= ~m;
        }
        SSTAMP(t_front);
        continue;
      }
      if (!nonempty) break;
    }
    // ---- take a batch: 64 slots of one class; at the end of the segment what is left in ALL rings goes out in mixed batches (the shading
    // code does not need a uniform class), so that a segment ends with one partial batch, not one per class ----
    wave_lds_sync();
    if (ready) {
      const uint32_t c = (uint32_t)__builtin_ctz(ready);
      const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)fillv, (int)c);     // 64 <= f <= 127
      const uint32_t rest = f - 64u;
      valid = true;
      item = ring[c * SHADE_RING + rest + lane];                                      // the 64 youngest: nothing has to move
      if (lane == c) fillv = rest;
      ready &= ~(1u << c);
      if (!rest) nonempty &= ~(1u << c);
    } else {
      uint32_t off = 0, left = nonempty;
      valid = false;
      while (left && off < 64u) {
        const uint32_t c = (uint32_t)__builtin_ctz(left);
        const uint32_t f = (uint32_t)__builtin_amdgcn_readlane((int)fillv, (int)c);   // 1 <= f <= 63
        const uint32_t take = f < 64u - off ? f : 64u - off;
        if (lane >= off && lane < off + take) { item = ring[c * SHADE_RING + f - take + (lane - off)]; valid = true; }
        if (lane == c) fillv = f - take;
        if (f == take) nonempty &= ~(1u << c);
        left &= ~(1u << c);
        off += take;
      }
    }
    }   // SORT
    const uint32_t slot = base + item;
    SSTAMP(t_front);

    bool alive = false, has_shadow = false;
    float4 oA, oB, oC, sA, sB, sC;
    oA = oB = oC = sA = sB = sC = make_float4(0, 0, 0, 0);
    if (valid) {
      const float4 A = rin.A[slot], Bq = rin.B[slot], Cq = rin.C[slot], H = q.hit[slot];
      const v3 d = V3(A.w, Bq.x, Bq.y);
      v3 T = V3(Bq.z, Bq.w, Cq.x);
      const float prev_pdf = Cq.y;
      const uint32_t path = __float_as_uint(Cq.z), key = __float_as_uint(Cq.w);
      const float ht = H.x, hu = H.z, hv = H.w;
      if (__float_as_int(H.y) < 0) {                                   // miss: environment radiance, MIS against env NEE
        if (SORT || has_env) {                                          // (the sort has dropped the misses of a scene without environment already)
        v3 Le; float pe; env_lookup(sc, d, Le, pe);
        float wgt = 1.0f;
        if (b > 0u) { const float pl = pe * p_env; const float pb2 = prev_pdf * prev_pdf; wgt = pb2 / pt_fma(pl, pl, pb2); }
        float4 L = q.lpath[path];
        L.x = pt_fma(T.x * Le.x, wgt, L.x); L.y = pt_fma(T.y * Le.y, wgt, L.y); L.z = pt_fma(T.z * Le.z, wgt, L.z);
        q.lpath[path] = L;
        }
      } else {
      const uint32_t prim = (uint32_t)__float_as_int(H.y) & ((1u << HIT_CLASS_SHIFT) - 1u);
      // ---- P5 surface reconstruction from the primitive's shading record (five 16-byte loads) ----
      const size_t rec = (size_t)prim * sc.shade_stride;
      const float4 r0 = ld4(sc.shade, rec), r1 = ld4(sc.shade, rec + 1), r2 = ld4(sc.shade, rec + 2), r3 = ld4(sc.shade, rec + 3), r4 = ld4(sc.shade, rec + 4);
      // a textured class (the hit word says so): the record's other six units leave with the first five, not one round trip later behind the material id
      const bool tex_cls = ((uint32_t)__float_as_int(H.y) >> HIT_CLASS_SHIFT) >= 2u;
      float4 x0, x1, x2, x3, x4, x5;
      x0 = x1 = x2 = x3 = x4 = x5 = make_float4(0, 0, 0, 0);
      if (tex_cls) { x0 = ld4(sc.shade, rec + 5); x1 = ld4(sc.shade, rec + 6); x2 = ld4(sc.shade, rec + 7); x3 = ld4(sc.shade, rec + 8); x4 = ld4(sc.shade, rec + 9); x5 = ld4(sc.shade, rec + 10); }
      // P7's environment sample depends on the path's random numbers only: its chain of dependent loads (guide, column cdf) runs while the record gather is in flight
      const uint32_t rb = b + 1u;
      bool use_env = false;
      if ((int)b < fr.max_bounces && env_nee) use_env = sc.n_lights == 0u || rng_f(key, rb, 7) < p_env;
      v3 wi_env = V3(0, 0, 0);
#ifdef SHADE_ENV_ALWAYS    // experiment (profiles/r04_shade_variants.txt): the environment sample's chain of loads issued by every lane, so that the env / area choice diverges in arithmetic only
      if ((int)b < fr.max_bounces && env_nee) wi_env = env_sample(sc, env_marg, env_marg_guide, rng_f(key, rb, 1), rng_f(key, rb, 2));
#else
      if (use_env) wi_env = env_sample(sc, env_marg, env_marg_guide, rng_f(key, rb, 1), rng_f(key, rb, 2));
#endif
      SSTAMP_LOADS(t_load);
      const v3 Pa = V3(r0.x, r0.y, r0.z), Pb = V3(r1.x, r1.y, r1.z), Pc = V3(r2.x, r2.y, r2.z);
      const v3 Na = V3(r2.w, r3.x, r3.y), Nb = V3(r3.z, r3.w, r4.x), Nc = V3(r4.y, r4.z, r4.w);
      const float hw = 1.0f - hu - hv;
      const v3 P = V3(pt_fma(Pc.x, hv, pt_fma(Pb.x, hu, Pa.x * hw)), pt_fma(Pc.y, hv, pt_fma(Pb.y, hu, Pa.y * hw)), pt_fma(Pc.z, hv, pt_fma(Pb.z, hu, Pa.z * hw)));
      v3 ng = normalize3(cross3(Pb - Pa, Pc - Pa));
      const v3 ni = V3(pt_fma(Nc.x, hv, pt_fma(Nb.x, hu, Na.x * hw)), pt_fma(Nc.y, hv, pt_fma(Nb.y, hu, Na.y * hw)), pt_fma(Nc.z, hv, pt_fma(Nb.z, hu, Na.z * hw)));
      v3 ns = normalize3(ni);
      const int mat = __float_as_int(r0.w);
      const float4 M0 = ld4(mats, (size_t)(mat * 4 + 0)), M1 = ld4(mats, (size_t)(mat * 4 + 1)), M2 = ld4(mats, (size_t)(mat * 4 + 2));
      float base_c[4] = {M0.x, M0.y, M0.z, M2.x};
      float metallic = M0.w, roughness = M1.w;
      const bool lambert = metallic == 0.0f && roughness >= 1.0f && __float_as_int(M2.w) < 0;
      if (__float_as_int(M2.y) >= 0 || __float_as_int(M2.z) >= 0 || __float_as_int(M2.w) >= 0)
        apply_textures(sc, x0, x1, x2, x3, x4, x5, hu, hv, hw, M2, __float_as_int(ld4(mats, (size_t)(mat * 4 + 3)).x), ni, base_c, metallic, roughness, ns);
      const v3 wo = -d;
      const bool front = dot3(ng, wo) > 0.0f;
      if (dot3(ns, ng) < 0.0f) ns = -ns;
      if (!front) { ng = -ng; ns = -ns; }
      if (!(dot3(ns, wo) > 0.0f)) ns = ng;
      // ---- emission (one-sided), MIS against next-event estimation ----
      const int li = __float_as_int(r1.w);
      if (li >= 0 && front) {
        const float4 l0 = ld4(lights, (size_t)(li * 5 + 0)), l1 = ld4(lights, (size_t)(li * 5 + 1)), l4 = ld4(lights, (size_t)(li * 5 + 4));
        float wgt = 1.0f;
        if (b > 0u) {
          const float cosl = dot3(ng, wo);
          const float pl = ((l1.w * (ht * ht)) / (l0.w * cosl)) * p_area;
          const float pb2 = prev_pdf * prev_pdf;
          wgt = pb2 / pt_fma(pl, pl, pb2);
        }
        float4 L = q.lpath[path];
        L.x = pt_fma(T.x * l4.x, wgt, L.x); L.y = pt_fma(T.y * l4.y, wgt, L.y); L.z = pt_fma(T.z * l4.z, wgt, L.z);
        q.lpath[path] = L;
      }
      if ((int)b < fr.max_bounces) {
        const bsdf_t bs = make_bsdf(V3(base_c[0], base_c[1], base_c[2]), metallic, roughness, lambert);
        v3 tx, ty; onb(ns, tx, ty);
        const v3 wol = V3(dot3(tx, wo), dot3(ty, wo), dot3(ns, wo));
        const float ps = spec_prob(bs, fmax2(wol.z, 1e-4f));
        const v3 porg = vfma(ng, sc.ray_eps, P);
        // ---- P7 next-event estimation: one light sample per bounce, environment or emissive triangle ----
        if (use_env) {
          const v3 wi = wi_env;
          const v3 wil = V3(dot3(tx, wi), dot3(ty, wi), dot3(ns, wi));
          if (wil.z > 0.0f && dot3(ng, wi) > 0.0f) {
            v3 Le; float pe; env_lookup(sc, wi, Le, pe);
            const float pl = pe * p_env;
            if (pl > 0.0f) {
              v3 f; float pb; bsdf_eval(bs, wol, wil, ps, f, pb);
              const float pl2 = pl * pl;
              const float wgt = pl2 / pt_fma(pb, pb, pl2);
              const float k = (wil.z * wgt) / pl;
              has_shadow = true;
              sA = make_float4(porg.x, porg.y, porg.z, wi.x);
              sB = make_float4(wi.y, wi.z, PT_T_INF, __uint_as_float(path));
              sC = make_float4(T.x * f.x * Le.x * k, T.y * f.y * Le.y * k, T.z * f.z * Le.z * k, 0.0f);
            }
          }
        } else if (sc.n_lights > 0u) {
          const float u0 = rng_f(key, rb, 0), r1 = rng_f(key, rb, 1), r2 = rng_f(key, rb, 2);
          const uint32_t lo = cdf_search(cdf, sc.n_lights, u0);
          const float4 l0 = ld4(lights, (size_t)(lo * 5 + 0)), l1 = ld4(lights, (size_t)(lo * 5 + 1)), l2 = ld4(lights, (size_t)(lo * 5 + 2)), l3 = ld4(lights, (size_t)(lo * 5 + 3)), l4 = ld4(lights, (size_t)(lo * 5 + 4));
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
              const float pl = ((l1.w * dist2) / (l0.w * cosl)) * p_area;
              v3 f; float pb; bsdf_eval(bs, wol, wil, ps, f, pb);
              const float pl2 = pl * pl;
              const float wgt = pl2 / pt_fma(pb, pb, pl2);
              const float k = (wil.z * wgt) / pl;
              has_shadow = true;
              // visibility: the segment from the offset origin porg to the sampled point, minus its last 0.1 %
              const v3 sv = y - porg;
              const float sd = pt_sqrt(dot3(sv, sv));
              const v3 sdir = sv * (1.0f / sd);
              sA = make_float4(porg.x, porg.y, porg.z, sdir.x);
              sB = make_float4(sdir.y, sdir.z, sd * 0.999f, __uint_as_float(path));
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
        }
      }
      }   // hit
    }
    SSTAMP(t_math);
    // ---- P9 back end: compaction into the wave's own segment of the output arrays ----
    const uint64_t ms = __ballot(has_shadow), ma = __ballot(alive);
    if (has_shadow) { const uint32_t o = base + out_s + mbcnt64(ms); q.shadow.A[o] = sA; q.shadow.B[o] = sB; q.shadow.C[o] = sC; }
    if (alive) { const uint32_t o = base + out_a + mbcnt64(ma); rout.A[o] = oA; rout.B[o] = oB; rout.C[o] = oC; }
    out_s += (uint32_t)__popcll(ms); out_a += (uint32_t)__popcll(ma);
    SSTAMP(t_back);
  }
  if (lane == 0) { q.seg_ray[qi ^ 1][seg] = out_a; q.seg_sh[seg] = out_s; }
#ifdef PT_STAMP_SHADE
  if (lane == 0) { atomicAdd(&q.stats[ST_DIAG_NODE_ITERS * ST_STRIDE], t_front); atomicAdd(&q.stats[ST_DIAG_TRI_ITERS * ST_STRIDE], t_load); atomicAdd(&q.stats[ST_DIAG_LEAF_VISITS * ST_STRIDE], t_math);
                   atomicAdd(&q.stats[ST_DIAG_ROUNDS * ST_STRIDE], t_back); }
#endif
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
// GBUF16 (PTC_INTEGRATOR_RASTER_GBUFFER16): the lighting pass reads what the reference's G-buffer holds
// (GBuffer.hpp:13-16): positions and normals as RGBA16F (fp32 → fp16, round to nearest even), albedo as RGBA16 UNORM
// (clamp to [0,1], round(v·65535)/65535); the normal is NOT re-normalised after the rounding (lighting.glsl:21,28).
// v_cvt_f16_f32 (RTE, denormals kept) + v_cvt_f32_f16.  The empty asm hides the producer of `v`: without it the compiler fuses
// an fma that feeds the conversion into v_fma_mixlo_f16, which rounds the exact product-sum ONCE to fp16, while the contract
// (and the oracle) round it to fp32 first — 3 pixels of a 160x90 frame differed by one fp16 ulp.
PT_DEV float round_f16(float v) { asm volatile("" : "+v"(v)); return (float)(_Float16)v; }
PT_DEV float round_unorm16(float v) {
  const float c = fmin2(fmax2(v, 0.0f), 1.0f);                                       // NaN → 0 (fmax2(NaN, 0) = 0)
  return (float)(uint32_t)(c * 65535.0f + 0.5f) / 65535.0f;
}
template <bool GBUF16>
__global__ __launch_bounds__(256) void k_shade_raster(DevScene sc, DevCamera cam, DevFrame fr, DevQueues q, float4* accum) {
  const uint32_t j = blockIdx.x * 256u + threadIdx.x;
  if (j >= fr.n_owned) return;
  const float4 H = q.hit[j];
  float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // G-buffer clear → colour 0
  const int pc = __float_as_int(H.y);
  if (pc >= 0) {
    const int prim = pc & ((1 << HIT_CLASS_SHIFT) - 1);
    const float hu = H.z, hv = H.w, hw = 1.0f - hu - hv;
    const float4* rec = sc.shade + (size_t)prim * sc.shade_stride;
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4];
    v3 P = V3(pt_fma(r2.x, hv, pt_fma(r1.x, hu, r0.x * hw)), pt_fma(r2.y, hv, pt_fma(r1.y, hu, r0.y * hw)), pt_fma(r2.z, hv, pt_fma(r1.z, hu, r0.z * hw)));
    const v3 ni = V3(pt_fma(r4.y, hv, pt_fma(r3.z, hu, r2.w * hw)), pt_fma(r4.z, hv, pt_fma(r3.w, hu, r3.x * hw)), pt_fma(r4.w, hv, pt_fma(r4.x, hu, r3.y * hw)));
    v3 N = normalize3(ni);
    const int mat = __float_as_int(r0.w);
    const float4 M0 = sc.mats[mat * 4 + 0], M1 = sc.mats[mat * 4 + 1], M2 = sc.mats[mat * 4 + 2];
    float base[4] = {M0.x, M0.y, M0.z, M2.x};
    float metallic = M0.w, roughness = M1.w;
    if (__float_as_int(M2.y) >= 0 || __float_as_int(M2.z) >= 0 || __float_as_int(M2.w) >= 0)
      apply_textures(sc, rec[5], rec[6], rec[7], rec[8], rec[9], rec[10], hu, hv, hw, M2, __float_as_int(sc.mats[mat * 4 + 3].x), ni, base, metallic, roughness, N);
    if (GBUF16) {
      P = V3(round_f16(P.x), round_f16(P.y), round_f16(P.z));
      N = V3(round_f16(N.x), round_f16(N.y), round_f16(N.z));
#pragma unroll
      for (int k = 0; k < 4; ++k) base[k] = round_unorm16(base[k]);
    }
    const v3 V = normalize3(V3(cam.pos[0], cam.pos[1], cam.pos[2]) - P);
    const v3 Hh = normalize3(V + V);
    const float ndv = fmax2(dot3(N, V), 0.0f), ndh = fmax2(dot3(N, Hh), 0.0f);
    const float s2 = ndh * ndh, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8, s32 = s16 * s16, spec = s32 * s32;
    o = make_float4(pt_fma(base[0], ndv, spec), pt_fma(base[1], ndv, spec), pt_fma(base[2], ndv, spec), pt_fma(base[3], ndv, spec));
  }
  accum[j] = o;
}

// RGBA32F → RGBA16F (the reference's HdrImage / lighting-pass target format, PbrRenderSystem.hpp:21, HdrImage.cpp:20):
// round to nearest even, overflow → inf, half denormals kept.  One pixel (8 bytes out) per thread.
__global__ __launch_bounds__(256) void k_to_half(const float4* in, uint2* out, uint32_t n) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const float4 c = in[i];
  const _Float16 hx = (_Float16)c.x, hy = (_Float16)c.y, hz = (_Float16)c.z, hw = (_Float16)c.w;
  const uint32_t bx = __builtin_bit_cast(unsigned short, hx), by = __builtin_bit_cast(unsigned short, hy),
                 bz = __builtin_bit_cast(unsigned short, hz), bw = __builtin_bit_cast(unsigned short, hw);
  out[i] = make_uint2(bx | (by << 16), bz | (bw << 16));
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
static size_t trace_lds_bytes(const LaunchCfg& cfg, const DevScene& sc, bool closest) {
  return (size_t)sc.n_lds_units * 16 + (size_t)TRACE_WAVES * cfg.stack_lds * 64 * 8 + (closest ? ORDER_TABLE_BYTES : 0) + (size_t)TRACE_WAVES * RING_FIELDS * TRACE_RING * 4;
}

int pt_trace_block_threads() { return TRACE_BLOCK; }
#define PT_STR2(x) #x
#define PT_STR(x) PT_STR2(x)
// the compile-time half of the launch policy (ptc_launch_policy)
const char* pt_kernel_policy() {
  return "trace_block=" PT_STR(TRACE_BLOCK) " trace_min_waves=" PT_STR(TRACE_MIN_WAVES) " chunk=" PT_STR(TRACE_CHUNK) " ring=" PT_STR(TRACE_RING) " refill_idle=" PT_STR(TRACE_REFILL_IDLE)
         " node_min=" PT_STR(TRACE_NODE_MIN) " leaf_extra=" PT_STR(LEAF_EXTRA) " shade_block=" PT_STR(SHADE_BLOCK) " shade_min_waves=" PT_STR(SHADE_MIN_WAVES);
}

size_t pt_trace_lds_bytes(const LaunchCfg& cfg, const DevScene& sc) { return trace_lds_bytes(cfg, sc, true); }

// Resident blocks per CU of the trace kernels with `lds` bytes of dynamic LDS (registers, static LDS and the launch bounds
// included: the runtime's own occupancy calculation), minimum over the closest-hit and any-hit kernels; <= 0 on error.
// Dynamic LDS above the default 64 KiB limit is enabled on every variant first.
int pt_trace_blocks_per_cu(size_t lds) {
  const void* fns[] = {(const void*)k_trace_closest<false>, (const void*)k_trace_closest<true>, (const void*)k_trace_any<false>, (const void*)k_trace_any<true>};
  int best = 1 << 30;
  for (const void* f : fns) {
    if (lds > 64u * 1024u && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -1;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, TRACE_BLOCK, lds) != hipSuccess) return -1;
    if (nb < best) best = nb;
  }
  return best;
}

static uint32_t trace_waves(const LaunchCfg& cfg) { return (uint32_t)(cfg.n_cu * cfg.trace_blocks_per_cu) * TRACE_WAVES; }
void pt_launch_set_counts(hipStream_t s, const LaunchCfg& cfg, const DevQueues& q, uint32_t n_rays, uint32_t n_shadow) {
  hipLaunchKernelGGL(k_set_counts, dim3(1), dim3(SCAN_BLOCK), 0, s, q, n_rays, n_shadow, trace_waves(cfg));
}
void pt_launch_scan(hipStream_t s, const LaunchCfg& cfg, const DevQueues& q, int qi_next) {
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(SCAN_BLOCK), 0, s, q, qi_next, trace_waves(cfg));
}
int pt_shade_block_threads() { return SHADE_BLOCK; }

void pt_launch_raygen(hipStream_t s, const DevCamera& cam, const DevFrame& fr, const DevQueues& q, uint32_t first_sample, uint32_t n_samples, bool raster) {
  const uint32_t n_paths = fr.n_owned * n_samples;
  const dim3 grid((n_paths + 255u) / 256u);
  if (raster) hipLaunchKernelGGL(k_raygen<true>, grid, dim3(256), 0, s, cam, fr, q, first_sample, n_paths);
  else hipLaunchKernelGGL(k_raygen<false>, grid, dim3(256), 0, s, cam, fr, q, first_sample, n_paths);
}

void pt_launch_trace_closest(hipStream_t s, const LaunchCfg& cfg, const DevScene& sc, const DevQueues& q, int qi, bool cull) {
  const dim3 grid((unsigned)(cfg.n_cu * cfg.trace_blocks_per_cu));
  const size_t lds = trace_lds_bytes(cfg, sc, true);
  if (cull) hipLaunchKernelGGL(k_trace_closest<true>, grid, dim3(TRACE_BLOCK), lds, s, sc, q, qi, cfg.stack_lds);
  else hipLaunchKernelGGL(k_trace_closest<false>, grid, dim3(TRACE_BLOCK), lds, s, sc, q, qi, cfg.stack_lds);
}

void pt_launch_trace_any(hipStream_t s, const LaunchCfg& cfg, const DevScene& sc, const DevQueues& q, uint8_t* debug_out) {
  const dim3 grid((unsigned)(cfg.n_cu * cfg.trace_blocks_per_cu));
  const size_t lds = trace_lds_bytes(cfg, sc, false);
  if (debug_out) hipLaunchKernelGGL(k_trace_any<true>, grid, dim3(TRACE_BLOCK), lds, s, sc, q, cfg.stack_lds, debug_out);
  else hipLaunchKernelGGL(k_trace_any<false>, grid, dim3(TRACE_BLOCK), lds, s, sc, q, cfg.stack_lds, debug_out);
}

void pt_launch_shade(hipStream_t s, const LaunchCfg& cfg, const DevScene* sc, const DevFrame& fr, const DevQueues& q, int qi, uint32_t bounce) {
  const dim3 grid((q.n_seg + SHADE_WAVES - 1u) / SHADE_WAVES);      // one wave per segment
  if (cfg.shade_sort) hipLaunchKernelGGL((k_shade<true, false>), grid, dim3(SHADE_BLOCK), 0, s, sc, fr, q, qi, bounce);
  else if (cfg.shade_tables_lds) hipLaunchKernelGGL((k_shade<false, true>), grid, dim3(SHADE_BLOCK), 0, s, sc, fr, q, qi, bounce);
  else hipLaunchKernelGGL((k_shade<false, false>), grid, dim3(SHADE_BLOCK), 0, s, sc, fr, q, qi, bounce);
}
// the emitter, material and environment-row tables of this scene all fit k_shade's LDS copies
bool pt_shade_tables_fit(const DevScene& sc) {
  return sc.n_lights <= SHADE_LDS_LIGHTS && sc.n_mats <= SHADE_LDS_MATS && (!sc.env_ok || sc.env_h <= (int)SHADE_LDS_ENV_ROWS);
}
void pt_launch_accumulate(hipStream_t s, const DevFrame& fr, const DevQueues& q, float4* accum, uint32_t n_samples) {
  hipLaunchKernelGGL(k_accumulate, dim3((fr.n_owned + 255u) / 256u), dim3(256), 0, s, fr, (const float4*)q.lpath, accum, n_samples);
}
void pt_launch_shade_raster(hipStream_t s, const DevScene& sc, const DevCamera& cam, const DevFrame& fr, const DevQueues& q, float4* accum, bool gbuffer16) {
  if (gbuffer16) hipLaunchKernelGGL(k_shade_raster<true>, dim3((fr.n_owned + 255u) / 256u), dim3(256), 0, s, sc, cam, fr, q, accum);
  else hipLaunchKernelGGL(k_shade_raster<false>, dim3((fr.n_owned + 255u) / 256u), dim3(256), 0, s, sc, cam, fr, q, accum);
}
void pt_launch_to_half(hipStream_t s, const float4* radiance, uint2* out, uint32_t n_pixels) {
  hipLaunchKernelGGL(k_to_half, dim3((n_pixels + 255u) / 256u), dim3(256), 0, s, radiance, out, n_pixels);
}
void pt_launch_resolve(hipStream_t s, const DevFrame& fr, const float4* accum, float4* radiance, float spp, bool raster) {
  hipLaunchKernelGGL(k_resolve, dim3((fr.n_owned + 255u) / 256u), dim3(256), 0, s, fr, accum, radiance, spp, raster ? 1 : 0);
}
void pt_launch_tonemap(hipStream_t s, const float4* radiance, uint32_t* rgba8, int w, int h) {
  hipLaunchKernelGGL(k_tonemap, dim3((unsigned)((w + 15) / 16), (unsigned)((h + 15) / 16)), dim3(256), 0, s, radiance, rgba8, w, h);
}
