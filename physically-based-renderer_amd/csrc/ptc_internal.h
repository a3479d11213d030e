// ptc_internal.h — structures shared by the host side (ptc_api.cpp, ptc_scene.cpp) and the
// kernels (pt_kernels.hip).  Nothing here crosses the C-ABI (include/ptc.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

// ---- HBM layout of the committed scene (DESIGN.md §"Data layout in HBM") -----------------------
struct DevScene {
  const float4* recs;         // the BVH: one array of 16-byte units; 64-byte 8-wide nodes (origin on a 16-bit scene grid, exponents, slot masks,
                              // children-block address, 8-bit child planes) and 48-byte triangle records (v0,prim) (e1,class) (e2,-) in the
                              // children blocks of their nodes; addresses are unit indices, the root is at 0 (layout: ptc_scene.cpp)
  float grid_lo[3], grid_step[3];   // node origin = fma(oq, grid_step, grid_lo)
  const float4* shade;        // shade_stride × float4 per original primitive id.  The first 5 (80 B): the three R1 vertex records de-indexed to what
                              // shading reads: (Pa,mat) (Pb,light) (Pc,Na.x) (Na.yz,Nb.xy) (Nb.z,Nc.xyz).  Scenes with a textured material append the 6
                              // units of uv ×3, world tangent ×3, world bitangent ×3 and one of padding: a 192-byte record on a 64-byte boundary, three
                              // whole sectors per gather instead of the four to five of an 80-byte and a 96-byte record in two tables
  uint32_t shade_stride;      // 5 or 12
  const float4* mats;         // 4 × float4 per material: (base.rgb, metallic) (emissive.rgb, roughness) (base.a, tex_color, tex_normal, tex_mr) (texture set, -, -, -)
  const uint32_t* texels;     // all RGBA8 textures back to back
  const int4* tex_info;       // per texture: (offset into texels, width, height, 0)
  const uint4* set_texels;    // texture SETS: the (colour, normal, metal-rough) textures of a material interleaved texel by texel — one 16-byte gather
                              // instead of three 4-byte ones from three lines — and stored in 8x8 tiles, Morton order inside a tile (a 128-byte line = a
                              // 4x2 block of texels); same texel values as `texels`, so NEAREST and LINEAR results are untouched
  const int4* set_info;       // per set: (offset into set_texels or -1 when the set's textures differ in size, width, height, tiles per row)
  int tex_linear;             // 0 = NEAREST (the reference's sampler), 1 = bilinear
  const float4* env;          // lat-long environment: (radiance.rgb, texel pmf) per texel, row 0 = +y
  const float* env_marg;      // row cdf (env_h)
  const float* env_cond;      // per-row column cdf (env_w × env_h)
  const uint16_t* env_marg_guide;   // [PTC_ENV_GUIDE + 1]: guide[b] = cdf_search(env_marg, r = b / PTC_ENV_GUIDE): the search for r runs inside [guide[b], guide[b+1]],
                                    // b = floor(PTC_ENV_GUIDE r) — same index as the search over the whole table, after ~1 instead of ~10 dependent loads
  const uint16_t* env_cond_guide;   // [env_h][PTC_ENV_GUIDE + 1]: the same per row
  int env_w, env_h, env_ok;   // env_ok: the map has non-zero power, i.e. it can be importance-sampled
  const float4* lights;       // 5 × float4 per emitter: (v0,area) (e1,pmf) (e2,-) (ng,-) (Le,-)
  const float* cdf;           // emitter power cdf
  uint32_t n_lights;
  uint32_t n_mats;
  uint32_t n_lds_units;       // leading units (the top of the tree, breadth-first) that the trace kernels stage in LDS
  uint2* stack_ovf;           // per-lane traversal-stack overflow: [wave][ovf_depth][64] entries of (base_child, hits<<8 | imask)
  uint32_t ovf_depth;
  float ray_eps;
};

struct DevCamera { float pos[3], f[3], s[3], u[3], sx, sy; };  // R5 basis, see ptc_scene.cpp

struct DevFrame {
  int w, h;
  uint32_t seed_hash;          // pcg(seed_lo + pcg(seed_hi))
  int max_bounces;
  uint32_t n_owned;            // pixels this context owns
  const uint32_t* owned;       // their indices y*w+x, tile-Morton order
};

// ---- wavefront queues (SoA of 16-byte lanes) ------------------------------------------------------
// ray record   : A=(o.xyz,d.x) B=(d.y,d.z,T.x,T.y) C=(T.z,prev_pdf,path_id,key)   (48 B; the bounce index is the same for a whole launch and travels as a kernel argument)
//   raster/debug rays reuse B.zw as (tmin,tmax)
// hit record   : H=(t,prim,u,v)                                                            (16 B)
// shadow record: A=(o.xyz,d.x) B=(d.y,d.z,tmax,path_id) C=(contrib.rgb,-)                  (48 B)
//
// SEGMENTED queues (round 3).  A queue of a batch is cut into `n_seg` segments of `seg_len` slots (a multiple of 64): segment s owns
// the slots [s*seg_len, (s+1)*seg_len) of every queue array and holds seg_*[s] live records packed at its front.  k_raygen fills the
// segments back to back (slot = path id), and ONE WAVE of k_shade owns one segment: it reads the segment's rays and writes the
// continuation rays and the shadow rays it produces to the front of the SAME segment of the output arrays, so output compaction needs
// no atomic and no block barrier (a wave never produces more records than it consumes).  The trace kernels balance their load over
// chunks of <= 512 rays that never straddle a segment: k_scan turns the per-segment counts into a prefix sum of chunks per segment
// (pre_*), and a trace wave maps a chunk number to (segment, offset) by a 64-ary search of that prefix.
struct RayQ { float4* A; float4* B; float4* C; };   // 48 B per ray; the bounce index is a launch constant, not a field
struct ShadowQ { float4* A; float4* B; float4* C; };

struct DevQueues {
  RayQ ray[2];     // ping-pong: bounce b reads ray[b&1], shade writes ray[(b+1)&1]
  ShadowQ shadow;
  float4* hit;     // hit record of ray slot i, written in place by k_trace_closest: (t, prim | class<<28, u, v)
  float4* lpath;   // per-path radiance (rgb,-), single owner
  uint32_t* cnt;   // header words and the two work counters of the trace kernels, see CNT_*
  uint32_t* seg_ray[2];   // live rays per segment of ray[k]
  uint32_t* seg_sh;       // live shadow rays per segment
  uint32_t* pre_ray;      // [n_seg + 1] exclusive prefix of chunks per segment of the CURRENT ray queue
  uint32_t* pre_sh;       // the same for the shadow queue
  unsigned long long* stats;  // device statistics, see ST_*
  uint32_t cap;    // queue capacity in paths (the arrays hold ptc_seg_slots(cap) slots)
  uint32_t n_seg, seg_len;   // segment layout of the batch in flight (host-chosen per batch: ptc_seg_layout)
};

// Header words in `cnt` (written by k_set_counts / k_scan, read by the trace kernels) and the two work counters, each of which takes
// atomics and therefore has a 128-byte line of its own (atomics on one line are served one at a time, ~11 ns each).
enum { CNT_RAY_TOTAL = 0, CNT_RAY_CHUNK, CNT_RAY_NCHUNKS, CNT_SH_TOTAL, CNT_SH_CHUNK, CNT_SH_NCHUNKS, CNT_WORK_TRACE = 32, CNT_WORK_SHADOW = 64, CNT_N = 96 };
#define PTC_MAX_SEGMENTS 16384u     // upper bound of n_seg (sizes the per-segment arrays)
#define PTC_ENV_GUIDE 256u          // buckets of the environment cdf guide tables (a power of two: r * PTC_ENV_GUIDE is exact)
#define PTC_MATERIAL_CLASSES 8      // classes a hit word can carry (3 bits above the 28-bit primitive id); class 7 = miss under an environment
// Every statistic word has a 128-byte line of its own (index i lies at stats[i * ST_STRIDE]): atomics on ONE line are served one at a time (~11 ns each), and until
// round 4 the trace kernels ended with four per WAVE on one line — 16-32 k atomics, 0.2-0.35 ms at the end of EVERY launch, which was most of a small launch
// (tools/viewer_loop.py).  Now: one per counter and BLOCK, four lines.
#define ST_STRIDE 16
enum { ST_SEGMENTS = 0, ST_SHADOW, ST_HITS, ST_NODES_C, ST_TRIS_C, ST_NODES_A, ST_TRIS_A,
       // wave-level iteration counts of the trace kernels' loops (filled only by a -DPT_DIAG build): lane
       // utilisation of a phase = lane-level count / (64 x wave-level count)
       ST_DIAG_NODE_ITERS, ST_DIAG_TRI_ITERS, ST_DIAG_LEAF_VISITS, ST_DIAG_ROUNDS, ST_DIAG_REFILLED, ST_N };

// Segment layout of a batch of n slots over at most max_seg segments: n_seg = min(max_seg, ceil(n / PTC_SEG_MIN_LEN)) (at least 1), seg_len =
// ceil(n / n_seg) rounded up to a multiple of 64.  n_seg * seg_len <= ptc_seg_slots(n, max_seg).
// A segment is at least one full trace chunk long (512 slots): with the 64 of round 3a a 1080p x 1 spp frame (2 M paths) was cut into 16384
// segments of 128 slots, hence >= 16384 chunks of a few rays each at the later bounces — every refill of a trace wave then cost an atomic and a
// search of the chunk prefix for a handful of rays, and EVERY trace launch of such a frame took 0.5-0.8 ms however few rays it had (11.5 ms per
// frame, tools/viewer_loop.py); batches of the benchmark's size (535 M slots) are at max_seg either way.
#define PTC_SEG_MIN_LEN 512u
inline void ptc_seg_layout(uint32_t n, uint32_t max_seg, uint32_t& n_seg, uint32_t& seg_len) {
  uint32_t s = (n + PTC_SEG_MIN_LEN - 1u) / PTC_SEG_MIN_LEN;
  if (s > max_seg) s = max_seg;
  if (s < 1u) s = 1u;
  const uint32_t per = (n + s - 1u) / s;
  n_seg = s; seg_len = per ? ((per + 63u) & ~63u) : 64u;
}
inline size_t ptc_seg_slots(uint32_t cap, uint32_t max_seg) { return (size_t)cap + 64u * (size_t)max_seg + 64u; }

struct LaunchCfg { int n_cu; int trace_blocks_per_cu; int stack_lds; /* stack entries kept in LDS per lane */ int shade_waves; /* waves of k_shade's grid = upper bound of n_seg */
                   int shade_sort; /* 1: k_shade sorts its slots by material class before shading (PTC_SHADE_SORT=1); 0: it takes them in queue order */
                   int shade_tables_lds; /* 1: the scene's emitter / material / environment-row tables all fit k_shade's LDS copies (pt_shade_tables_fit): the variant without global fallbacks runs */ };

// ---- kernel launchers (pt_kernels.hip) ------------------------------------------------------------
const char* pt_kernel_policy();  // compile-time constants of the kernels as "name=value ..." (ptc_launch_policy)
int pt_trace_block_threads();   // threads per block of the trace kernels (compile-time constant of pt_kernels.hip)
size_t pt_trace_lds_bytes(const LaunchCfg&, const DevScene&);   // dynamic LDS of a trace block (the larger, closest-hit, figure)
int pt_trace_blocks_per_cu(size_t lds_bytes);                    // resident trace blocks per CU at that LDS size (runtime occupancy query)
void pt_launch_set_counts(hipStream_t, const LaunchCfg&, const DevQueues&, uint32_t n_rays, uint32_t n_shadow);   // identity layout: ray i at slot i
void pt_launch_scan(hipStream_t, const LaunchCfg&, const DevQueues&, int qi_next);   // after k_shade: chunk prefixes of the rays it wrote to ray[qi_next] and of its shadow rays
int pt_shade_block_threads();
bool pt_shade_tables_fit(const DevScene&);
void pt_launch_raygen(hipStream_t, const DevCamera&, const DevFrame&, const DevQueues&, uint32_t first_sample, uint32_t n_samples, bool raster);
void pt_launch_trace_closest(hipStream_t, const LaunchCfg&, const DevScene&, const DevQueues&, int qi, bool cull);
void pt_launch_shade(hipStream_t, const LaunchCfg&, const DevScene* scene_on_device, const DevFrame&, const DevQueues&, int qi, uint32_t bounce);   // every ray of a wavefront launch is at the same bounce
void pt_launch_trace_any(hipStream_t, const LaunchCfg&, const DevScene&, const DevQueues&, uint8_t* debug_out /*or null*/);
void pt_launch_accumulate(hipStream_t, const DevFrame&, const DevQueues&, float4* accum, uint32_t n_samples);
void pt_launch_shade_raster(hipStream_t, const DevScene&, const DevCamera&, const DevFrame&, const DevQueues&, float4* accum, bool gbuffer16);
void pt_launch_to_half(hipStream_t, const float4* radiance, uint2* out_rgba16f, uint32_t n_pixels);
void pt_launch_resolve(hipStream_t, const DevFrame&, const float4* accum, float4* radiance, float inv_spp_divisor, bool raster);
void pt_launch_tonemap(hipStream_t, const float4* radiance, uint32_t* rgba8, int w, int h);

// ---- host-side scene build (ptc_scene.cpp) ----------------------------------------------------------
struct HostMaterial { float base[4]; float metallic, roughness; float emissive[3]; int tex_color, tex_normal, tex_mr; };
struct HostVertex { float position[3], normal[3], tangent[4], texcoord[2]; };
struct HostTexture { std::vector<uint8_t> px; int w, h; };
struct HostEnv { std::vector<float> rgb; int w = 0, h = 0; };
struct HostMesh { std::vector<HostVertex> v; std::vector<uint32_t> idx; int material; };
struct HostInstance { int mesh; float m[16]; };   // column-major model matrix

struct HostBuilt {
  uint32_t n_wverts = 0;         // world vertices of the scene (wverts.size() once the array is on the host: a commit or a tree made on the device leaves it unsized until somebody asks)
  std::vector<HostVertex> wverts;
  std::vector<uint32_t> widx;
  std::vector<int32_t> tri_mat;
  std::vector<int32_t> prim_light;
  std::vector<float> shade;      // 4 * shade_stride floats per primitive (see DevScene::shade)
  uint32_t shade_stride = 5;
  std::vector<uint32_t> texels;  // RGBA8 texels of all textures
  std::vector<int32_t> tex_info; // 4 ints per texture
  std::vector<uint32_t> set_texels;   // 4 words per texel of every texture set (see DevScene::set_texels)
  std::vector<int32_t> set_info;      // 4 ints per set
  std::vector<uint16_t> env_marg_guide, env_cond_guide;
  std::vector<float> env, env_marg, env_cond;   // 4 floats per texel; cdfs
  int env_w = 0, env_h = 0, env_ok = 0;
  std::vector<float> recs;       // the BVH as 16-byte units (4 floats each): nodes + triangle records (see DevScene::recs)
  float grid_lo[3] = {0, 0, 0}, grid_step[3] = {1, 1, 1};
  std::vector<float> mats;       // 16 floats per material
  std::vector<float> lights;     // 20 floats per emitter
  std::vector<float> cdf;
  uint32_t n_nodes = 0, n_tris = 0, n_tri_records = 0, n_lights = 0, max_depth = 0, n_units = 0, n_lds_units = 0;
  float ray_eps = 0.0f;
  uint64_t sa_cost_fixed = 0;    // surface-area cost of the 8-wide tree in units of 2^-20 (ptc_stats.bvh_sa_cost; the device refit sums the same terms: pt_refit.hip)
  float sa_unit = 0.0f;          // its unit: the half area of the scene box WHEN THE TOPOLOGY WAS BUILT (a refit keeps it, so that the cost of a moved scene is in the
                                 // units of the commit's and the two can be compared; a scene box that grows with the motion would otherwise hide the growth)
  std::shared_ptr<void> topology;   // what ptc_refit_scene keeps of the build (ptc_scene.cpp: Topology)
};

// returns empty string on success, else the error text
std::string ptc_build_scene(const std::vector<HostMaterial>&, const std::vector<HostMesh>&, const std::vector<HostInstance>&,
                            const std::vector<HostTexture>&, const HostEnv&, uint32_t toplet_budget, int bvh_builder /* PTC_BVH_* */,
                            HostBuilt& out);
// a commit whose flatten, shading records and tree are made on the device: the host's share (ptc_scene.cpp)
std::string ptc_build_skeleton(const std::vector<HostMaterial>&, const std::vector<HostMesh>&, const std::vector<HostInstance>&,
                               const std::vector<HostTexture>&, const HostEnv&, uint32_t toplet_budget, HostBuilt& out);
// the instances' matrices changed since ptc_build_scene filled `out`: same tree, new boxes / records (ptc_scene.cpp)
std::string ptc_refit_scene(const std::vector<HostMaterial>&, const std::vector<HostMesh>&, const std::vector<HostInstance>&,
                            const std::vector<HostTexture>&, const HostEnv&, HostBuilt& out);
// material class of every primitive (word 7 of its triangle record; ptc_scene.cpp, "texture sets and material classes")
void ptc_prim_classes(const std::vector<HostMaterial>&, const std::vector<int32_t>& tri_mat, std::vector<uint32_t>& out);
void ptc_trs_to_matrix(const float t[3], const float q_wxyz[4], const float s[3], float m16[16]);
void ptc_make_camera(const float pos[3], const float target[3], float fov, float aspect, DevCamera& cam);
// pixels owned by (rank,count) in tile-Morton order
void ptc_owned_pixels(int w, int h, int tile_rank, int tile_count, std::vector<uint32_t>& out);
