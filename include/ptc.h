/* ptc.h — C-ABI boundary of the MI355X path-tracing core ("ptc").
 *
 * This is the drop-in boundary of SURVEY.md §8(b).  The reference
 * (WeaponizedSchizophrenia/physically-based-renderer) has no FFI/plugin
 * interface for its render path; the seam is the C++ member call
 *
 *   pbr::PbrRenderSystem::render(cmd, scene, gBuffer, renderTarget, extent)
 *       src/pbr_engine/engine/pbr/PbrRenderSystem.hpp:46-47
 *   called from app::App::recordCommands  src/gltf_viewer/App.cpp:387-388
 *
 * fed by  gltf::Loader::loadAsset / Asset::loadScene
 *       src/pbr_engine/gltf/pbr/gltf/Loader.hpp:20-21, Asset.hpp:76-78
 * and consumed by TonemapperSystem::run
 *       src/pbr_engine/engine/pbr/TonemapperSystem.cpp:97-134.
 *
 * Each entry point below names the reference interface it replaces.  Plain C:
 * opaque handle, plain pointers and sizes, int status (0 = ok, <0 = error,
 * text via ptc_last_error).  The caller owns every input array (copied during
 * the call); the library owns all device memory.  A context is bound to one
 * HIP device and is not re-entrant; distinct contexts are independent.
 *
 * The library behind this header is HIP-only.  There is no CPU fallback:
 * ptc_create fails (returns NULL, ptc_last_error(NULL) says why) when no
 * gfx950 device is usable.
 */
#ifndef PTC_H
#define PTC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTC_ABI_VERSION 4

typedef struct ptc_ctx ptc_ctx;

/* status codes */
enum {
  PTC_OK = 0,
  PTC_E_ARG = -1,      /* bad argument (null pointer, index out of range, bad size) */
  PTC_E_STATE = -2,    /* call out of order (e.g. render before scene_commit)       */
  PTC_E_DEVICE = -3,   /* HIP / RCCL error (text carries hipGetErrorString)         */
  PTC_E_NOMEM = -4     /* hipErrorOutOfMemory: queues or scene do not fit the device */
};

/* integrators */
enum {
  PTC_INTEGRATOR_PATH = 0,          /* wavefront path tracer (SURVEY §8a-2 P1–P10)                  */
  PTC_INTEGRATOR_RASTER_COMPAT = 1, /* primary hit + the reference's Blinn-Phong pass (R6,R7,R8):
                                       assets/shaders/pbr/lighting.glsl:19-29, lit from fp32 P, N, albedo */
  PTC_INTEGRATOR_RASTER_GBUFFER16 = 2 /* the same pass lit from what the reference's G-buffer holds
                                       (engine/pbr/GBuffer.hpp:13-16): positions and normals rounded to
                                       RGBA16F (round to nearest even), albedo to RGBA16 UNORM, the normal not
                                       re-normalised; read the result with ptc_read_radiance_rgba16f for the
                                       RGBA16F lighting target (PbrRenderSystem.hpp:21)              */
};

/* Vertex record == pbr::MeshVertex (src/pbr_engine/engine/pbr/MeshVertex.hpp:14-19):
 * {vec3 position; vec3 normal; vec4 tangent; vec2 texCoords}, 48 bytes, tightly packed. */
typedef struct ptc_vertex {
  float position[3];
  float normal[3];
  float tangent[4];
  float texcoord[2];
} ptc_vertex;

/* Counters of the last frame ("counted, not estimated"). */
typedef struct ptc_stats {
  uint64_t paths;             /* camera samples traced                                   */
  uint64_t segments;          /* closest-hit rays cast (camera + continuation)           */
  uint64_t shadow_rays;       /* any-hit (NEE) rays cast                                 */
  uint64_t hits;              /* closest-hit rays that hit a surface                     */
  uint64_t node_visits_closest;
  uint64_t tri_tests_closest;
  uint64_t node_visits_any;
  uint64_t tri_tests_any;
  uint64_t algorithmic_bytes; /* SURVEY §8d byte formula evaluated on the counters above */
  double seconds_render;      /* device time of the frame's kernels (HIP events)         */
  double seconds_trace_closest; /* HIP-event time of the dominant kernel (sum over launches).  The three per-kernel sums cover the batches whose
                                 * kernels run one after the other; a small batch (<= 2^26 paths) runs k_trace_any(b) beside k_trace_closest(b + 1) and
                                 * carries the batch's span only (seconds_render) — PTC_TIMING=2 records per-kernel spans for those too (they then
                                 * include each other), PTC_TIMING=0 records nothing */
  double seconds_trace_any;
  double seconds_shade;
  double seconds_commit;      /* flatten + BVH build + upload                               */
  double seconds_reduce;      /* HIP-event time of ptc_comm_reduce_radiance on this rank's stream (includes waiting for the slowest rank) */
  double seconds_refit;       /* the last ptc_scene_refit: re-flatten + refit of the committed tree + upload                */
  uint32_t launches_trace_closest;
  uint32_t launches_trace_any;
  uint32_t n_triangles;
  uint32_t n_bvh_nodes;
  uint32_t n_emitters;
  uint32_t bvh_max_depth;
  /* ABI 4 (round 4): what a moved scene's tree costs, so that a caller can decide when a refit is no longer enough.
   * bvh_sa_cost = the surface-area cost of the 8-wide tree as it lies in HBM: sum over the nodes' child slots of half_area(child box) /
   * half_area(scene box when the tree's topology was made), a two-triangle leaf counted twice — the expected number of node visits + triangle
   * tests of a random long ray, up to a constant; the unit stays through refits, so the figures of a moving scene can be compared.  Written by ptc_scene_commit, by every ptc_scene_refit and ptc_scene_rebuild (one reduction inside the node pass, in
   * fixed point: the same bits whatever the order).  bvh_sa_cost_built = its value when the tree's TOPOLOGY was made (commit or rebuild): the ratio
   * of the two is what examples/viewer_shim.cpp watches. */
  double bvh_sa_cost;
  double bvh_sa_cost_built;
  double seconds_rebuild;     /* the last ptc_scene_rebuild: flatten + LBVH build + refit pass, all on the device                */
} ptc_stats;

/* ---- context ------------------------------------------------------------------------------
 * Replaces core::makeGpuHandle + MemoryAllocator bring-up
 * (src/pbr_engine/core/pbr/core/GpuHandle.cpp:94-101, engine/pbr/memory/MemoryAllocator.cpp:68-88).
 * device_id: HIP ordinal.  Returns NULL on failure.
 * PTC_DEVICE_NONE gives a description-only context: the scene calls (begin/add/commit: flatten +
 * BVH build on the host) and the ptc_debug_get_* hooks work, every call that needs the GPU fails with
 * PTC_E_DEVICE.  It exists so that the host logic can be checked without a GPU; it renders nothing. */
#define PTC_DEVICE_NONE (-1)
ptc_ctx* ptc_create(int device_id);
void ptc_destroy(ptc_ctx*);
/* Last error text of the context (or of the last failed ptc_create when ctx == NULL). */
const char* ptc_last_error(const ptc_ctx*);
int ptc_abi_version(void);
/* "ptc abi N gfx950 kernels-sha256 <64 hex digits>": the hash is over the kernel sources this library was built from
 * (csrc/Makefile); bench.py compares it with the hash in the committed kernel model its roofline block is calibrated on. */
const char* ptc_build_info(void);
/* "name=value ..." of everything that decides how the kernels are launched: their compile-time constants (block sizes, chunk, ring, thresholds), the
 * context's knobs after the environment was read (lanes, batch size, overlap mode, nodelets, builder ...) and, once a scene is committed on a device, what
 * followed from them (trace blocks per CU, stack entries in LDS, queue segments).  ctx == NULL: the built-in defaults, no device needed.  A measured
 * per-kernel figure is only valid for the policy it was measured under: tools/make_kernel_model.py records this string with the profile, bench.py prints
 * it and reports model_stale when it differs, tests/test_profiles.py fails when the defaults move without a new profile.  The pointer is valid until the
 * calling thread's next call. */
const char* ptc_launch_policy(const ptc_ctx*);

/* ---- scene description --------------------------------------------------------------------
 * Replaces gltf::Asset::loadScene → MeshBuilder::build → TransferStager
 * (src/pbr_engine/gltf/pbr/gltf/Asset.cpp:135-273, engine/pbr/MeshBuilder.cpp:16-55,
 *  engine/pbr/TransferStager.cpp:51-177). */
int ptc_scene_begin(ptc_ctx*);

/* pbr::MaterialData{vec4 color} (engine/pbr/Material.hpp:14-16) widened with the glTF
 * metal-rough + emissive factors the reference ignores (gltf/Asset.cpp:142-150).
 * tex_* are texture ids from ptc_add_texture_rgba8 or -1.  Returns material id >= 0. */
int ptc_add_material(ptc_ctx*, const float base_color[4], float metallic, float roughness,
                     const float emissive[3], int tex_color, int tex_normal, int tex_mr);

/* image::loadImage2D output: RGBA8, 4 channels forced (image/pbr/image/LoadImage.cpp:56-73);
 * sampled NEAREST/REPEAT like the reference's default sampler (gltf/Asset.cpp:116-117).
 * Returns texture id >= 0. */
int ptc_add_texture_rgba8(ptc_ctx*, const uint8_t* px, int w, int h);

/* One MeshBuilder::Primitive (engine/pbr/MeshBuilder.hpp:14-18) with indices widened to u32
 * (reference: u16, Asset.cpp:197-201).  Triangle list, indices primitive-local.
 * Returns mesh id >= 0. */
int ptc_add_mesh(ptc_ctx*, const ptc_vertex* verts, uint32_t n_verts, const uint32_t* indices,
                 uint32_t n_indices, int material);

/* pbr::Transform + makeModelPushConstant (engine/pbr/Scene.hpp:19-23,
 * ModelPushConstant.hpp:33-46): model = T·R·S, quaternion order (w,x,y,z). */
int ptc_add_instance(ptc_ctx*, int mesh, const float t[3], const float q_wxyz[4],
                     const float s[3]);
/* makeModelPushConstant(glm::mat4x4 model) (ModelPushConstant.hpp:33-38): an instance by its column-major
 * 4x4 model matrix (used by the glTF loader, which composes parent transforms). */
int ptc_add_instance_matrix(ptc_ctx*, int mesh, const float model[16]);

/* Scene dynamics.  The viewer turns its nodes every frame (src/gltf_viewer/App.cpp:306-313) and draws each node with its own
 * model matrix (the push constant of PbrRenderSystem.cpp:444-448), so a transform change costs the reference nothing.  Here the
 * triangles live in one world-space tree, so a change is two steps: ptc_update_instance / ptc_update_instance_matrix give instance
 * `instance` (the value ptc_add_instance* returned) a new transform, any number of them; ptc_scene_refit then re-flattens the
 * vertices and REFITS the committed tree — same topology, same slots, same layout; every box re-computed bottom-up and re-quantised,
 * triangle records, shading records and emitters rewritten — in place in HBM (textures, environment and materials are not
 * touched).  The refit runs ON THE DEVICE (csrc/pt_refit.hip: the same arithmetic as the host's, element-parallel; 84 bytes per
 * instance and the emitter table are all that crosses the bus); PTC_REFIT=host in the environment, a description-only context, or
 * a move that changes which triangles are emitters (a zero-area scale) take the host refit + upload instead — both give the same
 * bytes (ptc_debug_get_bvh / ptc_debug_get_shading_tables).  No re-build either way (ptc_stats.seconds_refit beside seconds_commit).  A refitted tree renders the same image as a fresh commit of the same transforms
 * (closest hit = minimum of (t, primitive id), whatever the tree); its traversal counters are those of the refitted tree, and the
 * oracle refits the same way.  Meshes, materials or the number of instances cannot change this way: that is a new scene.
 * A frame in progress ends (call ptc_frame_begin again).  PTC_E_STATE before the first ptc_scene_commit. */
int ptc_update_instance(ptc_ctx*, int instance, const float t[3], const float q_wxyz[4], const float s[3]);
int ptc_update_instance_matrix(ptc_ctx*, int instance, const float model[16]);
int ptc_scene_refit(ptc_ctx*);
/* A refit keeps the tree of the commit: topology and octant slots follow the geometry they were made for, and the viewer turns its nodes without
 * bound (App.cpp:306-313: rotate(rotation, deltaTime) every frame) — after a third of a turn the atrium's rays visit 1.5x the nodes
 * (profiles/r04_refit_curve.txt).  ptc_scene_rebuild applies the pending instance transforms like ptc_scene_refit and then builds a NEW tree for the
 * geometry as it now lies in HBM, ON THE DEVICE (csrc/pt_build.hip): 63-bit Morton codes, LDS radix sort, the radix tree, bottom-up boxes and collapse
 * costs, the same cost-optimal 8-wide collapse, octant slots, quantisation and unit layout as the host's LBVH builder (PTC_BVH_LBVH) — the bytes a
 * fresh ptc_scene_commit of the moved scene with that builder would upload (tests/test_gpu_parity.py compares them), in milliseconds and without
 * the description crossing the bus again.  The image does not depend on the tree; counters and speed are those of the new tree.  Whatever builder
 * the commit used, the rebuilt tree is the LBVH.  Needs a device; PTC_E_STATE before the first commit; a move that changes which triangles are
 * emitters falls back to a host build + upload (as ptc_scene_refit does).  ptc_stats.bvh_sa_cost / bvh_sa_cost_built say when it is worth calling. */
int ptc_scene_rebuild(ptc_ctx*);

/* pbr::makeCameraData (engine/pbr/CameraData.hpp:22-32): lookAtRH(pos,target,up=(0,-1,0)),
 * perspective fovY/aspect; y-down un-flipped viewport (PbrRenderSystem.cpp:425-430). */
int ptc_set_camera(ptc_ctx*, const float pos[3], const float target[3], float fov_y,
                   float aspect);

/* Lat-long environment light (BASELINE config 5; no counterpart in the reference, which has no lights): w*h RGB
 * fp32 texels, row 0 = +y, u = atan2(d.z, d.x)/(2 pi) + 1/2, piecewise-constant radiance, importance-sampled by
 * luminance x sin(theta).  rgb == NULL removes it.  Call before ptc_scene_commit. */
int ptc_set_env_latlong_rgb32f(ptc_ctx*, const float* rgb, int w, int h);

/* Texture filter of the scene being described (applies to every texture; reset to NEAREST by ptc_scene_begin).
 * PTC_FILTER_NEAREST is what the reference renders with — its samplers are default-constructed (gltf/Asset.cpp:116-117:
 * NEAREST, REPEAT, no mips).  PTC_FILTER_LINEAR is an option the reference does not have: bilinear over the 4 nearest
 * texels (centres at i + 1/2), REPEAT wrap, lerp(a, b, t) = fma(t, b - a, a) along x then y. */
enum { PTC_FILTER_NEAREST = 0, PTC_FILTER_LINEAR = 1 };
int ptc_set_texture_filter(ptc_ctx*, int filter);

/* BVH builder of the scene being described (reset by ptc_scene_begin to the context's default: PTC_BVH_SAH, or PTC_BVH_LBVH when
 * the environment has PTC_BVH=lbvh).  Both give a binary tree over single triangles that the same cost-optimal collapse turns into
 * the 8-wide quantised tree; images are identical up to the order-independence of closest hit, traversal counters differ.
 *   PTC_BVH_SAH   top-down binned surface-area splits (32 bins): the default; on the benchmark scene 17 % fewer node visits per
 *                 closest-hit ray and 28 % fewer per shadow ray than the LBVH
 *   PTC_BVH_LBVH  the radix tree of 63-bit Morton codes of the triangle-box centres (BASELINE.json north_star's "flattened LBVH") */
enum { PTC_BVH_SAH = 0, PTC_BVH_LBVH = 1 };
int ptc_set_bvh_builder(ptc_ctx*, int builder);

/* Flatten instances to world space (geometry_pass/vertex.glsl:25-36), build + flatten the BVH,
 * build the emitter CDF, upload everything to HBM.  With PTC_BVH_SAH (the default) flatten and build run on the host's
 * thread pool (75 ms at 250 k triangles).  With PTC_BVH_LBVH on a device context the host only describes (indices,
 * materials, emitter table, textures) and the DEVICE flattens the vertices, writes the shading records and builds the
 * tree (csrc/pt_refit.hip, csrc/pt_build.hip): 3-5 ms at 250 k triangles, the arrays in HBM byte for byte those of the
 * host's LBVH commit (PTC_COMMIT=host in the environment keeps that path; ptc_debug_get_internals [7] bit 1 says which ran). */
int ptc_scene_commit(ptc_ctx*);

/* ---- rendering ----------------------------------------------------------------------------
 * Replaces PbrRenderSystem::render (engine/pbr/PbrRenderSystem.cpp:357-365).
 * ptc_render == frame_begin + frame_add_samples(spp) + frame_resolve. */
int ptc_render(ptc_ctx*, int w, int h, int spp, uint64_t seed, int max_bounces, int integrator);

/* Progressive form.  tile_rank/tile_count select the 32×32-pixel tiles this context owns
 * (tile t along a Morton walk belongs to rank t mod tile_count); 0/1 = whole frame.
 * spp_total is the sample BUDGET of the frame: frame_add_samples fails with PTC_E_ARG beyond it. */
int ptc_frame_begin(ptc_ctx*, int w, int h, int spp_total, uint64_t seed, int max_bounces,
                    int integrator, int tile_rank, int tile_count);
/* Accept the next n_samples samples of every owned pixel.  Asynchronous: full wavefront batches (as many samples as
 * fit the path budget, PTC_BATCH_PATHS, split over the lanes) are queued on the device at once; a remainder is held
 * back and merged with the samples of later calls, so that many small calls still produce full-width launches.
 * ptc_frame_resolve / ptc_sync / ptc_get_stats queue whatever is still held back.  Queues are sized by the batches
 * actually issued: adding one sample per call needs queues for one sample per pixel. */
int ptc_frame_add_samples(ptc_ctx*, int n_samples);
/* Optional: allocate the current frame's wavefront queues for FULL batches now (min(batch, spp_total) samples per owned pixel and
 * lane).  Without it the queues grow with the batches issued — right for a viewer that adds a sample per displayed frame, but a
 * growth step drains the device and reallocates; an offline render that will spend its budget calls this once after
 * ptc_frame_begin so that nothing is allocated while it renders.  PTC_E_NOMEM if the queues do not fit. */
int ptc_frame_reserve(ptc_ctx*);
/* sum / (samples accumulated so far) → full-frame RGBA32F (zeros in pixels this context does not own): after k of
 * N samples the buffer holds the k-sample image, correctly exposed (progressive display). */
int ptc_frame_resolve(ptc_ctx*);
/* Block until all queued device work of this context has finished. */
int ptc_sync(ptc_ctx*);

/* Output == the HdrImage the tonemapper consumes (engine/pbr/HdrImage.cpp:12-45), as fp32:
 * w*h*4 floats, row-major, y-down, alpha = 1 where owned. */
int ptc_read_radiance_rgba32f(ptc_ctx*, float* out);
/* Device pointer of that buffer (w*h*4 floats) for in-place RCCL reduction by the caller. */
void* ptc_radiance_device_ptr(ptc_ctx*);
/* Overwrite the radiance buffer from host (e.g. after a reduce) before tonemapping. */
int ptc_write_radiance_rgba32f(ptc_ctx*, const float* in);
/* The same image in the reference's own HdrImage format, vk::Format::eR16G16B16A16Sfloat
 * (engine/pbr/PbrRenderSystem.hpp:21, HdrImage.cpp:20): w*h*4 IEEE binary16 values (as uint16_t), fp32 → fp16 by
 * round-to-nearest-even, overflow → inf, half denormals kept.  This is what a viewer shim copies into the HdrImage
 * (INTEGRATION.md §2).  The device-pointer form converts on the context's stream, waits for the conversion and
 * returns a buffer that stays valid until the next call of either function (NULL on error). */
int ptc_read_radiance_rgba16f(ptc_ctx*, uint16_t* out);
void* ptc_radiance_rgba16f_device_ptr(ptc_ctx*);

/* TonemapperSystem::run + tonemappers/aces+gamma.glsl:10-40 on the radiance buffer → RGBA8. */
int ptc_tonemap_rgba8(ptc_ctx*, uint8_t* out);

/* ---- checkpoint / resume, sample ranges (SURVEY §5 "optional later", §8e "kept as an option").  The RNG is counter-based — sample k of pixel p draws from
 * hash(seed, p, k) — and a pixel's sum is taken in sample order, so a frame is resumable by its sums and a count.
 *   ptc_frame_checkpoint: everything queued is finished; the per-pixel sums (n_owned x RGBA fp32, in the order of the frame's owned pixels: an opaque blob for
 *     ptc_frame_restore; NULL to query the sizes only) and the number of samples in them.
 *   ptc_frame_restore: right after a ptc_frame_begin with the SAME parameters (size, seed, bounces, tile share; spp_total may be larger): the sums and the
 *     count are put back, the next sample added is sample `samples_done`.  checkpoint after k samples + restore + the remaining samples = the uninterrupted
 *     frame, bit for bit.
 *   ptc_frame_set_sample_range: right after ptc_frame_begin: the frame's samples have the indices first_sample, first_sample + 1, ...; with resolve_divisor != 0
 *     the resolve divides by it instead of by the samples accumulated.  Sharding a frame by SAMPLES instead of by tiles: rank r of N renders all pixels
 *     (tile_count 1), spp/N samples from first_sample = r x spp/N, resolve_divisor = spp: the reduce's sum of the partial means is the frame (fp32 sums in another
 *     order than on one GPU: equal to rounding, not bit for bit — which is why tiles are the default). */
int ptc_frame_checkpoint(ptc_ctx*, float* accum_rgba, uint64_t* n_owned_pixels, uint32_t* samples_done);
int ptc_frame_restore(ptc_ctx*, const float* accum_rgba, uint64_t n_owned_pixels, uint32_t samples_done);
int ptc_frame_set_sample_range(ptc_ctx*, uint32_t first_sample, uint32_t resolve_divisor);

int ptc_get_stats(ptc_ctx*, ptc_stats* out);

/* ---- multi-GPU: tiles shard over devices, one RCCL reduce brings the framebuffer to the root -----------
 * The reference has no multi-device path (one vk::Device, core/GpuHandle.cpp:94-101); this is BASELINE.json's
 * "independent pixel/sample tiles shard across the 8 GPUs of one node with an RCCL reduce onto rank 0".
 *
 * One process per GPU: rank 0 calls ptc_comm_unique_id and ships the 128 bytes to the other ranks by whatever channel
 * the host has (MPI, torch.distributed, a file); every rank then calls ptc_comm_init on its context (collective:
 * ncclCommInitRank), renders its tiles (ptc_frame_begin with tile_rank/tile_count) and, after ptc_frame_resolve,
 * ptc_comm_reduce_radiance(root): ncclReduce(fp32, sum) in place on the full-frame radiance buffer, queued on the
 * context's stream behind the resolve (asynchronous; ptc_sync or a read-back waits for it).  Tiles are disjoint and the
 * other pixels are zero, so the sum is x + 0: the root's image is bit-identical to the single-GPU image. */
#define PTC_COMM_ID_BYTES 128
int ptc_comm_unique_id(uint8_t out[PTC_COMM_ID_BYTES]);
int ptc_comm_init(ptc_ctx*, const uint8_t id[PTC_COMM_ID_BYTES], int rank, int n_ranks);
int ptc_comm_reduce_radiance(ptc_ctx*, int root);
int ptc_comm_destroy(ptc_ctx*);

/* One process driving several GPUs: n contexts (one per device id) + ncclCommInitAll.  Describe and commit the same
 * scene on every ptc_group_ctx(g, i); ptc_group_render then traces device i's tiles on device i (all devices are queued
 * before anything is waited for), reduces onto device 0 and syncs: ptc_group_ctx(g, 0) holds the whole frame
 * (ptc_read_radiance_* / ptc_tonemap_rgba8 on it).  ptc_group_create returns NULL on failure (ptc_group_last_error(NULL)). */
typedef struct ptc_group ptc_group;
ptc_group* ptc_group_create(const int* device_ids, int n_devices);   /* every id PTC_DEVICE_NONE: a description-only group (no GPU, no RCCL): the host half of the group calls */
int ptc_group_size(const ptc_group*);
/* Commit ONE scene to every device of the group: the scene is described on ptc_group_ctx(g, 0) only; this call flattens it and
 * builds the BVH once on the host (unless device 0 has committed it already: that build is then used) and uploads that one build to
 * every device (N contexts committing on their own would each repeat the build, one after the other on the calling thread). */
int ptc_group_scene_commit(ptc_group*);
/* Scene dynamics for a group: ptc_update_instance* on ptc_group_ctx(g, 0), then every device refits its copy in place (or, on the
 * host path, ONE refit on the host whose arrays go to every device). */
int ptc_group_scene_refit(ptc_group*);
ptc_ctx* ptc_group_ctx(ptc_group*, int i);
int ptc_group_render(ptc_group*, int w, int h, int spp, uint64_t seed, int max_bounces, int integrator);
const char* ptc_group_last_error(const ptc_group*);
void ptc_group_destroy(ptc_group*);

/* ---- test hooks (parity of single stages; not needed by a renderer) -------------------------
 * Closest-hit of n explicit rays through the same trace kernel: origins/dirs are n*3 floats;
 * out_t n floats (t, or -1 on miss), out_prim n int32 (original primitive id or -1),
 * out_uv n*2 floats. */
int ptc_debug_trace_closest(ptc_ctx*, const float* origins, const float* dirs, uint32_t n,
                            float* out_t, int32_t* out_prim, float* out_uv);
/* Any-hit of n explicit rays with tmax each; out_occluded n uint8. */
int ptc_debug_trace_any(ptc_ctx*, const float* origins, const float* dirs, const float* tmax,
                        uint32_t n, uint8_t* out_occluded);
/* World-space flattened geometry as committed: verts (n_verts*12 floats = ptc_vertex),
 * indices (n_tris*3 u32), per-triangle material.  Pass NULL to query sizes only. */
int ptc_debug_get_flat_scene(ptc_ctx*, uint32_t* n_verts, uint32_t* n_tris, ptc_vertex* verts,
                             uint32_t* indices, int32_t* tri_material);

/* The scene description as received (valid from ptc_scene_begin on, committed or not): number of materials and
 * textures; material i as 12 values (base rgba, metallic, roughness, emissive rgb as floats; tex_color, tex_normal,
 * tex_mr as ints); texture i's size and, when rgba is non-NULL, its w*h*4 bytes. */
int ptc_debug_get_description(ptc_ctx*, int* n_materials, int* n_textures);
int ptc_debug_get_material(ptc_ctx*, int index, float out_factors[9], int out_textures[3]);
int ptc_debug_get_texture(ptc_ctx*, int index, int* w, int* h, uint8_t* rgba);

/* Raw device counter array of the current frame (segments, shadow rays, hits, node/triangle counts, then the
 * loop-iteration diagnostics a -DPT_DIAG build fills).  Returns the number of counters the library keeps. */
int ptc_debug_get_counters(ptc_ctx*, uint64_t* out, int n);

/* The flattened 8-wide BVH as committed: ONE array of n_units 16-byte units (n_units*4 32-bit words) holding 64-byte nodes
 * (origin on a 16-bit grid over the scene box, exponents, interior / leaf slot masks, the unit address of the children block,
 * 8-bit quantised child planes of the 8 slots) and 48-byte triangle records (v0,prim | e1,class | e2,0) inside the children
 * blocks; the root is the node at unit 0 (layout in csrc/ptc_scene.cpp).  grid: scene_lo.xyz, step.xyz of the origin grid.
 * Pass NULL to query sizes only. */
int ptc_debug_get_bvh(ptc_ctx*, uint32_t* n_nodes, uint32_t* n_tris, uint32_t* n_units, float* units, float grid[6]);

/* Identity (an address, as a number) of the host build the context renders from.  The contexts of a ptc_group share ONE build after
 * ptc_group_scene_commit / ptc_group_scene_refit (one flatten + BVH build for N devices): equal values say so.  0 for a null context. */
uint64_t ptc_debug_host_build_id(const ptc_ctx*);
/* Context internals for tests of the host logic: [0] HIP events created so far, [1] timing spans waiting to be
 * collected, [2] queue capacity (paths) of a lane, [3] samples of one full batch, [4] samples accepted but not yet
 * issued, [5] trace blocks per CU, [6] stack entries per lane kept in LDS, [7] bit 0: the last ptc_scene_refit ran on the
 * device (csrc/pt_refit.hip), not on the host; bit 1: the last ptc_scene_commit flattened and built on the device
 * (csrc/pt_refit.hip + csrc/pt_build.hip: the LBVH builder on a device context). */
int ptc_debug_get_internals(ptc_ctx*, uint64_t out[8]);

/* The host's share of a commit ON THE DEVICE (the LBVH builder on a device context: indices and material per primitive, materials, the emitter table from the
 * emissive primitives alone, textures, environment tables — no flatten), computed again for the committed description and held against the host build, without a device.
 * out: [0] primitives, [1] emitters, [2] indices + material per primitive agree, [3] emitter index per primitive, emitter table and cdf agree bit for bit, [4] material table,
 * [5] textures / texture sets / environment tables, [6] shading-record stride and vertex count — 1 each when equal. */
int ptc_debug_commit_host_parts(ptc_ctx*, uint64_t out[8]);

/* The tables shading reads besides the BVH, as they lie in HBM: the per-primitive shading records (4 * stride floats each,
 * stride 5 or 12), the emitters (20 floats each) and their power cdf (max(n_lights, 1) floats).  Arrays may be NULL (sizes only).
 * With ptc_debug_get_bvh this is everything a refit rewrites: the tests hold the refit on the device against the one on the
 * host and the oracle's through these two calls. */
int ptc_debug_get_shading_tables(ptc_ctx*, uint32_t* stride, float* shade, uint32_t* n_lights, float* lights, float* cdf);

/* The HOST's share of a refit on the device, runnable without one (CPU tests, sanitizer builds): the plan of the committed scene (vertex ->
 * instance, node addresses by level) and the emitter table of the current transforms from the emissive primitives alone, checked against the
 * host build; call it after ptc_scene_refit on a description-only context.  out: [0] world vertices, [1] primitives, [2] nodes in the level
 * lists, [3] levels, [4] primitives with an emissive material, [5] 1 if the lists hold every node once with children in earlier levels,
 * [6] 1 if emitter table and cdf equal the host refit's bit for bit, [7] 1 if every instance transform is finite. */
int ptc_debug_refit_host_parts(ptc_ctx*, uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* PTC_H */
