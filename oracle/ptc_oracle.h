/* ptc_oracle.h — C API of the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (physically-based-renderer_amd/) never includes, links or calls anything here.
 *
 * PARITY UNPINNED against the reference: WeaponizedSchizophrenia/physically-based-renderer holds
 * no path tracer, no CPU render path and no numerical fixtures.  What this
 * oracle restates from the reference are the scene/camera/material/tonemap CONVENTIONS around the
 * hot path (rows R1–R9 of SURVEY §8a), each cited at its definition in ptc_oracle.c; the path
 * tracer itself (P1–P10) is specified by this file and is the law the HIP kernels are held to.
 */
#ifndef PTC_ORACLE_H
#define PTC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_ctx ora_ctx;

typedef struct ora_stats {
  uint64_t paths, segments, shadow_rays, hits;
  uint64_t node_visits_closest, tri_tests_closest, node_visits_any, tri_tests_any;
  uint64_t algorithmic_bytes;
  double seconds_render;
  uint32_t n_triangles, n_bvh_nodes, n_emitters, bvh_max_depth;
} ora_stats;

ora_ctx* ora_create(void);
void ora_destroy(ora_ctx*);
const char* ora_last_error(const ora_ctx*);

int ora_scene_begin(ora_ctx*);
int ora_add_material(ora_ctx*, const float base_color[4], float metallic, float roughness,
                     const float emissive[3], int tex_color, int tex_normal, int tex_mr);
int ora_add_texture_rgba8(ora_ctx*, const uint8_t* px, int w, int h);
int ora_add_mesh(ora_ctx*, const void* verts48, uint32_t n_verts, const uint32_t* indices,
                 uint32_t n_indices, int material);
int ora_add_instance(ora_ctx*, int mesh, const float t[3], const float q_wxyz[4], const float s[3]);
int ora_add_instance_matrix(ora_ctx*, int mesh, const float model16[16]);
int ora_set_camera(ora_ctx*, const float pos[3], const float target[3], float fov_y, float aspect);
int ora_set_env_latlong_rgb32f(ora_ctx*, const float* rgb, int w, int h);   /* NULL clears */
int ora_set_texture_filter(ora_ctx*, int mode);                              /* 0 nearest (default, the reference's), 1 bilinear */
int ora_set_bvh_builder(ora_ctx*, int mode);                                 /* 0 binned SAH (default), 1 Morton-order LBVH; reset by scene_begin */
int ora_scene_commit(ora_ctx*);
/* new transform for a committed instance (the id ora_add_instance* returned), then a refit of the committed tree: same topology and slots,
 * every box, record and emitter recomputed from the moved vertices (mirrors ptc_update_instance / ptc_scene_refit) */
int ora_update_instance(ora_ctx*, int instance, const float t[3], const float q_wxyz[4], const float s[3]);
int ora_update_instance_matrix(ora_ctx*, int instance, const float model16[16]);
int ora_scene_refit(ora_ctx*);

/* Renders into out_rgba (w*h*4 floats, y-down).  Pixels not owned by (tile_rank, tile_count)
 * stay 0.  n_threads <= 0 means "all online cores".  integrator: 0 path tracer, 1 raster-compat (the reference's
 * Blinn-Phong pass from fp32 inputs), 2 the same pass from the reference's G-buffer formats (RGBA16F P/N, UNORM16 albedo). */
int ora_render(ora_ctx*, int w, int h, int spp, uint64_t seed, int max_bounces, int integrator,
               int tile_rank, int tile_count, int n_threads, float* out_rgba);
int ora_get_stats(ora_ctx*, ora_stats* out);

int ora_trace_closest(ora_ctx*, const float* origins, const float* dirs, uint32_t n, float* out_t,
                      int32_t* out_prim, float* out_uv);
int ora_trace_any(ora_ctx*, const float* origins, const float* dirs, const float* tmax, uint32_t n,
                  uint8_t* out_occluded);
int ora_get_flat_scene(ora_ctx*, uint32_t* n_verts, uint32_t* n_tris, void* verts48,
                       uint32_t* indices, int32_t* tri_material);

/* BVH as built: 8-wide quantised nodes, n_nodes*65 32-bit words each: org[3] (float bits), oq[3] (the origin's 16-bit grid
 * coordinates: org = fmaf(oq, (scene_hi - scene_lo) / 65535, scene_lo)), e[3], qlo[3][8], qhi[3][8],
 * code[8] (>= 0 node index, < 0 leaf ~(first | (count-1)<<28), 0x80000000 empty slot; the product stores the same
 * information packed into 80 bytes; tests decode both), sorted triangles n_tris*12 floats (v0,prim | e1,class | e2,0).
 * NULL pointers to query sizes. */
int ora_get_bvh(ora_ctx*, uint32_t* n_nodes, uint32_t* n_tris, float* nodes, float* tris);

/* stand-alone pieces for known-answer tests */
void ora_make_model(const float t[3], const float q_wxyz[4], const float s[3], float model16[16],
                    float normal9[9]);                       /* column-major, like glm */
void ora_make_camera(const float pos[3], const float target[3], float fov_y, float aspect,
                     float view16[16], float proj16[16]);   /* column-major, like glm */
void ora_tonemap_rgba8(const float* rgba, uint32_t n_pixels, uint8_t* out_rgba8);
void ora_sincos2pi(float u, float* s, float* c);
float ora_powf(float x, float y);
float ora_atan2f(float y, float x);
uint16_t ora_f32_to_f16(float f);                            /* binary32 -> binary16, round to nearest even */
float ora_f16_to_f32(uint16_t h);
uint32_t ora_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t dim);
/* tile ownership: owner rank of pixel (x,y) for a w×h frame split over tile_count ranks */
int ora_tile_owner(int w, int h, int x, int y, int tile_count);
int ora_hw_threads(void);

#ifdef __cplusplus
}
#endif
#endif
