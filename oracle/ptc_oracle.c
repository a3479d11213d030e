/* ptc_oracle.c — scalar CPU restatement of the path-tracing hot path.  TEST INFRASTRUCTURE ONLY
 * (see ptc_oracle.h: who may load it, and "parity unpinned").
 *
 * Arithmetic contract shared with the HIP kernels (DESIGN.md §"Arithmetic contract"):
 *   - IEEE-754 binary32 everywhere; + - * / sqrt correctly rounded; no implicit contraction
 *     (built with -ffp-contract=off); a fused multiply-add happens exactly where fmaf() is written.
 *   - no libm transcendental on the render path: sin/cos/pow are the polynomials below.
 *   - closest hit = lexicographic minimum of (t, original primitive id); traversal order is fixed
 *     (a node's leaf triangles first, then its interior children in the order of the ray's direction
 *     octant), so node/triangle counters are reproducible too.
 *   - RNG is a counter-based hash of (seed, pixel, sample, bounce, dim).
 *
 * Reference conventions restated here (file:line under /root/reference):
 *   R1 vertex record            src/pbr_engine/engine/pbr/MeshVertex.hpp:14-19
 *   R2 primitive concatenation  src/pbr_engine/engine/pbr/MeshBuilder.cpp:16-55
 *   R3 model / normal matrix    src/pbr_engine/engine/pbr/ModelPushConstant.hpp:33-46
 *   R4 world N/T/B              assets/shaders/geometry_pass/vertex.glsl:25-36
 *   R5 camera                   src/pbr_engine/engine/pbr/CameraData.hpp:22-32
 *   R6 winding / culling        src/pbr_engine/engine/pbr/PbrRenderSystem.cpp:186
 *   R7 albedo / normal map      assets/shaders/geometry_pass/fragment.glsl:19-31
 *   R8 Blinn-Phong lighting     assets/shaders/pbr/lighting.glsl:19-29, BlinnPhong.lib.glsl:4-10
 *   R9 tonemap                  assets/shaders/tonemappers/aces+gamma.glsl:10-40, Gamma.lib.glsl:4-6
 * Third-party closed forms restated (not in /root/reference; pinned cmake/Dependencies.cmake:6):
 *   glm 1.0.1 lookAtRH, perspectiveRH_NO, translate·toMat4·scale, inverse, transpose.
 */
#define _GNU_SOURCE
#include "ptc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

/* ------------------------------------------------------------------------------------------ */
/* constants of the specification                                                              */
#define ORA_LEAF_MAX 2          /* triangles per BVH leaf                                      */
#define ORA_W 8                 /* child slots per BVH node                                    */
#define ORA_STACK 128           /* traversal stack entries (at most one group per level)       */
#define ORA_EMPTY ((int32_t)0x80000000)  /* unused child slot of a wide node                    */
#define ORA_RR_START 3          /* Russian roulette from this bounce index on                  */
#define ORA_RR_PMIN 0.05f
#define ORA_ALPHA_MIN 0.001f
#define ORA_T_INF 3.0e38f
#define ORA_TILE 32
#define ORA_ZNEAR 0.01f         /* CameraData.hpp:25 */
#define ORA_ZFAR 1024.0f        /* CameraData.hpp:26 */
#define ORA_PI 3.14159265358979323846f
#define ORA_INV_PI 0.31830988618379067154f
#define ORA_HALF_PI 1.57079632679489661923f

/* struct sizes of the byte model (DESIGN.md §"Algorithmic bytes") */
#define S_RAY 56u
#define S_HIT 16u
#define S_SHADOW 44u
#define S_NODE 64u
#define S_TRI 48u
#define S_SURF 176u
#define S_FB 16u

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
  return V3(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline v3 normalize3(v3 a) {
  float inv = 1.0f / sqrtf(dot3(a, a));
  return vscale(a, inv);
}
/* a*s + b */
static inline v3 vfma(v3 a, float s, v3 b) { return V3(fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)); }
static inline float fmin2(float a, float b) { return a < b ? a : b; }
static inline float fmax2(float a, float b) { return a > b ? a : b; }
static inline float max3c(v3 a) { return fmax2(fmax2(a.x, a.y), a.z); }
static inline float luminance(v3 c) { return fmaf(c.z, 0.0722f, fmaf(c.y, 0.7152f, c.x * 0.2126f)); }

/* ------------------------------------------------------------------------------------------ */
/* RNG: counter-based, keyed (seed, pixel, sample) then (bounce, dim)                           */
static inline uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (w >> 22) ^ w;
}
static inline uint32_t path_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
  return pcg(pixel + pcg(sample + pcg((uint32_t)seed + pcg((uint32_t)(seed >> 32)))));
}
static inline uint32_t rng_u32(uint32_t key, uint32_t bounce, uint32_t dim) {
  return pcg(pcg(bounce * 8u + dim) ^ key);
}
static inline float rng_f(uint32_t key, uint32_t bounce, uint32_t dim) {
  return (float)(rng_u32(key, bounce, dim) >> 8) * (1.0f / 16777216.0f);
}
uint32_t ora_rng_u32(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t dim) {
  return rng_u32(path_key(seed, pixel, sample), bounce, dim);
}

/* ------------------------------------------------------------------------------------------ */
/* transcendental replacements                                                                 */
/* sin(2πu), cos(2πu) for u in [0,1): quadrant + octant reduction, Taylor on [0, π/4]. */
static inline void sincos2pi(float u, float* so, float* co) {
  float x4 = u * 4.0f;
  int q = (int)x4;
  if (q > 3) q = 3;
  float r = x4 - (float)q;
  int swap = r > 0.5f;
  float rr = swap ? 1.0f - r : r;
  float x = rr * ORA_HALF_PI;
  float x2 = x * x;
  float ps = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
  float s = x * ps;
  float c = fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
  if (swap) { float t = s; s = c; c = t; }
  float S, C;
  if (q == 0) { S = s; C = c; }
  else if (q == 1) { S = c; C = -s; }
  else if (q == 2) { S = -s; C = -c; }
  else { S = -c; C = s; }
  *so = S; *co = C;
}
void ora_sincos2pi(float u, float* s, float* c) { sincos2pi(u, s, c); }

/* x^y for x > 0 (x <= 0 → 0): exp2(y·log2 x), both by polynomial. */
static inline float pt_log2(float x) {
  union { float f; uint32_t u; } b; b.f = x;
  int e = (int)((b.u >> 23) & 0xffu) - 127;
  b.u = (b.u & 0x007fffffu) | 0x3f800000u;   /* m in [1,2) */
  float m = b.f;
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  float z = (m - 1.0f) / (m + 1.0f);        /* log2 m = 2/ln2 · atanh z */
  float z2 = z * z;
  float p = fmaf(z2, fmaf(z2, fmaf(z2, fmaf(z2, 0.3205989f, 0.4121984f), 0.5770780f), 0.9617967f), 2.8853901f);
  return fmaf(z, p, (float)e);
}
static inline float pt_exp2(float x) {
  if (x < -126.0f) return 0.0f;
  if (x > 127.0f) x = 127.0f;
  float fl = floorf(x);
  float f = x - fl;                           /* [0,1) */
  float p = fmaf(f, fmaf(f, fmaf(f, fmaf(f, fmaf(f, 1.8775767e-3f, 8.9893397e-3f), 5.5826318e-2f), 2.4015361e-1f), 6.9315308e-1f), 9.9999994e-1f);
  union { float f; uint32_t u; } b;
  b.u = (uint32_t)((int)fl + 127) << 23;
  return p * b.f;
}
static inline float pt_pow(float x, float y) {
  if (!(x > 0.0f)) return 0.0f;
  return pt_exp2(y * pt_log2(x));
}
float ora_powf(float x, float y) { return pt_pow(x, y); }

/* atan2(y, x) in (-pi, pi]: octant reduction + odd minimax polynomial on [0,1] (max error ~1e-5 rad).  Only the
 * lat-long environment lookup uses it, where 1e-5 rad is a few thousandths of a texel. */
static inline float pt_atan2(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmax2(ax, ay), mn = fmin2(ax, ay);
  if (!(mx > 0.0f)) return 0.0f;
  float a = mn / mx, a2 = a * a;
  float p = fmaf(a2, fmaf(a2, fmaf(a2, fmaf(a2, fmaf(a2, -0.01172120f, 0.05265332f), -0.11643287f), 0.19354346f), -0.33262347f), 0.99997726f);
  float r = a * p;
  if (ay > ax) r = ORA_HALF_PI - r;
  if (x < 0.0f) r = ORA_PI - r;
  if (y < 0.0f) r = -r;
  return r;
}
float ora_atan2f(float y, float x) { return pt_atan2(y, x); }

/* ------------------------------------------------------------------------------------------ */
/* scene containers                                                                            */
typedef struct { float position[3], normal[3], tangent[4], texcoord[2]; } vert48; /* R1 */

typedef struct {
  float base[4]; float metallic, roughness; float emissive[3];
  int tex_color, tex_normal, tex_mr;
} material_t;

typedef struct { vert48* v; uint32_t nv; uint32_t* idx; uint32_t ni; int material; } mesh_t;
typedef struct { int mesh; float m[16]; } inst_t;   /* column-major model matrix */
typedef struct { uint8_t* px; int w, h; } tex_t;

typedef struct {                 /* 64-byte interior node of the BINARY tree (single-triangle leaves): both child boxes + child codes */
  float lo0[3], hi0[3], lo1[3], hi1[3];
  int32_t c0, c1;                /* >=0 interior index; <0 leaf: ~(first | (count-1)<<28) */
  uint32_t lo, hi;               /* the sorted positions below this node */
} node_t;

/* 8-wide node with quantised child boxes: the traversal structure (the binary node_t above is only the
 * build's intermediate).  A child plane on axis k is  org[k] + q * 2^(e[k]-127)  with q an 8-bit integer
 * (lower planes rounded down, upper planes rounded up, so the quantised box contains the exact one).
 * The node's own origin is kept to 16 bits per axis on a grid over the scene box (org = grid_lo + oq * grid_step,
 * rounded down), which lets the product store a node in 64 bytes.
 * A child sits in the slot whose octant it lies in (assign_slots below), so a ray can order the slots by
 * its direction signs alone.  code: >=0 node index; <0 leaf (as in node_t); ORA_EMPTY unused slot. */
typedef struct {
  float org[3];                              /* = fmaf(oq, grid_step, grid_lo): the node's origin on the scene's 16-bit grid */
  uint32_t oq[3];                            /* 0..65535 */
  uint32_t e[3];
  uint32_t qlo[3][ORA_W], qhi[3][ORA_W];     /* [axis][slot], values 0..255 */
  int32_t code[ORA_W];
} wnode_t;

typedef struct { v3 v0, e1, e2, ng; float area; v3 Le; float pmf; uint32_t prim; } light_t;

struct ora_ctx {
  char err[256];
  /* description */
  material_t* mats; int n_mats;
  mesh_t* meshes; int n_meshes;
  inst_t* insts; int n_insts;
  tex_t* texs; int n_texs;
  float cam_pos[3], cam_target[3], cam_fov, cam_aspect; int have_cam;
  /* committed */
  int committed;
  vert48* wv; uint32_t n_wv;           /* world-space vertices  */
  float* wbt;                          /* world-space bitangent per vertex, 3 floats (vertex.glsl:35) */
  uint32_t* widx; uint32_t n_tris;     /* 3 per triangle        */
  int32_t* tri_mat;
  /* BVH (triangles in sorted order) */
  int tex_linear;                      /* 0 = NEAREST (the reference's sampler), 1 = bilinear */
  int bvh_builder;                     /* 0 = binned SAH (default), 1 = Morton-order LBVH; see morton_split */
  uint32_t* order;                     /* sorted position → original prim id */
  v3 *tv0, *te1, *te2;                 /* per sorted position */
  node_t* nodes; uint32_t n_nodes; uint32_t max_depth;   /* binary radix tree (intermediate) */
  wnode_t* wnodes; uint32_t n_wnodes; uint32_t wdepth;   /* 8-wide collapse: what rays traverse */
  int32_t* prim_light;                 /* original prim id → light index or -1 */
  light_t* lights; float* cdf; uint32_t n_lights;
  float scene_lo[3], scene_hi[3]; float ray_eps;
  float grid_step[3];                  /* node origins: scene_lo + q * grid_step, q = 0..65535 */
  /* lat-long environment light: radiance texels, pmf, row-marginal and per-row conditional cdfs */
  float* env_px; int env_w, env_h; float* env_pmf; float* env_marg; float* env_cond; int env_ok;
  ora_stats stats;
};

static int fail(ora_ctx* c, const char* msg) { snprintf(c->err, sizeof c->err, "%s", msg); return -1; }
const char* ora_last_error(const ora_ctx* c) { return c ? c->err : "null ctx"; }
int ora_hw_threads(void) { long n = sysconf(_SC_NPROCESSORS_ONLN); return n > 0 ? (int)n : 1; }

ora_ctx* ora_create(void) {
  if (!__builtin_cpu_supports("fma")) return NULL;   /* the arithmetic contract needs hardware fma */
  ora_ctx* c = (ora_ctx*)calloc(1, sizeof *c);
  return c;
}
static void free_committed(ora_ctx* c) {
  free(c->wv); free(c->wbt); free(c->widx); free(c->tri_mat); free(c->order); free(c->tv0); free(c->te1); free(c->te2);
  free(c->nodes); free(c->wnodes); free(c->prim_light); free(c->lights); free(c->cdf);
  c->wv = NULL; c->wbt = NULL; c->widx = NULL; c->tri_mat = NULL; c->order = NULL; c->tv0 = c->te1 = c->te2 = NULL;
  c->nodes = NULL; c->wnodes = NULL; c->prim_light = NULL; c->lights = NULL; c->cdf = NULL; c->committed = 0;
}
static void free_description(ora_ctx* c) {
  for (int i = 0; i < c->n_meshes; ++i) { free(c->meshes[i].v); free(c->meshes[i].idx); }
  for (int i = 0; i < c->n_texs; ++i) free(c->texs[i].px);
  free(c->meshes); free(c->mats); free(c->insts); free(c->texs);
  free(c->env_px); free(c->env_pmf); free(c->env_marg); free(c->env_cond);
  c->env_px = c->env_pmf = c->env_marg = c->env_cond = NULL; c->env_w = c->env_h = 0; c->env_ok = 0;
  c->meshes = NULL; c->mats = NULL; c->insts = NULL; c->texs = NULL;
  c->n_meshes = c->n_mats = c->n_insts = c->n_texs = 0;
}
void ora_destroy(ora_ctx* c) { if (!c) return; free_committed(c); free_description(c); free(c); }

int ora_scene_begin(ora_ctx* c) { free_committed(c); free_description(c); c->have_cam = 0; c->tex_linear = 0; c->bvh_builder = 0; return 0; }

int ora_add_material(ora_ctx* c, const float base[4], float metallic, float roughness,
                     const float emissive[3], int tc, int tn, int tmr) {
  if (!base || !emissive) return fail(c, "add_material: null pointer");
  if (tc >= c->n_texs || tn >= c->n_texs || tmr >= c->n_texs) return fail(c, "add_material: texture id out of range");
  c->mats = (material_t*)realloc(c->mats, sizeof(material_t) * (size_t)(c->n_mats + 1));
  material_t* m = &c->mats[c->n_mats];
  memcpy(m->base, base, 16); m->metallic = metallic; m->roughness = roughness;
  memcpy(m->emissive, emissive, 12); m->tex_color = tc; m->tex_normal = tn; m->tex_mr = tmr;
  return c->n_mats++;
}
int ora_add_texture_rgba8(ora_ctx* c, const uint8_t* px, int w, int h) {
  if (!px || w <= 0 || h <= 0) return fail(c, "add_texture: bad argument");
  c->texs = (tex_t*)realloc(c->texs, sizeof(tex_t) * (size_t)(c->n_texs + 1));
  tex_t* t = &c->texs[c->n_texs];
  t->px = (uint8_t*)malloc((size_t)w * h * 4); memcpy(t->px, px, (size_t)w * h * 4); t->w = w; t->h = h;
  return c->n_texs++;
}
int ora_add_mesh(ora_ctx* c, const void* verts, uint32_t nv, const uint32_t* idx, uint32_t ni, int material) {
  if (!verts || !idx || nv == 0 || ni == 0 || ni % 3u) return fail(c, "add_mesh: bad argument");
  if (material < 0 || material >= c->n_mats) return fail(c, "add_mesh: material out of range");
  for (uint32_t i = 0; i < ni; ++i) if (idx[i] >= nv) return fail(c, "add_mesh: index out of range");
  c->meshes = (mesh_t*)realloc(c->meshes, sizeof(mesh_t) * (size_t)(c->n_meshes + 1));
  mesh_t* m = &c->meshes[c->n_meshes];
  m->v = (vert48*)malloc(sizeof(vert48) * nv); memcpy(m->v, verts, sizeof(vert48) * nv); m->nv = nv;
  m->idx = (uint32_t*)malloc(4u * ni); memcpy(m->idx, idx, 4u * ni); m->ni = ni; m->material = material;
  return c->n_meshes++;
}
static void make_model(const float t[3], const float q[4], const float s[3], float M[16], float N[9]);
int ora_add_instance(ora_ctx* c, int mesh, const float t[3], const float q[4], const float s[3]) {
  if (!t || !q || !s) return fail(c, "add_instance: null pointer");
  if (mesh < 0 || mesh >= c->n_meshes) return fail(c, "add_instance: mesh out of range");
  c->insts = (inst_t*)realloc(c->insts, sizeof(inst_t) * (size_t)(c->n_insts + 1));
  inst_t* in = &c->insts[c->n_insts];
  float N[9];
  in->mesh = mesh; make_model(t, q, s, in->m, N);
  return c->n_insts++;
}
int ora_add_instance_matrix(ora_ctx* c, int mesh, const float m[16]) {
  if (!m) return fail(c, "add_instance_matrix: null pointer");
  if (mesh < 0 || mesh >= c->n_meshes) return fail(c, "add_instance_matrix: mesh out of range");
  c->insts = (inst_t*)realloc(c->insts, sizeof(inst_t) * (size_t)(c->n_insts + 1));
  inst_t* in = &c->insts[c->n_insts];
  in->mesh = mesh; memcpy(in->m, m, 64);
  return c->n_insts++;
}
int ora_set_camera(ora_ctx* c, const float pos[3], const float target[3], float fov, float aspect) {
  if (!pos || !target) return fail(c, "set_camera: null pointer");
  memcpy(c->cam_pos, pos, 12); memcpy(c->cam_target, target, 12); c->cam_fov = fov; c->cam_aspect = aspect;
  c->have_cam = 1; return 0;
}

/* Lat-long environment map: w*h RGB texels, row 0 = +y (up), u = atan2(d.z, d.x)/(2pi) + 0.5.  Piecewise-constant
 * radiance; sampled proportionally to luminance x sin(theta_row) (row marginal + per-row conditional cdf). */
int ora_set_texture_filter(ora_ctx* c, int mode) {
  if (mode != 0 && mode != 1) return fail(c, "set_texture_filter: mode must be 0 (nearest) or 1 (linear)");
  c->tex_linear = mode;
  return 0;
}
int ora_set_bvh_builder(ora_ctx* c, int mode) {
  if (mode != 0 && mode != 1) return fail(c, "set_bvh_builder: mode must be 0 (SAH) or 1 (LBVH)");
  c->bvh_builder = mode;
  return 0;
}

int ora_set_env_latlong_rgb32f(ora_ctx* c, const float* px, int w, int h) {
  free(c->env_px); free(c->env_pmf); free(c->env_marg); free(c->env_cond);
  c->env_px = c->env_pmf = c->env_marg = c->env_cond = NULL; c->env_w = c->env_h = 0; c->env_ok = 0;
  if (!px) return 0;
  if (w <= 0 || h <= 0) return fail(c, "set_env: bad size");
  size_t n = (size_t)w * h;
  c->env_px = (float*)malloc(12 * n); memcpy(c->env_px, px, 12 * n);
  c->env_pmf = (float*)malloc(4 * n); c->env_cond = (float*)malloc(4 * n); c->env_marg = (float*)malloc(4 * (size_t)h);
  c->env_w = w; c->env_h = h;
  float* rowsum = (float*)malloc(4 * (size_t)h);
  float total = 0.0f;
  for (int y = 0; y < h; ++y) {
    float sr = (float)sin(3.14159265358979323846 * ((double)y + 0.5) / (double)h);
    float run = 0.0f;
    for (int x = 0; x < w; ++x) {
      const float* t = &c->env_px[((size_t)y * w + x) * 3];
      float f = luminance(V3(t[0], t[1], t[2])) * sr;
      if (!(f > 0.0f)) f = 0.0f;
      c->env_pmf[(size_t)y * w + x] = f;
      run += f;
      c->env_cond[(size_t)y * w + x] = run;
    }
    rowsum[y] = run; total += run;
    for (int x = 0; x < w; ++x) c->env_cond[(size_t)y * w + x] = run > 0.0f ? c->env_cond[(size_t)y * w + x] / run : (float)(x + 1) / (float)w;
    c->env_cond[(size_t)y * w + (w - 1)] = 1.0f;
  }
  if (total > 0.0f) {
    float run = 0.0f;
    for (int y = 0; y < h; ++y) { run += rowsum[y]; c->env_marg[y] = run / total; }
    c->env_marg[h - 1] = 1.0f;
    for (size_t i = 0; i < n; ++i) c->env_pmf[i] = c->env_pmf[i] / total;
    c->env_ok = 1;
  }
  free(rowsum);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* R3: model = translate(p)·toMat4(q)·scale(s); normalModel = mat3(transpose(inverse(model))).   */
/* Column-major (glm): m[col*4+row].  Quaternion order (w,x,y,z) — gltf/Asset.cpp:242.           */
static void normal_matrix(const float M[16], float N[9]) {
  /* inverse-transpose of the upper 3x3 A of M = cofactor(A)/det(A); a(r,c) = M[c*4+r] */
#define a(r, c) M[(c) * 4 + (r)]
  float c00 = a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1);
  float c01 = a(1, 2) * a(2, 0) - a(1, 0) * a(2, 2);
  float c02 = a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0);
  float c10 = a(0, 2) * a(2, 1) - a(0, 1) * a(2, 2);
  float c11 = a(0, 0) * a(2, 2) - a(0, 2) * a(2, 0);
  float c12 = a(0, 1) * a(2, 0) - a(0, 0) * a(2, 1);
  float c20 = a(0, 1) * a(1, 2) - a(0, 2) * a(1, 1);
  float c21 = a(0, 2) * a(1, 0) - a(0, 0) * a(1, 2);
  float c22 = a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0);
  float det = a(0, 0) * c00 + a(0, 1) * c01 + a(0, 2) * c02;
#undef a
  float id = 1.0f / det;
  /* N(r,c) = cof(r,c)/det, stored column-major N[c*3+r] */
  N[0] = c00 * id; N[1] = c10 * id; N[2] = c20 * id;
  N[3] = c01 * id; N[4] = c11 * id; N[5] = c21 * id;
  N[6] = c02 * id; N[7] = c12 * id; N[8] = c22 * id;
}
static void make_model(const float t[3], const float q[4], const float s[3], float M[16], float N[9]) {
  float w = q[0], x = q[1], y = q[2], z = q[3];
  float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z;
  float qwx = w * x, qwy = w * y, qwz = w * z;
  float R[9]; /* R[col*3+row], glm::mat3_cast */
  R[0] = 1.0f - 2.0f * (qyy + qzz); R[1] = 2.0f * (qxy + qwz);        R[2] = 2.0f * (qxz - qwy);
  R[3] = 2.0f * (qxy - qwz);        R[4] = 1.0f - 2.0f * (qxx + qzz); R[5] = 2.0f * (qyz + qwx);
  R[6] = 2.0f * (qxz + qwy);        R[7] = 2.0f * (qyz - qwx);        R[8] = 1.0f - 2.0f * (qxx + qyy);
  for (int col = 0; col < 3; ++col) { for (int row = 0; row < 3; ++row) M[col * 4 + row] = R[col * 3 + row] * s[col]; M[col * 4 + 3] = 0.0f; }
  M[12] = t[0]; M[13] = t[1]; M[14] = t[2]; M[15] = 1.0f;
  normal_matrix(M, N);
}
void ora_make_model(const float t[3], const float q[4], const float s[3], float M[16], float N[9]) { make_model(t, q, s, M, N); }

static inline v3 mat3_mul(const float N[9], v3 v) { /* col0*x + col1*y + col2*z, glm order */
  return V3(N[0] * v.x + N[3] * v.y + N[6] * v.z, N[1] * v.x + N[4] * v.y + N[7] * v.z, N[2] * v.x + N[5] * v.y + N[8] * v.z);
}
static inline v3 mat4_point(const float M[16], v3 p) { /* m[0]*x + m[1]*y + m[2]*z + m[3]*1 */
  return V3(M[0] * p.x + M[4] * p.y + M[8] * p.z + M[12], M[1] * p.x + M[5] * p.y + M[9] * p.z + M[13],
            M[2] * p.x + M[6] * p.y + M[10] * p.z + M[14]);
}

/* R5: camera basis of glm::lookAtRH(pos, target, up=(0,-1,0)): f, s, u.                          */
typedef struct { v3 pos, f, s, u; float sx, sy; } camera_t;
static camera_t make_camera_basis(const float pos[3], const float target[3], float fov, float aspect) {
  camera_t c;
  c.pos = V3(pos[0], pos[1], pos[2]);
  c.f = normalize3(vsub(V3(target[0], target[1], target[2]), c.pos));
  c.s = normalize3(cross3(c.f, V3(0.0f, -1.0f, 0.0f)));
  c.u = cross3(c.s, c.f);
  float th = (float)tan((double)fov * 0.5);
  c.sy = th;
  c.sx = aspect * th;
  return c;
}
void ora_make_camera(const float pos[3], const float target[3], float fov, float aspect, float V[16], float P[16]) {
  camera_t c = make_camera_basis(pos, target, fov, aspect);
  /* glm::lookAtRH result, column-major */
  V[0] = c.s.x; V[4] = c.s.y; V[8] = c.s.z;  V[12] = -dot3(c.s, c.pos);
  V[1] = c.u.x; V[5] = c.u.y; V[9] = c.u.z;  V[13] = -dot3(c.u, c.pos);
  V[2] = -c.f.x; V[6] = -c.f.y; V[10] = -c.f.z; V[14] = dot3(c.f, c.pos);
  V[3] = 0.0f; V[7] = 0.0f; V[11] = 0.0f; V[15] = 1.0f;
  /* glm::perspectiveRH_NO(fov, aspect, ZNEAR, ZFAR) */
  memset(P, 0, 64);
  float th = (float)tan((double)fov * 0.5);
  P[0] = 1.0f / (aspect * th);
  P[5] = 1.0f / th;
  P[10] = -(ORA_ZFAR + ORA_ZNEAR) / (ORA_ZFAR - ORA_ZNEAR);
  P[11] = -1.0f;
  P[14] = -(2.0f * ORA_ZFAR * ORA_ZNEAR) / (ORA_ZFAR - ORA_ZNEAR);
}

/* ------------------------------------------------------------------------------------------ */
/* P2: BVH.  Top-down binned-SAH binary tree over the triangle boxes (sah_split below) down to     */
/* single triangles; then the 8-wide collapse (cost-optimal, by dynamic programming, leaves of    */
/* <= ORA_LEAF_MAX triangles), the octant slot assignment and the 8-bit quantisation.             */
static inline int32_t leaf_code(uint32_t first, uint32_t count) { return (int32_t)~(first | ((count - 1u) << 28)); }

typedef struct { ora_ctx* c; const float* tlo; const float* thi; uint32_t next; uint32_t depth_max; const uint64_t* codes; } build_t;

static void range_box(build_t* b, uint32_t lo, uint32_t hi, float blo[3], float bhi[3]) {
  for (int k = 0; k < 3; ++k) { blo[k] = INFINITY; bhi[k] = -INFINITY; }
  for (uint32_t i = lo; i <= hi; ++i) {
    uint32_t p = b->c->order[i];
    for (int k = 0; k < 3; ++k) {
      blo[k] = fmin2(blo[k], b->tlo[p * 3 + k]);
      bhi[k] = fmax2(bhi[k], b->thi[p * 3 + k]);
    }
  }
}
/* Binned surface-area split of order[lo..hi]: per axis, ORA_SAH_BINS equal bins over the range's centroid bounds,
 * cost(split) = half_area(L)*nL + half_area(R)*nR, minimum over axes and bin boundaries (ties: lowest axis, then
 * lowest boundary); order[lo..hi] is then partitioned STABLY (left = bin <= boundary).  When all centroids
 * coincide the range is cut at its middle index.  Returns the last index of the left part. */
#define ORA_SAH_BINS 32
static inline float half_area(const float lo[3], const float hi[3]);
static uint32_t sah_split(build_t* b, uint32_t lo, uint32_t hi) {
  uint32_t* ord = b->c->order;
  float cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = lo; i <= hi; ++i) {
    uint32_t p = ord[i];
    for (int k = 0; k < 3; ++k) { float ctr = 0.5f * (b->tlo[p * 3 + k] + b->thi[p * 3 + k]); cl[k] = fmin2(cl[k], ctr); ch[k] = fmax2(ch[k], ctr); }
  }
  float best = INFINITY; int best_axis = -1, best_bin = 0;
  for (int k = 0; k < 3; ++k) {
    if (!(ch[k] > cl[k])) continue;
    float scale = (float)ORA_SAH_BINS / (ch[k] - cl[k]);
    uint32_t cnt[ORA_SAH_BINS]; float blo[ORA_SAH_BINS][3], bhi[ORA_SAH_BINS][3];
    for (int j = 0; j < ORA_SAH_BINS; ++j) { cnt[j] = 0; for (int a = 0; a < 3; ++a) { blo[j][a] = INFINITY; bhi[j][a] = -INFINITY; } }
    for (uint32_t i = lo; i <= hi; ++i) {
      uint32_t p = ord[i];
      int j = (int)((0.5f * (b->tlo[p * 3 + k] + b->thi[p * 3 + k]) - cl[k]) * scale);
      if (j > ORA_SAH_BINS - 1) j = ORA_SAH_BINS - 1;
      cnt[j]++;
      for (int a = 0; a < 3; ++a) { blo[j][a] = fmin2(blo[j][a], b->tlo[p * 3 + a]); bhi[j][a] = fmax2(bhi[j][a], b->thi[p * 3 + a]); }
    }
    float rarea[ORA_SAH_BINS]; uint32_t rcnt[ORA_SAH_BINS];
    { float rl[3] = {INFINITY, INFINITY, INFINITY}, rh[3] = {-INFINITY, -INFINITY, -INFINITY}; uint32_t rc = 0;
      for (int j = ORA_SAH_BINS - 1; j >= 1; --j) {
        rc += cnt[j];
        for (int a = 0; a < 3; ++a) { rl[a] = fmin2(rl[a], blo[j][a]); rh[a] = fmax2(rh[a], bhi[j][a]); }
        rcnt[j] = rc; rarea[j] = rc ? half_area(rl, rh) : 0.0f;
      } }
    float ll[3] = {INFINITY, INFINITY, INFINITY}, lh[3] = {-INFINITY, -INFINITY, -INFINITY}; uint32_t lc = 0;
    for (int j = 0; j < ORA_SAH_BINS - 1; ++j) {
      lc += cnt[j];
      for (int a = 0; a < 3; ++a) { ll[a] = fmin2(ll[a], blo[j][a]); lh[a] = fmax2(lh[a], bhi[j][a]); }
      if (lc == 0 || rcnt[j + 1] == 0) continue;
      float cost = half_area(ll, lh) * (float)lc + rarea[j + 1] * (float)rcnt[j + 1];
      if (cost < best) { best = cost; best_axis = k; best_bin = j; }
    }
  }
  if (best_axis < 0) return lo + (hi - lo) / 2;
  {
    int k = best_axis; float scale = (float)ORA_SAH_BINS / (ch[k] - cl[k]);
    uint32_t n = hi - lo + 1, nl = 0, nr = 0;
    uint32_t* tmp = (uint32_t*)malloc(4u * n);
    for (uint32_t i = lo; i <= hi; ++i) {
      uint32_t p = ord[i];
      int j = (int)((0.5f * (b->tlo[p * 3 + k] + b->thi[p * 3 + k]) - cl[k]) * scale);
      if (j > ORA_SAH_BINS - 1) j = ORA_SAH_BINS - 1;
      if (j <= best_bin) ord[lo + nl++] = p; else tmp[nr++] = p;
    }
    memcpy(&ord[lo + nl], tmp, 4u * nr);
    free(tmp);
    return lo + nl - 1;
  }
}
/* LBVH (BASELINE north_star: "flattened LBVH"; Lauterbach et al. 2009, Karras 2012): the binary tree is the radix tree of the
 * triangles' Morton codes.  Code of a triangle: its box centre, per axis q = (uint32)((ctr - cl) * (2097152 / (ch - cl))) clamped to
 * 2^21 - 1 (cl, ch = bounds of all centres; 0 on a degenerate axis), bits interleaved x -> bit 0, y -> bit 1, z -> bit 2 (63 bits).
 * The initial order is ascending (code, primitive id).  The tree is the radix tree of the keys (code, sorted position) — Karras 2012,
 * sec. 4: duplicate codes are told apart by their position, so the rule below is one rule and every internal node can be found on
 * its own (the device build, csrc/pt_build.hip, does): a range splits where the highest bit in which its first and last KEY differ
 * flips from 0 to 1 — a bit of the code, or, in a range of one code, a bit of the position.  (Until round 4 a range of one code was
 * cut at its middle index, which no per-node rule reproduces.) */
static inline uint64_t expand21(uint32_t v) {
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffull;
  x = (x | x << 16) & 0x1f0000ff0000ffull;
  x = (x | x << 8) & 0x100f00f00f00f00full;
  x = (x | x << 4) & 0x10c30c30c30c30c3ull;
  x = (x | x << 2) & 0x1249249249249249ull;
  return x;
}
typedef struct { uint64_t code; uint32_t prim; } mkey_t;
static int mkey_cmp(const void* a, const void* b) {
  const mkey_t* x = (const mkey_t*)a; const mkey_t* y = (const mkey_t*)b;
  if (x->code != y->code) return x->code < y->code ? -1 : 1;
  return x->prim < y->prim ? -1 : (x->prim > y->prim ? 1 : 0);
}
/* fills order[] (sorted) and returns the codes in sorted order (caller frees) */
static uint64_t* morton_order(ora_ctx* c, const float* tlo, const float* thi, uint32_t n) {
  float cl[3] = {INFINITY, INFINITY, INFINITY}, ch[3] = {-INFINITY, -INFINITY, -INFINITY}, scale[3];
  for (uint32_t p = 0; p < n; ++p)
    for (int k = 0; k < 3; ++k) { float ctr = 0.5f * (tlo[p * 3 + k] + thi[p * 3 + k]); cl[k] = fmin2(cl[k], ctr); ch[k] = fmax2(ch[k], ctr); }
  for (int k = 0; k < 3; ++k) scale[k] = ch[k] > cl[k] ? 2097152.0f / (ch[k] - cl[k]) : 0.0f;
  mkey_t* keys = (mkey_t*)malloc(sizeof(mkey_t) * n);
  for (uint32_t p = 0; p < n; ++p) {
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
      float ctr = 0.5f * (tlo[p * 3 + k] + thi[p * 3 + k]);
      float f = (ctr - cl[k]) * scale[k];
      q[k] = f >= 2097151.0f ? 2097151u : (uint32_t)f;
    }
    keys[p].code = expand21(q[0]) | expand21(q[1]) << 1 | expand21(q[2]) << 2;
    keys[p].prim = p;
  }
  qsort(keys, n, sizeof(mkey_t), mkey_cmp);
  uint64_t* codes = (uint64_t*)malloc(8u * n);
  for (uint32_t i = 0; i < n; ++i) { c->order[i] = keys[i].prim; codes[i] = keys[i].code; }
  free(keys);
  return codes;
}
static uint32_t morton_split(const build_t* b, uint32_t lo, uint32_t hi) {
  const uint64_t c0 = b->codes[lo], c1 = b->codes[hi];
  if (c0 == c1) {                          /* the key goes on with the sorted position: split where the highest bit in which lo and hi differ turns 1 */
    const uint32_t pbit = 1u << (31 - __builtin_clz(lo ^ hi));
    return (hi & ~(pbit - 1u)) - 1u;
  }
  const uint64_t bit = 1ull << (63 - __builtin_clzll(c0 ^ c1));
  uint32_t a = lo, z = hi;                 /* codes[a] has the bit clear, codes[z] has it set */
  while (z - a > 1) { uint32_t m = a + (z - a) / 2; if (b->codes[m] & bit) z = m; else a = m; }
  return a;
}
/* Emit the interior node covering [lo,hi] (count >= 2: the binary tree goes down to single triangles; which
 * subtrees become leaves of the wide tree is decided by the collapse below); returns its index. */
static int32_t build_node(build_t* b, uint32_t lo, uint32_t hi, uint32_t depth) {
  uint32_t me = b->next++;
  if (depth > b->depth_max) b->depth_max = depth;
  uint32_t split = b->codes ? morton_split(b, lo, hi) : sah_split(b, lo, hi);
  node_t tmp; memset(&tmp, 0, sizeof tmp);
  range_box(b, lo, split, tmp.lo0, tmp.hi0);
  range_box(b, split + 1, hi, tmp.lo1, tmp.hi1);
  uint32_t n0 = split - lo + 1, n1 = hi - split;
  tmp.lo = lo; tmp.hi = hi;
  tmp.c0 = n0 == 1 ? leaf_code(lo, 1) : build_node(b, lo, split, depth + 1);
  tmp.c1 = n1 == 1 ? leaf_code(split + 1, 1) : build_node(b, split + 1, hi, depth + 1);
  b->c->nodes[me] = tmp;
  return (int32_t)me;
}

/* 8-wide collapse of the binary tree by dynamic programming over its surface-area cost (after Ylitie, Karras,
 * Laine, "Efficient Incoherent Ray Traversal on GPUs Through Compressed Wide BVHs", 2017, sec. 3.1).  For a binary
 * node n with box half-area A_n and P_n triangles, C(n,i) = the least cost of representing its subtree by at most i
 * roots (each root a leaf or an 8-wide node), i = 1..7:
 *     C(n,1) = min(C_leaf(n), C_int(n))                 C(n,i) = min(C_dist(n,i), C(n,i-1))
 *     C_leaf(n) = A_n * P_n * ORA_C_PRIM  if P_n <= ORA_LEAF_MAX, else +inf
 *     C_int(n)  = C_dist(n,8) + A_n * ORA_C_NODE        C_dist(n,j) = min over 0<k<j of C(left,k) + C(right,j-k)
 * (a single triangle costs A * ORA_C_PRIM for every i).  Ties: a leaf over an interior node, fewer roots over more,
 * the smallest k.  The root is always an 8-wide node; its children are the roots of C_dist(root,8), left to right. */
#define ORA_C_NODE 1.0f
#define ORA_C_PRIM 1.0f
typedef struct { float c[8]; uint8_t leaf1; uint8_t same[8]; uint8_t k[9]; } dp_t;   /* c[i], same[i]: i = 1..7; k[j]: j = 2..8 */
typedef struct { float lo[3], hi[3]; int32_t code; } wchild;
static inline float half_area(const float lo[3], const float hi[3]) {
  float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
  return ex * ey + ey * ez + ez * ex;
}
static inline wchild bin_child(const node_t* n, int k) {
  wchild w;
  memcpy(w.lo, k ? n->lo1 : n->lo0, 12); memcpy(w.hi, k ? n->hi1 : n->hi0, 12); w.code = k ? n->c1 : n->c0;
  return w;
}
/* Quantise the children's [lo,hi] on one axis against the node's own [org, nhi]: scale 2^(e-127) is the
 * smallest power of two with (nhi-org)/scale <= 255; lower planes floor, upper planes ceil, each then nudged
 * until the float expression  org + q*scale  really brackets the exact plane. */
static inline float pow2_biased(uint32_t e) { union { float f; uint32_t u; } b; b.u = e << 23; return b.f; }
static void quantize_axis(const float* clo, const float* chi, int n, float org, float nhi, uint32_t* e_out, uint32_t* qlo, uint32_t* qhi) {
  float f = (nhi - org) / 255.0f;
  union { float f; uint32_t u; } b; b.f = f;
  uint32_t e = (b.u >> 23) & 255u;
  if (b.u & 0x007fffffu) e += 1u;
  if (e < 1u) e = 1u;
  for (;; ++e) {
    float s = pow2_biased(e);
    int ok = 1;
    for (int i = 0; i < n && ok; ++i) {
      float fl = floorf((clo[i] - org) / s);
      if (fl < 0.0f) fl = 0.0f;
      if (fl > 255.0f) fl = 255.0f;
      int q = (int)fl;
      while (q > 0 && org + (float)q * s > clo[i]) --q;
      qlo[i] = (uint32_t)q;
      float ce = ceilf((chi[i] - org) / s);
      if (ce < 0.0f) ce = 0.0f;
      if (ce > 255.0f) { ok = 0; break; }
      int q2 = (int)ce;
      while (q2 < 255 && org + (float)q2 * s < chi[i]) ++q2;
      if (org + (float)q2 * s < chi[i]) { ok = 0; break; }
      qhi[i] = (uint32_t)q2;
    }
    if (ok) break;
  }
  *e_out = e;
}

/* Slot assignment: child i of the (<= 8) children goes to the free slot s that maximises
 *   score(i, s) = (+-)dx + (+-)dy + (+-)dz,  d = centre(child) - centre(node),  sign of axis k = bit k of s
 * (slot s is "the octant s of the node"), greedily over all (child, slot) pairs: largest score first, ties to the
 * lowest child, then the lowest slot.  Centres are 0.5f*(lo+hi); the sum is evaluated left to right. */
static void assign_slots(const wchild* list, int n, const float nlo[3], const float nhi[3], int slot_of[ORA_W]) {
  float d[ORA_W][3];
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) d[i][k] = 0.5f * (list[i].lo[k] + list[i].hi[k]) - 0.5f * (nlo[k] + nhi[k]);
  int child_done[ORA_W] = {0}, slot_used[ORA_W] = {0};
  for (int round = 0; round < n; ++round) {
    int bi = -1, bs = -1; float bsc = 0.0f;
    for (int i = 0; i < n; ++i) {
      if (child_done[i]) continue;
      for (int sl = 0; sl < ORA_W; ++sl) {
        if (slot_used[sl]) continue;
        float sc = ((sl & 1) ? d[i][0] : -d[i][0]) + ((sl & 2) ? d[i][1] : -d[i][1]) + ((sl & 4) ? d[i][2] : -d[i][2]);
        if (bi < 0 || sc > bsc) { bi = i; bs = sl; bsc = sc; }
      }
    }
    child_done[bi] = 1; slot_used[bs] = 1; slot_of[bi] = bs;
  }
}

static inline int min7(int k) { return k > 7 ? 7 : k; }
static float dp_cost(const dp_t* dp, const node_t* parent, int k, int i) {   /* C(child k of parent, i) */
  int32_t link = k ? parent->c1 : parent->c0;
  if (link < 0) return half_area(k ? parent->lo1 : parent->lo0, k ? parent->hi1 : parent->hi0) * 1.0f * ORA_C_PRIM;
  return dp[link].c[i];
}
static void dp_compute(const ora_ctx* c, dp_t* dp, int32_t n, float area) {
  const node_t* nd = &c->nodes[n];
  if (nd->c0 >= 0) dp_compute(c, dp, nd->c0, half_area(nd->lo0, nd->hi0));
  if (nd->c1 >= 0) dp_compute(c, dp, nd->c1, half_area(nd->lo1, nd->hi1));
  dp_t* d = &dp[n];
  float dist[9];
  for (int j = 2; j <= 8; ++j) {
    float best = INFINITY; int bk = 1;
    for (int k = 1; k < j; ++k) {
      float v = dp_cost(dp, nd, 0, min7(k)) + dp_cost(dp, nd, 1, min7(j - k));
      if (v < best) { best = v; bk = k; }
    }
    dist[j] = best; d->k[j] = (uint8_t)bk;
  }
  uint32_t P = nd->hi - nd->lo + 1u;
  float cleaf = P <= ORA_LEAF_MAX ? area * (float)P * ORA_C_PRIM : INFINITY;
  float cint = dist[8] + area * ORA_C_NODE;
  d->leaf1 = cleaf <= cint; d->c[1] = d->leaf1 ? cleaf : cint;
  for (int i = 2; i <= 7; ++i) {
    if (dist[i] < d->c[i - 1]) { d->c[i] = dist[i]; d->same[i] = 0; } else { d->c[i] = d->c[i - 1]; d->same[i] = 1; }
  }
}
/* the roots of the best forest of at most i roots below child k of `parent`, appended left to right */
static void dp_forest(const ora_ctx* c, const dp_t* dp, const node_t* parent, int k, int i, wchild* list, int* n) {
  int32_t link = k ? parent->c1 : parent->c0;
  wchild w = bin_child(parent, k);
  if (link < 0) { list[(*n)++] = w; return; }
  const node_t* nd = &c->nodes[link]; const dp_t* d = &dp[link];
  while (i > 1 && d->same[i]) --i;
  if (i == 1) {
    if (d->leaf1) w.code = leaf_code(nd->lo, nd->hi - nd->lo + 1u);
    list[(*n)++] = w; return;
  }
  int kk = d->k[i];
  dp_forest(c, dp, nd, 0, min7(kk), list, n);
  dp_forest(c, dp, nd, 1, min7(i - kk), list, n);
}

/* Origin and quantised planes of a wide node from the exact boxes of its children (by slot; used[s] = slot s holds a child): the origin on
 * the 16-bit scene grid, rounded down (the largest q with fmaf(q, step, lo) <= the node's lower bound), then quantize_axis per axis. */
static void quantise_wnode(const ora_ctx* c, wnode_t* w, float blo[ORA_W][3], float bhi[ORA_W][3], const int used[ORA_W]) {
  for (int k = 0; k < 3; ++k) {
    float clo[ORA_W], chi[ORA_W]; uint32_t qlo[ORA_W], qhi[ORA_W]; int sl_of[ORA_W], n = 0;
    float nlo = INFINITY, nhi = -INFINITY;
    for (int sl = 0; sl < ORA_W; ++sl) {
      if (!used[sl]) continue;
      sl_of[n] = sl; clo[n] = blo[sl][k]; chi[n] = bhi[sl][k];
      nlo = fmin2(nlo, clo[n]); nhi = fmax2(nhi, chi[n]); ++n;
    }
    float fq = floorf((nlo - c->scene_lo[k]) / c->grid_step[k]);
    if (fq < 0.0f) fq = 0.0f;
    if (fq > 65535.0f) fq = 65535.0f;
    uint32_t oq = (uint32_t)fq;
    while (oq > 0u && fmaf((float)oq, c->grid_step[k], c->scene_lo[k]) > nlo) --oq;
    w->oq[k] = oq; w->org[k] = fmaf((float)oq, c->grid_step[k], c->scene_lo[k]);
    quantize_axis(clo, chi, n, w->org[k], nhi, &w->e[k], qlo, qhi);
    for (int sl = 0; sl < ORA_W; ++sl) { w->qlo[k][sl] = 255; w->qhi[k][sl] = 0; }
    for (int i = 0; i < n; ++i) { w->qlo[k][sl_of[i]] = qlo[i]; w->qhi[k][sl_of[i]] = qhi[i]; }
  }
}

static int32_t widen(ora_ctx* c, const dp_t* dp, int32_t bin_idx, uint32_t depth, uint32_t* next, uint32_t* maxd) {
  uint32_t me = (*next)++;
  if (depth > *maxd) *maxd = depth;
  wchild list[ORA_W]; int n = 0;
  {
    const node_t* nd = &c->nodes[bin_idx]; int kk = dp[bin_idx].k[8];
    dp_forest(c, dp, nd, 0, min7(kk), list, &n);
    dp_forest(c, dp, nd, 1, min7(8 - kk), list, &n);
  }
  wnode_t w; memset(&w, 0, sizeof w);
  float nlo3[3], nhi3[3];
  for (int k = 0; k < 3; ++k) {
    float nlo = list[0].lo[k], nhi = list[0].hi[k];
    for (int i = 0; i < n; ++i) { nlo = fmin2(nlo, list[i].lo[k]); nhi = fmax2(nhi, list[i].hi[k]); }
    nlo3[k] = nlo; nhi3[k] = nhi;
  }
  int slot_of[ORA_W];
  assign_slots(list, n, nlo3, nhi3, slot_of);
  {
    float blo[ORA_W][3], bhi[ORA_W][3]; int used[ORA_W] = {0};
    for (int i = 0; i < n; ++i) { used[slot_of[i]] = 1; memcpy(blo[slot_of[i]], list[i].lo, 12); memcpy(bhi[slot_of[i]], list[i].hi, 12); }
    quantise_wnode(c, &w, blo, bhi, used);
  }
  /* recurse in slot order */
  int child_of_slot[ORA_W];
  for (int sl = 0; sl < ORA_W; ++sl) child_of_slot[sl] = -1;
  for (int i = 0; i < n; ++i) child_of_slot[slot_of[i]] = i;
  for (int sl = 0; sl < ORA_W; ++sl) {
    int i = child_of_slot[sl];
    if (i < 0) { w.code[sl] = ORA_EMPTY; continue; }
    w.code[sl] = list[i].code < 0 ? list[i].code : widen(c, dp, list[i].code, depth + 1, next, maxd);
  }
  c->wnodes[me] = w;
  return (int32_t)me;
}

/* triangle boxes; scene bounds, ray offset and the 16-bit origin grid that follow from them */
static void triangle_boxes_and_bounds(ora_ctx* c, float* tlo, float* thi) {
  uint32_t n = c->n_tris;
  for (int k = 0; k < 3; ++k) { c->scene_lo[k] = INFINITY; c->scene_hi[k] = -INFINITY; }
  for (uint32_t p = 0; p < n; ++p) {
    for (int k = 0; k < 3; ++k) {
      float a = c->wv[c->widx[p * 3 + 0]].position[k], b = c->wv[c->widx[p * 3 + 1]].position[k], d = c->wv[c->widx[p * 3 + 2]].position[k];
      float lo = fmin2(fmin2(a, b), d), hi = fmax2(fmax2(a, b), d);
      tlo[p * 3 + k] = lo; thi[p * 3 + k] = hi;
      c->scene_lo[k] = fmin2(c->scene_lo[k], lo); c->scene_hi[k] = fmax2(c->scene_hi[k], hi);
    }
  }
  float diag = fmax2(fmax2(c->scene_hi[0] - c->scene_lo[0], c->scene_hi[1] - c->scene_lo[1]), c->scene_hi[2] - c->scene_lo[2]);
  c->ray_eps = 1e-4f * fmax2(diag, 1e-6f);
  for (int k = 0; k < 3; ++k) { float st = (c->scene_hi[k] - c->scene_lo[k]) / 65535.0f; c->grid_step[k] = st > 0.0f ? st : 1.0f; }
}
/* (v0, e1, e2) per sorted position */
static void sorted_triangles(ora_ctx* c) {
  uint32_t n = c->n_tris;
  free(c->tv0); free(c->te1); free(c->te2);
  c->tv0 = (v3*)malloc(sizeof(v3) * n); c->te1 = (v3*)malloc(sizeof(v3) * n); c->te2 = (v3*)malloc(sizeof(v3) * n);
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t p = c->order[i];
    const float* a = c->wv[c->widx[p * 3 + 0]].position; const float* bq = c->wv[c->widx[p * 3 + 1]].position; const float* d = c->wv[c->widx[p * 3 + 2]].position;
    c->tv0[i] = V3(a[0], a[1], a[2]);
    c->te1[i] = V3(bq[0] - a[0], bq[1] - a[1], bq[2] - a[2]);
    c->te2[i] = V3(d[0] - a[0], d[1] - a[1], d[2] - a[2]);
  }
}
/* P7: emitter table in original primitive order, power pmf/cdf */
static void emitter_table(ora_ctx* c) {
  uint32_t n = c->n_tris;
  free(c->prim_light); free(c->lights); free(c->cdf);
  c->prim_light = (int32_t*)malloc(4u * n);
  uint32_t nl = 0;
  for (uint32_t p = 0; p < n; ++p) {
    const material_t* m = &c->mats[c->tri_mat[p]];
    c->prim_light[p] = -1;
    if (m->emissive[0] > 0.0f || m->emissive[1] > 0.0f || m->emissive[2] > 0.0f) nl++;
  }
  c->lights = (light_t*)malloc(sizeof(light_t) * (nl ? nl : 1)); c->cdf = (float*)malloc(4u * (nl ? nl : 1));
  nl = 0; float total = 0.0f;
  for (uint32_t p = 0; p < n; ++p) {
    const material_t* m = &c->mats[c->tri_mat[p]];
    if (!(m->emissive[0] > 0.0f || m->emissive[1] > 0.0f || m->emissive[2] > 0.0f)) continue;
    const float* a = c->wv[c->widx[p * 3 + 0]].position; const float* bb = c->wv[c->widx[p * 3 + 1]].position; const float* d = c->wv[c->widx[p * 3 + 2]].position;
    light_t L; L.v0 = V3(a[0], a[1], a[2]);
    L.e1 = V3(bb[0] - a[0], bb[1] - a[1], bb[2] - a[2]); L.e2 = V3(d[0] - a[0], d[1] - a[1], d[2] - a[2]);
    v3 cr = cross3(L.e1, L.e2); float len = sqrtf(dot3(cr, cr));
    L.area = 0.5f * len; L.Le = V3(m->emissive[0], m->emissive[1], m->emissive[2]);
    float wgt = L.area * luminance(L.Le);
    if (!(wgt > 0.0f)) continue;
    L.ng = vscale(cr, 1.0f / len); L.pmf = wgt; L.prim = p;
    c->prim_light[p] = (int32_t)nl; c->lights[nl++] = L; total += wgt;
  }
  c->n_lights = nl;
  float run = 0.0f;
  for (uint32_t i = 0; i < nl; ++i) { run += c->lights[i].pmf; c->cdf[i] = run / total; c->lights[i].pmf = c->lights[i].pmf / total; }
  if (nl) c->cdf[nl - 1] = 1.0f;
}
/* R2/R3/R4: instances in insertion order, triangles in index order -> world-space vertices (c->wv, c->wbt), indices, materials.
 * (Re)allocates the arrays; ora_scene_refit runs it again after the instances' matrices changed. */
static int flatten(ora_ctx* c) {
  free(c->wv); free(c->wbt); free(c->widx); free(c->tri_mat);
  c->wv = NULL; c->wbt = NULL; c->widx = NULL; c->tri_mat = NULL;
  /* R2/R3/R4 flatten: instances in insertion order, triangles in index order. */
  uint64_t nv = 0, nt = 0;
  for (int i = 0; i < c->n_insts; ++i) { nv += c->meshes[c->insts[i].mesh].nv; nt += c->meshes[c->insts[i].mesh].ni / 3; }
  if (nt >= (1u << 28)) return fail(c, "scene_commit: too many triangles");
  c->wv = (vert48*)malloc(sizeof(vert48) * nv); c->n_wv = (uint32_t)nv;
  c->wbt = (float*)malloc(12 * nv);
  c->widx = (uint32_t*)malloc(12u * nt); c->tri_mat = (int32_t*)malloc(4u * nt); c->n_tris = (uint32_t)nt;
  uint32_t vb = 0, tb = 0;
  for (int i = 0; i < c->n_insts; ++i) {
    const inst_t* in = &c->insts[i]; const mesh_t* m = &c->meshes[in->mesh];
    float M[16], N[9]; memcpy(M, in->m, 64); normal_matrix(M, N);
    for (uint32_t k = 0; k < m->nv; ++k) {
      const vert48* s = &m->v[k]; vert48* d = &c->wv[vb + k];
      v3 p = mat4_point(M, V3(s->position[0], s->position[1], s->position[2]));
      v3 n0 = V3(s->normal[0], s->normal[1], s->normal[2]);
      v3 t0 = V3(s->tangent[0], s->tangent[1], s->tangent[2]);
      v3 n = normalize3(mat3_mul(N, n0));                 /* vertex.glsl:33 */
      v3 t = normalize3(mat3_mul(N, t0));                 /* vertex.glsl:34 */
      d->position[0] = p.x; d->position[1] = p.y; d->position[2] = p.z;
      d->normal[0] = n.x; d->normal[1] = n.y; d->normal[2] = n.z;
      d->tangent[0] = t.x; d->tangent[1] = t.y; d->tangent[2] = t.z; d->tangent[3] = s->tangent[3];
      v3 bt = normalize3(mat3_mul(N, vscale(cross3(n0, t0), s->tangent[3])));   /* vertex.glsl:35 */
      c->wbt[(size_t)(vb + k) * 3 + 0] = bt.x; c->wbt[(size_t)(vb + k) * 3 + 1] = bt.y; c->wbt[(size_t)(vb + k) * 3 + 2] = bt.z;
      d->texcoord[0] = s->texcoord[0]; d->texcoord[1] = s->texcoord[1];
    }
    for (uint32_t k = 0; k < m->ni / 3; ++k) {
      c->widx[(tb + k) * 3 + 0] = vb + m->idx[k * 3 + 0];
      c->widx[(tb + k) * 3 + 1] = vb + m->idx[k * 3 + 1];
      c->widx[(tb + k) * 3 + 2] = vb + m->idx[k * 3 + 2];
      c->tri_mat[tb + k] = m->material;
    }
    vb += m->nv; tb += m->ni / 3;
  }
  for (uint32_t i = 0; i < c->n_wv; ++i) for (int k = 0; k < 3; ++k)
    if (!isfinite(c->wv[i].position[k])) return fail(c, "scene_commit: non-finite vertex position after the instance transform");
  return 0;
}

int ora_scene_commit(ora_ctx* c) {
  free_committed(c);
  if (!c->have_cam) return fail(c, "scene_commit: no camera");
  if (c->n_insts == 0) return fail(c, "scene_commit: no instances");
  { int rf = flatten(c); if (rf) return rf; }
  uint32_t n = c->n_tris;
  float* tlo = (float*)malloc(12u * n); float* thi = (float*)malloc(12u * n);
  triangle_boxes_and_bounds(c, tlo, thi);
  c->order = (uint32_t*)malloc(4u * n);
  for (uint32_t i = 0; i < n; ++i) c->order[i] = i;      /* initial order: original primitive order */
  uint64_t* codes = c->bvh_builder == 1 ? morton_order(c, tlo, thi, n) : NULL;
  c->nodes = (node_t*)calloc(n > 1 ? n : 1, sizeof(node_t));
  build_t b = {c, tlo, thi, 0, 0, codes};
  if (n == 1) {                 /* a single triangle: both children are that leaf */
    node_t r; memset(&r, 0, sizeof r);
    range_box(&b, 0, 0, r.lo0, r.hi0); range_box(&b, 0, 0, r.lo1, r.hi1);
    r.c0 = leaf_code(0, 1); r.c1 = leaf_code(0, 1); r.lo = 0; r.hi = 0;
    c->nodes[0] = r; b.next = 1;
  } else {
    build_node(&b, 0, n - 1, 0);
  }
  c->n_nodes = b.next; c->max_depth = b.depth_max;
  sorted_triangles(c);
  c->wnodes = (wnode_t*)calloc(c->n_nodes, sizeof(wnode_t));
  {
    dp_t* dp = (dp_t*)calloc(c->n_nodes, sizeof(dp_t));
    float rlo[3], rhi[3];
    for (int k = 0; k < 3; ++k) { rlo[k] = fmin2(c->nodes[0].lo0[k], c->nodes[0].lo1[k]); rhi[k] = fmax2(c->nodes[0].hi0[k], c->nodes[0].hi1[k]); }
    dp_compute(c, dp, 0, half_area(rlo, rhi));
    uint32_t next = 0, maxd = 0; widen(c, dp, 0, 0, &next, &maxd); c->n_wnodes = next; c->wdepth = maxd;
    free(dp);
  }
  free(tlo); free(thi); free(codes);
  emitter_table(c);
  uint32_t nl = c->n_lights;
  memset(&c->stats, 0, sizeof c->stats);
  c->stats.n_triangles = n; c->stats.n_bvh_nodes = c->n_wnodes; c->stats.n_emitters = nl; c->stats.bvh_max_depth = c->wdepth;
  c->committed = 1;
  return 0;
}

/* ---- scene dynamics: new transforms for committed instances, then a REFIT of the committed tree (ptc.h: ptc_update_instance,
 * ptc_scene_refit).  Same order of the triangles, same binary tree, same wide nodes and slots; what follows the vertices is recomputed
 * with the arithmetic of the commit: flatten, triangle boxes, scene box (ray offset, origin grid), the exact box of every child of every
 * wide node (bottom-up: a leaf child's box is that of its triangles, an interior child's the union of ITS children's boxes — the box of
 * the binary subtree it stands for), quantised planes, sorted triangles, emitters. */
int ora_update_instance_matrix(ora_ctx* c, int instance, const float m[16]) {
  if (!m) return fail(c, "update_instance_matrix: null pointer");
  if (instance < 0 || instance >= c->n_insts) return fail(c, "update_instance: instance out of range");
  memcpy(c->insts[instance].m, m, 64);
  return 0;
}
int ora_update_instance(ora_ctx* c, int instance, const float t[3], const float q[4], const float s[3]) {
  if (!t || !q || !s) return fail(c, "update_instance: null pointer");
  if (instance < 0 || instance >= c->n_insts) return fail(c, "update_instance: instance out of range");
  float N[9];
  make_model(t, q, s, c->insts[instance].m, N);
  return 0;
}
int ora_scene_refit(ora_ctx* c) {
  if (!c->committed) return fail(c, "scene_refit: scene not committed");
  const uint32_t n_before = c->n_tris, nv_before = c->n_wv;
  c->committed = 0;
  { int rf = flatten(c); if (rf) return rf; }
  if (c->n_tris != n_before || c->n_wv != nv_before) return fail(c, "scene_refit: the scene's meshes or instances changed since the commit (only transforms may)");
  uint32_t n = c->n_tris;
  float* tlo = (float*)malloc(12u * n); float* thi = (float*)malloc(12u * n);
  triangle_boxes_and_bounds(c, tlo, thi);
  sorted_triangles(c);
  /* exact child boxes, bottom-up: widen numbers a node before its children, so a child's index is larger than its parent's */
  float (*xlo)[ORA_W][3] = malloc(sizeof(float[ORA_W][3]) * c->n_wnodes);
  float (*xhi)[ORA_W][3] = malloc(sizeof(float[ORA_W][3]) * c->n_wnodes);
  for (uint32_t i = c->n_wnodes; i-- > 0;) {
    wnode_t* w = &c->wnodes[i];
    int used[ORA_W];
    for (int sl = 0; sl < ORA_W; ++sl) {
      used[sl] = w->code[sl] != ORA_EMPTY;
      if (!used[sl]) continue;
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      if (w->code[sl] < 0) {
        uint32_t code = (uint32_t)~w->code[sl], first = code & 0x0fffffffu, count = (code >> 28) + 1u;
        for (uint32_t t = first; t < first + count; ++t) {
          uint32_t p = c->order[t];
          for (int k = 0; k < 3; ++k) { lo[k] = fmin2(lo[k], tlo[p * 3 + k]); hi[k] = fmax2(hi[k], thi[p * 3 + k]); }
        }
      } else {
        const wnode_t* ch = &c->wnodes[w->code[sl]];
        for (int s2 = 0; s2 < ORA_W; ++s2) {
          if (ch->code[s2] == ORA_EMPTY) continue;
          for (int k = 0; k < 3; ++k) { lo[k] = fmin2(lo[k], xlo[w->code[sl]][s2][k]); hi[k] = fmax2(hi[k], xhi[w->code[sl]][s2][k]); }
        }
      }
      memcpy(xlo[i][sl], lo, 12); memcpy(xhi[i][sl], hi, 12);
    }
    quantise_wnode(c, w, xlo[i], xhi[i], used);
  }
  free(xlo); free(xhi); free(tlo); free(thi);
  emitter_table(c);
  c->stats.n_emitters = c->n_lights;
  c->committed = 1;
  return 0;
}

int ora_get_flat_scene(ora_ctx* c, uint32_t* nv, uint32_t* nt, void* verts, uint32_t* idx, int32_t* tm) {
  if (!c->committed) return fail(c, "get_flat_scene: scene not committed");
  if (nv) *nv = c->n_wv;
  if (nt) *nt = c->n_tris;
  if (verts) memcpy(verts, c->wv, sizeof(vert48) * c->n_wv);
  if (idx) memcpy(idx, c->widx, 12u * c->n_tris);
  if (tm) memcpy(tm, c->tri_mat, 4u * c->n_tris);
  return 0;
}

/* Material class of a triangle (the key the product's shading kernel sorts hits by; the oracle itself has no use for it and only
 * exports it with the triangles): a texture SET is a distinct (colour, normal, metal-rough) triple of texture ids among the materials
 * that have a texture, numbered in material order; class 0 = untextured Lambert (metallic 0, roughness >= 1), 1 = untextured GGX,
 * 2 + set % 5 = textured (7 is the product's class of an environment miss). */
static uint32_t material_class(const ora_ctx* c, int32_t mi) {
  const material_t* m = &c->mats[mi];
  if (m->tex_color < 0 && m->tex_normal < 0 && m->tex_mr < 0) return (m->metallic == 0.0f && m->roughness >= 1.0f) ? 0u : 1u;
  uint32_t set = 0;
  for (int i = 0; i < mi; ++i) {                 /* distinct textured triples before this material's first occurrence */
    const material_t* a = &c->mats[i];
    if (a->tex_color < 0 && a->tex_normal < 0 && a->tex_mr < 0) continue;
    if (a->tex_color == m->tex_color && a->tex_normal == m->tex_normal && a->tex_mr == m->tex_mr) break;   /* mi's set was first seen here */
    int seen = 0;
    for (int k = 0; k < i; ++k) {
      const material_t* b = &c->mats[k];
      if (b->tex_color == a->tex_color && b->tex_normal == a->tex_normal && b->tex_mr == a->tex_mr && !(b->tex_color < 0 && b->tex_normal < 0 && b->tex_mr < 0)) { seen = 1; break; }
    }
    if (!seen) ++set;
  }
  return 2u + set % 5u;
}
int ora_get_bvh(ora_ctx* c, uint32_t* nn, uint32_t* nt, float* nodes, float* tris) {
  if (!c->committed) return fail(c, "get_bvh: scene not committed");
  if (nn) *nn = c->n_wnodes;
  if (nt) *nt = c->n_tris;
  if (nodes) memcpy(nodes, c->wnodes, sizeof(wnode_t) * c->n_wnodes);
  if (tris) for (uint32_t i = 0; i < c->n_tris; ++i) {
    float* o = tris + (size_t)i * 12; uint32_t p = c->order[i];
    uint32_t cls = material_class(c, c->tri_mat[p]);
    o[0] = c->tv0[i].x; o[1] = c->tv0[i].y; o[2] = c->tv0[i].z; memcpy(&o[3], &p, 4);
    o[4] = c->te1[i].x; o[5] = c->te1[i].y; o[6] = c->te1[i].z; memcpy(&o[7], &cls, 4);
    o[8] = c->te2[i].x; o[9] = c->te2[i].y; o[10] = c->te2[i].z; o[11] = 0.0f;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* P3/P4: traversal                                                                             */
typedef struct { uint64_t nodes, tris; } trav_count;

typedef struct { v3 o, d, inv, ood; } ray_t;
static inline float safe_dir(float d) { return fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d; }
static inline ray_t make_ray(v3 o, v3 d) {
  ray_t r; r.o = o; r.d = d;
  r.inv = V3(1.0f / safe_dir(d.x), 1.0f / safe_dir(d.y), 1.0f / safe_dir(d.z));
  r.ood = V3(o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z);
  return r;
}
/* slab test of one child box against [tmin, tlimit]; returns hit, writes entry distance */
static inline int box_hit(const ray_t* r, const float lo[3], const float hi[3], float tmin, float tlimit, float* tn) {
  float x0 = fmaf(lo[0], r->inv.x, -r->ood.x), x1 = fmaf(hi[0], r->inv.x, -r->ood.x);
  float y0 = fmaf(lo[1], r->inv.y, -r->ood.y), y1 = fmaf(hi[1], r->inv.y, -r->ood.y);
  float z0 = fmaf(lo[2], r->inv.z, -r->ood.z), z1 = fmaf(hi[2], r->inv.z, -r->ood.z);
  float tnear = fmax2(fmax2(fmin2(x0, x1), fmin2(y0, y1)), fmax2(fmin2(z0, z1), tmin));
  float tfar = fmin2(fmin2(fmax2(x0, x1), fmax2(y0, y1)), fmin2(fmax2(z0, z1), tlimit));
  *tn = tnear;
  return tnear <= tfar;
}
/* Möller–Trumbore on (v0,e1,e2).  cull: R6 back-face culling (front = det > 0). */
static inline int tri_test(const ray_t* r, v3 v0, v3 e1, v3 e2, int cull, float* t, float* u, float* v) {
  v3 p = cross3(r->d, e2);
  float det = dot3(e1, p);
  if (cull ? !(det > 0.0f) : (det == 0.0f)) return 0;
  float inv = 1.0f / det;
  v3 tv = vsub(r->o, v0);
  float uu = dot3(tv, p) * inv;
  if (!(uu >= 0.0f && uu <= 1.0f)) return 0;
  v3 q = cross3(tv, e1);
  float vv = dot3(r->d, q) * inv;
  if (!(vv >= 0.0f && uu + vv <= 1.0f)) return 0;
  *t = dot3(e2, q) * inv; *u = uu; *v = vv;
  return 1;
}

typedef struct { float t, u, v; int32_t prim; uint32_t pos; } hit_t;

/* slab test of child slot i of a quantised node: plane distance = fma(q, scale*inv, fma(org, inv, -ood)) */
static inline int qbox_hit(const wnode_t* n, int i, const ray_t* r, float tmin, float tlimit) {
  float ax = pow2_biased(n->e[0]) * r->inv.x, ay = pow2_biased(n->e[1]) * r->inv.y, az = pow2_biased(n->e[2]) * r->inv.z;
  float bx = fmaf(n->org[0], r->inv.x, -r->ood.x), by = fmaf(n->org[1], r->inv.y, -r->ood.y), bz = fmaf(n->org[2], r->inv.z, -r->ood.z);
  float x0 = fmaf((float)n->qlo[0][i], ax, bx), x1 = fmaf((float)n->qhi[0][i], ax, bx);
  float y0 = fmaf((float)n->qlo[1][i], ay, by), y1 = fmaf((float)n->qhi[1][i], ay, by);
  float z0 = fmaf((float)n->qlo[2][i], az, bz), z1 = fmaf((float)n->qhi[2][i], az, bz);
#ifdef ORA_EXP_FP16_SLAB
  /* EXPERIMENT (tools/tree_quality.py with an oracle built -DORA_EXP_FP16_SLAB; never part of the specification): what a slab test in packed fp16 would cost in visits.
   * The planes as (1024 + q) * s + (b - 1024 s) with s, b - 1024 s and the result rounded to 11 bits, outward: every plane distance is widened by a bound on those three
   * roundings, 2^-11 (|b - 1024 a| + 1279 |a| + |t|); the interval's ends likewise. */
  {
    const float u = 1.0f / 2048.0f;
    const float ex0 = u * (fabsf(bx - 1024.0f * ax) + 1279.0f * fabsf(ax)), ey0 = u * (fabsf(by - 1024.0f * ay) + 1279.0f * fabsf(ay)), ez0 = u * (fabsf(bz - 1024.0f * az) + 1279.0f * fabsf(az));
    const float xn = fmin2(x0, x1), xf = fmax2(x0, x1), yn = fmin2(y0, y1), yf = fmax2(y0, y1), zn = fmin2(z0, z1), zf = fmax2(z0, z1);
    float tn16 = fmax2(fmax2(xn - (ex0 + u * fabsf(xn)), yn - (ey0 + u * fabsf(yn))), fmax2(zn - (ez0 + u * fabsf(zn)), tmin - u * fabsf(tmin)));
    float tf16 = fmin2(fmin2(xf + (ex0 + u * fabsf(xf)), yf + (ey0 + u * fabsf(yf))), fmin2(zf + (ez0 + u * fabsf(zf)), tlimit + u * fabsf(tlimit)));
    return tn16 <= tf16;
  }
#endif
  float tnear = fmax2(fmax2(fmin2(x0, x1), fmin2(y0, y1)), fmax2(fmin2(z0, z1), tmin));
  float tfar = fmin2(fmin2(fmax2(x0, x1), fmax2(y0, y1)), fmin2(fmax2(z0, z1), tlimit));
  return tnear <= tfar;
}
/* One node visit: the 8 slots against [tmin, tlimit] -> bit masks (bit s = slot s) of the hit interior children and
 * of the hit leaf children. */
static inline void node_visit(const wnode_t* n, const ray_t* r, float tmin, float tlimit, uint32_t* ihits, uint32_t* lhits) {
  uint32_t ih = 0, lh = 0;
  for (int i = 0; i < ORA_W; ++i) {
    if (n->code[i] == ORA_EMPTY) continue;
    if (!qbox_hit(n, i, r, tmin, tlimit)) continue;
    if (n->code[i] >= 0) ih |= 1u << i; else lh |= 1u << i;
  }
  *ihits = ih; *lhits = lh;
}
/* Direction octant of a ray: bit k set when component k of the direction is >= 0.  A child in slot s lies towards
 * +axis k when bit k of s is set, so the slots a ray enters first are those whose bits differ from the ray's:
 * slots are visited in DESCENDING order of (s ^ octant). */
static inline uint32_t ray_octant(const ray_t* r) { return (r->inv.x >= 0.0f ? 1u : 0u) | (r->inv.y >= 0.0f ? 2u : 0u) | (r->inv.z >= 0.0f ? 4u : 0u); }
static inline int next_slot(uint32_t hits, uint32_t oct) {
  int best = -1; uint32_t bk = 0;
  for (int s = 0; s < ORA_W; ++s) if (hits & (1u << s)) { uint32_t k = (uint32_t)s ^ oct; if (best < 0 || k > bk) { best = s; bk = k; } }
  return best;
}

typedef struct { int32_t node; uint32_t hits; } group_t;    /* interior children of `node` still to be visited */

/* Closest hit.  Per node: test the 8 slots against [tmin, best.t]; intersect the triangles of the hit leaf slots (slot order);
 * then descend into the hit interior children, nearest octant first; the others wait on the stack as one group (no
 * per-child entry distance is kept: a group is re-examined only through its children's own box tests). */
static hit_t trace_closest(const ora_ctx* c, v3 o, v3 d, float tmin, float tmax, int cull, trav_count* cnt) {
  hit_t best; best.t = tmax; best.prim = 0x7fffffff; best.u = best.v = 0.0f; best.pos = 0; int found = 0;
  ray_t r = make_ray(o, d);
  const uint32_t oct = ray_octant(&r);
  group_t stack[ORA_STACK]; int sp = 0; int32_t cur = 0;
  for (;;) {
    const wnode_t* n = &c->wnodes[cur];
    uint32_t ih, lh; cnt->nodes++;
    node_visit(n, &r, tmin, best.t, &ih, &lh);
    for (int s = 0; s < ORA_W; ++s) {
      if (!(lh & (1u << s))) continue;
      uint32_t code = (uint32_t)~n->code[s]; uint32_t first = code & 0x0fffffffu, count = (code >> 28) + 1u;
      for (uint32_t i = first; i < first + count; ++i) {
        float t, u, v; cnt->tris++;
        if (!tri_test(&r, c->tv0[i], c->te1[i], c->te2[i], cull, &t, &u, &v)) continue;
        int32_t pid = (int32_t)c->order[i];
        if (t > tmin && (t < best.t || (t == best.t && pid < best.prim))) { best.t = t; best.u = u; best.v = v; best.prim = pid; best.pos = i; found = 1; }
      }
    }
    group_t g; g.node = cur; g.hits = ih;
    if (!g.hits) { if (sp == 0) break; g = stack[--sp]; }
    int s = next_slot(g.hits, oct);
    g.hits &= ~(1u << s);
    cur = c->wnodes[g.node].code[s];
    if (g.hits) stack[sp++] = g;
  }
  if (!found) { best.t = -1.0f; best.prim = -1; }
  return best;
}
/* Any hit: same nodes; occlusion needs no order, so hit slots are taken in ascending slot order and the first
 * triangle hit inside (tmin, tmax) ends the ray. */
static int trace_any(const ora_ctx* c, v3 o, v3 d, float tmin, float tmax, trav_count* cnt) {
  ray_t r = make_ray(o, d);
  group_t stack[ORA_STACK]; int sp = 0; int32_t cur = 0;
  for (;;) {
    const wnode_t* n = &c->wnodes[cur];
    uint32_t ih, lh; cnt->nodes++;
    node_visit(n, &r, tmin, tmax, &ih, &lh);
    for (int s = 0; s < ORA_W; ++s) {
      if (!(lh & (1u << s))) continue;
      uint32_t code = (uint32_t)~n->code[s]; uint32_t first = code & 0x0fffffffu, count = (code >> 28) + 1u;
      for (uint32_t i = first; i < first + count; ++i) {
        float t, u, v; cnt->tris++;
        if (tri_test(&r, c->tv0[i], c->te1[i], c->te2[i], 0, &t, &u, &v) && t > tmin && t < tmax) return 1;
      }
    }
    group_t g; g.node = cur; g.hits = ih;
    if (!g.hits) { if (sp == 0) return 0; g = stack[--sp]; }
    int s = 0; while (!(g.hits & (1u << s))) ++s;
    g.hits &= ~(1u << s);
    cur = c->wnodes[g.node].code[s];
    if (g.hits) stack[sp++] = g;
  }
}

int ora_trace_closest(ora_ctx* c, const float* o, const float* d, uint32_t n, float* ot, int32_t* op, float* ouv) {
  if (!c->committed) return fail(c, "trace_closest: scene not committed");
  trav_count cnt = {0, 0};
  for (uint32_t i = 0; i < n; ++i) {
    hit_t h = trace_closest(c, V3(o[i * 3], o[i * 3 + 1], o[i * 3 + 2]), V3(d[i * 3], d[i * 3 + 1], d[i * 3 + 2]), 0.0f, ORA_T_INF, 0, &cnt);
    ot[i] = h.t; op[i] = h.prim; ouv[i * 2] = h.u; ouv[i * 2 + 1] = h.v;
  }
  c->stats.node_visits_closest = cnt.nodes; c->stats.tri_tests_closest = cnt.tris;
  return 0;
}
int ora_trace_any(ora_ctx* c, const float* o, const float* d, const float* tmax, uint32_t n, uint8_t* occ) {
  if (!c->committed) return fail(c, "trace_any: scene not committed");
  trav_count cnt = {0, 0};
  for (uint32_t i = 0; i < n; ++i)
    occ[i] = (uint8_t)trace_any(c, V3(o[i * 3], o[i * 3 + 1], o[i * 3 + 2]), V3(d[i * 3], d[i * 3 + 1], d[i * 3 + 2]), 0.0f, tmax[i], &cnt);
  c->stats.node_visits_any = cnt.nodes; c->stats.tri_tests_any = cnt.tris;
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* P6: BSDF in the local frame of the shading normal (z = n).                                    */
typedef struct { v3 cd, f0; float alpha; int ggx; } bsdf_t;

static inline bsdf_t make_bsdf(v3 base, float mt, float roughness, int lambert_class) {
  bsdf_t b;
  b.ggx = !lambert_class;                              /* metallic 0 & roughness >= 1 & no mr texture → pure Lambert class */
  b.cd = vscale(base, 1.0f - mt);
  b.f0 = V3(fmaf(base.x, mt, 0.04f * (1.0f - mt)), fmaf(base.y, mt, 0.04f * (1.0f - mt)), fmaf(base.z, mt, 0.04f * (1.0f - mt)));
  float a = roughness * roughness;
  b.alpha = fmax2(a, ORA_ALPHA_MIN);
  return b;
}
static inline float smith_g1(float x, float a2) { return (2.0f * x) / (x + sqrtf(fmaf(1.0f - a2, x * x, a2))); }
static inline v3 schlick(v3 f0, float voh) {
  float m = 1.0f - voh; if (m < 0.0f) m = 0.0f;
  float m2 = m * m; float m5 = m2 * m2 * m;
  return V3(fmaf(1.0f - f0.x, m5, f0.x), fmaf(1.0f - f0.y, m5, f0.y), fmaf(1.0f - f0.z, m5, f0.z));
}
/* probability of choosing the specular lobe, from wo only */
static inline float spec_prob(const bsdf_t* b, float nov) {
  if (!b->ggx) return 0.0f;
  float ld = luminance(b->cd);
  if (!(ld > 0.0f)) return 1.0f;
  float lf = luminance(schlick(b->f0, nov));
  float p = lf / (lf + ld);
  return fmin2(fmax2(p, 0.1f), 0.9f);
}
/* f (rgb) and pdf (solid angle) for local wo, wi (both z > 0) */
static inline void bsdf_eval(const bsdf_t* b, v3 wo, v3 wi, float ps, v3* f, float* pdf) {
  float nol = wi.z;
  v3 fd = vscale(b->cd, ORA_INV_PI);
  float pd = nol * ORA_INV_PI;
  if (!b->ggx) { *f = fd; *pdf = pd; return; }
  float nov = fmax2(wo.z, 1e-4f);
  v3 h = normalize3(vadd(wo, wi));
  float noh = h.z, voh = dot3(wo, h);
  float a2 = b->alpha * b->alpha;
  float dd = fmaf(noh * noh, a2 - 1.0f, 1.0f);
  float D = a2 / (ORA_PI * dd * dd);
  float gv = smith_g1(nov, a2), gl = smith_g1(nol, a2);
  v3 F = schlick(b->f0, voh);
  float sp = (D * gv * gl) / (4.0f * nov * nol);
  *f = V3(fmaf(F.x, sp, fd.x), fmaf(F.y, sp, fd.y), fmaf(F.z, sp, fd.z));
  float pspec = (gv * D) / (4.0f * nov);
  *pdf = fmaf(ps, pspec, (1.0f - ps) * pd);
}
/* sample local wi; returns 0 when the sample is unusable */
static inline int bsdf_sample(const bsdf_t* b, v3 wo, float ps, float ul, float s1, float s2, v3* wi) {
  float sn, cs; sincos2pi(s2, &sn, &cs);
  float r = sqrtf(s1);
  if (ul < ps) {  /* GGX VNDF (Heitz 2018) */
    float a = b->alpha;
    v3 vh = normalize3(V3(a * wo.x, a * wo.y, wo.z));
    float lensq = fmaf(vh.y, vh.y, vh.x * vh.x);
    v3 t1 = V3(1.0f, 0.0f, 0.0f);
    if (lensq > 0.0f) { float il = 1.0f / sqrtf(lensq); t1 = V3(-vh.y * il, vh.x * il, 0.0f); }
    v3 t2 = cross3(vh, t1);
    float p1 = r * cs, p2 = r * sn;
    float s = 0.5f * (1.0f + vh.z);
    p2 = fmaf(1.0f - s, sqrtf(fmax2(0.0f, 1.0f - p1 * p1)), s * p2);
    float pz = sqrtf(fmax2(0.0f, 1.0f - p1 * p1 - p2 * p2));
    v3 nh = vfma(t1, p1, vfma(t2, p2, vscale(vh, pz)));
    v3 h = normalize3(V3(a * nh.x, a * nh.y, fmax2(0.0f, nh.z)));
    float voh = dot3(wo, h);
    *wi = V3(fmaf(2.0f * voh, h.x, -wo.x), fmaf(2.0f * voh, h.y, -wo.y), fmaf(2.0f * voh, h.z, -wo.z));
  } else {        /* cosine hemisphere */
    *wi = V3(r * cs, r * sn, sqrtf(fmax2(0.0f, 1.0f - s1)));
  }
  return wi->z > 0.0f;
}
/* branchless orthonormal basis (Duff et al. 2017) */
static inline void onb(v3 n, v3* t, v3* b) {
  float sg = copysignf(1.0f, n.z);
  float a = -1.0f / (sg + n.z);
  float bb = n.x * n.y * a;
  *t = V3(fmaf(sg * n.x, n.x * a, 1.0f), sg * bb, -sg * n.x);
  *b = V3(bb, fmaf(n.y, n.y * a, sg), -n.y);
}

/* ------------------------------------------------------------------------------------------ */
/* tile ownership: 32×32 tiles, Morton walk over the tile grid, tile t → rank t mod n */
static inline uint32_t compact1by1(uint32_t x) {
  x &= 0x55555555u; x = (x ^ (x >> 1)) & 0x33333333u; x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
  x = (x ^ (x >> 4)) & 0x00ff00ffu; x = (x ^ (x >> 8)) & 0x0000ffffu; return x;
}
/* rank of tile (tx,ty) in the Morton walk restricted to the tiles_x × tiles_y grid */
static uint32_t* tile_ranks(int w, int h, uint32_t* n_tiles_out) {
  uint32_t tx = (uint32_t)(w + ORA_TILE - 1) / ORA_TILE, ty = (uint32_t)(h + ORA_TILE - 1) / ORA_TILE;
  uint32_t* rank = (uint32_t*)malloc(4u * tx * ty);
  uint32_t side = 1; while (side < tx || side < ty) side <<= 1;
  uint32_t t = 0;
  for (uint32_t m = 0; m < side * side; ++m) {
    uint32_t x = compact1by1(m), y = compact1by1(m >> 1);
    if (x < tx && y < ty) rank[y * tx + x] = t++;
  }
  *n_tiles_out = t;
  return rank;
}
int ora_tile_owner(int w, int h, int x, int y, int tile_count) {
  uint32_t nt; uint32_t* rank = tile_ranks(w, h, &nt);
  uint32_t tx = (uint32_t)(w + ORA_TILE - 1) / ORA_TILE;
  int o = (int)(rank[(uint32_t)(y / ORA_TILE) * tx + (uint32_t)(x / ORA_TILE)] % (uint32_t)tile_count);
  free(rank); return o;
}

/* ------------------------------------------------------------------------------------------ */
/* P1 + P5–P10: one camera sample → radiance                                                   */
typedef struct {
  uint64_t segments, shadow_rays, hits; trav_count closest, any;
} path_count;

typedef struct { v3 p, ng, ns; int front; const material_t* mat; float base[4]; float metallic, roughness; int lambert; } surf_t;

/* R7 sampler: the reference creates samplers with default create-info (gltf/Asset.cpp:116-117): NEAREST, REPEAT, no mips.
 * RGBA8 UNORM texel → float /255 (image/pbr/image/LoadImage.cpp:27). */
static inline void tex_fetch(const tex_t* t, float u, float v, int linear, float out[4]) {
  float fu = u - floorf(u), fv = v - floorf(v);
  if (!linear) {
    int x = (int)(fu * (float)t->w), y = (int)(fv * (float)t->h);
    if (x > t->w - 1) x = t->w - 1;
    if (y > t->h - 1) y = t->h - 1;
    const uint8_t* p = &t->px[((size_t)y * t->w + x) * 4];
    for (int k = 0; k < 4; ++k) out[k] = (float)p[k] / 255.0f;
    return;
  }
  /* PTC_FILTER_LINEAR (not what the reference does; an option): texel centres at i + 0.5, REPEAT wrap,
   * lerp(a, b, t) = fma(t, b - a, a), first along x then along y */
  float x = fmaf(fu, (float)t->w, -0.5f), y = fmaf(fv, (float)t->h, -0.5f);
  float x0f = floorf(x), y0f = floorf(y);
  float tx = x - x0f, ty = y - y0f;
  int x0 = (int)x0f, y0 = (int)y0f;
  int x1 = x0 + 1, y1 = y0 + 1;
  if (x0 < 0) x0 += t->w;
  if (y0 < 0) y0 += t->h;
  if (x1 > t->w - 1) x1 -= t->w;
  if (y1 > t->h - 1) y1 -= t->h;
  const uint8_t* p00 = &t->px[((size_t)y0 * t->w + x0) * 4]; const uint8_t* p10 = &t->px[((size_t)y0 * t->w + x1) * 4];
  const uint8_t* p01 = &t->px[((size_t)y1 * t->w + x0) * 4]; const uint8_t* p11 = &t->px[((size_t)y1 * t->w + x1) * 4];
  for (int k = 0; k < 4; ++k) {
    float c00 = (float)p00[k] / 255.0f, c10 = (float)p10[k] / 255.0f, c01 = (float)p01[k] / 255.0f, c11 = (float)p11[k] / 255.0f;
    float a = fmaf(tx, c10 - c00, c00), b = fmaf(tx, c11 - c01, c01);
    out[k] = fmaf(ty, b - a, a);
  }
}

#define LERP3(fa, fb, fc, k) fmaf((fc)[k], v, fmaf((fb)[k], u, (fa)[k] * w))

/* interpolated surface attributes at a hit; textures (base colour, normal map, metallic-roughness) applied */
static inline surf_t reconstruct(const ora_ctx* c, v3 d, const hit_t* h, int two_sided_fixup) {
  surf_t s; uint32_t p = (uint32_t)h->prim;
  uint32_t i0 = c->widx[p * 3 + 0], i1 = c->widx[p * 3 + 1], i2 = c->widx[p * 3 + 2];
  const vert48* a = &c->wv[i0]; const vert48* b = &c->wv[i1]; const vert48* e = &c->wv[i2];
  float u = h->u, v = h->v, w = 1.0f - u - v;
  s.p = V3(LERP3(a->position, b->position, e->position, 0), LERP3(a->position, b->position, e->position, 1), LERP3(a->position, b->position, e->position, 2));
  v3 e1 = V3(b->position[0] - a->position[0], b->position[1] - a->position[1], b->position[2] - a->position[2]);
  v3 e2 = V3(e->position[0] - a->position[0], e->position[1] - a->position[1], e->position[2] - a->position[2]);
  s.ng = normalize3(cross3(e1, e2));
  v3 ni = V3(LERP3(a->normal, b->normal, e->normal, 0), LERP3(a->normal, b->normal, e->normal, 1), LERP3(a->normal, b->normal, e->normal, 2));
  const material_t* m = &c->mats[c->tri_mat[p]];
  s.mat = m;
  memcpy(s.base, m->base, 16); s.metallic = m->metallic; s.roughness = m->roughness;
  s.lambert = (m->metallic == 0.0f && m->roughness >= 1.0f && m->tex_mr < 0);
  s.ns = normalize3(ni);
  if (m->tex_color >= 0 || m->tex_normal >= 0 || m->tex_mr >= 0) {
    float tu = LERP3(a->texcoord, b->texcoord, e->texcoord, 0), tv = LERP3(a->texcoord, b->texcoord, e->texcoord, 1);
    float t4[4];
    if (m->tex_color >= 0) { tex_fetch(&c->texs[m->tex_color], tu, tv, c->tex_linear, t4); for (int k = 0; k < 4; ++k) s.base[k] = m->base[k] * t4[k]; }   /* fragment.glsl:30 */
    if (m->tex_mr >= 0) { tex_fetch(&c->texs[m->tex_mr], tu, tv, c->tex_linear, t4); s.roughness = m->roughness * t4[1]; s.metallic = m->metallic * t4[2]; }   /* glTF: G = roughness, B = metallic */
    if (m->tex_normal >= 0) {                                                                                       /* fragment.glsl:24-27 */
      tex_fetch(&c->texs[m->tex_normal], tu, tv, c->tex_linear, t4);
      float nx = 2.0f * t4[0] - 1.0f, ny = 2.0f * t4[1] - 1.0f, nz = 2.0f * t4[2] - 1.0f;
      v3 ti = V3(LERP3(a->tangent, b->tangent, e->tangent, 0), LERP3(a->tangent, b->tangent, e->tangent, 1), LERP3(a->tangent, b->tangent, e->tangent, 2));
      const float* ba = &c->wbt[(size_t)i0 * 3]; const float* bb = &c->wbt[(size_t)i1 * 3]; const float* be = &c->wbt[(size_t)i2 * 3];
      v3 bi = V3(LERP3(ba, bb, be, 0), LERP3(ba, bb, be, 1), LERP3(ba, bb, be, 2));
      s.ns = normalize3(vfma(ti, nx, vfma(bi, ny, vscale(ni, nz))));
    }
  }
  v3 wo = vneg(d);
  s.front = dot3(s.ng, wo) > 0.0f;
  if (two_sided_fixup) {
    if (dot3(s.ns, s.ng) < 0.0f) s.ns = vneg(s.ns);
    if (!s.front) { s.ng = vneg(s.ng); s.ns = vneg(s.ns); }
    if (!(dot3(s.ns, wo) > 0.0f)) s.ns = s.ng;
  }
  return s;
}

/* environment radiance, texel pmf and solid-angle pdf for a unit direction */
static inline void env_lookup(const ora_ctx* c, v3 d, v3* Le, float* pdf) {
  float u = pt_atan2(d.z, d.x) * (0.5f * ORA_INV_PI) + 0.5f;
  float dy = fmin2(fmax2(d.y, -1.0f), 1.0f);
  float st = sqrtf(fmax2(0.0f, 1.0f - dy * dy));
  float vv = pt_atan2(st, dy) * ORA_INV_PI;
  int x = (int)(u * (float)c->env_w), y = (int)(vv * (float)c->env_h);
  if (x > c->env_w - 1) x = c->env_w - 1;
  if (x < 0) x = 0;
  if (y > c->env_h - 1) y = c->env_h - 1;
  if (y < 0) y = 0;
  const float* t = &c->env_px[((size_t)y * c->env_w + x) * 3];
  *Le = V3(t[0], t[1], t[2]);
  *pdf = c->env_ok ? (c->env_pmf[(size_t)y * c->env_w + x] * (float)c->env_w * (float)c->env_h) / (2.0f * ORA_PI * ORA_PI * fmax2(st, 1e-6f)) : 0.0f;
}
static inline uint32_t cdf_search(const float* cdf, uint32_t n, float r) {
  uint32_t lo = 0, hi = n - 1;
  while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (cdf[mid] > r) hi = mid; else lo = mid + 1; }
  return lo;
}
/* importance-sample a direction from the environment: row by the marginal cdf, column by the row's conditional cdf,
 * uniform inside the texel (the leftover of each random number) */
static inline v3 env_sample(const ora_ctx* c, float r1, float r2) {
  uint32_t y = cdf_search(c->env_marg, (uint32_t)c->env_h, r1);
  float m0 = y ? c->env_marg[y - 1] : 0.0f, m1 = c->env_marg[y];
  float xi_v = m1 > m0 ? (r1 - m0) / (m1 - m0) : 0.5f;
  const float* cc = &c->env_cond[(size_t)y * c->env_w];
  uint32_t x = cdf_search(cc, (uint32_t)c->env_w, r2);
  float c0 = x ? cc[x - 1] : 0.0f, c1 = cc[x];
  float xi_u = c1 > c0 ? (r2 - c0) / (c1 - c0) : 0.5f;
  xi_u = fmin2(fmax2(xi_u, 0.0f), 0.999999f); xi_v = fmin2(fmax2(xi_v, 0.0f), 0.999999f);
  float u = ((float)x + xi_u) / (float)c->env_w, vv = ((float)y + xi_v) / (float)c->env_h;
  float s2, c2, st, ct;
  sincos2pi(u, &s2, &c2);                 /* phi = 2 pi u - pi  →  cos phi = -cos(2 pi u), sin phi = -sin(2 pi u) */
  sincos2pi(0.5f * vv, &st, &ct);         /* theta = pi v */
  return V3(st * -c2, ct, st * -s2);
}

static v3 trace_path(const ora_ctx* c, const camera_t* cam, int w, int h, uint32_t px, uint32_t py, uint32_t sample,
                     uint64_t seed, int max_bounces, path_count* pc) {
  uint32_t key = path_key(seed, py * (uint32_t)w + px, sample);
  float jx = rng_f(key, 0, 0), jy = rng_f(key, 0, 1);
  float fx = ((float)px + jx) / (float)w, fy = ((float)py + jy) / (float)h;
  float dvx = (2.0f * fx - 1.0f) * cam->sx, dvy = (2.0f * fy - 1.0f) * cam->sy;
  v3 o = cam->pos;
  v3 d = normalize3(vfma(cam->s, dvx, vfma(cam->u, dvy, cam->f)));
  v3 T = V3(1.0f, 1.0f, 1.0f), L = V3(0.0f, 0.0f, 0.0f);
  float prev_pdf = 0.0f;
  /* light-kind selection probabilities for NEE: environment vs emissive triangles */
  const int has_env = c->env_px && c->env_ok;
  const float p_env = has_env ? (c->n_lights > 0 ? 0.5f : 1.0f) : 0.0f, p_area = 1.0f - p_env;
  for (int b = 0;; ++b) {
    pc->segments++;
    hit_t hit = trace_closest(c, o, d, 0.0f, ORA_T_INF, 0, &pc->closest);
    if (hit.prim < 0) {                                       /* miss: environment radiance (0 without one), MIS vs env NEE */
      if (c->env_px) {
        v3 Le; float pe; env_lookup(c, d, &Le, &pe);
        float wgt = 1.0f;
        if (b > 0) { float pl = pe * p_env; float pb2 = prev_pdf * prev_pdf; wgt = pb2 / fmaf(pl, pl, pb2); }
        L = V3(fmaf(T.x * Le.x, wgt, L.x), fmaf(T.y * Le.y, wgt, L.y), fmaf(T.z * Le.z, wgt, L.z));
      }
      break;
    }
    pc->hits++;
    surf_t s = reconstruct(c, d, &hit, 1);
    v3 wo = vneg(d);
    int li = c->prim_light[hit.prim];
    if (li >= 0 && s.front) {                                 /* emission, one-sided, MIS vs NEE */
      const light_t* lt = &c->lights[li];
      float wgt = 1.0f;
      if (b > 0) {
        float cosl = dot3(s.ng, wo);
        float pl = ((lt->pmf * (hit.t * hit.t)) / (lt->area * cosl)) * p_area;
        float pb2 = prev_pdf * prev_pdf;
        wgt = pb2 / fmaf(pl, pl, pb2);
      }
      L = V3(fmaf(T.x * lt->Le.x, wgt, L.x), fmaf(T.y * lt->Le.y, wgt, L.y), fmaf(T.z * lt->Le.z, wgt, L.z));
    }
    if (b >= max_bounces) break;
    bsdf_t bs = make_bsdf(V3(s.base[0], s.base[1], s.base[2]), s.metallic, s.roughness, s.lambert);
    v3 tx, ty; onb(s.ns, &tx, &ty);
    v3 wol = V3(dot3(tx, wo), dot3(ty, wo), dot3(s.ns, wo));
    float ps = spec_prob(&bs, fmax2(wol.z, 1e-4f));
    v3 porg = vfma(s.ng, c->ray_eps, s.p);
    uint32_t rb = (uint32_t)b + 1u;
    /* P7 next-event estimation: one light sample per bounce, environment or emissive triangle */
    int use_env = 0;
    if (has_env) use_env = c->n_lights == 0 || rng_f(key, rb, 7) < p_env;
    if (use_env) {
      float r1 = rng_f(key, rb, 1), r2 = rng_f(key, rb, 2);
      v3 wi = env_sample(c, r1, r2);
      v3 wil = V3(dot3(tx, wi), dot3(ty, wi), dot3(s.ns, wi));
      if (wil.z > 0.0f && dot3(s.ng, wi) > 0.0f) {
        v3 Le; float pe; env_lookup(c, wi, &Le, &pe);
        float pl = pe * p_env;
        if (pl > 0.0f) {
          v3 f; float pb; bsdf_eval(&bs, wol, wil, ps, &f, &pb);
          float pl2 = pl * pl;
          float wgt = pl2 / fmaf(pb, pb, pl2);
          float k = (wil.z * wgt) / pl;
          v3 contrib = V3(T.x * f.x * Le.x * k, T.y * f.y * Le.y * k, T.z * f.z * Le.z * k);
          pc->shadow_rays++;
          if (!trace_any(c, porg, wi, 0.0f, ORA_T_INF, &pc->any)) L = vadd(L, contrib);
        }
      }
    } else if (c->n_lights > 0) {
      float u0 = rng_f(key, rb, 0), r1 = rng_f(key, rb, 1), r2 = rng_f(key, rb, 2);
      uint32_t lo = cdf_search(c->cdf, c->n_lights, u0);
      const light_t* lt = &c->lights[lo];
      float su = sqrtf(r1); float bu = su * (1.0f - r2), bv = su * r2;
      v3 y = vfma(lt->e2, bv, vfma(lt->e1, bu, lt->v0));
      v3 dv = vsub(y, s.p);
      float dist2 = dot3(dv, dv);
      if (dist2 > 0.0f) {
        float dist = sqrtf(dist2);
        v3 wi = vscale(dv, 1.0f / dist);
        float cosl = -dot3(lt->ng, wi);
        v3 wil = V3(dot3(tx, wi), dot3(ty, wi), dot3(s.ns, wi));
        if (cosl > 0.0f && wil.z > 0.0f && dot3(s.ng, wi) > 0.0f) {
          float pl = ((lt->pmf * dist2) / (lt->area * cosl)) * p_area;
          v3 f; float pb; bsdf_eval(&bs, wol, wil, ps, &f, &pb);
          float pl2 = pl * pl;
          float wgt = pl2 / fmaf(pb, pb, pl2);
          float k = (wil.z * wgt) / pl;
          v3 contrib = V3(T.x * f.x * lt->Le.x * k, T.y * f.y * lt->Le.y * k, T.z * f.z * lt->Le.z * k);
          pc->shadow_rays++;
          /* visibility: the segment from the offset origin porg = P + ng*eps to the sampled point, minus its last 0.1 %
           * (aiming from porg along wi would miss the point by eps*sin and, with a large scene eps, run into the emitter
           * itself before dist*0.999) */
          v3 sv = vsub(y, porg);
          float sd = sqrtf(dot3(sv, sv));
          v3 sdir = vscale(sv, 1.0f / sd);
          if (!trace_any(c, porg, sdir, 0.0f, sd * 0.999f, &pc->any)) L = vadd(L, contrib);
        }
      }
    }
    /* P6 continuation */
    float ul = rng_f(key, rb, 3), s1 = rng_f(key, rb, 4), s2 = rng_f(key, rb, 5);
    v3 wil;
    if (!bsdf_sample(&bs, wol, ps, ul, s1, s2, &wil)) break;
    v3 wi = vfma(tx, wil.x, vfma(ty, wil.y, vscale(s.ns, wil.z)));
    if (!(dot3(s.ng, wi) > 0.0f)) break;
    v3 f; float pdf; bsdf_eval(&bs, wol, wil, ps, &f, &pdf);
    if (!(pdf > 0.0f)) break;
    float k = wil.z / pdf;
    T = V3(T.x * f.x * k, T.y * f.y * k, T.z * f.z * k);
    /* P8 Russian roulette */
    if (rb >= ORA_RR_START) {
      float q = max3c(T);
      if (!(q > 0.0f)) break;
      float pr = fmin2(fmax2(q, ORA_RR_PMIN), 1.0f);
      float ur = rng_f(key, rb, 6);
      if (ur >= pr) break;
      T = V3(T.x / pr, T.y / pr, T.z / pr);
    }
    o = porg; d = wi; prev_pdf = pdf;
  }
  return L;
}

/* IEEE binary32 -> binary16, round to nearest even, overflow -> inf, half denormals kept (what a Vulkan RGBA16F
 * render target stores of an fp32 shader output, RTE being one of the two roundings the specification allows; it is
 * the one v_cvt_f16_f32 performs), and back.  Written out in integer arithmetic: no F16C dependency. */
static inline uint16_t f32_to_f16(float f) {
  union { float f; uint32_t u; } b; b.f = f;
  uint32_t sign = (b.u >> 16) & 0x8000u, x = b.u & 0x7fffffffu;
  if (x > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((x >> 13) & 0x3ffu));     /* NaN: quiet, payload truncated */
  if (x >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);                           /* >= 2^16 (and inf) -> inf */
  if (x >= 0x38800000u) {                                                            /* normal half: >= 2^-14 */
    uint32_t m = x - 0x38000000u;                                                    /* exponent rebias 127 -> 15 */
    m += 0x0fffu + ((m >> 13) & 1u);                                                 /* round to nearest even; a carry may reach inf */
    return (uint16_t)(sign | (m >> 13));
  }
  if (x < 0x33000000u) return (uint16_t)sign;                                        /* < 2^-25 -> 0 (2^-25 itself ties to even: 0) */
  {                                                                                  /* half denormal: value = mant * 2^(e-150), unit 2^-24 */
    uint32_t e = x >> 23, mant = (x & 0x007fffffu) | 0x00800000u;
    uint32_t shift = 126u - e;                                                       /* 14..23 for e = 112..103; e = 102 -> 24 */
    uint32_t q = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1u);
    if (rem > half || (rem == half && (q & 1u))) q += 1u;
    return (uint16_t)(sign | q);
  }
}
static inline float f16_to_f32(uint16_t h) {
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
  union { float f; uint32_t u; } b;
  if (e == 31u) b.u = sign | 0x7f800000u | (m << 13);
  else if (e) b.u = sign | ((e + 112u) << 23) | (m << 13);
  else if (!m) b.u = sign;
  else { b.f = (float)m * 5.9604644775390625e-08f; b.u |= sign; }                    /* m * 2^-24, exact */
  return b.f;
}
uint16_t ora_f32_to_f16(float f) { return f32_to_f16(f); }
float ora_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
static inline float round_f16(float v) { return f16_to_f32(f32_to_f16(v)); }
/* RGBA16 UNORM as a G-buffer stores it: clamp to [0,1] (NaN -> 0), round(v * 65535) / 65535 */
static inline float round_unorm16(float v) {
  float cl = fmin2(fmax2(v, 0.0f), 1.0f);
  return (float)(uint32_t)(cl * 65535.0f + 0.5f) / 65535.0f;
}

/* Raster-compat: the reference's deferred Blinn-Phong result by ray casting the primary hit.    */
/* gbuf16: light from what the reference's G-buffer holds (GBuffer.hpp:13-16): positions and normals rounded to RGBA16F, */
/* albedo to RGBA16 UNORM, the normal not re-normalised (lighting.glsl:21,28).                                              */
static void raster_compat_pixel(const ora_ctx* c, const camera_t* cam, int w, int h, uint32_t px, uint32_t py, int gbuf16, float out[4], path_count* pc) {
  float fx = ((float)px + 0.5f) / (float)w, fy = ((float)py + 0.5f) / (float)h;
  float dvx = (2.0f * fx - 1.0f) * cam->sx, dvy = (2.0f * fy - 1.0f) * cam->sy;
  v3 dn = vfma(cam->s, dvx, vfma(cam->u, dvy, cam->f));
  float len = sqrtf(fmaf(dvy, dvy, fmaf(dvx, dvx, 1.0f)));
  v3 d = normalize3(dn);
  /* Vulkan clips NDC z to [0,1] under a -1..1 projection: near = 2fn/(f+n) */
  float dnear = (2.0f * ORA_ZFAR * ORA_ZNEAR) / (ORA_ZFAR + ORA_ZNEAR);
  pc->segments++;
  hit_t hit = trace_closest(c, cam->pos, d, dnear * len, ORA_ZFAR * len, 1, &pc->closest);
  out[0] = out[1] = out[2] = out[3] = 0.0f;                  /* G-buffer clear → colour 0 */
  if (hit.prim < 0) return;
  pc->hits++;
  /* fragment.glsl:19-31: N = normalize(TBN * (texel*2-1)) (flat texel (0.5,0.5,1) without a normal map: N = normalize(interpolated
   * normal)), albedo = color * texel */
  surf_t sf = reconstruct(c, d, &hit, 0);
  v3 P = sf.p, N = sf.ns;
  if (gbuf16) {
    P = V3(round_f16(P.x), round_f16(P.y), round_f16(P.z));
    N = V3(round_f16(N.x), round_f16(N.y), round_f16(N.z));
    for (int k = 0; k < 4; ++k) sf.base[k] = round_unorm16(sf.base[k]);
  }
  v3 V = normalize3(vsub(cam->pos, P));                      /* lighting.glsl:25 */
  v3 H = normalize3(vadd(V, V));                             /* L = V; BlinnPhong.lib.glsl:6 */
  float ndv = fmax2(dot3(N, V), 0.0f), ndh = fmax2(dot3(N, H), 0.0f);
  float s2 = ndh * ndh, s4 = s2 * s2, s8 = s4 * s4, s16 = s8 * s8, s32 = s16 * s16, spec = s32 * s32; /* pow(.,64) */
  for (int k = 0; k < 4; ++k) out[k] = fmaf(sf.base[k], ndv, spec);
}

typedef struct {
  const ora_ctx* c; camera_t cam; int w, h, spp, max_bounces, integrator, tile_rank, tile_count;
  uint64_t seed; float* out; const uint32_t* trank; uint32_t tiles_x;
  volatile int next_row; path_count pc; pthread_mutex_t* mu; path_count* total;
} job_t;

static void* worker(void* arg) {
  job_t* j = (job_t*)arg;
  path_count pc; memset(&pc, 0, sizeof pc);
  for (;;) {
    int y = __sync_fetch_and_add(&j->next_row, 1);
    if (y >= j->h) break;
    for (int x = 0; x < j->w; ++x) {
      float* o = j->out + ((size_t)y * j->w + x) * 4;
      if (j->tile_count > 1) {
        uint32_t t = j->trank[(uint32_t)(y / ORA_TILE) * j->tiles_x + (uint32_t)(x / ORA_TILE)];
        if ((int)(t % (uint32_t)j->tile_count) != j->tile_rank) { o[0] = o[1] = o[2] = o[3] = 0.0f; continue; }
      }
      if (j->integrator != 0) { raster_compat_pixel(j->c, &j->cam, j->w, j->h, (uint32_t)x, (uint32_t)y, j->integrator == 2, o, &pc); continue; }
      v3 sum = V3(0.0f, 0.0f, 0.0f);
      for (int s = 0; s < j->spp; ++s) {                       /* P10: sample-index order */
        v3 L = trace_path(j->c, &j->cam, j->w, j->h, (uint32_t)x, (uint32_t)y, (uint32_t)s, j->seed, j->max_bounces, &pc);
        sum = vadd(sum, L);
      }
      float fs = (float)j->spp;
      o[0] = sum.x / fs; o[1] = sum.y / fs; o[2] = sum.z / fs; o[3] = 1.0f;
    }
  }
  pthread_mutex_lock(j->mu);
  j->total->segments += pc.segments; j->total->shadow_rays += pc.shadow_rays; j->total->hits += pc.hits;
  j->total->closest.nodes += pc.closest.nodes; j->total->closest.tris += pc.closest.tris;
  j->total->any.nodes += pc.any.nodes; j->total->any.tris += pc.any.tris;
  pthread_mutex_unlock(j->mu);
  return NULL;
}

int ora_render(ora_ctx* c, int w, int h, int spp, uint64_t seed, int max_bounces, int integrator,
               int tile_rank, int tile_count, int n_threads, float* out) {
  if (!c->committed) return fail(c, "render: scene not committed");
  if (w <= 0 || h <= 0 || spp <= 0 || max_bounces < 0 || !out) return fail(c, "render: bad argument");
  if (integrator < 0 || integrator > 2) return fail(c, "render: unknown integrator");
  if (tile_count < 1 || tile_rank < 0 || tile_rank >= tile_count) return fail(c, "render: bad tile rank/count");
  if (n_threads <= 0) n_threads = ora_hw_threads();
  if (n_threads > 256) n_threads = 256;
  uint32_t nt; uint32_t* trank = tile_ranks(w, h, &nt);
  pthread_mutex_t mu; pthread_mutex_init(&mu, NULL);
  path_count total; memset(&total, 0, sizeof total);
  job_t j; memset(&j, 0, sizeof j);
  j.c = c; j.cam = make_camera_basis(c->cam_pos, c->cam_target, c->cam_fov, c->cam_aspect);
  j.w = w; j.h = h; j.spp = spp; j.max_bounces = max_bounces; j.integrator = integrator;
  j.tile_rank = tile_rank; j.tile_count = tile_count; j.seed = seed; j.out = out; j.trank = trank;
  j.tiles_x = (uint32_t)(w + ORA_TILE - 1) / ORA_TILE; j.next_row = 0; j.mu = &mu; j.total = &total;
  struct timespec t0, t1; clock_gettime(CLOCK_MONOTONIC, &t0);
  pthread_t th[256];
  for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, worker, &j);
  for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  pthread_mutex_destroy(&mu); free(trank);
  ora_stats* s = &c->stats;
  uint64_t owned = 0;
  {
    uint32_t nt2; uint32_t* tr = tile_ranks(w, h, &nt2); uint32_t tx = (uint32_t)(w + ORA_TILE - 1) / ORA_TILE;
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x)
      owned += ((int)(tr[(uint32_t)(y / ORA_TILE) * tx + (uint32_t)(x / ORA_TILE)] % (uint32_t)tile_count) == tile_rank);
    free(tr);
  }
  s->paths = owned * (uint64_t)(integrator != 0 ? 1 : spp);
  s->segments = total.segments; s->shadow_rays = total.shadow_rays; s->hits = total.hits;
  s->node_visits_closest = total.closest.nodes; s->tri_tests_closest = total.closest.tris;
  s->node_visits_any = total.any.nodes; s->tri_tests_any = total.any.tris;
  s->algorithmic_bytes = s->segments * (2u * S_RAY + 2u * S_HIT) + s->node_visits_closest * S_NODE + s->tri_tests_closest * S_TRI
                       + s->hits * S_SURF + s->shadow_rays * (2u * S_SHADOW) + s->node_visits_any * S_NODE + s->tri_tests_any * S_TRI
                       + s->paths * (2u * S_FB);
  s->seconds_render = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
  return 0;
}
int ora_get_stats(ora_ctx* c, ora_stats* out) { if (!out) return fail(c, "get_stats: null"); *out = c->stats; return 0; }

/* ------------------------------------------------------------------------------------------ */
/* R9: ACES fit + gamma 2.2 + clamp → RGBA8.  GLSL mat3x3(a,b,c, d,e,f, g,h,i) is column-major,   */
/* so the matrices written row-wise in aces+gamma.glsl:10-19 act as their transposes. */
static inline float rrt_odt(float c) {
  float num = c * (c + 0.0245786f) - 0.000090537f;
  float den = c * (0.983729f * c + 0.4329510f) + 0.238081f;
  return num / den;
}
void ora_tonemap_rgba8(const float* rgba, uint32_t n, uint8_t* out) {
  for (uint32_t i = 0; i < n; ++i) {
    float r = rgba[i * 4], g = rgba[i * 4 + 1], b = rgba[i * 4 + 2], a = rgba[i * 4 + 3];
    float ir = 0.59719f * r + 0.07600f * g + 0.02840f * b;
    float ig = 0.35458f * r + 0.90834f * g + 0.13383f * b;
    float ib = 0.04823f * r + 0.01566f * g + 0.83777f * b;
    float fr = rrt_odt(ir), fg = rrt_odt(ig), fb = rrt_odt(ib);
    float orr = 1.60475f * fr + -0.10208f * fg + -0.00327f * fb;
    float og = -0.53108f * fr + 1.10813f * fg + -0.07276f * fb;
    float ob = -0.07367f * fr + -0.00605f * fg + 1.07602f * fb;
    float c4[4] = {orr, og, ob, a};
    for (int k = 0; k < 4; ++k) {
      float v = c4[k];
      if (k < 3) v = pt_pow(fmax2(v, 0.0f), 1.0f / 2.2f);      /* clamp before pow: documented deviation */
      v = fmin2(fmax2(v, 0.0f), 1.0f);
      out[i * 4 + k] = (uint8_t)(int)(v * 255.0f + 0.5f);      /* UNORM8 round-to-nearest */
    }
  }
}
