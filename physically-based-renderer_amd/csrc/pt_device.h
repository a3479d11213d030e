// pt_device.h — device-side arithmetic of the path-tracing core (gfx950).
//
// Arithmetic contract (DESIGN.md §"Arithmetic contract"): IEEE binary32, correctly rounded
// + - * / sqrt (-fhip-fp32-correctly-rounded-divide-sqrt), NO implicit contraction
// (-ffp-contract=off): a fused multiply-add happens exactly where __builtin_fmaf is written.
// No libm / ocml transcendental on the render path: sin/cos/pow are the polynomials below.
// These rules make every value reproducible on any IEEE machine, which is what lets the parity
// tests demand (near) bit-equality instead of a statistical comparison.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__

#define PT_LEAF_MAX 2
#define PT_RR_START 3u
#define PT_RR_PMIN 0.05f
#define PT_ALPHA_MIN 0.001f
#define PT_T_INF 3.0e38f
#define PT_PI 3.14159265358979323846f
#define PT_INV_PI 0.31830988618379067154f
#define PT_HALF_PI 1.57079632679489661923f
#define PT_ZNEAR 0.01f   // CameraData.hpp:25
#define PT_ZFAR 1024.0f  // CameraData.hpp:26

struct v3 { float x, y, z; };

PT_DEV v3 V3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV v3 operator+(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV v3 operator-(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV v3 operator-(v3 a) { return V3(-a.x, -a.y, -a.z); }
PT_DEV v3 operator*(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
PT_DEV float pt_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_DEV float dot3(v3 a, v3 b) { return pt_fma(a.z, b.z, pt_fma(a.y, b.y, a.x * b.x)); }
PT_DEV v3 cross3(v3 a, v3 b) {
  return V3(pt_fma(a.y, b.z, -(a.z * b.y)), pt_fma(a.z, b.x, -(a.x * b.z)), pt_fma(a.x, b.y, -(a.y * b.x)));
}
PT_DEV float pt_sqrt(float x) { return __builtin_sqrtf(x); }
PT_DEV v3 normalize3(v3 a) { float inv = 1.0f / pt_sqrt(dot3(a, a)); return a * inv; }
PT_DEV v3 vfma(v3 a, float s, v3 b) { return V3(pt_fma(a.x, s, b.x), pt_fma(a.y, s, b.y), pt_fma(a.z, s, b.z)); }
PT_DEV float fmin2(float a, float b) { return a < b ? a : b; }
PT_DEV float fmax2(float a, float b) { return a > b ? a : b; }
PT_DEV float max3c(v3 a) { return fmax2(fmax2(a.x, a.y), a.z); }
PT_DEV float luminance(v3 c) { return pt_fma(c.z, 0.0722f, pt_fma(c.y, 0.7152f, c.x * 0.2126f)); }

// ---- RNG: counter-based hash keyed (seed, pixel, sample) then (bounce, dim) -------------------
PT_DEV uint32_t pcg(uint32_t v) {
  uint32_t s = v * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28) + 4u)) ^ s) * 277803737u;
  return (w >> 22) ^ w;
}
PT_DEV uint32_t path_key(uint32_t seed_hash, uint32_t pixel, uint32_t sample) {
  return pcg(pixel + pcg(sample + seed_hash));   // seed_hash = pcg(seed_lo + pcg(seed_hi)), host-side
}
PT_DEV float rng_f(uint32_t key, uint32_t bounce, uint32_t dim) {
  uint32_t x = pcg(pcg(bounce * 8u + dim) ^ key);
  return (float)(x >> 8) * (1.0f / 16777216.0f);
}

// ---- sin(2πu), cos(2πu), u ∈ [0,1): quadrant + octant reduction, Taylor on [0, π/4] ------------
PT_DEV void sincos2pi(float u, float& so, float& co) {
  float x4 = u * 4.0f;
  int q = (int)x4;
  if (q > 3) q = 3;
  float r = x4 - (float)q;
  bool swap = r > 0.5f;
  float rr = swap ? 1.0f - r : r;
  float x = rr * PT_HALF_PI;
  float x2 = x * x;
  float ps = pt_fma(x2, pt_fma(x2, pt_fma(x2, pt_fma(x2, 2.7557319e-6f, -1.9841270e-4f), 8.3333333e-3f), -1.6666667e-1f), 1.0f);
  float s = x * ps;
  float c = pt_fma(x2, pt_fma(x2, pt_fma(x2, pt_fma(x2, 2.4801587e-5f, -1.3888889e-3f), 4.1666667e-2f), -0.5f), 1.0f);
  if (swap) { float t = s; s = c; c = t; }
  float S, C;
  if (q == 0) { S = s; C = c; }
  else if (q == 1) { S = c; C = -s; }
  else if (q == 2) { S = -s; C = -c; }
  else { S = -c; C = s; }
  so = S; co = C;
}

// ---- x^y for x > 0 (else 0): exp2(y·log2 x) by polynomial --------------------------------------
PT_DEV float pt_log2(float x) {
  uint32_t u = __float_as_uint(x);
  int e = (int)((u >> 23) & 0xffu) - 127;
  float m = __uint_as_float((u & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
  float z = (m - 1.0f) / (m + 1.0f);
  float z2 = z * z;
  float p = pt_fma(z2, pt_fma(z2, pt_fma(z2, pt_fma(z2, 0.3205989f, 0.4121984f), 0.5770780f), 0.9617967f), 2.8853901f);
  return pt_fma(z, p, (float)e);
}
PT_DEV float pt_exp2(float x) {
  if (x < -126.0f) return 0.0f;
  if (x > 127.0f) x = 127.0f;
  float fl = __builtin_floorf(x);
  float f = x - fl;
  float p = pt_fma(f, pt_fma(f, pt_fma(f, pt_fma(f, pt_fma(f, 1.8775767e-3f, 8.9893397e-3f), 5.5826318e-2f), 2.4015361e-1f), 6.9315308e-1f), 9.9999994e-1f);
  return p * __uint_as_float((uint32_t)((int)fl + 127) << 23);
}
PT_DEV float pt_pow(float x, float y) {
  if (!(x > 0.0f)) return 0.0f;
  return pt_exp2(y * pt_log2(x));
}

// ---- atan2(y, x) in (-pi, pi]: octant reduction + odd minimax polynomial on [0,1] (max error ~1e-5 rad) --------
PT_DEV float pt_atan2(float y, float x) {
  const float ax = __builtin_fabsf(x), ay = __builtin_fabsf(y);
  const float mx = fmax2(ax, ay), mn = fmin2(ax, ay);
  if (!(mx > 0.0f)) return 0.0f;
  const float a = mn / mx, a2 = a * a;
  const float p = pt_fma(a2, pt_fma(a2, pt_fma(a2, pt_fma(a2, pt_fma(a2, -0.01172120f, 0.05265332f), -0.11643287f), 0.19354346f), -0.33262347f), 0.99997726f);
  float r = a * p;
  if (ay > ax) r = PT_HALF_PI - r;
  if (x < 0.0f) r = PT_PI - r;
  if (y < 0.0f) r = -r;
  return r;
}

// ---- P6: BSDF in the local frame of the shading normal (z = n) ---------------------------------
struct bsdf_t { v3 cd, f0; float alpha; int ggx; };

PT_DEV bsdf_t make_bsdf(v3 base, float metallic, float roughness, bool lambert_class) {
  bsdf_t b;
  float mt = metallic;
  b.ggx = !lambert_class;
  b.cd = base * (1.0f - mt);
  float d = 0.04f * (1.0f - mt);
  b.f0 = V3(pt_fma(base.x, mt, d), pt_fma(base.y, mt, d), pt_fma(base.z, mt, d));
  float a = roughness * roughness;
  b.alpha = fmax2(a, PT_ALPHA_MIN);
  return b;
}
PT_DEV float smith_g1(float x, float a2) { return (2.0f * x) / (x + pt_sqrt(pt_fma(1.0f - a2, x * x, a2))); }
PT_DEV v3 schlick(v3 f0, float voh) {
  float m = 1.0f - voh; if (m < 0.0f) m = 0.0f;
  float m2 = m * m; float m5 = m2 * m2 * m;
  return V3(pt_fma(1.0f - f0.x, m5, f0.x), pt_fma(1.0f - f0.y, m5, f0.y), pt_fma(1.0f - f0.z, m5, f0.z));
}
PT_DEV float spec_prob(const bsdf_t& b, float nov) {
  if (!b.ggx) return 0.0f;
  float ld = luminance(b.cd);
  if (!(ld > 0.0f)) return 1.0f;
  float lf = luminance(schlick(b.f0, nov));
  float p = lf / (lf + ld);
  return fmin2(fmax2(p, 0.1f), 0.9f);
}
PT_DEV void bsdf_eval(const bsdf_t& b, v3 wo, v3 wi, float ps, v3& f, float& pdf) {
  float nol = wi.z;
  v3 fd = b.cd * PT_INV_PI;
  float pd = nol * PT_INV_PI;
  if (!b.ggx) { f = fd; pdf = pd; return; }
  float nov = fmax2(wo.z, 1e-4f);
  v3 h = normalize3(wo + wi);
  float noh = h.z, voh = dot3(wo, h);
  float a2 = b.alpha * b.alpha;
  float dd = pt_fma(noh * noh, a2 - 1.0f, 1.0f);
  float D = a2 / (PT_PI * dd * dd);
  float gv = smith_g1(nov, a2), gl = smith_g1(nol, a2);
  v3 F = schlick(b.f0, voh);
  float sp = (D * gv * gl) / (4.0f * nov * nol);
  f = V3(pt_fma(F.x, sp, fd.x), pt_fma(F.y, sp, fd.y), pt_fma(F.z, sp, fd.z));
  float pspec = (gv * D) / (4.0f * nov);
  pdf = pt_fma(ps, pspec, (1.0f - ps) * pd);
}
PT_DEV bool bsdf_sample(const bsdf_t& b, v3 wo, float ps, float ul, float s1, float s2, v3& wi) {
  float sn, cs; sincos2pi(s2, sn, cs);
  float r = pt_sqrt(s1);
  if (ul < ps) {  // GGX VNDF (Heitz 2018)
    float a = b.alpha;
    v3 vh = normalize3(V3(a * wo.x, a * wo.y, wo.z));
    float lensq = pt_fma(vh.y, vh.y, vh.x * vh.x);
    v3 t1 = V3(1.0f, 0.0f, 0.0f);
    if (lensq > 0.0f) { float il = 1.0f / pt_sqrt(lensq); t1 = V3(-vh.y * il, vh.x * il, 0.0f); }
    v3 t2 = cross3(vh, t1);
    float p1 = r * cs, p2 = r * sn;
    float s = 0.5f * (1.0f + vh.z);
    p2 = pt_fma(1.0f - s, pt_sqrt(fmax2(0.0f, 1.0f - p1 * p1)), s * p2);
    float pz = pt_sqrt(fmax2(0.0f, 1.0f - p1 * p1 - p2 * p2));
    v3 nh = vfma(t1, p1, vfma(t2, p2, vh * pz));
    v3 h = normalize3(V3(a * nh.x, a * nh.y, fmax2(0.0f, nh.z)));
    float voh = dot3(wo, h);
    wi = V3(pt_fma(2.0f * voh, h.x, -wo.x), pt_fma(2.0f * voh, h.y, -wo.y), pt_fma(2.0f * voh, h.z, -wo.z));
  } else {        // cosine hemisphere
    wi = V3(r * cs, r * sn, pt_sqrt(fmax2(0.0f, 1.0f - s1)));
  }
  return wi.z > 0.0f;
}
// branchless orthonormal basis (Duff et al. 2017)
PT_DEV void onb(v3 n, v3& t, v3& b) {
  float sg = __builtin_copysignf(1.0f, n.z);
  float a = -1.0f / (sg + n.z);
  float bb = n.x * n.y * a;
  t = V3(pt_fma(sg * n.x, n.x * a, 1.0f), sg * bb, -sg * n.x);
  b = V3(bb, pt_fma(n.y, n.y * a, sg), -n.y);
}

// ---- ray / box / triangle ----------------------------------------------------------------------
struct ray_t { v3 o, d, inv, ood; };
PT_DEV float safe_dir(float d) { return __builtin_fabsf(d) < 1e-20f ? __builtin_copysignf(1e-20f, d) : d; }
PT_DEV ray_t make_ray(v3 o, v3 d) {
  ray_t r; r.o = o; r.d = d;
#ifdef PT_EXP_RCP      // timing experiment only (results differ): what the three IEEE divisions of a refill cost — the upper bound of carrying inv from the producer
  r.inv = V3(__builtin_amdgcn_rcpf(safe_dir(d.x)), __builtin_amdgcn_rcpf(safe_dir(d.y)), __builtin_amdgcn_rcpf(safe_dir(d.z)));
#else
  r.inv = V3(1.0f / safe_dir(d.x), 1.0f / safe_dir(d.y), 1.0f / safe_dir(d.z));
#endif
  r.ood = V3(o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z);
  return r;
}
PT_DEV bool box_hit(const ray_t& r, float lx, float ly, float lz, float hx, float hy, float hz, float tmin, float tlimit, float& tn) {
  float x0 = pt_fma(lx, r.inv.x, -r.ood.x), x1 = pt_fma(hx, r.inv.x, -r.ood.x);
  float y0 = pt_fma(ly, r.inv.y, -r.ood.y), y1 = pt_fma(hy, r.inv.y, -r.ood.y);
  float z0 = pt_fma(lz, r.inv.z, -r.ood.z), z1 = pt_fma(hz, r.inv.z, -r.ood.z);
  float tnear = fmax2(fmax2(fmin2(x0, x1), fmin2(y0, y1)), fmax2(fmin2(z0, z1), tmin));
  float tfar = fmin2(fmin2(fmax2(x0, x1), fmax2(y0, y1)), fmin2(fmax2(z0, z1), tlimit));
  tn = tnear;
  return tnear <= tfar;
}
// Möller–Trumbore on (v0,e1,e2); CULL = R6 back-face culling (front = det > 0)
template <bool CULL>
PT_DEV bool tri_test(const ray_t& r, v3 v0, v3 e1, v3 e2, float& t, float& u, float& v) {
  v3 p = cross3(r.d, e2);
  float det = dot3(e1, p);
  if (CULL ? !(det > 0.0f) : (det == 0.0f)) return false;
  float inv = 1.0f / det;
  v3 tv = r.o - v0;
  float uu = dot3(tv, p) * inv;
  if (!(uu >= 0.0f && uu <= 1.0f)) return false;
  v3 q = cross3(tv, e1);
  float vv = dot3(r.d, q) * inv;
  if (!(vv >= 0.0f && uu + vv <= 1.0f)) return false;
  t = dot3(e2, q) * inv; u = uu; v = vv;
  return true;
}

// ---- wave64 helpers ------------------------------------------------------------------------------
PT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
PT_DEV uint32_t mbcnt64(uint64_t m) {   // number of set bits of m below this lane
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
