// jpeg_decode.hpp — dependency-free JPEG reader → RGBA8 (textures of a glTF asset).
//
// The reference decodes images with its vendored stb_image, 4 channels requested
// (src/pbr_engine/image/pbr/image/LoadImage.cpp:56-73 → stb/stb_image.h load_jpeg_image :3865-4030).  A JPEG decoder
// is only fixed by the standard up to the inverse DCT, the chroma upsampling filter and the colour conversion, so this
// one restates the choices stb makes there, and is pinned against the reference's own stb build (oracle/_ref, see
// tests/test_jpeg.py) bit for bit:
//   inverse DCT        the 12-bit fixed-point "islow" form with stb's rounding: column pass keeps 2 extra bits
//                      (+512 >> 10), row pass adds 65536 + (128 << 17) and shifts by 17 (stb_image.h:2430-2520)
//   coefficients       16-bit, dequantised by a 16-bit multiply (:2226, :3073-3078)
//   chroma upsampling  per output row: nearest / 3:1 vertical / 3:1 horizontal / 9:3:3:1 "hv" filter, or replication
//                      for other ratios, with stb's near/far row walk (:3465-3528, :3646-3657, :3927-3945)
//   YCbCr → RGB        20-bit fixed point with constants rounded to 12 bits (:3658-3685); RGB-tagged and Adobe
//                      CMYK / YCCK files follow :3879, :3954-3975
// Handled: baseline, extended-sequential and progressive Huffman JPEG, 8 bits, 1 / 3 / 4 components, sampling factors
// 1..4 with integer ratios, restart intervals, 8- and 16-bit quantisation tables.  Not handled (error): arithmetic
// coding, lossless and hierarchical processes, 12-bit samples.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pbr::image {

namespace jpeg_detail {

inline const std::uint8_t* zigzag() {
  static const std::uint8_t z[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                     41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                     30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  return z;
}

struct HuffTable {
  bool defined = false;
  std::uint8_t symbols[256];
  int mincode[17], maxcode[18], valptr[17];   // per code length (T.81 F.2.2.3)
  void build(const int counts[16], const std::uint8_t* syms, int n) {
    std::memcpy(symbols, syms, (std::size_t)n);
    int code = 0, k = 0;
    for (int len = 1; len <= 16; ++len) {
      valptr[len] = k; mincode[len] = code;
      code += counts[len - 1]; k += counts[len - 1];
      maxcode[len] = counts[len - 1] ? code - 1 : -1;
      if (code > (1 << len)) throw std::runtime_error("JPEG: bad Huffman code lengths");
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    defined = true;
  }
};

struct Component {
  int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
  int x = 0, y = 0;            // samples that carry picture
  int bw = 0, bh = 0;          // blocks allocated (whole MCUs)
  int dc_pred = 0;
  std::vector<std::int16_t> coef;   // bw*bh blocks of 64, natural (de-zigzagged) order
  std::vector<std::uint8_t> plane;  // (bw*8) x (bh*8) samples
};

// Entropy-coded segment reader: bytes → bits, 0xFF00 unstuffed, stops feeding (zero bits) at a marker.
class BitReader {
public:
  BitReader(const std::uint8_t* d, std::size_t n) : d_(d), n_(n) {}
  std::size_t pos = 0;
  int marker = -1;                     // marker met inside entropy data, -1 = none
  void reset() { acc_ = 0; cnt_ = 0; marker = -1; }
  int bit() { return (int)bits(1); }
  unsigned bits(int n) {
    if (n == 0) return 0;
    while (cnt_ < n) feed();
    cnt_ -= n;
    return (unsigned)((acc_ >> cnt_) & ((1u << n) - 1u));
  }
  // value of an n-bit magnitude category (T.81 F.2.2.1 EXTEND)
  int receive_extend(int n) {
    if (n == 0) return 0;
    const int v = (int)bits(n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
  }
  int decode(const HuffTable& t) {
    if (!t.defined) throw std::runtime_error("JPEG: scan uses an undefined Huffman table");
    int code = 0;
    for (int len = 1; len <= 16; ++len) {
      code = (code << 1) | bit();
      if (t.maxcode[len] >= 0 && code <= t.maxcode[len] && code >= t.mincode[len]) return t.symbols[t.valptr[len] + code - t.mincode[len]];
    }
    throw std::runtime_error("JPEG: bad Huffman code");
  }
  // at a restart boundary / end of scan: drop the bit buffer, find the marker that follows
  int next_marker() {
    acc_ = 0; cnt_ = 0;
    if (marker >= 0) { const int m = marker; marker = -1; return m; }
    while (pos < n_) {
      if (d_[pos++] != 0xFF) continue;
      while (pos < n_ && d_[pos] == 0xFF) ++pos;
      if (pos >= n_) break;
      const int m = d_[pos++];
      if (m != 0) return m;
    }
    return -1;
  }

private:
  const std::uint8_t* d_; std::size_t n_;
  std::uint64_t acc_ = 0; int cnt_ = 0;
  void feed() {
    unsigned b = 0;
    if (marker < 0 && pos < n_) {
      b = d_[pos++];
      if (b == 0xFF) {
        while (pos < n_ && d_[pos] == 0xFF) ++pos;   // fill bytes
        const unsigned c = pos < n_ ? d_[pos++] : 0xD9u;
        if (c != 0) { marker = (int)c; b = 0; }
      }
    }
    acc_ = (acc_ << 8) | b; cnt_ += 8;
  }
};

// One 1-D pass of the fixed-point inverse DCT (constants × 4096, rounded).  in[8] → the 4 even / odd partial sums.
// 64-bit intermediates: identical to 32-bit arithmetic for every valid stream, and defined (no signed overflow) for
// corrupt ones whose coefficients are out of range.
struct Idct1D {
  typedef long long I;
  I x0, x1, x2, x3, t0, t1, t2, t3;
  static int fx(double c) { return (int)(c * 4096 + 0.5); }
  Idct1D(I s0, I s1, I s2, I s3, I s4, I s5, I s6, I s7) {
    static const int c0541 = fx(0.5411961f), c1847 = fx(-1.847759065f), c0765 = fx(0.765366865f), c1175 = fx(1.175875602f), c0298 = fx(0.298631336f),
                     c2053 = fx(2.053119869f), c3072 = fx(3.072711026f), c1501 = fx(1.501321110f), c0899 = fx(-0.899976223f), c2562 = fx(-2.562915447f),
                     c1961 = fx(-1.961570560f), c0390 = fx(-0.390180644f);
    I p1 = (s2 + s6) * c0541;
    const I e2 = p1 + s6 * c1847, e3 = p1 + s2 * c0765;
    const I e0 = (s0 + s4) * 4096, e1 = (s0 - s4) * 4096;
    x0 = e0 + e3; x3 = e0 - e3; x1 = e1 + e2; x2 = e1 - e2;
    I o0 = s7, o1 = s5, o2 = s3, o3 = s1;
    I p3 = o0 + o2, p4 = o1 + o3;
    p1 = o0 + o3;
    I p2 = o1 + o2;
    const I p5 = (p3 + p4) * c1175;
    o0 *= c0298; o1 *= c2053; o2 *= c3072; o3 *= c1501;
    p1 = p5 + p1 * c0899; p2 = p5 + p2 * c2562; p3 *= c1961; p4 *= c0390;
    t3 = o3 + p1 + p4; t2 = o2 + p2 + p3; t1 = o1 + p2 + p4; t0 = o0 + p1 + p3;
  }
};
inline std::uint8_t clamp8(long long v) { return (std::uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

inline void idct_block(const std::int16_t* d, std::uint8_t* out, int stride) {
  long long tmp[64];
  for (int c = 0; c < 8; ++c) {
    const std::int16_t* s = d + c;
    if (s[8] == 0 && s[16] == 0 && s[24] == 0 && s[32] == 0 && s[40] == 0 && s[48] == 0 && s[56] == 0) {
      const long long dc = s[0] * 4;
      for (int r = 0; r < 8; ++r) tmp[r * 8 + c] = dc;
      continue;
    }
    Idct1D k(s[0], s[8], s[16], s[24], s[32], s[40], s[48], s[56]);
    const long long r = 512;
    tmp[0 * 8 + c] = (k.x0 + r + k.t3) >> 10; tmp[7 * 8 + c] = (k.x0 + r - k.t3) >> 10;
    tmp[1 * 8 + c] = (k.x1 + r + k.t2) >> 10; tmp[6 * 8 + c] = (k.x1 + r - k.t2) >> 10;
    tmp[2 * 8 + c] = (k.x2 + r + k.t1) >> 10; tmp[5 * 8 + c] = (k.x2 + r - k.t1) >> 10;
    tmp[3 * 8 + c] = (k.x3 + r + k.t0) >> 10; tmp[4 * 8 + c] = (k.x3 + r - k.t0) >> 10;
  }
  for (int rr = 0; rr < 8; ++rr) {
    const long long* v = tmp + rr * 8;
    std::uint8_t* o = out + (std::size_t)rr * stride;
    Idct1D k(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    const long long r = 65536 + (128 << 17);
    o[0] = clamp8((k.x0 + r + k.t3) >> 17); o[7] = clamp8((k.x0 + r - k.t3) >> 17);
    o[1] = clamp8((k.x1 + r + k.t2) >> 17); o[6] = clamp8((k.x1 + r - k.t2) >> 17);
    o[2] = clamp8((k.x2 + r + k.t1) >> 17); o[5] = clamp8((k.x2 + r - k.t1) >> 17);
    o[3] = clamp8((k.x3 + r + k.t0) >> 17); o[4] = clamp8((k.x3 + r - k.t0) >> 17);
  }
}

// one output row of a component from its near / far sample rows, w input samples, horizontal factor hs, vertical vs
inline void upsample_row(std::uint8_t* out, const std::uint8_t* near_, const std::uint8_t* far_, int w, int hs, int vs) {
  if (hs == 1 && vs == 1) { std::memcpy(out, near_, (std::size_t)w); return; }
  if (hs == 1 && vs == 2) { for (int i = 0; i < w; ++i) out[i] = (std::uint8_t)((3 * near_[i] + far_[i] + 2) >> 2); return; }
  if (hs == 2 && vs == 1) {
    if (w == 1) { out[0] = out[1] = near_[0]; return; }
    out[0] = near_[0];
    out[1] = (std::uint8_t)((near_[0] * 3 + near_[1] + 2) >> 2);
    int i;
    for (i = 1; i < w - 1; ++i) {
      const int n = 3 * near_[i] + 2;
      out[i * 2] = (std::uint8_t)((n + near_[i - 1]) >> 2);
      out[i * 2 + 1] = (std::uint8_t)((n + near_[i + 1]) >> 2);
    }
    out[i * 2] = (std::uint8_t)((near_[w - 2] * 3 + near_[w - 1] + 2) >> 2);
    out[i * 2 + 1] = near_[w - 1];
    return;
  }
  if (hs == 2 && vs == 2) {
    if (w == 1) { out[0] = out[1] = (std::uint8_t)((3 * near_[0] + far_[0] + 2) >> 2); return; }
    int t1 = 3 * near_[0] + far_[0];
    out[0] = (std::uint8_t)((t1 + 2) >> 2);
    for (int i = 1; i < w; ++i) {
      const int t0 = t1;
      t1 = 3 * near_[i] + far_[i];
      out[i * 2 - 1] = (std::uint8_t)((3 * t0 + t1 + 8) >> 4);
      out[i * 2] = (std::uint8_t)((3 * t1 + t0 + 8) >> 4);
    }
    out[w * 2 - 1] = (std::uint8_t)((t1 + 2) >> 2);
    return;
  }
  for (int i = 0; i < w; ++i)
    for (int j = 0; j < hs; ++j) out[i * hs + j] = near_[i];
}

inline std::uint8_t mul8(unsigned x, unsigned y) { const unsigned t = x * y + 128; return (std::uint8_t)((t + (t >> 8)) >> 8); }

inline void ycc_to_rgba(std::uint8_t* out, const std::uint8_t* y, const std::uint8_t* cb, const std::uint8_t* cr, int n) {
  auto fixed = [](float c) { return ((int)(c * 4096.0f + 0.5f)) << 8; };
  static const int k_r = fixed(1.40200f), k_g1 = fixed(0.71414f), k_g2 = fixed(0.34414f), k_b = fixed(1.77200f);
  for (int i = 0; i < n; ++i) {
    const int yf = (y[i] << 20) + (1 << 19), r_ = cr[i] - 128, b_ = cb[i] - 128;
    const int r = (yf + r_ * k_r) >> 20;
    const int g = (int)((unsigned)(yf + r_ * -k_g1) + ((unsigned)(b_ * -k_g2) & 0xffff0000u)) >> 20;
    const int b = (yf + b_ * k_b) >> 20;
    out[i * 4] = clamp8(r); out[i * 4 + 1] = clamp8(g); out[i * 4 + 2] = clamp8(b); out[i * 4 + 3] = 255;
  }
}

}  // namespace jpeg_detail

inline bool is_jpeg(const std::uint8_t* d, std::size_t n) { return n >= 3 && d[0] == 0xFF && d[1] == 0xD8 && d[2] == 0xFF; }

// Decodes a JPEG file image into w·h·4 bytes (R, G, B, 255; row 0 on top).  Throws std::runtime_error.
inline std::vector<std::uint8_t> decode_jpeg(const std::uint8_t* d, std::size_t n, int& w_out, int& h_out) {
  using namespace jpeg_detail;
  if (!is_jpeg(d, n)) throw std::runtime_error("JPEG: no SOI marker");
  const std::uint8_t* zz = zigzag();
  std::uint16_t qt[4][64] = {};
  HuffTable hdc[4], hac[4];
  std::vector<Component> comp;
  int W = 0, H = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0, restart_interval = 0;
  bool progressive = false, have_frame = false, jfif = false;
  int adobe_transform = -1, rgb_ids = 0;
  BitReader br(d, n);
  br.pos = 2;
  auto need = [&](std::size_t p, std::size_t k) { if (p + k > n) throw std::runtime_error("JPEG: truncated segment"); };
  auto be16 = [&](std::size_t p) { return (int)((d[p] << 8) | d[p + 1]); };

  int m = br.next_marker();
  for (;;) {
    if (m < 0) throw std::runtime_error(have_frame ? "JPEG: missing EOI marker" : "JPEG: no frame header");
    if (m == 0xD9) break;
    if (m >= 0xD0 && m <= 0xD7) { m = br.next_marker(); continue; }   // stray restart marker between segments
    need(br.pos, 2);
    const int L = be16(br.pos);
    if (L < 2) throw std::runtime_error("JPEG: bad segment length");
    need(br.pos, (std::size_t)L);
    const std::size_t s = br.pos + 2, e = br.pos + (std::size_t)L;
    if (m == 0xDB) {                                                     // DQT
      std::size_t p = s;
      while (p < e) {
        const int pq = d[p] >> 4, tq = d[p] & 15; ++p;
        if (pq > 1 || tq > 3) throw std::runtime_error("JPEG: bad DQT");
        if (p + (pq ? 128u : 64u) > e) throw std::runtime_error("JPEG: bad DQT length");
        for (int i = 0; i < 64; ++i) { qt[tq][zz[i]] = (std::uint16_t)(pq ? be16(p) : d[p]); p += pq ? 2 : 1; }
      }
    } else if (m == 0xC4) {                                              // DHT
      std::size_t p = s;
      while (p < e) {
        if (p + 17 > e) throw std::runtime_error("JPEG: bad DHT length");
        const int tc = d[p] >> 4, th = d[p] & 15; ++p;
        if (tc > 1 || th > 3) throw std::runtime_error("JPEG: bad DHT header");
        int counts[16], total = 0;
        for (int i = 0; i < 16; ++i) { counts[i] = d[p + (std::size_t)i]; total += counts[i]; }
        p += 16;
        if (total > 256 || p + (std::size_t)total > e) throw std::runtime_error("JPEG: bad DHT header");
        (tc ? hac[th] : hdc[th]).build(counts, d + p, total);
        p += (std::size_t)total;
      }
    } else if (m == 0xDD) {                                              // DRI
      if (L != 4) throw std::runtime_error("JPEG: bad DRI length");
      restart_interval = be16(s);
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {                    // SOF0/1/2
      if (have_frame) throw std::runtime_error("JPEG: more than one frame");
      if (L < 11) throw std::runtime_error("JPEG: bad SOF length");
      if (d[s] != 8) throw std::runtime_error("JPEG: only 8-bit samples are supported");
      H = be16(s + 1); W = be16(s + 3);
      const int nc = d[s + 5];
      if (H == 0 || W == 0) throw std::runtime_error("JPEG: zero image dimension");
      if (nc != 1 && nc != 3 && nc != 4) throw std::runtime_error("JPEG: bad component count");
      if (L != 8 + 3 * nc) throw std::runtime_error("JPEG: bad SOF length");
      comp.resize((std::size_t)nc);
      for (int i = 0; i < nc; ++i) {
        Component& c = comp[(std::size_t)i];
        c.id = d[s + 6 + 3 * (std::size_t)i]; c.h = d[s + 7 + 3 * (std::size_t)i] >> 4; c.v = d[s + 7 + 3 * (std::size_t)i] & 15; c.tq = d[s + 8 + 3 * (std::size_t)i];
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) throw std::runtime_error("JPEG: bad sampling factor or table id");
        if (nc == 3 && c.id == "RGB"[i]) ++rgb_ids;
        if (c.h > hmax) hmax = c.h;
        if (c.v > vmax) vmax = c.v;
      }
      for (const Component& c : comp) if (hmax % c.h || vmax % c.v) throw std::runtime_error("JPEG: fractional sampling ratios are not supported");
      if ((std::uint64_t)W * (std::uint64_t)H > (1ull << 28)) throw std::runtime_error("JPEG: image too large");
      mcux = (W + hmax * 8 - 1) / (hmax * 8); mcuy = (H + vmax * 8 - 1) / (vmax * 8);
      for (Component& c : comp) {
        c.x = (W * c.h + hmax - 1) / hmax; c.y = (H * c.v + vmax - 1) / vmax;
        c.bw = mcux * c.h; c.bh = mcuy * c.v;
        c.coef.assign((std::size_t)c.bw * c.bh * 64, 0);
      }
      progressive = m == 0xC2; have_frame = true;
    } else if (m == 0xDA) {                                              // SOS + entropy-coded data
      if (!have_frame) throw std::runtime_error("JPEG: scan before the frame header");
      const int ns = d[s];
      if (ns < 1 || ns > 4 || ns > (int)comp.size() || L != 6 + 2 * ns) throw std::runtime_error("JPEG: bad SOS");
      int order[4];
      for (int i = 0; i < ns; ++i) {
        const int id = d[s + 1 + 2 * (std::size_t)i], tab = d[s + 2 + 2 * (std::size_t)i];
        int which = -1;
        for (std::size_t k = 0; k < comp.size(); ++k) if (comp[k].id == id) { which = (int)k; break; }
        if (which < 0 || (tab >> 4) > 3 || (tab & 15) > 3) throw std::runtime_error("JPEG: bad SOS component");
        comp[(std::size_t)which].td = tab >> 4; comp[(std::size_t)which].ta = tab & 15; order[i] = which;
      }
      int ss = d[s + 1 + 2 * (std::size_t)ns], se = d[s + 2 + 2 * (std::size_t)ns];
      const int ah = d[s + 3 + 2 * (std::size_t)ns] >> 4, al = d[s + 3 + 2 * (std::size_t)ns] & 15;
      if (progressive) {
        if (ss > 63 || se > 63 || ss > se || ah > 13 || al > 13) throw std::runtime_error("JPEG: bad SOS");
        if (ss == 0 && se != 0) throw std::runtime_error("JPEG: a progressive scan cannot mix DC and AC");
        if (ss > 0 && ns != 1) throw std::runtime_error("JPEG: AC scans hold one component");
      } else {
        if (ss != 0 || ah != 0 || al != 0) throw std::runtime_error("JPEG: bad SOS");
        se = 63;
      }
      br.pos = e;
      br.reset();
      for (Component& c : comp) c.dc_pred = 0;
      int eob_run = 0, todo = restart_interval ? restart_interval : 0x7fffffff;

      auto block = [&](Component& c, std::int16_t* blk) {
        if (!progressive) {                                             // sequential: DC difference + run/size coded AC
          const int t = br.decode(hdc[c.td]);
          if (t > 15) throw std::runtime_error("JPEG: bad DC category");
          c.dc_pred += br.receive_extend(t);
          blk[0] = (std::int16_t)c.dc_pred;
          for (int k = 1; k < 64;) {
            const int rs = br.decode(hac[c.ta]), r = rs >> 4, sz = rs & 15;
            if (sz == 0) { if (rs != 0xF0) break; k += 16; continue; }
            k += r;
            if (k > 63) throw std::runtime_error("JPEG: AC run past the block");
            blk[zz[k++]] = (std::int16_t)br.receive_extend(sz);
          }
        } else if (ss == 0) {                                           // progressive DC: first pass or one refinement bit
          if (ah == 0) {
            const int t = br.decode(hdc[c.td]);
            if (t > 15) throw std::runtime_error("JPEG: bad DC category");
            c.dc_pred += br.receive_extend(t);
            blk[0] = (std::int16_t)(c.dc_pred * (1 << al));
          } else if (br.bit()) blk[0] = (std::int16_t)(blk[0] + (1 << al));
        } else if (ah == 0) {                                           // progressive AC, first pass (G.1.2.2)
          if (eob_run) { --eob_run; return; }
          for (int k = ss; k <= se;) {
            const int rs = br.decode(hac[c.ta]), r = rs >> 4, sz = rs & 15;
            if (sz == 0) {
              if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += (int)br.bits(r); break; }
              k += 16; continue;
            }
            k += r;
            if (k > 63) throw std::runtime_error("JPEG: AC run past the block");
            blk[zz[k++]] = (std::int16_t)(br.receive_extend(sz) * (1 << al));
          }
        } else {                                                        // progressive AC, refinement (G.1.2.3)
          const std::int16_t bit = (std::int16_t)(1 << al);
          auto refine = [&](std::int16_t& p) {
            if (br.bit() && (p & bit) == 0) p = (std::int16_t)(p > 0 ? p + bit : p - bit);
          };
          if (eob_run) {
            --eob_run;
            for (int k = ss; k <= se; ++k) { std::int16_t& p = blk[zz[k]]; if (p != 0) refine(p); }
            return;
          }
          for (int k = ss; k <= se;) {
            const int rs = br.decode(hac[c.ta]);
            int r = rs >> 4, sz = rs & 15, val = 0;
            if (sz == 0) {
              if (r < 15) { eob_run = (1 << r) - 1; if (r) eob_run += (int)br.bits(r); r = 64; }
            } else {
              if (sz != 1) throw std::runtime_error("JPEG: bad refinement code");
              val = br.bit() ? bit : -bit;
            }
            while (k <= se) {
              std::int16_t& p = blk[zz[k++]];
              if (p != 0) refine(p);
              else { if (r == 0) { p = (std::int16_t)val; break; } --r; }
            }
          }
        }
      };
      auto restart_due = [&]() -> bool {      // true: the scan ends here (no restart marker where one is due)
        if (--todo > 0) return false;
        const int mk = br.next_marker();
        if (mk < 0xD0 || mk > 0xD7) { br.marker = mk; return true; }
        for (Component& c : comp) c.dc_pred = 0;
        eob_run = 0; todo = restart_interval ? restart_interval : 0x7fffffff;
        return false;
      };
      bool ended = false;
      if (ns == 1) {                          // non-interleaved: the component's own blocks in raster order
        Component& c = comp[(std::size_t)order[0]];
        const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
        for (int j = 0; j < bh && !ended; ++j)
          for (int i = 0; i < bw && !ended; ++i) {
            block(c, &c.coef[((std::size_t)j * c.bw + (std::size_t)i) * 64]);
            ended = restart_due();
          }
      } else {
        for (int j = 0; j < mcuy && !ended; ++j)
          for (int i = 0; i < mcux && !ended; ++i) {
            for (int k = 0; k < ns; ++k) {
              Component& c = comp[(std::size_t)order[k]];
              for (int y = 0; y < c.v; ++y)
                for (int x = 0; x < c.h; ++x) block(c, &c.coef[((std::size_t)(j * c.v + y) * c.bw + (std::size_t)(i * c.h + x)) * 64]);
            }
            ended = restart_due();
          }
      }
      m = br.next_marker();
      while (m >= 0xD0 && m <= 0xD7) m = br.next_marker();
      continue;
    } else if (m == 0xE0 && L >= 7 && !std::memcmp(d + s, "JFIF\0", 5)) jfif = true;
    else if (m == 0xEE && L >= 14 && !std::memcmp(d + s, "Adobe\0", 6)) adobe_transform = d[s + 11];
    else if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE || m == 0xDC) { /* APPn, COM, DNL: skipped */ }
    else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) throw std::runtime_error("JPEG: lossless, hierarchical and arithmetic-coded processes are not supported");
    else throw std::runtime_error("JPEG: unknown marker");
    br.pos = e;
    m = br.next_marker();
  }
  if (!have_frame) throw std::runtime_error("JPEG: no frame header");

  // dequantise (16-bit multiply) + inverse DCT of every block → sample planes
  for (Component& c : comp) {
    c.plane.assign((std::size_t)c.bw * 8 * c.bh * 8, 0);
    const std::uint16_t* q = qt[c.tq];
    for (int j = 0; j < c.bh; ++j)
      for (int i = 0; i < c.bw; ++i) {
        std::int16_t* blk = &c.coef[((std::size_t)j * c.bw + (std::size_t)i) * 64];
        for (int k = 0; k < 64; ++k) blk[k] = (std::int16_t)(blk[k] * q[k]);
        idct_block(blk, &c.plane[((std::size_t)j * 8 * c.bw + (std::size_t)i) * 8], c.bw * 8);
      }
    c.coef.clear(); c.coef.shrink_to_fit();
  }

  // upsample row by row, convert to RGBA
  const int nc = (int)comp.size();
  const bool is_rgb = nc == 3 && (rgb_ids == 3 || (adobe_transform == 0 && !jfif));
  struct Walk { int hs, vs, ystep, ypos, w_lo; const std::uint8_t *line0, *line1; std::vector<std::uint8_t> buf; };
  std::vector<Walk> wk((std::size_t)nc);
  for (int k = 0; k < nc; ++k) {
    Walk& r = wk[(std::size_t)k];
    r.hs = hmax / comp[(std::size_t)k].h; r.vs = vmax / comp[(std::size_t)k].v;
    r.ystep = r.vs >> 1; r.ypos = 0; r.w_lo = (W + r.hs - 1) / r.hs;
    r.line0 = r.line1 = comp[(std::size_t)k].plane.data();
    r.buf.assign((std::size_t)W + 8 + (std::size_t)r.hs * 2, 0);
  }
  std::vector<std::uint8_t> out((std::size_t)W * H * 4);
  for (int j = 0; j < H; ++j) {
    const std::uint8_t* row[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < nc; ++k) {
      Walk& r = wk[(std::size_t)k];
      const bool bottom = r.ystep >= (r.vs >> 1);
      upsample_row(r.buf.data(), bottom ? r.line1 : r.line0, bottom ? r.line0 : r.line1, r.w_lo, r.hs, r.vs);
      row[k] = r.buf.data();
      if (++r.ystep >= r.vs) {
        r.ystep = 0; r.line0 = r.line1;
        if (++r.ypos < comp[(std::size_t)k].y) r.line1 += (std::size_t)comp[(std::size_t)k].bw * 8;
      }
    }
    std::uint8_t* o = &out[(std::size_t)j * W * 4];
    if (nc == 1) {
      for (int i = 0; i < W; ++i) { o[i * 4] = o[i * 4 + 1] = o[i * 4 + 2] = row[0][i]; o[i * 4 + 3] = 255; }
    } else if (nc == 3) {
      if (is_rgb) for (int i = 0; i < W; ++i) { o[i * 4] = row[0][i]; o[i * 4 + 1] = row[1][i]; o[i * 4 + 2] = row[2][i]; o[i * 4 + 3] = 255; }
      else ycc_to_rgba(o, row[0], row[1], row[2], W);
    } else if (adobe_transform == 0) {                                   // CMYK
      for (int i = 0; i < W; ++i) { const unsigned k4 = row[3][i]; o[i * 4] = mul8(row[0][i], k4); o[i * 4 + 1] = mul8(row[1][i], k4); o[i * 4 + 2] = mul8(row[2][i], k4); o[i * 4 + 3] = 255; }
    } else {
      ycc_to_rgba(o, row[0], row[1], row[2], W);
      if (adobe_transform == 2)                                          // YCCK
        for (int i = 0; i < W; ++i) { const unsigned k4 = row[3][i]; o[i * 4] = mul8(255u - o[i * 4], k4); o[i * 4 + 1] = mul8(255u - o[i * 4 + 1], k4); o[i * 4 + 2] = mul8(255u - o[i * 4 + 2], k4); }
    }
  }
  w_out = W; h_out = H;
  return out;
}

}  // namespace pbr::image
