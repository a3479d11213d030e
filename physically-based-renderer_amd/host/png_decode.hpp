// png_decode.hpp — dependency-free PNG reader → RGBA8 (textures of a glTF asset).
//
// The reference decodes images with the vendored stb_image and forces 4 channels of 8 bits
// (src/pbr_engine/image/pbr/image/LoadImage.cpp:56-73, format RGBA8 UNORM :27); this is the same contract —
// any PNG colour type / bit depth in, w·h·4 bytes out, rows top to bottom — written from the PNG (ISO 15948)
// and DEFLATE (RFC 1950/1951) specifications.  16-bit samples keep their high byte, palette and grey images
// are expanded, tRNS gives alpha, Adam7 interlacing is undone.  CRCs of the critical chunks are verified.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pbr::image {

namespace detail {

// ---- RFC 1951 inflate -------------------------------------------------------------------------------
class Inflater {
public:
  Inflater(const std::uint8_t* data, std::size_t n) : in_(data), n_(n) {}

  std::vector<std::uint8_t> run(std::size_t size_hint) {
    std::vector<std::uint8_t> out;
    out.reserve(size_hint);
    for (bool last = false; !last;) {
      last = bits(1) != 0;
      switch (bits(2)) {
        case 0: stored(out); break;
        case 1: fixed_tables(); codes(out); break;
        case 2: dynamic_tables(); codes(out); break;
        default: throw std::runtime_error("PNG: bad deflate block type");
      }
    }
    return out;
  }
  std::size_t consumed() const { return pos_; }

private:
  // canonical Huffman code: count of codes per length + symbols ordered by (length, value)
  struct Huff { std::uint16_t count[16]; std::uint16_t symbol[288]; };

  const std::uint8_t* in_; std::size_t n_; std::size_t pos_ = 0;
  std::uint32_t hold_ = 0; int held_ = 0;
  Huff lit_{}, dist_{};

  std::uint32_t bits(int need) {
    while (held_ < need) {
      if (pos_ >= n_) throw std::runtime_error("PNG: deflate stream truncated");
      hold_ |= (std::uint32_t)in_[pos_++] << held_;
      held_ += 8;
    }
    const std::uint32_t v = hold_ & ((need == 32) ? 0xffffffffu : ((1u << need) - 1u));
    hold_ >>= need; held_ -= need;
    return v;
  }
  static void build(Huff& h, const std::uint8_t* lengths, int n) {
    std::memset(h.count, 0, sizeof h.count);
    for (int i = 0; i < n; ++i) h.count[lengths[i]]++;
    int left = 1;
    for (int len = 1; len < 16; ++len) {
      left = (left << 1) - h.count[len];
      if (left < 0) throw std::runtime_error("PNG: over-subscribed Huffman code");
    }
    std::uint16_t offs[16];
    offs[1] = 0;
    for (int len = 1; len < 15; ++len) offs[len + 1] = (std::uint16_t)(offs[len] + h.count[len]);
    for (int i = 0; i < n; ++i) if (lengths[i]) h.symbol[offs[lengths[i]]++] = (std::uint16_t)i;
  }
  int decode(const Huff& h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; ++len) {
      code |= (int)bits(1);
      const int cnt = h.count[len];
      if (code - cnt < first) return h.symbol[index + (code - first)];
      index += cnt; first += cnt; first <<= 1; code <<= 1;
    }
    throw std::runtime_error("PNG: invalid Huffman code");
  }
  void stored(std::vector<std::uint8_t>& out) {
    hold_ = 0; held_ = 0;                                     // skip to the byte boundary
    if (pos_ + 4 > n_) throw std::runtime_error("PNG: stored block truncated");
    const unsigned len = in_[pos_] | (in_[pos_ + 1] << 8), nlen = in_[pos_ + 2] | (in_[pos_ + 3] << 8);
    pos_ += 4;
    if ((len ^ 0xffffu) != nlen) throw std::runtime_error("PNG: stored block length check failed");
    if (pos_ + len > n_) throw std::runtime_error("PNG: stored block truncated");
    out.insert(out.end(), in_ + pos_, in_ + pos_ + len);
    pos_ += len;
  }
  void fixed_tables() {
    std::uint8_t l[288];
    for (int i = 0; i < 144; ++i) l[i] = 8;
    for (int i = 144; i < 256; ++i) l[i] = 9;
    for (int i = 256; i < 280; ++i) l[i] = 7;
    for (int i = 280; i < 288; ++i) l[i] = 8;
    build(lit_, l, 288);
    std::uint8_t d[30];
    for (int i = 0; i < 30; ++i) d[i] = 5;
    build(dist_, d, 30);
  }
  void dynamic_tables() {
    static const std::uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    const int nlen = (int)bits(5) + 257, ndist = (int)bits(5) + 1, ncode = (int)bits(4) + 4;
    if (nlen > 286 || ndist > 30) throw std::runtime_error("PNG: bad deflate code counts");
    std::uint8_t l[320];
    std::memset(l, 0, sizeof l);
    for (int i = 0; i < ncode; ++i) l[order[i]] = (std::uint8_t)bits(3);
    Huff cl{};
    build(cl, l, 19);
    std::uint8_t lengths[320];
    std::memset(lengths, 0, sizeof lengths);
    for (int i = 0; i < nlen + ndist;) {
      const int sym = decode(cl);
      if (sym < 16) { lengths[i++] = (std::uint8_t)sym; continue; }
      int rep; std::uint8_t val = 0;
      if (sym == 16) { if (i == 0) throw std::runtime_error("PNG: repeat without a previous length"); val = lengths[i - 1]; rep = 3 + (int)bits(2); }
      else if (sym == 17) rep = 3 + (int)bits(3);
      else rep = 11 + (int)bits(7);
      if (i + rep > nlen + ndist) throw std::runtime_error("PNG: code length repeat overruns");
      while (rep--) lengths[i++] = val;
    }
    if (lengths[256] == 0) throw std::runtime_error("PNG: no end-of-block code");
    build(lit_, lengths, nlen);
    build(dist_, lengths + nlen, ndist);
  }
  void codes(std::vector<std::uint8_t>& out) {
    static const std::uint16_t lbase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const std::uint8_t lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const std::uint16_t dbase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const std::uint8_t dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    for (;;) {
      int sym = decode(lit_);
      if (sym < 256) { out.push_back((std::uint8_t)sym); continue; }
      if (sym == 256) return;
      sym -= 257;
      if (sym >= 29) throw std::runtime_error("PNG: bad length symbol");
      const std::size_t len = lbase[sym] + bits(lext[sym]);
      const int ds = decode(dist_);
      if (ds >= 30) throw std::runtime_error("PNG: bad distance symbol");
      const std::size_t dist = dbase[ds] + bits(dext[ds]);
      if (dist > out.size()) throw std::runtime_error("PNG: distance reaches before the start of the data");
      const std::size_t from = out.size() - dist;
      for (std::size_t k = 0; k < len; ++k) out.push_back(out[from + k]);   // may overlap: byte by byte
    }
  }
};

inline std::uint32_t png_crc(const std::uint8_t* p, std::size_t n) {
  static std::uint32_t table[256];
  static bool ready = false;
  if (!ready) {
    for (std::uint32_t i = 0; i < 256; ++i) { std::uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    ready = true;
  }
  std::uint32_t c = 0xffffffffu;
  for (std::size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 255u] ^ (c >> 8);
  return c ^ 0xffffffffu;
}
inline std::uint32_t read_be32(const std::uint8_t* p) { return ((std::uint32_t)p[0] << 24) | ((std::uint32_t)p[1] << 16) | ((std::uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace detail

inline bool is_png(const std::uint8_t* d, std::size_t n) {
  static const std::uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  return n >= 8 && std::memcmp(d, sig, 8) == 0;
}

// Decodes a PNG file image into w·h·4 bytes (R, G, B, A; row 0 on top).  Throws std::runtime_error.
inline std::vector<std::uint8_t> decode_png(const std::uint8_t* d, std::size_t n, int& w, int& h) {
  using namespace detail;
  if (!is_png(d, n)) throw std::runtime_error("PNG: bad signature");
  std::size_t p = 8;
  std::uint32_t W = 0, H = 0; int depth = 0, ctype = -1, interlace = 0;
  std::vector<std::uint8_t> idat, plte, trns;
  bool have_ihdr = false, done = false;
  while (!done) {
    if (p + 12 > n) throw std::runtime_error("PNG: truncated chunk");
    const std::uint32_t len = read_be32(d + p);
    if (len > n - p - 12) throw std::runtime_error("PNG: chunk exceeds the file");
    const std::uint8_t* type = d + p + 4; const std::uint8_t* body = d + p + 8;
    const bool critical = (type[0] & 0x20) == 0;
    if (critical && png_crc(type, (std::size_t)len + 4) != read_be32(body + len)) throw std::runtime_error("PNG: CRC mismatch");
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) throw std::runtime_error("PNG: bad IHDR");
      W = read_be32(body); H = read_be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
      if (body[10] != 0 || body[11] != 0 || interlace > 1) throw std::runtime_error("PNG: unsupported compression / filter / interlace method");
      if (W == 0 || H == 0 || W > 32768u || H > 32768u) throw std::runtime_error("PNG: bad dimensions");
      have_ihdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
    else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
    else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if (!std::memcmp(type, "IEND", 4)) done = true;
    else if (critical) throw std::runtime_error("PNG: unknown critical chunk");
    p += (std::size_t)len + 12;
  }
  if (!have_ihdr || idat.empty()) throw std::runtime_error("PNG: missing IHDR or IDAT");
  int channels;
  switch (ctype) {
    case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break; case 4: channels = 2; break; case 6: channels = 4; break;
    default: throw std::runtime_error("PNG: bad colour type");
  }
  const bool depth_ok = (ctype == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                        (ctype == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                        ((ctype == 2 || ctype == 4 || ctype == 6) && (depth == 8 || depth == 16));
  if (!depth_ok) throw std::runtime_error("PNG: bit depth not allowed for the colour type");
  if (ctype == 3 && (plte.empty() || plte.size() % 3)) throw std::runtime_error("PNG: palette image without a valid PLTE");
  // zlib wrapper (RFC 1950): CM = 8, no preset dictionary
  if (idat.size() < 6 || (idat[0] & 15) != 8 || ((idat[0] << 8) | idat[1]) % 31 != 0 || (idat[1] & 0x20)) throw std::runtime_error("PNG: bad zlib header");
  const int bpp_bits = channels * depth;
  const std::size_t bpp = (std::size_t)(bpp_bits + 7) / 8;   // filter unit
  auto row_bytes = [&](std::uint32_t pw) { return ((std::size_t)pw * bpp_bits + 7) / 8; };
  struct Pass { std::uint32_t x0, y0, dx, dy; };
  static const Pass adam7[7] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
  static const Pass whole = {0, 0, 1, 1};
  std::size_t expect = 0;
  for (int k = 0; k < (interlace ? 7 : 1); ++k) {
    const Pass& ps = interlace ? adam7[k] : whole;
    const std::uint32_t pw = (W > ps.x0) ? (W - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = (H > ps.y0) ? (H - ps.y0 + ps.dy - 1) / ps.dy : 0;
    if (pw && ph) expect += (row_bytes(pw) + 1) * ph;
  }
  Inflater inf(idat.data() + 2, idat.size() - 2);
  std::vector<std::uint8_t> raw = inf.run(expect);
  if (raw.size() != expect) throw std::runtime_error("PNG: decompressed size does not match the header");
  {  // Adler-32 of the decompressed data
    std::uint32_t a = 1, b = 0;
    for (std::size_t i = 0; i < raw.size(); ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
    const std::size_t tail = 2 + inf.consumed();
    if (tail + 4 > idat.size() || read_be32(idat.data() + tail) != ((b << 16) | a)) throw std::runtime_error("PNG: Adler-32 mismatch");
  }
  std::vector<std::uint8_t> out((std::size_t)W * H * 4);
  auto sample = [&](const std::uint8_t* row, std::size_t index) -> unsigned {   // index-th `depth`-bit sample of a row
    if (depth == 8) return row[index];
    if (depth == 16) return ((unsigned)row[index * 2] << 8) | row[index * 2 + 1];
    const std::size_t bit = index * depth;
    return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
  };
  auto to8 = [&](unsigned v) -> std::uint8_t {
    switch (depth) { case 16: return (std::uint8_t)(v >> 8); case 8: return (std::uint8_t)v; case 4: return (std::uint8_t)(v * 17u); case 2: return (std::uint8_t)(v * 85u); default: return (std::uint8_t)(v * 255u); }
  };
  std::size_t rp = 0;
  std::vector<std::uint8_t> prev, cur;
  for (int k = 0; k < (interlace ? 7 : 1); ++k) {
    const Pass& ps = interlace ? adam7[k] : whole;
    const std::uint32_t pw = (W > ps.x0) ? (W - ps.x0 + ps.dx - 1) / ps.dx : 0, ph = (H > ps.y0) ? (H - ps.y0 + ps.dy - 1) / ps.dy : 0;
    if (!pw || !ph) continue;
    const std::size_t rb = row_bytes(pw);
    prev.assign(rb, 0); cur.assign(rb, 0);
    for (std::uint32_t y = 0; y < ph; ++y) {
      const int ft = raw[rp++];
      const std::uint8_t* src = &raw[rp];
      rp += rb;
      for (std::size_t i = 0; i < rb; ++i) {
        const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
        int v;
        switch (ft) {
          case 0: v = src[i]; break;
          case 1: v = src[i] + a; break;
          case 2: v = src[i] + b; break;
          case 3: v = src[i] + ((a + b) >> 1); break;
          case 4: v = src[i] + paeth(a, b, c); break;
          default: throw std::runtime_error("PNG: bad filter type");
        }
        cur[i] = (std::uint8_t)v;
      }
      std::uint8_t* orow = &out[((std::size_t)(ps.y0 + y * ps.dy) * W) * 4];
      for (std::uint32_t x = 0; x < pw; ++x) {
        std::uint8_t* px = orow + (std::size_t)(ps.x0 + x * ps.dx) * 4;
        switch (ctype) {
          case 0: {
            const unsigned g = sample(cur.data(), x);
            px[0] = px[1] = px[2] = to8(g);
            px[3] = (trns.size() >= 2 && g == (((unsigned)trns[0] << 8) | trns[1])) ? 0 : 255;
          } break;
          case 2: {
            const unsigned r = sample(cur.data(), x * 3u), g = sample(cur.data(), x * 3u + 1), b = sample(cur.data(), x * 3u + 2);
            px[0] = to8(r); px[1] = to8(g); px[2] = to8(b);
            px[3] = (trns.size() >= 6 && r == (((unsigned)trns[0] << 8) | trns[1]) && g == (((unsigned)trns[2] << 8) | trns[3]) && b == (((unsigned)trns[4] << 8) | trns[5])) ? 0 : 255;
          } break;
          case 3: {
            const unsigned i = sample(cur.data(), x);
            if ((std::size_t)i * 3 + 2 >= plte.size()) throw std::runtime_error("PNG: palette index out of range");
            px[0] = plte[i * 3]; px[1] = plte[i * 3 + 1]; px[2] = plte[i * 3 + 2];
            px[3] = i < trns.size() ? trns[i] : 255;
          } break;
          case 4: {
            px[0] = px[1] = px[2] = to8(sample(cur.data(), x * 2u));
            px[3] = to8(sample(cur.data(), x * 2u + 1));
          } break;
          default: {
            px[0] = to8(sample(cur.data(), x * 4u)); px[1] = to8(sample(cur.data(), x * 4u + 1));
            px[2] = to8(sample(cur.data(), x * 4u + 2)); px[3] = to8(sample(cur.data(), x * 4u + 3));
          } break;
        }
      }
      prev.swap(cur);
    }
  }
  w = (int)W; h = (int)H;
  return out;
}

}  // namespace pbr::image
