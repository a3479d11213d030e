// misc_decode.hpp — the other file images the reference's image path takes: Windows BMP, Truevision TGA, binary PGM / PPM, Photoshop PSD, GIF, Softimage PIC.
//
// image::loadImage2D (src/pbr_engine/image/pbr/image/LoadImage.cpp:56-73) hands whatever bytes a glTF image holds to
// stbi_load_from_memory(..., 4): besides PNG and JPEG (png_decode.hpp, jpeg_decode.hpp) that is BMP, GIF, PSD, PIC, PNM, HDR and TGA, tried
// in that order (src/pbr_engine/image/stb/stb_image.h, stbi__load_main).  This header restates six of them (Radiance-as-texture is in image_io.hpp) — the formats' own rules, plus
// the vendored decoder's choices where the formats leave room — so that the texels handed to ptc_add_texture_rgba8 are the reference's,
// byte for byte (tests/test_misc_images.py against oracle/_ref, the reference's stb_image translation unit compiled in place, and against
// fixtures generated from it):
//   BMP  core (12-byte), info (40 / 56), V4 / V5 headers; 1 / 4 / 8 bits with a palette, 16 / 24 / 32 bits direct or through bit fields
//        (each channel widened to 8 bits by bit replication); bottom-up or top-down rows; a 32-bit image whose alpha is 0 everywhere is opaque;
//        run-length and embedded PNG / JPEG compressions are refused, as there.
//   TGA  types 1 / 2 / 3 and their run-length forms 9 / 10 / 11; 8-bit grey, 16-bit grey + alpha, 15 / 16-bit colour (5-5-5, c * 255 / 31,
//        no alpha), 24 / 32-bit; colour maps of those entry sizes; the descriptor's bit 5 says top-down.  A run-length packet may cross rows.
//   PNM  "P5" / "P6" with "#" comments; maxval <= 255 gives 8-bit samples, <= 65535 16-bit ones — of which the reference keeps the byte at the
//        ODD offset (it reads the big-endian samples as native little-endian words and keeps their high half); so does this.
// A byte past the end of the file reads as 0 (stb's stbi__get8), so a BMP or a run-length TGA cut short decodes with a black tail there and
// here; where the reference reads whole rows at once (raw TGA rows, the PNM body) and a short file leaves them undefined or is refused, this
// decoder refuses.  Grey → (y, y, y, 255), grey + alpha → (y, y, y, a), RGB → (r, g, b, 255) (stbi__convert_format).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace pbr::image {
namespace misc_detail {
constexpr int kMaxDim = 1 << 24;        // STBI_MAX_DIMENSIONS

struct Bytes {                           // forward reader over the file; past the end every byte is 0
  const std::uint8_t* d; std::size_t n, at = 0;
  Bytes(const std::uint8_t* d_, std::size_t n_) : d(d_), n(n_) {}
  bool eof() const { return at >= n; }
  int u8() { return at < n ? d[at++] : 0; }
  int u16() { const int lo = u8(); return lo | (u8() << 8); }
  std::uint32_t u32() { const std::uint32_t lo = (std::uint32_t)u16(); return lo | ((std::uint32_t)u16() << 16); }
  void skip(long long k) {               // a negative distance jumps to the end; a positive one may pass it (everything there reads as 0)
    if (k < 0) { at = n; return; }
    at += (std::size_t)k;
  }
};

inline void check_area(long long w, long long h, long long bytes_per_pixel) {
  if (w < 0 || h < 0 || (w && h && bytes_per_pixel * w > 0x7fffffffLL / h)) throw std::runtime_error("image too large");
  if (w * h > (1LL << 28)) throw std::runtime_error("image too large");          // (as decode_hdr: nothing a texture can be; the reference would try to allocate it)
}
inline int top_bit(std::uint32_t v) { int k = -1; while (v) { ++k; v >>= 1; } return k; }
inline int bit_count(std::uint32_t v) { int k = 0; while (v) { k += (int)(v & 1u); v >>= 1; } return k; }
// the field `v & mask` of `bits` bits whose top bit sits at `top`, widened to 8 bits by repeating its bit pattern
inline int widen_field(std::uint32_t v, int top, int bits) {
  static const std::uint32_t mul[9] = {0, 0xff, 0x55, 0x49, 0x11, 0x21, 0x41, 0x81, 0x01};
  static const int shr[9] = {0, 0, 0, 1, 0, 2, 4, 6, 0};
  const int shift = top - 7;
  v = shift < 0 ? v << -shift : v >> shift;          // the field's top bit now at bit 7
  v >>= 8 - bits;
  return (int)(v * mul[bits]) >> shr[bits];
}
}  // namespace misc_detail

inline bool is_bmp(const std::uint8_t* d, std::size_t n) {
  misc_detail::Bytes r(d, n);
  if (r.u8() != 'B' || r.u8() != 'M') return false;
  r.skip(12);
  const std::uint32_t hs = r.u32();
  return hs == 12 || hs == 40 || hs == 56 || hs == 108 || hs == 124;
}

inline std::vector<std::uint8_t> decode_bmp(const std::uint8_t* d, std::size_t n, int& w_out, int& h_out) {
  using namespace misc_detail;
  Bytes r(d, n);
  if (r.u8() != 'B' || r.u8() != 'M') throw std::runtime_error("not a BMP image");
  r.skip(8);
  const int offset = (int)r.u32();
  const int hs = (int)r.u32();
  if (offset < 0) throw std::runtime_error("BMP: bad pixel offset");
  if (hs != 12 && hs != 40 && hs != 56 && hs != 108 && hs != 124) throw std::runtime_error("BMP: unknown header");
  int w, hraw;
  if (hs == 12) { w = r.u16(); hraw = r.u16(); } else { w = (int)r.u32(); hraw = (int)r.u32(); }
  if (r.u16() != 1) throw std::runtime_error("BMP: bad plane count");
  const int bpp = r.u16();
  std::uint32_t mr = 0, mg = 0, mb = 0, ma = 0;
  std::uint32_t alpha_seen = 255;        // OR of every alpha read; starts non-zero unless the default 32-bit layout is in force
  int header_extra = 14;                 // bytes in front of the info header, plus bit-field masks that follow a 40 / 56-byte one
  auto default_masks = [&]() {
    if (bpp == 16) { mr = 31u << 10; mg = 31u << 5; mb = 31u; }
    else if (bpp == 32) { mr = 0xffu << 16; mg = 0xffu << 8; mb = 0xffu; ma = 0xffu << 24; alpha_seen = 0; }
    else mr = mg = mb = ma = 0;
  };
  if (hs != 12) {
    const int compress = (int)r.u32();
    if (compress == 1 || compress == 2) throw std::runtime_error("BMP: run-length compression is not supported");
    if (compress >= 4) throw std::runtime_error("BMP: embedded JPEG / PNG is not supported");
    if (compress == 3 && bpp != 16 && bpp != 32) throw std::runtime_error("BMP: bit fields need 16 or 32 bits per pixel");
    r.skip(20);                          // image size, resolutions, colour counts
    if (hs == 40 || hs == 56) {
      if (hs == 56) r.skip(16);
      if (bpp == 16 || bpp == 32) {
        if (compress == 0) default_masks();
        else if (compress == 3) {
          mr = r.u32(); mg = r.u32(); mb = r.u32();
          header_extra += 12;
          if (mr == mg && mg == mb) throw std::runtime_error("BMP: bad bit fields");
        } else throw std::runtime_error("BMP: bad compression");
      }
    } else {
      mr = r.u32(); mg = r.u32(); mb = r.u32(); ma = r.u32();
      if (compress == 0) default_masks();            // the header's masks count in bit-field mode only
      else if (compress != 3) { /* negative values: the masks stay as read */ }
      r.skip(4 + 48);                    // colour space and its parameters
      if (hs == 124) r.skip(16);
    }
  }
  const bool bottom_up = hraw > 0;
  if (hraw == (-2147483647 - 1)) throw std::runtime_error("BMP: image too large");
  const int h = std::abs(hraw);
  if (h > kMaxDim || w > kMaxDim || w < 0) throw std::runtime_error("BMP: image too large");
  int psize = 0;
  if (hs == 12) { if (bpp < 24) psize = (offset - header_extra - 24) / 3; }
  else if (bpp < 16) psize = (offset - header_extra - hs) >> 2;
  if (psize == 0) {                      // no palette: the pixels start at `offset`, which may not lie inside the header nor far behind it
    const long long here = (long long)r.at;
    if (here <= 0 || here > 1024) throw std::runtime_error("BMP: bad header");
    if (offset < here || offset - here > 1024) throw std::runtime_error("BMP: bad pixel offset");
    r.skip(offset - here);
  }
  check_area(w, h, 4);
  std::vector<std::uint8_t> out((std::size_t)w * (std::size_t)h * 4);
  std::size_t z = 0;
  if (bpp < 16) {
    if (psize == 0 || psize > 256) throw std::runtime_error("BMP: bad palette");
    std::uint8_t pal[256][3] = {};
    for (int i = 0; i < psize; ++i) {
      pal[i][2] = (std::uint8_t)r.u8(); pal[i][1] = (std::uint8_t)r.u8(); pal[i][0] = (std::uint8_t)r.u8();
      if (hs != 12) r.u8();
    }
    r.skip((long long)offset - header_extra - hs - (long long)psize * (hs == 12 ? 3 : 4));
    int row_bytes;
    if (bpp == 1) row_bytes = (w + 7) >> 3;
    else if (bpp == 4) row_bytes = (w + 1) >> 1;
    else if (bpp == 8) row_bytes = w;
    else throw std::runtime_error("BMP: bad bits per pixel");
    const int pad = (-row_bytes) & 3;
    auto put = [&](int idx) { out[z++] = pal[idx][0]; out[z++] = pal[idx][1]; out[z++] = pal[idx][2]; out[z++] = 255; };
    for (int j = 0; j < h; ++j) {
      if (bpp == 1) {
        // a byte is fetched for the row's first pixel and after every eighth one but the last: a row of 8k pixels takes k bytes, and an empty row one
        int bit = 7, v = r.u8();
        for (int i = 0; i < w; ++i) {
          put((v >> bit) & 1);
          if (i + 1 == w) break;
          if (--bit < 0) { bit = 7; v = r.u8(); }
        }
      } else {
        for (int i = 0; i < w; i += 2) {
          int v = r.u8(), v2 = 0;
          if (bpp == 4) { v2 = v & 15; v >>= 4; }
          put(v);
          if (i + 1 == w) break;
          put(bpp == 8 ? r.u8() : v2);
        }
      }
      r.skip(pad);
    }
  } else {
    r.skip((long long)offset - header_extra - hs);
    const int row_bytes = bpp == 24 ? 3 * w : (bpp == 16 ? 2 * w : 0);
    const int pad = (-row_bytes) & 3;
    int easy = 0;
    if (bpp == 24) easy = 1;
    else if (bpp == 32 && mb == 0xffu && mg == 0xff00u && mr == 0x00ff0000u && ma == 0xff000000u) easy = 2;
    int tr = 0, tg = 0, tb = 0, ta = 0, cr = 0, cg = 0, cb = 0, ca = 0;
    if (!easy) {
      if (!mr || !mg || !mb) throw std::runtime_error("BMP: bad bit fields");
      tr = top_bit(mr); cr = bit_count(mr); tg = top_bit(mg); cg = bit_count(mg); tb = top_bit(mb); cb = bit_count(mb); ta = top_bit(ma); ca = bit_count(ma);
      if (cr > 8 || cg > 8 || cb > 8 || ca > 8) throw std::runtime_error("BMP: bad bit fields");
    }
    for (int j = 0; j < h; ++j) {
      for (int i = 0; i < w; ++i) {
        if (easy) {
          const int b = r.u8(), g = r.u8(), rr = r.u8();
          const int a = easy == 2 ? r.u8() : 255;
          alpha_seen |= (std::uint32_t)a;
          out[z++] = (std::uint8_t)rr; out[z++] = (std::uint8_t)g; out[z++] = (std::uint8_t)b; out[z++] = (std::uint8_t)a;
        } else {
          const std::uint32_t v = bpp == 16 ? (std::uint32_t)r.u16() : r.u32();
          out[z++] = (std::uint8_t)widen_field(v & mr, tr, cr);
          out[z++] = (std::uint8_t)widen_field(v & mg, tg, cg);
          out[z++] = (std::uint8_t)widen_field(v & mb, tb, cb);
          const int a = ma ? widen_field(v & ma, ta, ca) : 255;
          alpha_seen |= (std::uint32_t)a;
          out[z++] = (std::uint8_t)a;
        }
      }
      r.skip(pad);
    }
  }
  if (alpha_seen == 0)                     // a 32-bit image that left its alpha byte empty is opaque
    for (std::size_t i = 3; i < out.size(); i += 4) out[i] = 255;
  if (bottom_up)
    for (int j = 0; j < h / 2; ++j) {
      std::uint8_t* a = &out[(std::size_t)j * (std::size_t)w * 4];
      std::uint8_t* b = &out[(std::size_t)(h - 1 - j) * (std::size_t)w * 4];
      for (std::size_t i = 0; i < (std::size_t)w * 4; ++i) { const std::uint8_t t = a[i]; a[i] = b[i]; b[i] = t; }
    }
  w_out = w; h_out = h;
  return out;
}

inline bool is_pnm(const std::uint8_t* d, std::size_t n) { return n >= 2 && d[0] == 'P' && (d[1] == '5' || d[1] == '6'); }

inline std::vector<std::uint8_t> decode_pnm(const std::uint8_t* d, std::size_t n, int& w_out, int& h_out) {
  using namespace misc_detail;
  if (!is_pnm(d, n)) throw std::runtime_error("not a binary PGM / PPM image");
  const int comp = d[1] == '6' ? 3 : 1;
  Bytes r(d, n);
  r.skip(2);
  char c = (char)r.u8();
  auto space = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\v' || ch == '\f' || ch == '\r'; };
  auto skip_blank = [&]() {                // white space and "#" comments up to the end of their line
    for (;;) {
      while (!r.eof() && space(c)) c = (char)r.u8();
      if (r.eof() || c != '#') break;
      while (!r.eof() && c != '\n' && c != '\r') c = (char)r.u8();
    }
  };
  auto number = [&]() {
    int v = 0;
    while (!r.eof() && c >= '0' && c <= '9') {
      v = v * 10 + (c - '0');
      c = (char)r.u8();
      if (v > 214748364 || (v == 214748364 && c > '7')) throw std::runtime_error("PNM: number too large");
    }
    return v;
  };
  skip_blank();
  const int w = number();
  if (w == 0) throw std::runtime_error("PNM: bad width");
  skip_blank();
  const int h = number();
  if (h == 0) throw std::runtime_error("PNM: bad height");
  skip_blank();
  const int maxv = number();               // the single byte behind it (normally a newline) has been consumed into c: the samples follow
  if (maxv > 65535) throw std::runtime_error("PNM: maximum value above 65535");
  const int bytes = maxv > 255 ? 2 : 1;
  if (h > kMaxDim || w > kMaxDim) throw std::runtime_error("PNM: image too large");
  check_area(w, h, (long long)comp * bytes);
  check_area(w, h, 4);
  const std::size_t need = (std::size_t)w * (std::size_t)h * (std::size_t)comp * (std::size_t)bytes;
  if (r.at > n || n - r.at < need) throw std::runtime_error("PNM: file truncated");
  const std::uint8_t* s = d + r.at;
  std::vector<std::uint8_t> out((std::size_t)w * (std::size_t)h * 4);
  for (std::size_t p = 0; p < (std::size_t)w * (std::size_t)h; ++p) {
    std::uint8_t v[3];
    for (int k = 0; k < comp; ++k) v[k] = bytes == 1 ? s[p * (std::size_t)comp + (std::size_t)k] : s[(p * (std::size_t)comp + (std::size_t)k) * 2 + 1];
    if (comp == 1) v[1] = v[2] = v[0];
    out[p * 4] = v[0]; out[p * 4 + 1] = v[1]; out[p * 4 + 2] = v[2]; out[p * 4 + 3] = 255;
  }
  w_out = w; h_out = h;
  return out;
}

// TGA has no signature: the header's fields have to be plausible (the reference tries it last, for that reason)
inline bool is_tga(const std::uint8_t* d, std::size_t n) {
  misc_detail::Bytes r(d, n);
  r.u8();
  const int map_type = r.u8();
  if (map_type > 1) return false;
  const int type = r.u8();
  if (map_type == 1) {
    if (type != 1 && type != 9) return false;
    r.skip(4);
    const int eb = r.u8();
    if (eb != 8 && eb != 15 && eb != 16 && eb != 24 && eb != 32) return false;
    r.skip(4);
  } else {
    if (type != 2 && type != 3 && type != 10 && type != 11) return false;
    r.skip(9);
  }
  if (r.u16() < 1 || r.u16() < 1) return false;
  const int bpp = r.u8();
  if (map_type == 1 && bpp != 8 && bpp != 16) return false;
  return bpp == 8 || bpp == 15 || bpp == 16 || bpp == 24 || bpp == 32;
}

inline std::vector<std::uint8_t> decode_tga(const std::uint8_t* d, std::size_t n, int& w_out, int& h_out) {
  using namespace misc_detail;
  if (!is_tga(d, n)) throw std::runtime_error("not a TGA image");
  Bytes r(d, n);
  const int id_len = r.u8();
  const bool mapped = r.u8() != 0;
  int type = r.u8();
  const int map_first = r.u16(), map_len = r.u16(), map_bits = r.u8();
  r.skip(4);
  const int w = r.u16(), h = r.u16(), bpp = r.u8();
  const bool bottom_up = ((r.u8() >> 5) & 1) == 0;
  bool rle = false;
  if (type >= 8) { type -= 8; rle = true; }
  // channels of a pixel (or of a colour-map entry); 15 / 16-bit colour is 5-5-5 without alpha
  bool c555 = false;
  auto channels = [&](int bits, bool grey) {
    switch (bits) {
      case 8: return 1;
      case 16: if (grey) return 2; c555 = true; return 3;
      case 15: c555 = true; return 3;
      case 24: return 3;
      case 32: return 4;
      default: return 0;
    }
  };
  const int comp = mapped ? channels(map_bits, false) : channels(bpp, type == 3);
  if (!comp) throw std::runtime_error("TGA: bad pixel format");
  check_area(w, h, comp);
  check_area(w, h, 4);
  r.skip(id_len);
  const std::size_t np = (std::size_t)w * (std::size_t)h;
  std::vector<std::uint8_t> px(np * (std::size_t)comp);
  auto read555 = [&](std::uint8_t* o) {
    const int v = r.u16();
    o[0] = (std::uint8_t)((((v >> 10) & 31) * 255) / 31); o[1] = (std::uint8_t)((((v >> 5) & 31) * 255) / 31); o[2] = (std::uint8_t)(((v & 31) * 255) / 31);
  };
  if (!mapped && !rle && !c555) {          // whole rows, in file order
    if (r.at > n || n - r.at < px.size()) throw std::runtime_error("TGA: file truncated");
    for (int i = 0; i < h; ++i) {
      const int row = bottom_up ? h - 1 - i : i;
      for (std::size_t k = 0; k < (std::size_t)w * (std::size_t)comp; ++k) px[(std::size_t)row * (std::size_t)w * (std::size_t)comp + k] = d[r.at++];
    }
  } else {
    std::vector<std::uint8_t> pal;
    if (mapped) {
      if (map_len == 0) throw std::runtime_error("TGA: empty colour map");
      r.skip(map_first);
      pal.resize((std::size_t)map_len * (std::size_t)comp);
      if (c555) for (int i = 0; i < map_len; ++i) read555(&pal[(std::size_t)i * 3]);
      else {
        if (r.at > n || n - r.at < pal.size()) throw std::runtime_error("TGA: colour map truncated");
        for (std::uint8_t& b : pal) b = d[r.at++];
      }
    }
    std::uint8_t cur[4] = {0, 0, 0, 0};
    int run = 0; bool repeat = false;
    for (std::size_t i = 0; i < np; ++i) {
      bool fetch = true;
      if (rle) {
        if (run == 0) { const int cmd = r.u8(); run = 1 + (cmd & 127); repeat = (cmd >> 7) != 0; }
        else if (repeat) fetch = false;
      }
      if (fetch) {
        if (mapped) {
          int idx = bpp == 8 ? r.u8() : r.u16();
          if (idx >= map_len) idx = 0;
          for (int k = 0; k < comp; ++k) cur[k] = pal[(std::size_t)idx * (std::size_t)comp + (std::size_t)k];
        } else if (c555) read555(cur);
        else for (int k = 0; k < comp; ++k) cur[k] = (std::uint8_t)r.u8();
      }
      for (int k = 0; k < comp; ++k) px[i * (std::size_t)comp + (std::size_t)k] = cur[k];
      --run;
    }
    if (bottom_up)
      for (int j = 0; j * 2 < h; ++j) {
        std::uint8_t* a = &px[(std::size_t)j * (std::size_t)w * (std::size_t)comp];
        std::uint8_t* b = &px[(std::size_t)(h - 1 - j) * (std::size_t)w * (std::size_t)comp];
        for (std::size_t i = 0; i < (std::size_t)w * (std::size_t)comp; ++i) { const std::uint8_t t = a[i]; a[i] = b[i]; b[i] = t; }
      }
  }
  std::vector<std::uint8_t> out(np * 4);
  for (std::size_t p = 0; p < np; ++p) {
    const std::uint8_t* s = &px[p * (std::size_t)comp];
    std::uint8_t* o = &out[p * 4];
    if (comp == 1) { o[0] = o[1] = o[2] = s[0]; o[3] = 255; }
    else if (comp == 2) { o[0] = o[1] = o[2] = s[0]; o[3] = s[1]; }
    else {
      const bool bgr = !c555;              // 24 / 32-bit pixels are stored blue first; the 5-5-5 ones were unpacked in RGB order
      o[0] = s[bgr ? 2 : 0]; o[1] = s[1]; o[2] = s[bgr ? 0 : 2]; o[3] = comp == 4 ? s[3] : 255;
    }
  }
  w_out = w; h_out = h;
  return out;
}
// ---- Photoshop PSD: the merged image only (what the reference's decoder reads): "8BPS" version 1, RGB mode, 8 or 16 bits, raw or PackBits rows ---------------------------
// Channels beyond the file's count are 0 (alpha: 255); 16-bit samples keep their high byte; a PackBits file is read as 8-bit whatever its depth says (the reference's
// choice); with four channels or more the colour is un-matted from white: c' = c / a + 255 (1 - 1 / a), a = alpha / 255, in single precision, truncated to a byte.
inline bool is_psd(const std::uint8_t* d, std::size_t n) { return n >= 4 && d[0] == '8' && d[1] == 'B' && d[2] == 'P' && d[3] == 'S'; }

inline std::vector<std::uint8_t> decode_psd(const std::uint8_t* d, std::size_t n, int& w_out, int& h_out) {
  using namespace misc_detail;
  Bytes r(d, n);
  auto be16 = [&]() { const int hi = r.u8(); return (hi << 8) | r.u8(); };
  auto be32 = [&]() { const std::uint32_t hi = (std::uint32_t)be16(); return (std::int32_t)((hi << 16) | (std::uint32_t)be16()); };
  if (be32() != 0x38425053) throw std::runtime_error("not a PSD image");
  if (be16() != 1) throw std::runtime_error("PSD: unsupported version");
  r.skip(6);
  const int channels = be16();
  if (channels < 0 || channels > 16) throw std::runtime_error("PSD: unsupported channel count");
  const int h = be32(), w = be32();
  if (h > kMaxDim || w > kMaxDim) throw std::runtime_error("PSD: image too large");
  const int depth = be16();
  if (depth != 8 && depth != 16) throw std::runtime_error("PSD: bit depth is not 8 or 16");
  if (be16() != 3) throw std::runtime_error("PSD: not in RGB colour mode");
  r.skip(be32()); r.skip(be32()); r.skip(be32());          // mode data, image resources, layer and mask information
  const int compression = be16();
  if (compression > 1) throw std::runtime_error("PSD: unknown compression");
  check_area(w, h, 4);
  const std::size_t np = (std::size_t)w * (std::size_t)h;
  std::vector<std::uint8_t> out(np * 4);
  if (compression) {
    r.skip((long long)h * channels * 2);                   // the rows' byte counts
    for (int c = 0; c < 4; ++c) {
      if (c >= channels) { for (std::size_t i = 0; i < np; ++i) out[i * 4 + (std::size_t)c] = c == 3 ? 255 : 0; continue; }
      std::size_t done = 0;
      while (done < np) {
        int len = r.u8();
        if (len == 128) continue;                          // (at the end of the file the bytes read as 0: a literal of one byte)
        if (len < 128) {
          ++len;
          if ((std::size_t)len > np - done) throw std::runtime_error("PSD: bad run-length data");
          for (; len; --len) out[(done++) * 4 + (std::size_t)c] = (std::uint8_t)r.u8();
        } else {
          len = 257 - len;
          if ((std::size_t)len > np - done) throw std::runtime_error("PSD: bad run-length data");
          const std::uint8_t v = (std::uint8_t)r.u8();
          for (; len; --len) out[(done++) * 4 + (std::size_t)c] = v;
        }
      }
    }
  } else {
    for (int c = 0; c < 4; ++c)
      for (std::size_t i = 0; i < np; ++i)
        out[i * 4 + (std::size_t)c] = c >= channels ? (c == 3 ? 255 : 0) : (depth == 16 ? (std::uint8_t)(be16() >> 8) : (std::uint8_t)r.u8());
  }
  if (channels >= 4)
    for (std::size_t i = 0; i < np; ++i) {
      std::uint8_t* px = &out[i * 4];
      if (px[3] != 0 && px[3] != 255) {
        const float a = px[3] / 255.0f, ra = 1.0f / a, inv_a = 255.0f * (1 - ra);
        for (int k = 0; k < 3; ++k) px[k] = (std::uint8_t)(int)(px[k] * ra + inv_a);
      }
    }
  w_out = w; h_out = h;
  return out;
}

// ---- GIF: the FIRST image of the file, as the reference's decoder composes it ------------------------------------------------------------------------------------------
// Canvas of the logical screen, transparent black; the first image descriptor's rectangle is decoded (LZW, interlaced or not, local or global colour table) and drawn
// through the table — an index marked transparent by a preceding graphic-control extension leaves the canvas as it is; then, when the header names a background index
// above 0, every pixel the image did NOT touch takes that entry of the global table, opaque — written in the table's own byte order, blue first (the reference's memcpy).
inline bool is_gif(const std::uint8_t* d, std::size_t n) {
  return n >= 6 && d[0] == 'G' && d[1] == 'I' && d[2] == 'F' && d[3] == '8' && (d[4] == '7' || d[4] == '9') && d[5] == 'a';
}

inline std::vector<std::uint8_t> decode_gif(const std::uint8_t* d, std::size_t n, int& w_out, int& h_out) {
  using namespace misc_detail;
  if (!is_gif(d, n)) throw std::runtime_error("not a GIF image");
  Bytes r(d, n);
  r.skip(6);
  const int W = r.u16(), H = r.u16(), flags = r.u8(), bgindex = r.u8();
  r.u8();                                                  // aspect ratio
  std::uint8_t pal[256][4] = {}, lpal[256][4] = {};        // entries as the file orders them reversed: (b, g, r, a)
  auto table = [&](std::uint8_t t[256][4], int entries, int transparent) {
    for (int i = 0; i < entries; ++i) { t[i][2] = (std::uint8_t)r.u8(); t[i][1] = (std::uint8_t)r.u8(); t[i][0] = (std::uint8_t)r.u8(); t[i][3] = transparent == i ? 0 : 255; }
  };
  if (flags & 0x80) table(pal, 2 << (flags & 7), -1);
  check_area(W, H, 4);
  const std::size_t np = (std::size_t)W * (std::size_t)H;
  std::vector<std::uint8_t> out(np * 4, 0), touched(np, 0);
  int eflags = 0, transparent = -1;
  for (;;) {
    const int tag = r.u8();
    if (tag == 0x2C) {
      const int x = r.u16(), y = r.u16(), w = r.u16(), h = r.u16();
      if (x + w > W || y + h > H) throw std::runtime_error("GIF: bad image descriptor");
      const long long line = (long long)W * 4;
      const long long start_x = (long long)x * 4, start_y = (long long)y * line, max_x = start_x + (long long)w * 4, max_y = start_y + (long long)h * line;
      long long cur_x = start_x, cur_y = w == 0 ? max_y : start_y, step;
      const int lflags = r.u8();
      int parse;
      if (lflags & 0x40) { step = 8 * line; parse = 3; } else { step = line; parse = 0; }
      const std::uint8_t (*ct)[4];
      if (lflags & 0x80) { table(lpal, 2 << (lflags & 7), (eflags & 1) ? transparent : -1); ct = lpal; }
      else if (flags & 0x80) ct = pal;
      else throw std::runtime_error("GIF: missing colour table");
      // ---- LZW raster ----
      const int lzw_cs = r.u8();
      if (lzw_cs > 12) throw std::runtime_error("GIF: bad code size");
      struct Code { std::int16_t prefix; std::uint8_t first, suffix; };
      std::vector<Code> codes(8193);
      const int clear = 1 << lzw_cs;
      for (int i = 0; i < clear; ++i) codes[(std::size_t)i] = {(std::int16_t)-1, (std::uint8_t)i, (std::uint8_t)i};
      int codesize = lzw_cs + 1, codemask = (1 << codesize) - 1, avail = clear + 2, oldcode = -1, valid_bits = 0, len = 0;
      std::int32_t bits = 0;
      bool first = true;
      std::vector<std::uint8_t> chain;
      auto emit = [&](int code) {                          // the string of a code, first symbol first
        chain.clear();
        for (int c = code; c >= 0; c = codes[(std::size_t)c].prefix) chain.push_back(codes[(std::size_t)c].suffix);
        for (std::size_t k = chain.size(); k-- > 0;) {
          if (cur_y >= max_y) return;
          const std::size_t idx = (std::size_t)(cur_x + cur_y);
          touched[idx / 4] = 1;
          const std::uint8_t* c = ct[chain[k]];
          if (c[3] > 128) { out[idx] = c[2]; out[idx + 1] = c[1]; out[idx + 2] = c[0]; out[idx + 3] = c[3]; }
          cur_x += 4;
          if (cur_x >= max_x) {
            cur_x = start_x; cur_y += step;
            while (cur_y >= max_y && parse > 0) { step = (1LL << parse) * line; cur_y = start_y + (step >> 1); --parse; }
          }
        }
      };
      for (;;) {
        if (valid_bits < codesize) {
          if (len == 0) { len = r.u8(); if (len == 0) break; }
          --len;
          bits |= (std::int32_t)((std::uint32_t)r.u8() << valid_bits);
          valid_bits += 8;
        } else {
          const int code = bits & codemask;
          bits >>= codesize; valid_bits -= codesize;
          if (code == clear) { codesize = lzw_cs + 1; codemask = (1 << codesize) - 1; avail = clear + 2; oldcode = -1; first = false; }
          else if (code == clear + 1) break;               // end of the image's data (what follows it in the file is not looked at)
          else if (code <= avail) {
            if (first) throw std::runtime_error("GIF: no clear code");
            if (oldcode >= 0) {
              Code& p = codes[(std::size_t)avail++];
              if (avail > 8192) throw std::runtime_error("GIF: too many codes");
              p.prefix = (std::int16_t)oldcode;
              p.first = codes[(std::size_t)oldcode].first;
              p.suffix = code == avail ? p.first : codes[(std::size_t)code].first;
            } else if (code == avail) throw std::runtime_error("GIF: illegal code");
            emit(code);
            if ((avail & codemask) == 0 && avail <= 0x0fff) { ++codesize; codemask = (1 << codesize) - 1; }
            oldcode = code;
          } else throw std::runtime_error("GIF: illegal code");
        }
      }
      if (bgindex > 0)
        for (std::size_t i = 0; i < np; ++i)
          if (!touched[i]) { out[i * 4] = pal[bgindex][0]; out[i * 4 + 1] = pal[bgindex][1]; out[i * 4 + 2] = pal[bgindex][2]; out[i * 4 + 3] = 255; }
      w_out = W; h_out = H;
      return out;
    } else if (tag == 0x21) {
      const int ext = r.u8();
      int len;
      if (ext == 0xF9) {
        len = r.u8();
        if (len == 4) {
          eflags = r.u8();
          r.u16();                                         // delay
          if (transparent >= 0) pal[transparent][3] = 255;
          if (eflags & 1) { transparent = r.u8(); pal[transparent][3] = 0; }
          else { r.skip(1); transparent = -1; }
        } else { r.skip(len); continue; }
      }
      while ((len = r.u8()) != 0) r.skip(len);
    } else if (tag == 0x3B) throw std::runtime_error("GIF: no image in the file");
    else throw std::runtime_error("GIF: unknown block");
  }
}
// ---- Softimage PIC: magic 53 80 F6 34, "PICT" at byte 88, big-endian size at 92; up to ten channel packets (8 bits each; channel mask 0x80 red ... 0x10 alpha), every
// row holding each packet's channels uncompressed (type 0), as runs of (count, value) (type 1) or as mixed runs (type 2: count >= 128 a run of count - 127, or of a 16-bit
// count when 128; below 128 that many + 1 literal pixels).  Channels no packet carries are 255.  Unlike the formats above, a file that ends early is refused (as there).
inline bool is_pic(const std::uint8_t* d, std::size_t n) {
  return n >= 92 && d[0] == 0x53 && d[1] == 0x80 && d[2] == 0xF6 && d[3] == 0x34 && d[88] == 'P' && d[89] == 'I' && d[90] == 'C' && d[91] == 'T';
}

inline std::vector<std::uint8_t> decode_pic(const std::uint8_t* d, std::size_t n, int& w_out, int& h_out) {
  using namespace misc_detail;
  if (!is_pic(d, n)) throw std::runtime_error("not a Softimage PIC image");
  Bytes r(d, n);
  r.skip(92);
  auto be16 = [&]() { const int hi = r.u8(); return (hi << 8) | r.u8(); };
  const int w = be16(), h = be16();
  if (r.eof()) throw std::runtime_error("PIC: file too short");
  check_area(w, h, 4);
  r.skip(8);                                               // ratio, fields, pad
  std::vector<std::uint8_t> out((std::size_t)w * (std::size_t)h * 4, 0xff);
  struct Packet { int type, channel; } packets[10];
  int n_packets = 0, chained;
  do {
    if (n_packets == 10) throw std::runtime_error("PIC: too many packets");
    chained = r.u8();
    const int size = r.u8();
    packets[n_packets].type = r.u8(); packets[n_packets].channel = r.u8();
    ++n_packets;
    if (r.eof()) throw std::runtime_error("PIC: file too short");
    if (size != 8) throw std::runtime_error("PIC: packet is not 8 bits per channel");
  } while (chained);
  auto read_value = [&](int channel, std::uint8_t* dst) {
    for (int i = 0, mask = 0x80; i < 4; ++i, mask >>= 1)
      if (channel & mask) { if (r.eof()) throw std::runtime_error("PIC: file too short"); dst[i] = (std::uint8_t)r.u8(); }
  };
  auto copy_value = [](int channel, std::uint8_t* dst, const std::uint8_t* src) {
    for (int i = 0, mask = 0x80; i < 4; ++i, mask >>= 1) if (channel & mask) dst[i] = src[i];
  };
  for (int y = 0; y < h; ++y)
    for (int k = 0; k < n_packets; ++k) {
      const Packet& p = packets[k];
      std::uint8_t* dst = &out[(std::size_t)y * (std::size_t)w * 4];
      if (p.type == 0) { for (int x = 0; x < w; ++x, dst += 4) read_value(p.channel, dst); }
      else if (p.type == 1) {
        int left = w;
        while (left > 0) {
          int count = r.u8();
          if (r.eof()) throw std::runtime_error("PIC: file too short");
          if (count > left) count = left & 255;
          std::uint8_t v[4];
          read_value(p.channel, v);
          for (int i = 0; i < count; ++i, dst += 4) copy_value(p.channel, dst, v);
          left -= count;
        }
      } else if (p.type == 2) {
        int left = w;
        while (left > 0) {
          int count = r.u8();
          if (r.eof()) throw std::runtime_error("PIC: file too short");
          if (count >= 128) {
            count = count == 128 ? be16() : count - 127;
            if (count > left) throw std::runtime_error("PIC: scanline overrun");
            std::uint8_t v[4];
            read_value(p.channel, v);
            for (int i = 0; i < count; ++i, dst += 4) copy_value(p.channel, dst, v);
          } else {
            ++count;
            if (count > left) throw std::runtime_error("PIC: scanline overrun");
            for (int i = 0; i < count; ++i, dst += 4) read_value(p.channel, dst);
          }
          left -= count;
        }
      } else throw std::runtime_error("PIC: bad compression type");
    }
  w_out = w; h_out = h;
  return out;
}
}  // namespace pbr::image
