// image_io.hpp — dependency-free image writers for the offline renderer: PFM for the fp32 radiance
// buffer, PNG (8-bit RGBA, stored-deflate: no compression library needed) for the tonemapped output that the
// reference only ever presents to a swapchain (src/gltf_viewer/App.cpp:384-393).
#pragma once
#include <cmath>
#include <cstdint>
#include <iterator>
#include <cstring>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace pbr::image {

// rgba: w*h*4 floats, row 0 = top (y-down like the HdrImage); PFM stores rows bottom-up, RGB, little endian (-1.0)
inline void write_pfm(const std::string& path, const float* rgba, int w, int h) {
  std::ofstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot write " + path);
  f << "PF\n" << w << " " << h << "\n-1.0\n";
  for (int y = h - 1; y >= 0; --y)
    for (int x = 0; x < w; ++x) f.write(reinterpret_cast<const char*>(&rgba[((std::size_t)y * w + x) * 4]), 12);
}

// PFM (PF = RGB, Pf = grey; little or big endian by the sign of the scale line) → w*h*3 floats, row 0 = top.
inline std::vector<float> read_pfm(const std::string& path, int& w, int& h) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot read " + path);
  std::string magic; double scale = 0.0;
  f >> magic >> w >> h >> scale;
  if ((magic != "PF" && magic != "Pf") || w <= 0 || h <= 0 || scale == 0.0 || (std::uint64_t)w * (std::uint64_t)h > (1u << 28)) throw std::runtime_error(path + ": not a PFM image");
  f.get();                                                   // the single whitespace byte after the scale line
  const int ch = magic == "PF" ? 3 : 1;
  std::vector<float> raw((std::size_t)w * h * ch);
  f.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)(raw.size() * 4));
  if (!f) throw std::runtime_error(path + ": truncated PFM image");
  if (scale > 0.0)                                           // big endian
    for (float& v : raw) { std::uint32_t u; std::memcpy(&u, &v, 4); u = (u >> 24) | ((u >> 8) & 0xff00u) | ((u << 8) & 0xff0000u) | (u << 24); std::memcpy(&v, &u, 4); }
  std::vector<float> out((std::size_t)w * h * 3);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      for (int c = 0; c < 3; ++c) out[((std::size_t)y * w + x) * 3 + c] = raw[((std::size_t)(h - 1 - y) * w + x) * ch + (ch == 3 ? c : 0)];
  return out;
}

// Radiance RGBE (.hdr / .pic) from memory → w*h*3 floats, row 0 = top: the format the reference's image::loadImage2D would take through stb_image's HDR path
// (src/pbr_engine/image/stb/stb_image.h: stbi__hdr_load).  Header: "#?RADIANCE" or "#?RGBE", lines up to an empty one, one of them "FORMAT=32-bit_rle_rgbe"; then
// "-Y h +X w" (the only orientation stb reads); scanlines flat (4 bytes per pixel) or new-style run-length encoded (2 2 hi lo, then the four channels one after
// the other: a count byte > 128 = a run of count - 128, else that many literal bytes).  Pixel: (r, g, b) * 2^(e - 136), e = 0 → black — ldexp, exact in float.
inline std::vector<float> decode_hdr(const std::uint8_t* d, std::size_t n, int& w, int& h) {
  std::size_t at = 0;
  auto line = [&]() { std::string l; while (at < n && d[at] != '\n') l.push_back((char)d[at++]); if (at < n) ++at; return l; };
  const std::string magic = line();
  if (magic != "#?RADIANCE" && magic != "#?RGBE") throw std::runtime_error("not a Radiance HDR image");
  bool fmt = false;
  for (;;) {
    if (at >= n) throw std::runtime_error("HDR: truncated header");
    const std::string l = line();
    if (l.empty()) break;
    if (l == "FORMAT=32-bit_rle_rgbe") fmt = true;
  }
  if (!fmt) throw std::runtime_error("HDR: unsupported format (only 32-bit_rle_rgbe)");
  const std::string dims = line();
  long hh = 0, ww = 0;
  {
    if (dims.compare(0, 3, "-Y ") != 0) throw std::runtime_error("HDR: unsupported data layout (only -Y h +X w)");
    char* end = nullptr;
    hh = std::strtol(dims.c_str() + 3, &end, 10);
    while (*end == ' ') ++end;
    if (std::strncmp(end, "+X ", 3) != 0) throw std::runtime_error("HDR: unsupported data layout (only -Y h +X w)");
    ww = std::strtol(end + 3, nullptr, 10);
  }
  if (hh <= 0 || ww <= 0 || hh > (1 << 24) || ww > (1 << 24) || (std::uint64_t)hh * (std::uint64_t)ww > (1u << 28)) throw std::runtime_error("HDR: bad size");
  w = (int)ww; h = (int)hh;
  std::vector<float> out((std::size_t)w * h * 3);
  auto convert = [](const std::uint8_t* px, float* o) {
    if (px[3] != 0) { const float f = std::ldexp(1.0f, (int)px[3] - (128 + 8)); o[0] = (float)px[0] * f; o[1] = (float)px[1] * f; o[2] = (float)px[2] * f; }
    else o[0] = o[1] = o[2] = 0.0f;
  };
  auto need = [&](std::size_t k) { if (at + k > n) throw std::runtime_error("HDR: truncated data"); };
  std::vector<std::uint8_t> scan((std::size_t)w * 4);
  bool flat = w < 8 || w >= 32768;                               // the run-length form exists for these widths only
  for (int y = 0; y < h; ++y) {
    if (!flat) {
      need(4);
      const std::uint8_t c1 = d[at], c2 = d[at + 1], len = d[at + 2];
      if (c1 != 2 || c2 != 2 || (len & 0x80)) {
        if (y != 0) throw std::runtime_error("HDR: corrupt scanline header");
        flat = true;                                             // an old, flat file: this was its first pixel
      } else {
        if ((((int)len << 8) | d[at + 3]) != w) throw std::runtime_error("HDR: scanline width does not match the header");
        at += 4;
        for (int k = 0; k < 4; ++k) {
          int i = 0;
          while (i < w) {
            need(1);
            int count = d[at++];
            if (count > 128) {                                   // a run
              count -= 128;
              if (count == 0 || count > w - i) throw std::runtime_error("HDR: corrupt run");
              need(1);
              const std::uint8_t v = d[at++];
              for (int z = 0; z < count; ++z) scan[(std::size_t)(i++) * 4 + k] = v;
            } else {                                             // literals
              if (count == 0 || count > w - i) throw std::runtime_error("HDR: corrupt literal block");
              need((std::size_t)count);
              for (int z = 0; z < count; ++z) scan[(std::size_t)(i++) * 4 + k] = d[at++];
            }
          }
        }
        for (int x = 0; x < w; ++x) convert(&scan[(std::size_t)x * 4], &out[((std::size_t)y * w + x) * 3]);
        continue;
      }
    }
    need((std::size_t)w * 4);                                    // flat: the rest of the file is 4 bytes per pixel, row after row
    for (int x = 0; x < w; ++x) convert(d + at + (std::size_t)x * 4, &out[((std::size_t)y * w + x) * 3]);
    at += (std::size_t)w * 4;
  }
  return out;
}
// A Radiance image used as an 8-bit TEXTURE: what the reference's stbi_load_from_memory(..., 4) makes of one (stb_image.h stbi__hdr_to_ldr with its default gamma 2.2 and
// scale 1): channel → (float) pow(v, 1 / 2.2f) * 255 + 0.5f, clamped to [0, 255] and truncated; alpha 255.  pow is the C library's double pow, as there.
inline bool is_hdr(const std::uint8_t* d, std::size_t n) {
  auto starts = [&](const char* sig) { const std::size_t k = std::strlen(sig); return n >= k && std::memcmp(d, sig, k) == 0; };
  return starts("#?RADIANCE\n") || starts("#?RGBE\n");
}
inline std::vector<std::uint8_t> decode_hdr_rgba8(const std::uint8_t* d, std::size_t n, int& w, int& h) {
  const std::vector<float> f = decode_hdr(d, n, w, h);
  std::vector<std::uint8_t> out((std::size_t)w * (std::size_t)h * 4);
  const float gamma_i = 1.0f / 2.2f, scale_i = 1.0f;
  for (std::size_t p = 0; p < (std::size_t)w * (std::size_t)h; ++p) {
    for (int k = 0; k < 3; ++k) {
      float z = (float)std::pow((double)(f[p * 3 + (std::size_t)k] * scale_i), (double)gamma_i) * 255 + 0.5f;
      if (z < 0) z = 0;
      if (z > 255) z = 255;
      out[p * 4 + (std::size_t)k] = (std::uint8_t)(int)z;
    }
    out[p * 4 + 3] = 255;
  }
  return out;
}

inline std::vector<float> read_hdr(const std::string& path, int& w, int& h) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot read " + path);
  const std::vector<std::uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  return decode_hdr(bytes.data(), bytes.size(), w, h);
}

namespace detail {
inline std::uint32_t crc32(const std::uint8_t* p, std::size_t n, std::uint32_t crc = 0) {
  static std::uint32_t table[256];
  static bool init = false;
  if (!init) {
    for (std::uint32_t i = 0; i < 256; ++i) { std::uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
    init = true;
  }
  crc = ~crc;
  for (std::size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
  return ~crc;
}
inline void be32(std::vector<std::uint8_t>& v, std::uint32_t x) { v.push_back((std::uint8_t)(x >> 24)); v.push_back((std::uint8_t)(x >> 16)); v.push_back((std::uint8_t)(x >> 8)); v.push_back((std::uint8_t)x); }
inline void chunk(std::vector<std::uint8_t>& out, const char type[4], const std::vector<std::uint8_t>& data) {
  be32(out, (std::uint32_t)data.size());
  const std::size_t start = out.size();
  out.insert(out.end(), type, type + 4);
  out.insert(out.end(), data.begin(), data.end());
  be32(out, crc32(&out[start], out.size() - start));
}
}  // namespace detail

// rgba8: w*h*4 bytes, row 0 = top.  PNG colour type 6 (RGBA), 8 bits, filter 0, zlib stream of stored blocks.
inline void write_png(const std::string& path, const std::uint8_t* rgba8, int w, int h) {
  using namespace detail;
  std::vector<std::uint8_t> raw;
  raw.reserve((std::size_t)h * ((std::size_t)w * 4 + 1));
  for (int y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), rgba8 + (std::size_t)y * w * 4, rgba8 + (std::size_t)(y + 1) * w * 4); }
  std::vector<std::uint8_t> z{0x78, 0x01};
  std::uint32_t a = 1, b = 0;                                   // Adler-32 of the raw stream
  for (std::uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
  for (std::size_t p = 0; p < raw.size() || p == 0;) {
    const std::size_t n = raw.size() - p < 65535 ? raw.size() - p : 65535;
    z.push_back(p + n >= raw.size() ? 1 : 0);                   // BFINAL, BTYPE = 00 (stored)
    z.push_back((std::uint8_t)(n & 0xFF)); z.push_back((std::uint8_t)(n >> 8));
    z.push_back((std::uint8_t)(~n & 0xFF)); z.push_back((std::uint8_t)((~n >> 8) & 0xFF));
    z.insert(z.end(), raw.begin() + (long)p, raw.begin() + (long)(p + n));
    p += n;
    if (n == 0) break;
  }
  be32(z, (b << 16) | a);
  std::vector<std::uint8_t> out{0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<std::uint8_t> ihdr;
  be32(ihdr, (std::uint32_t)w); be32(ihdr, (std::uint32_t)h);
  ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});
  chunk(out, "IHDR", ihdr);
  chunk(out, "IDAT", z);
  chunk(out, "IEND", {});
  std::ofstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot write " + path);
  f.write(reinterpret_cast<const char*>(out.data()), (std::streamsize)out.size());
}

}  // namespace pbr::image
