// ptc_gltf.cpp — C-ABI wrapper of gltf_loader.hpp (include/ptc_gltf.h).
#include "gltf_loader.hpp"
#include "image_io.hpp"

#include <ptc_gltf.h>

#include <cstdio>
#include <cstring>

extern "C" long long ptc_gltf_load(ptc_ctx* ctx, const char* path, int scene_index, int compose_parents, float bbox6[6], char* err, int err_len) {
  auto say = [&](const std::string& m) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", m.c_str()); };
  if (!ctx || !path) { say("ptc_gltf_load: null argument"); return PTC_E_ARG; }
  try {
    const pbr::gltf::FlatScene s = pbr::gltf::load(path, scene_index, compose_parents != 0);
    const int rc = pbr::gltf::upload(ctx, s);
    if (rc < 0) { say(std::string("ptc_gltf_load: ") + ptc_last_error(ctx)); return rc; }
    if (bbox6) { for (int k = 0; k < 3; ++k) { bbox6[k] = s.bbox_lo[k]; bbox6[3 + k] = s.bbox_hi[k]; } }
    return (long long)s.n_triangles;
  } catch (std::exception const& e) {
    say(std::string("ptc_gltf_load: ") + e.what());
    return PTC_E_ARG;
  }
}

namespace {
template <class Decode>
int decode_into(const char* who, Decode&& dec, unsigned char* out, unsigned long long out_capacity, int* w, int* h, char* err, int err_len) {
  auto say = [&](const std::string& m) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", m.c_str()); };
  if (!w || !h) { say(std::string(who) + ": null argument"); return PTC_E_ARG; }
  try {
    const std::vector<std::uint8_t> px = dec(*w, *h);
    if (out) {
      if (out_capacity < px.size()) { say(std::string(who) + ": output buffer too small"); return PTC_E_ARG; }
      std::memcpy(out, px.data(), px.size());
    }
    return PTC_OK;
  } catch (std::exception const& e) {
    say(std::string(who) + ": " + e.what());
    return PTC_E_ARG;
  }
}
}  // namespace

extern "C" int ptc_jpeg_decode_rgba8(const unsigned char* data, unsigned long long n, unsigned char* out, unsigned long long out_capacity, int* w, int* h, char* err, int err_len) {
  if (!data) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "ptc_jpeg_decode_rgba8: null argument"); return PTC_E_ARG; }
  return decode_into("ptc_jpeg_decode_rgba8", [&](int& ww, int& hh) { return pbr::image::decode_jpeg(data, (std::size_t)n, ww, hh); }, out, out_capacity, w, h, err, err_len);
}

extern "C" int ptc_png_decode_rgba8(const unsigned char* data, unsigned long long n, unsigned char* out, unsigned long long out_capacity, int* w, int* h, char* err, int err_len) {
  auto say = [&](const std::string& m) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", m.c_str()); };
  if (!data || !w || !h) { say("ptc_png_decode_rgba8: null argument"); return PTC_E_ARG; }
  try {
    const std::vector<std::uint8_t> px = pbr::image::decode_png(data, (std::size_t)n, *w, *h);
    if (out) {
      if (out_capacity < px.size()) { say("ptc_png_decode_rgba8: output buffer too small"); return PTC_E_ARG; }
      std::memcpy(out, px.data(), px.size());
    }
    return PTC_OK;
  } catch (std::exception const& e) {
    say(std::string("ptc_png_decode_rgba8: ") + e.what());
    return PTC_E_ARG;
  }
}

extern "C" int ptc_hdr_decode_rgb32f(const unsigned char* data, unsigned long long n, float* out, unsigned long long out_capacity_floats, int* w, int* h, char* err, int err_len) {
  auto say = [&](const std::string& m) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "%s", m.c_str()); };
  if (!data || !w || !h) { say("ptc_hdr_decode_rgb32f: null argument"); return PTC_E_ARG; }
  try {
    const std::vector<float> px = pbr::image::decode_hdr(data, (std::size_t)n, *w, *h);
    if (out) {
      if (out_capacity_floats < px.size()) { say("ptc_hdr_decode_rgb32f: output buffer too small"); return PTC_E_ARG; }
      std::memcpy(out, px.data(), px.size() * sizeof(float));
    }
    return PTC_OK;
  } catch (std::exception const& e) {
    say(std::string("ptc_hdr_decode_rgb32f: ") + e.what());
    return PTC_E_ARG;
  }
}


// BMP / TGA / binary PGM-PPM (host/misc_decode.hpp); kind: 0 = by content in the reference's order (BMP, GIF, PSD, PIC, PNM, Radiance, TGA), 1 BMP, 2 TGA, 3 PNM, 4 Radiance as an 8-bit texture, 5 GIF, 6 PSD, 7 PIC
extern "C" int ptc_image_decode_rgba8(int kind, const unsigned char* data, unsigned long long n, unsigned char* out, unsigned long long out_capacity, int* w, int* h, char* err, int err_len) {
  if (!data) { if (err && err_len > 0) std::snprintf(err, (size_t)err_len, "ptc_image_decode_rgba8: null argument"); return PTC_E_ARG; }
  return decode_into("ptc_image_decode_rgba8", [&](int& ww, int& hh) {
    const std::size_t nn = (std::size_t)n;
    if (kind == 1 || (kind == 0 && pbr::image::is_bmp(data, nn))) return pbr::image::decode_bmp(data, nn, ww, hh);
    if (kind == 5 || (kind == 0 && pbr::image::is_gif(data, nn))) return pbr::image::decode_gif(data, nn, ww, hh);
    if (kind == 6 || (kind == 0 && pbr::image::is_psd(data, nn))) return pbr::image::decode_psd(data, nn, ww, hh);
    if (kind == 7 || (kind == 0 && pbr::image::is_pic(data, nn))) return pbr::image::decode_pic(data, nn, ww, hh);
    if (kind == 3 || (kind == 0 && pbr::image::is_pnm(data, nn))) return pbr::image::decode_pnm(data, nn, ww, hh);
    if (kind == 4 || (kind == 0 && pbr::image::is_hdr(data, nn))) return pbr::image::decode_hdr_rgba8(data, nn, ww, hh);
    if (kind == 2 || (kind == 0 && pbr::image::is_tga(data, nn))) return pbr::image::decode_tga(data, nn, ww, hh);
    throw std::runtime_error("not a BMP, GIF, PSD, PIC, PGM / PPM, Radiance or TGA image");
  }, out, out_capacity, w, h, err, err_len);
}
