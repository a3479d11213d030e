// Fuzz entry points for tools/sanitize.sh (host-only, built with -fsanitize=address,undefined).
#ifdef FUZZ_PNG
#include "png_decode.hpp"
extern "C" int png_try(const unsigned char* d, unsigned long long n) {
  try { int w, h; auto v = pbr::image::decode_png(d, (size_t)n, w, h); return (int)(v.size() & 0x7fffffff); } catch (std::exception const&) { return -1; }
}
#endif
#ifdef FUZZ_JPEG
#include "jpeg_decode.hpp"
extern "C" int jpeg_try(const unsigned char* d, unsigned long long n) {
  try { int w, h; auto v = pbr::image::decode_jpeg(d, (size_t)n, w, h); return (int)(v.size() & 0x7fffffff); } catch (std::exception const&) { return -1; }
}
#endif
#ifdef FUZZ_GLTF
#include "gltf_loader.hpp"
extern "C" {
// link-time stand-ins so that the header's upload() resolves; the fuzz target only runs load()
int ptc_add_material(ptc_ctx*, const float*, float, float, const float*, int, int, int) { return 0; }
int ptc_add_texture_rgba8(ptc_ctx*, const uint8_t*, int, int) { return 0; }
int ptc_add_mesh(ptc_ctx*, const ptc_vertex*, uint32_t, const uint32_t*, uint32_t, int) { return 0; }
int ptc_add_instance_matrix(ptc_ctx*, int, const float*) { return 0; }
int gltf_try(const char* path) {
  try { auto s = pbr::gltf::load(path); return (int)(s.n_triangles & 0x7fffffff); } catch (std::exception const&) { return -1; }
}
}
#endif
#ifdef FUZZ_MISC
#include "image_io.hpp"
#include "misc_decode.hpp"
// kind as ptc_image_decode_rgba8: 1 BMP, 2 TGA, 3 PNM, 4 Radiance as a texture, 5 GIF, 6 PSD, 7 PIC
extern "C" int misc_try(int kind, const unsigned char* d, unsigned long long n) {
  try {
    int w, h;
    std::vector<std::uint8_t> v;
    switch (kind) {
      case 1: v = pbr::image::decode_bmp(d, (size_t)n, w, h); break;
      case 2: v = pbr::image::decode_tga(d, (size_t)n, w, h); break;
      case 3: v = pbr::image::decode_pnm(d, (size_t)n, w, h); break;
      case 4: v = pbr::image::decode_hdr_rgba8(d, (size_t)n, w, h); break;
      case 5: v = pbr::image::decode_gif(d, (size_t)n, w, h); break;
      case 6: v = pbr::image::decode_psd(d, (size_t)n, w, h); break;
      default: v = pbr::image::decode_pic(d, (size_t)n, w, h); break;
    }
    return (int)(v.size() & 0x7fffffff);
  } catch (std::exception const&) { return -1; }
}
#endif
