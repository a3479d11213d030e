/* ptc_gltf.h — C-ABI of the glTF 2.0 / GLB loader (libptc_gltf.so), SURVEY.md §8f-1.
 *
 * Replaces, for the path tracer, gltf::Loader::loadAsset + Asset::loadScene
 * (src/pbr_engine/gltf/pbr/gltf/Loader.hpp:20-21, Asset.hpp:76-78): reads a .gltf / .glb file and issues the
 * ptc_add_texture_rgba8 / ptc_add_material / ptc_add_mesh / ptc_add_instance_matrix calls of include/ptc.h on `ctx`.
 * Images (Asset::loadImage2D, Asset.cpp:121-133; bufferView, file URI or base64 data URI) are decoded to RGBA8 like
 * image::loadImage2D does (LoadImage.cpp:56-73: 4 channels of 8 bits whatever the file holds) by the library's own PNG
 * and JPEG decoders (and BMP, TGA, PGM / PPM: ptc_image_decode_rgba8 below), which reproduce the output of the reference's vendored stb_image bit for bit (tests/test_jpeg.py,
 * tests/test_png.py against oracle/_ref).  Samplers are ignored: the reference creates
 * default samplers (NEAREST, REPEAT) whatever the asset says (Asset.cpp:116-117).
 * Call between ptc_scene_begin and ptc_scene_commit; the camera stays the caller's (the reference ignores glTF
 * cameras too and injects its own, Asset.cpp:262-265).  Host-only: works on a PTC_DEVICE_NONE context.
 */
#ifndef PTC_GLTF_H
#define PTC_GLTF_H
#include "ptc.h"
#ifdef __cplusplus
extern "C" {
#endif

/* scene_index: -1 = the asset's default scene.  compose_parents: 1 = glTF-correct world transforms,
 * 0 = each node with its local transform only (the reference's behaviour, PbrRenderSystem.cpp:444-446).
 * bbox6 (may be NULL) receives the world bounds lo.xyz, hi.xyz.  err (may be NULL) receives the error text.
 * Returns the number of triangles instanced (>= 0) or a negative PTC_E_* code. */
long long ptc_gltf_load(ptc_ctx* ctx, const char* path, int scene_index, int compose_parents, float bbox6[6],
                        char* err, int err_len);

/* PNG file image (any colour type / bit depth / interlacing) → w*h*4 bytes RGBA8, row 0 on top; the decoder the loader
 * uses for glTF images (replaces stbi_load_from_memory(..., STBI_rgb_alpha), LoadImage.cpp:56-73).  out may be NULL to
 * query the size only.  Returns 0 or a negative PTC_E_* code with the text in err. */
int ptc_png_decode_rgba8(const unsigned char* data, unsigned long long n, unsigned char* out, unsigned long long out_capacity,
                         int* w, int* h, char* err, int err_len);

/* JPEG file image (baseline / extended / progressive Huffman, 8 bit, 1, 3 or 4 components) → w*h*4 bytes RGBA8 with
 * alpha 255; inverse DCT, chroma upsampling and colour conversion as stb_image does them (stb/stb_image.h:2430-2520,
 * 3465-3528, 3658-3685), so the texels equal the reference's.  Same calling convention as ptc_png_decode_rgba8. */
int ptc_jpeg_decode_rgba8(const unsigned char* data, unsigned long long n, unsigned char* out, unsigned long long out_capacity,
                          int* w, int* h, char* err, int err_len);

/* Radiance RGBE (.hdr) file image → w*h*3 floats, row 0 on top: what stbi_loadf(..., 3) gives for such a file (stb/stb_image.h, stbi__hdr_load: "-Y h +X w" layout,
 * flat or new-style run-length scanlines, pixel = (r, g, b) * 2^(e - 136)) — bit for bit (tests/test_hdr.py against oracle/_ref).  The lat-long environment maps one
 * finds in the wild are in this format: `ptc_render --env sky.hdr` reads them through it (ptc_set_env_latlong_rgb32f takes the floats).  out may be NULL to query the
 * size only; out_capacity_floats counts floats. */
int ptc_hdr_decode_rgb32f(const unsigned char* data, unsigned long long n, float* out, unsigned long long out_capacity_floats,
                          int* w, int* h, char* err, int err_len);

/* The other still-image formats the reference's image path decodes besides PNG and JPEG and that this loader takes too: Windows BMP (palettes, 16 / 24 / 32 bits,
 * bit fields; no run-length forms), Truevision TGA (types 1 / 2 / 3 / 9 / 10 / 11) and binary PGM / PPM ("P5" / "P6", 8- or 16-bit) → w*h*4 bytes RGBA8, row 0 on
 * top: what stbi_load_from_memory(..., 4) gives (stb/stb_image.h stbi__bmp_load, stbi__tga_load, stbi__pnm_load), byte for byte (tests/test_misc_images.py against
 * oracle/_ref).  Also a Radiance .hdr file used as an 8-bit texture: every channel (float) pow(v, 1 / 2.2f) * 255 + 0.5f, clamped and truncated, alpha 255 (stbi__hdr_to_ldr at
 * its defaults); the FIRST image of a GIF (LZW, interlacing, local tables, transparency, the background index as the reference applies it); the merged image of an RGB
 * Photoshop PSD (8 / 16 bits, raw or PackBits, un-matted from white); Softimage PIC (uncompressed, pure and mixed run-length packets).  With PNG and JPEG that is every
 * format the reference's stb build decodes.  kind: 0 = by content, in the reference's order (BMP, GIF, PSD, PIC, PNM, Radiance, TGA — TGA has no signature and is tried
 * last), 1 = BMP, 2 = TGA, 3 = PNM, 4 = Radiance, 5 = GIF, 6 = PSD, 7 = PIC.  Same calling convention as ptc_png_decode_rgba8. */
int ptc_image_decode_rgba8(int kind, const unsigned char* data, unsigned long long n, unsigned char* out, unsigned long long out_capacity,
                           int* w, int* h, char* err, int err_len);

#ifdef __cplusplus
}
#endif
#endif
