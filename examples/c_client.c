/* c_client.c — the C-ABI used from plain C99 (no C++, no HIP, no torch types): builds a two-triangle scene with an
 * emitter, renders it, tone-maps it and prints a checksum.  With device = -1 (PTC_DEVICE_NONE) it only describes and
 * commits the scene (host flatten + BVH) and shows that every render call fails cleanly without a GPU.
 *   gcc -std=c99 -Wall -Iinclude examples/c_client.c -Lphysically-based-renderer_amd/lib -lptc -Wl,-rpath,$PWD/physically-based-renderer_amd/lib -o c_client
 *   ./c_client 0        (GPU 0)      ./c_client -1   (description only)                                                   */
#include <ptc.h>
#include <ptc_gltf.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call) do { int rc_ = (call); if (rc_ < 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ptc_last_error(ctx)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int device = argc > 1 ? atoi(argv[1]) : 0;
  ptc_ctx* ctx = ptc_create(device);
  if (!ctx) { fprintf(stderr, "ptc_create(%d) failed: %s\n", device, ptc_last_error(NULL)); return 2; }
  printf("abi %d\n", ptc_abi_version());
  CHECK(ptc_scene_begin(ctx));
  const float white[4] = {0.8f, 0.8f, 0.8f, 1.0f}, none[3] = {0, 0, 0}, glow[3] = {10.0f, 10.0f, 10.0f};
  const int m_floor = ptc_add_material(ctx, white, 0.0f, 1.0f, none, -1, -1, -1);
  const int m_light = ptc_add_material(ctx, white, 0.0f, 1.0f, glow, -1, -1, -1);
  CHECK(m_floor); CHECK(m_light);
  ptc_vertex quad[4];
  memset(quad, 0, sizeof quad);
  const float P[4][3] = {{-1, 0, -1}, {1, 0, -1}, {1, 0, 1}, {-1, 0, 1}};
  for (int i = 0; i < 4; ++i) {
    memcpy(quad[i].position, P[i], 12);
    quad[i].normal[1] = -1.0f;                       /* the reference's world is y-down: "up" is -y (CameraData.hpp:28) */
    quad[i].tangent[0] = 1.0f; quad[i].tangent[3] = 1.0f;
    quad[i].texcoord[0] = P[i][0] * 0.5f + 0.5f; quad[i].texcoord[1] = P[i][2] * 0.5f + 0.5f;
  }
  const uint32_t idx[6] = {0, 2, 1, 0, 3, 2};
  const int mesh_floor = ptc_add_mesh(ctx, quad, 4, idx, 6, m_floor);
  const int mesh_light = ptc_add_mesh(ctx, quad, 4, idx, 6, m_light);
  CHECK(mesh_floor); CHECK(mesh_light);
  const float t0[3] = {0, 0.5f, -3}, q0[4] = {1, 0, 0, 0}, s0[3] = {2, 1, 2};
  const float t1[3] = {0, -1.5f, -3}, q1[4] = {0, 1, 0, 0}, s1[3] = {0.5f, 1, 0.5f};      /* 180 deg about x: faces down */
  CHECK(ptc_add_instance(ctx, mesh_floor, t0, q0, s0));
  CHECK(ptc_add_instance(ctx, mesh_light, t1, q1, s1));
  const float eye[3] = {0, -0.5f, 0}, target[3] = {0, 0, -3};
  CHECK(ptc_set_camera(ctx, eye, target, 1.0f, 1.0f));
  CHECK(ptc_scene_commit(ctx));
  ptc_stats st;
  CHECK(ptc_get_stats(ctx, &st));
  printf("scene: %u triangles, %u BVH nodes, %u emitters\n", (unsigned)st.n_triangles, (unsigned)st.n_bvh_nodes, (unsigned)st.n_emitters);
  const int w = 64, h = 64;
  int rc = ptc_render(ctx, w, h, 16, 1u, 4, PTC_INTEGRATOR_PATH);
  if (device < 0) {
    printf("render without a device: rc %d (%s)\n", rc, ptc_last_error(ctx));
    ptc_destroy(ctx);
    return rc == PTC_E_DEVICE ? 0 : 3;
  }
  CHECK(rc);
  float* img = (float*)malloc(sizeof(float) * 4 * w * h);
  unsigned char* ldr = (unsigned char*)malloc(4u * w * h);
  CHECK(ptc_read_radiance_rgba32f(ctx, img));
  CHECK(ptc_tonemap_rgba8(ctx, ldr));
  double sum = 0.0; unsigned long lsum = 0;
  for (int i = 0; i < w * h; ++i) { sum += img[i * 4] + img[i * 4 + 1] + img[i * 4 + 2]; lsum += ldr[i * 4]; }
  CHECK(ptc_get_stats(ctx, &st));
  printf("rendered %llu paths, %llu segments; radiance sum %.6f, ldr sum %lu\n", (unsigned long long)st.paths, (unsigned long long)st.segments, sum, lsum);
  free(img); free(ldr);
  ptc_destroy(ctx);
  return sum > 0.0 ? 0 : 4;
}
