// gather_bench.hip — what does a random gather of small records cost on gfx950, per CU?
//
// The trace kernels fetch one BVH node per lane and visit: 64 lanes x 64..80 B from 64 unrelated addresses of an
// L2-resident array.  This measures the per-CU rate of that access pattern in the forms the kernels could use:
//   own      every lane loads its own record with R dwordx4 loads (R = record bytes / 16): each wave-instruction touches 64 lines
//   quad     the record of lane q is loaded by R adjacent lanes (one dwordx4 each): a wave-instruction touches 64/R records,
//            whole 64-B (R = 4) records per lane quad; data reaches its owner through LDS (ds_write_b128 + ds_read_b128)
//   quad_dma the same addresses, loaded straight into LDS (global_load_lds_dwordx4: wave-uniform base + lane x 16), owner reads LDS
// for 64-B records (R = 4) and 80-B records (R = 5, records not line-aligned), table sizes that sit in L2 (2 MB) and beyond it
// (64 MB: Infinity Cache).  Output: one JSON line (profiles/r02_gather_bench.json).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_bench tools/gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

typedef float fx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) fx4 lds_f4;
typedef __attribute__((address_space(1))) const fx4 glb_f4;

constexpr int kIters = 512;
constexpr int kWaves = 4;   // per block

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }
__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// MODE 0 own, 1 quad through registers + LDS, 2 quad by LDS-DMA.  R = 16-byte pieces per record.
template <int MODE, int R>
__global__ __launch_bounds__(256) void k_gather(const float4* table, uint32_t n_rec, float* out) {
  extern __shared__ float4 lds[];                       // [waves][64 * R] pieces
  const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
  float4* stage = lds + (size_t)wave * 64 * R;
  uint32_t seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  fx4 acc = {0, 0, 0, 0};
  for (int it = 0; it < kIters; ++it) {
    const uint32_t rec = lcg(seed) % n_rec;              // this lane's record for this iteration
    if (MODE == 0) {
      glb_f4* p = (glb_f4*)table + (size_t)rec * R;
#pragma unroll
      for (int k = 0; k < R; ++k) acc += p[k];
    } else {
      // piece p = 64 j + lane of the wave's 64 R pieces belongs to owner p / R, piece p % R of its record
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const uint32_t p = 64u * (uint32_t)j + lane, owner = p / (uint32_t)R, k = p % (uint32_t)R;
        const uint32_t orec = (uint32_t)__shfl((int)rec, (int)owner);
        glb_f4* src = (glb_f4*)table + (size_t)orec * R + k;
        if (MODE == 1) {
          ((lds_f4*)stage)[p] = *src;
        } else {
          // LDS-DMA: destination = wave-uniform base (M0) + lane * 16
          const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(stage + 64 * j));
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(src), "s"(lds_dst) : "memory");
        }
      }
      if (MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int k = 0; k < R; ++k) acc += ((lds_f4*)stage)[lane * R + k];
      __builtin_amdgcn_wave_barrier();
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE, int R>
double run(const float4* table, uint32_t n_rec, float* out, int grid) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t lds = MODE ? (size_t)kWaves * 64 * R * 16 : 0;
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_gather<MODE, R>), dim3(grid), dim3(256), lds, 0, table, n_rec, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  CHECK(hipGetLastError());
  return (double)ms;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  float4* table; float* out;
  const size_t table_bytes = (size_t)64 << 20;
  CHECK(hipMalloc(&table, table_bytes + 256));
  CHECK(hipMemset(table, 0, table_bytes + 256));
  CHECK(hipMalloc(&out, (size_t)n_cu * 8 * 256 * sizeof(float)));
  std::printf("{\"device\": \"%s\", \"n_cu\": %d, \"iters\": %d, \"runs\": [", prop.gcnArchName, n_cu, kIters);
  bool first = true;
  const char* names[] = {"own", "quad", "quad_dma"};
  for (size_t tb : {(size_t)2 << 20, (size_t)64 << 20})
    for (int bpc : {4, 7})
      for (int r : {4, 5})
        for (int mode = 0; mode < 3; ++mode) {
          const uint32_t n_rec = (uint32_t)(tb / (16 * r));
          const int grid = n_cu * bpc;
          double ms = 0;
#define RUN(M, R) ms = run<M, R>(table, n_rec, out, grid)
          if (r == 4) { if (mode == 0) RUN(0, 4); else if (mode == 1) RUN(1, 4); else RUN(2, 4); }
          else { if (mode == 0) RUN(0, 5); else if (mode == 1) RUN(1, 5); else RUN(2, 5); }
          const double bytes = (double)grid * 256 * kIters * 16.0 * r;
          const double gbs = bytes / (ms * 1e-3) / 1e9;
          std::printf("%s{\"mode\": \"%s\", \"record_bytes\": %d, \"table_mb\": %zu, \"blocks_per_cu\": %d, \"ms\": %.4f, \"chip_GBs\": %.1f, \"GBs_per_cu\": %.2f, \"ns_per_wave_gather\": %.2f}",
                      first ? "" : ", ", names[mode], 16 * r, tb >> 20, bpc, ms, gbs, gbs / n_cu, ms * 1e6 / ((double)bpc * kWaves * kIters));
          first = false;
        }
  std::printf("]}\n");
  return 0;
}
