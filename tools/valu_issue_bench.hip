// valu_issue_bench.hip — how many cycles does one wave64 VALU instruction occupy its SIMD on gfx950?
//
// Streams of independent v_fma_f32 (16 accumulators per lane, so no dependency stalls) at 1, 2, 4 and 7
// resident waves per SIMD, on every CU at once.  Reports, per occupancy:
//   cyc/instr/wave  — s_memtime ticks of one wave ÷ its instruction count (what ONE wave sees)
//   cyc/instr/SIMD  — the same ÷ waves per SIMD (what the SIMD's issue port delivers: the roofline figure)
// and the same for a stream with an 8-cycle transcendental (v_rcp_f32) mixed in 1:8.
// Output: one JSON line (committed under profiles/ as r02_valu_issue.json).
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue_bench tools/valu_issue_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kAcc = 16;
constexpr int kRep = 8;         // the 16-instruction body is repeated kRep times per loop iteration (loop overhead: 3 scalar instructions per 128)
constexpr int kIters = 8192;    // x kRep x kAcc instructions per lane

// MODE 0: v_fma_f32 only; 1: every 8th instruction a v_rcp_f32; 2: v_pk_fma_f32 (two fp32 fmas per lane and instruction).
// Inline asm, so that the compiler neither packs the scalar stream into v_pk_fma_f32 nor unpacks the packed one.
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k_fma(float* out, unsigned long long* ticks, float a, float b) {
  float acc[kAcc];
  f2 acc2[kAcc];
  const f2 a2 = {a, a}, b2 = {b, b};
#pragma unroll
  for (int i = 0; i < kAcc; ++i) { acc[i] = (float)(threadIdx.x + i); acc2[i] = f2{acc[i], acc[i] + 0.5f}; }
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int rep = 0; rep < kRep; ++rep)
#pragma unroll
    for (int i = 0; i < kAcc; ++i) {
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
      else if (MODE == 1 && (i & 7) == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(acc[i]));
      else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < kAcc; ++i) s += acc[i] + acc2[i].x + acc2[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  const int occ[] = {1, 2, 4, 7};
  float* out; unsigned long long* ticks;
  CHECK(hipMalloc(&out, (size_t)n_cu * 8 * 256 * sizeof(float)));
  CHECK(hipMalloc(&ticks, (size_t)n_cu * 8 * 4 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::printf("{\"device\": \"%s\", \"n_cu\": %d, \"clock_mhz_prop\": %d, \"fma_per_wave\": %d, \"runs\": [", prop.gcnArchName, n_cu, prop.clockRate / 1000, kIters * kRep * kAcc);
  bool first = true;
  for (int rcp = 0; rcp < 3; ++rcp)
    for (int w : occ) {
      // a 256-thread block = 4 waves = one wave per SIMD; w blocks per CU = w waves per SIMD
      const int grid = n_cu * w;
      for (int rep = 0; rep < 2; ++rep) {   // first repetition warms the clocks
        CHECK(hipEventRecord(e0));
        if (rcp == 2) hipLaunchKernelGGL(k_fma<2>, dim3(grid), dim3(256), 0, 0, out, ticks, 1.0000001f, 1e-9f);
        else if (rcp) hipLaunchKernelGGL(k_fma<1>, dim3(grid), dim3(256), 0, 0, out, ticks, 1.0000001f, 1e-9f);
        else hipLaunchKernelGGL(k_fma<0>, dim3(grid), dim3(256), 0, 0, out, ticks, 1.0000001f, 1e-9f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
      }
      float ms = 0.0f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h((size_t)grid * 4);
      CHECK(hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      double sum = 0.0; unsigned long long mx = 0;
      for (auto t : h) { sum += (double)t; if (t > mx) mx = t; }
      const double mean = sum / (double)h.size();
      const double n_instr = (double)kIters * kRep * kAcc;
      // s_memtime ticks are shader cycles on gfx950 (MI355X_MICROARCH.md, cycle-constants table), so ticks ÷ instructions is the
      // cycles one wave needs per instruction, and ÷ waves per SIMD what the SIMD's issue port delivers; the launch's event
      // time gives the same in ns (launch overhead included) and, divided by the ticks, the clock the chip held.
      const double ns_per_instr_simd = (double)ms * 1e6 / (n_instr * w);
      const double cyc_wave = mean / n_instr, cyc_simd = cyc_wave / w;
      std::printf("%s{\"stream\": \"%s\", \"waves_per_simd\": %d, \"launch_ms\": %.4f, \"memtime_ticks_mean\": %.0f, \"memtime_ticks_max\": %llu, "
                  "\"cycles_per_instr_per_wave\": %.3f, \"cycles_per_instr_per_simd\": %.3f, \"cycles_per_instr_per_simd_slowest_wave\": %.3f, \"ns_per_instr_per_simd\": %.5f, \"implied_clock_ghz\": %.3f}",
                  first ? "" : ", ", rcp == 2 ? "v_pk_fma_f32" : (rcp ? "7 v_fma_f32 : 1 v_rcp_f32" : "v_fma_f32"), w, ms, mean, mx, cyc_wave, cyc_simd, (double)mx / n_instr / w, ns_per_instr_simd, (double)mx / ((double)ms * 1e6));
      first = false;
    }
  std::printf("]}\n");
  return 0;
}
