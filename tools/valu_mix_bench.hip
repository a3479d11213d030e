// valu_mix_bench.hip — issue cost of the instruction kinds the trace kernels' node visit is made of (gfx950).
//
// valu_issue_bench.hip measured the issue ceiling with v_fma_f32 only.  A node visit is 48 v_cvt_f32_ubyteN + 48 v_fma_f32 + 32 min/max +
// 8 x (v_cmp_le_f32, v_addc_co_u32) + selects and integer bookkeeping, so the question this answers is whether all of those cost one
// fma slot — they do not: see profiles/r03_valu_mix.json.  Each stream is 16 independent instructions of one kind per lane (no
// dependency stalls), 8 waves per SIMD on every CU (the occupancy of the trace kernels), timed with events over the launch.  Mixed
// streams (an fma next to another kind) show whether two kinds overlap or add.  Output: one JSON line with ns per wave64 instruction
// per SIMD and the cost relative to v_fma_f32.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_mix_bench tools/valu_mix_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kAcc = 16, kRep = 8, kIters = 2048;

// X(name, instructions per unit, asm text).  Operands: %0 float acc (rw), %1 uint acc (rw), %2 float a, %3 float b, %4 uint word, %5 64-bit pair acc
#define STREAMS(X) \
  X(fma, 1, "v_fma_f32 %0, %0, %2, %3") \
  X(mul, 1, "v_mul_f32 %0, %0, %2") \
  X(add, 1, "v_add_f32 %0, %0, %2") \
  X(sub, 1, "v_sub_f32 %0, %0, %2") \
  X(fmac, 1, "v_fmac_f32 %0, %2, %3") \
  X(mov, 1, "v_mov_b32 %1, %4") \
  X(fma_mix_f16src, 1, "v_fma_mix_f32 %0, %4, %2, %3 op_sel_hi:[1,0,0]") \
  X(fma_mix_f16src_hi, 1, "v_fma_mix_f32 %0, %4, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]") \
  X(pk_fma_f32, 1, "v_pk_fma_f32 %5, %5, %5, %5") \
  X(pk_mul_f32, 1, "v_pk_mul_f32 %5, %5, %5") \
  X(pk_add_f32, 1, "v_pk_add_f32 %5, %5, %5") \
  X(pk_add_f16, 1, "v_pk_add_f16 %1, %1, %4") \
  X(pk_fma_f16, 1, "v_pk_fma_f16 %1, %1, %4, %4") \
  X(cvt_f32_ubyte0, 1, "v_cvt_f32_ubyte0 %0, %4") \
  X(cvt_f32_ubyte3, 1, "v_cvt_f32_ubyte3 %0, %4") \
  X(cvt_f32_f16, 1, "v_cvt_f32_f16 %0, %4") \
  X(cvt_f32_i32, 1, "v_cvt_f32_i32 %0, %4") \
  X(cvt_f16_f32, 1, "v_cvt_f16_f32 %1, %2") \
  X(rcp, 1, "v_rcp_f32 %0, %0") \
  X(max, 1, "v_max_f32 %0, %0, %2") \
  X(max3, 1, "v_max3_f32 %0, %0, %2, %3") \
  X(min3, 1, "v_min3_f32 %0, %0, %2, %3") \
  X(med3, 1, "v_med3_f32 %0, %0, %2, %3") \
  X(cmp_to_sgpr, 1, "v_cmp_le_f32 s[20:21], %0, %2") \
  X(cmp_vcc_addc, 2, "v_cmp_le_f32 vcc, %2, %0\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc") \
  X(cndmask_sgpr, 1, "v_cndmask_b32 %1, %1, %4, s[22:23]") \
  X(add_u32, 1, "v_add_u32 %1, %1, %4") \
  X(and_b32, 1, "v_and_b32 %1, %1, %4") \
  X(xor_b32, 1, "v_xor_b32 %1, %1, %4") \
  X(lshlrev, 1, "v_lshlrev_b32 %1, 1, %1") \
  X(lshl_add, 1, "v_lshl_add_u32 %1, %1, 1, %4") \
  X(lshl_or, 1, "v_lshl_or_b32 %1, %1, 1, %4") \
  X(alignbit, 1, "v_alignbit_b32 %1, %1, %4, 31") \
  X(bfe, 1, "v_bfe_u32 %1, %4, 8, 8") \
  X(perm, 1, "v_perm_b32 %1, %1, %4, %4") \
  X(bcnt, 1, "v_bcnt_u32_b32 %1, %4, %1") \
  X(mbcnt_lo, 1, "v_mbcnt_lo_u32_b32 %1, %4, %1") \
  X(mad_u32_u24, 1, "v_mad_u32_u24 %1, %1, %4, %4") \
  X(mul_lo_u32, 1, "v_mul_lo_u32 %1, %1, %4") \
  X(readlane, 1, "v_readlane_b32 s20, %1, 3") \
  X(readfirstlane, 1, "v_readfirstlane_b32 s20, %1") \
  X(bitop3, 1, "v_bitop3_b32 %1, %1, %4, %4 bitop3:0x80") \
  X(mix_fma_cvt_1_1, 2, "v_fma_f32 %0, %0, %2, %3\n\tv_cvt_f32_ubyte0 %5, %4") \
  X(mix_fma_cvt_1_2, 3, "v_fma_f32 %0, %0, %2, %3\n\tv_cvt_f32_ubyte0 %5, %4\n\tv_and_b32 %1, %1, %4") \
  X(mix_fma_cvt_2_1, 3, "v_fma_f32 %0, %0, %2, %3\n\tv_mul_f32 %5, %2, %3\n\tv_and_b32 %1, %1, %4") \
  X(mix_fma_max3_1_1, 2, "v_fma_f32 %0, %0, %2, %3\n\tv_max3_f32 %5, %5, %2, %3") \
  X(mix_fmamix_max3_1_1, 2, "v_fma_mix_f32 %0, %4, %2, %3 op_sel_hi:[1,0,0]\n\tv_max3_f32 %5, %5, %2, %3") \
  X(mul_sdwa_b0, 1, "v_mul_f32_sdwa %0, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD") \
  X(mul_sdwa_b2, 1, "v_mul_f32_sdwa %0, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD") \
  X(add_sdwa_b1, 1, "v_add_f32_sdwa %0, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
  X(mov_sdwa_b3, 1, "v_mov_b32_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3") \
  X(mul_clamp, 1, "v_mul_f32_e64 %0, %0, %2 clamp") \
  X(mul_e64, 1, "v_mul_f32_e64 %0, %0, %2") \
  X(fma_clamp, 1, "v_fma_f32 %0, %0, %2, %3 clamp") \
  X(add_neg_abs, 1, "v_add_f32_e64 %0, -%0, |%2|") \
  X(or_b32, 1, "v_or_b32 %1, %1, %4") \
  X(sub_u32, 1, "v_sub_u32 %1, %1, %4") \
  X(add3_u32, 1, "v_add3_u32 %1, %1, %4, %4") \
  X(not_b32, 1, "v_not_b32 %1, %4") \
  X(lshl_add_u64, 1, "v_lshl_add_u64 %9, %9, 4, %9") \
  X(mix_and_cvt_1_1, 2, "v_and_b32 %1, %1, %4\n\tv_cvt_f32_ubyte0 %5, %4") \
  X(mix_and_fma_1_1, 2, "v_and_b32 %1, %1, %4\n\tv_fma_f32 %0, %0, %2, %3") \
  X(mix_fma_cvt_3_1, 4, "v_fma_f32 %0, %0, %2, %3\n\tv_mul_f32 %5, %2, %3\n\tv_add_f32 %6, %2, %3\n\tv_cvt_f32_ubyte0 %5, %4") \
  X(slot_v3_allfast_20, 20, \
    "v_mul_f32_sdwa %0, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\tv_mul_f32_sdwa %5, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
    "v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max_f32 %0, %0, %5\n\t" \
    "v_mul_f32_sdwa %5, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max3_f32 %0, %0, %5, 0\n\t" \
    "v_mul_f32_sdwa %5, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\tv_fma_f32 %5, %5, %2, %3\n\t" \
    "v_mul_f32_sdwa %6, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min_f32 %5, %6, %5\n\t" \
    "v_mul_f32_sdwa %6, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min3_f32 %5, %5, %6, %3\n\t" \
    "v_sub_f32 %5, %0, %5\n\tv_mul_f32 %5, %5, %2\n\tv_mul_f32_e64 %5, %5, %2 clamp\n\tv_fma_f32 %0, %5, %3, %0") \
  X(slot_v3_2cvt_20, 20, \
    "v_cvt_f32_ubyte0 %0, %4\n\tv_mul_f32_sdwa %5, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
    "v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max_f32 %0, %0, %5\n\t" \
    "v_mul_f32_sdwa %5, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max3_f32 %0, %0, %5, 0\n\t" \
    "v_cvt_f32_ubyte3 %5, %4\n\tv_fma_f32 %5, %5, %2, %3\n\t" \
    "v_mul_f32_sdwa %6, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min_f32 %5, %6, %5\n\t" \
    "v_mul_f32_sdwa %6, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min3_f32 %5, %5, %6, %3\n\t" \
    "v_sub_f32 %5, %0, %5\n\tv_mul_f32 %5, %5, %2\n\tv_mul_f32_e64 %5, %5, %2 clamp\n\tv_fma_f32 %0, %5, %3, %0") \
  X(slot_v3_cmp_18, 18, \
    "v_mul_f32_sdwa %0, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\tv_mul_f32_sdwa %5, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\t" \
    "v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max_f32 %0, %0, %5\n\t" \
    "v_mul_f32_sdwa %5, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max3_f32 %0, %0, %5, 0\n\t" \
    "v_mul_f32_sdwa %5, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n\tv_fma_f32 %5, %5, %2, %3\n\t" \
    "v_mul_f32_sdwa %6, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min_f32 %5, %6, %5\n\t" \
    "v_mul_f32_sdwa %6, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min3_f32 %5, %5, %6, %3\n\t" \
    "v_cmp_le_f32 vcc, %0, %5\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc") \
  X(slot_B_hitfast_20, 20, \
    "v_cvt_f32_ubyte0 %0, %4\n\tv_cvt_f32_ubyte1 %5, %4\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max_f32 %0, %0, %5\n\t" \
    "v_cvt_f32_ubyte2 %5, %4\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max3_f32 %0, %0, %5, 0\n\t" \
    "v_cvt_f32_ubyte3 %5, %4\n\tv_fma_f32 %5, %5, %2, %3\n\tv_cvt_f32_ubyte0 %6, %1\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min_f32 %5, %6, %5\n\t" \
    "v_cvt_f32_ubyte1 %6, %1\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min3_f32 %5, %5, %6, %3\n\t" \
    "v_sub_f32 %5, %5, %0\n\tv_and_b32 %5, 0x80000000, %5\n\tv_or_b32 %5, 1.0, %5\n\tv_fma_f32 %0, %5, %3, %0") \
  X(slot_C_half_fastconv_26, 26, \
    "v_and_b32 %0, 0xff, %4\n\tv_or_b32 %0, 0x4b000000, %0\n\tv_sub_f32 %0, %0, %2\n\tv_and_b32 %5, 0xff00, %4\n\tv_or_b32 %5, 0x4b000000, %5\n\tv_sub_f32 %5, %5, %2\n\t" \
    "v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max_f32 %0, %0, %5\n\t" \
    "v_cvt_f32_ubyte2 %5, %4\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max3_f32 %0, %0, %5, 0\n\t" \
    "v_cvt_f32_ubyte3 %5, %4\n\tv_fma_f32 %5, %5, %2, %3\n\tv_and_b32 %6, 0xff, %1\n\tv_or_b32 %6, 0x4b000000, %6\n\tv_sub_f32 %6, %6, %2\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min_f32 %5, %6, %5\n\t" \
    "v_cvt_f32_ubyte1 %6, %1\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min3_f32 %5, %5, %6, %3\n\t" \
    "v_sub_f32 %5, %5, %0\n\tv_and_b32 %5, 0x80000000, %5\n\tv_or_b32 %5, 1.0, %5\n\tv_fma_f32 %0, %5, %3, %0") \
  X(slot_D_all_fastconv_32, 32, \
    "v_and_b32 %0, 0xff, %4\n\tv_or_b32 %0, 0x4b000000, %0\n\tv_sub_f32 %0, %0, %2\n\tv_and_b32 %5, 0xff00, %4\n\tv_or_b32 %5, 0x4b000000, %5\n\tv_sub_f32 %5, %5, %2\n\t" \
    "v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max_f32 %0, %0, %5\n\t" \
    "v_and_b32 %5, 0xff, %1\n\tv_or_b32 %5, 0x4b000000, %5\n\tv_sub_f32 %5, %5, %2\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max3_f32 %0, %0, %5, 0\n\t" \
    "v_and_b32 %5, 0xff00, %1\n\tv_or_b32 %5, 0x4b000000, %5\n\tv_sub_f32 %5, %5, %2\n\tv_fma_f32 %5, %5, %2, %3\n\tv_and_b32 %6, 0xff, %1\n\tv_or_b32 %6, 0x4b000000, %6\n\tv_sub_f32 %6, %6, %2\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min_f32 %5, %6, %5\n\t" \
    "v_and_b32 %6, 0xff00, %4\n\tv_or_b32 %6, 0x4b000000, %6\n\tv_sub_f32 %6, %6, %2\n\tv_fma_f32 %6, %6, %2, %3\n\tv_min3_f32 %5, %5, %6, %3\n\t" \
    "v_sub_f32 %5, %5, %0\n\tv_and_b32 %5, 0x80000000, %5\n\tv_or_b32 %5, 1.0, %5\n\tv_fma_f32 %0, %5, %3, %0") \
  X(mix_F_S_2_1, 3, "v_fma_f32 %0, %0, %2, %3\n\tv_and_b32 %1, %1, %4\n\tv_cvt_f32_ubyte0 %5, %4") \
  X(mix_F_S_4_1, 5, "v_fma_f32 %0, %0, %2, %3\n\tv_and_b32 %1, %1, %4\n\tv_sub_f32 %6, %2, %3\n\tv_or_b32 %7, %4, %4\n\tv_cvt_f32_ubyte0 %5, %4") \
  X(bfi, 1, "v_bfi_b32 %1, %4, %1, %4") \
  X(cndmask_vcc_e32_b, 1, "v_cndmask_b32_e32 %1, %1, %4, vcc") \
  X(slot_now_18, 18, \
    "v_cvt_f32_ubyte0 %0, %4\n\tv_cvt_f32_ubyte1 %5, %4\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max_f32 %0, %0, %5\n\t" \
    "v_cvt_f32_ubyte2 %5, %4\n\tv_fma_f32 %5, %5, %2, %3\n\tv_max3_f32 %0, %0, %5, 0\n\t" \
    "v_cvt_f32_ubyte3 %5, %4\n\tv_fma_f32 %5, %5, %2, %3\n\tv_cvt_f32_ubyte0 %0, %1\n\tv_fma_f32 %0, %0, %2, %3\n\tv_min_f32 %5, %0, %5\n\t" \
    "v_cvt_f32_ubyte1 %0, %1\n\tv_fma_f32 %0, %0, %2, %3\n\tv_min3_f32 %5, %5, %0, %3\n\t" \
    "v_cmp_le_f32 vcc, %0, %5\n\tv_addc_co_u32 %1, vcc, %1, %1, vcc") \
  X(slot_new_12, 12, \
    "v_fma_mix_f32 %0, %4, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %4, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_max_f32 %0, %0, %5\n\t" \
    "v_fma_mix_f32 %5, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_max3_f32 %0, %0, %5, 0\n\t" \
    "v_fma_mix_f32 %5, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %6, %4, %3, %2 op_sel_hi:[1,0,0]\n\tv_min_f32 %5, %6, %5\n\t" \
    "v_fma_mix_f32 %6, %4, %3, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_min3_f32 %5, %5, %6, %3\n\t" \
    "v_sub_f32 %5, %5, %0\n\tv_alignbit_b32 %1, %1, %5, 31") \
  X(bpermute, 1, "ds_bpermute_b32 %1, %4, %1\n\ts_waitcnt lgkmcnt(6)") \
  X(ds_read_b32, 1, "ds_read_b32 %1, %7\n\ts_waitcnt lgkmcnt(6)") \
  X(ds_read_b128, 1, "ds_read_b128 %8, %7\n\ts_waitcnt lgkmcnt(6)")

enum {
#define X(name, n, text) S_##name,
  STREAMS(X)
#undef X
  S_N
};
static const char* kNames[S_N] = {
#define X(name, n, text) #name,
  STREAMS(X)
#undef X
};
static const int kInstr[S_N] = {
#define X(name, n, text) n,
  STREAMS(X)
#undef X
};

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int S>
__global__ __launch_bounds__(256, 8) void k_stream(float* out, float a, float b, unsigned w0) {
  __shared__ float lds[2048];
  float acc[kAcc]; unsigned iacc[kAcc]; float t5[kAcc]; float t6 = 0.0f; f4 q = {0, 0, 0, 0}; unsigned long long a64 = threadIdx.x;
  unsigned wq = (w0 + threadIdx.x) | 0x3c003c00u;     // two finite halves when read as f16
  unsigned laddr = (threadIdx.x & 63u) * 16u;
#pragma unroll
  for (int i = 0; i < kAcc; ++i) { acc[i] = (float)(threadIdx.x + i); iacc[i] = threadIdx.x * 4u + (unsigned)i; t5[i] = 1.0f; }
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = a;
  __syncthreads();
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int rep = 0; rep < kRep; ++rep)
#pragma unroll
    for (int i = 0; i < kAcc; ++i) {
      switch (S) {
#define X(name, n, text) \
        case S_##name: \
          if (S_##name == S_pk_fma_f32 || S_##name == S_pk_mul_f32 || S_##name == S_pk_add_f32) { f2 p = {acc[i], t5[i]}; asm volatile(text : "+v"(acc[i]), "+v"(iacc[i]), "+v"(a), "+v"(b), "+v"(w0), "+v"(p), "+v"(t6), "+v"(w0), "+v"(q), "+v"(a64)); acc[i] = p.x; t5[i] = p.y; } \
          else asm volatile(text : "+v"(acc[i]), "+v"(iacc[i]), "+v"(a), "+v"(b), "+v"(wq), "+v"(t5[i]), "+v"(t6), "+v"(laddr), "+v"(q), "+v"(a64) : : "vcc", "s20", "s21", "s22", "s23"); \
          break;
        STREAMS(X)
#undef X
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)");
  float s = t6 + q.x + (float)a64;
#pragma unroll
  for (int i = 0; i < kAcc; ++i) s += acc[i] + (float)iacc[i] + t5[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + lds[(threadIdx.x * 7) & 2047];
}

static double g_ns_fma = 1.0;
template <int S> void run(int n_cu, float* out, hipEvent_t e0, hipEvent_t e1) {
  const int grid = n_cu * 8;   // 256-thread block = one wave per SIMD; 8 blocks per CU = 8 waves per SIMD
  float ms = 0.0f;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_stream<S>, dim3(grid), dim3(256), 0, 0, out, 1.0000001f, 1e-9f, 0x01020304u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
  }
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double units = (double)kIters * kRep * kAcc;
  const double ns_instr = (double)ms * 1e6 / (units * 8.0) / kInstr[S];
  if (S == S_fma) g_ns_fma = ns_instr;
  std::printf("%s{\"stream\": \"%s\", \"instr_per_unit\": %d, \"launch_ms\": %.4f, \"ns_per_instr_per_simd\": %.4f, \"fma_slots_per_instr\": %.3f}", S == 0 ? "" : ", ", kNames[S], kInstr[S], ms, ns_instr, ns_instr / g_ns_fma);
  std::fflush(stdout);
}
template <int S> struct RunAll { static void go(int n_cu, float* out, hipEvent_t e0, hipEvent_t e1) { RunAll<S - 1>::go(n_cu, out, e0, e1); run<S>(n_cu, out, e0, e1); } };
template <> struct RunAll<-1> { static void go(int, float*, hipEvent_t, hipEvent_t) {} };

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  float* out;
  CHECK(hipMalloc(&out, (size_t)n_cu * 8 * 256 * sizeof(float)));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::printf("{\"device\": \"%s\", \"n_cu\": %d, \"waves_per_simd\": 8, \"runs\": [", prop.gcnArchName, n_cu);
  RunAll<S_N - 1>::go(n_cu, out, e0, e1);
  std::printf("]}\n");
  return 0;
}
