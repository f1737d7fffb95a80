// Shared pieces of the fused disentangled-attention kernels (forward and backward "A") for gfx950.
//
// Tile geometry: 64 "rows" x 64 "cols" per step (queries x keys in the forward, keys x queries in the backward).
// For such a pair the relative index idx(i-j) spans <= 127 consecutive rows of the position tables, so a 128-row WINDOW
// of PK and PQ (starting at r_lo = idx(min delta of the pair)) is staged in LDS.  Within the pair a 16-row MFMA group
// (16 queries against 64 keys, or 16 keys against 64 queries) only needs 79 consecutive window rows, so each bias GEMM
// is done on an 80-row sub-window (5 MFMA row tiles) at a wave-specific offset and stored as fp16 (T1/T2 tiles,
// |T| <= 65504 saturating): 2 x 11 KiB instead of 2 x 33 KiB fp32 -> 73 KiB LDS per workgroup, two workgroups per CU.
#pragma once
#include "fbl_common.h"

typedef _Float16 f16;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;

namespace attn {

constexpr int LT = 88;   // fp16 row stride of the T1/T2 tiles (80 used)
constexpr int LDV = 72;  // bf16 row stride of transposed-operand tiles ([64 d][64 + 8])
constexpr int TW = 5;    // MFMA row tiles per 80-row sub-window

__device__ __forceinline__ bf16x8 lds_frag(const char* base, int row, int chunk) {
  return *(const bf16x8*)(base + row * 128 + ((chunk ^ (row & 7)) << 4));  // [rows][8 x 16B] image, chunk ^= row&7
}
__device__ __forceinline__ void lds_put(char* base, int row, int chunk, bf16x8 v) {
  *(bf16x8*)(base + row * 128 + ((chunk ^ (row & 7)) << 4)) = v;
}
__device__ __forceinline__ f16x4 to_f16x4(f32x4 v) {
  f16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (f16)fminf(fmaxf(v[e], -65504.f), 65504.f);
  return o;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// T[col c][sub-window] = X_c . TAB[off + .]   for the 16 columns held as B-fragments xb0/xb1 (k-steps of d);
// tab: swizzled [128][64] bf16 LDS image of the table window; dst: fp16 tile, row stride LT, this lane's row = c.
__device__ __forceinline__ void bias_tile(const char* tab, int off, bf16x8 xb0, bf16x8 xb1, f16* dst_row_c, int c, int g) {
#pragma unroll
  for (int wt = 0; wt < TW; ++wt) {
    const int row = min(off + wt * 16 + c, 127);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(tab, row, g), xb0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(tab, row, 4 + g), xb1, acc, 0, 0, 0);
    *(f16x4*)(dst_row_c + wt * 16 + g * 4) = to_f16x4(acc);
  }
}

}  // namespace attn
