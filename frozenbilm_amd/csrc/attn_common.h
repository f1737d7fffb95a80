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
  f32x4 m;
#pragma unroll
  for (int e = 0; e < 4; ++e) m[e] = __builtin_amdgcn_fmed3f(v[e], -65504.f, 65504.f);
  return __builtin_convertvector(m, f16x4);
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// Workgroup -> (tile x, head h, sample b) for a 1-D grid of nx * nh * B workgroups.
//  * the nx tiles of one (sample, head) share their K/V (or Q/dO, or transposed-operand) rows: they are placed on ONE XCD
//    (the dispatcher hands block L to XCD L % 8 -- observed, used for L2 affinity only) and dispatched back to back, so
//    those rows are fetched into one L2 once instead of into up to nx of them;
//  * samples are walked in the caller's `border` order (longest first): the work per sample grows with klen^2 and the
//    grid is several waves deep, so the long samples must not be the last ones dispatched.
struct WgCoord { int x, h, b; };
#ifdef FBL_DEBUG_SWITCHES
__device__ int g_attn_plainmap;  // measurement switch (debug builds only): 1 = plain b-major mapping
__device__ int g_attn_dbgbits;   // FBL_ATTN_DBG (backward kernels: 64 = no zero fill of G^T, 128 = no G^T stores)
#define FBL_ATTN_PLAINMAP g_attn_plainmap
#define FBL_ATTN_DBGBITS g_attn_dbgbits
static inline void attn_debug_init() {
  static bool once = false;
  if (!once) {
    const int v = FBL_ENV_INT("FBL_ATTN_PLAINMAP", 0);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_plainmap), &v, sizeof(v));
    const int d = FBL_ENV_INT("FBL_ATTN_DBG", 0);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbgbits), &d, sizeof(d));
    once = true;
  }
}
#else
#define FBL_ATTN_PLAINMAP 0
#define FBL_ATTN_DBGBITS 0
static inline void attn_debug_init() {}
#endif
__device__ __forceinline__ WgCoord wg_coord(int nx, int nh, int B, const int32_t* border) {
  const int L = blockIdx.x, ng = nh * B;
  int G, x;
  if ((ng & 7) == 0 && !FBL_ATTN_PLAINMAP) {
    const int s = L >> 3;
    G = (s / nx) * 8 + (L & 7);
    x = s % nx;
  } else {
    G = L / nx;
    x = L % nx;
  }
  const int bs = G / nh;
  WgCoord c;
  c.x = x;
  c.h = G - bs * nh;
  c.b = border ? border[bs] : bs;
  return c;
}

// ds_read_b64_tr_b16: the 16 lanes of a group each pass the address of 4 contiguous 16-bit elements -- together a
// [4 rows][16 columns] block of a row-major LDS image (lane i: row i/4, columns (i%4)*4..+3) -- and lane i receives
// column i of that block, i.e. 4 consecutive rows of one column (lane mapping probed in tools/probe/tr16_probe.hip).
typedef __attribute__((ext_vector_type(4))) short tr16x4;
__device__ __forceinline__ tr16x4 lds_tr16(const bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr16x4*)p);
}

constexpr float LOG2E = 1.4426950408889634f;

// Buffer addressing for kernels that stream several tensors per loop iteration: a 128-bit descriptor of a WAVE-UNIFORM base in
// scalar registers + a 32-bit per-lane byte offset + a scalar byte offset (the per-iteration part) -- no 64-bit per-lane
// pointers (2 VGPRs and a 64-bit add per stream and iteration otherwise).  The base passes through readfirstlane so that the
// compiler can prove it uniform (anything derived from blockIdx-dependent LOADS is "divergent" to it and every access would
// be wrapped in a waterfall loop).  Offsets must stay below 4 GiB from the base.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* p) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0xffffffffu, 0x00020000);
}
__device__ __forceinline__ bf16x8 buf_ld16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void buf_st16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, bf16x8 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}

// Relative-index table padded to the 64-tile grid: idxp[t] = relidx[clamp(t - (Sp - S), 0, 2S-2)] with t = i - j + Sp - 1,
// so that every (i, j) of a tile (valid or padding) indexes it without a clamp.  2*Sp - 1 <= 1023 entries.
__device__ __forceinline__ void load_idx_padded(int16_t* idxp, const int16_t* relidx, int S, int Sp, int tid, int nthr) {
  for (int t = tid; t < 2 * Sp - 1; t += nthr) idxp[t] = relidx[clampi(t - (Sp - S), 0, 2 * S - 2)];
}

// T[col c][sub-window] = X_c . TAB[off + .]   for the 16 columns held as B-fragments xb0/xb1 (k-steps of d);
// tab: swizzled [128][64] bf16 LDS image of the table window; dst: fp16 tile, row stride LT, this lane's row = c.
// colvalid = false writes -inf for the whole row (a masked key: its score becomes -inf without a per-element test).
__device__ __forceinline__ void bias_tile(const char* tab, int off, bf16x8 xb0, bf16x8 xb1, f16* dst_row_c, int c, int g,
                                          bool colvalid = true) {
  const f16 ninf = (f16)(-INFINITY);
#pragma unroll
  for (int wt = 0; wt < TW; ++wt) {
    const int row = min(off + wt * 16 + c, 127);
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(tab, row, g), xb0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(tab, row, 4 + g), xb1, acc, 0, 0, 0);
    const f16x4 t = to_f16x4(acc);
    *(f16x4*)(dst_row_c + wt * 16 + g * 4) = colvalid ? t : (f16x4){ninf, ninf, ninf, ninf};
  }
}

// ---- attention-probability dropout RNG (shared by the forward and the backward-A kernels).
// One 32-bit counter hash serves a 2x2 block of (query, key) pairs: four 16-bit fields, field (i&1)*2 + (j&1), each
// compared with a 16-bit threshold (p quantised to 1/65536; the kept values are scaled by the exact 1/(1-p_q)).
// 32-bit integer multiplies are quarter rate on CDNA, so this costs ~6x fewer issue slots than one 64-bit mix per
// element.  Keys are derived per (seed, batch*head) with the common 64-bit mixer.
struct DropKey {
  uint32_t k1, k2, thr16;
  float inv_keep;
};
__device__ __forceinline__ DropKey attn_drop_key(uint64_t seed, int bh, float p) {
  DropKey k;
  k.k1 = fbl_hash(seed, 2ull * (uint64_t)bh);
  k.k2 = fbl_hash(seed, 2ull * (uint64_t)bh + 1ull);
  const float t = p * 65536.f + 0.5f;
  k.thr16 = p > 0.f ? (uint32_t)fminf(t, 65535.f) : 0u;
  k.inv_keep = 65536.f / (65536.f - (float)k.thr16);
  return k;
}
// the two words of block (bi, bj) = (i >> 1, j >> 1); Sp2 = Sp / 2
__device__ __forceinline__ void attn_drop_block(const DropKey& k, int bi, int bj, int Sp2, uint32_t* x_out, uint32_t* y_out) {
  uint32_t x = (uint32_t)(bi * Sp2 + bj) * 0x9E3779B1u + k.k1;
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  uint32_t y = (x ^ k.k2) * 0x2c1b3c6du;
  y ^= y >> 15;
  *x_out = x;
  *y_out = y;
}
// keep factor (0 or 1/(1-p)) of the element with parities (pi, pj) inside its block
__device__ __forceinline__ float attn_drop_keep(const DropKey& k, uint32_t x, uint32_t y, int pi, int pj) {
  const uint32_t wsel = pi ? y : x;
  const uint32_t f = pj ? (wsel >> 16) : (wsel & 0xffffu);
  return f >= k.thr16 ? k.inv_keep : 0.f;
}

}  // namespace attn
