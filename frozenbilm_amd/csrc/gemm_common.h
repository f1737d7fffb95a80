// Pieces shared by the bf16 MFMA GEMM kernels of libfbl (gemm.hip: 2-stage tiles; gemm8.hip: 8-phase 256x256 tile):
// the argument block, the LDS-DMA helper, the XCD-aware tile mapping and the fused epilogue.
#pragma once
#include <type_traits>

#include "fbl_common.h"
#include "../../include/fbl.h"

namespace fblgemm {

constexpr int BK = 64;
constexpr int NXCD = 8;

struct GemmArgs {
  const bf16* A;
  const bf16* B;
  long lda, ldb;
  int M, N, K;
  const float* bias;      // [N] or null
  const float* rowscale;  // [M] or null: multiplies (alpha*acc + bias) per row before the activation
  float alpha;
  int act, aux_kind;
  const void* aux;
  long ld_aux;
  float* out_f32;
  bf16* out_bf16;
  bf16* out_pre;  // pre-activation copy (bf16) or null
  long ldc;
  long sA, sB, sC, sAux, sBias;  // batch strides in elements
  int splitk;
  int k8_per;  // 8-phase kernel with splitk > 1: K-tiles per slice (even, >= 4; the last slice takes what is left, even too)
  int tiles_m, tiles_n;
  float* ws;  // split-K partials [batch][splitk][M][Nw] (Nw = N rounded up to 4) or null -> atomicAdd into out_f32
  int Nw;
  const int32_t* kskip_len;  // optional: K is made of samples of kskip_steps k-steps; step j of sample b is all zero when 64*j >= kskip_len[b]
  int kskip_steps;
  const uint32_t* kskip_tilemask;  // optional, with kskip_len: per 64-wide k-step, bit t set <=> rows [128t, 128t+128) of A can be non-zero
  long a_kblk;  // 0: A rows are K-contiguous.  >0: A is stored in 32-wide k blocks: A[m][k] at m*lda + (k/32)*a_kblk + k%32
  // post-activation dropout of the epilogue (adapter bottleneck, model/adapter.py:39-41): element (m, n) is keyed by
  // (drop_seed, m*drop_ld + n) exactly like fbl_dropout_bf16 on the [M, drop_ld] output; drop_thresh == 0 -> off
  uint64_t drop_seed;
  const uint64_t* drop_seed_dev;  // optional device word added to drop_seed (fbl_seed)
  uint32_t drop_thresh;
  float drop_inv_keep;
  long drop_ld;
  long drop_row0;  // row offset added to m in the dropout key (the rows of a split launch keep their global keys)
  // Second output segment (fbl_dense_adapter_down_fwd): columns n >= seg_n are NOT part of C but the adapter bottleneck
  // z[m, n - seg_n] = dropout(relu(v)) (bf16, row stride seg_ld; dropout = drop_* keyed by m*seg_ld + n - seg_n), whatever
  // ACT / AUX the kernel was instantiated with.  seg_n == 0: off.  seg_n % 4 == 0.
  int seg_n;
  bf16* seg_out;
  long seg_ld;
  // Adapter tail (FBL_AUX_ADAPTER_TAIL, fbl_adapter_up_resid_fwd): out_f32 = dropout(alpha*acc + bias + aux_bf16) + resid, the
  // residual either plain (r_stats == null: r_t[m,n]) or in LayerNorm-normalised form ((r_t - mean[m]) * rstd[m] * r_gamma[n]
  // + r_beta[n]) * r_rowmask[m]; dropout = drop_* keyed by (m + drop_row0) * drop_ld + n like fbl_ln_fwd
  const float* r_t;
  long ld_r;
  const float* r_stats;
  const float* r_gamma;
  const float* r_beta;
  const int32_t* r_rowmask;
};

__device__ __forceinline__ void glds16(const void* g, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Tile mapping: XCD-aware (block b runs on XCD b%8; bijective remap so each XCD owns a contiguous run of tiles) +
// grouped along M so neighbours share the B panel in their XCD's L2.
__device__ __forceinline__ void tile_of_block(int pid, int tiles_m, int tiles_n, int* tm, int* tn) {
  const int ntiles = tiles_m * tiles_n;
  {
    const int q = ntiles / NXCD, r = ntiles % NXCD;
    const int xcd = pid % NXCD, idx = pid / NXCD;
    pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  constexpr int GROUP_M = 8;
  const int width = GROUP_M * tiles_n;
  const int group = pid / width;
  const int first_m = group * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  *tm = first_m + (pid % width) % gsz;
  *tn = (pid % width) / gsz;
}

// ---- epilogue.  Accumulators (lane: C[m = ..+mi*16+(lane&15)][4 consecutive n]) go through a wave-private LDS tile
// [64 m][64 n] so that global traffic is row-contiguous: one wave instruction covers 4 rows x 256 B (fp32) /
// 128 B (bf16) of C and of the aux operand, instead of 16 rows x 64/32 B.
//   acc[ni][mi]: 16x16 tile ni (of 4) along n, mi (of MI) along m.  The wave's rows leave in slabs of 64: slab `half`
//   starts at global row m_first + half*m_slab_stride; this lane's 4 staged columns (lane&15)*4.. map to the global
//   columns n4..n4+3.
// Epilogue of the tile columns that belong to the second output segment of a merged dense + adapter-down GEMM
// (GemmArgs::seg_n): z[m, n - seg_n] = dropout(relu(alpha*acc + bias[n])), bf16.  Kept out of gemm_epilogue's row loop on
// purpose: that loop is fully unrolled, and a larger body makes the compiler give up unrolling the slab loop around it
// (the accumulator array then lives in scratch memory).
template <int MI>
__device__ __forceinline__ void gemm_epilogue_seg(const GemmArgs& g, char* smem, int wave, int lane, f32x4 (&acc)[4][MI],
                                               int m_first, int m_slab_stride, int n4, int slab_rows) {
  constexpr int LDW = 68;
  float* stage = (float*)smem + wave * (64 * LDW);
  const int frow = lane & 15, fg = lane >> 4;
  const int er = lane >> 4, ec = (lane & 15) * 4;
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (n4 + r < g.N) ? g.bias[n4 + r] : 0.f;
  }
  const int nz = n4 - g.seg_n;
  const bool vec = (n4 + 3 < g.N) && ((g.seg_ld & 3) == 0);
  const uint64_t dseed = g.drop_thresh ? fbl_seed(g.drop_seed, g.drop_seed_dev) : 0;
#pragma unroll
  for (int half = 0; half < (MI + 3) / 4; ++half) {
    const int cnt = (MI - half * 4 < 4) ? MI - half * 4 : 4;
    const int mslab = m_first + half * m_slab_stride;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      if (half * 4 + mi < MI) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          *(f32x4*)(stage + (mi * 16 + frow) * LDW + ni * 16 + fg * 4) = acc[ni][half * 4 + mi];
      }
    if (n4 < g.N) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        if (it >= cnt * 4) break;
        const int row = it * 4 + er;
        const int m = mslab + row;
        if (m >= g.M || row >= slab_rows) continue;
        const f32x4 a4 = *(const f32x4*)(stage + row * LDW + ec);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaxf(a4[r] * g.alpha + bv[r], 0.f);
          if (g.drop_thresh)
            v[r] *= fbl_dropout_scale(dseed, (uint64_t)(m + g.drop_row0) * (uint64_t)g.seg_ld + (uint64_t)(nz + r), g.drop_thresh, g.drop_inv_keep);
        }
        bf16* zp = g.seg_out + (long)m * g.seg_ld + nz;
        if (vec) {
          *(bf16x4*)zp = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n4 + r < g.N) zp[r] = f2bf(v[r]);
        }
      }
    }
  }
}


// Epilogue of the adapter tail (GemmArgs::r_t).  The per-row LayerNorm statistics of the residual travel through the four
// padding floats of each staged row (columns 64..67 of the wave-private tile), so they cost no registers in the row loop.
template <int MI>
__device__ __forceinline__ void gemm_epilogue_tail(const GemmArgs& g, char* smem, int wave, int lane, f32x4 (&acc)[4][MI],
                                                int m_first, int m_slab_stride, int n4, int slab_rows) {
  constexpr int LDW = 68;
  float* stage = (float*)smem + wave * (64 * LDW);
  const int frow = lane & 15, fg = lane >> 4;
  const int er = lane >> 4, ec = (lane & 15) * 4;
  const bool live = n4 + 3 < g.N;       // (N % 4 == 0 is checked on the host)
  const int nc = live ? n4 : 0;         // loads of dead lanes stay in bounds
  float bv[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {1.f, 1.f, 1.f, 1.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
  if (g.bias) {
    const f32x4 t = *(const f32x4*)(g.bias + nc);
    bv[0] = t[0]; bv[1] = t[1]; bv[2] = t[2]; bv[3] = t[3];
  }
  const bool normed = g.r_stats != nullptr;
  const uint64_t dseed = g.drop_thresh ? fbl_seed(g.drop_seed, g.drop_seed_dev) : 0;
  if (normed) {
    const f32x4 t = *(const f32x4*)(g.r_gamma + nc), u = *(const f32x4*)(g.r_beta + nc);
    gg[0] = t[0]; gg[1] = t[1]; gg[2] = t[2]; gg[3] = t[3];
    bb[0] = u[0]; bb[1] = u[1]; bb[2] = u[2]; bb[3] = u[3];
  }
#pragma unroll
  for (int half = 0; half < (MI + 3) / 4; ++half) {
    const int cnt = (MI - half * 4 < 4) ? MI - half * 4 : 4;
    const int mslab = m_first + half * m_slab_stride;
    bf16x4 xa[16];
    f32x4 ra[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (it < cnt * 4) {
        const long m = min(mslab + it * 4 + er, g.M - 1);
        xa[it] = *(const bf16x4*)((const bf16*)g.aux + m * g.ld_aux + nc);
        ra[it] = *(const f32x4*)(g.r_t + m * g.ld_r + nc);
      }
    }
    float mean = 0.f, rstd = 1.f, rm = 1.f;  // statistics of row mslab + lane
    if (normed) {
      const long m = min(mslab + lane, g.M - 1);
      const f32x2 st = *(const f32x2*)(g.r_stats + 2 * m);
      mean = st[0];
      rstd = st[1];
      if (g.r_rowmask) rm = (float)g.r_rowmask[m];
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      if (half * 4 + mi < MI) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          *(f32x4*)(stage + (mi * 16 + frow) * LDW + ni * 16 + fg * 4) = acc[ni][half * 4 + mi];
      }
    *(f32x4*)(stage + lane * LDW + 64) = (f32x4){mean, rstd, rm, 0.f};
    // every global load of the slab has landed before its first store (see gemm_epilogue)
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      asm volatile("" : "+v"(xa[it]));
      asm volatile("" : "+v"(ra[it]));
    }
    if (live) {
      const long er_c = (long)(mslab + er) * g.ldc + n4;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        if (it >= cnt * 4) break;
        const int row = it * 4 + er;
        const int m = mslab + row;
        if (m >= g.M || row >= slab_rows) continue;
        const f32x4 a4 = *(const f32x4*)(stage + row * LDW + ec);
        const f32x4 st = *(const f32x4*)(stage + row * LDW + 64);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = a4[r] * g.alpha + bv[r] + bf2f(xa[it][r]);
          if (g.drop_thresh)
            v[r] *= fbl_dropout_scale(dseed, (uint64_t)(m + g.drop_row0) * (uint64_t)g.drop_ld + (uint64_t)(n4 + r), g.drop_thresh, g.drop_inv_keep);
          v[r] += normed ? ((ra[it][r] - st[0]) * st[1] * gg[r] + bb[r]) * st[2] : ra[it][r];
        }
        *(f32x4*)(g.out_f32 + er_c + (long)(it * 4) * g.ldc) = (f32x4){v[0], v[1], v[2], v[3]};
      }
    }
  }
}

//   slab_rows (<= 64): rows of a slab that belong to this wave (the 224-row configuration of the 8-phase kernel gives
//   its second wave row 48 of them); rows beyond it are not stored.
// DBG (measurement builds only): bit 0 = everything but the global stores (values kept alive), bit 1 = the stores of all
// tile rows alias the first 256 rows of C (the output stays L2-resident: store issue without HBM write traffic)
// SHORT48: slab_rows may be 48 (second wave row of the 224-row 8-phase tile): gets its own branch-free row loop.
// TPS: 16-row tiles per slab that carry results (4; 2 for the 128-row 8-phase tile, whose waves own 32 rows of each half).
template <int ACT, int AUX, bool SPLITK, int MI, int DBG = 0, bool SHORT48 = false, int TPS = 4>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& g, char* smem, int wave, int lane, f32x4 (&acc)[4][MI],
                                              int m_first, int m_slab_stride, int n4, int batch, int ks,
                                              int slab_rows = 64) {
  if constexpr (AUX == FBL_AUX_ADAPTER_TAIL) {
    gemm_epilogue_tail<MI>(g, smem, wave, lane, acc, m_first, m_slab_stride, n4, slab_rows);
    return;
  }
  if (!SPLITK && g.seg_n > 0 && n4 >= g.seg_n) {  // (uniform per workgroup: seg_n is a multiple of the tile width)
    gemm_epilogue_seg<MI>(g, smem, wave, lane, acc, m_first, m_slab_stride, n4, slab_rows);
    return;
  }
  constexpr int LDW = 68;  // floats per staged row (64 + 4: conflict-free b128 writes)
  float* stage = (float*)smem + wave * (64 * LDW);
  const int frow = lane & 15, fg = lane >> 4;
  const float* bias = g.bias ? g.bias + (long)batch * g.sBias : nullptr;
  const long cbase = (long)batch * g.sC;
  const long xbase = (long)batch * g.sAux;
  const bool vec_ok = ((g.ldc & 3) == 0);
  const int er = lane >> 4, ec = (lane & 15) * 4;  // this lane's row-within-quad and staged column offset
  float bv[4] = {0.f, 0.f, 0.f, 0.f};
  if (bias && (!SPLITK || (ks == 0 && !g.ws))) {
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (n4 + r < g.N) ? bias[n4 + r] : 0.f;
  }
  const bool full = (n4 + 3 < g.N);
  const uint64_t dseed = (ACT == FBL_ACT_RELU && g.drop_thresh) ? fbl_seed(g.drop_seed, g.drop_seed_dev) : 0;
  // every lane of the wave stores whole 4-column vectors (wave-uniform): together with "all rows of the slab exist" this
  // selects the branch-free copy of the row loop below
  const bool wave_full = vec_ok && __builtin_amdgcn_ballot_w64(full) == ~0ull;
  // (the two slabs are two calls of one generic lambda, not a loop: whatever the unroller thinks of the body's size, the
  //  accumulator indices stay compile-time constants -- a rolled loop would put the accumulator array into scratch memory)
  auto slab = [&](auto halfc) {
    constexpr int half = decltype(halfc)::value;
    constexpr int cnt0 = (MI - half * 4 < 4) ? MI - half * 4 : 4;
    constexpr int cnt = cnt0 < TPS ? cnt0 : TPS;  // 16-row tiles in this slab
    const int mslab = m_first + half * m_slab_stride;
    // The aux operand of the slab is requested BEFORE the accumulators make their LDS round trip: issued inside the
    // row loop, every iteration exposed a full global-load latency (32 dependent loads per wave on a 256x256 tile).
    constexpr bool AUX_F32 = (AUX == FBL_AUX_ADD_F32);
    const bool aux_fast = (AUX != FBL_AUX_NONE) && !SPLITK && full && ((g.ld_aux & 3) == 0);
    f32x4 xa32[AUX_F32 ? 16 : 1];
    bf16x4 xa16[(AUX != FBL_AUX_NONE && !AUX_F32) ? 16 : 1];
    if (AUX != FBL_AUX_NONE && aux_fast) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int m = min(mslab + it * 4 + er, g.M - 1);
        const long ao = xbase + (long)m * g.ld_aux + n4;
        if (it < cnt * 4) {
          if (AUX_F32) xa32[it] = *(const f32x4*)((const float*)g.aux + ao);
          else xa16[it] = *(const bf16x4*)((const bf16*)g.aux + ao);
        }
      }
    }
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
      if (half * 4 + mi < MI && mi < cnt) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          *(f32x4*)(stage + (mi * 16 + frow) * LDW + ni * 16 + fg * 4) = acc[ni][half * 4 + mi];
      }
    // per-row scale of the slab, loaded up front: a global load inside the row loop would make every iteration wait
    // (vmcnt is one in-order counter for loads AND stores) for all stores of the previous rows to be acknowledged
    float rsv[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) rsv[it] = 1.0f;
    if (g.rowscale) {
#pragma unroll
      for (int it = 0; it < 16; ++it)
        if (it < cnt * 4) rsv[it] = g.rowscale[min(mslab + it * 4 + er, g.M - 1)];
    }
    // Every global load of the slab (aux operand, row scales) is COMPLETE before the first store: the stores below sit in
    // branches, so the compiler cannot count them and would otherwise wait with vmcnt(0) in every row iteration -- i.e.
    // for the acknowledgement of all earlier stores -- to be sure a load issued up here has landed.
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      asm volatile("" : "+v"(rsv[it]));
      if (AUX_F32) asm volatile("" : "+v"(xa32[AUX_F32 ? it : 0]));
      if (AUX != FBL_AUX_NONE && !AUX_F32) asm volatile("" : "+v"(xa16[(AUX != FBL_AUX_NONE && !AUX_F32) ? it : 0]));
    }
    if (n4 < g.N) {
    // element offsets of this lane's first row of the slab; the per-iteration part (4*it rows) is wave-uniform
    const long er_c = cbase + (long)((DBG & 2) ? ((mslab + er) & 255) : (mslab + er)) * g.ldc + n4;
    const long er_x = xbase + (long)(mslab + er) * g.ld_aux + n4;
    // (two copies of the row loop, selected by the wave-uniform aux_fast: the slow path loads aux inside the loop, and a
    //  join of "maybe a load is pending" with the fast path would put a vmcnt(0) into every iteration of both)
    // Three more copies per aux mode, selected by wave-uniform conditions: INNER = 64 / 48: an interior slab -- all its rows
    // exist, every lane stores full vectors -- runs WITHOUT per-row predicates, so the compiler interleaves the LDS reads,
    // transcendental chains and stores of the 16 (12) row iterations instead of branching around each of them (the predicated
    // copy spends most of its time in exposed latencies: 4-13 us per 256x256 tile); INNER = 0: edge slabs, predicated.
    // The interior copies are further specialised by the set of outputs (OUTS: 1 fp32, 2 bf16, 6 bf16 + pre-activation, 0 =
    // any, tested per row): a run-time "is this output wanted" test around a store is a branch per row too.
    auto row_loop = [&](auto fastc, auto innerc, auto outsc) {
    constexpr bool AUXFAST = decltype(fastc)::value;
    constexpr int INNER = decltype(innerc)::value;
    constexpr int OUTS = decltype(outsc)::value;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (it >= cnt * 4) break;
      if (INNER && it * 4 >= INNER) break;
      const int row = it * 4 + er;
      const int m = mslab + row;
      if (!INNER && (m >= g.M || row >= slab_rows)) continue;
      const f32x4 a4 = *(const f32x4*)(stage + row * LDW + ec);
      const float rs = rsv[it];
      float v[4], pre[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (a4[r] * g.alpha + bv[r]) * rs;
      if (SPLITK) {
        if (g.ws) {  // plain 16-byte stores of the partial tile; folded into out_f32 by splitk_reduce_kernel
          *(f32x4*)(g.ws + (((long)(batch * g.splitk + ks) * g.M + m) * g.Nw + n4)) = (f32x4){v[0], v[1], v[2], v[3]};
        } else {
          for (int r = 0; r < 4; ++r)
            if (n4 + r < g.N) unsafeAtomicAdd(g.out_f32 + cbase + (long)m * g.ldc + n4 + r, v[r]);
        }
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pre[r] = v[r];
        if (ACT == FBL_ACT_GELU) v[r] = gelu_erf(v[r]);
        else if (ACT == FBL_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
        else if (ACT == FBL_ACT_GELU_GRAD) gelu_and_grad(pre[r], &v[r], &pre[r]);
      }
      if (ACT == FBL_ACT_RELU && g.drop_thresh) {  // dropout(relu(.)) of the adapter bottleneck, same keys as fbl_dropout_bf16
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] *= fbl_dropout_scale(dseed, (uint64_t)(m + g.drop_row0) * (uint64_t)g.drop_ld + (uint64_t)(n4 + r), g.drop_thresh, g.drop_inv_keep);
      }
      if (AUX != FBL_AUX_NONE) {
        const long ao = er_x + (long)(it * 4) * g.ld_aux;
        float x[4] = {0.f, 0.f, 0.f, 0.f};
        if (AUX == FBL_AUX_ADD_F32) {
          if (AUXFAST) {
            const f32x4 t = xa32[AUX_F32 ? it : 0];
            x[0] = t[0]; x[1] = t[1]; x[2] = t[2]; x[3] = t[3];
          } else {
            for (int r = 0; r < 4 && n4 + r < g.N; ++r) x[r] = ((const float*)g.aux)[ao + r];
          }
        } else {
          if (AUXFAST) {
            const bf16x4 t = xa16[(AUX != FBL_AUX_NONE && !AUX_F32) ? it : 0];
            x[0] = bf2f(t[0]); x[1] = bf2f(t[1]); x[2] = bf2f(t[2]); x[3] = bf2f(t[3]);
          } else {
            for (int r = 0; r < 4 && n4 + r < g.N; ++r) x[r] = bf2f(((const bf16*)g.aux)[ao + r]);
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (AUX == FBL_AUX_ADD_F32 || AUX == FBL_AUX_ADD_BF16) v[r] += x[r];
          else if (AUX == FBL_AUX_MUL_DGELU_BF16) v[r] *= dgelu_erf(x[r]);
          else if (AUX == FBL_AUX_MUL_POS_BF16) v[r] = (x[r] > 0.f) ? v[r] : 0.f;
          else if (AUX == FBL_AUX_MUL_BF16) v[r] *= x[r];
        }
      }
      const long co = er_c + (long)(it * 4) * g.ldc;
      if (DBG & 1) {
        asm volatile("" ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(pre[0]), "v"(pre[1]), "v"(pre[2]), "v"(pre[3]), "v"(co));
      } else if (INNER && OUTS) {
        if (OUTS & 1) *(f32x4*)(g.out_f32 + co) = (f32x4){v[0], v[1], v[2], v[3]};
        if (OUTS & 2) *(bf16x4*)(g.out_bf16 + co) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        if (OUTS & 4) *(bf16x4*)(g.out_pre + co) = (bf16x4){f2bf(pre[0]), f2bf(pre[1]), f2bf(pre[2]), f2bf(pre[3])};
      } else if (INNER || (full && vec_ok)) {
        if (g.out_f32) *(f32x4*)(g.out_f32 + co) = (f32x4){v[0], v[1], v[2], v[3]};
        if (g.out_bf16) *(bf16x4*)(g.out_bf16 + co) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        if (g.out_pre) *(bf16x4*)(g.out_pre + co) = (bf16x4){f2bf(pre[0]), f2bf(pre[1]), f2bf(pre[2]), f2bf(pre[3])};
      } else {
        for (int r = 0; r < 4; ++r) {
          if (n4 + r >= g.N) break;
          if (g.out_f32) g.out_f32[co + r] = v[r];
          if (g.out_bf16) g.out_bf16[co + r] = f2bf(v[r]);
          if (g.out_pre) g.out_pre[co + r] = f2bf(pre[r]);
        }
      }
    }
    };
    const int rows_here = min(slab_rows, cnt * 16);
    const bool inner = !SPLITK && wave_full && mslab + rows_here <= g.M;
    using I0 = std::integral_constant<int, 0>;
    using I48 = std::integral_constant<int, 48>;
    using I64 = std::integral_constant<int, 64>;
    const int outs = (g.out_f32 ? 1 : 0) | (g.out_bf16 ? 2 : 0) | (g.out_pre ? 4 : 0);
    auto by_outs = [&](auto fastc, auto innerc) {
      using O = std::integral_constant<int, 0>;
      if (outs == 2) row_loop(fastc, innerc, std::integral_constant<int, 2>{});
      else if (outs == 1) row_loop(fastc, innerc, std::integral_constant<int, 1>{});
      else if (outs == 6 && ACT != FBL_ACT_NONE) row_loop(fastc, innerc, std::integral_constant<int, 6>{});
      else row_loop(fastc, innerc, O{});
    };
    if (AUX != FBL_AUX_NONE && aux_fast) {
      if (inner && rows_here == cnt * 16) by_outs(std::true_type{}, I64{});
      else if (SHORT48 && inner && rows_here == 48) by_outs(std::true_type{}, I48{});
      else row_loop(std::true_type{}, I0{}, I0{});
    } else if (AUX == FBL_AUX_NONE) {
      if (inner && rows_here == cnt * 16) by_outs(std::false_type{}, I64{});
      else if (SHORT48 && inner && rows_here == 48) by_outs(std::false_type{}, I48{});
      else row_loop(std::false_type{}, I0{}, I0{});
    } else {
      row_loop(std::false_type{}, I0{}, I0{});
    }
    }
  };
  slab(std::integral_constant<int, 0>{});
  if constexpr ((MI + 3) / 4 > 1) slab(std::integral_constant<int, 1>{});
}

// 8-phase 256x256 kernel (gemm8.hip).  Returns 0 when launched, FBL_ERR_ARG for an epilogue combination it does not
// instantiate (the caller then uses the 2-stage kernel).
int launch_gemm8(const GemmArgs& g, int act, int aux_kind, int rows, dim3 grid, hipStream_t stream);  // rows: 256 / 224 / 128
bool gemm8_eligible(const GemmArgs& g);
int launch_gemm8_splitk(const GemmArgs& g, hipStream_t stream);  // plain 256-row tiles, g.splitk slices of g.k8_per K-tiles -> g.ws


}  // namespace fblgemm
