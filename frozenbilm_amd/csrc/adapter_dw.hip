// Weight / bias gradients of a GROUP of bottleneck adapters in one launch (autograd of model/adapter.py:38-42):
//     dWu[H,A] += dy^T . z        dWd[A,H] += dz^T . x        dbd[A] += colsum(dz)
// (dy, x: [N,H] bf16; z = dropout(relu(.)), dz: [N,Ap] bf16, zero beyond A.)  Both products contract over the N token
// ROWS of a wide operand (dy / x, 26 MB at the bench shape) and a narrow one (z / dz): 5 GFLOP against 59 MB per adapter,
// i.e. HBM-bound, but one adapter alone has only 2 x H/64 = 48 output tiles -- too few for 256 CUs without a split over
// the rows, and the split costs a partial-sum round trip as large as the operands.  The adapter gradients of several
// layers are therefore computed TOGETHER once their operands exist (the caller keeps dy / dz alive until then): a workgroup
// owns one [64 x Ap] (dWu) or [Ap x 64] (dWd) output tile of one adapter over ALL rows, accumulates in registers and adds
// into the gradient exactly once -- no workspace, no fold, deterministic.  An adapter that ran more than once in the
// forward pass (the last encoder layer, model/deberta.py:1151-1176: its two enhanced-mask-decoder executions) hands in one
// SEGMENT per execution; the workgroup walks them back to back.
//
// Tile: [64 rows] x [64 wide cols | Ap narrow cols] staged row-major in LDS (16-byte global loads, next tile prefetched
// into registers); the MFMA fragments -- 8 consecutive rows of one column -- come out of the row-major image through
// ds_read_b64_tr_b16 as in gemm_bf16_tn_kernel.  4 waves = 2 (wide halves of 32) x 2 (narrow halves of Ap/2); 167 VGPRs,
// 35 KiB LDS at Ap = 192 -> three workgroups per CU; 16 adapters = 768 workgroups = one full round of the chip.
// Measured (N = 8512, H = 1536, A = 192, 16 adapters): 337 us = 21 us per adapter (the per-adapter route through
// fbl_gemm_bf16_tn_acc: 72 us alone for 6 launches); the wide operand arrives in 128-byte row pieces (2.6 TB/s), which
// is what bounds the pass -- matrix pipe 23 %, LDS 43 % busy.
// which = 0: C[wide col][narrow col] (dWu, a lane owns 4 consecutive a); which = 1: the MFMA operands trade places and a
// lane owns 4 consecutive h of one a: dWd[a][h] is stored row-major without a transposing pass.  The column sums of dz (dbd)
// ride along in the first dWd tile column of each adapter.
#include "fbl_common.h"
#include "../../include/fbl.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short tr16x4;
__device__ __forceinline__ tr16x4 lds_tr16(const bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr16x4*)p);
}

constexpr int ADW_MAX_OUT = FBL_ADW_MAX_ADAPTERS;
constexpr int ADW_MAX_SEG = FBL_ADW_MAX_SEGMENTS;

struct AdwSeg { const bf16* dy; const bf16* z; const bf16* dz; const bf16* x; };
struct AdwArgs {
  AdwSeg seg[ADW_MAX_SEG];
  float* dWu[ADW_MAX_OUT]; float* dWd[ADW_MAX_OUT]; float* dbd[ADW_MAX_OUT];
  int seg_first[ADW_MAX_OUT + 1];  // adapter o owns segments [seg_first[o], seg_first[o+1])
  int ld_dy[ADW_MAX_OUT], ld_z[ADW_MAX_OUT], ld_dz[ADW_MAX_OUT], ld_x[ADW_MAX_OUT];  // row strides, per adapter
  int N, H, A, tiles_h, n_out;
};

constexpr int WI = 2;             // 16-column MFMA tiles per wave along the wide operand (4: +5 % at Ap = 192, spills at 256)
constexpr int WT = WI * 32;       // wide columns of a tile
constexpr int ADW_LDW = WT + 16;  // bf16 row stride of the staged wide tile [64][WT]: 40 dwords = 8 x odd (see the fragment reads)

template <int NT, int WHICH>
__device__ __forceinline__ void adw_tile(const AdwArgs& g, bf16* sW, bf16* sS, const int th, const int o) {
  constexpr int AP = NT * 32;       // narrow columns
  constexpr int LDS_ = AP + 16;     // bf16 row stride of the staged narrow tile [64][AP]: (AP + 16) / 2 dwords = 8 x odd for AP = 64 k
  constexpr int NCH = AP / 8;       // 16-byte chunks per narrow row
  constexpr int SPT = (64 * NCH) / 256;  // narrow chunks per thread (AP = 64 -> 2 ... 256 -> 8)
  constexpr int WCH = WT / 8, WPT = (64 * WCH) / 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int h0 = th * WT;
  const long ldw = WHICH ? g.ld_x[o] : g.ld_dy[o], lds = WHICH ? g.ld_dz[o] : g.ld_z[o];
  const int nk = (g.N + 63) / 64;
  const int s0 = g.seg_first[o], s1 = g.seg_first[o + 1];
  const int total = (s1 - s0) * nk;  // row tiles over all segments
  const int frow = lane & 15, fg = lane >> 4;

  f32x4 acc[NT][WI];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int i = 0; i < WI; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Per-thread staging roles, fixed for the whole pass: element offsets inside a row tile of the operands (added to a
  // wave-uniform tile base -> scalar base + 32-bit lane offset addressing, no per-load 64-bit lane arithmetic) and inside
  // the LDS images.  (The first version recomputed rows / chunks / predicates per load: 130 VALU instructions per row tile
  // against 24 MFMAs -- the pass was instruction-issue bound at 17 % matrix-pipe utilisation.)
  bf16x8 rw[WPT], rs[SPT];
  int gW[WPT], lW[WPT], rW[WPT], gS[SPT], lS[SPT], rS[SPT];
#pragma unroll
  for (int t = 0; t < WPT; ++t) {
    const int id = tid + t * 256;
    const int r = id / WCH, ch = id % WCH;
    // wide columns beyond H (last tile of a ragged H) re-read the last 8 valid columns: an MFMA output row / column
    // depends only on its own operand row, and those outputs are never stored -- no predicate on the loads
    const int c = min(h0 + ch * 8, g.H - 8);
    rW[t] = r;
    gW[t] = (int)(r * ldw + c);
    lW[t] = r * ADW_LDW + ch * 8;
  }
#pragma unroll
  for (int t = 0; t < SPT; ++t) {
    const int id = tid + t * 256;
    const int r = id / NCH, ch = id % NCH;
    rS[t] = r;
    gS[t] = (int)(r * lds + ch * 8);
    lS[t] = r * LDS_ + ch * 8;
  }
  const bf16x8 zero8 = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  const bool ragged = (g.N & 63) != 0;
  int sg_n = s0, kt_n = 0;  // the next row tile to fetch (wave-uniform)
  auto load_tile = [&]() {
    const bf16* W = (WHICH ? g.seg[sg_n].x : g.seg[sg_n].dy) + (long)kt_n * 64 * ldw;
    const bf16* S = (WHICH ? g.seg[sg_n].dz : g.seg[sg_n].z) + (long)kt_n * 64 * lds;
    if (ragged && kt_n == nk - 1) {
      // the last row tile of a segment: the NARROW rows beyond N read as zero; the wide ones re-read row N-1 (finite x 0)
      const int left = g.N - kt_n * 64;
#pragma unroll
      for (int t = 0; t < WPT; ++t) rw[t] = *(const bf16x8*)(W + gW[t] - (long)max(rW[t] - (left - 1), 0) * ldw);
#pragma unroll
      for (int t = 0; t < SPT; ++t) rs[t] = rS[t] < left ? *(const bf16x8*)(S + gS[t]) : zero8;
    } else {
#pragma unroll
      for (int t = 0; t < WPT; ++t) rw[t] = *(const bf16x8*)(W + gW[t]);
#pragma unroll
      for (int t = 0; t < SPT; ++t) rs[t] = *(const bf16x8*)(S + gS[t]);
    }
    if (++kt_n == nk) { kt_n = 0; ++sg_n; }
  };
  const bool do_cs = WHICH == 1 && th == 0 && g.dbd[o] != nullptr;
  const bf16x2 ones2 = {(bf16)1.0f, (bf16)1.0f};
  float cs[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) cs[j] = 0.f;
  // fragment read bases of this lane: row (fg*4 + frow/4) of a 32-row half, 4-column sub-block (frow%4) -- everything else
  // is a compile-time offset.  The k slots of a lane group (8 per MFMA) are rows fg*4 .. fg*4+3 (first read) and 16 + fg*4 ..
  // (second read) of the half -- the same permutation of the 32 rows for both operands, so the contraction is unchanged.
  // Why not rows fg*8 .. fg*8+7: ds_read_b64_tr_b16 serves lanes 0-31 (groups 0, 1) in one pass, i.e. eight 32-byte row
  // pieces; with rows {0-3, 8-11} no row pitch puts them on disjoint banks (r05_pmc.md: 33 % of the LDS cycles of this kernel
  // were bank conflicts), with the eight CONSECUTIVE rows {0-7} a pitch of 8 x odd dwords tiles the 64 banks exactly.
  const bf16* fW = sW + (fg * 4 + (frow >> 2)) * ADW_LDW + (frow & 3) * 4 + wm * (WI * 16);
  const bf16* fS = sS + (fg * 4 + (frow >> 2)) * LDS_ + (frow & 3) * 4 + wn * (NT * 16);
  if (total > 0) load_tile();
  for (int it = 0; it < total; ++it) {
#pragma unroll
    for (int t = 0; t < WPT; ++t) *(bf16x8*)(sW + lW[t]) = rw[t];
#pragma unroll
    for (int t = 0; t < SPT; ++t) *(bf16x8*)(sS + lS[t]) = rs[t];
    __syncthreads();
    if (it + 1 < total) load_tile();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 wf[WI], sf[NT];
#pragma unroll
      for (int i = 0; i < WI; ++i) {
        // ds_read_b64_tr_b16: the 16 lanes of a group hand in a 4x16 block (rows k..k+3, 16 columns) and each lane gets
        // its own column back -- 4 consecutive k of one column; two reads per MFMA fragment
        const bf16* p = fW + s * 32 * ADW_LDW + i * 16;
        union { tr16x4 h[2]; bf16x8 v; } u;
        u.h[0] = lds_tr16(p);
        u.h[1] = lds_tr16(p + 16 * ADW_LDW);
        wf[i] = u.v;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16* p = fS + s * 32 * LDS_ + j * 16;
        union { tr16x4 h[2]; bf16x8 v; } u;
        u.h[0] = lds_tr16(p);
        u.h[1] = lds_tr16(p + 16 * LDS_);
        sf[j] = u.v;
      }
      if (WHICH == 1 && do_cs && s == wm) {
        // column sums of dz (dbd) ride along in the first tile column of an adapter: the narrow fragments are dz itself --
        // a lane adds up its 8 rows of column (j*16 + frow) with v_dot2c_f32_bf16 against (1, 1); the wave row wm takes the
        // 32-row half s = wm, so each of the four waves pays 4 x NT extra VALU instructions per row tile
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          union { bf16x8 v; bf16x2 p[4]; } u;
          u.v = sf[j];
#pragma unroll
          for (int q = 0; q < 4; ++q) cs[j] = __builtin_amdgcn_fdot2_f32_bf16(u.p[q], ones2, cs[j], false);
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < WI; ++i) {
          if (WHICH == 0) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf[j], wf[i], acc[j][i], 0, 0, 0);
          else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], sf[j], acc[j][i], 0, 0, 0);
        }
    }
    __syncthreads();
  }
  // the tile is added into the gradient once (this workgroup is its only writer in the launch)
  const int A = g.A;
  if (WHICH == 0) {
    float* out = g.dWu[o];
    if (!out) return;
#pragma unroll
    for (int i = 0; i < WI; ++i) {
      const int h = h0 + wm * (WI * 16) + i * 16 + frow;
      if (h >= g.H) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int a4 = wn * (NT * 16) + j * 16 + fg * 4;
        float* p = out + (long)h * A + a4;
        if (a4 + 4 <= A && ((uintptr_t)p & 15) == 0) *(f32x4*)p += acc[j][i];
        else
          for (int r = 0; r < 4; ++r)
            if (a4 + r < A) p[r] += acc[j][i][r];
      }
    }
  } else {
    if (do_cs) {  // partial sums of the 8 (wave row, row group) pairs are folded in index order
      float* red = (float*)sS;  // [8][AP] (the row-tile loop ended with a barrier)
#pragma unroll
      for (int j = 0; j < NT; ++j) red[(wm * 4 + fg) * AP + wn * (NT * 16) + j * 16 + frow] = cs[j];
      __syncthreads();
      if (tid < A) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += red[k * AP + tid];
        g.dbd[o][tid] += t;
      }
    }
    float* out = g.dWd[o];
    if (!out) return;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int a = wn * (NT * 16) + j * 16 + frow;
      if (a >= A) continue;
#pragma unroll
      for (int i = 0; i < WI; ++i) {
        const int h4 = h0 + wm * (WI * 16) + i * 16 + fg * 4;
        float* p = out + (long)a * g.H + h4;
        if (h4 + 4 <= g.H && ((uintptr_t)p & 15) == 0) *(f32x4*)p += acc[j][i];
        else
          for (int r = 0; r < 4; ++r)
            if (h4 + r < g.H) p[r] += acc[j][i][r];
      }
    }
  }
}

template <int NT>
__global__ __launch_bounds__(256, 2) void adapter_dw_kernel(AdwArgs g) {
  __shared__ __attribute__((aligned(16))) bf16 sW[64 * ADW_LDW];
  __shared__ __attribute__((aligned(16))) bf16 sS[64 * (NT * 32 + 16)];
  // Workgroup -> (problem, tile column).  The hardware deals consecutive workgroups round-robin to the 8 XCDs, each with
  // its own L2; all tile columns of one problem (adapter, which) read the SAME narrow operand, so they are given to one XCD
  // (problem p lives on XCD p % 8): the narrow rows are fetched from memory once instead of once per XCD (measured before:
  // the pass pulled 1.8x its operand bytes through the fabric and no two workgroups of a CU overlapped).
  const int lin = blockIdx.x, xcd = lin & 7, slot = lin >> 3;
  const int p = (slot / g.tiles_h) * 8 + xcd, th = slot % g.tiles_h;
  if (p >= 2 * g.n_out) return;
  const int o = p >> 1;
  if ((p & 1) == 0) adw_tile<NT, 0>(g, sW, sS, th, o);
  else adw_tile<NT, 1>(g, sW, sS, th, o);
}

}  // namespace

extern "C" int fbl_adapter_bwd_dw(int n_adapters, const int32_t* seg_first, const void* const* dy_bf16, const void* const* z_bf16,
                                  const void* const* dz_bf16, const void* const* x_bf16, const int64_t* ld_dy,
                                  const int64_t* ld_z, const int64_t* ld_dz, const int64_t* ld_x, int N, int H, int A, int Ap,
                                  float* const* dWu, float* const* dWd, float* const* dbd, void* stream) {
  if (n_adapters <= 0 || N <= 0 || H <= 0 || A <= 0) return 0;
  if (n_adapters > ADW_MAX_OUT || !seg_first || seg_first[0] != 0) return FBL_ERR_ARG;
  const int nseg = seg_first[n_adapters];
  if (nseg <= 0 || nseg > ADW_MAX_SEG) return FBL_ERR_ARG;
  if (Ap < A || (Ap % 64) || Ap > 256 || (H % 8)) return FBL_ERR_SHAPE;
  if (!dy_bf16 || !z_bf16 || !dz_bf16 || !x_bf16 || !ld_dy || !ld_z || !ld_dz || !ld_x) return FBL_ERR_ARG;
  AdwArgs g{};
  for (int o = 0; o < n_adapters; ++o) {
    if (seg_first[o + 1] < seg_first[o]) return FBL_ERR_ARG;
    g.seg_first[o] = seg_first[o];
    g.dWu[o] = dWu ? dWu[o] : nullptr;
    g.dWd[o] = dWd ? dWd[o] : nullptr;
    g.dbd[o] = dbd ? dbd[o] : nullptr;
    if ((ld_dy[o] % 8) || (ld_z[o] % 8) || (ld_dz[o] % 8) || (ld_x[o] % 8)) return FBL_ERR_ALIGN;
    if (ld_dy[o] < H || ld_x[o] < H || ld_z[o] < Ap || ld_dz[o] < Ap) return FBL_ERR_ARG;
    if ((ld_dy[o] | ld_x[o] | ld_z[o] | ld_dz[o]) >> 24) return FBL_ERR_ARG;  // 64-row tile offsets stay inside 32 bits
    g.ld_dy[o] = (int)ld_dy[o]; g.ld_z[o] = (int)ld_z[o]; g.ld_dz[o] = (int)ld_dz[o]; g.ld_x[o] = (int)ld_x[o];
  }
  g.seg_first[n_adapters] = nseg;
  for (int s = 0; s < nseg; ++s) {
    if (!dy_bf16[s] || !z_bf16[s] || !dz_bf16[s] || !x_bf16[s]) return FBL_ERR_ARG;
    g.seg[s] = AdwSeg{(const bf16*)dy_bf16[s], (const bf16*)z_bf16[s], (const bf16*)dz_bf16[s], (const bf16*)x_bf16[s]};
  }
  g.N = N; g.H = H; g.A = A;
  g.tiles_h = (H + WT - 1) / WT;
  g.n_out = n_adapters;
  dim3 grid((unsigned)(((2 * n_adapters + 7) / 8) * 8 * g.tiles_h));
  switch (Ap / 64) {
    case 1: hipLaunchKernelGGL(adapter_dw_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, g); break;
    case 2: hipLaunchKernelGGL(adapter_dw_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, g); break;
    case 3: hipLaunchKernelGGL(adapter_dw_kernel<6>, grid, dim3(256), 0, (hipStream_t)stream, g); break;
    default: hipLaunchKernelGGL(adapter_dw_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, g); break;
  }
  FBL_CHECK_LAUNCH();
  return 0;
}
