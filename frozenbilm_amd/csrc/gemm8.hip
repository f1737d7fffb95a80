// 8-phase bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( alpha * A[M,K] . B[N,K]^T ), 256x256x64 tile, 512 threads.
//
// Same contract and epilogue as gemm.hip's kernel (reference: the nn.Linear calls of model/deberta.py:255,311,329,
// 757-765,1545,1550 and their dX backward); this is the main loop for the large GEMMs of the step, built for the
// CDNA4 execution model instead of the "load stage / barrier / compute stage" loop:
//
//  * The K-tile (A 256x64 | B 256x64, 64 KiB) is cut into four HALF-TILES of 16 KiB -- A rows 0..127 (A0), B rows
//    0..127 (B0), B rows 128..255 (B1), A rows 128..255 (A1) -- and the LDS holds a ring of 8 of them (2 K-tiles,
//    128 KiB).  Waves are a 2 (M) x 4 (N) grid, but a wave's 128 x 64 outputs are INTERLEAVED over the halves: 64 rows
//    of A0 + 64 rows of A1, 32 columns of B0 + 32 of B1.  One K-tile is then four phases, each one 64x32 quadrant of the
//    wave's tile (16 v_mfma_f32_16x16x32_bf16) that needs at most one new half-tile:
//        q0: read A0, B0 -> A0xB0     q1: read B1 -> A0xB1     q2: read A1 -> A1xB1     q3: -> A1xB0 (B0 still in VGPRs)
//  * Every phase issues the LDS-DMA (global_load_lds_dwordx4, 2 per thread) of exactly one half-tile, six half-tiles
//    ahead of its consumption, and waits with a COUNTED s_waitcnt vmcnt(8): four half-tiles stay in flight across the
//    barriers, the queue is never drained inside the loop (raw s_barrier -- __syncthreads() would emit vmcnt(0)).
//    RAW: the half-tile read in phase p+1 is waited for before the first barrier of phase p.  WAR: a ring slot is
//    re-filled two or three phases after the phase that read it.
//  * The two wave rows (waves 0-3 / 4-7: one wave of each per SIMD) run half a phase apart (the second row takes one
//    extra barrier at the start): while one wave of a SIMD issues its 16 MFMAs under s_setprio 1, the other one does its
//    ds_read_b128 fragment reads and DMA issue -- the matrix pipe always has a wave that is not waiting on LDS.
//  * LDS image of a half-tile: [128 rows][8 x 16 B], chunk ^= row & 7 applied on the DMA SOURCE address (LDS-DMA
//    destinations are lane-linear) and again on the ds_read_b128 side (conflict-free fragment reads); MFMA operands
//    swapped (weights as "A", activations as "B") so a lane owns 4 consecutive output columns.
//
// K must be a multiple of 128 with K >= 256 (an even number >= 4 of K-tiles; every large GEMM of the step is), operands
// below 4 GiB (32-bit byte offsets against a scalar base): gemm8_eligible(); everything else stays on gemm.hip's kernels.
#include <stdlib.h>
#include <type_traits>

#include "gemm_common.h"

namespace fblgemm {
namespace {

template <int V> using IC = std::integral_constant<int, V>;

// LDS-DMA with a scalar base and a 32-bit per-lane byte offset.  Inline asm: the request is invisible to the compiler's
// s_waitcnt bookkeeping (which is the point -- the pipeline below counts vmcnt by hand) and M0, the LDS destination, is
// written in the same statement that uses it and restored afterwards.
__device__ __forceinline__ void glds16_saddr(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

constexpr int HT_BYTES = 128 * BK * 2;       // one half-tile: 16 KiB
constexpr int RING_BYTES = 8 * HT_BYTES;     // 128 KiB
constexpr int SMEM8_BYTES = (8 * 64 * 68 * 4 > RING_BYTES) ? 8 * 64 * 68 * 4 : RING_BYTES;  // epilogue staging is larger

// VAR bit 0: stagger the two wave rows by half a phase; bit 1: s_setprio around the MFMA cluster (experiment switches)
// RT = 1 (R224): 224-row tile (halves of 112 rows: the first wave row owns 64 rows of each half, the second one 48 = three
// 16-row MFMA tiles) for the M = 8512, N = 1536 GEMMs of the step: 38 x 6 = 228 tiles instead of 204 that are 14 % bigger.
// RT = 2 (R128): 128-row tile (halves of 64 rows, BOTH wave rows own 32 of each = two MFMA tiles) for row counts at which
// neither of the tall tiles covers the chip (packed ragged batches, small batches: M = 5322, N = 1536 -> 252 tiles).
// SK: split-K -- blockIdx.y = slice ks of g.splitk: K-tiles [ks * k8_per, + k8_per) (the last slice: what is left), partial tile
// stored to the workspace g.ws[ks] (folded by splitk_reduce_kernel): few rows against a very long K (the prediction head's
// backward [~700 x 1536 x 128128]: 18 tiles x 14 slices instead of 72 tiles x 15 slices of the 128 x 128 two-stage kernel).
template <int ACT, int AUX, int VAR, int RT = 0, bool SK = false>
__global__ __launch_bounds__(512, 2) void gemm8_kernel(GemmArgs g) {
  constexpr bool R224 = RT == 1, R128 = RT == 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  int tm, tn;
  tile_of_block(blockIdx.x, g.tiles_m, g.tiles_n, &tm, &tn);
  constexpr int HROWS = R224 ? 112 : R128 ? 64 : 128;  // rows of an A half-tile
  constexpr int WROWS = R128 ? 32 : 64;                // rows of a half-tile that one wave row owns (the first one)
  constexpr int NRT = R128 ? 2 : 4;                    // its 16-row MFMA tiles
  const int m0 = tm * (2 * HROWS), n0 = tn * 256;
  const int ks = SK ? (int)blockIdx.y : 0;
  const int kt0 = SK ? ks * g.k8_per : 0;
  const int nk = SK ? min(g.K / BK - kt0, g.k8_per) : g.K / BK;  // even, >= 4
  const bool short_rows = R224 && wm == 1;  // this wave owns 3 (not 4) row tiles per half

  // ---- LDS-DMA source offsets (bytes from A / B, K-tile 0).  One instruction of the workgroup covers 64 rows of a
  // half-tile (wave w: rows 8w..8w+7 of them, lane -> row lane>>3, physical chunk lane&7), two instructions a half-tile.
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;  // logical 16-byte chunk fetched by this lane (row & 7 == lrow)
  uint32_t a_off[2][2], b_off[2][2];     // [half][instruction]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = (j * 8 + wave) * 8 + lrow;  // row inside the half-tile slot
      // (224-row tiles: slot rows 112..127 are never read -- 128-row tiles: 64..127 --; their lanes re-request the last
      //  row so that every wave issues the same number of DMA instructions and one vmcnt count serves all)
      const int am = min(m0 + h * HROWS + min(rr, HROWS - 1), g.M - 1);
      const int bn = min(n0 + h * 128 + rr, g.N - 1);
      a_off[h][j] = (uint32_t)(((long)am * g.lda + lchunk * 8) * 2);
      b_off[h][j] = (uint32_t)(((long)bn * g.ldb + lchunk * 8) * 2);
    }
  const char* Ab = (const char*)g.A + (long)kt0 * (BK * 2);
  const char* Bb = (const char*)g.B + (long)kt0 * (BK * 2);
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem + wave * 1024;

  // half-tile S (0 = A0, 1 = B0, 2 = B1, 3 = A1) of K-tile kt -> ring slot (BUF, S).  Scalar base + 32-bit lane offset
  // (the saddr form: no 64-bit VALU address arithmetic per request).
  auto issue = [&](auto Sc, auto Bc, int kt) {
    constexpr int S = decltype(Sc)::value, BUF = decltype(Bc)::value;
    const char* base = ((S == 0 || S == 3) ? Ab : Bb) + (long)kt * (BK * 2);
    const uint32_t o0 = (S == 0) ? a_off[0][0] : (S == 3) ? a_off[1][0] : (S == 1) ? b_off[0][0] : b_off[1][0];
    const uint32_t o1 = (S == 0) ? a_off[0][1] : (S == 3) ? a_off[1][1] : (S == 1) ? b_off[0][1] : b_off[1][1];
    const uint32_t dst = lds0 + (BUF * 4 + S) * HT_BYTES;
    glds16_saddr(base, o0, dst);
    glds16_saddr(base, o1, dst + 8192);
  };

  f32x4 acc[4][8];  // [ch*2 + j][rh*4 + i]: column half ch, 16-column tile j; row half rh, 16-row tile i
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside a half-tile: row*128 + ((s*4 + lane>>4) ^ (row&7))*16; k-sub-step s=1 flips bit 6
  const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
  const int a_rd0 = (wm * WROWS + frow) * 128 + ((fg ^ fsw) * 16);
  const int a_rd1 = a_rd0 ^ 64;
  const int b_rd0 = (wn * 32 + frow) * 128 + ((fg ^ fsw) * 16);
  const int b_rd1 = b_rd0 ^ 64;

  bf16x8 af[4][2], bf0[2][2], bf1[2][2];  // [tile][k-sub-step]
  auto read_a = [&](auto Hc, auto Bc) {
    constexpr int SLOT = (decltype(Bc)::value * 4 + (decltype(Hc)::value ? 3 : 0)) * HT_BYTES;
#pragma unroll
    for (int i = 0; i < NRT; ++i) {  // (the fourth row tile of a short wave reads slot rows it never uses: harmless)
      af[i][0] = *(const bf16x8*)(smem + SLOT + i * 2048 + a_rd0);
      af[i][1] = *(const bf16x8*)(smem + SLOT + i * 2048 + a_rd1);
    }
  };
  auto read_b = [&](auto Hc, auto Bc) {
    constexpr int H = decltype(Hc)::value;
    constexpr int SLOT = (decltype(Bc)::value * 4 + 1 + H) * HT_BYTES;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (H == 0) {
        bf0[j][0] = *(const bf16x8*)(smem + SLOT + j * 2048 + b_rd0);
        bf0[j][1] = *(const bf16x8*)(smem + SLOT + j * 2048 + b_rd1);
      } else {
        bf1[j][0] = *(const bf16x8*)(smem + SLOT + j * 2048 + b_rd0);
        bf1[j][1] = *(const bf16x8*)(smem + SLOT + j * 2048 + b_rd1);
      }
    }
  };
  auto mma_n = [&](auto RHc, auto CHc, auto NRc) {
    constexpr int RH = decltype(RHc)::value, CH = decltype(CHc)::value, NR = decltype(NRc)::value;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (CH == 0)
            acc[j][RH * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf0[j][s], af[i][s], acc[j][RH * 4 + i], 0, 0, 0);
          else
            acc[2 + j][RH * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf1[j][s], af[i][s], acc[2 + j][RH * 4 + i], 0, 0, 0);
        }
  };
  // One phase of K-tile kt (ring buffer BUF).  TAIL < 0: steady state (issue the half-tile six ahead, keep four in
  // flight); TAIL = k >= 0: nothing left to issue, k half-tiles may stay in flight.
  auto phase = [&](auto Qc, auto Bc, auto Tc, int kt, auto NRc) {
    constexpr int Q = decltype(Qc)::value, BUF = decltype(Bc)::value, TAIL = decltype(Tc)::value;
    if constexpr (Q == 0) {
      read_b(IC<0>{}, Bc);
      __builtin_amdgcn_sched_barrier(0);
      read_a(IC<0>{}, Bc);
    } else if constexpr (Q == 1) {
      read_b(IC<1>{}, Bc);
    } else if constexpr (Q == 2) {
      read_a(IC<1>{}, Bc);
    }
    if constexpr (TAIL < 0) {
      // phase P = 4*kt + Q issues half-tile sequence number P + 6: slot (Q+2)&3 of K-tile kt+1 (Q < 2) / kt+2
      issue(IC<(Q + 2) & 3>{}, IC<(Q < 2) ? (BUF ^ 1) : BUF>{}, kt + (Q < 2 ? 1 : 2));
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if constexpr (TAIL == 3) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if constexpr (TAIL == 2) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if constexpr (TAIL == 1) {
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (VAR & 2) __builtin_amdgcn_s_setprio(1);
    if constexpr (Q == 0) mma_n(IC<0>{}, IC<0>{}, NRc);
    else if constexpr (Q == 1) mma_n(IC<0>{}, IC<1>{}, NRc);
    else if constexpr (Q == 2) mma_n(IC<1>{}, IC<1>{}, NRc);
    else mma_n(IC<1>{}, IC<0>{}, NRc);
    if constexpr (VAR & 2) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };

  // ---- prologue: half-tile sequence numbers 0..5 (all of K-tile 0, A0/B0 of K-tile 1); phase 0 reads 0 and 1
  issue(IC<0>{}, IC<0>{}, 0);
  issue(IC<1>{}, IC<0>{}, 0);
  issue(IC<2>{}, IC<0>{}, 0);
  issue(IC<3>{}, IC<0>{}, 0);
  issue(IC<0>{}, IC<1>{}, 1);
  issue(IC<1>{}, IC<1>{}, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (VAR & 1) {
    if (wm == 1) __builtin_amdgcn_s_barrier();
  }

  // The whole K loop as one function of the number of row tiles this wave multiplies (4; 3 for the second wave row of a
  // 224-row tile): two complete copies of the loop instead of a branch around every MFMA group -- the barriers pair up
  // across the copies, and each copy gets its own register allocation.
  auto k_loop = [&](auto NRc) {
    for (int kt = 0; kt + 2 < nk; kt += 2) {  // K-tile pairs with a full pipeline behind them
      phase(IC<0>{}, IC<0>{}, IC<-1>{}, kt, NRc);
      phase(IC<1>{}, IC<0>{}, IC<-1>{}, kt, NRc);
      phase(IC<2>{}, IC<0>{}, IC<-1>{}, kt, NRc);
      phase(IC<3>{}, IC<0>{}, IC<-1>{}, kt, NRc);
      phase(IC<0>{}, IC<1>{}, IC<-1>{}, kt + 1, NRc);
      phase(IC<1>{}, IC<1>{}, IC<-1>{}, kt + 1, NRc);
      phase(IC<2>{}, IC<1>{}, IC<-1>{}, kt + 1, NRc);
      phase(IC<3>{}, IC<1>{}, IC<-1>{}, kt + 1, NRc);
    }
    {  // last pair: the pipeline drains (sequence numbers stop at 4*nk - 1)
      const int kt = nk - 2;
      phase(IC<0>{}, IC<0>{}, IC<-1>{}, kt, NRc);
      phase(IC<1>{}, IC<0>{}, IC<-1>{}, kt, NRc);
      phase(IC<2>{}, IC<0>{}, IC<3>{}, kt, NRc);
      phase(IC<3>{}, IC<0>{}, IC<2>{}, kt, NRc);
      phase(IC<0>{}, IC<1>{}, IC<1>{}, kt + 1, NRc);
      phase(IC<1>{}, IC<1>{}, IC<0>{}, kt + 1, NRc);
      phase(IC<2>{}, IC<1>{}, IC<0>{}, kt + 1, NRc);
      phase(IC<3>{}, IC<1>{}, IC<0>{}, kt + 1, NRc);
    }
  };
  if constexpr (R224) {
    if (short_rows) k_loop(IC<3>{});
    else k_loop(IC<4>{});
  } else if constexpr (R128) {
    k_loop(IC<2>{});
  } else {
    k_loop(IC<4>{});
  }
  if constexpr (VAR & 1) {
    if (wm == 0) __builtin_amdgcn_s_barrier();  // balances the extra barrier of the second wave row
  }
  // every fragment read of the ring retired (each wave waited lgkmcnt(0) before its last barriers): the epilogue may
  // reuse the LDS
  if constexpr (VAR & 4) {  // experiment: main loop only (accumulators kept alive, nothing stored)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
  }
  const int ec = (lane & 15) * 4;
  gemm_epilogue<ACT, AUX, SK, 8, (VAR >> 3) & 3, R224, NRT>(g, smem, wave, lane, acc, m0 + wm * WROWS, HROWS,
                                                            n0 + (ec >> 5) * 128 + wn * 32 + (ec & 31), 0, ks,
                                                            R128 ? 32 : (short_rows ? 48 : 64));
}

template <int ACT, int AUX, int RT>
int launch_variant(const GemmArgs& g, dim3 grid, hipStream_t stream) {
  static const int var = FBL_ENV_INT("FBL_GEMM8_VAR", 3);
#define FBL_G8_LAUNCH(VAR_)                                                                                     \
  do {                                                                                                          \
    static bool attr_set = false;                                                                               \
    auto kfn = gemm8_kernel<ACT, AUX, VAR_, RT>;                                                                \
    if (!attr_set) {                                                                                            \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM8_BYTES); \
      if (e != hipSuccess) return (int)e;                                                                       \
      attr_set = true;                                                                                          \
    }                                                                                                           \
    hipLaunchKernelGGL(kfn, grid, dim3(512), SMEM8_BYTES, stream, g);                                           \
  } while (0)
#ifdef FBL_DEBUG_SWITCHES
  if constexpr ((ACT == FBL_ACT_NONE || ACT == FBL_ACT_GELU_GRAD) && AUX == FBL_AUX_NONE && RT == 0) {  // experiment variants
    if (var == 0) FBL_G8_LAUNCH(0);
    else if (var == 1) FBL_G8_LAUNCH(1);
    else if (var == 2) FBL_G8_LAUNCH(2);
    else if (var == 7) FBL_G8_LAUNCH(7);
    else if (var == 11) FBL_G8_LAUNCH(11);  // no global stores
    else if (var == 19) FBL_G8_LAUNCH(19);  // stores alias 256 rows
    else FBL_G8_LAUNCH(3);
  } else {
    FBL_G8_LAUNCH(3);
  }
#else
  FBL_G8_LAUNCH(3);
#endif
#undef FBL_G8_LAUNCH
  FBL_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// split-K launch of the plain 256-row configuration: grid (tiles, slices)
int launch_gemm8_splitk(const GemmArgs& g, hipStream_t stream) {
  static bool attr_set = false;
  auto kfn = gemm8_kernel<FBL_ACT_NONE, FBL_AUX_NONE, 3, 0, true>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM8_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3(g.tiles_m * g.tiles_n, g.splitk), dim3(512), SMEM8_BYTES, stream, g);
  FBL_CHECK_LAUNCH();
  return 0;
}

bool gemm8_eligible(const GemmArgs& g) {
  const int nk = g.K / BK;
  if (g.K % BK || nk < 4 || (nk & 1)) return false;
  if (g.a_kblk || g.kskip_len || g.splitk != 1 || (g.drop_thresh && g.seg_n <= 0)) return false;
  if ((long)g.M * g.lda * 2 >= (1l << 32) || (long)g.N * g.ldb * 2 >= (1l << 32)) return false;
  return true;
}

// rows: 256, 224 or 128 (tile height)
int launch_gemm8(const GemmArgs& g, int act, int aux_kind, int rows, dim3 grid, hipStream_t stream) {
  if (rows == 224) {  // the N = 1536 GEMMs of the step: plain (one or two outputs) and residual-add epilogues
    if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_NONE) return launch_variant<FBL_ACT_NONE, FBL_AUX_NONE, 1>(g, grid, stream);
    if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_ADD_F32) return launch_variant<FBL_ACT_NONE, FBL_AUX_ADD_F32, 1>(g, grid, stream);
    return FBL_ERR_ARG;
  }
  if (rows == 128) {  // the same GEMMs at row counts that leave the tall tiles' grids half empty
    if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_NONE) return launch_variant<FBL_ACT_NONE, FBL_AUX_NONE, 2>(g, grid, stream);
    if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_ADD_F32) return launch_variant<FBL_ACT_NONE, FBL_AUX_ADD_F32, 2>(g, grid, stream);
    return FBL_ERR_ARG;
  }
  if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_NONE) return launch_variant<FBL_ACT_NONE, FBL_AUX_NONE, 0>(g, grid, stream);
  if (act == FBL_ACT_GELU && aux_kind == FBL_AUX_NONE) return launch_variant<FBL_ACT_GELU, FBL_AUX_NONE, 0>(g, grid, stream);
  if (act == FBL_ACT_GELU_GRAD && aux_kind == FBL_AUX_NONE) return launch_variant<FBL_ACT_GELU_GRAD, FBL_AUX_NONE, 0>(g, grid, stream);
  if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_ADD_F32) return launch_variant<FBL_ACT_NONE, FBL_AUX_ADD_F32, 0>(g, grid, stream);
  if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_ADD_BF16) return launch_variant<FBL_ACT_NONE, FBL_AUX_ADD_BF16, 0>(g, grid, stream);
  if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_MUL_BF16) return launch_variant<FBL_ACT_NONE, FBL_AUX_MUL_BF16, 0>(g, grid, stream);
  if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_MUL_DGELU_BF16) return launch_variant<FBL_ACT_NONE, FBL_AUX_MUL_DGELU_BF16, 0>(g, grid, stream);
  return FBL_ERR_ARG;
}

}  // namespace fblgemm
