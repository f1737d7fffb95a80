// bf16 MFMA GEMM for gfx950:  C[M,N] = epilogue( alpha * A[M,K] . B[N,K]^T )   ("NT": both operands K-contiguous)
//
// This one kernel serves every dense contraction of the FrozenBiLM hot path (reference: the nn.Linear calls of
// model/deberta.py:255,311,329,757-765,847-853,1545 and model/adapter.py:38,42; their backward dX = dY.W is the
// same kernel on the pre-transposed frozen weight).
//
// Structure (CDNA4): 128x128x64 tile, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16
// accumulators.  Operand tiles go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double-buffered,
// one barrier per K-step.  LDS image is [128 rows][8 x 16B chunks] with chunk ^= row&7 applied on the SOURCE
// address (LDS-DMA destinations are lane-linear) and again on the ds_read_b128 side -> conflict-free fragment reads.
// MFMA operands are swapped (weights as "A", activations as "B") so each lane owns 4 consecutive output columns
// -> 16-byte fp32 / 8-byte bf16 epilogue stores.
#include <stdlib.h>

#include "gemm_common.h"

using namespace fblgemm;

namespace {

// Tile configurations (NW waves, wave grid WR x WC, each wave (MI*16) x 64 outputs; BM = BN = 32*NW):
//   small: 128x128, 4 waves (2x2), MI=4  -> 69.6 KiB LDS, 2 workgroups/CU   (narrow / short GEMMs)
//   big:   256x256, 8 waves (2x4), MI=8  -> 136 KiB LDS, 1 workgroup/CU     (1/3 fewer LDS bytes per MFMA)
// (the large square-ish problems run the 8-phase kernel of gemm8.hip instead of the NW = 8 configuration)
constexpr int KSKIP_MAX_STEPS = 2048;  // K-steps per split of a launch that skips zero blocks (LDS list of the valid ones)
constexpr int KSKIP_SMEM_BYTES = 3 * (128 * BK * 2 + 64 * BK * 2) + KSKIP_MAX_STEPS * 2 + 16;

template <int NW> struct TileCfg {
  static constexpr int BM = 32 * NW, BN = 32 * NW;
  static constexpr int TILE_BYTES = BM * BK * 2;
  static constexpr int STAGE_BYTES = 2 * TILE_BYTES;
  static constexpr int SMEM_BYTES = (NW * 64 * 68 * 4 > 2 * STAGE_BYTES) ? NW * 64 * 68 * 4 : 2 * STAGE_BYTES;
};

// MI_: 16-row MFMA tiles per wave along M.  The tile is (32*MI_) x (32*NW); MI_ = NW gives the square 128/256 tiles,
// MI_ = 7 with NW = 8 a 224x256 tile for shapes whose 256x256 grid leaves CUs idle (8512 rows = 38 x 224 exactly:
// 228 tiles on 256 CUs for N = 1536 instead of 204 bigger ones).  LDS keeps the 32*NW-row A image; the unused rows are
// simply not fetched.
template <int NW, int ACT, int AUX, bool SPLITK, int SCHED = 0, int MI_ = NW, bool KSKIP = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 2) void gemm_bf16_nt_kernel(GemmArgs g) {
  using Cfg = TileCfg<NW>;
  constexpr int MI = MI_;           // 16-row MFMA tiles per wave along M; 4 tiles (64 cols) along N
  constexpr int BM = 32 * MI;
  // LDS stage = A image | B image.  The 2-stage loops reserve the square tile's A image even for shorter tiles; the
  // ring of SCHED 5 packs the BM rows it really has (three stages of 24 KiB: two workgroups per CU still fit)
  constexpr int BN = Cfg::BN;
  constexpr int TILE_BYTES = (SCHED == 5) ? BM * BK * 2 : Cfg::TILE_BYTES;
  constexpr int STAGE_BYTES = (SCHED == 5) ? TILE_BYTES + Cfg::TILE_BYTES : Cfg::STAGE_BYTES;
  constexpr int WC = NW / 2;        // waves along N (2 rows of waves along M)
  constexpr int WROWS = MI * 16;    // rows of C per wave
  static_assert(SCHED == 0 || SCHED == 5 || MI_ == NW, "the counted-vmcnt schedules assume the square tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x (A tile | B tile)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WC, wn = wave % WC;

  int tm, tn;
  int by = blockIdx.y;
  if constexpr (KSKIP) {
    // The batched position-table products are a 2-D grid (few tiles x many (batch, K-slice) pairs): consecutive workgroups
    // -- x fastest -- go round-robin to the 8 XCDs, which would put the row tiles of ONE pair on different XCDs, each
    // fetching that pair's B operand (Q^T / K^T, 1.3 MB) into its own L2.  Re-deal: 8 pairs at a time, workgroup L of the
    // group of 8 * tiles on XCD L % 8 takes pair L % 8, tile L / 8 -- all tiles of a pair share one L2.
    const int gx = gridDim.x;
    if ((gridDim.y & 7) == 0) {
      const int lin = blockIdx.y * gx + blockIdx.x;
      const int grp = lin / (8 * gx), r = lin - grp * 8 * gx;
      by = grp * 8 + (r & 7);
      tile_of_block(r >> 3, g.tiles_m, g.tiles_n, &tm, &tn);
    } else {
      tile_of_block(blockIdx.x, g.tiles_m, g.tiles_n, &tm, &tn);
    }
  } else {
    tile_of_block(blockIdx.x, g.tiles_m, g.tiles_n, &tm, &tn);
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const int batch = by / g.splitk;
  const int ks = by % g.splitk;
  const int nk_total = g.K / BK;
  const int per = (nk_total + g.splitk - 1) / g.splitk;
  const int kt0 = ks * per;
  const int kt1 = min(nk_total, kt0 + per);
  if (kt0 >= kt1) return;

  const bf16* A = g.A + (long)batch * g.sA;
  const bf16* B = g.B + (long)batch * g.sB;

  // ---- LDS-DMA source addressing: instruction q covers rows (q*4+wave)*8 .. +8; lane -> row lane>>3, phys chunk lane&7
  const int lrow = lane >> 3;
  const int lchunk = (lane & 7) ^ lrow;  // logical 16B chunk fetched by this lane (row&7 == lrow)
  const bf16* a_src[4];
  const bf16* b_src[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = (q * NW + wave) * 8 + lrow;
    const int am = min(m0 + row, g.M - 1);
    const int bn = min(n0 + row, g.N - 1);
    a_src[q] = A + (long)am * g.lda + (g.a_kblk ? (long)(lchunk >> 2) * g.a_kblk + (lchunk & 3) * 8 : (long)lchunk * 8);
    b_src[q] = B + (long)bn * g.ldb + lchunk * 8;
  }
  auto issue = [&](int kt, int stage) {
    char* base = smem + stage * STAGE_BYTES;
    const long koff = (long)kt * BK;
    const long koff_a = g.a_kblk ? (long)kt * 2 * g.a_kblk : koff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int off = (q * NW + wave) * 1024;
      if ((q * NW + wave) * 8 < BM) glds16(a_src[q] + koff_a, base + off);  // (always true for the square tiles)
      glds16(b_src[q] + koff, base + TILE_BYTES + off);
    }
  };

  f32x4 acc[4][MI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (bytes) inside a tile: row*128 + ((s*4 + lane>>4) ^ (row&7))*16
  const int frow = lane & 15, fg = lane >> 4, fsw = lane & 7;
  const int a_off0 = (wm * WROWS + frow) * 128;                 // + i*16*128 per M tile
  const int b_off0 = TILE_BYTES + (wn * 64 + frow) * 128;       // + i*16*128 per N tile

  auto mfma_block = [&](const bf16x8* af, const bf16x8* bfg) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[ni], af[mi], acc[ni][mi], 0, 0, 0);
  };
  auto read_frags = [&](const char* base, int s_, bf16x8* af, bf16x8* bfg) {
    const int pc = ((s_ * 4 + fg) ^ fsw) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) bfg[i] = *(const bf16x8*)(base + b_off0 + i * 2048 + pc);
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(base + a_off0 + i * 2048 + pc);
  };
  if constexpr (NW == 8 && (SCHED == 1 || SCHED == 4)) {
    // ---- counted-vmcnt schedule: every tile is requested TWO K-steps before it is consumed (raw s_barrier +
    // explicit s_waitcnt: __syncthreads() would drain the in-flight DMA).
    issue(kt0, 0);
    if (kt0 + 1 < kt1) issue(kt0 + 1, 1);
    for (int kt = kt0; kt < kt1; ++kt) {
      const int stage = (kt - kt0) & 1;
      const char* base = smem + stage * STAGE_BYTES;
      if (kt + 1 < kt1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      bf16x8 af0[MI], bf0[4], af1[MI], bf1[4];
      read_frags(base, 0, af0, bf0);
      if constexpr (SCHED == 1) mfma_block(af0, bf0);
      read_frags(base, 1, af1, bf1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment of this stage is in registers
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 2 < kt1) issue(kt + 2, stage);
      if constexpr (SCHED == 4) mfma_block(af0, bf0);
      mfma_block(af1, bf1);
    }
    __builtin_amdgcn_s_barrier();  // all LDS tile reads retired before the epilogue reuses the memory
  } else if constexpr (SCHED == 5) {
    // ---- ring of NS = 3 stages for the tall, narrow problems (adapter bottleneck projections: about one 64x128 workgroup
    // per CU, so nothing else on the CU hides a load): two K-steps of operands in flight, ONE raw barrier per K-step.
    //   iteration kt:  wait until stage kt has landed (one younger stage may stay in flight) -> barrier (every wave's part
    //   of stage kt is in LDS AND every wave is done reading stage kt-1) -> request stage kt+2 into the slot of kt-1 ->
    //   fragments + MFMAs of stage kt.
    constexpr int NS = 3;
    constexpr int PS = 4 + ((BM + NW * 8 - 1) / (NW * 8) < 4 ? (BM + NW * 8 - 1) / (NW * 8) : 4);  // LDS-DMA requests per thread and stage
    static_assert(PS == 6, "the vmcnt immediates below are written for the 64-row configuration (2 A + 4 B requests)");
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
      if (kt0 + s_ < kt1) issue(kt0 + s_, s_);
    for (int kt = kt0; kt < kt1; ++kt) {
      if (kt + 1 < kt1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + NS - 1 < kt1) issue(kt + NS - 1, (kt - kt0 + NS - 1) % NS);
      const char* base = smem + ((kt - kt0) % NS) * STAGE_BYTES;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[MI], bfg[4];
        read_frags(base, s, af, bfg);
        mfma_block(af, bfg);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of the stage are complete
    }
    __builtin_amdgcn_s_barrier();  // all LDS tile reads retired before the epilogue reuses the memory
  } else if constexpr (KSKIP) {
    // K-steps whose operand block is known to be all zero (rows beyond a sample's last valid position in the G^T
    // operand of the position-table gradients) are neither fetched nor multiplied.  The problem is a pure stream -- 332 x 64
    // outputs against megabytes of G^T per (layer execution, head), N = 64 -- so what matters is how many bytes a workgroup
    // keeps in flight: the valid K-steps are listed ONCE (wave 0, ballot compaction into LDS: no scalar table lookup in
    // front of every request) and run through a ring of three 24 KiB stages (A 128 rows, B 64 rows: the clamped duplicate
    // rows of a 128-row B image are not fetched) with counted vmcnt and one raw barrier per step, as in the SCHED 5 ring:
    // two steps of operands in flight instead of one fetched behind a drained queue (round 4: 2-stage loop, __syncthreads
    // per step, a dependent scalar load + division per step: 3.3 us per K-step).
    constexpr int KS_STAGE = Cfg::TILE_BYTES + 64 * BK * 2;  // 16 KiB + 8 KiB
    constexpr int KS_NS = 3;
    int16_t* klist = (int16_t*)(smem + KS_NS * KS_STAGE);    // up to KSKIP_MAX_STEPS entries (host-checked)
    int* kcount = (int*)(smem + KS_NS * KS_STAGE + KSKIP_MAX_STEPS * 2);
    if (wave == 0) {
      int cnt = 0;
      for (int base = kt0; base < kt1; base += 64) {
        const int kt = base + lane;
        bool valid = false;
        if (kt < kt1) {
          const int b = kt / g.kskip_steps;
          valid = (kt - b * g.kskip_steps) * BK < g.kskip_len[b];
          // (the 128-row tile of A this workgroup owns is all zero -- and unwritten -- in k-steps whose mask bit is clear)
          if (valid && g.kskip_tilemask) valid = (g.kskip_tilemask[kt] >> tm) & 1u;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(valid);
        if (valid) klist[cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int16_t)kt;
        cnt += __builtin_popcountll(m);
      }
      if (lane == 0) *kcount = cnt;
    }
    __syncthreads();
    const int nv = __builtin_amdgcn_readfirstlane(*kcount);
    auto issue_ks = [&](int i, int slot) {
      const int kt = __builtin_amdgcn_readfirstlane((int)klist[i]);
      char* base = smem + slot * KS_STAGE;
      const long koff = (long)kt * BK;
      const long koff_a = g.a_kblk ? (long)kt * 2 * g.a_kblk : koff;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int off = (q * NW + wave) * 1024;
        glds16(a_src[q] + koff_a, base + off);
        if ((q * NW + wave) * 8 < 64) glds16(b_src[q] + koff, base + Cfg::TILE_BYTES + off);  // (q < 2: six requests per thread)
      }
    };
#pragma unroll
    for (int s_ = 0; s_ < KS_NS - 1; ++s_)
      if (s_ < nv) issue_ks(s_, s_);
    for (int i = 0; i < nv; ++i) {
      if (i + 1 < nv) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (i + KS_NS - 1 < nv) issue_ks(i + KS_NS - 1, (i + KS_NS - 1) % KS_NS);
      const char* base = smem + (i % KS_NS) * KS_STAGE;
      if (n0 + wn * 64 < g.N) {  // (N = 64 here: the second column of waves only helps with the staging)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          bf16x8 af[MI], bfg[4];
          read_frags(base, s, af, bfg);
          mfma_block(af, bfg);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's fragment reads of the stage are complete
    }
    __builtin_amdgcn_s_barrier();  // all LDS tile reads retired before the epilogue reuses the memory
  } else {
  issue(kt0, 0);
  __syncthreads();  // drains the LDS-DMA (vmcnt(0)) and publishes stage 0
  for (int kt = kt0; kt < kt1; ++kt) {
    const int stage = (kt - kt0) & 1;
    if (kt + 1 < kt1) issue(kt + 1, stage ^ 1);
    const char* base = smem + stage * STAGE_BYTES;
    if constexpr (SCHED == 2) {  // all 24 fragment reads first, then 64 MFMAs
      bf16x8 af0[MI], bf0[4], af1[MI], bf1[4];
      read_frags(base, 0, af0, bf0);
      read_frags(base, 1, af1, bf1);
      mfma_block(af0, bf0);
      mfma_block(af1, bf1);
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[MI], bfg[4];
        read_frags(base, s, af, bfg);
        if constexpr (SCHED == 3) __builtin_amdgcn_s_setprio(1);
        mfma_block(af, bfg);
        if constexpr (SCHED == 3) __builtin_amdgcn_s_setprio(0);
      }
    }
    __syncthreads();
  }
  }

  // ---- epilogue (gemm_common.h): slabs of 64 rows of this wave's tile through a wave-private LDS staging tile
  gemm_epilogue<ACT, AUX, SPLITK, MI>(g, smem, wave, lane, acc, m0 + wm * WROWS, 64, n0 + wn * 64 + (lane & 15) * 4, batch, ks);
}

// ---------------------------------------------------------------------------------------------------------------
// "TN" variant for the trainable-weight gradients:  C[M,N] += sum_k A[k,m] * B[k,n]   (A [K,M], B [K,N] row-major, the
// contraction runs over ROWS: dW = X^T . dY without materialising X^T / dY^T in HBM).  128x128x64 tile, 4 waves.
// Tiles are staged row-major in LDS (coalesced 16-byte global loads, next tile prefetched into registers) and the
// MFMA fragments -- 8 consecutive k for one m -- come out of the row-major image through the hardware transpose read
// ds_read_b64_tr_b16 (two per fragment).
// Always split-K with a workspace fold (accumulates into C).
struct GemmTnArgs {
  const bf16* A; const bf16* B; long lda, ldb;
  int M, N, K;
  float* ws; int Nw;  // partials [splitk][M][Nw]
  int splitk, tiles_m, tiles_n;
};
typedef __attribute__((ext_vector_type(4))) short tr16x4;
__device__ __forceinline__ tr16x4 lds_tr16(const bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr16x4*)p);
}
constexpr int TN_LD = 136;  // bf16 row stride of the staged [64 k][128 cols] tiles (272 B: 16B-aligned rows)

__global__ __launch_bounds__(256, 2) void gemm_bf16_tn_kernel(GemmTnArgs g) {
  __shared__ __attribute__((aligned(16))) bf16 sA[64 * TN_LD];
  __shared__ __attribute__((aligned(16))) bf16 sB[64 * TN_LD];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tm = blockIdx.x % g.tiles_m, tn = blockIdx.x / g.tiles_m;
  const int m0 = tm * 128, n0 = tn * 128;
  const int ks = blockIdx.y;
  const int nk = (g.K + 63) / 64;
  const int per = (nk + g.splitk - 1) / g.splitk;
  const int kt0 = ks * per, kt1 = min(nk, kt0 + per);
  const int frow = lane & 15, fg = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging role: 64 rows x 16 chunks(8 cols) per operand = 1024 chunks -> 4 per thread
  bf16x8 ra[4], rb[4];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int id = tid + t * 256;
      const int r = id >> 4, ch = id & 15;
      const int k = kt * 64 + r;
      const bool kok = k < g.K;
      const int ca = m0 + ch * 8, cb = n0 + ch * 8;
      ra[t] = (kok && ca + 8 <= g.M) ? *(const bf16x8*)(g.A + (long)k * g.lda + ca) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      rb[t] = (kok && cb + 8 <= g.N) ? *(const bf16x8*)(g.B + (long)k * g.ldb + cb) : (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    }
  };
  if (kt0 < kt1) load_tile(kt0);
  for (int kt = kt0; kt < kt1; ++kt) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int id = tid + t * 256;
      *(bf16x8*)(sA + (id >> 4) * TN_LD + (id & 15) * 8) = ra[t];
      *(bf16x8*)(sB + (id >> 4) * TN_LD + (id & 15) * 8) = rb[t];
    }
    __syncthreads();
    if (kt + 1 < kt1) load_tile(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[4], bfg[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // ds_read_b64_tr_b16: the 16 lanes of a group hand in the 4x16 block (rows k..k+3, 16 columns) as 16 x 8 bytes
        // and each lane gets its own column back -- 4 consecutive k of one m: two reads per MFMA fragment instead
        // of eight ds_read_u16 (semantics probed on hardware: tools/probe/tr16_probe.hip)
        const int kr = (s * 32 + fg * 8 + (frow >> 2)) * TN_LD + (frow & 3) * 4;
        const tr16x4 a0 = lds_tr16(sA + kr + wm * 64 + i * 16), a1 = lds_tr16(sA + kr + 4 * TN_LD + wm * 64 + i * 16);
        const tr16x4 b0 = lds_tr16(sB + kr + wn * 64 + i * 16), b1 = lds_tr16(sB + kr + 4 * TN_LD + wn * 64 + i * 16);
        union { tr16x4 h[2]; bf16x8 v; } ua, ub;
        ua.h[0] = a0; ua.h[1] = a1; ub.h[0] = b0; ub.h[1] = b1;
        af[i] = ua.v;
        bfg[i] = ub.v;
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfg[ni], af[mi], acc[ni][mi], 0, 0, 0);
    }
    __syncthreads();
  }
  // partial tile -> workspace (lane owns C[m][4 consecutive n]); rows/cols beyond M/N are skipped
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    const int m = m0 + wm * 64 + mi * 16 + frow;
    if (m >= g.M) continue;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n4 = n0 + wn * 64 + ni * 16 + fg * 4;
      if (n4 >= g.N) continue;
      *(f32x4*)(g.ws + (((long)ks * g.M + m) * g.Nw + n4)) = acc[ni][mi];
    }
  }
}

// 256x256 (or 224x256) tiles only where both dimensions fill them and the grid still covers half the chip
static inline bool big_tile_shape(int M, int N, int batch) {
  // (M >= 512 with a very wide N: the vocabulary GEMM of the loss on the labelled rows, [~700 x 128100 x 1536] -- 3 x 501 big tiles)
  static const int wide = FBL_ENV_INT("FBL_GEMM_WIDE", 1);
  return batch == 1 && (M >= 2048 || (wide && M >= 512 && N >= 16384)) && N >= 1024 &&
         ((long)((M + 255) / 256) * ((N + 255) / 256) >= 128);
}

constexpr double FBL_R128_US_PER_KTILE = 0.86;  // 128-row 8-phase tile: time per K-tile of one workgroup (measured: [5322,1536,6144] 88 us)

static int device_cu_count() {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
    if (n_cu <= 0) n_cu = 256;
  }
  return n_cu;
}


// out[b][m][n] += sum_ks ws[b][ks][m][n]   (deterministic split-K fold)
__global__ void splitk_reduce_kernel(const float* ws, int splitk, int M, int N, int Nw, float* out, long ldc, long sC) {
  const int b = blockIdx.y;
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index inside [M][Nw/4]
  const int nq = Nw >> 2;
  if (q >= (long)M * nq) return;
  const int m = (int)(q / nq), n4 = (int)(q % nq) * 4;
  const float* p = ws + ((long)b * splitk * M + m) * Nw + n4;
  const long ks = (long)M * Nw;
  float* o = out + b * sC + (long)m * ldc + n4;
  const bool vec = (n4 + 3 < N) && ((ldc & 3) == 0) && ((sC & 3) == 0);
  f32x4 acc = vec ? *(const f32x4*)o : (f32x4){0.f, 0.f, 0.f, 0.f};
  // partials are summed in index order (deterministic); 8 independent loads in flight per thread
  int k = 0;
  for (; k + 8 <= splitk; k += 8) {
    f32x4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = *(const f32x4*)(p + (k + u) * ks);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += t[u];
  }
  for (; k < splitk; ++k) acc += *(const f32x4*)(p + k * ks);
  if (vec) {
    *(f32x4*)o = acc;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (n4 + r < N) o[r] += acc[r];
  }
}

}  // namespace

// Fork / join events of one (stream, aux_stream) pair.  The library owns no stream; the events are its only process
// state: created on first use for the pair (on the device that is current in the calling thread, which must be the
// streams' device), kept for the life of the process, looked up under a mutex.  One pair of events per pair of streams
// is enough: a stream is fed by one thread at a time, and re-recording an event only affects waits issued afterwards.
#include <map>
#include <mutex>
#include <utility>
static bool fork_join_events(hipStream_t s, hipStream_t aux, hipEvent_t* fork, hipEvent_t* join) {
  static std::mutex mu;
  static std::map<std::pair<hipStream_t, hipStream_t>, std::pair<hipEvent_t, hipEvent_t>> pool;
  std::lock_guard<std::mutex> lock(mu);
  auto it = pool.find({s, aux});
  if (it == pool.end()) {
    hipEvent_t f = nullptr, j = nullptr;
    if (hipEventCreateWithFlags(&f, hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&j, hipEventDisableTiming) != hipSuccess) {
      (void)hipEventDestroy(f);
      return false;
    }
    it = pool.emplace(std::make_pair(s, aux), std::make_pair(f, j)).first;
  }
  *fork = it->second.first;
  *join = it->second.second;
  return true;
}

// residual operand of the adapter-tail epilogue (fbl_adapter_up_resid_fwd)
struct TailArgs {
  const float* r_t; int64_t ld_r; const float* r_stats; const float* r_gamma; const float* r_beta; const int32_t* r_rowmask;
};

// p_drop > 0 (ReLU epilogue only): dropout of the activated output, element (m, n) keyed by (drop_seed, m*ldc + n)
static int gemm_nt_impl(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                        const float* bias, const float* rowscale, float alpha, int act, int aux_kind,
                        const void* aux, int64_t ld_aux, float* out_f32, void* out_bf16, void* out_pre_bf16,
                        int64_t ldc, int batch, int64_t strideA, int64_t strideB, int64_t strideC,
                        int64_t strideAux, int64_t strideBias, int splitk, float* splitk_ws,
                        int64_t splitk_ws_floats, int64_t a_kblock_stride, const int32_t* kskip_len, int kskip_steps,
                        float p_drop, uint64_t drop_seed, void* stream, int seg_n = 0, void* seg_out = nullptr,
                        int64_t seg_ld = 0, int64_t drop_row0 = 0, void* aux_stream = nullptr,
                        const TailArgs* tail = nullptr, const uint64_t* drop_seed_dev = nullptr,
                        const uint32_t* kskip_tilemask = nullptr) {
  if (M <= 0 || N <= 0 || batch <= 0) return 0;
  if ((aux_kind == FBL_AUX_ADAPTER_TAIL) != (tail != nullptr)) return FBL_ERR_ARG;
  if (K <= 0 || (K % BK) != 0) return FBL_ERR_SHAPE;           // K must be a multiple of 64 (callers zero-pad)
  if ((lda % 8) != 0 || (ldb % 8) != 0) return FBL_ERR_ALIGN;  // 16-byte operand rows
  if (splitk < 1) splitk = 1;
  const bool accumulate = splitk > 1;  // callers ask for splitk >= 2 when they want "out += A.B^T"
  if (splitk > 1) {  // every split must own at least one K tile
    const int nk = K / BK;
    const int per = (nk + splitk - 1) / splitk;
    splitk = (nk + per - 1) / per;
  }
  const int Nw = (N + 3) & ~3;
  if (accumulate && splitk_ws && (int64_t)batch * splitk * M * Nw > splitk_ws_floats) splitk_ws = nullptr;  // too small
  if (accumulate && bias && splitk_ws) return FBL_ERR_ARG;
  if (accumulate && (!out_f32 || out_bf16 || out_pre_bf16 || act != FBL_ACT_NONE || aux_kind != FBL_AUX_NONE))
    return FBL_ERR_ARG;  // split-K only accumulates (atomicAdd) into a pre-initialised fp32 output
  if (!out_f32 && !out_bf16) return FBL_ERR_ARG;
  GemmArgs g;
  g.A = (const bf16*)A; g.B = (const bf16*)B; g.lda = lda; g.ldb = ldb;
  g.M = M; g.N = N; g.K = K;
  g.bias = bias; g.rowscale = rowscale; g.alpha = alpha; g.act = act; g.aux_kind = aux_kind;
  g.aux = aux; g.ld_aux = ld_aux;
  g.out_f32 = out_f32; g.out_bf16 = (bf16*)out_bf16; g.out_pre = (bf16*)out_pre_bf16; g.ldc = ldc;
  g.sA = strideA; g.sB = strideB; g.sC = strideC; g.sAux = strideAux; g.sBias = strideBias;
  g.splitk = splitk;
  g.k8_per = 0;
  g.ws = accumulate ? splitk_ws : nullptr;
  g.Nw = Nw;
  g.a_kblk = a_kblock_stride;
  g.kskip_len = kskip_len;
  g.kskip_steps = kskip_steps;
  g.kskip_tilemask = kskip_tilemask;
  g.drop_thresh = 0; g.drop_seed = drop_seed; g.drop_seed_dev = drop_seed_dev; g.drop_inv_keep = 1.f; g.drop_ld = ldc;
  g.seg_n = seg_n; g.seg_out = (bf16*)seg_out; g.seg_ld = seg_ld; g.drop_row0 = drop_row0;
  g.r_t = nullptr; g.ld_r = 0; g.r_stats = nullptr; g.r_gamma = nullptr; g.r_beta = nullptr; g.r_rowmask = nullptr;
  if (tail) {
    if (accumulate || batch != 1 || seg_n > 0 || !out_f32 || out_bf16 || out_pre_bf16 || act != FBL_ACT_NONE || rowscale ||
        !aux || !tail->r_t || (N & 3) || (ldc & 3) || (ld_aux & 3) || (tail->ld_r & 3) ||
        (tail->r_stats && (!tail->r_gamma || !tail->r_beta)))
      return FBL_ERR_ARG;
    g.r_t = tail->r_t; g.ld_r = tail->ld_r; g.r_stats = tail->r_stats; g.r_gamma = tail->r_gamma; g.r_beta = tail->r_beta;
    g.r_rowmask = tail->r_rowmask;
    g.drop_ld = N;  // keys of fbl_ln_fwd: (seed, m*H + n)
  }
  if (seg_n > 0) {
    // The epilogue branches per LANE on "column >= seg_n" but stages accumulators through per-WAVE LDS patches that all 64
    // lanes fill: the segment boundary must not cut a wave's column range.  A wave of the 2-stage kernels owns 64
    // contiguous columns, a wave of the 256-wide tiles (2-stage and 8-phase) two 32-column ranges 128 apart -> the
    // boundary has to be a multiple of 64, and of 256 for the wide tiles (otherwise the narrow tiles take the problem).
    if ((seg_n & 63) || seg_n >= N || !seg_out || accumulate || batch != 1) return FBL_ERR_ARG;
    g.drop_ld = seg_ld;
  }
  if (p_drop > 0.f) {
    if ((act != FBL_ACT_RELU && seg_n <= 0 && !tail) || p_drop >= 1.f || batch != 1) return FBL_ERR_ARG;
    g.drop_thresh = fbl_drop_thresh(p_drop);
    g.drop_inv_keep = 1.f / (1.f - p_drop);
  }
  if (kskip_len && (kskip_steps <= 0 || !accumulate)) return FBL_ERR_ARG;  // only the split-K (accumulating) path skips
  if (kskip_tilemask && (!kskip_len || M > 32 * 128)) return FBL_ERR_ARG;
  // big tiles only where both dimensions fill them and the grid still covers the chip
  static const int force_small = FBL_ENV_INT("FBL_GEMM_SMALL", 0);
  const bool big = !force_small && !accumulate && (p_drop <= 0.f || seg_n > 0 || tail) && (seg_n <= 0 || (seg_n & 255) == 0) &&
                   big_tile_shape(M, N, batch);
  // A multi-round problem of the 8-phase kernel whose partial last round still uses a good part of the chip (96..192 of 256
  // CUs; the QKV projection [9024,4608,1536]: 648 tiles of 256 rows, 738 of 224) runs as ONE plain launch instead of "whole
  // rounds + 128x128 remainder" (122 us as 224-row tiles; the split: 146).  With a nearly empty last round (FFN-up: 816
  // tiles, 48 left) the split stays better: an epilogue costs a CU 10-20 us of VALU / store time that nothing on that CU
  // overlaps, so a fourth round of full tiles (222 us) loses to three rounds plus small tiles (208 us).
  // (Rounds 2-3 delayed part of the first round by a spin-wait on the real-time clock to take the CUs out of lockstep: worth
  //  13 us on the 256-row launch, nothing on the 224-row one the problem takes now -- removed.)
  static const int gemm8_on = FBL_ENV_INT("FBL_GEMM8", 3);
  bool single_launch = false;
  if (big && splitk_ws_floats >= 0 && gemm8_on > 0 && gemm8_eligible(g)) {
    const long total = (long)((N + 255) / 256) * ((M + 255) / 256);
    const long rem = total % 256;
    single_launch = total > 256 && rem >= 96 && rem <= 192;
  }
  if (big && splitk_ws_floats >= 0 && !single_launch && !tail) {  // (negative values mark the two halves of an already split launch)
    // Wave quantisation: one 256x256 workgroup per CU, so a grid of T tiles costs ceil(T/CUs) rounds.  When the last
    // round would be mostly empty, give the big tiles only as many M rows as fill whole rounds and run the remaining
    // rows with the 128x128 configuration (2 workgroups/CU, 1/4 of the work per tile) right behind.
    static int n_cu = 0;
    if (!n_cu) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
      if (n_cu <= 0) n_cu = 256;
    }
    const int tn = (N + 255) / 256, tm = (M + 255) / 256;
    const long total = (long)tn * tm;
    const long rem = total % n_cu;
    if (total > n_cu && rem != 0 && rem * 3 < (long)n_cu * 2) {  // last round less than 2/3 full
      const int tm_big = (int)((total - rem) / tn);             // whole rounds worth of M tiles
      const int m_big = tm_big * 256;
      if (tm_big >= 1 && m_big < M && M - m_big >= 64 && !splitk_ws) {
        // The remainder rows run on the CALLER's aux_stream (if one is given), launched BEFORE the big tiles and forked
        // from / joined back into `stream` by events: its workgroups take their CUs first, so those CUs reach their first
        // big tile a fraction of a tile late -- the chip leaves lockstep (the epilogue of a round is an HBM burst: every CU
        // stores its tile at the same moment while the memory system idles during the main loops) and the remainder costs
        // no round of its own.  Without an aux stream the remainder simply precedes the big tiles on `stream`.
        // FBL_GEMM_REM (debug builds) bit 0: use the aux stream; bit 1: remainder in 64x128 tiles (320 rows x 6144: 240
        // workgroups instead of 144).
        static const int rem_mode = FBL_ENV_INT("FBL_GEMM_REM", 3);
        bool forked = false;
        hipEvent_t ev_fork = nullptr, ev_join = nullptr;
        void* s_rem = stream;
        if ((rem_mode & 1) && aux_stream && aux_stream != stream) {
          if (!fork_join_events((hipStream_t)stream, (hipStream_t)aux_stream, &ev_fork, &ev_join)) return FBL_ERR_ARG;
          if (hipEventRecord(ev_fork, (hipStream_t)stream) != hipSuccess) return FBL_ERR_ARG;
          if (hipStreamWaitEvent((hipStream_t)aux_stream, ev_fork, 0) != hipSuccess) return FBL_ERR_ARG;
          s_rem = aux_stream;
          forked = true;
        }
        const size_t aux_es = (aux_kind == FBL_AUX_ADD_F32) ? 4 : 2;
        int rc = gemm_nt_impl((const char*)A + (size_t)m_big * lda * 2, lda, B, ldb, M - m_big, N, K, bias,
                              rowscale ? rowscale + m_big : nullptr, alpha, act, aux_kind,
                              aux ? (const char*)aux + (size_t)m_big * ld_aux * aux_es : nullptr, ld_aux,
                              out_f32 ? out_f32 + (size_t)m_big * ldc : nullptr,
                              out_bf16 ? (char*)out_bf16 + (size_t)m_big * ldc * 2 : nullptr,
                              out_pre_bf16 ? (char*)out_pre_bf16 + (size_t)m_big * ldc * 2 : nullptr, ldc, 1, 0, 0, 0, 0, 0,
                              1, nullptr, (rem_mode & 2) ? -3 : -2, a_kblock_stride, nullptr, 0, p_drop, drop_seed, s_rem, seg_n,
                              seg_out ? (char*)seg_out + (size_t)m_big * seg_ld * 2 : nullptr, seg_ld, drop_row0 + m_big, nullptr,
                              nullptr, drop_seed_dev);
        if (rc) return rc;
        if (forked && hipEventRecord(ev_join, (hipStream_t)aux_stream) != hipSuccess) return FBL_ERR_ARG;
        rc = gemm_nt_impl(A, lda, B, ldb, m_big, N, K, bias, rowscale, alpha, act, aux_kind, aux, ld_aux, out_f32,
                          out_bf16, out_pre_bf16, ldc, 1, 0, 0, 0, 0, 0, 1, nullptr, -1, a_kblock_stride, nullptr, 0,
                          p_drop, drop_seed, stream, seg_n, seg_out, seg_ld, drop_row0, nullptr, nullptr, drop_seed_dev);
        if (rc) return rc;
        if (forked && hipStreamWaitEvent((hipStream_t)stream, ev_join, 0) != hipSuccess) return FBL_ERR_ARG;
        return 0;
      }
    }
  }
  const bool use_big = big && splitk_ws_floats > -2;  // -2 / -3: remainder rows of a split launch -> 128x128 / 64x128 tiles
  const int BT = use_big ? 256 : 128;
  // 224x256 tiles when they cover the problem in fewer (rounds x tile area) than 256x256 -- the N = 1536 GEMMs of the
  // step: 38 x 6 = 228 tiles in one round instead of 204 tiles that are 14 % bigger
  bool use_224 = false;
  static const int no224 = FBL_ENV_INT("FBL_GEMM_NO224", 0);
  if (use_big && splitk_ws_floats >= 0 && !no224) {
    int dev = 0, n_cu = 256;
    static int cu_cached = 0;
    if (!cu_cached) {
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cu_cached = prop.multiProcessorCount;
      if (cu_cached <= 0) cu_cached = 256;
    }
    n_cu = cu_cached;
    const long tn = (N + 255) / 256;
    const long t256 = tn * ((M + 255) / 256), t224 = tn * ((M + 223) / 224);
    const long c256 = ((t256 + n_cu - 1) / n_cu) * 256, c224 = ((t224 + n_cu - 1) / n_cu) * 224;
    use_224 = c224 * 100 < c256 * 97;
  }
  // 64x128 tiles for tall, narrow problems whose 128x128 grid covers less than the chip (the adapter bottleneck
  // projections: 8512 x 192 -> 134 workgroups): twice the workgroups, so twice the CUs pull operands from L2
  static const int no64 = FBL_ENV_INT("FBL_GEMM_NO64", 0);
  const bool use_64 = !use_big && !accumulate && !no64 && batch == 1 &&
                      ((M >= 2048 && splitk_ws_floats >= 0) || splitk_ws_floats == -3) &&
                      (long)((M + 127) / 128) * ((N + 127) / 128) < 200;
  g.tiles_m = use_224 ? (M + 223) / 224 : use_64 ? (M + 63) / 64 : (M + BT - 1) / BT;
  g.tiles_n = (N + BT - 1) / BT;
  dim3 grid(g.tiles_m * g.tiles_n, batch * splitk);
  // The 8-phase kernel (gemm8.hip) takes every launch of the 256- / 224-row configurations it is instantiated for.
  // FBL_GEMM8=0 switches it off, 1 keeps the 224x256 shapes on the 2-stage kernel, 2 runs them as 256x256 8-phase tiles
  // (measured: [8512,1536,6144] 915 -> 1139 TFLOP/s although only 204 of 256 CUs get a tile), 3 (default) as 224x256
  // 8-phase tiles (228 tiles).
  static const int gemm8_mode = FBL_ENV_INT("FBL_GEMM8", 3);
  if (use_big && gemm8_mode > 0 && (!use_224 || gemm8_mode >= 2) && gemm8_eligible(g)) {
    GemmArgs g8 = g;
    const bool r224 = use_224 && gemm8_mode >= 3;
    for (int attempt = r224 ? 0 : 1; attempt < 2; ++attempt) {  // 224-row tiles first where they apply, then 256-row tiles
      const int bm = attempt == 0 ? 224 : 256;
      g8.tiles_m = (M + bm - 1) / bm;
      g8.tiles_n = (N + 255) / 256;
      const int rc8 = launch_gemm8(g8, act, aux_kind, bm, dim3(g8.tiles_m * g8.tiles_n, 1), (hipStream_t)stream);
      if (rc8 != FBL_ERR_ARG) return rc8;
    }
  }
  // 128-row 8-phase tiles for problems the tall tiles leave half the chip idle on (the N = 1536 dX GEMMs at the row counts
  // of packed ragged batches / small batches: M = 5322 -> 21 x 6 = 126 tiles of 256 rows, 42 x 6 = 252 of 128 rows) and
  // that would otherwise run on the 128x128 two-stage kernel.  Chosen by a two-line cost model from measured per-K-tile
  // times (tools/bench_gemm.py --set packed): rounds x (K-tiles x us per K-tile + epilogue).  FBL_GEMM_R128 (measurement
  // builds): 0 = never, 2 = whenever the kernel can take the launch.
  static const int r128_mode = FBL_ENV_INT("FBL_GEMM_R128", 1);
  if (!use_big && r128_mode > 0 && gemm8_mode > 0 && !accumulate && batch == 1 && splitk_ws_floats >= 0 && !tail && seg_n <= 0 &&
      p_drop <= 0.f && M >= 1024 && N >= 1024 && N <= 8192 && gemm8_eligible(g)) {
    const int n_cu = device_cu_count();
    const long t128 = (long)((M + 127) / 128) * ((N + 255) / 256), tsm = (long)((M + 127) / 128) * ((N + 127) / 128);
    const double nk = K / 64.0;
    const double us128 = (double)((t128 + n_cu - 1) / n_cu) * (nk * FBL_R128_US_PER_KTILE + 6.0);
    const double ussm = (double)((tsm + 2 * n_cu - 1) / (2 * n_cu)) * (nk * 1.17 + 5.0);
    if (r128_mode >= 2 || us128 < 0.95 * ussm) {
      GemmArgs g8 = g;
      g8.tiles_m = (M + 127) / 128;
      g8.tiles_n = (N + 255) / 256;
      const int rc8 = launch_gemm8(g8, act, aux_kind, 128, dim3(g8.tiles_m * g8.tiles_n, 1), (hipStream_t)stream);
      if (rc8 != FBL_ERR_ARG) return rc8;
    }
  }
#define FBL_GEMM_LAUNCH_NW(NW_, ACT_, AUX_, SK_, MI_)                                                           \
  do {                                                                                                         \
    static bool attr_set = false;                                                                              \
    auto kfn = gemm_bf16_nt_kernel<NW_, ACT_, AUX_, SK_, 0, MI_>;                                              \
    constexpr int smem_bytes = TileCfg<NW_>::SMEM_BYTES;                                                       \
    if (!attr_set) {                                                                                           \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes); \
      if (e != hipSuccess) return (int)e;                                                                      \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    hipLaunchKernelGGL(kfn, grid, dim3(NW_ * 64), smem_bytes, (hipStream_t)stream, g);                         \
  } while (0)
  static const int deep64 = FBL_ENV_INT("FBL_GEMM_DEEP", 1);
#define FBL_GEMM_LAUNCH_DEEP(ACT_, AUX_)                                                                        \
  do {                                                                                                         \
    static bool attr_set = false;                                                                              \
    auto kfn = gemm_bf16_nt_kernel<4, ACT_, AUX_, false, 5, 2>;                                                \
    constexpr int smem_bytes = 3 * (64 * BK * 2 + TileCfg<4>::TILE_BYTES);                                     \
    static_assert(smem_bytes >= 4 * 64 * 68 * 4, "ring must cover the epilogue staging");                      \
    if (!attr_set) {                                                                                           \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes); \
      if (e != hipSuccess) return (int)e;                                                                      \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    hipLaunchKernelGGL(kfn, grid, dim3(256), smem_bytes, (hipStream_t)stream, g);                              \
  } while (0)
#define FBL_GEMM_LAUNCH(ACT_, AUX_, SK_)                              \
  do {                                                                \
    if (use_224) FBL_GEMM_LAUNCH_NW(8, ACT_, AUX_, false, 7);         \
    else if (use_big) FBL_GEMM_LAUNCH_NW(8, ACT_, AUX_, false, 8);    \
    else if (use_64 && deep64) FBL_GEMM_LAUNCH_DEEP(ACT_, AUX_);      \
    else if (use_64) FBL_GEMM_LAUNCH_NW(4, ACT_, AUX_, false, 2);     \
    else FBL_GEMM_LAUNCH_NW(4, ACT_, AUX_, SK_, 4);                   \
  } while (0)
#define FBL_GEMM_LAUNCH_SCHED(SCHED_)                                                                          \
  do {                                                                                                         \
    static bool attr_set = false;                                                                              \
    auto kfn = gemm_bf16_nt_kernel<8, FBL_ACT_NONE, FBL_AUX_NONE, false, SCHED_>;                              \
    constexpr int smem_bytes = TileCfg<8>::SMEM_BYTES;                                                         \
    if (!attr_set) {                                                                                           \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes); \
      if (e != hipSuccess) return (int)e;                                                                      \
      attr_set = true;                                                                                         \
    }                                                                                                          \
    hipLaunchKernelGGL(kfn, grid, dim3(512), smem_bytes, (hipStream_t)stream, g);                              \
  } while (0)
  static const int exp_sched = FBL_ENV_INT("FBL_GEMM_SCHED", 0);
  if (use_big && exp_sched > 0 && act == FBL_ACT_NONE && aux_kind == FBL_AUX_NONE) {  // experiment switch (plain epilogue)
    if (exp_sched == 1) FBL_GEMM_LAUNCH_SCHED(1);
    else if (exp_sched == 2) FBL_GEMM_LAUNCH_SCHED(2);
    else if (exp_sched == 3) FBL_GEMM_LAUNCH_SCHED(3);
    else FBL_GEMM_LAUNCH_SCHED(4);
    FBL_CHECK_LAUNCH();
    return 0;
  }
  // Few rows against a very long K with a workspace at hand (the prediction head's backward dh = dlogits . E: [~700 x 1536 x
  // 128128]): 8-phase 256 x 256 tiles, K cut into as many slices as fill the chip (18 tiles x 14 slices), partial tiles folded by
  // splitk_reduce_kernel as below -- 463 -> ~235 us.  FBL_GEMM8_SK (measurement builds): 0 = the 128 x 128 two-stage kernel.
  static const int sk8_on = FBL_ENV_INT("FBL_GEMM8_SK", 1);
  bool done = false;
  if (accumulate && sk8_on && g.ws && batch == 1 && !kskip_len && !a_kblock_stride && M >= 512 && N >= 1024 && (K / BK) % 2 == 0 &&
      K / BK >= 128 && (long)M * lda * 2 < (1l << 32) && (long)N * ldb * 2 < (1l << 32)) {
    const int nk = K / BK, tiles = ((M + 255) / 256) * ((N + 255) / 256), n_cu = device_cu_count();
    int want = n_cu / tiles;
    if (want >= 2) {
      if (want > nk / 8) want = nk / 8;
      int per = (nk + want - 1) / want;
      per += per & 1;  // even
      int s8 = (nk + per - 1) / per;
      if (nk - (s8 - 1) * per < 4) {  // a last slice of 2 K-tiles is below the kernel's shortest pipeline
        per += 2;
        s8 = (nk + per - 1) / per;
      }
      if (s8 >= 2 && per >= 8 && nk - (s8 - 1) * per >= 4 && (int64_t)s8 * M * Nw <= splitk_ws_floats) {
        GemmArgs g8 = g;
        g8.splitk = s8;
        g8.k8_per = per;
        g8.tiles_m = (M + 255) / 256;
        g8.tiles_n = (N + 255) / 256;
        const int rc8 = launch_gemm8_splitk(g8, (hipStream_t)stream);
        if (rc8) return rc8;
        splitk = s8;
        done = true;
      }
    }
  }
  if (done) {
  } else if (accumulate && kskip_len) {
    static bool attr_ks = false;
    auto kfn = gemm_bf16_nt_kernel<4, FBL_ACT_NONE, FBL_AUX_NONE, true, 0, 4, true>;
    constexpr int smem_bytes = KSKIP_SMEM_BYTES > TileCfg<4>::SMEM_BYTES ? KSKIP_SMEM_BYTES : TileCfg<4>::SMEM_BYTES;
    if (N > 64 || (K / BK + splitk - 1) / splitk > KSKIP_MAX_STEPS) return FBL_ERR_SHAPE;  // (64 B rows staged, LDS list of valid K-steps)
    if (K / BK >= 32768) return FBL_ERR_SHAPE;  // the list holds ABSOLUTE k-step indices as int16
    if (!attr_ks) {
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
      if (e != hipSuccess) return (int)e;
      attr_ks = true;
    }
    hipLaunchKernelGGL(kfn, grid, dim3(256), smem_bytes, (hipStream_t)stream, g);
  } else if (accumulate) FBL_GEMM_LAUNCH(FBL_ACT_NONE, FBL_AUX_NONE, true);
  else if (act == FBL_ACT_GELU && aux_kind == FBL_AUX_NONE) FBL_GEMM_LAUNCH(FBL_ACT_GELU, FBL_AUX_NONE, false);
  else if (act == FBL_ACT_RELU && aux_kind == FBL_AUX_NONE) FBL_GEMM_LAUNCH(FBL_ACT_RELU, FBL_AUX_NONE, false);
  else if (act == FBL_ACT_GELU_GRAD && aux_kind == FBL_AUX_NONE) FBL_GEMM_LAUNCH(FBL_ACT_GELU_GRAD, FBL_AUX_NONE, false);
  else if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_MUL_BF16) FBL_GEMM_LAUNCH(FBL_ACT_NONE, FBL_AUX_MUL_BF16, false);
  else if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_NONE) FBL_GEMM_LAUNCH(FBL_ACT_NONE, FBL_AUX_NONE, false);
  else if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_ADD_F32) FBL_GEMM_LAUNCH(FBL_ACT_NONE, FBL_AUX_ADD_F32, false);
  else if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_ADD_BF16) FBL_GEMM_LAUNCH(FBL_ACT_NONE, FBL_AUX_ADD_BF16, false);
  else if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_MUL_DGELU_BF16) FBL_GEMM_LAUNCH(FBL_ACT_NONE, FBL_AUX_MUL_DGELU_BF16, false);
  else if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_MUL_POS_BF16) FBL_GEMM_LAUNCH(FBL_ACT_NONE, FBL_AUX_MUL_POS_BF16, false);
  else if (act == FBL_ACT_NONE && aux_kind == FBL_AUX_ADAPTER_TAIL) FBL_GEMM_LAUNCH(FBL_ACT_NONE, FBL_AUX_ADAPTER_TAIL, false);
  else return FBL_ERR_ARG;
#undef FBL_GEMM_LAUNCH
#undef FBL_GEMM_LAUNCH_NW
  FBL_CHECK_LAUNCH();
  if (accumulate && g.ws) {
    const long nq = (long)M * (Nw >> 2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nq + 255) / 256), batch), dim3(256), 0, (hipStream_t)stream,
                       (const float*)g.ws, splitk, M, N, Nw, out_f32, (long)ldc, (long)strideC);
  }
  FBL_CHECK_LAUNCH();
  return 0;
}

// Host-side query (no launch): which kernel fbl_gemm_bf16_nt gives a plain launch of this shape to -- 8: the 8-phase
// kernel of gemm8.hip (whole rounds; a remainder of rows may go to 64x128 tiles of the 2-stage kernel), 2: the 2-stage kernel.
extern "C" int fbl_gemm_plan(int M, int N, int K, int batch, int splitk) {
  static const int force_small = FBL_ENV_INT("FBL_GEMM_SMALL", 0);
  static const int gemm8_on = FBL_ENV_INT("FBL_GEMM8", 3);
  if (force_small || gemm8_on <= 0 || splitk > 1 || !big_tile_shape(M, N, batch)) return 2;
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.splitk = 1;
  return gemm8_eligible(g) ? 8 : 2;
}

extern "C" int fbl_gemm_bf16_nt(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                                const float* bias, const float* rowscale, float alpha, int act, int aux_kind,
                                const void* aux, int64_t ld_aux, float* out_f32, void* out_bf16, void* out_pre_bf16,
                                int64_t ldc, int batch, int64_t strideA, int64_t strideB, int64_t strideC,
                                int64_t strideAux, int64_t strideBias, int splitk, float* splitk_ws,
                                int64_t splitk_ws_floats, int64_t a_kblock_stride, const int32_t* kskip_len, int kskip_steps,
                                const uint32_t* kskip_tilemask, void* stream, void* aux_stream) {
  if (aux_kind == FBL_AUX_ADAPTER_TAIL) return FBL_ERR_ARG;  // (has its own entry point: fbl_adapter_up_resid_fwd)
  return gemm_nt_impl(A, lda, B, ldb, M, N, K, bias, rowscale, alpha, act, aux_kind, aux, ld_aux, out_f32, out_bf16,
                      out_pre_bf16, ldc, batch, strideA, strideB, strideC, strideAux, strideBias, splitk, splitk_ws,
                      splitk_ws_floats, a_kblock_stride, kskip_len, kskip_steps, 0.f, 0, stream, 0, nullptr, 0, 0, aux_stream,
                      nullptr, nullptr, kskip_tilemask);
}

// z[M, A] = dropout(relu(x[M,K] . Wd[A,K]^T + bd)): the adapter's down-projection with ReLU AND dropout in the GEMM
// epilogue (one launch instead of GEMM + fbl_dropout_bf16).  Element (m, a) is keyed by (seed, m*ldz + a).
extern "C" int fbl_adapter_down_fwd(const void* x_bf16, int64_t ldx, const void* wd_bf16, int64_t ldw, int M, int A, int K,
                                    const float* bias, float p_drop, uint64_t seed, const uint64_t* seed_dev, void* z_bf16,
                                    int64_t ldz, void* stream) {
  return gemm_nt_impl(x_bf16, ldx, wd_bf16, ldw, M, A, K, bias, nullptr, 1.0f, FBL_ACT_RELU, FBL_AUX_NONE, nullptr, 0,
                      nullptr, z_bf16, nullptr, ldz, 1, 0, 0, 0, 0, 0, 1, nullptr, 0, 0, nullptr, 0, p_drop, seed, stream, 0,
                      nullptr, 0, 0, nullptr, nullptr, seed_dev);
}

// One GEMM for a dense layer AND the down-projection of the adapter that follows it (model/deberta.py:255-257, 329-331:
// dense -> adapter; model/adapter.py:38-41): with Wm = [W ; Wd.W] ([N1 + A, K]; the lower A rows are the down-projection
// composed with the dense weight, rebuilt by the caller whenever Wd changes) and bm = [b ; Wd.b + bd],
//     [ y | z_pre ] = x . Wm^T + bm,     C = y (fp32 and/or bf16, N1 columns),     z = dropout_p(relu(z_pre)) (bf16, A columns).
// The bottleneck activations come out of the epilogue of the tile column(s) beyond N1 -- no separate K = N1 GEMM that
// re-reads y, no launch.  Dropout keys as in fbl_adapter_down_fwd: (seed, m*ldz + a).  N1 % 64 == 0 (a wave's column range
// must not straddle the segment boundary); the 256-wide tiles additionally need N1 % 256 == 0 and are not used otherwise.
extern "C" int fbl_dense_adapter_down_fwd(const void* x_bf16, int64_t ldx, const void* wm_bf16, int64_t ldw, int M, int N1,
                                          int A, int K, const float* bias_m, float* y_f32, void* y_bf16, int64_t ldy,
                                          float p_drop, uint64_t seed, const uint64_t* seed_dev, void* z_bf16, int64_t ldz,
                                          void* stream, void* aux_stream) {
  if (A <= 0 || (N1 & 63)) return FBL_ERR_ARG;
  return gemm_nt_impl(x_bf16, ldx, wm_bf16, ldw, M, N1 + A, K, bias_m, nullptr, 1.0f, FBL_ACT_NONE, FBL_AUX_NONE, nullptr, 0,
                      y_f32, y_bf16, nullptr, ldy, 1, 0, 0, 0, 0, 0, 1, nullptr, 0, 0, nullptr, 0, p_drop, seed, stream, N1,
                      z_bf16, ldz, 0, aux_stream, nullptr, seed_dev);
}

extern "C" int fbl_adapter_up_resid_fwd(const void* z_bf16, int64_t ldz, const void* wu_bf16, int64_t ldw, int M, int H, int A,
                                        const float* bias_u, const void* x_bf16, int64_t ldx, float p_drop, uint64_t seed,
                                        const uint64_t* seed_dev, const float* r_t, int64_t ld_r, const float* r_stats, const float* r_gamma,
                                        const float* r_beta, const int32_t* r_rowmask, float* out_t, int64_t ldt,
                                        void* stream) {
  if (!z_bf16 || !wu_bf16 || !x_bf16 || !r_t || !out_t) return FBL_ERR_ARG;
  if (p_drop < 0.f || p_drop >= 1.f) return FBL_ERR_ARG;
  if (ldx % 8) return FBL_ERR_ALIGN;
  const TailArgs tail{r_t, ld_r, r_stats, r_gamma, r_beta, r_rowmask};
  return gemm_nt_impl(z_bf16, ldz, wu_bf16, ldw, M, H, A, bias_u, nullptr, 1.0f, FBL_ACT_NONE, FBL_AUX_ADAPTER_TAIL, x_bf16, ldx,
                      out_t, nullptr, nullptr, ldt, 1, 0, 0, 0, 0, 0, 1, nullptr, 0, 0, nullptr, 0, p_drop, seed, stream, 0,
                      nullptr, 0, 0, nullptr, &tail, seed_dev);
}

extern "C" int fbl_gemm_bf16_tn_acc(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K,
                                    float* out_f32, int64_t ldc, int splitk, float* splitk_ws, int64_t splitk_ws_floats,
                                    void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if ((lda % 8) || (ldb % 8) || (M % 8) || (N % 8)) return FBL_ERR_ALIGN;
  if (!out_f32 || !splitk_ws) return FBL_ERR_ARG;
  if (splitk < 1) splitk = 1;
  const int nk = (K + 63) / 64;
  const int per = (nk + splitk - 1) / splitk;
  splitk = (nk + per - 1) / per;
  const int Nw = (N + 3) & ~3;
  if ((int64_t)splitk * M * Nw > splitk_ws_floats) return FBL_ERR_ARG;
  GemmTnArgs g{(const bf16*)A, (const bf16*)B, lda, ldb, M, N, K, splitk_ws, Nw, splitk, (M + 127) / 128, (N + 127) / 128};
  hipLaunchKernelGGL(gemm_bf16_tn_kernel, dim3(g.tiles_m * g.tiles_n, splitk), dim3(256), 0, (hipStream_t)stream, g);
  FBL_CHECK_LAUNCH();
  const long nq = (long)M * (Nw >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((nq + 255) / 256), 1), dim3(256), 0, (hipStream_t)stream,
                     (const float*)splitk_ws, splitk, M, N, Nw, out_f32, (long)ldc, 0L);
  FBL_CHECK_LAUNCH();
  return 0;
}
