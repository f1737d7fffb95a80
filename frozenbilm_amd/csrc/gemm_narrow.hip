// Tall-and-narrow bf16 MFMA GEMM for gfx950:  C[M, N] = epilogue( alpha * A[M,K] . B[N,K]^T ),  N <= 192, K <= 1536, M large.
//
// The adapter backward's dz = (dy . Wu) (*) gate  (autograd of model/adapter.py:38-42: [8512 x 1536] . [1536 x 192]) is 5 GFLOP
// against 32 MB of operands -- HBM-bound at ~8 us -- but on the generic tiles it is a latency chain: 64x128 tiles give 266
// workgroups (one per CU) that EACH stream the whole 0.6 MB weight through a 3-stage LDS ring, 24 dependent K-steps behind
// one another (36 us in the step, 18 us alone; N = 192 also wastes a quarter of the second 128-wide tile column).
//
// Here the roles are turned around: the narrow operand stays put, the tall one streams.
//   * a workgroup (4 waves, one per SIMD) owns a 48-column slice of B -- [48][K] bf16, 147 KiB at K = 1536 -- copied into
//     LDS ONCE (row stride K*2 + 16 bytes: the 16 lanes of a fragment read hit 16 distinct bank groups);
//   * it then walks 64-row tiles of A (tile t = row group + i * row groups), each wave 16 rows: the MFMA fragments of A come
//     straight from global memory into registers (a lane's fragment is 16 contiguous bytes of one row), a chunk of 24
//     K-steps (96 registers) ahead of the chunk being multiplied -- no LDS staging, no barrier after the prologue;
//   * per K-step three v_mfma_f32_16x16x32_bf16 (weights as "A" operand: a lane owns 4 consecutive output columns) against
//     three ds_read_b128 of the resident slice.
// grid = ceil(N / 48) column slices x (256 / slices) row groups: one workgroup per CU, every CU busy, every byte of A read once
// per column slice (4 x 26 MB through L2, once from HBM), B read once per workgroup from L2.
//
// Epilogue (fused like gemm_common.h's, for the cases this shape occurs in): + bias, optional gate by a bf16 aux operand
// (FBL_AUX_MUL_POS_BF16: v if aux > 0 else 0 -- ReLU gate and dropout mask of the bottleneck in one test), bf16 output.
#include "gemm_common.h"

namespace fblgemm {
namespace {

constexpr int NARROW_NC = 48;       // columns per workgroup
constexpr int NARROW_CHUNK = 24;    // K-steps (of 32) per register chunk of A fragments
constexpr int NARROW_MAX_K = 1536;

template <int AUX>
__global__ __launch_bounds__(256) void gemm_narrow_kernel(GemmArgs g, int row_groups) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, q = lane >> 4;
  // block b runs on XCD b % 8 (observed dispatch order, used for L2 affinity only): the column slices of one row group are
  // placed on ONE XCD, back to back, so a row tile of A is fetched into one L2 once instead of into up to four
  int cg, rg;
  {
    const int nb = gridDim.x, b = blockIdx.x;
    if ((nb % NXCD) == 0 && ((nb / NXCD) % g.tiles_n) == 0) {
      const int idx = b / NXCD, xcd = b % NXCD;
      rg = xcd * ((nb / NXCD) / g.tiles_n) + idx / g.tiles_n;
      cg = idx % g.tiles_n;
    } else {
      cg = b % g.tiles_n;
      rg = b / g.tiles_n;
    }
  }
  const int n_base = cg * NARROW_NC;
  const int K = g.K;
  const int ldb_s = K * 2 + 16;  // bytes per LDS row

  // ---- B slice -> LDS (rows beyond N: clamped, their results are never stored)
  {
    const int cpr = K / 8;  // 16-byte chunks per row
    const int total = NARROW_NC * cpr;
    for (int i0 = tid; i0 < total; i0 += 256 * 4) {
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(i0 + u * 256, total - 1);
        const int r = i / cpr, ch = i - r * cpr;
        v[u] = *(const bf16x8*)(g.B + (long)min(n_base + r, g.N - 1) * g.ldb + ch * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 256;
        if (i < total) {
          const int r = i / cpr, ch = i - r * cpr;
          *(bf16x8*)(smem + r * ldb_s + ch * 16) = v[u];
        }
      }
    }
  }
  __syncthreads();

  const int nsteps = K / 32;                                         // K-steps of one MFMA
  const int nchunks = (nsteps + NARROW_CHUNK - 1) / NARROW_CHUNK;    // register chunks per row tile
  const int ntiles_m = (g.M + 63) / 64;
  const int my_tiles = rg < ntiles_m ? (ntiles_m - rg + row_groups - 1) / row_groups : 0;
  const int total_chunks = my_tiles * nchunks;
  if (total_chunks == 0) return;
  const char* bfrag = smem + c * ldb_s + q * 16;  // + t*16*ldb_s + ks*64
  float bv[3][4];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n_base + t * 16 + q * 4 + r;
      bv[t][r] = (g.bias && n < g.N) ? g.bias[n] : 0.f;
    }

  auto a_ptr = [&](int tile_i, int chunk) -> const bf16* {
    const int m = min((rg + tile_i * row_groups) * 64 + wave * 16 + c, g.M - 1);
    return g.A + (long)m * g.lda + chunk * (NARROW_CHUNK * 32) + q * 8;
  };
  auto load_chunk = [&](int flat, bf16x8* dst) {
    const int tile_i = flat / nchunks, chunk = flat - tile_i * nchunks;
    const bf16* p = a_ptr(tile_i, chunk);
    const int steps = min(NARROW_CHUNK, nsteps - chunk * NARROW_CHUNK);
#pragma unroll
    for (int s = 0; s < NARROW_CHUNK; ++s)
      if (s < steps) dst[s] = *(const bf16x8*)(p + s * 32);
  };

  f32x4 acc[3];
  bf16x8 fa[NARROW_CHUNK], fb[NARROW_CHUNK];
  auto mma_chunk = [&](int flat, const bf16x8* src) {
    const int tile_i = flat / nchunks, chunk = flat - tile_i * nchunks;
    if (chunk == 0) {
#pragma unroll
      for (int t = 0; t < 3; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int steps = min(NARROW_CHUNK, nsteps - chunk * NARROW_CHUNK);
    const char* bp = bfrag + chunk * (NARROW_CHUNK * 64);
#pragma unroll
    for (int s = 0; s < NARROW_CHUNK; ++s) {
      if (s < steps) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(bp + t * 16 * ldb_s + s * 64), src[s], acc[t], 0, 0, 0);
      }
    }
    if (chunk == nchunks - 1) {  // this row tile is complete: epilogue for the wave's 16 rows x 48 columns
      const int m = (rg + tile_i * row_groups) * 64 + wave * 16 + c;
      if (m < g.M) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int n4 = n_base + t * 16 + q * 4;
          if (n4 + 3 < g.N) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[t][r] * g.alpha + bv[t][r];
            if (AUX == FBL_AUX_MUL_POS_BF16) {
              const bf16x4 x = *(const bf16x4*)((const bf16*)g.aux + (long)m * g.ld_aux + n4);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = bf2f(x[r]) > 0.f ? v[r] : 0.f;
            }
            *(bf16x4*)(g.out_bf16 + (long)m * g.ldc + n4) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
          }
        }
      }
    }
  };

  // two register chunks in flight: the loads of chunk i+1 are issued before chunk i is multiplied
  load_chunk(0, fa);
  for (int i = 0; i < total_chunks; i += 2) {
    if (i + 1 < total_chunks) load_chunk(i + 1, fb);
    mma_chunk(i, fa);
    if (i + 2 < total_chunks) load_chunk(i + 2, fa);
    if (i + 1 < total_chunks) mma_chunk(i + 1, fb);
  }
}

}  // namespace

bool gemm_narrow_eligible(const GemmArgs& g, int act, int aux_kind) {
  if (act != FBL_ACT_NONE || (aux_kind != FBL_AUX_NONE && aux_kind != FBL_AUX_MUL_POS_BF16)) return false;
  if (g.N > 192 || (g.N % 16) || g.K > NARROW_MAX_K || (g.K % 32) || g.M < 2048) return false;
  if (g.splitk != 1 || g.a_kblk || g.kskip_len || g.rowscale || g.seg_n > 0 || g.drop_thresh || g.r_t) return false;
  if (!g.out_bf16 || g.out_f32 || g.out_pre || (g.ldc % 4)) return false;
  if (aux_kind == FBL_AUX_MUL_POS_BF16 && (!g.aux || (g.ld_aux % 4))) return false;
  return true;
}

int launch_gemm_narrow(const GemmArgs& g0, int aux_kind, int n_cu, hipStream_t stream) {
  GemmArgs g = g0;
  g.tiles_n = (g.N + NARROW_NC - 1) / NARROW_NC;
  int row_groups = n_cu / g.tiles_n;
  const int ntiles_m = (g.M + 63) / 64;
  if (row_groups > ntiles_m) row_groups = ntiles_m;
  if (row_groups < 1) row_groups = 1;
  const int smem_bytes = NARROW_NC * (g.K * 2 + 16);
  const dim3 grid((unsigned)(g.tiles_n * row_groups));
#define FBL_NARROW_LAUNCH(AUX_)                                                                                   \
  do {                                                                                                            \
    static int attr_bytes = 0;                                                                                    \
    auto kfn = gemm_narrow_kernel<AUX_>;                                                                          \
    if (smem_bytes > attr_bytes) {                                                                                \
      hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes); \
      if (e != hipSuccess) return (int)e;                                                                         \
      attr_bytes = smem_bytes;                                                                                    \
    }                                                                                                             \
    hipLaunchKernelGGL(kfn, grid, dim3(256), smem_bytes, stream, g, row_groups);                                  \
  } while (0)
  if (aux_kind == FBL_AUX_MUL_POS_BF16) FBL_NARROW_LAUNCH(FBL_AUX_MUL_POS_BF16);
  else FBL_NARROW_LAUNCH(FBL_AUX_NONE);
#undef FBL_NARROW_LAUNCH
  FBL_CHECK_LAUNCH();
  return 0;
}

}  // namespace fblgemm
