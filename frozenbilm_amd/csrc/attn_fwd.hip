// Fused disentangled self-attention forward for gfx950 (DeBERTa-v2: c2p + p2c relative-position bias, share_att_key).
//
//   score[i,j] = scale * ( Q_i.K_j + Q_i.PK[idx(i-j)] + K_j.PQ[idx(i-j)] ),   ctx = dropout(softmax_masked(score)) . V
//   reference: model/deberta.py:717-818 + :820-947 + XSoftmax :100-138   
//
// One workgroup (4 waves) = one (batch, head, 64-query tile); it sweeps the keys in tiles of 64 with an online
// softmax.  For a (query tile, key tile) pair idx(i-j) spans <= 127 consecutive rows of the position tables (idx is
// monotone with slope <= 1), so both bias terms are MFMA GEMMs against a 128-row WINDOW of PK / PQ staged in LDS,
//     T1[i][w] = Q_i . PKwin[w]   (per wave: its 16 queries need an 80-row sub-window)
//     T2[j][w] = K_j . PQwin[w]   (per 16-key tile: 80-row sub-window; wave w computes key tile w, all waves read it)
// followed by an LDS gather  c2p[i,j] = T1[i][idx(i-j)-..],  p2c[i,j] = T2[j][idx(i-j)-..].  Nothing of size SxS or
// Sx512 ever reaches HBM.  All MFMAs are "swapped" (keys/positions as A rows, queries as B columns): a lane owns ONE
// query column, so softmax statistics are in-lane + 2 shuffles, and P feeds the P.V MFMA straight from registers
// (the k-slot order of that MFMA is permuted identically on the V^T operand).  The next key tile's global loads are
// issued into registers before the current tile is computed (HBM/L2 latency hides under the MFMA + LDS work).
#include <stdlib.h>

#include "attn_common.h"
#include "../../include/fbl.h"

namespace {
using namespace attn;

struct AttnArgs {
  const bf16* q; const bf16* k; const bf16* v; const bf16* pk; const bf16* pq;
  long ldq, ldk, ldp;
  long ldv;
  const int16_t* relidx;
  const int32_t* mask;
  const int32_t* klen;  // [B] last valid position + 1 (tiles beyond it are exactly zero and skipped) or null
  float scale, p_drop;
  uint64_t seed;
  bf16* ctx;
  long ldo;
  float* lse;
  int B, S, Sp, nh, span2;
  int dbg;  // ablation switches (FBL_ATTN_DBG; 0 in production)
};

constexpr int SM_KS = 0;                        // [64][64] bf16 swizzled
constexpr int SM_VT = SM_KS + 8192;             // V tile, row-major [64 keys][72] bf16 (read transposed by ds_read_b64_tr_b16)
constexpr int SM_PK = SM_VT + 64 * LDV * 2;     // [128][64] bf16 swizzled
constexpr int SM_PQ = SM_PK + 16384;
constexpr int SM_T1 = SM_PQ + 16384;            // [4 waves][16][LT] fp16
constexpr int SM_T2 = SM_T1 + 4 * 16 * LT * 2;  // [64 keys][LT] fp16
constexpr int SM_IDX = SM_T2 + 64 * LT * 2;     // int16 [1024]
constexpr int SM_KM = SM_IDX + 2048;            // float [64] key validity
constexpr int SM_TOTAL = SM_KM + 256;

struct TileRegs {  // one key tile in flight: 12 x 16 B per thread
  bf16x8 k[2], v[2], pk[4], pq[4];
  float km;
};

__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int i0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int S = a.S, Sp = a.Sp;
  const int i = i0 + w * 16 + c;  // this lane's query row
  const int ic = min(i, S - 1);

  int16_t* idx = (int16_t*)(smem + SM_IDX);
  float* kms = (float*)(smem + SM_KM);
  f16* T1w = (f16*)(smem + SM_T1) + w * 16 * LT;
  f16* T2 = (f16*)(smem + SM_T2);

  load_idx_padded(idx, a.relidx, S, Sp, tid, 256);

  bf16x8 qf[2];
  {
    const bf16* qp = a.q + ((long)b * S + ic) * a.ldq + h * 64 + g * 8;
    qf[0] = *(const bf16x8*)qp;
    qf[1] = *(const bf16x8*)(qp + 32);
  }
  const bool qvalid = i < S && a.mask[(long)b * S + ic] != 0;

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const DropKey dk = attn_drop_key(a.seed, b * a.nh + h, a.p_drop);
  const float k2 = a.scale * LOG2E;  // scores stay unscaled; softmax runs in the exp2 domain
  const int kl = a.klen ? min(a.klen[b], S) : S;
  const int nkt = (i0 < kl) ? (kl + 63) / 64 : 0;  // masked key tiles contribute exactly 0; a fully masked query tile outputs 0
  const int tq = Sp - 1;  // idx[i - j + tq]
  __syncthreads();  // idx table visible

  const int srow = tid >> 3, sch = tid & 7;  // staging role of this thread: row (0..31) and 16-byte chunk
  auto load_tile = [&](int jt, TileRegs& R) {
    const int j0 = jt * 64;
    const int r_lo = idx[i0 - (j0 + 63) + tq];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = srow + t * 32;
      const int j = min(j0 + row, S - 1);
      R.k[t] = *(const bf16x8*)(a.k + ((long)b * S + j) * a.ldk + h * 64 + sch * 8);
      R.v[t] = *(const bf16x8*)(a.v + ((long)b * S + j) * a.ldv + h * 64 + sch * 8);
    }
    if (!(a.dbg & 16))
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int r = min(r_lo + srow + t * 32, a.span2 - 1);
      const long off = (long)r * a.ldp + h * 64 + sch * 8;
      R.pk[t] = *(const bf16x8*)(a.pk + off);
      R.pq[t] = *(const bf16x8*)(a.pq + off);
    }
    R.km = 0.f;
    if (tid < 64) {
      const int j = j0 + tid;
      R.km = (j < S && a.mask[(long)b * S + min(j, S - 1)] != 0) ? 1.f : 0.f;
    }
  };
  auto store_tile = [&](const TileRegs& R) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = srow + t * 32;
      lds_put(smem + SM_KS, row, sch, R.k[t]);
      *(bf16x8*)(smem + SM_VT + row * (LDV * 2) + sch * 16) = R.v[t];
    }
    if (!(a.dbg & 16))
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = srow + t * 32;
      lds_put(smem + SM_PK, row, sch, R.pk[t]);
      lds_put(smem + SM_PQ, row, sch, R.pq[t]);
    }
    if (tid < 64) kms[tid] = R.km;
  };

  TileRegs R;
  if (nkt > 0) load_tile(0, R);
  for (int jt = 0; jt < nkt; ++jt) {
    const int j0 = jt * 64;
    const int r_lo = idx[i0 - (j0 + 63) + tq];
    store_tile(R);
    __syncthreads();
    if (jt + 1 < nkt) load_tile(jt + 1, R);  // in flight while this tile is computed

    // sub-window offsets: this wave's 16 queries (T1) and each 16-key tile (T2)
    const int base1 = idx[i0 + w * 16 - (j0 + 63) + tq];  // absolute table row of T1w[.][0]
    const int off1 = base1 - r_lo;
    int base2[4];                                          // absolute table row of T2[key tile nt][.][0]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) base2[nt] = idx[i0 - (j0 + nt * 16 + 15) + tq];
    const int off2w = idx[i0 - (j0 + w * 16 + 15) + tq] - r_lo;

    // ---- (1) content scores, transposed: sacc[nt][r] = Q_i . K_j,  j = j0 + nt*16 + g*4 + r
    f32x4 sacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + SM_KS, nt * 16 + c, g), qf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + SM_KS, nt * 16 + c, 4 + g), qf[1], acc, 0, 0, 0);
      sacc[nt] = acc;
    }
    // ---- (2) T1 for this wave's queries, (3) T2 for key tile w
    if (!(a.dbg & 2)) {
    bias_tile(smem + SM_PK, off1, qf[0], qf[1], T1w + c * LT, c, g);
    bias_tile(smem + SM_PQ, off2w, lds_frag(smem + SM_KS, w * 16 + c, g), lds_frag(smem + SM_KS, w * 16 + c, 4 + g),
              T2 + (w * 16 + c) * LT, c, g, kms[w * 16 + c] != 0.f);  // masked keys: whole T2 row = -inf
    }
    __syncthreads();

    // ---- (4) gather the bias terms, online softmax.  No clamps: in-range (i, j) always land inside the 80-wide
    // sub-windows; padding queries read finite garbage that stays in their own lane column and is dropped at the end,
    // padding / masked keys carry -inf through their T2 row.
    float p[16];
    float mx = -INFINITY;
    {
      const int16_t* ib = idx + (i - j0 - g * 4 + tq - 63);
      const f16* t1row = T1w + c * LT - base1;
      const f16* t2g = T2 + g * 4 * LT;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f16* t2n = t2g - base2[nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float s = sacc[nt][r];
          if (!(a.dbg & 1)) {
            const int wi = ib[63 - nt * 16 - r];
            s += (float)t1row[wi] + (float)t2n[(nt * 16 + r) * LT + wi];
          }
          p[nt * 4 + r] = s;
          mx = fmaxf(mx, s);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float alpha = 1.f, psum = 0.f;
    if (m_new == -INFINITY) {
#pragma unroll
      for (int e = 0; e < 16; ++e) p[e] = 0.f;
    } else {
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * k2);  // m_run = -inf -> 0
      const float mk = -m_new * k2;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        p[e] = __builtin_amdgcn_exp2f(fmaf(p[e], k2, mk));
        psum += p[e];
      }
    }
    psum += __shfl_xor(psum, 16, 64);
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
    if (a.p_drop > 0.f && !(a.dbg & 8)) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          uint32_t x, y;
          attn_drop_block(dk, i >> 1, (j0 + nt * 16 + g * 4 + bb * 2) >> 1, Sp >> 1, &x, &y);
          p[nt * 4 + bb * 2] *= attn_drop_keep(dk, x, y, i & 1, 0);
          p[nt * 4 + bb * 2 + 1] *= attn_drop_keep(dk, x, y, i & 1, 1);
        }
    }
    // ---- (5) O^T += V^T . P^T ; k-slot e of step kk  <->  key kk*32 + (e>>2)*16 + g*4 + (e&3)
    if (!(a.dbg & 4))
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[e] = f2bf(p[(2 * kk) * 4 + e]);
        pf[4 + e] = f2bf(p[(2 * kk + 1) * 4 + e]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // V^T fragment (row d = dt*16 + c, keys kk*32 + g*4 + {0..3} and +16) straight from the row-major V tile:
        // the 16 lanes of a group present a [4 keys][16 d] block and the transpose read hands each lane its column
        const bf16* vblk = (const bf16*)(smem + SM_VT) + (kk * 32 + g * 4 + (c >> 2)) * LDV + dt * 16 + (c & 3) * 4;
        union { tr16x4 h[2]; bf16x8 v; } u;
        u.h[0] = lds_tr16(vblk);
        u.h[1] = lds_tr16(vblk + 16 * LDV);
        const bf16x8 vf = u.v;
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
      }
    }
    __syncthreads();  // LDS tiles are overwritten by the next key tile
  }

  const float inv_l = (qvalid && l_run > 0.f) ? 1.f / l_run : 0.f;
  if (i < S) {
    bf16* op = a.ctx + ((long)b * S + i) * a.ldo + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      f32x4 v = o[dt] * inv_l;
      if (inv_l == 0.f) v = (f32x4){0.f, 0.f, 0.f, 0.f};  // masked query rows are exactly zero (their lane may hold garbage)
      *(bf16x4*)(op + dt * 16) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    }
    if (g == 0 && a.lse)
      a.lse[((long)b * a.nh + h) * S + i] = (qvalid && l_run > 0.f) ? m_run * a.scale + __logf(l_run) : INFINITY;
  }
}

}  // namespace

extern "C" int fbl_disent_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                   const void* pk, const void* pq, int64_t ldp,
                                   const int16_t* relidx, const int32_t* mask, const int32_t* klen, float scale,
                                   float p_drop, uint64_t seed,
                                   void* ctx, int64_t ldo, float* lse, int B, int S, int Sp, int nh, int span2,
                                   void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64) return FBL_ERR_SHAPE;
  if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldp % 8) || (ldo % 4)) return FBL_ERR_ALIGN;
  if (B <= 0 || nh <= 0) return 0;
  AttnArgs a{(const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)pk, (const bf16*)pq, ldq, ldk, ldp, ldv, relidx, mask, klen, scale, p_drop, seed, (bf16*)ctx, ldo, lse, B, S, Sp, nh, span2, 0};
  static const int dbg = getenv("FBL_ATTN_DBG") ? atoi(getenv("FBL_ATTN_DBG")) : 0;
  a.dbg = dbg;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SM_TOTAL);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid((S + 63) / 64, nh, B);
  hipLaunchKernelGGL(attn_fwd_kernel, grid, dim3(256), SM_TOTAL, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}
