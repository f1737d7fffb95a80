// Fused disentangled self-attention forward for gfx950 (DeBERTa-v2: c2p + p2c relative-position bias, share_att_key).
//
//   score[i,j] = scale * ( Q_i.K_j + Q_i.PK[idx(i-j)] + K_j.PQ[idx(i-j)] ),   ctx = dropout(softmax_masked(score)) . V
//   reference: model/deberta.py:717-818 + :820-947 + XSoftmax :100-138   
//
// One workgroup (4 waves) = one (batch, head, 64-query tile); it sweeps the keys in tiles of 64 with an online
// softmax.  For a (query tile, key tile) pair idx(i-j) spans <= 127 consecutive rows of the position tables (idx is
// monotone with slope <= 1), so both bias terms are MFMA GEMMs against a 128-row WINDOW of PK / PQ,
//     T1[i][w] = Q_i . PKwin[w]   (a 16-query group needs an 80-row sub-window of it; 96 outside the identity band)
//     T2[j][w] = K_j . PQwin[w]   (a 16-key group likewise)
// followed by an LDS gather  c2p[i,j] = T1[i][idx(i-j)-..],  p2c[i,j] = T2[j][idx(i-j)-..].  Nothing of size SxS or
// Sx512 ever reaches HBM.  All MFMAs are "swapped" (keys/positions as A rows, queries as B columns): a lane owns ONE
// query column, so softmax statistics are in-lane + 2 shuffles, and P feeds the P.V MFMA straight from registers
// (the k-slot order of that MFMA is permuted identically on the V^T operand).  The window is never staged in LDS: see
// the note at the kernel (row-tile split of the bias GEMMs); inside the identity band of the relative-position map the
// gather addresses are lane constants + immediates (no index table).  Workgroups are dispatched longest sample first,
// the tiles of one (sample, head) on one XCD (attn_common.h: wg_coord).
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
  const int32_t* border;  // [B] order in which the samples are dispatched (longest first) or null
  float scale, p_drop;
  uint64_t seed; const uint64_t* seed_dev;
  bf16* ctx;
  long ldo;
  float* lse;
  int B, S, Sp, nh, span2;
  int lin;  // |d| < lin: idx(d) = idx(0) + d (identity buckets); 0 = unknown
  int dbg;  // ablation switches (FBL_ATTN_DBG of debug builds; compiled out of the product library)
  const int32_t* row0;  // [B+1] packed-row layout (PACKED kernels): sample b owns activation rows [row0[b], row0[b+1]) =
                        // its positions 0 .. row0[b+1]-row0[b]-1; null = the padded [B, S] grid (row b*S + s)
  // SAVEP kernels (training): what the backward would otherwise recompute -- psave[b,h,i,j] = exp2(k2*(s_ij - m)) (bf16,
  // [B,nh,Sp,Sp], BEFORE dropout, sign bit set where dropout dropped the pair) with m the running row maximum at key tile j/64, and msave[b,h,j/64,i] = k2*m (fp32,
  // [B,nh,Sp/64,S]): P_ij = psave * exp2(msave - lse*log2(e)).  Only the tile pairs the forward visits are written.
  bf16* psave;
  float* msave;
};
#ifdef FBL_DEBUG_SWITCHES
#define ATTN_DBG(bit) (a.dbg & (bit))
#else
#define ATTN_DBG(bit) 0
#endif

// LDS per workgroup: Q, K, V tiles 3 x 8 KiB + T1 13 KiB + T2 13 KiB + key mask + index table = 52 KiB at Sp = 320:
// three workgroups per CU.  The position tables are NOT staged in LDS.  The 128-row window [r_lo, r_lo + 128) of a tile
// pair is cut into 8 row tiles of 16; the 80-row (96 outside the identity band) sub-window of a 16-query / 16-key group
// is a run of 5 (6) of them.  Each WAVE owns two row tiles: it loads their MFMA A-fragments straight from global memory
// (2 x 2 x 16 bytes per lane and table; the tables of all heads are 2.4 MB, L2-resident) and multiplies them against
// EVERY query group (T1, B = the groups' Q fragments out of the Q tile) and every key group (T2, B = K fragments out of
// the K tile) whose sub-window contains the tile -- 5 + 5 MFMA pairs per wave and table as before, but 8 instead
// of 20 fragment loads per wave, every table byte fetched once per workgroup and no LDS window (32 KiB less).
constexpr int LTW = 100;                        // fp16 row stride of the T1/T2 tiles (6 row tiles = 96 used): 50 dwords, 16 rows -> 16 banks
constexpr int SM_KS = 0;                        // [64 keys][64] bf16 swizzled
constexpr int SM_VS = SM_KS + 8192;             // [64 keys][64] bf16 swizzled (read transposed by ds_read_b64_tr_b16)
constexpr int SM_T1 = SM_VS + 8192;             // [64 queries][LTW] fp16
constexpr int SM_T2 = SM_T1 + 64 * LTW * 2;     // [64 keys][LTW] fp16
constexpr int SM_KM = SM_T2 + 64 * LTW * 2;     // float [64] key validity
constexpr int SM_QS = SM_KM + 256;              // [64 queries][64] bf16 swizzled (staged once)
constexpr int SM_IDX = SM_QS + 8192;            // int16 [2*Sp]
__host__ __device__ constexpr int sm_total(int Sp) { return SM_IDX + 2 * Sp * 2; }

struct TileRegs {  // one key tile in flight: 4 x 16 B per thread
  bf16x8 k[2], v[2];
  int km;  // mask word of key j0 + lane
};

// PACKED: q / k / v / ctx rows follow AttnArgs::row0 (ragged batches without their padding rows); mask and lse keep the
// padded [B, S] indexing.  A separate instantiation: the padded kernel's code is unchanged.
template <int OCC, bool PF, bool PACKED = false, bool SAVEP = false>
__global__ __launch_bounds__(256, OCC) void attn_fwd_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, Sp = a.Sp;
  const WgCoord wc = wg_coord((S + 63) / 64, a.nh, a.B, a.border);
  const int i0 = wc.x * 64, h = wc.h, b = wc.b;
  const int i = i0 + w * 16 + c;  // this lane's query row
  const int ic = min(i, S - 1);

  int16_t* idx = (int16_t*)(smem + SM_IDX);
  float* kms = (float*)(smem + SM_KM);
  f16* T1 = (f16*)(smem + SM_T1);
  f16* T2 = (f16*)(smem + SM_T2);

  const int kl = a.klen ? min(a.klen[b], S) : S;
  const int nkt = (i0 < kl) ? (kl + 63) / 64 : 0;  // masked key tiles contribute exactly 0; a fully masked query tile outputs 0
  // first activation row of this sample and the number of its rows that exist (kl <= lim: the engine keeps every valid row)
  const long rb = PACKED ? (long)a.row0[b] : (long)b * S;
  const int lim = PACKED ? min(a.row0[b + 1] - a.row0[b], S) : S;
  if (nkt > 0) load_idx_padded(idx, a.relidx, S, Sp, tid, 256);

  // (the Q tile lives in LDS: B operands of Q.K^T and of the T1 row tiles are read from it where they are used)
  const bool qvalid = i < S && a.mask[(long)b * S + ic] != 0;

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const DropKey dk = attn_drop_key(a.p_drop > 0.f ? fbl_seed(a.seed, a.seed_dev) : 0, b * a.nh + h, a.p_drop);
  const float k2 = a.scale * LOG2E;  // scores stay unscaled; softmax runs in the exp2 domain
  const int tq = Sp - 1;             // idx[i - j + tq]
  const int izero = a.relidx[S - 1];  // idx(0)
  const bf16* pkh = a.pk + h * 64 + g * 8;
  const bf16* pqh = a.pq + h * 64 + g * 8;
  // the two window row tiles of this wave: {3,0} {4,7} {2,6} {5,1} -- inside the band tile t serves 1,2,3,4,4,3,2,1
  // groups (t = 0..7), so every wave issues 5 MFMA pairs per table
  const int my_t0 = (0x5243 >> (w * 4)) & 15, my_t1 = (0x1670 >> (w * 4)) & 15;

  const int srow = tid >> 3, sch = tid & 7;  // staging role of this thread: row (0..31) and 16-byte chunk
  // LDS addressing as lane constants + immediates (anything recomputed per fragment is hoisted out of the loop by the
  // compiler and then lives in a register each): in the swizzled [rows][8 x 16 B] images the swizzle term of row
  // x*16 + c (or srow + 32 t) only depends on c (srow), and chunk 4+g is chunk g with bit 6 of the address flipped.
  const int fb0 = c * 128 + ((g ^ (c & 7)) << 4), fb1 = fb0 ^ 64;  // fragment of row x*16 + c: + x*2048
  const int sb = srow * 128 + ((sch ^ (srow & 7)) << 4);           // staging slot of row srow + 32 t: + t*4096
  auto load_tile = [&](int jt, TileRegs& R) {
    const int j0 = jt * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = srow + t * 32;
      const int j = min(j0 + row, lim - 1);
      R.k[t] = *(const bf16x8*)(a.k + (rb + j) * a.ldk + h * 64 + sch * 8);
      R.v[t] = *(const bf16x8*)(a.v + (rb + j) * a.ldv + h * 64 + sch * 8);
    }
    // (every wave loads the mask word of key j0 + lane, clamped, and nothing here USES it: a use inside a tid < 64 branch would
    //  park wave 0 -- and with it the workgroup's next barrier -- until the K / V requests in front of it have returned)
    R.km = a.mask[(long)b * S + min(j0 + lane, S - 1)];
  };
  auto store_tile = [&](const TileRegs& R, int j0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      *(bf16x8*)(smem + SM_KS + sb + t * 4096) = R.k[t];
      *(bf16x8*)(smem + SM_VS + sb + t * 4096) = R.v[t];
    }
    if (tid < 64) kms[tid] = (j0 + tid < S && R.km != 0) ? 1.f : 0.f;
  };

  TileRegs R;
  if (PF && nkt > 0) load_tile(0, R);
  if (nkt > 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = srow + t * 32;
      *(bf16x8*)(smem + SM_QS + sb + t * 4096) = *(const bf16x8*)(a.q + (rb + min(i0 + row, lim - 1)) * a.ldq + h * 64 + sch * 8);
    }
    __syncthreads();  // index table and Q tile visible
  }
  for (int jt = 0; jt < nkt; ++jt) {
    const int j0 = jt * 64;
    // Inside the identity band of the relative-position map (|i - j| < lin for every pair of the tile) the map is
    // affine, idx(d) = idx(0) + d: window, sub-window bases and gather offsets are then lane constants + immediates.
    const bool band = !ATTN_DBG(32) && max(abs(i0 - (j0 + 63)), abs(i0 + 63 - j0)) < a.lin;
    // window start, and the first row tile (t0) of every query group's / key group's sub-window
    int r_lo, t0q[4], t0k[4];
    if (band) {
      r_lo = izero + i0 - (j0 + 63);
#pragma unroll
      for (int x = 0; x < 4; ++x) { t0q[x] = x; t0k[x] = 3 - x; }
    } else {
      r_lo = __builtin_amdgcn_readfirstlane((int)idx[i0 - (j0 + 63) + tq]);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        t0q[x] = (__builtin_amdgcn_readfirstlane((int)idx[i0 + x * 16 - (j0 + 63) + tq]) - r_lo) >> 4;
        t0k[x] = (__builtin_amdgcn_readfirstlane((int)idx[i0 - (j0 + x * 16 + 15) + tq]) - r_lo) >> 4;
      }
    }
    const int ntile = band ? 5 : 6;  // row tiles per sub-window (outside the band its start is not 16-aligned: 79 + 15 rows)
    // table A-fragments of this wave's two row tiles (in flight across the tile staging + barrier)
    bf16x8 apk[2][2], apq[2][2];
    if (!ATTN_DBG(2)) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long ro = (long)min(r_lo + (u ? my_t1 : my_t0) * 16 + c, a.span2 - 1) * a.ldp;
        apk[u][0] = *(const bf16x8*)(pkh + ro);
        apk[u][1] = *(const bf16x8*)(pkh + ro + 32);
        apq[u][0] = *(const bf16x8*)(pqh + ro);
        apq[u][1] = *(const bf16x8*)(pqh + ro + 32);
      }
    }
    if (!PF) load_tile(jt, R);  // no register prefetch: the other workgroups of the CU cover the load latency
    store_tile(R, j0);
    __syncthreads();  // K / V tile visible

    // ---- (1) content scores, transposed: sacc[nt][r] = Q_i . K_j,  j = j0 + nt*16 + g*4 + r
    f32x4 sacc[4];
    const bf16x8 qf[2] = {*(const bf16x8*)(smem + SM_QS + w * 2048 + fb0), *(const bf16x8*)(smem + SM_QS + w * 2048 + fb1)};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + SM_KS + nt * 2048 + fb0), qf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + SM_KS + nt * 2048 + fb1), qf[1], acc, 0, 0, 0);
      sacc[nt] = acc;
    }
    // ---- (2) this wave's row tiles of T1[query][sub-window row] = Q_i . PK[.] and T2[key][.] = K_j . PQ[.]  (fp16)
    if (!ATTN_DBG(2)) {
      const f16 ninf = (f16)(-INFINITY);
      f16* t1s = T1 + c * LTW + g * 4;  // + (x*16)*LTW + rel*16
      f16* t2s = T2 + c * LTW + g * 4;
      auto pair1 = [&](const bf16x8& a0, const bf16x8& a1, int x, int rel) {  // row tile x query group x
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, *(const bf16x8*)(smem + SM_QS + x * 2048 + fb0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, *(const bf16x8*)(smem + SM_QS + x * 2048 + fb1), acc, 0, 0, 0);
        *(f16x4*)(t1s + x * 16 * LTW + rel * 16) = to_f16x4(acc);
      };
      auto pair2 = [&](const bf16x8& a0, const bf16x8& a1, int x, int rel) {  // row tile x key group x
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, *(const bf16x8*)(smem + SM_KS + x * 2048 + fb0), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, *(const bf16x8*)(smem + SM_KS + x * 2048 + fb1), acc, 0, 0, 0);
        const f16x4 tv = to_f16x4(acc);
        // masked / padding keys: the whole T2 row is -inf, their score needs no per-element test
        *(f16x4*)(t2s + x * 16 * LTW + rel * 16) = kms[x * 16 + c] != 0.f ? tv : (f16x4){ninf, ninf, ninf, ninf};
      };
      {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int t = u ? my_t1 : my_t0;
#pragma unroll
          for (int x = 0; x < 4; ++x) {
            const int rel1 = t - t0q[x];
            if (rel1 >= 0 && rel1 < ntile) pair1(apk[u][0], apk[u][1], x, rel1);
            const int rel2 = t - t0k[x];
            if (rel2 >= 0 && rel2 < ntile) pair2(apq[u][0], apq[u][1], x, rel2);
          }
        }
      }
    }
    __syncthreads();
    if (PF) load_tile(min(jt + 1, nkt - 1), R);  // next tile in flight during the gather / softmax / P.V phase (unconditional: behind a
                                                 // branch the compiler cannot count the requests and later waits cover them)

    // ---- (3) gather the bias terms, online softmax.  No clamps: in-range (i, j) always land inside the sub-windows;
    // padding queries read finite-or-not garbage that stays in their own lane column and is dropped at the end,
    // padding / masked keys carry -inf through their T2 row.
    float p[16];
    float mx = -INFINITY;
    if (ATTN_DBG(1)) {
#pragma unroll
      for (int e = 0; e < 16; ++e) { p[e] = sacc[e >> 2][e & 3]; mx = fmaxf(mx, p[e]); }
    } else if (band) {
      // c2p: T1[i][(i - j) - (r_lo + 16 w - izero)] = T1[i][c + 63 - (nt*16 + g*4 + r)]
      // p2c: T2[j][(i - j) - (r_lo + 16 (3 - nt) - izero)] = T2[nt*16 + g*4 + r][w*16 + c + 15 - (g*4 + r)]
      const f16* t1p = T1 + (w * 16 + c) * LTW + c + 63 - g * 4;
      const f16* t2p = T2 + g * 4 * LTW + w * 16 + c + 15 - g * 4;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s = sacc[nt][r] + (float)t1p[-(nt * 16 + r)] + (float)t2p[(nt * 16 + r) * LTW - r];
          p[nt * 4 + r] = s;
          mx = fmaxf(mx, s);
        }
    } else {
      const int16_t* ib = idx + (i - j0 - g * 4 + tq - 63);
      int sbw = r_lo;  // first table row of this wave's T1 rows
#pragma unroll
      for (int x = 0; x < 4; ++x) sbw = (w == x) ? r_lo + 16 * t0q[x] : sbw;
      const f16* t1row = T1 + (w * 16 + c) * LTW - sbw;
      const f16* t2g = T2 + g * 4 * LTW - r_lo;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const f16* t2n = t2g - 16 * t0k[nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wi = ib[63 - nt * 16 - r];
          const float s = sacc[nt][r] + (float)t1row[wi] + (float)t2n[(nt * 16 + r) * LTW + wi];
          p[nt * 4 + r] = s;
          mx = fmaxf(mx, s);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float alpha = 1.f, psum = 0.f;
    if (!(m_new > -INFINITY)) {  // nothing seen yet (or a NaN column of a padding query: dropped at the end)
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
    float kf[16];  // dropout keep factors (0 or 1/(1-p))
#pragma unroll
    for (int e = 0; e < 16; ++e) kf[e] = 1.f;
    if (a.p_drop > 0.f && !ATTN_DBG(8)) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          uint32_t x, y;
          attn_drop_block(dk, i >> 1, (j0 + nt * 16 + g * 4 + bb * 2) >> 1, Sp >> 1, &x, &y);
          kf[nt * 4 + bb * 2] = attn_drop_keep(dk, x, y, i & 1, 0);
          kf[nt * 4 + bb * 2 + 1] = attn_drop_keep(dk, x, y, i & 1, 1);
        }
    }
    if constexpr (SAVEP) {
      // the un-normalised probabilities of this pair leave as they are, BEFORE dropout, with the dropout decision in the sign
      // bit (P >= 0: a set sign = dropped) -- the backward then neither recomputes the scores nor regenerates the mask (4
      // consecutive keys = 8 bytes per lane and key group; padding query rows i < Sp exist in the buffer and may receive
      // garbage: their lse is +inf, the backward reads P = 0)
      bf16* pp = a.psave + (((long)b * a.nh + h) * Sp + i) * Sp + j0 + g * 4;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bf16x4 v4;
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = f2bf(kf[nt * 4 + r] == 0.f ? -p[nt * 4 + r] : p[nt * 4 + r]);
        *(bf16x4*)(pp + nt * 16) = v4;
      }
      if (g == 0 && i < S) a.msave[(((long)b * a.nh + h) * (Sp >> 6) + jt) * S + i] = m_new * k2;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) p[e] *= kf[e];
    // ---- (4) O^T += V^T . P^T ; k-slot e of step kk  <->  key kk*32 + (e>>2)*16 + g*4 + (e&3)
    if (!ATTN_DBG(4))
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
        // V^T fragment (row d = dt*16 + c, keys kk*32 + g*4 + {0..3} and +16) straight from the swizzled row-major V
        // tile: the 16 lanes of a group present a [4 keys][16 d] block (lane: key c>>2, 4 contiguous d at (c&3)*4) and
        // the transpose read hands each lane its column
        // (row r = kk*32 + g*4 + (c>>2) and r + 16: the swizzle term only depends on the lane and on dt)
        const int r = g * 4 + (c >> 2);
        const int ch = dt * 2 + ((c >> 1) & 1), sub = (c & 1) * 8;
        const char* vb = smem + SM_VS + r * 128 + ((ch ^ (r & 7)) << 4) + sub;
        union { tr16x4 h[2]; bf16x8 v; } u;
        u.h[0] = lds_tr16((const bf16*)(vb + kk * 4096));
        u.h[1] = lds_tr16((const bf16*)(vb + kk * 4096 + 2048));
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf, o[dt], 0, 0, 0);
      }
    }
    __syncthreads();  // LDS tiles are overwritten by the next key tile
  }

  const float inv_l = (qvalid && l_run > 0.f) ? 1.f / l_run : 0.f;
  if (i < S) {
    if (i < lim) {
      bf16* op = a.ctx + (rb + i) * a.ldo + h * 64 + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        f32x4 v = o[dt] * inv_l;
        if (inv_l == 0.f) v = (f32x4){0.f, 0.f, 0.f, 0.f};  // masked query rows are exactly zero (their lane may hold garbage)
        *(bf16x4*)(op + dt * 16) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
      }
    }
    if (g == 0 && a.lse)
      a.lse[((long)b * a.nh + h) * S + i] = (qvalid && l_run > 0.f) ? m_run * a.scale + __logf(l_run) : INFINITY;
  }
}

// ---- attention probabilities on request (output_attentions=True: model/deberta.py:789-818 return_att, :544-560).
// The fused kernel never materialises P; this plain kernel does, for callers that ask: one workgroup per (sample, head,
// query row), thread j-strided over the keys, P[i,j] = exp(scale*(Q_i.K_j + Q_i.PK[idx] + K_j.PQ[idx]) - lse_i) with the
// log-sum-exp the fused forward stored -- 0 for masked pairs and for masked query rows (XSoftmax).  Not on the hot path.
__global__ __launch_bounds__(128) void attn_probs_kernel(const bf16* q, const bf16* k, long ldq, const bf16* pk, const bf16* pq,
                                                         long ldp, const int16_t* relidx, const int32_t* mask, const float* lse,
                                                         float scale, float* probs, int B, int S, int nh) {
  __shared__ float qs[64];
  const int i = blockIdx.x % S, h = (blockIdx.x / S) % nh, b = blockIdx.x / (S * nh);
  const long row = (long)b * S + i;
  if (threadIdx.x < 64) qs[threadIdx.x] = bf2f(q[row * ldq + h * 64 + threadIdx.x]);
  __syncthreads();
  const bool qvalid = mask[row] != 0;
  const float l = lse[((long)b * nh + h) * S + i];
  float* out = probs + (((long)b * nh + h) * S + i) * S;
  for (int j = threadIdx.x; j < S; j += blockDim.x) {
    float p = 0.f;
    if (qvalid && mask[(long)b * S + j] != 0) {
      const int r = relidx[i - j + S - 1];
      const bf16* kj = k + ((long)b * S + j) * ldq + h * 64;
      const bf16* pkr = pk + (long)r * ldp + h * 64;
      const bf16* pqr = pq + (long)r * ldp + h * 64;
      float s = 0.f;
#pragma unroll 8
      for (int d = 0; d < 64; ++d) {
        const float kd = bf2f(kj[d]);
        s += qs[d] * (kd + bf2f(pkr[d])) + kd * bf2f(pqr[d]);
      }
      p = __expf(s * scale - l);
    }
    out[j] = p;
  }
}

}  // namespace

extern "C" int fbl_disent_attn_probs(const void* q, const void* k, int64_t ldq, const void* pk, const void* pq, int64_t ldp,
                                     const int16_t* relidx, const int32_t* mask, const float* lse, float scale,
                                     float* probs, int B, int S, int nh, void* stream) {
  if (S < 1 || S > 512) return FBL_ERR_SHAPE;
  if (!q || !k || !pk || !pq || !relidx || !mask || !lse || !probs) return FBL_ERR_ARG;
  if (B <= 0 || nh <= 0) return 0;
  hipLaunchKernelGGL(attn_probs_kernel, dim3((unsigned)(B * nh * S)), dim3(128), 0, (hipStream_t)stream, (const bf16*)q,
                     (const bf16*)k, (long)ldq, (const bf16*)pk, (const bf16*)pq, (long)ldp, relidx, mask, lse, scale, probs, B, S,
                     nh);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_disent_attn_fwd(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                   const void* pk, const void* pq, int64_t ldp,
                                   const int16_t* relidx, const int32_t* mask, const int32_t* klen,
                                   const int32_t* border, float scale, float p_drop, uint64_t seed,
                                   const uint64_t* seed_dev, void* ctx, int64_t ldo, float* lse, int B, int S, int Sp, int nh, int span2,
                                   int lin_span, const int32_t* row0, void* psave, float* msave, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64) return FBL_ERR_SHAPE;
  if ((ldq % 8) || (ldk % 8) || (ldv % 8) || (ldp % 8) || (ldo % 4)) return FBL_ERR_ALIGN;
  if (lin_span < 0 || 2 * lin_span > span2) return FBL_ERR_ARG;  // idx(0) +- (lin_span - 1 + 79) must stay inside the table
  if (row0 && !klen) return FBL_ERR_ARG;  // the packed layout is defined by the samples' lengths
  if ((psave != nullptr) != (msave != nullptr) || (psave && !lse)) return FBL_ERR_ARG;
  if (B <= 0 || nh <= 0) return 0;
  AttnArgs a{(const bf16*)q, (const bf16*)k, (const bf16*)v, (const bf16*)pk, (const bf16*)pq, ldq, ldk, ldp, ldv, relidx, mask, klen, border, scale, p_drop, seed, seed_dev, (bf16*)ctx, ldo, lse, B, S, Sp, nh, span2, lin_span, 0, row0,
             (bf16*)psave, msave};
  static const int dbg = FBL_ENV_INT("FBL_ATTN_DBG", 0);
  a.dbg = dbg;
  attn_debug_init();
  const int smem_bytes = sm_total(Sp);
  // three workgroups per CU (<= 168 VGPRs), no register prefetch of the next key tile: the other two workgroups of the CU
  // cover the load (measured 88.6 us vs 97.2 with the prefetch, whose registers spill)
  static const int pf = FBL_ENV_INT("FBL_ATTN_PF", 1);  // (measurement builds) register prefetch of the next key tile
  static int attr_bytes = 0;
  if (smem_bytes > attr_bytes) {
    const void* fns[8] = {(const void*)attn_fwd_kernel<3, false>, (const void*)attn_fwd_kernel<3, false, true>,
                          (const void*)attn_fwd_kernel<3, false, false, true>, (const void*)attn_fwd_kernel<3, false, true, true>,
                          (const void*)attn_fwd_kernel<3, true>, (const void*)attn_fwd_kernel<3, true, true>,
                          (const void*)attn_fwd_kernel<3, true, false, true>, (const void*)attn_fwd_kernel<3, true, true, true>};
    for (int f = 0; f < 8; ++f) {
      hipError_t e = hipFuncSetAttribute(fns[f], hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
      if (e != hipSuccess) return (int)e;
    }
    attr_bytes = smem_bytes;
  }
  dim3 grid((unsigned)((S + 63) / 64) * nh * B);
#define FBL_FWD_LAUNCH(PF_)                                                                                             \
  do {                                                                                                                  \
    if (row0 && psave)                                                                                                  \
      hipLaunchKernelGGL((attn_fwd_kernel<3, PF_, true, true>), grid, dim3(256), smem_bytes, (hipStream_t)stream, a);   \
    else if (row0)                                                                                                      \
      hipLaunchKernelGGL((attn_fwd_kernel<3, PF_, true>), grid, dim3(256), smem_bytes, (hipStream_t)stream, a);         \
    else if (psave)                                                                                                     \
      hipLaunchKernelGGL((attn_fwd_kernel<3, PF_, false, true>), grid, dim3(256), smem_bytes, (hipStream_t)stream, a);  \
    else                                                                                                                \
      hipLaunchKernelGGL((attn_fwd_kernel<3, PF_>), grid, dim3(256), smem_bytes, (hipStream_t)stream, a);               \
  } while (0)
  if (pf) FBL_FWD_LAUNCH(true);
  else FBL_FWD_LAUNCH(false);
#undef FBL_FWD_LAUNCH
  FBL_CHECK_LAUNCH();
  return 0;
}
