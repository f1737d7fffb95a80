// Backward of the fused disentangled attention (gfx950).  reference: autograd of model/deberta.py:717-947 incl.
// XSoftmax.backward (:134-138) and XDropout.backward (:185-190).
//
// With  s[i,j] = scale*(Q_i.K_j + Q_i.PK[idx(i-j)] + K_j.PQ[idx(i-j)]),  P = softmax_masked(s),  O = drop(P).V :
//   dP = dO.V^T (through the dropout mask),  dS = P*(dP - D_i)*scale,  D_i = dO_i.O_i
//   dV = drop(P)^T.dO
//   dQ = dS.K   + G1.PK,    G1[i,r] = sum_{j: idx(i-j)=r} dS[i,j]        (c2p)
//   dK = dS^T.Q + G2.PQ,    G2[j,r] = sum_{i: idx(i-j)=r} dS[i,j]        (p2c)
//   dPK = sum_b G1^T.Q ,  dPQ = sum_b G2^T.K     (per head; done by the GEMM kernel on G1^T/G2^T written here)
//
// The launches of one layer execution (round 6; frozenbilm_amd/attn_bwd.py):
//   attn_bwd_prep  D = rowdot(dO, O) and the position tables expanded by the index map, PQX / PKX[h][delta + Sp][d] = table[idx(delta)]
//                  (only what the chosen route needs: the transposed copies K^T, Q^T, PK^T, PQ^T of rounds 1-5 on request).
//   attn_bwd_dspk  key-major: one workgroup per (b, h, 64-key tile) sweeps the query tiles; P from the probabilities the training
//                  forward saved (dropout decision in their sign bit), dS, dV AND dK (dS^T.Q + the p2c term as a Toeplitz product of
//                  a sheared LDS tile against PQX); writes dS and dS^T (bf16, zero where masked).
//   attn_bwd_dq    query-major: reads dS, dQ = dS.K + the c2p term as a Toeplitz product against PKX.
//   pos_grad       dPK / dPQ of ALL layer executions at the end of backward, straight from dS / dS^T (fbl_attn_pos_grad).
// Earlier routes, kept behind engine options for A/B measurements and for calls without saved probabilities:
//   attn_bwd_ds    kernel A that RECOMPUTES P exactly like the forward (row-tile split fp16 T1/T2 bias GEMMs + LDS gather, same
//                  rounding) with the keys as lane columns; attn_bwd_dsp: kernel A from saved probabilities without dK.
//   attn_bwd_shear<NEG> ("kernel BC"): one workgroup per (b, h, 32 rows): X_out = dSx.Y + G.Ptab with the scatter
//                  G[row, idx(+-(row-col))] += dSx[row,col] done by LDS stores / atomics into a [32 x W] bf16 tile, W = the index
//                  range the 32 rows can reach (~S+32 <= 512); optionally writes G^T for the position-table GEMMs of rounds 1-5.
#include "attn_common.h"
#include "../../include/fbl.h"

namespace {
using namespace attn;

// ------------------------------------------------------------------------------------------- D_i = dO_i . O_i
// 8 lanes per (row, head): lane c reads the c-th 16-byte chunk of both 128-byte head rows (a wave covers 8 consecutive
// heads = 1 KiB contiguous per operand), then a 3-step shuffle reduction.
__global__ void rowdot_kernel(const bf16* dO, const bf16* O, long ld, float* out, int B, int S, int nh) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long idx = gid >> 3;  // (row, head)
  const int c = (int)(gid & 7);
  const long total = (long)B * S * nh;
  const bool live = idx < total;
  const long row = live ? idx / nh : 0;
  const int h = live ? (int)(idx % nh) : 0;
  float s = 0.f;
  if (live) {
    const bf16x8 x = *(const bf16x8*)(dO + row * ld + h * 64 + c * 8);
    const bf16x8 y = *(const bf16x8*)(O + row * ld + h * 64 + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f(x[e]) * bf2f(y[e]);
  }
  s += __shfl_xor(s, 1, 64);
  s += __shfl_xor(s, 2, 64);
  s += __shfl_xor(s, 4, 64);
  if (live && c == 0) {
    const int bb = (int)(row / S), ss = (int)(row % S);
    out[((long)bb * nh + h) * S + ss] = s;
  }
}

// ------------------------------------------------------------------------------------------- backward preparation
// Everything the backward needs before kernel A as ONE launch instead of five: the position-contiguous copies K^T, Q^T
// (head-major [nh,64,B,Sp]: operands of the shear passes and of the position-table gradient GEMMs), PK^T, PQ^T
// ([nh,64,span2]) and D = rowdot(dO, O).  Five small kernels (two of them 1.5 MB each, i.e. pure launch latency) cannot
// fill the chip one after the other; as block ranges of one grid they run side by side.
struct PrepArgs {
  const bf16* q; const bf16* k; long ldq;
  const bf16* pq; const bf16* pk; long ldp;
  const bf16* dO; const bf16* O; long ldo;
  bf16* QT; bf16* KT; bf16* PQT; bf16* PKT; float* Dv;
  int B, S, Sp, nh, span2;
  int n_kt, n_qt, n_pkt, n_pqt;  // blocks of the K / Q transposes and of the PK / PQ table transposes (0: output not wanted)
  const int32_t* row0;  // [B+1] packed-row layout of q / k / dO / O (see attn_fwd.hip) or null; the outputs keep [B, S(p)]
  // tables expanded by the relative-index map (the fused key-/query-major passes): X[h][t][d] = tab[relidx[t - Sp + S - 1]][h*64 + d],
  // t in [0, 2 Sp) standing for delta = i - j = t - Sp (indices beyond the map's range are clamped: dS is zero there)
  const int16_t* relidx;
  bf16* PQX; bf16* PKX;
  int n_pqx, n_pkx;
};

// vt[h*sh + b*sb + d*sd + s] = v[rb+s, h*64+d] (s < S), 0 for S <= s < Sp: one (64-position tile, head, sample); rb = first
// row of the sample (b*S in the padded layout), S = number of its rows that exist
__device__ __forceinline__ void head_transpose_tile(uint32_t* tile, const bf16* v, long ldv, bf16* vt, int S, int Sp, long sh,
                                                    long sb, long sd, int s0, int h, int b, long rb) {
  const int t = threadIdx.x;
  {
    const int row = t >> 2, c0 = (t & 3) * 2;
    const int s = s0 + row;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 x = make_uint4(0, 0, 0, 0);
      if (s < S) x = *(const uint4*)(v + (rb + s) * ldv + h * 64 + (c0 + c) * 8);
      uint32_t* d = tile + row * 33 + (c0 + c) * 4;
      d[0] = x.x; d[1] = x.y; d[2] = x.z; d[3] = x.w;
    }
  }
  __syncthreads();
  const uint16_t* t16 = (const uint16_t*)tile;
  const int sc = t & 7;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int d = (t >> 3) + pass * 32;
    uint32_t w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t lo = t16[(sc * 8 + 2 * k) * 66 + d], hi = t16[(sc * 8 + 2 * k + 1) * 66 + d];
      w[k] = lo | (hi << 16);
    }
    const int s = s0 + sc * 8;
    if (s < Sp) *(uint4*)(vt + h * sh + b * sb + d * sd + s) = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

__global__ __launch_bounds__(256) void attn_bwd_prep_kernel(PrepArgs a) {
  __shared__ uint32_t tile[64 * 33];
  int id = blockIdx.x;
  if (id < a.n_kt + a.n_qt) {  // K^T (first n_kt blocks) and Q^T, head-major
    const bool isq = id >= a.n_kt;
    if (isq) id -= a.n_kt;
    const int nst = a.Sp / 64;
    const int st = id % nst, h = (id / nst) % a.nh, b = id / (nst * a.nh);
    const long rb = a.row0 ? (long)a.row0[b] : (long)b * a.S;
    const int lim = a.row0 ? min(a.row0[b + 1] - a.row0[b], a.S) : a.S;  // (rows beyond it read as zero, like S <= s < Sp)
    head_transpose_tile(tile, isq ? a.q : a.k, a.ldq, isq ? a.QT : a.KT, lim, a.Sp, 64l * a.B * a.Sp, a.Sp, (long)a.B * a.Sp,
                        st * 64, h, b, rb);
    return;
  }
  id -= a.n_kt + a.n_qt;
  if (id < a.n_pkt + a.n_pqt) {  // PK^T, PQ^T: [nh][64][span2]
    const bool isq = id >= a.n_pkt;
    if (isq) id -= a.n_pkt;
    const int nst = a.span2 / 64;
    const int st = id % nst, h = id / nst;
    head_transpose_tile(tile, isq ? a.pq : a.pk, a.ldp, isq ? a.PQT : a.PKT, a.span2, a.span2, 64l * a.span2,
                        (long)a.nh * 64 * a.span2, a.span2, st * 64, h, 0, 0);
    return;
  }
  id -= a.n_pkt + a.n_pqt;
  if (id < a.n_pqx + a.n_pkx) {  // expanded tables [nh][2 Sp][64]: a row gather, 64 rows x one head per block
    const bool isk = id >= a.n_pqx;
    if (isk) id -= a.n_pqx;
    const int W = 2 * a.Sp, nst = W / 64;
    const int st = id % nst, h = id / nst;
    const int t = st * 64 + (threadIdx.x >> 2), c0 = (threadIdx.x & 3) * 2;
    const int src = min((int)a.relidx[min(max(t - a.Sp + a.S - 1, 0), 2 * a.S - 2)], a.span2 - 1);
    const bf16* sp = (isk ? a.pk : a.pq) + (long)src * a.ldp + h * 64 + c0 * 8;
    bf16* dp = (isk ? a.PKX : a.PQX) + ((long)h * W + t) * 64 + c0 * 8;
    *(uint4*)dp = *(const uint4*)sp;
    *(uint4*)(dp + 8) = *(const uint4*)(sp + 8);
    return;
  }
  id -= a.n_pqx + a.n_pkx;
  {  // D[b,h,s] = dO_row . O_row over the head's 64 columns: 8 lanes per (row, head)
    const long gid = (long)id * 256 + threadIdx.x;
    const long idx = gid >> 3;
    const int c = (int)(gid & 7);
    const long total = (long)a.B * a.S * a.nh;
    const bool live = idx < total;
    const long row = live ? idx / a.nh : 0;  // position of the padded [B, S] grid
    const int h = live ? (int)(idx % a.nh) : 0;
    const int bb = (int)(row / a.S), ss = (int)(row % a.S);
    long arow = row;  // its activation row (packed layout: positions without one get D = 0)
    bool has = live;
    if (a.row0 && live) {
      has = ss < a.row0[bb + 1] - a.row0[bb];
      arow = (long)a.row0[bb] + ss;
    }
    float s = 0.f;
    if (has) {
      const bf16x8 x = *(const bf16x8*)(a.dO + arow * a.ldo + h * 64 + c * 8);
      const bf16x8 y = *(const bf16x8*)(a.O + arow * a.ldo + h * 64 + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += bf2f(x[e]) * bf2f(y[e]);
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (live && c == 0) a.Dv[((long)bb * a.nh + h) * a.S + ss] = s;
  }
}

// ------------------------------------------------------------------------------------------- kernel A
struct BwdAArgs {
  const bf16* q; const bf16* k; const bf16* v; long ldq;  // row-major [B*S, ld], head h at col h*64
  const bf16* dO; long ldo;                               // row-major [B*S, ldo]
  const bf16* pk; const bf16* pq; long ldp;
  const int16_t* relidx; const int32_t* mask; const int32_t* klen; const int32_t* border;
  const float* lse; const float* Dv;                      // [B,nh,S]
  float scale, p_drop; uint64_t seed;
  bf16* dV; long lddv;                                    // row-major out, head h at col h*64
  bf16* dS; bf16* dST;                                    // [B,nh,Sp,Sp]
  int B, S, Sp, nh, span2;
  int lin;  // |d| < lin: idx(d) = idx(0) + d (identity buckets); 0 = unknown
  const uint64_t* seed_dev;  // optional device word added to seed (fbl_seed); last: the offsets of the fields above are what the
                             // register allocation of this kernel was tuned with (any field in front of them costs spills)
  const int32_t* row0;       // [B+1] packed-row layout of q / k / v / dO / dV (PACKED kernels; see attn_fwd.hip) or null
};

// LDS: Q, dO tiles (per iteration) and this workgroup's K tile 3 x 8 KiB, T1 / T2 2 x 13 KiB (reused as the dS / dS^T
// staging tiles), per-row lse / D, key validity, index table = 52.5 KiB at Sp = 320: three workgroups per CU.  As in the
// forward (attn_fwd.hip) the position tables are not staged: each wave owns two of the eight 16-row tiles of the pair's
// table window, loads their MFMA A-fragments from global memory once and multiplies them against every query group
// (T1) and key group (T2) whose sub-window contains the tile.
constexpr int LTW = 100;                         // fp16 row stride of the T1/T2 tiles (6 row tiles = 96 used): 50 dwords, 16 rows -> 16 banks
constexpr int A_QS = 0;                          // [64 i][64] swz
constexpr int A_DOS = A_QS + 8192;               // [64 i][64] swz
constexpr int A_KS = A_DOS + 8192;               // [64 j][64] swz (staged once)
constexpr int A_T1 = A_KS + 8192;                // [64 i][LTW] fp16 -- reused as the [64][72] bf16 dS staging tile
constexpr int A_T2 = A_T1 + 64 * LTW * 2;        // [64 j][LTW] fp16 -- reused as the dS^T staging tile
constexpr int A_ROW = A_T2 + 64 * LTW * 2;       // float lse[64], D[64], key validity[64]
constexpr int A_IDX = A_ROW + 768;               // int16[2*Sp]
__host__ __device__ constexpr int a_total(int Sp) { return A_IDX + 2 * Sp * 2; }
static_assert(64 * LTW * 2 >= 64 * LDV * 2, "dS staging tile must fit in the T1 region");

struct ATileRegs {
  bf16x8 q[2], d[2];
  float lse, D;
};

// DEVSEED: the dropout seed gets the device word added (launch graphs).  Two instantiations on purpose: this kernel sits at
// its register cap, and the scalar load + add in front of the key derivation costs the common (DEVSEED = false) launch
// 24 more bytes of spills per lane and 14 % of its time (120 -> 136 us, same box) if it shares one body with it.
template <bool DEVSEED, bool PACKED = false>
__global__ __launch_bounds__(256, 3) void attn_bwd_ds_kernel(BwdAArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, Sp = a.Sp;
  const WgCoord wc = wg_coord(Sp / 64, a.nh, a.B, a.border);
  const int j0 = wc.x * 64, h = wc.h, b = wc.b;
  const int j = j0 + w * 16 + c;  // this lane's key
  // first activation row of this sample and the number of its rows that exist (PACKED: ragged batches without padding rows)
  const long rb = PACKED ? (long)a.row0[b] : (long)b * S;
  const int lim = PACKED ? min(a.row0[b + 1] - a.row0[b], S) : S;
  const int jc = min(j, lim - 1);
  const int tq = Sp - 1;  // idx[i - j + tq]

  int16_t* idx = (int16_t*)(smem + A_IDX);
  f16* T1 = (f16*)(smem + A_T1);
  f16* T2 = (f16*)(smem + A_T2);
  float* rlse = (float*)(smem + A_ROW);
  float* rD = rlse + 64;
  float* kms = rlse + 128;
  bf16* dst = (bf16*)(smem + A_T1);   // dS tile [query][key] after the gather
  bf16* dstT = (bf16*)(smem + A_T2);  // dS^T tile [key][query]

  const int kl = a.klen ? min(a.klen[b], S) : S;
  const int nqt = (j0 < kl) ? (kl + 63) / 64 : 0;  // tiles beyond the last valid position: dS = 0 (never read), dV = 0
  const int srow = tid >> 3, sch = tid & 7;
  // LDS addressing as lane constants + immediates (see attn_fwd.hip)
  const int fb0 = c * 128 + ((g ^ (c & 7)) << 4), fb1 = fb0 ^ 64;  // fragment of row x*16 + c: + x*2048
  const int sb = srow * 128 + ((sch ^ (srow & 7)) << 4);           // staging slot of row srow + 32 t: + t*4096

  bf16x8 vf[2];  // V fragments of this lane's key (the K fragments are read from the K tile in LDS where they are used)
  {
    const long off = (rb + jc) * a.ldq + h * 64 + g * 8;
    vf[0] = *(const bf16x8*)(a.v + off);
    vf[1] = *(const bf16x8*)(a.v + off + 32);
  }
  if (nqt > 0) {
    load_idx_padded(idx, a.relidx, S, Sp, tid, 256);
#pragma unroll
    for (int t = 0; t < 2; ++t)
      *(bf16x8*)(smem + A_KS + sb + t * 4096) =
          *(const bf16x8*)(a.k + (rb + min(j0 + srow + t * 32, lim - 1)) * a.ldq + h * 64 + sch * 8);
    if (tid < 64) kms[tid] = (j0 + tid < S && a.mask[(long)b * S + min(j0 + tid, S - 1)] != 0) ? 1.f : 0.f;
    // T1 / T2 slots that no row tile of a pair covers are still gathered by padding rows / columns (whose P is forced
    // to 0 through lse = +inf or a -inf T2 row): they must hold finite-or--inf values, never NaN bit patterns
    for (int t = tid; t < (2 * 64 * LTW * 2) / 16; t += 256) *(bf16x8*)(smem + A_T1 + t * 16) = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }

  f32x4 dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const DropKey dk = attn_drop_key(DEVSEED ? a.seed + *a.seed_dev : a.seed, b * a.nh + h, a.p_drop);
  const float k2 = a.scale * LOG2E;
  const long sbase = ((long)b * a.nh + h) * Sp * Sp;
  const int izero = a.relidx[S - 1];  // idx(0)
  const bf16* pkh = a.pk + h * 64 + g * 8;
  const bf16* pqh = a.pq + h * 64 + g * 8;
  const int my_t0 = (0x5243 >> (w * 4)) & 15, my_t1 = (0x1670 >> (w * 4)) & 15;  // this wave's window row tiles

  auto load_qd = [&](int it, ATileRegs& R) {
    const int i0 = it * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = min(i0 + srow + t * 32, lim - 1);
      R.q[t] = *(const bf16x8*)(a.q + (rb + i) * a.ldq + h * 64 + sch * 8);
      R.d[t] = *(const bf16x8*)(a.dO + (rb + i) * a.ldo + h * 64 + sch * 8);
    }
    R.lse = INFINITY; R.D = 0.f;  // lse = +inf (padding and masked queries: the forward stores +inf) -> P = 0
    if (tid < 64) {
      const int i = i0 + tid;
      if (i < lim) {
        const long o = ((long)b * a.nh + h) * S + i;
        R.lse = a.lse[o] * LOG2E;
        R.D = a.Dv[o];
      }
    }
  };
  auto store_tile = [&](const ATileRegs& R) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      *(bf16x8*)(smem + A_QS + sb + t * 4096) = R.q[t];
      *(bf16x8*)(smem + A_DOS + sb + t * 4096) = R.d[t];
    }
    if (tid < 64) {
      rlse[tid] = R.lse;
      rD[tid] = R.D;
    }
  };

  for (int it = 0; it < nqt; ++it) {
    const int i0 = it * 64;
    const bool band = max(abs(i0 - (j0 + 63)), abs(i0 + 63 - j0)) < a.lin;
    int r_lo, t0q[4], t0k[4];
    if (band) {
      r_lo = izero + i0 - (j0 + 63);
#pragma unroll
      for (int x = 0; x < 4; ++x) { t0q[x] = x; t0k[x] = 3 - x; }
    } else {
      if (it == 0) __syncthreads();  // index table
      r_lo = __builtin_amdgcn_readfirstlane((int)idx[i0 - (j0 + 63) + tq]);
#pragma unroll
      for (int x = 0; x < 4; ++x) {
        t0q[x] = (__builtin_amdgcn_readfirstlane((int)idx[i0 + x * 16 - (j0 + 63) + tq]) - r_lo) >> 4;
        t0k[x] = (__builtin_amdgcn_readfirstlane((int)idx[i0 - (j0 + x * 16 + 15) + tq]) - r_lo) >> 4;
      }
    }
    const int ntile = band ? 5 : 6;
    bf16x8 apk[2][2], apq[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long ro = (long)min(r_lo + (u ? my_t1 : my_t0) * 16 + c, a.span2 - 1) * a.ldp;
      apk[u][0] = *(const bf16x8*)(pkh + ro);
      apk[u][1] = *(const bf16x8*)(pkh + ro + 32);
      apq[u][0] = *(const bf16x8*)(pqh + ro);
      apq[u][1] = *(const bf16x8*)(pqh + ro + 32);
    }
    {
      ATileRegs R;
      load_qd(it, R);  // no register prefetch: the other workgroups of the CU cover the load latency
      store_tile(R);
    }
    __syncthreads();

    // ---- (1) scores: sacc[nt][r] = Q_i . K_j,  i = i0 + nt*16 + g*4 + r, key column c
    f32x4 sacc[4];
    const bf16x8 kf[2] = {*(const bf16x8*)(smem + A_KS + w * 2048 + fb0), *(const bf16x8*)(smem + A_KS + w * 2048 + fb1)};
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + A_QS + nt * 2048 + fb0), kf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + A_QS + nt * 2048 + fb1), kf[1], acc, 0, 0, 0);
      sacc[nt] = acc;
    }
    // ---- (2) this wave's row tiles of T1[query][.] = Q_i . PK[.] and T2[key][.] = K_j . PQ[.]  (fp16, as the forward)
    {
      const f16 ninf = (f16)(-INFINITY);
      f16* t1s = T1 + c * LTW + g * 4;  // + (x*16)*LTW + rel*16
      f16* t2s = T2 + c * LTW + g * 4;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int t = u ? my_t1 : my_t0;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
          const int rel1 = t - t0q[x];
          if (rel1 >= 0 && rel1 < ntile) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(apk[u][0], *(const bf16x8*)(smem + A_QS + x * 2048 + fb0), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(apk[u][1], *(const bf16x8*)(smem + A_QS + x * 2048 + fb1), acc, 0, 0, 0);
            *(f16x4*)(t1s + x * 16 * LTW + rel1 * 16) = to_f16x4(acc);
          }
          const int rel2 = t - t0k[x];
          if (rel2 >= 0 && rel2 < ntile) {
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(apq[u][0], *(const bf16x8*)(smem + A_KS + x * 2048 + fb0), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(apq[u][1], *(const bf16x8*)(smem + A_KS + x * 2048 + fb1), acc, 0, 0, 0);
            const f16x4 tv = to_f16x4(acc);
            // masked / padding keys: T2 row = -inf -> P = 0
            *(f16x4*)(t2s + x * 16 * LTW + rel2 * 16) = kms[x * 16 + c] != 0.f ? tv : (f16x4){ninf, ninf, ninf, ninf};
          }
        }
      }
    }
    __syncthreads();

    // ---- (3) P (same arithmetic as the forward)
    float p[16];
    if (band) {
      // c2p: T1[i][(i - j) - (r_lo + 16 nt - izero)] = T1[nt*16 + g*4 + r][g*4 + r + 63 - 16 w - c]
      // p2c: T2[j][(i - j) - (r_lo + 16 (3 - w) - izero)] = T2[16 w + c][nt*16 + g*4 + r + 15 - c]
      const f16* t1p = T1 + g * 4 * (LTW + 1) + 63 - 16 * w - c;
      const f16* t2p = T2 + (w * 16 + c) * LTW + g * 4 + 15 - c;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float s = sacc[nt][r] + (float)t1p[nt * 16 * LTW + r * (LTW + 1)] + (float)t2p[nt * 16 + r];
          p[nt * 4 + r] = __builtin_amdgcn_exp2f(fmaf(s, k2, -rlse[nt * 16 + g * 4 + r]));
        }
    } else {
      // the sub-window offsets are clamped here (one v_med3 each): padding queries must give an exact P = 0 (finite or
      // -inf score, lse = +inf), not garbage, because their dS rows are stored
      const int16_t* ib = idx + (i0 + g * 4 - j + tq);
      int sb2 = r_lo;  // first table row of this wave's T2 rows
#pragma unroll
      for (int x = 0; x < 4; ++x) sb2 = (w == x) ? r_lo + 16 * t0k[x] : sb2;
      const f16* t1g = T1 + g * 4 * LTW;
      const f16* t2row = T2 + (w * 16 + c) * LTW;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int sb1 = r_lo + 16 * t0q[nt];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int wi = ib[nt * 16 + r];
          const int w1 = clampi(wi - sb1, 0, 95), w2 = clampi(wi - sb2, 0, 95);
          const float s = sacc[nt][r] + (float)t1g[(nt * 16 + r) * LTW + w1] + (float)t2row[w2];
          p[nt * 4 + r] = __builtin_amdgcn_exp2f(fmaf(s, k2, -rlse[nt * 16 + g * 4 + r]));
        }
      }
    }
    // ---- dP = dO.V^T, dS = P*(dP - D)*scale; packed to bf16 at once (dsb: dS, pfh: dropped-out P for the dV MFMA)
    bf16x4 dsb[4], pfh[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + A_DOS + nt * 2048 + fb0), vf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + A_DOS + nt * 2048 + fb1), vf[1], acc, 0, 0, 0);
      float keep[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.p_drop > 0.f) {
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          uint32_t x, y;
          attn_drop_block(dk, (i0 + nt * 16 + g * 4 + bb * 2) >> 1, j >> 1, Sp >> 1, &x, &y);
          keep[bb * 2] = attn_drop_keep(dk, x, y, 0, j & 1);
          keep[bb * 2 + 1] = attn_drop_keep(dk, x, y, 1, j & 1);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = nt * 16 + g * 4 + r;
        const float pv = p[nt * 4 + r];
        dsb[nt][r] = f2bf(pv * (acc[r] * keep[r] - rD[il]) * a.scale);
        pfh[nt][r] = f2bf(pv * keep[r]);
      }
    }
    __syncthreads();  // every wave is done gathering from T1 / T2 (reused below as dS staging)
    // ---- (4) dV^T += dO^T . drop(P):  k-slot e of step kk <-> query kk*32 + (e>>2)*16 + g*4 + (e&3)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[e] = pfh[2 * kk][e];
        pf[4 + e] = pfh[2 * kk + 1][e];
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // dO^T fragment (row d = dt*16 + c, queries kk*32 + g*4 + {0..3} and +16) read transposed out of the swizzled
        // row-major dO tile that already feeds the dP MFMAs: lane (g, c) hands in 4 contiguous d of query row r
        const int r = g * 4 + (c >> 2);
        const int ch = dt * 2 + ((c >> 1) & 1), sub = (c & 1) * 8;
        const char* db = smem + A_DOS + r * 128 + ((ch ^ (r & 7)) << 4) + sub;
        union { tr16x4 h[2]; bf16x8 v; } u;
        u.h[0] = lds_tr16((const bf16*)(db + kk * 4096));
        u.h[1] = lds_tr16((const bf16*)(db + kk * 4096 + 2048));
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf, dv[dt], 0, 0, 0);
      }
    }
    // ---- (5) dS and dS^T leave through LDS transposes (T1 / T2 regions): 16-byte rows
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      *(bf16x4*)(dstT + (w * 16 + c) * LDV + nt * 16 + g * 4) = dsb[nt];  // [key][query]
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(nt * 16 + g * 4 + r) * LDV + w * 16 + c] = dsb[nt][r];  // [query][key]
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = srow + t * 32;
      *(bf16x8*)(a.dS + sbase + (long)(i0 + row) * Sp + j0 + sch * 8) = *(const bf16x8*)(dst + row * LDV + sch * 8);
      *(bf16x8*)(a.dST + sbase + (long)(j0 + row) * Sp + i0 + sch * 8) = *(const bf16x8*)(dstT + row * LDV + sch * 8);
    }
    __syncthreads();
  }

  if (j < lim) {
    bf16* op = a.dV + (rb + j) * a.lddv + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(bf16x4*)(op + dt * 16) = (bf16x4){f2bf(dv[dt][0]), f2bf(dv[dt][1]), f2bf(dv[dt][2]), f2bf(dv[dt][3])};
  }
}

// ------------------------------------------------------------------------------------------- kernel A, P saved by the forward
// The training forward (attn_fwd.hip, SAVEP) leaves the un-normalised probabilities of every visited tile pair in HBM --
// psave[b,h,i,j] = exp2(k2*(s_ij - m)), bf16, with m the running row maximum at key tile j/64 (msave) -- so this pass does
// not repeat Q.K^T, the two bias GEMMs against the position-table windows and the LDS gather (kernel A above: 28 of its 44
// MFMA pairs, ten fp16 tile stores, 32 element reads per lane and a barrier per tile pair; 288 GB of HBM hold the 157 MB
// per layer execution easily):  P_ij = psave_ij * exp2(msave_i - lse_i*log2(e)).  Everything after P is kernel A's:
// dP = dO.V^T through the dropout mask (the sign bits of psave), dS = P*(dP - D)*scale, dV += drop(P)^T.dO, dS and dS^T staged through
// LDS.  Two barriers per tile pair, 37 KiB of LDS, no index table, no position tables.
struct BwdPArgs {
  const bf16* psave; const float* msave;
  const bf16* v; long ldv;
  const bf16* dO; long ldo;
  const int32_t* klen; const int32_t* border;
  const float* lse; const float* Dv;  // [B,nh,S]
  float scale, p_drop; uint64_t seed; const uint64_t* seed_dev;
  bf16* dV; long lddv;
  bf16* dS; bf16* dST;                // [B,nh,Sp,Sp]
  int B, S, Sp, nh;
  const int32_t* row0;                // packed-row layout of v / dO / dV or null
};
constexpr int LDP = 80;                     // bf16 row stride of the P tile: 40 dwords = 8 x odd -> the eight consecutive rows a
                                            // ds_read_b64_tr_b16 pass touches (lanes 0-31: 32 bytes each) tile the 64 banks
constexpr int P_DOS = 0;                    // [64 i][64] bf16 swizzled
constexpr int P_PS = P_DOS + 8192;          // [64 i][LDP] bf16
constexpr int P_ST = P_PS + 64 * LDP * 2;   // dS staging [64 i][LDV]
constexpr int P_STT = P_ST + 64 * LDV * 2;  // dS^T staging [64 j][LDV]
constexpr int P_ROW = P_STT + 64 * LDV * 2; // float f[64], D[64]
constexpr int P_TOTAL = P_ROW + 512;

struct PTileRegs {
  bf16x8 d[2], p[2];
  float f, D;
};

template <bool PACKED = false>
__global__ __launch_bounds__(256, 3) void attn_bwd_dsp_kernel(BwdPArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, Sp = a.Sp;
  const WgCoord wc = wg_coord(Sp / 64, a.nh, a.B, a.border);
  const int j0 = wc.x * 64, h = wc.h, b = wc.b;
  const int j = j0 + w * 16 + c;  // this lane's key
  const long rb = PACKED ? (long)a.row0[b] : (long)b * S;
  const int lim = PACKED ? min(a.row0[b + 1] - a.row0[b], S) : S;
  const int jc = min(j, lim - 1);
  float* rF = (float*)(smem + P_ROW);
  float* rD = rF + 64;
  bf16* dst = (bf16*)(smem + P_ST);    // dS tile [query][key]
  bf16* dstT = (bf16*)(smem + P_STT);  // dS^T tile [key][query]

  const int kl = a.klen ? min(a.klen[b], S) : S;
  const int nqt = (j0 < kl) ? (kl + 63) / 64 : 0;
  const int srow = tid >> 3, sch = tid & 7;
  const int fb0 = c * 128 + ((g ^ (c & 7)) << 4), fb1 = fb0 ^ 64;  // dO fragment of row x*16 + c: + x*2048
  const int sb = srow * 128 + ((sch ^ (srow & 7)) << 4);           // dO staging slot of row srow + 32 t: + t*4096
  const int sp = srow * (LDP * 2) + sch * 16;                      // P staging slot: + t*32*LDP*2

  bf16x8 vf[2];
  {
    const long off = (rb + jc) * a.ldv + h * 64 + g * 8;
    vf[0] = *(const bf16x8*)(a.v + off);
    vf[1] = *(const bf16x8*)(a.v + off + 32);
  }
  f32x4 dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const DropKey dk = attn_drop_key(a.p_drop > 0.f ? fbl_seed(a.seed, a.seed_dev) : 0, b * a.nh + h, a.p_drop);
  const long sbase = ((long)b * a.nh + h) * Sp * Sp;
  const float* mrow = a.msave + (((long)b * a.nh + h) * (Sp >> 6) + wc.x) * S;

  auto load_tile = [&](int it, PTileRegs& R) {
    const int i0 = it * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int il = i0 + srow + t * 32;
      R.d[t] = *(const bf16x8*)(a.dO + (rb + min(il, lim - 1)) * a.ldo + h * 64 + sch * 8);
      R.p[t] = *(const bf16x8*)(a.psave + sbase + (long)il * Sp + j0 + sch * 8);
    }
    R.f = 0.f; R.D = 0.f;  // padding and masked queries: lse = +inf in the forward -> P = 0
    if (tid < 64) {
      const int i = i0 + tid;
      if (i < lim) {
        const long o = ((long)b * a.nh + h) * S + i;
        const float l = a.lse[o];
        R.f = (l < INFINITY) ? __builtin_amdgcn_exp2f(mrow[i] - l * LOG2E) : 0.f;
        R.D = a.Dv[o];
      }
    }
  };
  auto store_tile = [&](const PTileRegs& R) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      *(bf16x8*)(smem + P_DOS + sb + t * 4096) = R.d[t];
      *(bf16x8*)(smem + P_PS + sp + t * (32 * LDP * 2)) = R.p[t];
    }
    if (tid < 64) {
      rF[tid] = R.f;
      rD[tid] = R.D;
    }
  };

  PTileRegs R;
  if (nqt > 0) load_tile(0, R);
  for (int it = 0; it < nqt; ++it) {
    const int i0 = it * 64;
    store_tile(R);
    __syncthreads();  // (also: every wave is done with the staging tiles of the previous pair)
    if (it + 1 < nqt) load_tile(it + 1, R);  // the next pair's operands fly during this pair's arithmetic

    bf16x4 dsb[4], pfh[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      // P~ of queries nt*16 + g*4 + {0..3}, key column c: the transposing read hands each lane its column of a [4 x 16] block
      union { tr16x4 t; bf16x4 v; } pu;
      pu.t = lds_tr16((const bf16*)(smem + P_PS + (nt * 16 + g * 4 + (c >> 2)) * (LDP * 2) + (w * 16 + (c & 3) * 4) * 2));
      const f32x4 f4 = *(const f32x4*)(rF + nt * 16 + g * 4);
      const f32x4 d4 = *(const f32x4*)(rD + nt * 16 + g * 4);
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + P_DOS + nt * 2048 + fb0), vf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + P_DOS + nt * 2048 + fb1), vf[1], acc, 0, 0, 0);
      // the forward left the dropout decision in the sign bit of the saved probability (P >= 0: set = dropped)
      float keep[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) keep[r] = pu.t[r] < 0 ? 0.f : dk.inv_keep;  // (the 16-bit pattern as a signed integer)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = f4[r] != 0.f ? fabsf(bf2f(pu.v[r])) * f4[r] : 0.f;  // (f = 0 rows may hold garbage in psave)
        dsb[nt][r] = f2bf(pv * (acc[r] * keep[r] - d4[r]) * a.scale);
        pfh[nt][r] = f2bf(pv * keep[r]);
      }
    }
    // ---- dV^T += dO^T . drop(P):  k-slot e of step kk <-> query kk*32 + (e>>2)*16 + g*4 + (e&3)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 pf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[e] = pfh[2 * kk][e];
        pf[4 + e] = pfh[2 * kk + 1][e];
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int r = g * 4 + (c >> 2);
        const int ch = dt * 2 + ((c >> 1) & 1), sub = (c & 1) * 8;
        const char* db = smem + P_DOS + r * 128 + ((ch ^ (r & 7)) << 4) + sub;
        union { tr16x4 h[2]; bf16x8 v; } u;
        u.h[0] = lds_tr16((const bf16*)(db + kk * 4096));
        u.h[1] = lds_tr16((const bf16*)(db + kk * 4096 + 2048));
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf, dv[dt], 0, 0, 0);
      }
    }
    // ---- dS and dS^T leave through LDS transposes: 16-byte rows
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      *(bf16x4*)(dstT + (w * 16 + c) * LDV + nt * 16 + g * 4) = dsb[nt];  // [key][query]
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(nt * 16 + g * 4 + r) * LDV + w * 16 + c] = dsb[nt][r];  // [query][key]
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = srow + t * 32;
      *(bf16x8*)(a.dS + sbase + (long)(i0 + row) * Sp + j0 + sch * 8) = *(const bf16x8*)(dst + row * LDV + sch * 8);
      *(bf16x8*)(a.dST + sbase + (long)(j0 + row) * Sp + i0 + sch * 8) = *(const bf16x8*)(dstT + row * LDV + sch * 8);
    }
  }

  if (j < lim) {
    bf16* op = a.dV + (rb + j) * a.lddv + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(bf16x4*)(op + dt * 16) = (bf16x4){f2bf(dv[dt][0]), f2bf(dv[dt][1]), f2bf(dv[dt][2]), f2bf(dv[dt][3])};
  }
}

// ------------------------------------------------------------------------------------------- kernel A + the key-major shear pass
// attn_bwd_dsp with dK formed in place (round 6): the workgroup of a 64-key tile already holds every dS[i, j] of its keys, one
// query tile at a time, so both terms of
//     dK_j = sum_i dS[i,j] Q_i  +  sum_i dS[i,j] PQ[idx(i-j)]
// are accumulated next to dV instead of by a second kernel that reads dS^T back (attn_bwd_shear<NEG = 1>: 78 us and 157 MB per
// layer execution at the bench shape):
//  * dK^T += Q^T . dS: the dS values are MFMA B operands exactly as they sit in the accumulator layout (like drop(P) for dV), the
//    Q^T fragments are transposing reads of a row-major Q tile staged beside the dO tile;
//  * the position term is a Toeplitz product, dK_j += sum_delta dS[j+delta, j] PQX[delta], with PQX the table expanded by the
//    relative-index map (fbl_attn_bwd_prep: PQX[h][d][delta + Sp] = PQ[idx(delta)], so log buckets and the identity band are
//    one case and the kernel holds no index table).  Each wave writes its 16 keys' dS values a second time into a SHEARED tile
//    G2[j][x], x = (i - i0) - (j - j0) + 64 (16-bit stores; the zero entries of the parallelogram are written once, before the
//    loop) and reads aligned 16-byte MFMA B fragments back from its own rows -- no barrier.  The 128 table rows a pair reaches
//    are two 64-row blocks of PQX (8 KiB each, contiguous); consecutive pairs share one, so every pair stages ONE new block in
//    LDS beside Q / dO (ring of two; a first version fetched the A fragments PQX^T[d][delta..+7] straight from a transposed
//    table: twelve 16-byte loads per lane and pair in 64-byte segments, three times the bytes, 20 of the kernel's 110 us) and
//    the A fragments are transposing LDS reads.  A wave's keys reach 96 of the 128 deltas of a pair: 3 k-steps x 4 MFMAs.
// dS / dS^T are still written (fbl_attn_pos_grad and the query-major pass read them).
// Swizzle of the [64 rows][8 x 16 B] images whose MFMA fragments are eight CONSECUTIVE rows of one column (natural k order:
// transposing reads of rows g*8 + {0..3} and + 4): the 32 lanes of a read pass touch rows {0..3} and {8..11} (+16 k), so the
// chunk index is XOR-ed with (row & 3) | (bit 3 of the row) << 2 -- with the usual row & 7 rows r and r + 8 collide (2-way on
// every read: 30 % of the LDS cycles of the first version of these kernels).
__device__ __forceinline__ int tswz(int r) { return (r & 3) | ((r >> 1) & 4); }

struct BwdPKArgs {
  BwdPArgs p;
  const bf16* q; long ldq;   // row-major like v, head h at column h*64
  const bf16* pqx;           // [nh][2 Sp][64]
  bf16* dK; long lddk;
};
constexpr int LDG2 = 144;                    // bf16 row stride of the sheared tile (128 used; 72 dwords: conflict-free b128 reads and 16-bit stores)
constexpr int K_DOS = 0;                     // [64 i][64] bf16 swizzled
constexpr int K_QS = K_DOS + 8192;           // [64 i][64] bf16 swizzled
constexpr int K_PS = K_QS + 8192;            // [64 i][LDP] bf16: P~, then (same columns per wave) the dS staging tile
constexpr int K_STT = K_PS + 64 * LDP * 2;   // dS^T staging [64 j][LDV]
constexpr int K_G2 = K_STT + 64 * LDV * 2;   // sheared dS^T [64 j][LDG2]
constexpr int K_TAB = K_G2 + 64 * LDG2 * 2;  // two [64 t][64] bf16 blocks of PQX, chunk ^= tswz(row): block beta in slot beta & 1
constexpr int K_ROW = K_TAB + 16384;         // float f[64], D[64]
constexpr int K_TOTAL = K_ROW + 512;
static_assert(2 * K_TOTAL <= 160 * 1024, "two workgroups per CU");

struct PKTileRegs {
  bf16x8 d[2], p[2], q[2], x[2];
  float m, l, D;  // raw: msave, lse, D of query row tid (f = exp2(m - l*log2 e) is formed when the tile is stored: a use right behind
                  // the loads would park wave 0 -- and with it the workgroup's next barrier -- for a full memory latency per pair)
};

template <bool PACKED = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dspk_kernel(BwdPKArgs ka) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const BwdPArgs& a = ka.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, Sp = a.Sp;
  const WgCoord wc = wg_coord(Sp / 64, a.nh, a.B, a.border);
  const int j0 = wc.x * 64, h = wc.h, b = wc.b;
  const int j = j0 + w * 16 + c;  // this lane's key
  const long rb = PACKED ? (long)a.row0[b] : (long)b * S;
  const int lim = PACKED ? min(a.row0[b + 1] - a.row0[b], S) : S;
  const int jc = min(j, lim - 1);
  float* rF = (float*)(smem + K_ROW);
  float* rD = rF + 64;
  bf16* dst = (bf16*)(smem + K_PS);     // dS tile [query][key], row stride LDP (aliases the P~ tile: a wave reads and writes
                                        // only the 16 key columns it owns)
  bf16* dstT = (bf16*)(smem + K_STT);   // dS^T tile [key][query]
  bf16* g2 = (bf16*)(smem + K_G2) + (w * 16 + c) * LDG2;  // this lane's row of the sheared tile

  const int kl = a.klen ? min(a.klen[b], S) : S;
  const int nqt = (j0 < kl) ? (kl + 63) / 64 : 0;
  const int srow = tid >> 3, sch = tid & 7;
  const int fb0 = c * 128 + ((g ^ (c & 7)) << 4), fb1 = fb0 ^ 64;  // dO fragment of row x*16 + c: + x*2048
  const int sb = srow * 128 + ((sch ^ (srow & 7)) << 4);           // dO / Q staging slot of row srow + 32 t: + t*4096
  const int sp = srow * (LDP * 2) + sch * 16;                      // P staging slot: + t*32*LDP*2
  const int sbx = srow * 128 + ((sch ^ tswz(srow)) << 4);          // table staging slot of row srow + 32 t: + t*4096 (tswz(r + 32) = tswz(r))

  bf16x8 vf[2];
  {
    const long off = (rb + jc) * a.ldv + h * 64 + g * 8;
    vf[0] = *(const bf16x8*)(a.v + off);
    vf[1] = *(const bf16x8*)(a.v + off + 32);
  }
  f32x4 dv[4], dk[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const DropKey dkey = attn_drop_key(a.p_drop > 0.f ? fbl_seed(a.seed, a.seed_dev) : 0, b * a.nh + h, a.p_drop);
  const long sbase = ((long)b * a.nh + h) * Sp * Sp;
  const float* mrow = a.msave + (((long)b * a.nh + h) * (Sp >> 6) + wc.x) * S;
  // the sheared tile starts as zeros; the loop only ever rewrites the entries inside the parallelogram
  if (nqt > 0)
    for (int t = tid; t < (64 * LDG2 * 2) / 16; t += 256) *(bf16x8*)(smem + K_G2 + t * 16) = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  // PQX^T fragments: row d = dt*16 + c, deltas t0 + ks*32 + g*8 .. +7 (t0 = i0 - j0 - 64 + Sp, a multiple of 64)
  const int W2 = 2 * Sp;
  const int ks0 = (w < 2) ? 1 : 0;  // this wave's keys reach x in [49 - 16 w, 128 - 16 w): three of the four 32-wide k-steps
  // Global addresses as a wave-uniform base (buffer descriptor in scalar registers, advanced per pair by a scalar offset) + a
  // 32-bit lane offset that does not change over the loop: 64-bit per-lane pointers for the seven streams of this kernel cost
  // 24 spilled registers
  const __amdgpu_buffer_rsrc_t xr = buf_rsrc(ka.pqx + (long)h * W2 * 64);
  const int beta0 = (Sp - j0 - 64) >> 6;  // first table block of pair 0 (pair it: beta0 + it and the next one)
  const __amdgpu_buffer_rsrc_t dOr = buf_rsrc(a.dO + rb * a.ldo + h * 64);
  const __amdgpu_buffer_rsrc_t qr = buf_rsrc(ka.q + rb * ka.ldq + h * 64);
  const __amdgpu_buffer_rsrc_t pr = buf_rsrc(a.psave + sbase + j0);
  const __amdgpu_buffer_rsrc_t dSr = buf_rsrc(a.dS + sbase + j0);
  const __amdgpu_buffer_rsrc_t dSTr = buf_rsrc(a.dST + sbase + (long)j0 * Sp);
  const uint32_t ldo2 = (uint32_t)(a.ldo * 2), ldq2 = (uint32_t)(ka.ldq * 2);
  const uint32_t po = (uint32_t)((srow * Sp + sch * 8) * 2);  // element (srow, sch*8) of a [64 x 64] block of an [Sp x Sp] matrix: + t*32*Sp*2
  auto load_tile = [&](int it, PKTileRegs& R) {
    const int i0 = it * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t ir = (uint32_t)min(i0 + srow + t * 32, lim - 1);
      R.d[t] = buf_ld16(dOr, ir * ldo2 + (uint32_t)(sch * 16), 0);
      R.p[t] = buf_ld16(pr, po + (uint32_t)(t * 32 * Sp * 2), (uint32_t)(i0 * Sp * 2));
    }
    // table block beta0 + it + 1 (block beta0 + it is the previous pair's second one)
#pragma unroll
    for (int t = 0; t < 2; ++t) R.x[t] = buf_ld16(xr, (uint32_t)((srow + t * 32) * 128 + sch * 16), (uint32_t)((beta0 + it + 1) * 8192));
    {  // (every wave loads, clamped: no branch, no wait; only wave 0's values are stored)
      const int i = min(i0 + lane, lim - 1);
      const long o = ((long)b * a.nh + h) * S + i;
      R.l = a.lse[o];
      R.m = mrow[i];
      R.D = a.Dv[o];
    }
  };
  // (the Q rows of the next pair are requested later in the pair than dO / P~: at the top of the pair the registers are full)
  auto load_q = [&](int it, PKTileRegs& R) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const uint32_t ir = (uint32_t)min(it * 64 + srow + t * 32, lim - 1);
      R.q[t] = (FBL_ATTN_DBGBITS & 16) ? vf[0] : buf_ld16(qr, ir * ldq2 + (uint32_t)(sch * 16), 0);
    }
  };
  auto store_tile = [&](const PKTileRegs& R, int it_i0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      *(bf16x8*)(smem + K_DOS + sb + t * 4096) = R.d[t];
      *(bf16x8*)(smem + K_QS + sb + t * 4096) = R.q[t];
      *(bf16x8*)(smem + K_TAB + (((it_i0 >> 6) + beta0 + 1) & 1) * 8192 + sbx + t * 4096) = R.x[t];
      *(bf16x8*)(smem + K_PS + sp + t * (32 * LDP * 2)) = R.p[t];
    }
    if (tid < 64) {  // padding and masked queries: lse = +inf in the forward -> P = 0
      const bool live = it_i0 + tid < lim && R.l < INFINITY;
      rF[tid] = live ? __builtin_amdgcn_exp2f(R.m - R.l * LOG2E) : 0.f;
      rD[tid] = live ? R.D : 0.f;
    }
  };

  PKTileRegs R;
  if (nqt > 0) {
    load_tile(0, R);
    load_q(0, R);
#pragma unroll
    for (int t = 0; t < 2; ++t)  // block beta0 (pair 0's first one)
      *(bf16x8*)(smem + K_TAB + (beta0 & 1) * 8192 + sbx + t * 4096) =
          buf_ld16(xr, (uint32_t)((srow + t * 32) * 128 + sch * 16), (uint32_t)(beta0 * 8192));
  }
  for (int it = 0; it < nqt; ++it) {
    const int i0 = it * 64;
    store_tile(R, i0);
    __syncthreads();  // (also: every wave is done with the staging tiles of the previous pair)
    {  // the next pair's operands fly during this pair's arithmetic.  Unconditional (the last pair re-requests its own tile):
       // behind a branch the compiler cannot count these requests and every later wait would cover them -- an HBM latency per pair
      const int itn = min(it + 1, nqt - 1);
      load_tile(itn, R);
      load_q(itn, R);
    }

    bf16x4 dsb[4], pfh[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      union { tr16x4 t; bf16x4 v; } pu;
      pu.t = lds_tr16((const bf16*)(smem + K_PS + (nt * 16 + g * 4 + (c >> 2)) * (LDP * 2) + (w * 16 + (c & 3) * 4) * 2));
      const f32x4 f4 = *(const f32x4*)(rF + nt * 16 + g * 4);
      const f32x4 d4 = *(const f32x4*)(rD + nt * 16 + g * 4);
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + K_DOS + nt * 2048 + fb0), vf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(smem + K_DOS + nt * 2048 + fb1), vf[1], acc, 0, 0, 0);
      // the forward left the dropout decision in the sign bit of the saved probability (P >= 0: set = dropped)
      float keep[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) keep[r] = pu.t[r] < 0 ? 0.f : dkey.inv_keep;  // (the 16-bit pattern as a signed integer)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = f4[r] != 0.f ? fabsf(bf2f(pu.v[r])) * f4[r] : 0.f;  // (f = 0 rows may hold garbage in psave)
        dsb[nt][r] = f2bf(pv * (acc[r] * keep[r] - d4[r]) * a.scale);
        pfh[nt][r] = f2bf(pv * keep[r]);
      }
    }
    // ---- dV^T += dO^T . drop(P) and dK^T += Q^T . dS:  k-slot e of step kk <-> query kk*32 + (e>>2)*16 + g*4 + (e&3)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 pf, sf;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[e] = pfh[2 * kk][e];
        pf[4 + e] = pfh[2 * kk + 1][e];
        sf[e] = dsb[2 * kk][e];
        sf[4 + e] = dsb[2 * kk + 1][e];
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int r = g * 4 + (c >> 2);
        const int ch = dt * 2 + ((c >> 1) & 1), sub = (c & 1) * 8;
        const int o = r * 128 + ((ch ^ (r & 7)) << 4) + sub + kk * 4096;
        union { tr16x4 h[2]; bf16x8 v; } u, uq;
        u.h[0] = lds_tr16((const bf16*)(smem + K_DOS + o));
        u.h[1] = lds_tr16((const bf16*)(smem + K_DOS + o + 2048));
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, pf, dv[dt], 0, 0, 0);
        if (FBL_ATTN_DBGBITS & 2) continue;  // (debug builds: no Q^T . dS)
        uq.h[0] = lds_tr16((const bf16*)(smem + K_QS + o));
        uq.h[1] = lds_tr16((const bf16*)(smem + K_QS + o + 2048));
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(uq.v, sf, dk[dt], 0, 0, 0);
      }
    }
    // ---- dS (row-major, over the P~ columns this wave has consumed), dS^T and the sheared copy
    {
      bf16* gs = g2 + 64 - (w * 16 + c) + g * 4;  // x of query nt*16 + g*4 + r: + nt*16 + r
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        *(bf16x4*)(dstT + (w * 16 + c) * LDV + nt * 16 + g * 4) = dsb[nt];  // [key][query]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dst[(nt * 16 + g * 4 + r) * LDP + w * 16 + c] = dsb[nt][r];  // [query][key]
          if (!(FBL_ATTN_DBGBITS & 4)) gs[nt * 16 + r] = dsb[nt][r];
        }
      }
    }
    // ---- dK^T += PQX^T . G2^T over this wave's three k-steps (before the barrier: the next pair's staging overwrites the older
    //      table block; the sheared rows are this wave's own).  A fragment of k-step ks, column tile dt: table rows x = ks*32 + g*8
    //      + {0..7} of the window (block x >> 6), column d = dt*16 + c -- two transposing reads of four rows each
    if (!(FBL_ATTN_DBGBITS & 8)) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int x0 = (ks0 + u) * 32 + g * 8;  // (+ c >> 2: the row this lane addresses)
        const int slot = (((i0 >> 6) + beta0 + (x0 >> 6)) & 1) * 8192;
        const int r = (x0 & 63) + (c >> 2);
        const bf16x8 gf = *(const bf16x8*)(g2 + (ks0 + u) * 32 + g * 8);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int ch = dt * 2 + ((c >> 1) & 1), sub = (c & 1) * 8;
          union { tr16x4 h[2]; bf16x8 v; } ut;
          ut.h[0] = lds_tr16((const bf16*)(smem + K_TAB + slot + r * 128 + ((ch ^ tswz(r)) << 4) + sub));
          ut.h[1] = lds_tr16((const bf16*)(smem + K_TAB + slot + (r + 4) * 128 + ((ch ^ tswz(r + 4)) << 4) + sub));
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ut.v, gf, dk[dt], 0, 0, 0);
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = srow + t * 32;
      buf_st16(dSr, po + (uint32_t)(t * 32 * Sp * 2), (uint32_t)(i0 * Sp * 2), *(const bf16x8*)(dst + row * LDP + sch * 8));
      buf_st16(dSTr, po + (uint32_t)(t * 32 * Sp * 2), (uint32_t)(i0 * 2), *(const bf16x8*)(dstT + row * LDV + sch * 8));
    }
  }

  if (j < lim) {
    bf16* op = a.dV + (rb + j) * a.lddv + h * 64 + g * 4;
    bf16* ok = ka.dK + (rb + j) * ka.lddk + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      *(bf16x4*)(op + dt * 16) = (bf16x4){f2bf(dv[dt][0]), f2bf(dv[dt][1]), f2bf(dv[dt][2]), f2bf(dv[dt][3])};
      *(bf16x4*)(ok + dt * 16) = (bf16x4){f2bf(dk[dt][0]), f2bf(dk[dt][1]), f2bf(dk[dt][2]), f2bf(dk[dt][3])};
    }
  }
}

// ------------------------------------------------------------------------------------------- query-major pass (Toeplitz form)
// dQ_i = sum_j dS[i,j] K_j + sum_j dS[i,j] PK[idx(i-j)]  from the dS tensor kernel A wrote -- what attn_bwd_shear<NEG = 0> computes by
// scattering dS into index space (LDS stores / atomics per element, an index table, a table GEMM over the whole window): here, like
// the key-major half inside attn_bwd_dspk, the position term is a Toeplitz product against PKX (the table expanded by the index
// map): the 64 x 64 dS tile of a pair is stored twice by the wave that owns its rows -- row-major (MFMA B fragments of dS.K) and
// sheared, G1[i][x], x = (i - i0) - (j - j0) + 64 -- and the 128 table rows of the pair are two 64-row blocks of PKX of which
// consecutive pairs share one.  One workgroup = (sample, head, 64 queries); it walks the key tiles below the sample's length.
// The dS rows and the sheared rows are wave-private (no barrier between their store and their use); the K tile and the table
// blocks are shared: two barriers per pair.
struct BwdQArgs {
  const bf16* dS;            // [B,nh,Sp,Sp]
  const bf16* k; long ldk;   // row-major rows b*S + s (or packed), head h at column h*64
  const bf16* pkx;           // [nh][2 Sp][64]
  const int32_t* klen; const int32_t* border;
  bf16* dQ; long lddq;
  int B, S, Sp, nh;
  const int32_t* row0;
};
constexpr int LDG1 = 104;                    // bf16 row stride of the sheared tile: the 96 columns x - 32*ks0 a wave's rows reach
constexpr int Q_DS = 0;                      // [64 i][64 j] bf16 swizzled (rows wave-private)
constexpr int Q_KS = Q_DS + 8192;            // [64 j][64 d] bf16 swizzled
constexpr int Q_G1 = Q_KS + 8192;            // [64 i][LDG1]
constexpr int Q_TAB = Q_G1 + 64 * LDG1 * 2;  // two [64 t][64] blocks of PKX: block beta in slot beta & 1
constexpr int Q_TOTAL = Q_TAB + 16384;
static_assert(3 * Q_TOTAL <= 160 * 1024, "three workgroups per CU");

template <bool PACKED = false>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(BwdQArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, Sp = a.Sp;
  const WgCoord wc = wg_coord(Sp / 64, a.nh, a.B, a.border);
  const int i0 = wc.x * 64, h = wc.h, b = wc.b;
  const int i = i0 + w * 16 + c;  // this lane's query
  const long rb = PACKED ? (long)a.row0[b] : (long)b * S;
  const int lim = PACKED ? min(a.row0[b + 1] - a.row0[b], S) : S;
  const int kl = a.klen ? min(a.klen[b], S) : S;
  const int nkt = (i0 < kl) ? (kl + 63) / 64 : 0;  // rows beyond the last valid position: dS = 0 (never written) -> dQ = 0
  const int srow = tid >> 3, sch = tid & 7;
  const int sb = srow * 128 + ((sch ^ tswz(srow)) << 4);  // K / table staging slot of row srow + 32 t: + t*4096 (tswz(r + 32) = tswz(r))
  const int ks0 = (w < 2) ? 0 : 1;  // this wave's rows reach x in [16 w + 1, 16 w + 79]: three of the four 32-wide k-steps
  bf16* g1 = (bf16*)(smem + Q_G1) + (w * 16 + c) * LDG1;  // this lane's row of the sheared tile (column x - 32*ks0)
  const int wr = w * 16 + c;                              // its row of the tile
  f32x4 dq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (nkt > 0) {  // the sheared tile starts as zeros (the loop only rewrites the entries inside the parallelogram)
    for (int t = tid; t < (64 * LDG1 * 2) / 16; t += 256) *(bf16x8*)(smem + Q_G1 + t * 16) = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
    __syncthreads();
  }

  const long sbase = ((long)b * a.nh + h) * Sp * Sp;
  const __amdgpu_buffer_rsrc_t sr = buf_rsrc(a.dS + sbase + (long)i0 * Sp);
  const __amdgpu_buffer_rsrc_t kr = buf_rsrc(a.k + rb * a.ldk + h * 64);
  const __amdgpu_buffer_rsrc_t xr = buf_rsrc(a.pkx + (long)h * 2 * Sp * 64);
  const uint32_t so = (uint32_t)((wr * Sp + g * 16) * 2);  // row wr, columns g*16 .. +15 of the tile
  const uint32_t ldk2 = (uint32_t)(a.ldk * 2);
  const int beta0 = (i0 - 64 + Sp) >> 6;  // first table block of the pair with key tile 0; key tile jt: blocks beta0 - jt, + 1

  struct Regs { bf16x8 s[2], k[2], x[2]; } R;
  auto load_pair = [&](int jt) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      R.s[t] = buf_ld16(sr, so + (uint32_t)(t * 16), (uint32_t)(jt * 128));
      const uint32_t jr = (uint32_t)min(jt * 64 + srow + t * 32, lim - 1);
      R.k[t] = buf_ld16(kr, jr * ldk2 + (uint32_t)(sch * 16), 0);
      R.x[t] = buf_ld16(xr, (uint32_t)((srow + t * 32) * 128 + sch * 16), (uint32_t)((beta0 - jt) * 8192));  // the pair's LOWER block
    }
  };
  if (nkt > 0) {
    load_pair(0);
#pragma unroll
    for (int t = 0; t < 2; ++t)  // pair 0's upper block
      *(bf16x8*)(smem + Q_TAB + ((beta0 + 1) & 1) * 8192 + sb + t * 4096) =
          buf_ld16(xr, (uint32_t)((srow + t * 32) * 128 + sch * 16), (uint32_t)((beta0 + 1) * 8192));
  }
  for (int jt = 0; jt < nkt; ++jt) {
    // ---- staging: K tile and the new table block (shared), this wave's dS rows row-major and sheared (private)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      *(bf16x8*)(smem + Q_KS + sb + t * 4096) = R.k[t];
      *(bf16x8*)(smem + Q_TAB + ((beta0 - jt) & 1) * 8192 + sb + t * 4096) = R.x[t];
      *(bf16x8*)(smem + Q_DS + wr * 128 + (((2 * g + t) ^ (wr & 7)) << 4)) = R.s[t];
    }
    {
      // element e of R.s[t]: key column cj = g*16 + t*8 + e -> x = wr - cj + 64, stored at column x - 32*ks0
      bf16* gs = g1 + wr + 64 - g * 16 - ks0 * 32;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) gs[-(t * 8 + e)] = R.s[t][e];
    }
    __syncthreads();
    load_pair(min(jt + 1, nkt - 1));  // (unconditional: see attn_bwd_dspk)
    // ---- dQ^T += K^T . dS^T: A = transposing reads of the K tile, B = this lane's dS row (k = key)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int rk = kk * 32 + g * 8 + (c >> 2);
      const bf16x8 sf = *(const bf16x8*)(smem + Q_DS + wr * 128 + (((kk * 4 + g) ^ (wr & 7)) << 4));
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int ch = dt * 2 + ((c >> 1) & 1), sub = (c & 1) * 8;
        union { tr16x4 h[2]; bf16x8 v; } u;
        u.h[0] = lds_tr16((const bf16*)(smem + Q_KS + rk * 128 + ((ch ^ tswz(rk)) << 4) + sub));
        u.h[1] = lds_tr16((const bf16*)(smem + Q_KS + (rk + 4) * 128 + ((ch ^ tswz(rk + 4)) << 4) + sub));
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.v, sf, dq[dt], 0, 0, 0);
      }
    }
    // ---- dQ^T += PKX^T . G1^T over this wave's three k-steps
#pragma unroll
    for (int u3 = 0; u3 < 3; ++u3) {
      const int x0 = (ks0 + u3) * 32 + g * 8;
      const int slot = ((beta0 - jt + (x0 >> 6)) & 1) * 8192;
      const int r = (x0 & 63) + (c >> 2);
      const bf16x8 gf = *(const bf16x8*)(g1 + u3 * 32 + g * 8);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int ch = dt * 2 + ((c >> 1) & 1), sub = (c & 1) * 8;
        union { tr16x4 h[2]; bf16x8 v; } ut;
        ut.h[0] = lds_tr16((const bf16*)(smem + Q_TAB + slot + r * 128 + ((ch ^ tswz(r)) << 4) + sub));
        ut.h[1] = lds_tr16((const bf16*)(smem + Q_TAB + slot + (r + 4) * 128 + ((ch ^ tswz(r + 4)) << 4) + sub));
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ut.v, gf, dq[dt], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done with the K tile and the older table block
  }
  if (i < lim) {
    bf16* op = a.dQ + (rb + i) * a.lddq + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(bf16x4*)(op + dt * 16) = (bf16x4){f2bf(dq[dt][0]), f2bf(dq[dt][1]), f2bf(dq[dt][2]), f2bf(dq[dt][3])};
  }
}

// ------------------------------------------------------------------------------------------- kernel BC
struct ShearArgs {
  const bf16* X;                          // dS (NEG=0) or dS^T (NEG=1): [B,nh,Sp,Sp], rows = output rows
  const bf16* YT; long y_sh, y_sb, y_sd;  // transposed K (NEG=0) / Q (NEG=1): index h*sh + b*sb + d*sd + s
  const bf16* PT;                         // transposed position table [nh][64][span2]
  const int16_t* relidx;
  const int32_t* klen;
  const int32_t* border;
  bf16* out; long ldout;                  // row-major, head h at col h*64
  bf16* GT;                               // [nh][B][Sp/32][rcnt][32]
  int B, S, Sp, nh, span2, Wg;            // Wg: columns of the G tile (multiple of 32)
  int rmin, rcnt;                         // only rows [rmin, rmin+rcnt) of G^T can be non-zero (range of relidx)
  int lin;                                // |delta| < lin: idx(delta) is injective (identity buckets) -> plain stores
  const int32_t* row0;                    // [B+1] packed-row layout of `out` (PACKED kernels; see attn_fwd.hip) or null
  const uint32_t* tilemask;               // [B][Sp/64] (fbl_gt_tilemask) or null: bit t set <=> rows [128t, 128t+128) of the rcnt
                                          // G^T rows can be non-zero in this 64-row k-step; other rows are NOT written (nor read)
};
// rows [rbase, rbase + nks*32) of the position tables that the 32 output rows r0.. of a sample with kl valid positions reach
__device__ __forceinline__ void gt_window(const int16_t* relidx, int S, int kl, int r0, bool neg, int Wg, int* rbase, int* nks) {
  const int hi = 2 * S - 2;
  const int dlo = neg ? -(r0 + 31) : r0 - (kl - 1);
  const int dhi = neg ? (kl - 1 - r0) : (r0 + 31);
  const int rb = (int)relidx[min(max(dlo + S - 1, 0), hi)] & ~7;
  const int rt = (int)relidx[min(max(dhi + S - 1, 0), hi)];
  *rbase = rb;
  *nks = min((rt - rb + 32) / 32, Wg / 32);
}
// one thread per (sample, 64-row k-step): which 128-row tiles of the rcnt G^T rows the step's two 32-row blocks can touch
__global__ void gt_tilemask_kernel(const int16_t* relidx, const int32_t* klen, int B, int S, int Sp, int neg, int rmin, int rcnt,
                                   int Wg, uint32_t* mask) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int steps = Sp / 64;
  if (t >= B * steps) return;
  const int b = t / steps, j = t - b * steps;
  const int kl = klen ? min(klen[b], S) : S;
  uint32_t m = 0;
  for (int half = 0; half < 2; ++half) {
    const int r0 = (2 * j + half) * 32;
    if (r0 >= kl) continue;
    int rbase, nks;
    gt_window(relidx, S, kl, r0, neg != 0, Wg, &rbase, &nks);
    const int lo = max(rbase - rmin, 0), hi = min(rbase + nks * 32 - rmin, rcnt);
    for (int tt = lo / 128; tt * 128 < hi; ++tt) m |= 1u << tt;
  }
  mask[t] = m;
}
constexpr int C_IDX = 0;           // int16[1024]: relative-index table padded to the tile grid
constexpr int C_G = C_IDX + 2048;  // [32][Wg + 8] bf16

// 2 waves x 16 rows; after the initial barrier (cooperative zeroing of G, index table) the waves never synchronise:
//  * the transposed-operand fragments come straight from global memory (L2-resident: every row tile of a (batch, head)
//    re-reads them) into registers one column tile ahead;
//  * each wave scatters only into its own 16 rows of G.  A 16x64 patch that lies entirely in the identity-bucket band
//    (|i-j| < lin, idx = delta + idx(0)) needs no table lookup and no atomics: one unconditional LDS store per element
//    (zeros of padded / masked positions are diverted to the row's padding slot);
//  * the table GEMM walks only the index range the valid columns [0, klen) can reach, and G^T leaves through the
//    matrix cores: D = G_frag . I puts 4 consecutive rows of one table index in a lane (an exact transpose of the bf16
//    values), so there is no column-wise LDS read-out.
template <bool NEG, bool PACKED = false>
__global__ __launch_bounds__(128) void attn_bwd_shear_kernel(ShearArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, Sp = a.Sp, hi = 2 * S - 2;
  const WgCoord wc = wg_coord(Sp / 32, a.nh, a.B, a.border);
  const int bx = wc.x, r0 = bx * 32, h = wc.h, b = wc.b;
  const int LDG = a.Wg + 8;  // bf16 elements; rows stay 16-byte aligned, 8 padding slots per row
  const int rl = w * 16 + c;  // local row
  const int row = r0 + rl;
  bf16* G = (bf16*)(smem + C_G);
  int16_t* idx = (int16_t*)(smem + C_IDX);
  const bool wgt = a.GT != nullptr;  // G^T wanted (the per-head position-table GEMMs of rounds 1-5; fbl_attn_pos_grad reads dS itself)
  bf16* gt = a.GT + ((((long)h * a.B + b) * (Sp / 32) + bx) * a.rcnt) * 32;
  const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  const int kl = a.klen ? min(a.klen[b], S) : S;
  const long rb = PACKED ? (long)a.row0[b] : (long)b * S;                  // first output row of this sample
  const int lim = PACKED ? min(a.row0[b + 1] - a.row0[b], S) : S;          // output rows that exist
  if (r0 >= kl) {  // rows entirely beyond the sample's last valid position: dS is zero -> zero output rows, zero G^T block
    // The consumer of G^T (the position-table GEMMs) skips a 64-wide k-step whose first row is beyond kl, so this block
    // only has to exist (as zeros) when it is the odd half of a step whose even half is valid.
    if (wgt && (bx & 1) && (r0 - 32 < kl)) {
      const uint32_t tm = a.tilemask ? a.tilemask[b * (Sp / 64) + (bx >> 1)] : ~0u;
      if (!(FBL_ATTN_DBGBITS & 64))
        for (int id = tid; id < a.rcnt * 4; id += 128)
          if ((tm >> (id >> 9)) & 1) *(bf16x8*)(gt + (long)id * 8) = z8;  // (id >> 2 = row, 128 rows per tile)
    }
    if (row < lim) {
      bf16* op = a.out + (rb + row) * a.ldout + h * 64 + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *(bf16x4*)(op + dt * 16) = (bf16x4){0, 0, 0, 0};
    }
    return;
  }
  // table rows reachable from these 32 rows x the valid columns [0, kl): [rbase, rbase + nks*32); rbase aligned down to 8 so the
  // PT fragments stay 16-byte aligned.  dS is only defined (and non-zero) inside [kl x kl].
  int rbase, nks;
  gt_window(a.relidx, S, kl, r0, NEG, a.Wg, &rbase, &nks);
  (void)hi;
  const int izero = (int)a.relidx[S - 1];  // idx(0)
  {
    const int vpr = nks * 4;  // 16-byte vectors per row
    for (int t = tid; t < 32 * vpr; t += 128) *(bf16x8*)(G + (t / vpr) * LDG + (t % vpr) * 8) = z8;
  }
  attn::load_idx_padded(idx, a.relidx, S, Sp, tid, 128);
  // G^T rows outside [rbase, rbase + nks*32) are zero: written straight from here -- only inside the 128-row tiles that the
  // consumer fetches for this 64-row k-step (tilemask: the tiles either of its two blocks can touch)
  const uint32_t tmask = a.tilemask ? a.tilemask[b * (Sp / 64) + (bx >> 1)] : ~0u;
  for (int id = tid; wgt && id < a.rcnt * 4; id += 128) {
    const int r = a.rmin + (id >> 2);
    if (!(FBL_ATTN_DBGBITS & 64) && ((tmask >> (id >> 9)) & 1) && (r < rbase || r >= rbase + nks * 32)) *(bf16x8*)(gt + (long)id * 8) = z8;
  }
  const long xbase = (((long)b * a.nh + h) * Sp + row) * Sp;
  f32x4 acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int nct = (kl + 63) / 64;
  const bf16* ytb = a.YT + h * a.y_sh + b * a.y_sb + (long)c * a.y_sd + g * 8;
  const int16_t* ib = idx + (NEG ? (Sp - 1 - row + g * 8) : (Sp - 1 + row - g * 8));  // idx(delta) = ib[+-(col - g*8)]
  bf16* grow = G + rl * LDG - rbase;  // indexed by the absolute table row
  const int wmax = rbase + nks * 32 - 1;
  const int dummy = rbase + a.Wg;      // the row's padding slots: never read
  // lane-linear part of the identity-band slot: slot = izero + delta, delta = +-(row - col)
  const int lin0 = izero + (NEG ? (g * 8 - row) : (row - g * 8));
  const int rw = r0 + w * 16;
  struct Frags { bf16x8 y[8], x[2]; };
  auto load_ct = [&](int ct, Frags& F) {
    const int c0 = ct * 64;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        F.y[kk * 4 + dt] = (FBL_ATTN_DBGBITS & 2048) ? z8 : *(const bf16x8*)(ytb + (long)(dt * 16) * a.y_sd + c0 + kk * 32);
    F.x[0] = *(const bf16x8*)(a.X + xbase + c0 + g * 8);
    F.x[1] = *(const bf16x8*)(a.X + xbase + c0 + 32 + g * 8);
  };
  auto step = [&](int ct, const Frags& F) {
    const int c0 = ct * 64;
    // delta range of this wave's 16 rows x these 64 columns (wave-uniform)
    const int d_a = rw - (c0 + 63), d_b = rw + 15 - c0;  // row - col in [d_a, d_b]
    const bool band = max(abs(d_a), abs(d_b)) < a.lin;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 xv = F.x[kk];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(F.y[kk * 4 + dt], xv, acc[dt], 0, 0, 0);
      const int cb = c0 + kk * 32;  // this lane's columns: cb + g*8 + e
      if (FBL_ATTN_DBGBITS & 512) continue;  // (debug builds: no scatter)
      if (band) {
        const int s0 = NEG ? (lin0 + cb) : (lin0 - cb);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int slot = NEG ? (s0 + e) : (s0 - e);
          grow[bf2f(xv[e]) != 0.f ? slot : dummy] = xv[e];  // injective inside the band and G starts at 0: a plain store is exact
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {  // ~2/3 of dS is exactly 0 (padding / masks) and is skipped
          const float x = bf2f(xv[e]);
          if (x != 0.f) {
            const int dlt = NEG ? (cb + g * 8 + e - row) : (row - cb - g * 8 - e);
            const int gi = clampi((int)(NEG ? ib[cb + e] : ib[-(cb + e)]), rbase, wmax);
            // LDS atomics (packed bf16 add on the element's aligned pair, the other half adds 0) only where log buckets
            // can collide; a bucket collects a handful of terms, each partial sum rounded to bf16 like G itself is
            // before it meets the matrix cores
            if (abs(dlt) < a.lin) {
              grow[gi] = xv[e];
            } else {
              bf16x2 pv = {(bf16)0.f, (bf16)0.f};
              pv[gi & 1] = xv[e];
              __builtin_amdgcn_ds_atomic_fadd_v2bf16((__attribute__((address_space(3))) bf16x2*)(grow + (gi & ~1)), pv);
            }
          }
        }
      }
    }
  };
  {
    Frags F0, F1;
    load_ct(0, F0);
    for (int ct = 0; ct < nct; ct += 2) {
      if (ct + 1 < nct) load_ct(ct + 1, F1);
      step(ct, F0);
      if (ct + 2 < nct) load_ct(ct + 2, F0);
      if (ct + 1 < nct) step(ct + 1, F1);
    }
  }
  // ---- table part: acc[d][row] += sum_r PT[d][rbase + r] * G[row][r], PT fragments double-buffered from L2;
  //      G^T[r][row] leaves via two identity MFMAs per 32 table rows
  const bf16* pt = a.PT + (long)h * 64 * a.span2 + rbase + g * 8;
  bf16x8 I0, I1;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    I0[e] = f2bf((g * 8 + e) == c ? 1.f : 0.f);
    I1[e] = f2bf((g * 8 + e) == c + 16 ? 1.f : 0.f);
  }
  auto load_pt = [&](int kk, bf16x8* dstf) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bool ok = rbase + kk * 32 + g * 8 + 8 <= a.span2;
      dstf[dt] = ok ? *(const bf16x8*)(pt + (long)(dt * 16 + c) * a.span2 + kk * 32) : z8;
    }
  };
  auto table_step = [&](int kk, const bf16x8* af) {
    const bf16x8 bfv = *(const bf16x8*)(G + rl * LDG + kk * 32 + g * 8);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[dt], bfv, acc[dt], 0, 0, 0);
    if (!wgt) return;
    const f32x4 zf = {0.f, 0.f, 0.f, 0.f};
    const f32x4 t0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv, I0, zf, 0, 0, 0);  // [row g*4+j][table row kk*32 + c]
    const f32x4 t1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv, I1, zf, 0, 0, 0);  // [row g*4+j][table row kk*32+16+c]
    const int ra = rbase + kk * 32 + c - a.rmin, rb = ra + 16;
    if (ra >= 0 && ra < a.rcnt && !(FBL_ATTN_DBGBITS & 128))
      *(bf16x4*)(gt + (long)ra * 32 + w * 16 + g * 4) = (bf16x4){f2bf(t0[0]), f2bf(t0[1]), f2bf(t0[2]), f2bf(t0[3])};
    if (rb >= 0 && rb < a.rcnt && !(FBL_ATTN_DBGBITS & 128))
      *(bf16x4*)(gt + (long)rb * 32 + w * 16 + g * 4) = (bf16x4){f2bf(t1[0]), f2bf(t1[1]), f2bf(t1[2]), f2bf(t1[3])};
  };
  // (a third k-step of fragments in flight, and 64-row workgroups of four waves, were measured: no gain / slower)
  bf16x8 pa[4], pb[4];
  const int nks_run = (FBL_ATTN_DBGBITS & 1024) ? 0 : nks;
  load_pt(0, pa);
  for (int kk = 0; kk < nks_run; kk += 2) {
    if (kk + 1 < nks) load_pt(kk + 1, pb);
    table_step(kk, pa);
    if (kk + 2 < nks) load_pt(kk + 2, pa);
    if (kk + 1 < nks) table_step(kk + 1, pb);
  }
  if (row < lim) {
    bf16* op = a.out + (rb + row) * a.ldout + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(bf16x4*)(op + dt * 16) = (bf16x4){f2bf(acc[dt][0]), f2bf(acc[dt][1]), f2bf(acc[dt][2]), f2bf(acc[dt][3])};
  }
}

// ------------------------------------------------------------------------------------------- position-table gradients
// dPK[h][r][d] = sum_b sum_{(i,j): idx(i-j) = r} dS_b[i,j] Q_b[i,d]        (NEG = 0: X = dS,   Y = Q, k index = query i)
// dPQ[h][r][d] = sum_b sum_{(i,j): idx(i-j) = r} dS_b[i,j] K_b[j,d]        (NEG = 1: X = dS^T, Y = K, k index = key j)
// of EVERY layer execution in one launch (autograd of model/deberta.py:870-918 through the c2p / p2c gathers), straight from
// the dS / dS^T tensors kernel A wrote: until round 5 the shear passes wrote a sheared copy G^T of them (2 x 163 MB per layer
// execution) that two batched split-K GEMMs fetched back at the end of backward (8 GB each way per step).  A table row r
// collects the deltas [dlo[r], dlo[r] + dcnt[r]) (one inside the identity band, a few per log bucket; idx is monotone), so
// with rows of X = dS (dS^T) staged in LDS the MFMA operand G[r][k] = sum_t X[k][k -+ (dlo[r] + t)] is a diagonal walk of
// 16-bit reads -- 8 per fragment, against the 4 MFMAs it feeds.  One workgroup (4 waves) = one (execution, head, 256 table
// rows): it walks all samples and their 32-row blocks, every wave keeps the [16 x 64] results of four 16-row tiles of table
// rows in registers, and the sum over the batch never leaves them: no partial sums, no workspace, deterministic.  The two
// workgroups of an (execution, head) sit on one XCD (they read the same rows of X).
constexpr int PG_MAX_E = 64;  // layer executions per launch (the pointers travel as kernel arguments: capturable)
struct PosGradArgs {
  const bf16* X[PG_MAX_E];  // dS or dS^T of each execution: [B,nh,Sp,Sp]
  const bf16* Y[PG_MAX_E];  // q or k: rows b*S + s (or packed: row0), head h at column h*64
  long ldy;
  const int16_t* dlo;     // [rcnt] first delta of table row rmin + r
  const int16_t* dcnt;    // [rcnt] number of deltas of that row
  const int32_t* klen;    // [B] or null
  const int32_t* row0;    // [B+1] packed rows of Y or null
  float* out;             // [E][nh][rcnt][64]
  int E, B, S, Sp, nh, rcnt;
  int cmax;               // max of dcnt (host)
};
constexpr int PG_PAD = 48;   // zero columns on both sides of the staged rows: a fragment of a tile that touches the valid range
                             // reaches at most 15 + 31 columns beyond it
constexpr int PG_LDY = 80;   // bf16 row stride of the Y tile (transposing reads, see LDP)
constexpr int PG_TPW = 4;    // 16-row tiles of table rows per wave
constexpr int PG_WAVES = 4, PG_THR = PG_WAVES * 64;
constexpr int PG_ROWS = PG_WAVES * PG_TPW * 16;  // table rows per workgroup
__host__ __device__ constexpr int pg_pitch(int Sp) { return Sp + 2 * PG_PAD; }          // bf16 elements
__host__ __device__ constexpr int pg_buf(int Sp) { return 32 * pg_pitch(Sp) * 2 + 32 * PG_LDY * 2; }  // one staging buffer
__host__ __device__ constexpr int pg_smem(int Sp) { return 2 * pg_buf(Sp); }

// CM: upper bound of dcnt (deltas per table row): 3 covers S <= 300 at the DeBERTa-v2 bucket map, 8 every S <= 512
// NXR: 16-byte chunks of X per thread and block (Sp / 64); OCC: workgroups per CU the register budget is set for (3: one staging
// buffer and two barriers per item, 2: two buffers and one barrier)
template <bool NEG, int CM, int NXR, int OCC>
__global__ __launch_bounds__(PG_THR, OCC) void pos_grad_kernel(PosGradArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int S = a.S, Sp = a.Sp, P = pg_pitch(Sp);
  // block -> (problem = execution * nh + head, part): the parts of a problem share its XCD (block L runs on XCD L % 8)
  const int nsplit = (a.rcnt + PG_ROWS - 1) / PG_ROWS;
  const int L8 = blockIdx.x >> 3, xcd = blockIdx.x & 7;
  const int prob = (L8 / nsplit) * 8 + xcd, part = L8 % nsplit;
  if (prob >= a.E * a.nh) return;
  const int e = prob / a.nh, h = prob % a.nh;
  const int bufsz = pg_buf(Sp);  // two staging buffers: [32][P] rows of X (column PG_PAD + j holds X[row][j]) + [32][PG_LDY] rows of Y
  const bf16* X = a.X[e];
  const bf16* Y = a.Y[e];
  const int ntiles = (a.rcnt + 15) / 16;

  // this wave's tiles: t = (w + 4 u) * nsplit + part (the parts and the waves take the tiles round-robin: the expensive
  // log-bucket tiles at both ends of the table and the tiles a short sample touches are spread evenly).  Per tile: the lane's table row, its delta range, and the tile's (wave-uniform) delta span
  int d0[PG_TPW], dn[PG_TPW], tlo[PG_TPW], thi[PG_TPW];
  bool simple[PG_TPW];
#pragma unroll
  for (int u = 0; u < PG_TPW; ++u) {
    const int t = (w + PG_WAVES * u) * nsplit + part;
    const int r = min(t * 16 + c, a.rcnt - 1);
    const bool live = t < ntiles && t * 16 + c < a.rcnt;
    d0[u] = live ? (int)a.dlo[r] : 0;
    dn[u] = live ? (int)a.dcnt[r] : 0;
    const int rf = min(t * 16, a.rcnt - 1), rl = min(t * 16 + 15, a.rcnt - 1);
    tlo[u] = t < ntiles ? (int)a.dlo[rf] : 1 << 20;
    thi[u] = t < ntiles ? (int)a.dlo[rl] + (int)a.dcnt[rl] - 1 : -(1 << 20);
    // identity band: one delta per row, consecutive rows <-> consecutive deltas (wave-uniform)
    simple[u] = t < ntiles && t * 16 + 15 < a.rcnt && (thi[u] - tlo[u]) == 15 && __builtin_amdgcn_ballot_w64(dn[u] != 1) == 0;
  }
  f32x4 acc[PG_TPW][4];
#pragma unroll
  for (int u = 0; u < PG_TPW; ++u)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[u][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the zero columns on both sides are written once (both buffers); the staging rewrites columns [0, Sp) of every row
  for (int id = tid; id < (OCC >= 3 ? 1 : 2) * 32 * 2 * (PG_PAD / 8); id += PG_THR) {
    const int bsel = id / (32 * 2 * (PG_PAD / 8)), r2 = id % (32 * 2 * (PG_PAD / 8));
    const int row = r2 / (2 * (PG_PAD / 8)), q = r2 % (2 * (PG_PAD / 8));
    const int col = q < PG_PAD / 8 ? q * 8 : PG_PAD + Sp + (q - PG_PAD / 8) * 8;
    *(bf16x8*)((bf16*)(smem + bsel * bufsz) + row * P + col) = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  }
  const bf16x8 z8 = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  // staging role of this thread: row tid / 8 of the block, 16-byte chunks (tid % 8) + 8 q  (Sp is a multiple of 64: nq = Sp / 64
  // chunks per thread; eight consecutive threads fetch 128 contiguous bytes) -- no integer division in the loop
  const int srow = tid >> 3, sch = tid & 7, nq = Sp >> 6;

  // work list: (sample b, 32-row block k0 < klen[b]) in order; the next item's operands are in flight (registers) during the
  // current item's arithmetic
  int b_n = 0, k_n = 0, kl_n = 0;
  auto seek = [&]() {  // first item at or after (b_n, k_n)
    while (b_n < a.B) {
      kl_n = a.klen ? min(a.klen[b_n], S) : S;
      if (k_n < kl_n) return true;
      ++b_n; k_n = 0;
    }
    return false;
  };
  bf16x8 rx[NXR], ry;
  int it_b = 0, it_k = 0, it_kl = 0;
  auto load_item = [&]() {  // loads item (b_n, k_n) and advances the iterator
    it_b = b_n; it_k = k_n; it_kl = kl_n;
    const int kl64 = (kl_n + 63) & ~63;  // kernel A wrote the tile pairs below it; anything beyond is unwritten memory
    const long rb = a.row0 ? (long)a.row0[b_n] : (long)b_n * S;
    const int lim = a.row0 ? min(a.row0[b_n + 1] - a.row0[b_n], S) : S;
    const bf16* Xb = X + (((long)b_n * a.nh + h) * Sp) * Sp + (long)k_n * Sp;
    const bf16* xr = Xb + (long)srow * Sp + sch * 8;
#pragma unroll
    for (int q = 0; q < NXR; ++q) rx[q] = (q < nq && sch * 8 + q * 64 < kl64) ? *(const bf16x8*)(xr + q * 64) : z8;
    {
      const int pos = k_n + srow;
      ry = (pos < lim) ? *(const bf16x8*)(Y + (rb + pos) * a.ldy + h * 64 + sch * 8) : z8;
    }
    k_n += 32;
  };
  int pb = 0;
  bool have = seek();
  if (have) load_item();
  while (have) {
    bf16* xs = (bf16*)(smem + pb * bufsz);
    bf16* ys = xs + 32 * P;
    if (OCC >= 3) __syncthreads();  // single buffer: the previous item's fragments are read
    {
      bf16* xw = xs + srow * P + PG_PAD + sch * 8;
#pragma unroll
      for (int q = 0; q < NXR; ++q)
        if (q < nq) *(bf16x8*)(xw + q * 64) = rx[q];
    }
    *(bf16x8*)(ys + srow * PG_LDY + sch * 8) = ry;
    const int k0 = it_k, kl = it_kl;
    __syncthreads();  // one barrier per item: the buffer written two items from now was read one item ago, before this barrier
    have = seek();
    if (have) load_item();
    // Y^T fragments (MFMA A operand: rows d = dt*16 + c, k = the 32 staged rows): transposing reads of the row-major tile
    bf16x8 yf[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16* yb = ys + (g * 8 + (c >> 2)) * PG_LDY + dt * 16 + (c & 3) * 4;
      union { tr16x4 hh[2]; bf16x8 v; } u2;
      u2.hh[0] = lds_tr16(yb);
      u2.hh[1] = lds_tr16(yb + 4 * PG_LDY);
      yf[dt] = u2.v;
    }
#pragma unroll
    for (int u = 0; u < PG_TPW; ++u) {
      // columns this tile reads in this block: k -+ delta for k in [k0, k0+32), delta in [tlo, thi]; valid columns [0, kl)
      const int cmin = NEG ? k0 + tlo[u] : k0 - thi[u];
      const int cmax = NEG ? k0 + 31 + thi[u] : k0 + 31 - tlo[u];
      if (cmax < 0 || cmin >= kl) continue;  // (wave-uniform)
      bf16x8 gf;
      // G[r][k] = sum_t X[k][k -+ (d0 + t)]: lane (table row c, k chunk g) walks diagonals of the staged rows
      const bf16* xrow = xs + (g * 8) * P + PG_PAD;
      const int cb = NEG ? (k0 + g * 8 + d0[u]) : (k0 + g * 8 - d0[u]);  // column of row g*8 (q = 0, t = 0)
      if (simple[u]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) gf[q] = xrow[q * P + cb + q];
      } else {
        // log buckets (a few deltas per row, <= a.cmax) and edge tiles: columns are clamped into the zero margins
        // (all 8 x CM reads are issued before the first one is used: a read-use-read chain costs an LDS latency per element)
        const int lo = -PG_PAD, hi = Sp + PG_PAD - 1;
        bf16 vv[8][CM];
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
          for (int t2 = 0; t2 < CM; ++t2) vv[q][t2] = xrow[q * P + clampi(NEG ? cb + q + t2 : cb + q - t2, lo, hi)];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          float sum = 0.f;
#pragma unroll
          for (int t2 = 0; t2 < CM; ++t2) sum += t2 < dn[u] ? bf2f(vv[q][t2]) : 0.f;
          gf[q] = f2bf(sum);
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[dt], gf, acc[u][dt], 0, 0, 0);
    }
    if (OCC < 3) pb ^= 1;
  }
  // lane: table row t*16 + c, columns d = dt*16 + g*4 .. +3
  float* ob = a.out + ((long)e * a.nh + h) * a.rcnt * 64;
#pragma unroll
  for (int u = 0; u < PG_TPW; ++u) {
    const int r = ((w + PG_WAVES * u) * nsplit + part) * 16 + c;
    if (r < a.rcnt) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) *(f32x4*)(ob + (long)r * 64 + dt * 16 + g * 4) = acc[u][dt];
    }
  }
}

}  // namespace

static inline int shear_wg(int S, int span2) {  // columns of the shear pass's G tile (a multiple of 32)
  int Wg = S + 31 + 7;
  if (Wg > span2) Wg = span2;
  return (Wg + 31) / 32 * 32;
}

extern "C" int fbl_attn_rowdot(const void* dO, const void* O, int64_t ld, float* out, int B, int S, int nh,
                               void* stream) {
  if (ld % 8) return FBL_ERR_ALIGN;
  const long total = (long)B * S * nh;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((total * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dO, (const bf16*)O, (long)ld, out, B, S, nh);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_disent_attn_bwd_ds(const void* q, const void* k, const void* v, int64_t ldq, const void* dO,
                                      int64_t ldo,
                                      const void* pk, const void* pq, int64_t ldp, const int16_t* relidx,
                                      const int32_t* mask, const int32_t* klen, const int32_t* border, const float* lse, const float* Dv, float scale, float p_drop,
                                      uint64_t seed, const uint64_t* seed_dev, void* dV, int64_t lddv, void* dS, void* dST, int B, int S, int Sp,
                                      int nh, int span2, int lin_span, const int32_t* row0, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64) return FBL_ERR_SHAPE;
  if ((ldq % 8) || (ldo % 8) || (ldp % 8) || (lddv % 4)) return FBL_ERR_ALIGN;
  if (lin_span < 0 || 2 * lin_span > span2) return FBL_ERR_ARG;
  if (row0 && !klen) return FBL_ERR_ARG;
  if (B <= 0 || nh <= 0) return 0;
  BwdAArgs a{(const bf16*)q, (const bf16*)k, (const bf16*)v, ldq, (const bf16*)dO, ldo, (const bf16*)pk, (const bf16*)pq, ldp, relidx, mask, klen, border, lse, Dv, scale, p_drop, seed, (bf16*)dV, lddv,
             (bf16*)dS, (bf16*)dST, B, S, Sp, nh, span2, lin_span, seed_dev, row0};
  attn_debug_init();
  const int smem_bytes = a_total(Sp);
  static int attr_bytes = 0;
  if (smem_bytes > attr_bytes) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_ds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_ds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_ds_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_ds_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e != hipSuccess) return (int)e;
    attr_bytes = smem_bytes;
  }
  const dim3 grid((unsigned)(Sp / 64) * nh * B);
  const bool devseed = seed_dev && p_drop > 0.f;
  if (row0 && devseed)
    hipLaunchKernelGGL((attn_bwd_ds_kernel<true, true>), grid, dim3(256), smem_bytes, (hipStream_t)stream, a);
  else if (row0)
    hipLaunchKernelGGL((attn_bwd_ds_kernel<false, true>), grid, dim3(256), smem_bytes, (hipStream_t)stream, a);
  else if (devseed)
    hipLaunchKernelGGL(attn_bwd_ds_kernel<true>, grid, dim3(256), smem_bytes, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_bwd_ds_kernel<false>, grid, dim3(256), smem_bytes, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_disent_attn_bwd_dsp(const void* psave, const float* msave, const void* v, int64_t ldv, const void* dO,
                                       int64_t ldo, const int32_t* klen, const int32_t* border, const float* lse, const float* Dv,
                                       float scale, float p_drop, uint64_t seed, const uint64_t* seed_dev, void* dV, int64_t lddv,
                                       void* dS, void* dST, int B, int S, int Sp, int nh, const int32_t* row0, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64) return FBL_ERR_SHAPE;
  if ((ldv % 8) || (ldo % 8) || (lddv % 4)) return FBL_ERR_ALIGN;
  if (!psave || !msave || !v || !dO || !lse || !Dv || !dV || !dS || !dST) return FBL_ERR_ARG;
  if (row0 && !klen) return FBL_ERR_ARG;
  if (B <= 0 || nh <= 0) return 0;
  BwdPArgs a{(const bf16*)psave, msave, (const bf16*)v, ldv, (const bf16*)dO, ldo, klen, border, lse, Dv, scale, p_drop, seed,
             seed_dev, (bf16*)dV, lddv, (bf16*)dS, (bf16*)dST, B, S, Sp, nh, row0};
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dsp_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, P_TOTAL);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dsp_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, P_TOTAL);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const dim3 grid((unsigned)(Sp / 64) * nh * B);
  if (row0)
    hipLaunchKernelGGL(attn_bwd_dsp_kernel<true>, grid, dim3(256), P_TOTAL, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_bwd_dsp_kernel<false>, grid, dim3(256), P_TOTAL, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_disent_attn_bwd_dspk(const void* psave, const float* msave, const void* q, int64_t ldq, const void* v, int64_t ldv,
                                        const void* dO, int64_t ldo, const void* pqx, const int32_t* klen, const int32_t* border,
                                        const float* lse, const float* Dv, float scale, float p_drop, uint64_t seed,
                                        const uint64_t* seed_dev, void* dK, int64_t lddk, void* dV, int64_t lddv, void* dS, void* dST,
                                        int B, int S, int Sp, int nh, const int32_t* row0, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64) return FBL_ERR_SHAPE;
  if ((ldv % 8) || (ldo % 8) || (ldq % 8) || (lddv % 4) || (lddk % 4)) return FBL_ERR_ALIGN;
  if (!psave || !msave || !q || !v || !dO || !pqx || !lse || !Dv || !dK || !dV || !dS || !dST) return FBL_ERR_ARG;
  if (row0 && !klen) return FBL_ERR_ARG;
  if (B <= 0 || nh <= 0) return 0;
  BwdPKArgs a{{(const bf16*)psave, msave, (const bf16*)v, ldv, (const bf16*)dO, ldo, klen, border, lse, Dv, scale, p_drop, seed,
               seed_dev, (bf16*)dV, lddv, (bf16*)dS, (bf16*)dST, B, S, Sp, nh, row0},
              (const bf16*)q, ldq, (const bf16*)pqx, (bf16*)dK, lddk};
  attn_debug_init();
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dspk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, K_TOTAL);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dspk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, K_TOTAL);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const dim3 grid((unsigned)(Sp / 64) * nh * B);
  if (row0)
    hipLaunchKernelGGL(attn_bwd_dspk_kernel<true>, grid, dim3(256), K_TOTAL, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_bwd_dspk_kernel<false>, grid, dim3(256), K_TOTAL, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_disent_attn_bwd_dq(const void* dS, const void* k, int64_t ldk, const void* pkx, const int32_t* klen,
                                      const int32_t* border, void* dQ, int64_t lddq, int B, int S, int Sp, int nh,
                                      const int32_t* row0, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64) return FBL_ERR_SHAPE;
  if ((ldk % 8) || (lddq % 4)) return FBL_ERR_ALIGN;
  if (!dS || !k || !pkx || !dQ) return FBL_ERR_ARG;
  if (row0 && !klen) return FBL_ERR_ARG;
  if (B <= 0 || nh <= 0) return 0;
  BwdQArgs a{(const bf16*)dS, (const bf16*)k, ldk, (const bf16*)pkx, klen, border, (bf16*)dQ, lddq, B, S, Sp, nh, row0};
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_TOTAL);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_TOTAL);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const dim3 grid((unsigned)(Sp / 64) * nh * B);
  if (row0)
    hipLaunchKernelGGL(attn_bwd_dq_kernel<true>, grid, dim3(256), Q_TOTAL, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_bwd_dq_kernel<false>, grid, dim3(256), Q_TOTAL, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_disent_attn_bwd_shear(int neg, const void* X, const void* YT, int64_t y_sh, int64_t y_sb,
                                         int64_t y_sd, const void* PT, const int16_t* relidx, const int32_t* klen,
                                         const int32_t* border, void* out, int64_t ldout, void* GT, int gt_rmin, int gt_rcnt, int lin_span, int B, int S,
                                         int Sp, int nh, int span2, const int32_t* row0, const uint32_t* gt_tilemask, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64 || span2 > 512 || span2 % 32) return FBL_ERR_SHAPE;
  if ((ldout % 4) || (y_sd % 8) || (y_sb % 8) || (y_sh % 8)) return FBL_ERR_ALIGN;
  if (B <= 0 || nh <= 0) return 0;
  // index range reachable from 32 consecutive rows: <= S + 31 entries (idx has slope <= 1), +7 for the 8-alignment
  const int Wg = shear_wg(S, span2);
  if (GT && (gt_rmin < 0 || gt_rcnt < 0 || gt_rmin + gt_rcnt > span2)) return FBL_ERR_ARG;
  if (row0 && !klen) return FBL_ERR_ARG;
  ShearArgs a{(const bf16*)X, (const bf16*)YT, y_sh, y_sb, y_sd, (const bf16*)PT, relidx, klen, border, (bf16*)out, ldout, (bf16*)GT,
              B, S, Sp, nh, span2, Wg, gt_rmin, gt_rcnt, lin_span, row0, gt_tilemask};
  attn_debug_init();
  const int smem_bytes = C_G + 32 * (Wg + 8) * 2;
  static int attr_bytes = 0;
  if (smem_bytes > attr_bytes) {
    hipError_t e1 = hipFuncSetAttribute((const void*)attn_bwd_shear_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    hipError_t e2 = hipFuncSetAttribute((const void*)attn_bwd_shear_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e1 != hipSuccess) return (int)e1;
    if (e2 != hipSuccess) return (int)e2;
    e1 = hipFuncSetAttribute((const void*)attn_bwd_shear_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    e2 = hipFuncSetAttribute((const void*)attn_bwd_shear_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
    if (e1 != hipSuccess) return (int)e1;
    if (e2 != hipSuccess) return (int)e2;
    attr_bytes = smem_bytes;
  }
  dim3 grid((unsigned)(Sp / 32) * nh * B);
  if (neg && row0)
    hipLaunchKernelGGL((attn_bwd_shear_kernel<true, true>), grid, dim3(128), smem_bytes, (hipStream_t)stream, a);
  else if (row0)
    hipLaunchKernelGGL((attn_bwd_shear_kernel<false, true>), grid, dim3(128), smem_bytes, (hipStream_t)stream, a);
  else if (neg)
    hipLaunchKernelGGL(attn_bwd_shear_kernel<true>, grid, dim3(128), smem_bytes, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_bwd_shear_kernel<false>, grid, dim3(128), smem_bytes, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_attn_pos_grad(int neg, const void* const* X, const void* const* Y, int64_t ldy, const int16_t* dlo,
                                 const int16_t* dcnt, int dcnt_max, const int32_t* klen, const int32_t* row0, float* out, int E,
                                 int B, int S, int Sp, int nh, int rcnt, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64 || rcnt < 1 || rcnt > 1024) return FBL_ERR_SHAPE;
  if (ldy % 8) return FBL_ERR_ALIGN;
  if (!X || !Y || !dlo || !dcnt || !out || dcnt_max < 1 || dcnt_max > 8) return FBL_ERR_ARG;
  if (row0 && !klen) return FBL_ERR_ARG;
  if (E <= 0 || B <= 0 || nh <= 0) return 0;
  for (int e = 0; e < E; ++e)
    if (!X[e] || !Y[e]) return FBL_ERR_ARG;
  // three workgroups per CU (one staging buffer) where the rows are short enough for five chunks per thread, two otherwise
  static const int occ_sw = FBL_ENV_INT("FBL_POSGRAD_OCC", 3);
  const bool occ3 = Sp <= 320 && dcnt_max <= 3 && occ_sw >= 3;
  const int smem_bytes = occ3 ? pg_buf(Sp) : pg_smem(Sp);
  static int attr_bytes = 0;
  if (pg_smem(Sp) > attr_bytes) {
    const void* fns[6] = {(const void*)pos_grad_kernel<false, 3, 5, 3>, (const void*)pos_grad_kernel<true, 3, 5, 3>,
                          (const void*)pos_grad_kernel<false, 3, 8, 2>, (const void*)pos_grad_kernel<true, 3, 8, 2>,
                          (const void*)pos_grad_kernel<false, 8, 8, 2>, (const void*)pos_grad_kernel<true, 8, 8, 2>};
    for (int i = 0; i < 6; ++i) {
      hipError_t e1 = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, pg_smem(Sp));
      if (e1 != hipSuccess) return (int)e1;
    }
    attr_bytes = pg_smem(Sp);
  }
  const int nsplit = (rcnt + PG_ROWS - 1) / PG_ROWS;
  for (int e0 = 0; e0 < E; e0 += PG_MAX_E) {
    PosGradArgs a{};
    const int ne = E - e0 < PG_MAX_E ? E - e0 : PG_MAX_E;
    for (int e = 0; e < ne; ++e) {
      a.X[e] = (const bf16*)X[e0 + e];
      a.Y[e] = (const bf16*)Y[e0 + e];
    }
    a.ldy = ldy; a.dlo = dlo; a.dcnt = dcnt; a.klen = klen; a.row0 = row0;
    a.out = out + (long)e0 * nh * rcnt * 64;
    a.E = ne; a.B = B; a.S = S; a.Sp = Sp; a.nh = nh; a.rcnt = rcnt; a.cmax = dcnt_max;
    const dim3 grid((unsigned)(((ne * nh + 7) / 8) * 8 * nsplit));
#define FBL_PG_LAUNCH(NEG_, CM_, NXR_, OCC_) \
  hipLaunchKernelGGL((pos_grad_kernel<NEG_, CM_, NXR_, OCC_>), grid, dim3(PG_THR), smem_bytes, (hipStream_t)stream, a)
    if (occ3) {
      if (neg) FBL_PG_LAUNCH(true, 3, 5, 3);
      else FBL_PG_LAUNCH(false, 3, 5, 3);
    } else if (dcnt_max <= 3) {
      if (neg) FBL_PG_LAUNCH(true, 3, 8, 2);
      else FBL_PG_LAUNCH(false, 3, 8, 2);
    } else {
      if (neg) FBL_PG_LAUNCH(true, 8, 8, 2);
      else FBL_PG_LAUNCH(false, 8, 8, 2);
    }
#undef FBL_PG_LAUNCH
    FBL_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int fbl_gt_tilemask(const int16_t* relidx, const int32_t* klen, int B, int S, int Sp, int span2, int neg, int gt_rmin,
                               int gt_rcnt, uint32_t* mask, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64 || span2 > 512 || span2 % 32 || !relidx || !mask) return FBL_ERR_SHAPE;
  if (gt_rmin < 0 || gt_rcnt < 0 || gt_rmin + gt_rcnt > span2 || gt_rcnt > 32 * 128) return FBL_ERR_ARG;
  if (B <= 0) return 0;
  const int n = B * (Sp / 64);
  hipLaunchKernelGGL(gt_tilemask_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, relidx, klen, B, S, Sp, neg, gt_rmin,
                     gt_rcnt, shear_wg(S, span2), mask);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_attn_bwd_prep(const void* q, const void* k, int64_t ldq, const void* pq, const void* pk, int64_t ldp,
                                 const void* dO, const void* O, int64_t ldo, void* QT, void* KT, void* PQT, void* PKT,
                                 float* Dv, const int16_t* relidx, void* PQX, void* PKX, int B, int S, int Sp, int nh, int span2,
                                 const int32_t* row0, void* stream) {
  if (S < 1 || Sp < S || Sp % 64 || span2 <= 0 || span2 % 64) return FBL_ERR_SHAPE;
  if ((ldq % 8) || (ldp % 8) || (ldo % 8)) return FBL_ERR_ALIGN;
  if ((PQX || PKX) && !relidx) return FBL_ERR_ARG;
  if (!Dv) return FBL_ERR_ARG;
  if (B <= 0 || nh <= 0) return 0;
  const int n_tr = (Sp / 64) * nh * B, n_tab = (span2 / 64) * nh, n_x = (2 * Sp / 64) * nh;
  PrepArgs a{(const bf16*)q, (const bf16*)k, ldq, (const bf16*)pq, (const bf16*)pk, ldp, (const bf16*)dO, (const bf16*)O, ldo,
             (bf16*)QT, (bf16*)KT, (bf16*)PQT, (bf16*)PKT, Dv, B, S, Sp, nh, span2, KT ? n_tr : 0, QT ? n_tr : 0, PKT ? n_tab : 0,
             PQT ? n_tab : 0, row0, relidx, (bf16*)PQX, (bf16*)PKX, PQX ? n_x : 0, PKX ? n_x : 0};
  const long n_dot = ((long)B * S * nh * 8 + 255) / 256;
  const long grid = (long)a.n_kt + a.n_qt + a.n_pkt + a.n_pqt + a.n_pqx + a.n_pkx + n_dot;
  hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}
