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
// Kernel A  (attn_bwd_ds):  one workgroup per (b, h, 64-key tile), sweeps the query tiles; recomputes P exactly like
//            the forward (windowed T1/T2 bias GEMMs + LDS gather) but with the KEYS as lane columns, so dV accumulates
//            in registers; writes dS and dS^T (bf16, zero where masked) -- 2 x SxS bf16 per head is the only extra HBM.
// Kernel BC (attn_bwd_shear<NEG>): one workgroup per (b, h, 32 rows): X_out = dSx.Y + G.Ptab with the scatter
//            G[row, idx(+-(row-col))] += dSx[row,col] done by LDS atomics into a [32 x 512] fp32 tile; also writes G^T.
#include "fbl_common.h"
#include "../../include/fbl.h"

namespace {

constexpr int LDT = 132;  // fp32 row stride of T1/T2 windows
constexpr int LDX = 72;   // bf16 row stride of transposed-operand tiles ([64 d][64 + 8])

__device__ __forceinline__ bf16x8 lds_frag(const char* base, int row, int chunk) {
  return *(const bf16x8*)(base + row * 128 + ((chunk ^ (row & 7)) << 4));
}
__device__ __forceinline__ void lds_put(char* base, int row, int chunk, bf16x8 v) {
  *(bf16x8*)(base + row * 128 + ((chunk ^ (row & 7)) << 4)) = v;
}

// ------------------------------------------------------------------------------------------- D_i = dO_i . O_i
__global__ void rowdot_kernel(const bf16* dO, const bf16* O, long ld, float* out, int B, int S, int nh) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;  // (row, head)
  const long total = (long)B * S * nh;
  if (idx >= total) return;
  const long row = idx / nh;
  const int h = (int)(idx % nh);
  const bf16* a = dO + row * ld + h * 64;
  const bf16* b = O + row * ld + h * 64;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const bf16x8 x = *(const bf16x8*)(a + c * 8);
    const bf16x8 y = *(const bf16x8*)(b + c * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += bf2f(x[e]) * bf2f(y[e]);
  }
  const int bb = (int)(row / S), ss = (int)(row % S);
  out[((long)bb * nh + h) * S + ss] = s;
}

// ------------------------------------------------------------------------------------------- kernel A
struct BwdAArgs {
  const bf16* q; const bf16* k; const bf16* v; long ldq;  // row-major [B*S, ld], head h at col h*64
  const bf16* dO; long ldo;                               // row-major [B*S, ldo]
  const bf16* dOT; long t_sh, t_sb, t_sd;                 // transposed dO: index h*sh + b*sb + d*sd + s
  const bf16* pk; const bf16* pq; long ldp;
  const int16_t* relidx; const int32_t* mask;
  const float* lse; const float* Dv;                      // [B,nh,S]
  float scale, p_drop; uint64_t seed;
  bf16* dV; long lddv;                                    // row-major out, head h at col h*64
  bf16* dS; bf16* dST;                                    // [B,nh,Sp,Sp]
  int B, S, Sp, nh, span2;
};

constexpr int A_QS = 0;                           // [64 i][64] swz
constexpr int A_DOS = A_QS + 8192;                // [64 i][64] swz
constexpr int A_DOT = A_DOS + 8192;               // [64 d][72]
constexpr int A_PK = A_DOT + 64 * LDX * 2;        // [128][64] swz
constexpr int A_PQ = A_PK + 16384;
constexpr int A_T1 = A_PQ + 16384;                // [64 i][LDT] fp32 (shared)
constexpr int A_T2 = A_T1 + 64 * LDT * 4;         // [4][16][LDT] fp32 (wave private, per key)
constexpr int A_DST = A_T2 + 64 * LDT * 4;        // [64 i][72] bf16 staging of the dS tile
constexpr int A_IDX = A_DST + 64 * LDX * 2;       // int16[1024]
constexpr int A_ROW = A_IDX + 2048;               // float lse[64], D[64], qvalid[64]
constexpr int A_TOTAL = A_ROW + 768;

__global__ __launch_bounds__(256) void attn_bwd_ds_kernel(BwdAArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int j0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int S = a.S, Sp = a.Sp;
  const int j = j0 + w * 16 + c;  // this lane's key
  const int jc = min(j, S - 1);

  int16_t* idx = (int16_t*)(smem + A_IDX);
  float* T1 = (float*)(smem + A_T1);
  float* T2w = (float*)(smem + A_T2) + w * 16 * LDT;
  float* rlse = (float*)(smem + A_ROW);
  float* rD = rlse + 64;
  float* rqv = rD + 64;
  bf16* dst = (bf16*)(smem + A_DST);

  for (int t = tid; t < 2 * S - 1; t += 256) idx[t] = a.relidx[t];

  bf16x8 kf[2], vf[2];
  {
    const long off = ((long)b * S + jc) * a.ldq + h * 64 + g * 8;
    kf[0] = *(const bf16x8*)(a.k + off);
    kf[1] = *(const bf16x8*)(a.k + off + 32);
    vf[0] = *(const bf16x8*)(a.v + off);
    vf[1] = *(const bf16x8*)(a.v + off + 32);
  }
  const float kvalid = (j < S && a.mask[(long)b * S + jc] != 0) ? 1.f : 0.f;

  f32x4 dv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint32_t thr = fbl_drop_thresh(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.f / (1.f - a.p_drop) : 1.f;
  const long sbase = ((long)b * a.nh + h) * Sp * Sp;
  const int nqt = Sp / 64;
  __syncthreads();

  for (int it = 0; it < nqt; ++it) {
    const int i0 = it * 64;
    const int dmin = min(max(i0 - (j0 + 63) + S - 1, 0), 2 * S - 2);
    const int r_lo = idx[dmin];
    // ---- stage Q tile, dO tile (row-major, swizzled), dO^T tile, PK/PQ windows, per-query scalars
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int id = tid + t * 256;
      const int row = id >> 3, ch = id & 7;
      const int i = min(i0 + row, S - 1);
      lds_put(smem + A_QS, row, ch, *(const bf16x8*)(a.q + ((long)b * S + i) * a.ldq + h * 64 + ch * 8));
      lds_put(smem + A_DOS, row, ch, *(const bf16x8*)(a.dO + ((long)b * S + i) * a.ldo + h * 64 + ch * 8));
      *(bf16x8*)(smem + A_DOT + row * (LDX * 2) + ch * 16) =
          *(const bf16x8*)(a.dOT + h * a.t_sh + b * a.t_sb + row * a.t_sd + i0 + ch * 8);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int id = tid + t * 256;
      const int row = id >> 3, ch = id & 7;
      const int r = min(r_lo + row, a.span2 - 1);
      const long off = (long)r * a.ldp + h * 64 + ch * 8;
      lds_put(smem + A_PK, row, ch, *(const bf16x8*)(a.pk + off));
      lds_put(smem + A_PQ, row, ch, *(const bf16x8*)(a.pq + off));
    }
    if (tid < 64) {
      const int i = i0 + tid;
      const bool ok = i < S;
      const long o = ((long)b * a.nh + h) * S + min(i, S - 1);
      rlse[tid] = ok ? a.lse[o] : INFINITY;
      rD[tid] = ok ? a.Dv[o] : 0.f;
      rqv[tid] = (ok && a.mask[(long)b * S + min(i, S - 1)] != 0) ? 1.f : 0.f;
    }
    __syncthreads();

    // ---- (1) scores: sacc[nt][r] = Q_i . K_j,  i = i0 + nt*16 + g*4 + r, key column c
    f32x4 sacc[4], dpacc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + A_QS, nt * 16 + c, g), kf[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + A_QS, nt * 16 + c, 4 + g), kf[1], acc, 0, 0, 0);
      sacc[nt] = acc;
      f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + A_DOS, nt * 16 + c, g), vf[0], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + A_DOS, nt * 16 + c, 4 + g), vf[1], acc2, 0, 0, 0);
      dpacc[nt] = acc2;
    }
    // ---- (2) T2 (per key, wave private) and T1 (per query, shared; this wave does queries 16w..16w+15)
    {
      const bf16x8 qb0 = lds_frag(smem + A_QS, w * 16 + c, g);
      const bf16x8 qb1 = lds_frag(smem + A_QS, w * 16 + c, 4 + g);
#pragma unroll
      for (int wt = 0; wt < 8; ++wt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + A_PQ, wt * 16 + c, g), kf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + A_PQ, wt * 16 + c, 4 + g), kf[1], acc, 0, 0, 0);
        *(f32x4*)(T2w + c * LDT + wt * 16 + g * 4) = acc;
        f32x4 acc2 = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + A_PK, wt * 16 + c, g), qb0, acc2, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_frag(smem + A_PK, wt * 16 + c, 4 + g), qb1, acc2, 0, 0, 0);
        *(f32x4*)(T1 + (w * 16 + c) * LDT + wt * 16 + g * 4) = acc2;
      }
    }
    __syncthreads();

    // ---- (3) P, dP, dS
    float p[16], ds[16];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = nt * 16 + g * 4 + r;
        const int i = i0 + il;
        const int di = min(max(i - j + S - 1, 0), 2 * S - 2);
        const int wi = min(max((int)idx[di] - r_lo, 0), 127);
        const float s = (sacc[nt][r] + T1[il * LDT + wi] + T2w[c * LDT + wi]) * a.scale;
        float pv = (kvalid * rqv[il] != 0.f) ? __expf(s - rlse[il]) : 0.f;
        float keep = 1.f;
        if (a.p_drop > 0.f)
          keep = fbl_dropout_scale(a.seed, (((uint64_t)b * a.nh + h) * S + (uint64_t)min(i, S - 1)) * S + (uint64_t)jc, thr,
                                   inv_keep);
        ds[nt * 4 + r] = pv * (dpacc[nt][r] * keep - rD[il]) * a.scale;
        p[nt * 4 + r] = pv * keep;
      }
    }
    // ---- (4) dV^T += dO^T . drop(P):  k-slot e of step kk <-> query kk*32 + (e>>2)*16 + g*4 + (e&3)
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
        const char* row = smem + A_DOT + (dt * 16 + c) * (LDX * 2) + (kk * 32 + g * 4) * 2;
        const bf16x4 v0 = *(const bf16x4*)row;
        const bf16x4 v1 = *(const bf16x4*)(row + 32);
        bf16x8 af;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          af[e] = v0[e];
          af[4 + e] = v1[e];
        }
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, pf, dv[dt], 0, 0, 0);
      }
    }
    // ---- (5) dS^T[j][i0 + ..] straight from registers (4 consecutive queries = 8 bytes); dS via an LDS transpose
    if (j < Sp) {
      bf16* o = a.dST + sbase + (long)j * Sp + i0 + g * 4;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        *(bf16x4*)(o + nt * 16) = (bf16x4){f2bf(ds[nt * 4]), f2bf(ds[nt * 4 + 1]), f2bf(ds[nt * 4 + 2]), f2bf(ds[nt * 4 + 3])};
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(nt * 16 + g * 4 + r) * LDX + w * 16 + c] = f2bf(ds[nt * 4 + r]);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int id = tid + t * 256;
      const int row = id >> 3, ch = id & 7;
      *(bf16x8*)(a.dS + sbase + (long)(i0 + row) * Sp + j0 + ch * 8) = *(const bf16x8*)(dst + row * LDX + ch * 8);
    }
    __syncthreads();
  }

  if (j < S) {
    bf16* op = a.dV + ((long)b * S + j) * a.lddv + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(bf16x4*)(op + dt * 16) = (bf16x4){f2bf(dv[dt][0]), f2bf(dv[dt][1]), f2bf(dv[dt][2]), f2bf(dv[dt][3])};
  }
}

// ------------------------------------------------------------------------------------------- kernel BC
struct ShearArgs {
  const bf16* X;                     // dS (NEG=0) or dS^T (NEG=1): [B,nh,Sp,Sp], rows = output rows
  const bf16* YT; long y_sh, y_sb, y_sd;   // transposed K (NEG=0) / Q (NEG=1): index h*sh + b*sb + d*sd + s
  const bf16* PT;                    // transposed position table [nh][64][span2]
  const int16_t* relidx;
  bf16* out; long ldout;             // row-major, head h at col h*64
  bf16* GT;                          // [nh][span2][B][Sp]
  int B, S, Sp, nh, span2;
};
constexpr int LDG = 516;  // fp32 row stride of the G tile (span2 = 512 max)
constexpr int C_G = 0;                       // [32][LDG] fp32
constexpr int C_YT = C_G + 32 * LDG * 4;     // [64 d][72] bf16
constexpr int C_IDX = C_YT + 64 * LDX * 2;   // int16[1024]
constexpr int C_TOTAL = C_IDX + 2048;

template <bool NEG>
__global__ __launch_bounds__(128) void attn_bwd_shear_kernel(ShearArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int r0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const int S = a.S, Sp = a.Sp;
  const int rl = w * 16 + c;  // local row
  const int row = r0 + rl;
  float* G = (float*)(smem + C_G);
  int16_t* idx = (int16_t*)(smem + C_IDX);
  for (int t = tid; t < 32 * LDG; t += 128) G[t] = 0.f;
  for (int t = tid; t < 2 * S - 1; t += 128) idx[t] = a.relidx[t];
  const long xbase = (((long)b * a.nh + h) * Sp + row) * Sp;
  f32x4 acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  const int nct = Sp / 64;
  for (int ct = 0; ct < nct; ++ct) {
    const int c0 = ct * 64;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int id = tid + t * 128;
      const int d = id >> 3, ch = id & 7;
      *(bf16x8*)(smem + C_YT + d * (LDX * 2) + ch * 16) =
          *(const bf16x8*)(a.YT + h * a.y_sh + b * a.y_sb + d * a.y_sd + c0 + ch * 8);
    }
    bf16x8 xb[2];
    xb[0] = *(const bf16x8*)(a.X + xbase + c0 + g * 8);
    xb[1] = *(const bf16x8*)(a.X + xbase + c0 + 32 + g * 8);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 af = *(const bf16x8*)(smem + C_YT + (dt * 16 + c) * (LDX * 2) + (kk * 32 + g * 8) * 2);
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, xb[kk], acc[dt], 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = bf2f(xb[kk][e]);
        if (x != 0.f) {
          const int col = c0 + kk * 32 + g * 8 + e;
          const int dlt = NEG ? (col - row) : (row - col);
          const int di = min(max(dlt + S - 1, 0), 2 * S - 2);
          atomicAdd(&G[rl * LDG + idx[di]], x);
        }
      }
    }
    __syncthreads();
  }
  // ---- table part: acc[d][row] += sum_r PT[d][r] * G[row][r]
  const int nks = a.span2 / 32;
  const bf16* pt = a.PT + (long)h * 64 * a.span2;
  for (int kk = 0; kk < nks; ++kk) {
    const float* gp = G + rl * LDG + kk * 32 + g * 8;
    const f32x4 g0 = *(const f32x4*)gp;
    const f32x4 g1 = *(const f32x4*)(gp + 4);
    bf16x8 bfv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bfv[e] = f2bf(g0[e]);
      bfv[4 + e] = f2bf(g1[e]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const bf16x8 af = *(const bf16x8*)(pt + (long)(dt * 16 + c) * a.span2 + kk * 32 + g * 8);
      acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfv, acc[dt], 0, 0, 0);
    }
  }
  if (row < S) {
    bf16* op = a.out + ((long)b * S + row) * a.ldout + h * 64 + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *(bf16x4*)(op + dt * 16) = (bf16x4){f2bf(acc[dt][0]), f2bf(acc[dt][1]), f2bf(acc[dt][2]), f2bf(acc[dt][3])};
  }
  // ---- G^T[h][r][b][r0 .. r0+31] (bf16): thread -> (r, 8-row chunk)
  for (int id = tid; id < a.span2 * 4; id += 128) {
    const int r = id >> 2, ch = id & 3;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = f2bf(G[(ch * 8 + e) * LDG + r]);
    *(bf16x8*)(a.GT + (((long)h * a.span2 + r) * a.B + b) * Sp + r0 + ch * 8) = v;
  }
}

}  // namespace

extern "C" int fbl_attn_rowdot(const void* dO, const void* O, int64_t ld, float* out, int B, int S, int nh,
                               void* stream) {
  if (ld % 8) return FBL_ERR_ALIGN;
  const long total = (long)B * S * nh;
  if (total <= 0) return 0;
  hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)dO, (const bf16*)O, (long)ld, out, B, S, nh);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_disent_attn_bwd_ds(const void* q, const void* k, const void* v, int64_t ldq, const void* dO,
                                      int64_t ldo, const void* dOT, int64_t t_sh, int64_t t_sb, int64_t t_sd,
                                      const void* pk, const void* pq, int64_t ldp, const int16_t* relidx,
                                      const int32_t* mask, const float* lse, const float* Dv, float scale, float p_drop,
                                      uint64_t seed, void* dV, int64_t lddv, void* dS, void* dST, int B, int S, int Sp,
                                      int nh, int span2, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64) return FBL_ERR_SHAPE;
  if ((ldq % 8) || (ldo % 8) || (ldp % 8) || (lddv % 4) || (t_sd % 8) || (t_sb % 8) || (t_sh % 8)) return FBL_ERR_ALIGN;
  if (B <= 0 || nh <= 0) return 0;
  BwdAArgs a{(const bf16*)q, (const bf16*)k, (const bf16*)v, ldq, (const bf16*)dO, ldo, (const bf16*)dOT, t_sh, t_sb,
             t_sd, (const bf16*)pk, (const bf16*)pq, ldp, relidx, mask, lse, Dv, scale, p_drop, seed, (bf16*)dV, lddv,
             (bf16*)dS, (bf16*)dST, B, S, Sp, nh, span2};
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)attn_bwd_ds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A_TOTAL);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_ds_kernel, dim3(Sp / 64, nh, B), dim3(256), A_TOTAL, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_disent_attn_bwd_shear(int neg, const void* X, const void* YT, int64_t y_sh, int64_t y_sb,
                                         int64_t y_sd, const void* PT, const int16_t* relidx, void* out, int64_t ldout,
                                         void* GT, int B, int S, int Sp, int nh, int span2, void* stream) {
  if (S < 1 || S > 512 || Sp < S || Sp % 64 || span2 > 512 || span2 % 32) return FBL_ERR_SHAPE;
  if ((ldout % 4) || (y_sd % 8) || (y_sb % 8) || (y_sh % 8)) return FBL_ERR_ALIGN;
  if (B <= 0 || nh <= 0) return 0;
  ShearArgs a{(const bf16*)X, (const bf16*)YT, y_sh, y_sb, y_sd, (const bf16*)PT, relidx, (bf16*)out, ldout, (bf16*)GT,
              B, S, Sp, nh, span2};
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e1 = hipFuncSetAttribute((const void*)attn_bwd_shear_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, C_TOTAL);
    hipError_t e2 = hipFuncSetAttribute((const void*)attn_bwd_shear_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, C_TOTAL);
    if (e1 != hipSuccess) return (int)e1;
    if (e2 != hipSuccess) return (int)e2;
    attr_set = true;
  }
  dim3 grid(Sp / 32, nh, B);
  if (neg)
    hipLaunchKernelGGL(attn_bwd_shear_kernel<true>, grid, dim3(128), C_TOTAL, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(attn_bwd_shear_kernel<false>, grid, dim3(128), C_TOTAL, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}
