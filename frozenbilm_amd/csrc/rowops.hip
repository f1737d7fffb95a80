// HBM-bound row / elementwise kernels of the FrozenBiLM hot path for gfx950 (wave64).
// One wave owns one row (H <= 2048): the row lives in registers, statistics via wave shuffles, 16-byte accesses.
#include "fbl_common.h"
#include "../../include/fbl.h"

namespace {

template <int VEC>
struct VecF {};
template <>
struct VecF<4> {
  typedef f32x4 T;
};
template <>
struct VecF<2> {
  typedef f32x2 T;
};

template <int VEC>
__device__ __forceinline__ void ldf(const float* p, float* v) {
  if (VEC == 4) {
    f32x4 x = *(const f32x4*)p;
    v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
  } else if (VEC == 2) {
    f32x2 x = *(const f32x2*)p;
    v[0] = x[0]; v[1] = x[1];
  } else {
    v[0] = p[0];
  }
}
template <int VEC>
__device__ __forceinline__ void stf(float* p, const float* v) {
  if (VEC == 4) *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]};
  else if (VEC == 2) *(f32x2*)p = (f32x2){v[0], v[1]};
  else p[0] = v[0];
}
template <int VEC>
__device__ __forceinline__ void stb(bf16* p, const float* v) {
  if (VEC == 4) *(bf16x4*)p = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
  else if (VEC == 2) *(bf16x2*)p = (bf16x2){f2bf(v[0]), f2bf(v[1])};
  else p[0] = f2bf(v[0]);
}

struct LnFwdArgs {
  const float* y; long ldy; float p_drop; uint64_t seed; const uint64_t* seed_dev;
  const float* r_plain; const float* r_t; const float* r_stats; const float* r_gamma; const float* r_beta;
  const int32_t* r_rowmask;
  const float* gamma; const float* beta; float eps; const int32_t* rowmask;
  float* out_t; float* out_stats; bf16* out_bf16; float* out_f32;
  int N, H;
};

// EPL = elements per lane = H/64
template <int EPL>
__global__ __launch_bounds__(256) void ln_fwd_kernel(LnFwdArgs a) {
  constexpr int VEC = (EPL % 4 == 0) ? 4 : ((EPL % 2 == 0) ? 2 : 1);
  constexpr int NIT = EPL / VEC;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.N) return;
  const int H = a.H;
  float v[EPL];
  const uint32_t thr = fbl_drop_thresh(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  const uint64_t seed = a.p_drop > 0.f ? fbl_seed(a.seed, a.seed_dev) : 0;
  float rmean = 0.f, rrstd = 0.f, rmask = 1.f;
  if (a.r_t) {
    rmean = a.r_stats[2 * (long)row];
    rrstd = a.r_stats[2 * (long)row + 1];
    if (a.r_rowmask) rmask = (float)a.r_rowmask[row];
  }
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int col = (it * 64 + lane) * VEC;
    float* vv = v + it * VEC;
    if (a.y) {
      ldf<VEC>(a.y + (long)row * a.ldy + col, vv);
      if (a.p_drop > 0.f) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) vv[c] *= fbl_dropout_scale(seed, (uint64_t)row * H + col + c, thr, inv_keep);
      }
    } else {
#pragma unroll
      for (int c = 0; c < VEC; ++c) vv[c] = 0.f;
    }
    if (a.r_plain) {
      float r[VEC];
      ldf<VEC>(a.r_plain + (long)row * H + col, r);
#pragma unroll
      for (int c = 0; c < VEC; ++c) vv[c] += r[c];
    }
    if (a.r_t) {
      float r[VEC], gg[VEC], bb[VEC];
      ldf<VEC>(a.r_t + (long)row * H + col, r);
      ldf<VEC>(a.r_gamma + col, gg);
      ldf<VEC>(a.r_beta + col, bb);
#pragma unroll
      for (int c = 0; c < VEC; ++c) vv[c] += ((r[c] - rmean) * rrstd * gg[c] + bb[c]) * rmask;
    }
    if (a.out_t) stf<VEC>(a.out_t + (long)row * H + col, vv);
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) s += v[e];
  const float mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < EPL; ++e) {
    const float d = v[e] - mean;
    q += d * d;
  }
  const float var = wave_sum(q) / (float)H;
  const float rstd = rsqrtf(var + a.eps);
  if (lane == 0 && a.out_stats) {
    a.out_stats[2 * (long)row] = mean;
    a.out_stats[2 * (long)row + 1] = rstd;
  }
  const float om = a.rowmask ? (float)a.rowmask[row] : 1.0f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int col = (it * 64 + lane) * VEC;
    float gg[VEC], bb[VEC], o[VEC];
    ldf<VEC>(a.gamma + col, gg);
    ldf<VEC>(a.beta + col, bb);
#pragma unroll
    for (int c = 0; c < VEC; ++c) o[c] = ((v[it * VEC + c] - mean) * rstd * gg[c] + bb[c]) * om;
    if (a.out_bf16) stb<VEC>(a.out_bf16 + (long)row * H + col, o);
    if (a.out_f32) stf<VEC>(a.out_f32 + (long)row * H + col, o);
  }
}

struct LnMatArgs {
  const float* t; const float* stats; const float* gamma; const float* beta; const int32_t* rowmask;
  const float* add_bcast; int S; float* out_f32; bf16* out_bf16; int N, H;
};
template <int EPL>
__global__ __launch_bounds__(256) void ln_mat_kernel(LnMatArgs a) {
  constexpr int VEC = (EPL % 4 == 0) ? 4 : ((EPL % 2 == 0) ? 2 : 1);
  constexpr int NIT = EPL / VEC;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.N) return;
  const int H = a.H;
  const float mean = a.stats[2 * (long)row], rstd = a.stats[2 * (long)row + 1];
  const float om = a.rowmask ? (float)a.rowmask[row] : 1.0f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int col = (it * 64 + lane) * VEC;
    float x[VEC], gg[VEC], bb[VEC], o[VEC];
    ldf<VEC>(a.t + (long)row * H + col, x);
    ldf<VEC>(a.gamma + col, gg);
    ldf<VEC>(a.beta + col, bb);
#pragma unroll
    for (int c = 0; c < VEC; ++c) o[c] = ((x[c] - mean) * rstd * gg[c] + bb[c]) * om;
    if (a.add_bcast) {
      float p[VEC];
      ldf<VEC>(a.add_bcast + (long)(row % a.S) * H + col, p);
#pragma unroll
      for (int c = 0; c < VEC; ++c) o[c] += p[c];
    }
    if (a.out_f32) stf<VEC>(a.out_f32 + (long)row * H + col, o);
    if (a.out_bf16) stb<VEC>(a.out_bf16 + (long)row * H + col, o);
  }
}

// ---- LayerNorm backward: persistent grid of LNB_BLOCKS blocks x 4 waves; each wave strides over rows and keeps its
// dgamma/dbeta partial in registers; block-reduced through LDS into ws[block][2][H]; a second tiny kernel folds ws
// into dgamma/dbeta (+=), deterministic.
constexpr int LNB_BLOCKS = 768;  // three 4-wave blocks per CU (the kernels run 3 waves per SIMD)
struct LnBwdArgs {
  const float* dout; const int32_t* rowmask; const float* t; const float* stats; const float* gamma;
  float p_drop; uint64_t seed; const uint64_t* seed_dev; float* out_dt; bf16* out_dy_bf16; float* out_dy_f32; float* ws; int N, H;
  long ld_dyb;  // row stride of out_dy_bf16 (elements)
};
template <int EPL>
__global__ __launch_bounds__(256) void ln_bwd_kernel(LnBwdArgs a) {
  constexpr int VEC = (EPL % 4 == 0) ? 4 : ((EPL % 2 == 0) ? 2 : 1);
  constexpr int NIT = EPL / VEC;
  __shared__ float red[4 * 64 * EPL];  // [wave][H]: cross-wave fold of one column sum at a time (dgamma, dbeta, colsum(dy))
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int H = a.H;
  float dg[EPL], db[EPL], dys[EPL], gam[EPL];
#pragma unroll
  for (int e = 0; e < EPL; ++e) dg[e] = db[e] = dys[e] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) ldf<VEC>(a.gamma + (it * 64 + lane) * VEC, gam + it * VEC);
  const uint32_t thr = fbl_drop_thresh(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  const uint64_t seed = a.p_drop > 0.f ? fbl_seed(a.seed, a.seed_dev) : 0;
  for (int row = blockIdx.x * 4 + wave; row < a.N; row += gridDim.x * 4) {
    const float mean = a.stats[2 * (long)row], rstd = a.stats[2 * (long)row + 1];
    const float om = a.rowmask ? (float)a.rowmask[row] : 1.0f;
    float g[EPL], xh[EPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int col = (it * 64 + lane) * VEC;
      float d[VEC], x[VEC];
      ldf<VEC>(a.dout + (long)row * H + col, d);
      ldf<VEC>(a.t + (long)row * H + col, x);
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        const int e = it * VEC + c;
        const float dd = d[c] * om;
        xh[e] = (x[c] - mean) * rstd;
        dg[e] += dd * xh[e];
        db[e] += dd;
        g[e] = dd * gam[e];
        s1 += g[e];
        s2 += g[e] * xh[e];
      }
    }
    s1 = wave_sum(s1) / (float)H;
    s2 = wave_sum(s2) / (float)H;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int col = (it * 64 + lane) * VEC;
      float dt[VEC], dy[VEC];
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        const int e = it * VEC + c;
        dt[c] = rstd * (g[e] - s1 - xh[e] * s2);
        dy[c] = dt[c];
        if (a.p_drop > 0.f) dy[c] *= fbl_dropout_scale(seed, (uint64_t)row * H + col + c, thr, inv_keep);
        dys[e] += dy[c];
      }
      if (a.out_dt) stf<VEC>(a.out_dt + (long)row * H + col, dt);
      if (a.out_dy_bf16) stb<VEC>(a.out_dy_bf16 + (long)row * a.ld_dyb + col, dy);
      if (a.out_dy_f32) stf<VEC>(a.out_dy_f32 + (long)row * H + col, dy);
    }
  }
  // the three column sums leave one after the other through the same [4][H] LDS tile (a third of the LDS of folding them
  // together: 24.6 KiB per block at H = 1536, so several blocks share a CU)
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    const float* src = which == 0 ? dg : (which == 1 ? db : dys);
    if (which) __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int col = (it * 64 + lane) * VEC;
#pragma unroll
      for (int c = 0; c < VEC; ++c) red[wave * H + col + c] = src[it * VEC + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += 256) {
      float s_ = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) s_ += red[w * H + i];
      a.ws[(long)blockIdx.x * 3 * H + which * H + i] = s_;
    }
  }
}
// Two waves per row (column halves), two rows per block iteration: half the per-lane state of ln_bwd_kernel (which needs
// > 256 registers at H = 1536 and therefore runs one wave per SIMD), so four waves per SIMD keep enough rows in flight to
// cover the HBM latency.  The two halves of a row exchange their (sum g, sum g*xhat) partials through LDS: one barrier
// per iteration, double-buffered.  Same arithmetic and the same deterministic ws / fold protocol as ln_bwd_kernel.
template <int EPL>
__global__ __launch_bounds__(256, 3) void ln_bwd2_kernel(LnBwdArgs a) {
  static_assert(EPL % 2 == 0, "two waves share a row");
  constexpr int EW = EPL / 2;  // elements per lane
  constexpr int VEC = (EW % 4 == 0) ? 4 : ((EW % 2 == 0) ? 2 : 1);
  constexpr int NIT = EW / VEC;
  __shared__ float red[2 * 64 * EPL];  // [row slot][H]
  __shared__ float xch[2][2][2][2];    // [parity][slot][half][s1, s2]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = wave >> 1, half = wave & 1;
  const int H = a.H;
  const int c0 = half * (H / 2);
  float dg[EW], db[EW], dys[EW], gam[EW];
#pragma unroll
  for (int e = 0; e < EW; ++e) dg[e] = db[e] = dys[e] = 0.f;
#pragma unroll
  for (int it = 0; it < NIT; ++it) ldf<VEC>(a.gamma + c0 + (it * 64 + lane) * VEC, gam + it * VEC);
  const uint32_t thr = fbl_drop_thresh(a.p_drop);
  const float inv_keep = a.p_drop > 0.f ? 1.0f / (1.0f - a.p_drop) : 1.0f;
  const uint64_t seed = a.p_drop > 0.f ? fbl_seed(a.seed, a.seed_dev) : 0;
  int par = 0;
  for (int base = blockIdx.x * 2; base < a.N; base += gridDim.x * 2, par ^= 1) {
    const int row = base + slot;
    const bool live = row < a.N;
    const int rr = live ? row : a.N - 1;
    const float mean = a.stats[2 * (long)rr], rstd = a.stats[2 * (long)rr + 1];
    const float om = live ? (a.rowmask ? (float)a.rowmask[rr] : 1.0f) : 0.f;
    float g[EW], xh[EW];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int col = c0 + (it * 64 + lane) * VEC;
      float d[VEC], x[VEC];
      ldf<VEC>(a.dout + (long)rr * H + col, d);
      ldf<VEC>(a.t + (long)rr * H + col, x);
#pragma unroll
      for (int c = 0; c < VEC; ++c) {
        const int e = it * VEC + c;
        const float dd = d[c] * om;
        xh[e] = (x[c] - mean) * rstd;
        dg[e] += dd * xh[e];
        db[e] += dd;
        g[e] = dd * gam[e];
        s1 += g[e];
        s2 += g[e] * xh[e];
      }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
      xch[par][slot][half][0] = s1;
      xch[par][slot][half][1] = s2;
    }
    __syncthreads();
    // (both halves add the two partials in the same order: bit-identical s1 / s2 on both waves of a row)
    s1 = (xch[par][slot][0][0] + xch[par][slot][1][0]) / (float)H;
    s2 = (xch[par][slot][0][1] + xch[par][slot][1][1]) / (float)H;
    if (live) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int col = c0 + (it * 64 + lane) * VEC;
        float dt[VEC], dy[VEC];
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          const int e = it * VEC + c;
          dt[c] = rstd * (g[e] - s1 - xh[e] * s2);
          dy[c] = dt[c];
          if (a.p_drop > 0.f) dy[c] *= fbl_dropout_scale(seed, (uint64_t)row * H + col + c, thr, inv_keep);
          dys[e] += dy[c];
        }
        if (a.out_dt) stf<VEC>(a.out_dt + (long)row * H + col, dt);
        if (a.out_dy_bf16) stb<VEC>(a.out_dy_bf16 + (long)row * a.ld_dyb + col, dy);
        if (a.out_dy_f32) stf<VEC>(a.out_dy_f32 + (long)row * H + col, dy);
      }
    }
  }
#pragma unroll
  for (int which = 0; which < 3; ++which) {
    const float* src = which == 0 ? dg : (which == 1 ? db : dys);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int col = c0 + (it * 64 + lane) * VEC;
#pragma unroll
      for (int c = 0; c < VEC; ++c) red[slot * H + col + c] = src[it * VEC + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H; i += 256) a.ws[(long)blockIdx.x * 3 * H + which * H + i] = red[i] + red[H + i];
  }
}

// fold [nblk][ncols] partial sums: a block owns 16 columns, its 16 row-groups stride over the partial rows
__device__ __forceinline__ float fold16(const float* ws, int nblk, int ncols, int col) {
  __shared__ float red[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float s = 0.f;
  if (col < ncols)
    for (int b = ty; b < nblk; b += 16) s += ws[(long)b * ncols + col];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0) {
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][tx];
  }
  return s;
}
// A block owns 32 consecutive columns of ws[nblk][3H]: 8 lanes x float4 read one 128-byte line of a partial row, the 32
// lane groups stride over the rows with eight independent loads in flight each (the fold is a latency chain, not a
// bandwidth problem: the 14 MB of partials have just been written and sit in the Infinity Cache; with 16 groups x 4 loads
// it took 11 us per LayerNorm, 54 times a step), then a fixed-order fold of the groups -- in-wave butterflies over the 8
// groups of a wave, the 4 waves through LDS -- deterministic.
__global__ __launch_bounds__(256) void ln_bwd_fold_kernel(const float* ws, int nblk, int H, float* dgamma, float* dbeta,
                                                          float* dysum) {
  __shared__ f32x4 red[4][8];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 32 + tx * 4;
  const int ncols = 3 * H;  // (H % 64 == 0: a block never straddles two of the three sums, nor the end)
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (col < ncols) {
    const float* p = ws + col;
    int b = ty;
    for (; b + 224 < nblk; b += 256) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(p + (long)(b + 32 * u) * ncols);
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; b < nblk; b += 32) s += *(const f32x4*)(p + (long)b * ncols);
  }
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] += __shfl_xor(s[r], o, 64);
  }
  if (lane < 8) red[wave][lane] = s;
  __syncthreads();
  if (threadIdx.x < 8 && col < ncols) {
    s = (red[0][tx] + red[1][tx]) + (red[2][tx] + red[3][tx]);
    float* dst = col < H ? dgamma : (col < 2 * H ? dbeta : dysum);
    if (dst) {
      float* q = dst + (col % H);
#pragma unroll
      for (int r = 0; r < 4; ++r) q[r] += s[r];
    }
  }
}

__global__ void embed_gather_kernel(const int64_t* ids, const float* E, const float* vproj, int B, int T, int L, int H,
                                    float* out) {
  const int S = T + L;
  const int row = blockIdx.x;  // b*S + s
  const int b = row / S, s = row % S;
  const float* src = (s < T) ? vproj + ((long)b * T + s) * H : E + ids[(long)b * L + (s - T)] * (long)H;
  float* dst = out + (long)row * H;
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) *(f32x4*)(dst + c) = *(const f32x4*)(src + c);
}

// ---- input side: variable-length fp16 CLIP feature clips -> fixed [B, T, F] fp32 (+ lengths, mask)
// One block per output row (b, j): 16-byte loads of 8 halves, fp32 stores; slot j of a clip with n > T rows reads row
// (j*n)/T, a clip with n <= T rows is copied and zero padded (datasets/videotext_dataset.py:27-43).
__global__ void video_stage_kernel(const _Float16* feats, const int64_t* row_off, const int32_t* n_rows, int T, int F,
                                   float* out, int64_t* video_len, int64_t* video_mask) {
  const int b = blockIdx.x / T, j = blockIdx.x % T;
  const int n = n_rows[b];
  const int len = n < T ? n : T;
  if (threadIdx.x == 0) {
    if (j == 0 && video_len) video_len[b] = len;
    if (video_mask) video_mask[(long)b * T + j] = j < len ? 1 : 0;
  }
  float* dst = out + ((long)b * T + j) * F;
  if (j >= len) {
    for (int c = threadIdx.x * 4; c < F; c += blockDim.x * 4) *(f32x4*)(dst + c) = (f32x4){0.f, 0.f, 0.f, 0.f};
    return;
  }
  const long r = n > T ? ((long)j * n) / T : j;
  const _Float16* src = feats + (row_off[b] + r) * (long)F;
  typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
  for (int c = threadIdx.x * 8; c < F; c += blockDim.x * 8) {
    const f16x8 h = *(const f16x8*)(src + c);
    *(f32x4*)(dst + c) = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    *(f32x4*)(dst + c + 4) = (f32x4){(float)h[4], (float)h[5], (float)h[6], (float)h[7]};
  }
}

// ---- masked-LM corruption on the device (util/misc.py:14-56 semantics, counter-based RNG): per token four draws
// keyed by (seed, 4*index + k): select with probability p unless the id is special / padding; of the selected 80 %
// become [MASK], half of the rest a uniformly random id, the rest stay; labels = original id where selected else -100.
__global__ void mask_tokens_kernel(int64_t* ids, int64_t* labels, long n, const int64_t* special, int n_special, float p,
                                   int64_t mask_id, int64_t vocab, uint64_t seed) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t id = ids[i];
  bool sp = false;
  for (int k = 0; k < n_special; ++k) sp |= (special[k] == id);
  const float inv = 1.0f / 4294967296.0f;
  const bool pick = !sp && (float)fbl_hash(seed, 4ull * i) * inv < p;
  labels[i] = pick ? id : -100;
  if (!pick) return;
  if ((float)fbl_hash(seed, 4ull * i + 1) * inv < 0.8f) ids[i] = mask_id;
  else if ((float)fbl_hash(seed, 4ull * i + 2) * inv < 0.5f) ids[i] = (int64_t)(fbl_hash(seed, 4ull * i + 3) % (uint64_t)vocab);
}

__global__ void im2col3_kernel(const bf16* x, bf16* out, int B, int S, int H) {
  const int row = blockIdx.x;
  const int s = row % S;
  const int chunks = 3 * H / 8;
  for (int i = threadIdx.x; i < chunks; i += blockDim.x) {
    const int k = (i * 8) / H, c = (i * 8) % H;
    const int ss = s + k - 1;
    bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (ss >= 0 && ss < S) v = *(const bf16x8*)(x + ((long)row + k - 1) * H + c);
    *(bf16x8*)(out + (long)row * 3 * H + i * 8) = v;
  }
}
__global__ void col2im3_kernel(const float* dcol, float* dx, int B, int S, int H, int accumulate) {
  const int row = blockIdx.x;
  const int s = row % S;
  for (int c = threadIdx.x * 4; c < H; c += blockDim.x * 4) {
    f32x4 acc = accumulate ? *(const f32x4*)(dx + (long)row * H + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ss = s - k + 1;  // output position whose tap k reads input position s
      if (ss >= 0 && ss < S) acc += *(const f32x4*)(dcol + ((long)row - k + 1) * 3 * H + k * H + c);
    }
    *(f32x4*)(dx + (long)row * H + c) = acc;
  }
}

__global__ void dropout_gelu_fwd_kernel(const float* c, float p, uint64_t seed0, const uint64_t* seed_dev, float* out, long n) {
  const uint64_t seed = p > 0.f ? fbl_seed(seed0, seed_dev) : 0;
  const uint32_t thr = fbl_drop_thresh(p);
  const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = c[i];
    if (p > 0.f) v *= fbl_dropout_scale(seed, (uint64_t)i, thr, ik);
    out[i] = gelu_erf(v);
  }
}
__global__ void dropout_gelu_bwd_kernel(const float* dy, const float* c, float p, uint64_t seed0, const uint64_t* seed_dev,
                                        bf16* ob, float* of, long n) {
  const uint64_t seed = p > 0.f ? fbl_seed(seed0, seed_dev) : 0;
  const uint32_t thr = fbl_drop_thresh(p);
  const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float sc = p > 0.f ? fbl_dropout_scale(seed, (uint64_t)i, thr, ik) : 1.f;
    const float g = dy[i] * dgelu_erf(c[i] * sc) * sc;
    if (ob) ob[i] = f2bf(g);
    if (of) of[i] = g;
  }
}

// 64x64 tile transpose through LDS; output rows are the input columns, zero-padded to rows_pad.
template <bool IN_BF16>
__global__ __launch_bounds__(256) void transpose_kernel(const void* in_, long ld_in, int rows, int cols, bf16* out,
                                                        long rows_pad) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols)
      v = IN_BF16 ? bf2f(((const bf16*)in_)[(long)r * ld_in + c]) : ((const float*)in_)[(long)r * ld_in + c];
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows_pad) out[(long)c * rows_pad + r] = f2bf(tile[tx][i]);
  }
}

constexpr int CS_BLOCKS = 512;
template <bool IN_BF16>
__global__ __launch_bounds__(256) void colsum_kernel(const void* in_, long ld_in, int rows, int cols, float* ws) {
  // block handles a strided set of rows for a 256-wide column slab
  const int c = blockIdx.y * 256 + threadIdx.x;
  float s = 0.f;
  if (c < cols)
    for (int r = blockIdx.x; r < rows; r += gridDim.x)
      s += IN_BF16 ? bf2f(((const bf16*)in_)[(long)r * ld_in + c]) : ((const float*)in_)[(long)r * ld_in + c];
  if (c < cols) ws[(long)blockIdx.x * cols + c] = s;
}
__global__ __launch_bounds__(256) void colsum_fold_kernel(const float* ws, int nblk, int cols, float* out) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  const float s = fold16(ws, nblk, cols, c);
  if ((threadIdx.x >> 4) == 0 && c < cols) out[c] += s;
}

// Vt[b,h,d,s] = V[b*S+s, h*64+d]; one block per (s-tile of 64, h, b).  16-byte global loads and stores: a thread
// loads two 8-element row chunks, the tile sits in LDS with a 33-dword row stride (conflict-free column walks), and a
// thread stores 8 consecutive s of one d, 8 lanes covering a full 128-byte output row segment.
__global__ __launch_bounds__(256) void head_transpose_kernel(const bf16* v, long ldv, bf16* vt, int B, int S, int Sp,
                                                             int nh, long sh, long sb, long sd) {
  __shared__ uint32_t tile[64 * 33];
  const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
  const int t = threadIdx.x;
  {
    const int row = t >> 2, c0 = (t & 3) * 2;
    const int s = s0 + row;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 x = make_uint4(0, 0, 0, 0);
      if (s < S) x = *(const uint4*)(v + ((long)b * S + s) * ldv + h * 64 + (c0 + c) * 8);
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

// Batched [rows, cols] -> [cols, rows] bf16 transposes of equally-shaped matrices living at element offsets
// src_off[z] / dst_off[z] of two base buffers (the adapters' trainable weights inside the flat bf16 parameter copy).
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16* src, const int64_t* src_off, bf16* dst,
                                                                const int64_t* dst_off, int rows, int cols) {
  __shared__ bf16 tile[64][66];
  const bf16* in = src + src_off[blockIdx.z];
  bf16* out = dst + dst_off[blockIdx.z];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(long)r * cols + c] : f2bf(0.f);
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) out[(long)c * rows + r] = tile[tx][i];
  }
}

// ---- cross entropy
// One block per row; a single pass over the 128100 logits with 16-byte loads: every thread keeps an online
// (max, sum exp) pair that is merged across the block at the end.
__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* logits, long ldv, const int64_t* labels, int N, int V,
                                                     float* row_lse, float* loss_sum_cnt) {
  const int row = blockIdx.x;
  const int64_t lab = labels[row];
  if (lab < 0) {
    if (threadIdx.x == 0) row_lse[row] = 0.f;
    return;
  }
  __shared__ float red_m[4], red_s[4];
  const float* x = logits + (long)row * ldv;
  float m = -INFINITY, s = 0.f;
  auto push = [&](float v) {
    if (v > m) {
      s = s * __expf(m - v) + 1.f;  // (m = -inf: s is 0)
      m = v;
    } else {
      s += __expf(v - m);
    }
  };
  const bool vec = ((ldv & 3) == 0) && (((uintptr_t)logits & 15) == 0);
  const int V4 = vec ? (V >> 2) : 0;
  for (int i = threadIdx.x; i < V4; i += 256) {
    const f32x4 v = *(const f32x4*)(x + 4 * i);
    const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    if (vm > m) {
      s *= __expf(m - vm);
      m = vm;
    }
    s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
  }
  for (int i = 4 * V4 + threadIdx.x; i < V; i += 256) push(x[i]);
  // merge (m, s) pairs: wave, then block
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
    const float mn = fmaxf(m, m2);
    s = (mn == -INFINITY) ? 0.f : s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
  }
  if ((threadIdx.x & 63) == 0) {
    red_m[threadIdx.x >> 6] = m;
    red_s[threadIdx.x >> 6] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3])), S = 0.f;
    for (int k = 0; k < 4; ++k) S += (red_m[k] == -INFINITY) ? 0.f : red_s[k] * __expf(red_m[k] - M);
    const float lse = M + logf(S);
    row_lse[row] = lse;
  }
}
// loss_sum_cnt[0] += sum over labelled rows of (lse - logit[label]), [1] += their count: ONE block, thread t takes rows
// t, t + 256, ..., fixed-order fold -- the loss is reproducible bit for bit (atomic adds from the row blocks were not).
__global__ __launch_bounds__(256) void ce_fold_kernel(const float* logits, long ldv, const int64_t* labels, int N,
                                                      const float* row_lse, float* loss_sum_cnt) {
  __shared__ float red_s[4], red_c[4];
  float s = 0.f, c = 0.f;
  for (int r = threadIdx.x; r < N; r += 256) {
    const int64_t lab = labels[r];
    if (lab >= 0) {
      s += row_lse[r] - logits[(long)r * ldv + lab];
      c += 1.f;
    }
  }
  s = wave_sum(s);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    red_s[threadIdx.x >> 6] = s;
    red_c[threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    loss_sum_cnt[0] += (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    loss_sum_cnt[1] += (red_c[0] + red_c[1]) + (red_c[2] + red_c[3]);
  }
}
__global__ __launch_bounds__(256) void ce_bwd_rows_kernel(const float* logits, long ldv, const int64_t* labels,
                                                          const int32_t* rows, int V, int Vp, const float* row_lse,
                                                          const float* loss_sum_cnt, float gscale, const float* gscale_dev,
                                                          bf16* out) {
  const int r = blockIdx.x;
  const int row = rows[r];
  const int lab = (int)labels[row];
  const float lse = row_lse[row];
  // (an ignored row -- label < 0: the padding entries of a fixed-capacity row list -- has exactly zero gradient)
  const float sc = lab < 0 ? 0.f : gscale * (gscale_dev ? gscale_dev[0] : 1.0f) / fmaxf(loss_sum_cnt[1], 1.0f);
  const float* x = logits + (long)row * ldv;
  bf16* o = out + (long)r * Vp;
  const bool vec = ((ldv & 3) == 0) && ((Vp & 3) == 0) && (((uintptr_t)logits & 15) == 0) && (((uintptr_t)out & 7) == 0);
  const int V4 = vec ? (V >> 2) : 0;
  for (int i = threadIdx.x; i < V4; i += 256) {
    const f32x4 v = *(const f32x4*)(x + 4 * i);
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = lab < 0 ? 0.f : (__expf(v[e] - lse) - ((4 * i + e) == lab ? 1.f : 0.f)) * sc;
    *(bf16x4*)(o + 4 * i) = (bf16x4){f2bf(g[0]), f2bf(g[1]), f2bf(g[2]), f2bf(g[3])};
  }
  for (int i = 4 * V4 + threadIdx.x; i < Vp; i += 256) {
    float g = 0.f;
    if (i < V && lab >= 0) g = (__expf(x[i] - lse) - (i == lab ? 1.f : 0.f)) * sc;
    o[i] = f2bf(g);
  }
}

__global__ void gather_rows_bf16_kernel(const bf16* in, long ld, const int32_t* rows, int cols, bf16* out) {
  const int r = blockIdx.x;
  const bf16* src = in + (long)rows[r] * ld;
  for (int c = threadIdx.x * 8; c < cols; c += blockDim.x * 8) *(bf16x8*)(out + (long)r * cols + c) = *(const bf16x8*)(src + c);
}
__global__ void scatter_rows_f32_kernel(const float* in, const int32_t* rows, int cols, float* out, long ld) {
  const int r = blockIdx.x;
  float* dst = out + (long)rows[r] * ld;
  for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[c] += in[(long)r * cols + c];
}

// sum of squares, reproducible bit for bit: every block leaves its partial sum in ws[block], a one-block second launch adds
// the partials in index order (an atomicAdd per block would make the clip factor -- and with it every parameter -- depend
// on the order in which the blocks happen to finish)
constexpr int SUMSQ_BLOCKS = 1024;
__global__ void sumsq_kernel(const float* x, long n, float* ws) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += x[i] * x[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) ws[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_fold_kernel(const float* ws, int nblk, float* out) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) s += ws[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] += red[0];
}
__global__ void adam_flat_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                                 float eps, float wd, float bc1, float bc2, const float* sumsq, float max_norm,
                                 float grad_scale) {
  float clip = grad_scale;
  if (sumsq && max_norm > 0.f) {
    const float norm = sqrtf(sumsq[0]) * grad_scale;
    clip *= fminf(1.0f, max_norm / (norm + 1e-6f));
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * clip;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}
__global__ void cast_bf16_kernel(const float* in, bf16* out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = f2bf(in[i]);
}
__global__ void dropout_f32_kernel(const float* in, float p, uint64_t seed0, const uint64_t* seed_dev, float* of, bf16* ob, long n) {
  const uint64_t seed = p > 0.f ? fbl_seed(seed0, seed_dev) : 0;
  const uint32_t thr = fbl_drop_thresh(p);
  const float ik = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = in[i];
    if (p > 0.f) v *= fbl_dropout_scale(seed, (uint64_t)i, thr, ik);
    if (of) of[i] = v;
    if (ob) ob[i] = f2bf(v);
  }
}
__global__ void dropout_bf16_kernel(bf16* x, float p, uint64_t seed0, const uint64_t* seed_dev, long n) {
  const uint64_t seed = fbl_seed(seed0, seed_dev);
  const uint32_t thr = fbl_drop_thresh(p);
  const float ik = 1.f / (1.f - p);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    x[i] = f2bf(bf2f(x[i]) * fbl_dropout_scale(seed, (uint64_t)i, thr, ik));
}

// out[i] = sum_s dropout_{seed_s}(x[s][i]): the per-layer-execution gradients of the shared relative-position table, each
// through the mask its forward drew (model/deberta.py:779 pos_dropout), folded into one tensor in slice order (deterministic)
struct DropSumArgs {
  uint64_t seeds[FBL_DROPSUM_MAX_SLICES];
  const float* x;
  const uint64_t* seed_dev;
  float* out;
  long n, key0;
  int ns;
  float p;
};
__global__ void dropout_sum_kernel(DropSumArgs a) {
  const uint32_t thr = fbl_drop_thresh(a.p);
  const float ik = a.p > 0.f ? 1.f / (1.f - a.p) : 1.f;
  const uint64_t word = (a.p > 0.f && a.seed_dev) ? *a.seed_dev : 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += (long)gridDim.x * blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < a.ns; ++s) {
      float v = a.x[(long)s * a.n + i];
      if (a.p > 0.f) v *= fbl_dropout_scale(a.seeds[s] + word, (uint64_t)(a.key0 + i), thr, ik);
      acc += v;
    }
    a.out[i] = acc;
  }
}

// dst[e][r][h*64 + c] (bf16, row stride ld_dst) = src[e][h][r][c] (fp32): the per-head position-table gradients of every layer
// execution -> the [rows, heads*64] operand layout of the projection GEMM; 8 lanes per 64-float head row (2 x 16-byte loads)
__global__ void heads_to_rows_bf16_kernel(const float* src, bf16* dst, int E, int nh, int rows, long ld_dst) {
  const long total = (long)E * nh * rows * 8;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(t & 7);
    const long q = t >> 3;  // (e, h, r)
    const int r = (int)(q % rows);
    const long eh = q / rows;
    const int h = (int)(eh % nh);
    const long e = eh / nh;
    const f32x4 a = *(const f32x4*)(src + q * 64 + c8 * 8), b = *(const f32x4*)(src + q * 64 + c8 * 8 + 4);
    *(bf16x8*)(dst + (e * rows + r) * ld_dst + h * 64 + c8 * 8) =
        (bf16x8){f2bf(a[0]), f2bf(a[1]), f2bf(a[2]), f2bf(a[3]), f2bf(b[0]), f2bf(b[1]), f2bf(b[2]), f2bf(b[3])};
  }
}

inline int grid1d(long n, int block = 256, int cap = 256 * 16) {
  long b = (n + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

#define FBL_EPL_DISPATCH(H, KERNEL, GRID, ARGS, STREAM)                                              \
  switch ((H) / 64) {                                                                                \
    case 1: hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(256), 0, STREAM, ARGS); break;                  \
    case 2: hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(256), 0, STREAM, ARGS); break;                  \
    case 4: hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), 0, STREAM, ARGS); break;                  \
    case 8: hipLaunchKernelGGL(KERNEL<8>, GRID, dim3(256), 0, STREAM, ARGS); break;                  \
    case 12: hipLaunchKernelGGL(KERNEL<12>, GRID, dim3(256), 0, STREAM, ARGS); break;                \
    case 16: hipLaunchKernelGGL(KERNEL<16>, GRID, dim3(256), 0, STREAM, ARGS); break;                \
    case 24: hipLaunchKernelGGL(KERNEL<24>, GRID, dim3(256), 0, STREAM, ARGS); break;                \
    case 32: hipLaunchKernelGGL(KERNEL<32>, GRID, dim3(256), 0, STREAM, ARGS); break;                \
    default: return FBL_ERR_SHAPE;                                                                   \
  }

extern "C" int fbl_abi_version(void) { return 8; }

extern "C" int fbl_embed_gather(const int64_t* ids, const float* E, const float* vproj, int B, int T, int L, int H,
                                float* out_t, void* stream) {
  if (H % 4) return FBL_ERR_SHAPE;
  if (T > 0 && !vproj) return FBL_ERR_ARG;
  if (B * (T + L) <= 0) return 0;
  hipLaunchKernelGGL(embed_gather_kernel, dim3(B * (T + L)), dim3(128), 0, (hipStream_t)stream, ids, E, vproj, B, T, L,
                     H, out_t);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_video_stage_f16(const void* feats_f16, const int64_t* row_off, const int32_t* n_rows, int B, int T, int F,
                                   float* out_f32, int64_t* video_len, int64_t* video_mask, void* stream) {
  if (F % 8 || ((uintptr_t)feats_f16 & 15) || ((uintptr_t)out_f32 & 15)) return FBL_ERR_ALIGN;
  if (B <= 0 || T <= 0) return 0;
  hipLaunchKernelGGL(video_stage_kernel, dim3(B * T), dim3(128), 0, (hipStream_t)stream, (const _Float16*)feats_f16, row_off,
                     n_rows, T, F, out_f32, video_len, video_mask);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_mask_tokens(int64_t* ids, int64_t* labels, int64_t n, const int64_t* special_ids, int n_special,
                               float mlm_probability, int64_t mask_token_id, int64_t vocab_size, uint64_t seed,
                               void* stream) {
  if (n <= 0) return 0;
  if (vocab_size <= 0 || mlm_probability < 0.f || mlm_probability > 1.f || n_special < 0) return FBL_ERR_ARG;
  hipLaunchKernelGGL(mask_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ids, labels,
                     (long)n, special_ids, n_special, mlm_probability, mask_token_id, vocab_size, seed);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_ln_fwd(const float* y, int64_t ldy, float p_drop, uint64_t seed, const uint64_t* seed_dev, const float* r_plain,
                          const float* r_t, const float* r_stats, const float* r_gamma, const float* r_beta,
                          const int32_t* r_rowmask, const float* gamma, const float* beta, float eps,
                          const int32_t* rowmask, float* out_t, float* out_stats, void* out_bf16, float* out_f32, int N,
                          int H, void* stream) {
  if (H % 64 || H > 2048) return FBL_ERR_SHAPE;
  if (r_t && (!r_stats || !r_gamma || !r_beta)) return FBL_ERR_ARG;
  if (N <= 0) return 0;
  LnFwdArgs a{y, ldy, p_drop, seed, seed_dev, r_plain, r_t, r_stats, r_gamma, r_beta, r_rowmask, gamma, beta, eps, rowmask,
              out_t, out_stats, (bf16*)out_bf16, out_f32, N, H};
  dim3 grid((N + 3) / 4);
  FBL_EPL_DISPATCH(H, ln_fwd_kernel, grid, a, (hipStream_t)stream);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_ln_materialize(const float* t, const float* stats, const float* gamma, const float* beta,
                                  const int32_t* rowmask, const float* add_bcast, int S, float* out_f32, void* out_bf16,
                                  int N, int H, void* stream) {
  if (H % 64 || H > 2048) return FBL_ERR_SHAPE;
  if (N <= 0) return 0;
  LnMatArgs a{t, stats, gamma, beta, rowmask, add_bcast, S > 0 ? S : 1, out_f32, (bf16*)out_bf16, N, H};
  dim3 grid((N + 3) / 4);
  FBL_EPL_DISPATCH(H, ln_mat_kernel, grid, a, (hipStream_t)stream);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t fbl_ln_bwd_ws_floats(int H) { return (int64_t)LNB_BLOCKS * 3 * H; }
extern "C" int fbl_ln_bwd(const float* dout, const int32_t* rowmask, const float* t, const float* stats,
                          const float* gamma, float p_drop, uint64_t seed, const uint64_t* seed_dev, float* out_dt, void* out_dy_bf16,
                          float* out_dy_f32, float* dgamma, float* dbeta, float* dysum, float* ws, int N, int H,
                          int64_t ld_dy_bf16, void* stream) {
  if (H % 64 || H > 2048) return FBL_ERR_SHAPE;
  if (N <= 0) return 0;
  if (ld_dy_bf16 == 0) ld_dy_bf16 = H;
  if (ld_dy_bf16 < H || (ld_dy_bf16 % 8)) return FBL_ERR_ALIGN;
  LnBwdArgs a{dout, rowmask, t, stats, gamma, p_drop, seed, seed_dev, out_dt, (bf16*)out_dy_bf16, out_dy_f32, ws, N, H, ld_dy_bf16};
  int nblk;
  if ((H / 64) % 2 == 0) {  // two waves per row, two rows per block iteration
    nblk = (N + 1) / 2;
    if (nblk > LNB_BLOCKS) nblk = LNB_BLOCKS;
    dim3 grid(nblk);
    switch (H / 64) {
      case 2: hipLaunchKernelGGL(ln_bwd2_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
      case 4: hipLaunchKernelGGL(ln_bwd2_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
      case 8: hipLaunchKernelGGL(ln_bwd2_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
      case 12: hipLaunchKernelGGL(ln_bwd2_kernel<12>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
      case 16: hipLaunchKernelGGL(ln_bwd2_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
      case 24: hipLaunchKernelGGL(ln_bwd2_kernel<24>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
      case 32: hipLaunchKernelGGL(ln_bwd2_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, a); break;
      default: return FBL_ERR_SHAPE;
    }
  } else {
    nblk = (N + 3) / 4;
    if (nblk > LNB_BLOCKS) nblk = LNB_BLOCKS;
    dim3 grid(nblk);
    FBL_EPL_DISPATCH(H, ln_bwd_kernel, grid, a, (hipStream_t)stream);
  }
  FBL_CHECK_LAUNCH();
  if (dgamma || dbeta || dysum) {
    hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3((3 * H + 31) / 32), dim3(256), 0, (hipStream_t)stream, ws, nblk, H,
                       dgamma, dbeta, dysum);
    FBL_CHECK_LAUNCH();
  }
  return 0;
}

extern "C" int fbl_im2col3(const void* x_bf16, void* out_bf16, int B, int S, int H, void* stream) {
  if (H % 8) return FBL_ERR_SHAPE;
  if (B * S <= 0) return 0;
  hipLaunchKernelGGL(im2col3_kernel, dim3(B * S), dim3(256), 0, (hipStream_t)stream, (const bf16*)x_bf16,
                     (bf16*)out_bf16, B, S, H);
  FBL_CHECK_LAUNCH();
  return 0;
}
extern "C" int fbl_col2im3(const float* dcol, float* dx, int B, int S, int H, int accumulate, void* stream) {
  if (H % 4) return FBL_ERR_SHAPE;
  if (B * S <= 0) return 0;
  hipLaunchKernelGGL(col2im3_kernel, dim3(B * S), dim3(256), 0, (hipStream_t)stream, dcol, dx, B, S, H, accumulate);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_dropout_gelu_fwd(const float* c, float p_drop, uint64_t seed, const uint64_t* seed_dev, float* out_f32,
                                    int64_t n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(dropout_gelu_fwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, c, p_drop, seed, seed_dev,
                     out_f32, (long)n);
  FBL_CHECK_LAUNCH();
  return 0;
}
extern "C" int fbl_dropout_gelu_bwd(const float* dy, const float* c, float p_drop, uint64_t seed, const uint64_t* seed_dev,
                                    void* out_bf16, float* out_f32, int64_t n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(dropout_gelu_bwd_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, dy, c, p_drop, seed, seed_dev,
                     (bf16*)out_bf16, out_f32, (long)n);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_transpose_to_bf16(const void* in, int in_is_bf16, int64_t ld_in, int rows, int cols, void* out_bf16,
                                     int64_t rows_pad, void* stream) {
  if (rows_pad < rows) return FBL_ERR_ARG;
  if (rows_pad <= 0 || cols <= 0) return 0;
  dim3 grid((unsigned)((rows_pad + 63) / 64), (cols + 63) / 64);
  if (in_is_bf16)
    hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, rows, cols,
                       (bf16*)out_bf16, (long)rows_pad);
  else
    hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, rows, cols,
                       (bf16*)out_bf16, (long)rows_pad);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_transpose_batched_bf16(const void* src_bf16, const int64_t* src_off, void* dst_bf16,
                                          const int64_t* dst_off, int count, int rows, int cols, void* stream) {
  if (count <= 0 || rows <= 0 || cols <= 0) return 0;
  hipLaunchKernelGGL(transpose_batched_kernel, dim3((rows + 63) / 64, (cols + 63) / 64, count), dim3(256), 0,
                     (hipStream_t)stream, (const bf16*)src_bf16, src_off, (bf16*)dst_bf16, dst_off, rows, cols);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t fbl_colsum_ws_floats(int cols) { return (int64_t)CS_BLOCKS * cols; }
extern "C" int fbl_colsum(const void* in, int in_is_bf16, int64_t ld_in, int rows, int cols, float* out, float* ws,
                          void* stream) {
  if (rows <= 0 || cols <= 0) return 0;
  int nblk = rows < CS_BLOCKS ? rows : CS_BLOCKS;
  dim3 grid(nblk, (cols + 255) / 256);
  if (in_is_bf16)
    hipLaunchKernelGGL(colsum_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, rows, cols, ws);
  else
    hipLaunchKernelGGL(colsum_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, in, (long)ld_in, rows, cols, ws);
  FBL_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((cols + 15) / 16), dim3(256), 0, (hipStream_t)stream, ws, nblk, cols,
                     out);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_head_transpose(const void* v_bf16, int64_t ldv, void* vt_bf16, int B, int S, int Sp, int nh,
                                  int64_t out_sh, int64_t out_sb, int64_t out_sd, void* stream) {
  if (Sp < S || Sp % 64) return FBL_ERR_SHAPE;
  if (ldv % 8 || out_sh % 8 || out_sb % 8 || out_sd % 8 || ((uintptr_t)v_bf16 & 15) || ((uintptr_t)vt_bf16 & 15)) return FBL_ERR_ALIGN;
  if (B * S <= 0) return 0;
  hipLaunchKernelGGL(head_transpose_kernel, dim3(Sp / 64, nh, B), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)v_bf16, (long)ldv, (bf16*)vt_bf16, B, S, Sp, nh, (long)out_sh, (long)out_sb,
                     (long)out_sd);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_ce_fwd(const float* logits, int64_t ldv, const int64_t* labels, int N, int V, float* row_lse,
                          float* loss_sum_cnt, void* stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(ce_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, logits, (long)ldv, labels, N, V,
                     row_lse, loss_sum_cnt);
  FBL_CHECK_LAUNCH();
  hipLaunchKernelGGL(ce_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, (long)ldv, labels, N, row_lse,
                     loss_sum_cnt);
  FBL_CHECK_LAUNCH();
  return 0;
}
extern "C" int fbl_ce_bwd_rows(const float* logits, int64_t ldv, const int64_t* labels, const int32_t* rows, int R,
                               int V, int Vp, const float* row_lse, const float* loss_sum_cnt, float gscale,
                               const float* gscale_dev, void* dlogits_bf16, void* stream) {
  if (R <= 0) return 0;
  if (Vp < V) return FBL_ERR_ARG;
  hipLaunchKernelGGL(ce_bwd_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, (long)ldv, labels, rows, V,
                     Vp, row_lse, loss_sum_cnt, gscale, gscale_dev, (bf16*)dlogits_bf16);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_gather_rows_bf16(const void* in, int64_t ld, const int32_t* rows, int R, int cols, void* out,
                                    void* stream) {
  if (R <= 0) return 0;
  if (cols % 8 || ld % 8) return FBL_ERR_ALIGN;
  hipLaunchKernelGGL(gather_rows_bf16_kernel, dim3(R), dim3(128), 0, (hipStream_t)stream, (const bf16*)in, (long)ld,
                     rows, cols, (bf16*)out);
  FBL_CHECK_LAUNCH();
  return 0;
}
extern "C" int fbl_scatter_rows_f32(const float* in, const int32_t* rows, int R, int cols, float* out, int64_t ld,
                                    void* stream) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(scatter_rows_f32_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, in, rows, cols, out,
                     (long)ld);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int64_t fbl_sumsq_ws_floats(void) { return SUMSQ_BLOCKS; }
extern "C" int fbl_sumsq(const float* x, int64_t n, float* out_sumsq, float* ws, void* stream) {
  if (n <= 0) return 0;
  if (!ws || !out_sumsq) return FBL_ERR_ARG;
  const int nblk = grid1d(n, 256, SUMSQ_BLOCKS);
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, (long)n, ws);
  FBL_CHECK_LAUNCH();
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nblk, out_sumsq);
  FBL_CHECK_LAUNCH();
  return 0;
}
extern "C" int fbl_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, const float* sumsq, float max_norm,
                             float grad_scale, void* stream) {
  if (n <= 0) return 0;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_flat_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr,
                     beta1, beta2, eps, weight_decay, bc1, bc2, sumsq, max_norm, grad_scale);
  FBL_CHECK_LAUNCH();
  return 0;
}

extern "C" int fbl_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, in, (bf16*)out, (long)n);
  FBL_CHECK_LAUNCH();
  return 0;
}
extern "C" int fbl_dropout_f32(const float* in, float p_drop, uint64_t seed, const uint64_t* seed_dev, float* out_f32,
                               void* out_bf16, int64_t n, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(dropout_f32_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, in, p_drop, seed, seed_dev, out_f32,
                     (bf16*)out_bf16, (long)n);
  FBL_CHECK_LAUNCH();
  return 0;
}
// p[0, bytes) = 0 as a KERNEL: 16-byte stores over the aligned body, byte stores for a ragged head / tail.  Not hipMemsetAsync:
// inside a captured step that becomes a memset node of the graph, the only non-kernel nodes the step would have -- and the buffers
// zeroed this way are all accumulated into by the next kernel (split-K partial sums, the loss accumulator, the position-table
// gradients), so a fill that is not ordered exactly like a kernel turns into garbage or NaN in every gradient.  Replayed steps
// produced non-finite gradients about once in 500 replays while the fills were memset nodes (bench.py `graphed_step`, round 5);
// eager steps never did.
__global__ void zero_kernel(unsigned char* p, long bytes, long head, long body16) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  uint4* b = (uint4*)(p + head);
  const uint4 z = make_uint4(0, 0, 0, 0);
  for (long i = t; i < body16; i += stride) b[i] = z;
  const long tail0 = head + body16 * 16;
  for (long i = t; i < head; i += stride) p[i] = 0;
  for (long i = tail0 + t; i < bytes; i += stride) p[i] = 0;
}
extern "C" int fbl_zero(void* p, int64_t bytes, void* stream) {
  if (bytes <= 0) return 0;
  if (!p) return FBL_ERR_ARG;
  long head = (long)((16 - ((uintptr_t)p & 15)) & 15);
  if (head > bytes) head = bytes;
  const long body16 = (bytes - head) / 16;
  long blocks = (body16 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 4096) blocks = 4096;  // grid-stride: 16 workgroups per CU keep every HBM channel busy
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned char*)p, (long)bytes, head, body16);
  FBL_CHECK_LAUNCH();
  return 0;
}
extern "C" int fbl_heads_to_rows_bf16(const float* src, void* dst_bf16, int E, int nh, int rows, int64_t ld_dst, void* stream) {
  if (E <= 0 || nh <= 0 || rows <= 0) return 0;
  if (!src || !dst_bf16 || ld_dst < (int64_t)nh * 64 || (ld_dst % 8)) return FBL_ERR_ARG;
  if (((uintptr_t)dst_bf16 & 15) || ((uintptr_t)src & 15)) return FBL_ERR_ALIGN;
  hipLaunchKernelGGL(heads_to_rows_bf16_kernel, dim3(grid1d((long)E * nh * rows * 8)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16*)dst_bf16, E, nh, rows, (long)ld_dst);
  FBL_CHECK_LAUNCH();
  return 0;
}
extern "C" int fbl_dropout_sum_f32(const float* x, int64_t n, int64_t key0, int n_slices, const uint64_t* seeds, float p_drop,
                                   const uint64_t* seed_dev, float* out_f32, void* stream) {
  if (n <= 0) return 0;
  if (n_slices < 1 || n_slices > FBL_DROPSUM_MAX_SLICES || !x || !out_f32 || (p_drop > 0.f && !seeds) || p_drop >= 1.f || key0 < 0) return FBL_ERR_ARG;
  DropSumArgs a{};
  for (int s = 0; s < n_slices; ++s) a.seeds[s] = (p_drop > 0.f) ? seeds[s] : 0;
  a.x = x; a.seed_dev = seed_dev; a.out = out_f32; a.n = (long)n; a.key0 = (long)key0; a.ns = n_slices; a.p = p_drop;
  hipLaunchKernelGGL(dropout_sum_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, a);
  FBL_CHECK_LAUNCH();
  return 0;
}
extern "C" int fbl_dropout_bf16(void* inout_bf16, float p_drop, uint64_t seed, const uint64_t* seed_dev, int64_t n,
                                void* stream) {
  if (n <= 0 || p_drop <= 0.f) return 0;
  hipLaunchKernelGGL(dropout_bf16_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, (bf16*)inout_bf16, p_drop,
                     seed, seed_dev, (long)n);
  FBL_CHECK_LAUNCH();
  return 0;
}
