// Device-side helpers shared by the gfx950 kernels of libfbl.  CDNA4 only: wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define FBL_WAVE 64

// Experiment switches (tile-shape overrides, ablations) exist only in builds made with -DFBL_DEBUG_SWITCHES
// (FBL_DEBUG_BUILD=1 python -m frozenbilm_amd.build): the product library reads no environment variable.
#ifdef FBL_DEBUG_SWITCHES
#include <stdlib.h>
#define FBL_ENV_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define FBL_ENV_INT(name, dflt) (dflt)
#endif

#define FBL_CHECK_LAUNCH()                                     \
  do {                                                         \
    hipError_t e__ = hipGetLastError();                        \
    if (e__ != hipSuccess) return (int)e__;                    \
  } while (0)

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }  // RNE (v_cvt_pk_bf16_f32 on gfx950)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7, ~12 VALU ops, no branches) -- the libm erff expansion made the
// GELU epilogues VALU-bound.  e = exp(-u^2) is returned so dGELU can reuse it as the Gaussian density.
__device__ __forceinline__ float fast_erf(float u, float* e_out) {
  const float au = fabsf(u);
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * au);
  const float e = __expf(-au * au);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  *e_out = e;
  return copysignf(1.0f - poly * e, u);
}
__device__ __forceinline__ float gelu_erf(float x) {
  float e;
  return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f, &e));
}
__device__ __forceinline__ float dgelu_erf(float x) {
  float e;  // exp(-x^2/2)
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f, &e));
  return cdf + x * 0.3989422804014327f * e;
}

// gelu(x) and gelu'(x) from one erf / one exp
__device__ __forceinline__ void gelu_and_grad(float x, float* y, float* dy) {
  float e;
  const float cdf = 0.5f * (1.0f + fast_erf(x * 0.70710678118654752f, &e));
  *y = x * cdf;
  *dy = cdf + x * 0.3989422804014327f * e;
}

// Counter-based dropout RNG: keep decision is a pure function of (seed, element index), so backward
// regenerates the mask instead of storing it.  32-bit arithmetic only: integer multiplies are quarter rate on CDNA and a
// 64 x 64 -> 64 bit product is four of them -- the first version of this mixer (three 64-bit multiplies, ~230 issue cycles per
// element and wave) was half of the run time of the HBM-bound row kernels that regenerate a mask per element (LayerNorm
// backward, adapter tail).  Now: index * odd constant, seed folded in by XOR (non-linear against the multiplies around it: two
// sites / steps do not see shifted copies of one sequence), then the two multiply-xorshift rounds of the "lowbias32"
// finaliser with the seed's upper word entering between them.  3 multiplies, ~85 issue cycles.  Restated on the host by
// tests/dropout_replay.py (train-mode parity against the oracle with replayed masks).
__device__ __forceinline__ uint32_t fbl_hash(uint64_t seed, uint64_t idx) {
  const uint32_t hi = (uint32_t)(idx >> 32);
  uint32_t h = (uint32_t)idx * 0x9E3779B1u;
  h ^= (uint32_t)seed;
  h ^= (hi << 13) | (hi >> 19);
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= h >> 15;
  h ^= (uint32_t)(seed >> 32);
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}
// returns the multiplicative factor: 0 (dropped) or 1/(1-p) (kept); p == 0 -> 1
__device__ __forceinline__ float fbl_dropout_scale(uint64_t seed, uint64_t idx, uint32_t thresh, float inv_keep) {
  return (fbl_hash(seed, idx) >= thresh) ? inv_keep : 0.0f;
}
// Effective seed of a dropout site: the launch-time value plus an optional device-resident 64-bit word.  With the word a
// captured launch (hipGraph replay) draws a new mask every time the host advances it -- the seed itself is a kernel argument
// and frozen into the graph.  NULL: the launch-time value alone.
__device__ __forceinline__ uint64_t fbl_seed(uint64_t seed, const uint64_t* seed_dev) {
#ifdef FBL_NO_SEED_DEV  // (measurement builds only: what the device word costs a kernel's code generation)
  return seed;
#else
  return seed_dev ? seed + *seed_dev : seed;
#endif
}
// host+device: p -> 32-bit threshold (drop iff hash < thresh)
static inline __host__ __device__ uint32_t fbl_drop_thresh(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}
