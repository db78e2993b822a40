// Shared host/device helpers for libimmoco_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/immoco_hip.h"

// A/B and diagnostics switches (IMMOCO_* environment variables: kernel variants, scheduling experiments, and a few
// timing-only ablations that compute WRONG results).  Only a library built with -DIMMOCO_DIAG (`make diag` ->
// libimmoco_hip_diag.so, never loaded by the package unless IMMOCO_LIB_PATH points at it) reads them; the shipped
// libimmoco_hip.so reads no environment variable at all and always runs the defaults.
#ifdef IMMOCO_DIAG
static inline const char* immoco_diag_env(const char* name) { return getenv(name); }
#else
static inline const char* immoco_diag_env(const char*) { return nullptr; }
#endif

namespace immoco {

// ---- error plumbing -------------------------------------------------------
void set_error(const char* fmt, ...);

#define IMMOCO_CHECK_HIP(expr)                                                              \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      immoco::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return IMMOCO_E_HIP;                                                                  \
    }                                                                                       \
  } while (0)

#define IMMOCO_REQUIRE(cond, ...)         \
  do {                                    \
    if (!(cond)) {                        \
      immoco::set_error(__VA_ARGS__);     \
      return IMMOCO_E_INVALID;            \
    }                                     \
  } while (0)

#define IMMOCO_LAUNCH_CHECK() IMMOCO_CHECK_HIP(hipGetLastError())

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- level table passed by value to kernels --------------------------------
struct Levels {
  uint32_t offset[IMMOCO_MAX_LEVELS + 1];
  uint32_t res[IMMOCO_MAX_LEVELS];
  uint32_t size[IMMOCO_MAX_LEVELS];
  float scale[IMMOCO_MAX_LEVELS];
  uint32_t hashed;  // bit l set: level l uses the prime hash
  uint32_t pow2;    // bit l set: size[l] is a power of two (use & instead of %)
  int32_t n_levels;
  int32_t dims;
};

int build_levels(const immoco_grid_cfg* cfg, Levels* out);  // host; validates cfg

// ---- device: tiny-cuda-nn grid semantics --------------
#define IMMOCO_PRIME1 2654435761u
#define IMMOCO_PRIME2 805459861u

// pos_fract(): pos = fmaf(scale, x, 0.5); cell = floor(pos) as wrapped uint32.
__device__ __forceinline__ void pos_fract(float x, float scale, uint32_t& cell, float& frac) {
  float pos = fmaf(scale, x, 0.5f);
  float fl = floorf(pos);
  cell = (uint32_t)(int32_t)fl;
  frac = pos - fl;
}

template <int D>
__device__ __forceinline__ uint32_t grid_index(const uint32_t (&c)[D], uint32_t size, uint32_t res,
                                               bool hashed, bool pow2) {
  uint32_t idx;
  if (hashed) {
    idx = c[0];
    if (D > 1) idx ^= c[1] * IMMOCO_PRIME1;
    if (D > 2) idx ^= c[2] * IMMOCO_PRIME2;
  } else {
    idx = 0;
    uint32_t stride = 1;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (stride <= size) {
        idx += c[d] * stride;
        stride *= res;
      }
    }
  }
  return pow2 ? (idx & (size - 1u)) : (idx % size);
}

// Un-contracted fp32 ops (hipcc's __fmul_rn/__fadd_rn are plain operators and get fused into
// FMAs under the default -ffp-contract=fast): used where bit-identity with the oracle matters.
__device__ __forceinline__ float mul_nc(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_nc(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_nc(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// PCG output hash shared with oracle.pcg_hash_u32.
__host__ __device__ __forceinline__ uint32_t pcg_hash(uint32_t x) {
  uint32_t state = x * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}

// n / d for index arithmetic.  hipcc expands a 64-bit (and even a 32-bit) integer division into a long
// multiply/correct sequence - ~100 VALU instructions per `(i / W) % H` on an int64 index, which made the
// per-thread index decomposition the largest VALU cost of the gather and pointwise kernels (round 2: the
// encode forward issued ~350 VALU instructions per (point, level), most of them three 64-bit divisions).
// For n < 2^24 the quotient comes from ONE correctly rounded float reciprocal: float(n) is exact, the product's
// relative error is below 2^-23, so the truncated quotient is off by at most one for d >= 2 and one correction
// step each way makes it exact (d = 1 is handled by the caller's layouts: a stride of 1 is never divided by).
__device__ __forceinline__ uint32_t fast_div(uint32_t n, uint32_t d) {
  if (n < (1u << 24) && d < (1u << 24)) {
    uint32_t q = (uint32_t)((float)n * (1.0f / (float)d));
    q -= (q * d > n) ? 1u : 0u;
    q += ((q + 1u) * d <= n) ? 1u : 0u;
    return q;
  }
  return n / d;
}

// wave64 sum via DPP-free shuffles (width 64).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace immoco
