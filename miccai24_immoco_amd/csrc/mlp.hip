// Bias-free one-hidden-layer MLP 32 -> HID -> 2, forward and backward, fp32.
// Replaces the network half of tinycudann.NetworkWithInputEncoding
// (CutlassMLP 32->256 ReLU->2 and FullyFusedMLP 32->64 Tanh->2; reference
// src/models/immoco.py:11-25,60-65).  W1 [HID][32], W2 [n_out_padded][HID]
// row-major, only rows 0..1 of W2 are live (tcnn pads the output width).
//
// v1 mapping: lane = point.  Weights are read with wave-uniform addresses, so
// hipcc turns them into scalar loads (s_load_dwordx8/16) feeding v_fmac with an
// SGPR operand: no LDS traffic for weights in the forward / recompute / d-input
// products.  The weight-gradient outer products need a reduction over points;
// they are staged through LDS ([point][hidden] image, padded) and accumulated
// with lane = hidden unit.
#include <stdlib.h>

#include "kernels.hpp"

namespace immoco {

int check_mlp_cfg(const immoco_mlp_cfg* cfg) {
  IMMOCO_REQUIRE(cfg != nullptr, "mlp cfg is NULL");
  IMMOCO_REQUIRE(cfg->n_in == 32, "mlp n_in must be 32 (16 levels x 2 features), got %d", cfg->n_in);
  IMMOCO_REQUIRE(cfg->n_hidden == 64 || cfg->n_hidden == 256, "mlp n_hidden must be 64 or 256, got %d",
                 cfg->n_hidden);
  IMMOCO_REQUIRE(cfg->n_out == 2 && cfg->n_out_padded >= 2, "mlp n_out must be 2");
  IMMOCO_REQUIRE(cfg->activation == IMMOCO_ACT_RELU || cfg->activation == IMMOCO_ACT_TANH,
                 "unknown activation %d", cfg->activation);
  return IMMOCO_OK;
}

template <int ACT>
__device__ __forceinline__ float act_fwd(float pre) {
  return ACT == IMMOCO_ACT_RELU ? fmaxf(pre, 0.0f) : tanhf(pre);
}
// derivative given pre-activation and activation value
template <int ACT>
__device__ __forceinline__ float act_bwd(float pre, float h) {
  return ACT == IMMOCO_ACT_RELU ? (pre > 0.0f ? 1.0f : 0.0f) : 1.0f - h * h;
}

__device__ __forceinline__ void load_enc(const float* in, int64_t p, int64_t ps, int64_t ls,
                                         float (&e)[32]) {
#pragma unroll
  for (int l = 0; l < 16; ++l) {
    const float2 v = *reinterpret_cast<const float2*>(in + p * ps + (int64_t)l * ls);
    e[2 * l] = v.x;
    e[2 * l + 1] = v.y;
  }
}

template <int HID, int ACT>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(const float* __restrict__ in, int64_t ps, int64_t ls,
                                                      int64_t n, const float* __restrict__ w1,
                                                      const float* __restrict__ w2, float* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  float e[32];
  load_enc(in, p, ps, ls, e);
  float o0 = 0.f, o1 = 0.f;
#pragma unroll 8
  for (int j = 0; j < HID; ++j) {
    float pre = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) pre = fmaf(w1[j * 32 + k], e[k], pre);
    const float h = act_fwd<ACT>(pre);
    o0 = fmaf(w2[j], h, o0);
    o1 = fmaf(w2[HID + j], h, o1);
  }
  *reinterpret_cast<float2*>(out + p * 2) = make_float2(o0, o1);
}

// ---------------------------------------------------------------------------
// backward.  Block = 128 threads = 2 waves; each wave owns batches of 64 points.
constexpr int BWD_THREADS = 128;
constexpr int A_LD = 65;  // padded leading dimension of the [point][hidden] LDS image

template <int HID, int ACT>
__global__ __launch_bounds__(BWD_THREADS) void mlp_bwd_kernel(
    const float* in /* may alias din */, int64_t ps, int64_t ls, int64_t n, const float* __restrict__ w1,
    const float* __restrict__ w2, const float* __restrict__ dout, float* din,
    float* __restrict__ dw1, float* __restrict__ dw2, int64_t n_batches, int64_t dout_plane) {
  constexpr int NCH = HID / 64;
  __shared__ float lds[2][64 * A_LD + 32 * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* __restrict__ A = lds[wave];
  float* __restrict__ E = A + 64 * A_LD;
  const int64_t per_iter = (int64_t)gridDim.x * 2;
  const int64_t n_iter = (n_batches + per_iter - 1) / per_iter;

  for (int64_t it = 0; it < n_iter; ++it) {
    const int64_t b = it * per_iter + (int64_t)blockIdx.x * 2 + wave;
    const int64_t p = b * 64 + lane;
    const bool valid = (b < n_batches) && (p < n);
    float e[32];
    float d0 = 0.f, d1 = 0.f;
    if (valid) {
      load_enc(in, p, ps, ls, e);
      if (dout_plane) {
        d0 = dout[p];
        d1 = dout[dout_plane + p];
      } else {
        const float2 d = *reinterpret_cast<const float2*>(dout + p * 2);
        d0 = d.x;
        d1 = d.y;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 32; ++k) e[k] = 0.f;
    }
    __syncthreads();  // previous iteration's readers of E/A are done
#pragma unroll
    for (int k = 0; k < 32; ++k) E[k * 64 + lane] = e[k];
    float de[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) de[k] = 0.f;

#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
      float dpre[64];
      // 1. recompute hidden chunk, stage h, keep dpre in registers
#pragma unroll
      for (int jj = 0; jj < 64; ++jj) {
        const int j = ch * 64 + jj;
        float pre = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) pre = fmaf(w1[j * 32 + k], e[k], pre);
        const float h = act_fwd<ACT>(pre);
        const float dh = fmaf(w2[j], d0, w2[HID + j] * d1);
        dpre[jj] = dh * act_bwd<ACT>(pre, h);
        A[lane * A_LD + jj] = h;
      }
      __syncthreads();
      // 2. dW2[o][j] += sum_p dout[p][o] * h[p][j]   (lane = hidden unit j)
      float a20 = 0.f, a21 = 0.f;
#pragma unroll
      for (int pp = 0; pp < 64; ++pp) {
        const float hp = A[pp * A_LD + lane];
        a20 = fmaf(__shfl(d0, pp, 64), hp, a20);
        a21 = fmaf(__shfl(d1, pp, 64), hp, a21);
      }
      unsafeAtomicAdd(dw2 + ch * 64 + lane, a20);
      unsafeAtomicAdd(dw2 + HID + ch * 64 + lane, a21);
      __syncthreads();
      // 3. stage dpre; d enc += W1^T dpre
#pragma unroll
      for (int jj = 0; jj < 64; ++jj) A[lane * A_LD + jj] = dpre[jj];
#pragma unroll
      for (int jj = 0; jj < 64; ++jj) {
        const int j = ch * 64 + jj;
#pragma unroll
        for (int k = 0; k < 32; ++k) de[k] = fmaf(w1[j * 32 + k], dpre[jj], de[k]);
      }
      __syncthreads();
      // 4. dW1[j][k] += sum_p dpre[p][j] * enc[p][k]   (lane = hidden unit j)
      float acc[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) acc[k] = 0.f;
#pragma unroll 4
      for (int pp = 0; pp < 64; ++pp) {
        const float dp = A[pp * A_LD + lane];
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k] = fmaf(dp, E[k * 64 + pp], acc[k]);
      }
      __syncthreads();
      // 5. flush through LDS so that each atomic wave-instruction covers 256 contiguous bytes
#pragma unroll
      for (int k = 0; k < 32; ++k) A[lane * 33 + k] = acc[k];
      __syncthreads();
#pragma unroll 4
      for (int r = 0; r < 32; ++r) {
        const int idx = r * 64 + lane;  // element (j = idx/32, k = idx%32) of this chunk's [64][32] tile
        unsafeAtomicAdd(dw1 + (size_t)ch * 2048 + idx, A[(idx >> 5) * 33 + (idx & 31)]);
      }
      __syncthreads();
    }
    if (valid) {
#pragma unroll
      for (int l = 0; l < 16; ++l)
        *reinterpret_cast<float2*>(din + p * ps + (int64_t)l * ls) = make_float2(de[2 * l], de[2 * l + 1]);
    }
  }
}

// IMMOCO_MLP_IMPL=valu selects the fp32 VALU kernels of this file (kept for A/B measurements);
// the default is the matrix-core implementation in mlp_mfma.hip.  valu-fwd / valu-bwd / valu-image /
// valu-motion restrict the switch to one direction or one network (hidden width 256 / 64).
static bool use_valu_impl(bool bwd, int n_hidden) {
  static const unsigned v = [] {
    const char* e = immoco_diag_env("IMMOCO_MLP_IMPL");
    if (!e) return 0u;
    if (strcmp(e, "valu") == 0) return 15u;
    if (strcmp(e, "valu-fwd") == 0) return 5u;
    if (strcmp(e, "valu-bwd") == 0) return 10u;
    if (strcmp(e, "valu-image") == 0) return 3u;
    if (strcmp(e, "valu-motion") == 0) return 12u;
    return 0u;
  }();   // bit 0 image forward, 1 image backward, 2 motion forward, 3 motion backward
  return (v >> ((n_hidden == 64 ? 2 : 0) + (bwd ? 1 : 0))) & 1u;
}

int launch_mlp_fwd(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                   const float* w1, const float* w2, float* out, hipStream_t st) {
  if (n == 0) return IMMOCO_OK;
  if (!use_valu_impl(false, cfg.n_hidden)) return launch_mlp_fwd_mfma(cfg, in, ps, ls, n, w1, w2, out, st);
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
  const unsigned grid = (unsigned)cdiv(n, 256);
#define IMMOCO_FWD(H, A) mlp_fwd_kernel<H, A><<<grid, 256, 0, st>>>(in, ps, ls, n, w1, w2, out)
  if (cfg.n_hidden == 64 && cfg.activation == IMMOCO_ACT_TANH) IMMOCO_FWD(64, IMMOCO_ACT_TANH);
  else if (cfg.n_hidden == 64) IMMOCO_FWD(64, IMMOCO_ACT_RELU);
  else if (cfg.activation == IMMOCO_ACT_TANH) IMMOCO_FWD(256, IMMOCO_ACT_TANH);
  else IMMOCO_FWD(256, IMMOCO_ACT_RELU);
#undef IMMOCO_FWD
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_mlp_bwd(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                   const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                   hipStream_t st, int64_t dout_plane) {
  if (n == 0) return IMMOCO_OK;
  if (!use_valu_impl(true, cfg.n_hidden)) return launch_mlp_bwd_mfma(cfg, in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
  const int64_t n_batches = cdiv(n, 64);
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_batches, 2), 4096);
#define IMMOCO_BWD(H, A) \
  mlp_bwd_kernel<H, A><<<grid, BWD_THREADS, 0, st>>>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, n_batches, dout_plane)
  if (cfg.n_hidden == 64 && cfg.activation == IMMOCO_ACT_TANH) IMMOCO_BWD(64, IMMOCO_ACT_TANH);
  else if (cfg.n_hidden == 64) IMMOCO_BWD(64, IMMOCO_ACT_RELU);
  else if (cfg.activation == IMMOCO_ACT_TANH) IMMOCO_BWD(256, IMMOCO_ACT_TANH);
  else IMMOCO_BWD(256, IMMOCO_ACT_RELU);
#undef IMMOCO_BWD
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

}  // namespace immoco

using namespace immoco;

extern "C" int immoco_mlp_fwd(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                              int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                              float* out, void* stream) {
  int rc = check_mlp_cfg(cfg);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (in && w1 && w2 && out)), "mlp_fwd: NULL buffer");
  return launch_mlp_fwd(*cfg, in, in_point_stride, in_level_stride, n, w1, w2, out, as_stream(stream));
}

extern "C" int immoco_mlp_bwd(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                              int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                              const float* dout, float* din, float* dw1, float* dw2, void* stream) {
  int rc = check_mlp_cfg(cfg);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (in && w1 && w2 && dout && din && dw1 && dw2)), "mlp_bwd: NULL buffer");
  return launch_mlp_bwd(*cfg, in, in_point_stride, in_level_stride, n, w1, w2, dout, din, dw1, dw2,
                        as_stream(stream), 0);
}

extern "C" int immoco_mlp_bwd_split(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                                    int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                                    const float* dout, float* din, float* dw1, float* dw2, void* stream) {
  int rc = check_mlp_cfg(cfg);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (in && w1 && w2 && dout && din && dw1 && dw2)), "mlp_bwd_split: NULL buffer");
  IMMOCO_REQUIRE(cfg->n_hidden == 256, "mlp_bwd_split: n_hidden must be 256 (got %d)", cfg->n_hidden);
  IMMOCO_REQUIRE(in != din, "mlp_bwd_split: din must not alias in (the dW kernel reads the encoding after din is written)");
  if ((rc = launch_mlp_bwd_denc(*cfg, in, in_point_stride, in_level_stride, n, w1, w2, dout, din, as_stream(stream), 0)))
    return rc;
  return launch_mlp_bwd_dw(*cfg, in, in_point_stride, in_level_stride, n, w1, w2, dout, dw1, dw2, as_stream(stream), 0);
}
