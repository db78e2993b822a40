// torch.optim.Adam (betas 0.9/0.999, eps 1e-8, no weight decay, no amsgrad) as the
// reference uses it (src/models/immoco.py:149-154,166,175), restated from the
// single-tensor torch implementation:
//   m = lerp(m, g, 1-b1); v = b2*v + (1-b2)*g*g
//   p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// HBM-bound: 16-byte loads/stores, grid-stride, 28 B/param (+4 B when the fused
// zero_grad write is enabled).
#include <hip/hip_fp16.h>

#include "kernels.hpp"

namespace immoco {

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float step_size, float bc2_sqrt,
                                         float b1, float b2, float eps) {
  m = m + (g - m) * (1.0f - b1);
  v = v * b2 + (1.0f - b2) * g * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

template <bool SCHED>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g, int n_gparts,
                                                   int64_t g_stride, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, int64_t zero_limit,
                                                   float step_size, float bc2_sqrt,
                                                   const float* __restrict__ sched,
                                                   const int32_t* __restrict__ iter_dev, float b1, float b2,
                                                   float eps, __half* __restrict__ shadow, int64_t shadow_begin) {
  if (SCHED) {
    const int it = *iter_dev;
    step_size = sched[2 * it];
    bc2_sqrt = sched[2 * it + 1];
  }
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = reinterpret_cast<const float4*>(g)[i];
    for (int q = 1; q < n_gparts; ++q) {  // partial gradient tables of the transposed-index backward
      const float4 gq = reinterpret_cast<const float4*>(g + q * g_stride)[i];
      gg.x += gq.x;
      gg.y += gq.y;
      gg.z += gq.z;
      gg.w += gq.w;
    }
    float4 mm = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, step_size, bc2_sqrt, b1, b2, eps);
    adam_one(pp.y, gg.y, mm.y, vv.y, step_size, bc2_sqrt, b1, b2, eps);
    adam_one(pp.z, gg.z, mm.z, vv.z, step_size, bc2_sqrt, b1, b2, eps);
    adam_one(pp.w, gg.w, mm.w, vv.w, step_size, bc2_sqrt, b1, b2, eps);
    reinterpret_cast<float4*>(p)[i] = pp;
    if (shadow && 4 * i >= shadow_begin) {  // shadow_begin is a multiple of 4
      __half2* sh = reinterpret_cast<__half2*>(shadow + (4 * i - shadow_begin));
      sh[0] = __floats2half2_rn(pp.x, pp.y);
      sh[1] = __floats2half2_rn(pp.z, pp.w);
    }
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (SCHED && 4 * i < zero_limit)  // fused zero_grad (skipped where the producer overwrites)
      for (int q = 0; q < n_gparts; ++q) reinterpret_cast<float4*>(g + q * g_stride)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // tail
  const int64_t t = n4 * 4 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t < n) {
    float gt = g[t];
    for (int q = 1; q < n_gparts; ++q) gt += g[q * g_stride + t];
    adam_one(p[t], gt, m[t], v[t], step_size, bc2_sqrt, b1, b2, eps);
    if (shadow && t >= shadow_begin) shadow[t - shadow_begin] = __float2half_rn(p[t]);
    if (SCHED && t < zero_limit)
      for (int q = 0; q < n_gparts; ++q) g[q * g_stride + t] = 0.f;
  }
}

// Touched-blocks Adam: workgroup b < n_wblocks handles 1024 float4 of the MLP weights, the others one slot
// block (<= 2048 slots = 1024 float4) of the table.  Table slots outside the listed blocks never receive a
// gradient (the block list is a constant of the lattice, csr.hip), so g = m = v = 0 there for ever and
// torch's update is exactly 0: skipping them is the "touched-only Adam" of SURVEY a14.
__global__ __launch_bounds__(256) void adam_blocks_kernel(float* __restrict__ p, float* __restrict__ g, int n_gparts,
                                                          int64_t g_stride, float* __restrict__ m,
                                                          float* __restrict__ v, int64_t n_w4, uint32_t n_wblocks,
                                                          const uint2* __restrict__ blocks,
                                                          const float* __restrict__ sched,
                                                          const int32_t* __restrict__ iter_dev, int iter_off,
                                                          float b1, float b2, float eps,
                                                          __half* __restrict__ shadow, int nt) {
  const int it = *iter_dev + iter_off;   // iter_off = -1: the deferred update of the PREVIOUS iteration (solver.hip)
  const float step_size = sched[2 * it], bc2_sqrt = sched[2 * it + 1];
  int64_t i0, cnt;
  bool zero;
  if (blockIdx.x < n_wblocks) {
    i0 = (int64_t)blockIdx.x * 1024;
    cnt = n_w4 - i0 < 1024 ? n_w4 - i0 : 1024;
    zero = true;
  } else {
    const uint2 b = blocks[blockIdx.x - n_wblocks];
    i0 = n_w4 + (int64_t)b.x / 2;            // 2 floats per slot, 4 per float4
    cnt = (b.y & 0x7FFFFFFFu) / 2;
    zero = b.y >> 31;
  }
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  auto ld = [&](const float* base, int64_t i) {   // streamed once per iteration: non-temporal
    if (!nt) return reinterpret_cast<const float4*>(base)[i];
    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base) + i);
    return make_float4(t.x, t.y, t.z, t.w);
  };
  auto st = [&](float* base, int64_t i, const float4& x) {
    if (!nt) {
      reinterpret_cast<float4*>(base)[i] = x;
      return;
    }
    f32x4 t;
    t.x = x.x;
    t.y = x.y;
    t.z = x.z;
    t.w = x.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(base) + i);
  };
  for (int64_t k = threadIdx.x; k < cnt; k += 256) {
    const int64_t i = i0 + k;
    float4 pp = reinterpret_cast<float4*>(p)[i];
    float4 gg = ld(g, i);
    for (int q = 1; q < n_gparts; ++q) {
      const float4 gq = ld(g + q * g_stride, i);
      gg.x += gq.x;
      gg.y += gq.y;
      gg.z += gq.z;
      gg.w += gq.w;
    }
    float4 mm = ld(m, i);
    float4 vv = ld(v, i);
    adam_one(pp.x, gg.x, mm.x, vv.x, step_size, bc2_sqrt, b1, b2, eps);
    adam_one(pp.y, gg.y, mm.y, vv.y, step_size, bc2_sqrt, b1, b2, eps);
    adam_one(pp.z, gg.z, mm.z, vv.z, step_size, bc2_sqrt, b1, b2, eps);
    adam_one(pp.w, gg.w, mm.w, vv.w, step_size, bc2_sqrt, b1, b2, eps);
    if (nt & 2) st(p, i, pp);
    else reinterpret_cast<float4*>(p)[i] = pp;
    if (shadow && i >= n_w4) {
      __half2* sh = reinterpret_cast<__half2*>(shadow + 4 * (i - n_w4));
      sh[0] = __floats2half2_rn(pp.x, pp.y);
      sh[1] = __floats2half2_rn(pp.z, pp.w);
    }
    st(m, i, mm);
    st(v, i, vv);
    if (zero)
      for (int q = 0; q < n_gparts; ++q) reinterpret_cast<float4*>(g + q * g_stride)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

static unsigned adam_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n / 4 + 1, 256), 2048)); }

int launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float step_size, float bc2_sqrt,
                float beta1, float beta2, float eps, hipStream_t st) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                     ((uintptr_t)v % 16) == 0, "adam: buffers must be 16-byte aligned");
  adam_kernel<false><<<adam_grid(n), 256, 0, st>>>(p, const_cast<float*>(g), 1, 0, m, v, n, n, step_size, bc2_sqrt,
                                                   nullptr, nullptr, beta1, beta2, eps, nullptr, 0);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_adam_sched(float* p, float* g, int n_gparts, int64_t g_stride, float* m, float* v, int64_t n,
                      int64_t zero_limit, const float* sched, const int32_t* iter_dev, float beta1, float beta2,
                      float eps, hipStream_t st, void* shadow, int64_t shadow_begin) {
  IMMOCO_REQUIRE(!shadow || (shadow_begin % 4) == 0, "adam: shadow_begin must be a multiple of 4");
  IMMOCO_REQUIRE(n_gparts >= 1 && (g_stride % 4) == 0, "adam: partial gradient stride must be a multiple of 4");
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                     ((uintptr_t)v % 16) == 0, "adam: buffers must be 16-byte aligned");
  adam_kernel<true><<<adam_grid(n), 256, 0, st>>>(p, g, n_gparts, g_stride, m, v, n, zero_limit, 0.f, 1.f, sched, iter_dev, beta1,
                                                  beta2, eps, reinterpret_cast<__half*>(shadow), shadow_begin);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_adam_blocks(float* p, float* g, int n_gparts, int64_t g_stride, float* m, float* v, int64_t n_w,
                       const uint2* blocks, uint32_t n_blocks, const float* sched, const int32_t* iter_dev,
                       float beta1, float beta2, float eps, hipStream_t st, void* shadow, int iter_off) {
  IMMOCO_REQUIRE((n_w % 4) == 0, "adam: the MLP weight count must be a multiple of 4 (got %lld)", (long long)n_w);
  IMMOCO_REQUIRE(n_gparts >= 1 && (g_stride % 4) == 0, "adam: partial gradient stride must be a multiple of 4");
  IMMOCO_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                     ((uintptr_t)v % 16) == 0, "adam: buffers must be 16-byte aligned");
  const uint32_t n_wblocks = (uint32_t)cdiv(n_w / 4, 1024);
  if (n_wblocks + n_blocks == 0) return IMMOCO_OK;
  // Gradients, moments and parameters stream through once per iteration: non-temporal loads and stores (motion grid
  // 0.0775 -> 0.0706 ms, graph iteration -0.5 %).  A/B switch (environment, read once): IMMOCO_ADAM_NT = 0 | 1 (g, m,
  // v) | 3 (and the parameter store; default).
  static const int adam_nt = [] { const char* e = immoco_diag_env("IMMOCO_ADAM_NT"); return e ? atoi(e) : 3; }();
  adam_blocks_kernel<<<n_wblocks + n_blocks, 256, 0, st>>>(p, g, n_gparts, g_stride, m, v, n_w / 4, n_wblocks, blocks,
                                                          sched, iter_dev, iter_off, beta1, beta2, eps,
                                                          reinterpret_cast<__half*>(shadow), adam_nt);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

}  // namespace immoco

using namespace immoco;

extern "C" int immoco_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                float beta2, float eps, int32_t step, void* stream) {
  IMMOCO_REQUIRE(n >= 0 && step >= 1 && (n == 0 || (p && g && m && v)), "adam_step: bad argument");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  return launch_adam(p, g, m, v, n, (float)((double)lr / bc1), (float)sqrt(bc2), beta1, beta2, eps,
                     as_stream(stream));
}
