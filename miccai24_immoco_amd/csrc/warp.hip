// Bilinear warp of ONE complex image at nM dense sampling grids, forward and
// backward: F.grid_sample(mode="bilinear", padding_mode="zeros",
// align_corners=False) as used by reference src/models/immoco.py:91,97-107
// (ATen grid_sampler_2d semantics: x_pix = ((g+1)*W - 1)/2, taps nw/ne/sw/se,
// out-of-bounds taps contribute 0; d/dgrid scaled by W/2, H/2).
#include <stdlib.h>

#include "kernels.hpp"

namespace immoco {

struct Taps {
  int x0, y0;
  float nw, ne, sw, se;  // weights
  float tx0, tx1, ty0, ty1;  // (x1-x),(x-x0),(y1-y),(y-y0)
};

__device__ __forceinline__ Taps make_taps(float gx, float gy, int H, int W) {
  Taps t;
  const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f;
  const float iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  const float fx = floorf(ix), fy = floorf(iy);
  // clamp the integer part so that wildly out-of-range grids cannot overflow int
  t.x0 = (int)fminf(fmaxf(fx, -2.f), (float)W + 1.f);
  t.y0 = (int)fminf(fmaxf(fy, -2.f), (float)H + 1.f);
  t.tx0 = (fx + 1.f) - ix;
  t.tx1 = ix - fx;
  t.ty0 = (fy + 1.f) - iy;
  t.ty1 = iy - fy;
  t.nw = t.tx0 * t.ty0;
  t.ne = t.tx1 * t.ty0;
  t.sw = t.tx0 * t.ty1;
  t.se = t.tx1 * t.ty1;
  return t;
}

__device__ __forceinline__ bool inb(int y, int x, int H, int W) {
  return (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
}

__device__ __forceinline__ float2 ld_or_zero(const float2* __restrict__ img, int y, int x, int H, int W) {
  return inb(y, x, H, W) ? img[(size_t)y * W + x] : make_float2(0.f, 0.f);
}

__device__ __forceinline__ float2 sample(const float2* __restrict__ img, const Taps& t, int H, int W) {
  const float2 a = ld_or_zero(img, t.y0, t.x0, H, W), b = ld_or_zero(img, t.y0, t.x0 + 1, H, W);
  const float2 c = ld_or_zero(img, t.y0 + 1, t.x0, H, W), d = ld_or_zero(img, t.y0 + 1, t.x0 + 1, H, W);
  // ATen accumulation order nw, ne, sw, se
  float2 o;
  o.x = a.x * t.nw;
  o.x += b.x * t.ne;
  o.x += c.x * t.sw;
  o.x += d.x * t.se;
  o.y = a.y * t.nw;
  o.y += b.y * t.ne;
  o.y += c.y * t.sw;
  o.y += d.y * t.se;
  return o;
}

// gradient wrt image (atomic scatter) and wrt the sampling position (returned, in grid units)
__device__ __forceinline__ float2 sample_bwd(const float2* __restrict__ img, float* __restrict__ dimg,
                                             const Taps& t, float2 go, int H, int W) {
  const int x0 = t.x0, y0 = t.y0;
  const float2 a = ld_or_zero(img, y0, x0, H, W), b = ld_or_zero(img, y0, x0 + 1, H, W);
  const float2 c = ld_or_zero(img, y0 + 1, x0, H, W), d = ld_or_zero(img, y0 + 1, x0 + 1, H, W);
  if (dimg) {
    if (inb(y0, x0, H, W)) {
      unsafeAtomicAdd(dimg + 2 * ((size_t)y0 * W + x0), t.nw * go.x);
      unsafeAtomicAdd(dimg + 2 * ((size_t)y0 * W + x0) + 1, t.nw * go.y);
    }
    if (inb(y0, x0 + 1, H, W)) {
      unsafeAtomicAdd(dimg + 2 * ((size_t)y0 * W + x0 + 1), t.ne * go.x);
      unsafeAtomicAdd(dimg + 2 * ((size_t)y0 * W + x0 + 1) + 1, t.ne * go.y);
    }
    if (inb(y0 + 1, x0, H, W)) {
      unsafeAtomicAdd(dimg + 2 * ((size_t)(y0 + 1) * W + x0), t.sw * go.x);
      unsafeAtomicAdd(dimg + 2 * ((size_t)(y0 + 1) * W + x0) + 1, t.sw * go.y);
    }
    if (inb(y0 + 1, x0 + 1, H, W)) {
      unsafeAtomicAdd(dimg + 2 * ((size_t)(y0 + 1) * W + x0 + 1), t.se * go.x);
      unsafeAtomicAdd(dimg + 2 * ((size_t)(y0 + 1) * W + x0 + 1) + 1, t.se * go.y);
    }
  }
  // ATen grid_sampler_2d_backward: sum over channels (re, im)
  float gix = 0.f, giy = 0.f;
  const float va = a.x * go.x + a.y * go.y, vb = b.x * go.x + b.y * go.y;
  const float vc = c.x * go.x + c.y * go.y, vd = d.x * go.x + d.y * go.y;
  gix -= va * t.ty0;
  giy -= va * t.tx0;
  gix += vb * t.ty0;
  giy -= vb * t.tx1;
  gix -= vc * t.ty1;
  giy += vc * t.tx0;
  gix += vd * t.ty1;
  giy += vd * t.tx1;
  return make_float2(gix * (0.5f * (float)W), giy * (0.5f * (float)H));
}

__global__ __launch_bounds__(256) void warp_fwd_kernel(const float2* __restrict__ img,
                                                       const float2* __restrict__ grids, int64_t n, int H,
                                                       int W, float2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float2 g = grids[i];
  out[i] = sample(img, make_taps(g.x, g.y, H, W), H, W);
}

__global__ __launch_bounds__(256) void warp_bwd_kernel(const float2* __restrict__ img,
                                                       const float2* __restrict__ grids,
                                                       const float2* __restrict__ dout, int64_t n, int H,
                                                       int W, float* __restrict__ dimg,
                                                       float2* __restrict__ dgrids) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float2 g = grids[i];
  dgrids[i] = sample_bwd(img, dimg, make_taps(g.x, g.y, H, W), dout[i], H, W);
}

// solver: o (motion MLP output) -> t = tanh(o); grid = t + identity (immoco.py:93-95);
// warped image, pre-multiplied by the FFT checkerboard sign, into fft slot 1+m.
__global__ __launch_bounds__(256) void motion_warp_fwd_kernel(const float2* __restrict__ img,
                                                              const float2* __restrict__ o,
                                                              const float* __restrict__ xs,
                                                              const float* __restrict__ ys, int64_t n, int H,
                                                              int W, float2* __restrict__ t_out,
                                                              float2* __restrict__ slots) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % W), r = (int)((i / W) % H);
  const float2 ov = o[i];
  const float2 t = make_float2(tanhf(ov.x), tanhf(ov.y));
  t_out[i] = t;
  float2 v = sample(img, make_taps(t.x + xs[c], t.y + ys[r], H, W), H, W);
  const float s = ((r + c) & 1) ? -1.f : 1.f;
  slots[i] = make_float2(v.x * s, v.y * s);
}

// Backward of the fused warp.  The image gradient is accumulated with float atomics into a PLANAR
// buffer dpl[0..P) = d/dRe, dpl[P..2P) = d/dIm, so that one atomic wave-instruction covers 256
// contiguous bytes (the fast shape; interleaved complex halves that).  Neighbouring lanes
// (neighbouring pixels of one row) nearly always sample neighbouring source columns: the right tap
// column of lane l is the left tap column of lane l+1.  Lanes exchange that column through
// shuffles and each source pixel is added once per pair, which halves the atomics again
// (rocprof before: 5.5 M atomic requests, 0.29 ms per iteration).
__global__ __launch_bounds__(256) void motion_warp_bwd_kernel(const float2* __restrict__ img,
                                                              const float2* __restrict__ t_in,
                                                              const float* __restrict__ xs,
                                                              const float* __restrict__ ys,
                                                              const float2* __restrict__ adj, int64_t n,
                                                              int H, int W, float* __restrict__ dpl,
                                                              float2* __restrict__ d_o) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool active = i < n;
  const int64_t P = (int64_t)H * W;
  int x0 = -1000, y0 = -1000;
  float2 L0 = make_float2(0.f, 0.f), L1 = L0, R0 = L0, R1 = L0;
  if (active) {
    const int c = (int)(i % W), r = (int)((i / W) % H);
    const float2 t = t_in[i];
    const float s = ((r + c) & 1) ? -1.f : 1.f;
    const float2 a = adj[i];
    const float2 go = make_float2(a.x * s, a.y * s);
    const Taps tp = make_taps(t.x + xs[c], t.y + ys[r], H, W);
    const float2 dg = sample_bwd(img, nullptr, tp, go, H, W);
    d_o[i] = make_float2(dg.x * (1.f - t.x * t.x), dg.y * (1.f - t.y * t.y));
    x0 = tp.x0;
    y0 = tp.y0;
    L0 = make_float2(tp.nw * go.x, tp.nw * go.y);
    R0 = make_float2(tp.ne * go.x, tp.ne * go.y);
    L1 = make_float2(tp.sw * go.x, tp.sw * go.y);
    R1 = make_float2(tp.se * go.x, tp.se * go.y);
  }
  const int px0 = __shfl_up(x0, 1, 64), py0 = __shfl_up(y0, 1, 64);
  const int nx0 = __shfl_down(x0, 1, 64), ny0 = __shfl_down(y0, 1, 64);
  const float pr0x = __shfl_up(R0.x, 1, 64), pr0y = __shfl_up(R0.y, 1, 64);
  const float pr1x = __shfl_up(R1.x, 1, 64), pr1y = __shfl_up(R1.y, 1, 64);
  const bool take_prev = lane > 0 && px0 + 1 == x0 && py0 == y0;
  const bool next_takes = lane < 63 && nx0 == x0 + 1 && ny0 == y0;
  if (take_prev) {
    L0.x += pr0x;
    L0.y += pr0y;
    L1.x += pr1x;
    L1.y += pr1y;
  }
  if (!active) return;
  if (inb(y0, x0, H, W)) {
    unsafeAtomicAdd(dpl + (size_t)y0 * W + x0, L0.x);
    unsafeAtomicAdd(dpl + P + (size_t)y0 * W + x0, L0.y);
  }
  if (inb(y0 + 1, x0, H, W)) {
    unsafeAtomicAdd(dpl + (size_t)(y0 + 1) * W + x0, L1.x);
    unsafeAtomicAdd(dpl + P + (size_t)(y0 + 1) * W + x0, L1.y);
  }
  if (!next_takes) {
    if (inb(y0, x0 + 1, H, W)) {
      unsafeAtomicAdd(dpl + (size_t)y0 * W + x0 + 1, R0.x);
      unsafeAtomicAdd(dpl + P + (size_t)y0 * W + x0 + 1, R0.y);
    }
    if (inb(y0 + 1, x0 + 1, H, W)) {
      unsafeAtomicAdd(dpl + (size_t)(y0 + 1) * W + x0 + 1, R1.x);
      unsafeAtomicAdd(dpl + P + (size_t)(y0 + 1) * W + x0 + 1, R1.y);
    }
  }
}

// Tiled variant (default in the solver).  rocprof on the lane-merged kernel: 4.15 M atomic requests,
// 83 % of wave cycles stalled at issue, i.e. it runs at the memory-side float-atomic rate
// (~20 G requests/s).  Rigid-ish motion maps a pixel tile onto a compact source window, and all
// motion groups of a tile land in nearly the same window, so: one workgroup = one 16x16 pixel tile
// x a chunk of motion groups; pass A finds the bounding box of all in-bounds taps, pass B
// accumulates dL/dimage in an LDS window (ds_add_f32) and the window is flushed once with
// row-contiguous atomics: ~20x fewer memory-side requests.  Windows larger than WIN_MAX pixels
// (wild displacement fields early in training are possible) fall back to direct atomics.
constexpr int WIN_MAX = 4096;

__global__ __launch_bounds__(256) void motion_warp_bwd_tiled_kernel(const float2* __restrict__ img,
                                                                    const float2* __restrict__ t_in,
                                                                    const float* __restrict__ xs,
                                                                    const float* __restrict__ ys,
                                                                    const float2* __restrict__ adj, int nM, int H,
                                                                    int W, int tiles_x, int m_per_chunk,
                                                                    float* __restrict__ dpl,
                                                                    float2* __restrict__ d_o) {
  __shared__ float win[2 * WIN_MAX];
  __shared__ int bb[4];
  const int tid = threadIdx.x;
  const int c = (blockIdx.x % tiles_x) * 16 + (tid & 15), r = (blockIdx.x / tiles_x) * 16 + (tid >> 4);
  const bool inside = c < W && r < H;
  const int m0 = blockIdx.y * m_per_chunk, m1 = min(nM, m0 + m_per_chunk);
  const int64_t P = (int64_t)H * W;
  if (tid == 0) {
    bb[0] = bb[1] = 1 << 30;
    bb[2] = bb[3] = -(1 << 30);
  }
  __syncthreads();
  const float gx0 = inside ? xs[c] : 0.f, gy0 = inside ? ys[r] : 0.f;
  // ---- pass A: bounding box of the in-bounds taps
  int mnx = 1 << 30, mny = 1 << 30, mxx = -(1 << 30), mxy = -(1 << 30);
  if (inside) {
    for (int m = m0; m < m1; ++m) {
      const float2 t = t_in[((int64_t)m * H + r) * W + c];
      const Taps tp = make_taps(t.x + gx0, t.y + gy0, H, W);
      // columns x0, x0+1 / rows y0, y0+1 clipped to the image
      const int ax = max(tp.x0, 0), bx = min(tp.x0 + 1, W - 1), ay = max(tp.y0, 0), by = min(tp.y0 + 1, H - 1);
      if (ax <= bx && ay <= by) {
        mnx = min(mnx, ax);
        mxx = max(mxx, bx);
        mny = min(mny, ay);
        mxy = max(mxy, by);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mnx = min(mnx, __shfl_xor(mnx, o, 64));
    mny = min(mny, __shfl_xor(mny, o, 64));
    mxx = max(mxx, __shfl_xor(mxx, o, 64));
    mxy = max(mxy, __shfl_xor(mxy, o, 64));
  }
  if ((tid & 63) == 0) {
    atomicMin(&bb[0], mnx);
    atomicMin(&bb[1], mny);
    atomicMax(&bb[2], mxx);
    atomicMax(&bb[3], mxy);
  }
  __syncthreads();
  const int wx0 = bb[0], wy0 = bb[1];
  const int ww = bb[2] - bb[0] + 1, wh = bb[3] - bb[1] + 1;
  const bool any = bb[2] >= bb[0] && bb[3] >= bb[1];
  const bool use_lds = any && (int64_t)ww * wh <= WIN_MAX;
  const int wn = use_lds ? ww * wh : 0;
  for (int i = tid; i < 2 * wn; i += 256) win[i] = 0.f;
  __syncthreads();
  // ---- pass B
  if (inside) {
    for (int m = m0; m < m1; ++m) {
      const int64_t i = ((int64_t)m * H + r) * W + c;
      const float2 t = t_in[i];
      const float s = ((r + c) & 1) ? -1.f : 1.f;
      const float2 a = adj[i];
      const float2 go = make_float2(a.x * s, a.y * s);
      const Taps tp = make_taps(t.x + gx0, t.y + gy0, H, W);
      const float2 dg = sample_bwd(img, nullptr, tp, go, H, W);
      d_o[i] = make_float2(dg.x * (1.f - t.x * t.x), dg.y * (1.f - t.y * t.y));
      const float wgt[4] = {tp.nw, tp.ne, tp.sw, tp.se};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int xx = tp.x0 + (k & 1), yy = tp.y0 + (k >> 1);
        if (inb(yy, xx, H, W)) {
          if (use_lds) {
            const int li = (yy - wy0) * ww + (xx - wx0);
            atomicAdd(&win[li], wgt[k] * go.x);
            atomicAdd(&win[wn + li], wgt[k] * go.y);
          } else {
            unsafeAtomicAdd(dpl + (size_t)yy * W + xx, wgt[k] * go.x);
            unsafeAtomicAdd(dpl + P + (size_t)yy * W + xx, wgt[k] * go.y);
          }
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < wn; i += 256) {
    const float re = win[i], im = win[wn + i];
    const size_t g = (size_t)(wy0 + i / ww) * W + wx0 + i % ww;
    if (re != 0.f) unsafeAtomicAdd(dpl + g, re);
    if (im != 0.f) unsafeAtomicAdd(dpl + P + g, im);
  }
}

// Round 4: the same kernel with the LDS window kept in 64-bit FIXED POINT.  tools/bench_lds_atomic.hip: ds_add_f32 costs
// ~3 clocks per active lane per CU (193 per wave instruction) where an integer LDS atomic costs 8 per wave instruction,
// and pass B issues 8.2 M float lane-atomics per launch at 320x320x10: 96 K clocks per CU = 0.046 of the kernel's
// 0.051 ms.  Every contribution wgt * go (one fp32 rounding, as before) is scaled by a per-workgroup power of two -
// 2^50 / (largest |adjoint| component of the workgroup's points, found in pass A) - rounded to an integer and added
// with ds_add_u64; the window sum is therefore the EXACT sum of the fp32 products to 2^-51 of the workgroup's largest
// term, independent of the order in which the lanes arrive (a float atomic rounds after every add), and is converted
// back to fp32 once when the window is flushed.  Range: at most 256 x (motion groups per workgroup) contributions per
// window cell, each below 2^tbits after scaling; the host picks tbits so that their sum stays below 2^62.
constexpr int WIN_MAX64 = 2048;

__global__ __launch_bounds__(256) void motion_warp_bwd_tiled_i64_kernel(const float2* __restrict__ img,
                                                                        const float2* __restrict__ t_in,
                                                                        const float* __restrict__ xs,
                                                                        const float* __restrict__ ys,
                                                                        const float2* __restrict__ adj, int nM, int H,
                                                                        int W, int tiles_x, int m_per_chunk, int tbits,
                                                                        float* __restrict__ dpl,
                                                                        float2* __restrict__ d_o) {
  __shared__ unsigned long long win[2 * WIN_MAX64];
  __shared__ int bb[4];
  __shared__ unsigned int amax_bits;
  const int tid = threadIdx.x;
  const int c = (blockIdx.x % tiles_x) * 16 + (tid & 15), r = (blockIdx.x / tiles_x) * 16 + (tid >> 4);
  const bool inside = c < W && r < H;
  const int m0 = blockIdx.y * m_per_chunk, m1 = min(nM, m0 + m_per_chunk);
  const int64_t P = (int64_t)H * W;
  if (tid == 0) {
    bb[0] = bb[1] = 1 << 30;
    bb[2] = bb[3] = -(1 << 30);
    amax_bits = 0u;
  }
  __syncthreads();
  const float gx0 = inside ? xs[c] : 0.f, gy0 = inside ? ys[r] : 0.f;
  // ---- pass A: bounding box of the in-bounds taps, largest adjoint component
  int mnx = 1 << 30, mny = 1 << 30, mxx = -(1 << 30), mxy = -(1 << 30);
  float am = 0.f;
  if (inside) {
    for (int m = m0; m < m1; ++m) {
      const int64_t i = ((int64_t)m * H + r) * W + c;
      const float2 t = t_in[i];
      const float2 a = adj[i];
      am = fmaxf(am, fmaxf(fabsf(a.x), fabsf(a.y)));
      const Taps tp = make_taps(t.x + gx0, t.y + gy0, H, W);
      const int ax = max(tp.x0, 0), bx = min(tp.x0 + 1, W - 1), ay = max(tp.y0, 0), by = min(tp.y0 + 1, H - 1);
      if (ax <= bx && ay <= by) {
        mnx = min(mnx, ax);
        mxx = max(mxx, bx);
        mny = min(mny, ay);
        mxy = max(mxy, by);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mnx = min(mnx, __shfl_xor(mnx, o, 64));
    mny = min(mny, __shfl_xor(mny, o, 64));
    mxx = max(mxx, __shfl_xor(mxx, o, 64));
    mxy = max(mxy, __shfl_xor(mxy, o, 64));
    am = fmaxf(am, __shfl_xor(am, o, 64));
  }
  if ((tid & 63) == 0) {
    atomicMin(&bb[0], mnx);
    atomicMin(&bb[1], mny);
    atomicMax(&bb[2], mxx);
    atomicMax(&bb[3], mxy);
    atomicMax(&amax_bits, __float_as_uint(am));   // non-negative floats order like their bit patterns (NaN: largest)
  }
  __syncthreads();
  const int wx0 = bb[0], wy0 = bb[1];
  const int ww = bb[2] - bb[0] + 1, wh = bb[3] - bb[1] + 1;
  const bool any = bb[2] >= bb[0] && bb[3] >= bb[1];
  // scale = 2^(tbits - e) with 2^(e-1) <= amax < 2^e (tbits = 53 - log2(contributions a cell can receive), host); amax = 0, inf or NaN: the direct float atomics below keep the
  // reference behaviour (zeros add nothing, inf / NaN propagate)
  const unsigned int ab = amax_bits;
  const int ex = (int)(ab >> 23) - 126;                     // frexp exponent of a normal float
  const bool fixed_ok = ab >= 0x00800000u && ab < 0x7F800000u;
  const bool use_lds = any && fixed_ok && (int64_t)ww * wh <= WIN_MAX64;
  const int k = min(max(tbits - ex, -100), 100);
  const float scale = __uint_as_float((unsigned int)(k + 127) << 23);
  const int wn = use_lds ? ww * wh : 0;
  for (int i = tid; i < 2 * wn; i += 256) win[i] = 0ull;
  __syncthreads();
  // ---- pass B
  if (inside) {
    for (int m = m0; m < m1; ++m) {
      const int64_t i = ((int64_t)m * H + r) * W + c;
      const float2 t = t_in[i];
      const float s = ((r + c) & 1) ? -1.f : 1.f;
      const float2 a = adj[i];
      const float2 go = make_float2(a.x * s, a.y * s);
      const Taps tp = make_taps(t.x + gx0, t.y + gy0, H, W);
      const float2 dg = sample_bwd(img, nullptr, tp, go, H, W);
      d_o[i] = make_float2(dg.x * (1.f - t.x * t.x), dg.y * (1.f - t.y * t.y));
      const float wgt[4] = {tp.nw, tp.ne, tp.sw, tp.se};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int xx = tp.x0 + (kk & 1), yy = tp.y0 + (kk >> 1);
        if (inb(yy, xx, H, W)) {
          const float vx = wgt[kk] * go.x, vy = wgt[kk] * go.y;
          if (use_lds) {
            const int li = (yy - wy0) * ww + (xx - wx0);
            atomicAdd(&win[li], (unsigned long long)__float2ll_rn(vx * scale));
            atomicAdd(&win[wn + li], (unsigned long long)__float2ll_rn(vy * scale));
          } else {
            unsafeAtomicAdd(dpl + (size_t)yy * W + xx, vx);
            unsafeAtomicAdd(dpl + P + (size_t)yy * W + xx, vy);
          }
        }
      }
    }
  }
  __syncthreads();
  const double inv = 1.0 / (double)scale;
  for (int i = tid; i < wn; i += 256) {
    const long long sre = (long long)win[i], sim = (long long)win[wn + i];
    const size_t g = (size_t)(wy0 + i / ww) * W + wx0 + i % ww;
    if (sre != 0) unsafeAtomicAdd(dpl + g, (float)((double)sre * inv));
    if (sim != 0) unsafeAtomicAdd(dpl + P + g, (float)((double)sim * inv));
  }
}

// ---- pruned path (round 4; SURVEY a9, reference src/models/immoco.py:97-111) ------------------------------------
// K = FFT(image) (1 - sum_m mask_m) + sum_m FFT(warp_m) mask_m with column-constant masks: of the nM 2-D transforms of
// the motion images only the k-space columns c with g(c) = m are ever used - W columns in all, whatever nM.  The row
// transform (along W) of motion image m is therefore evaluated as a DIRECT DFT for its own columns only, fused into
// the warp: Z[c][r] = sum_w warp_m[r][w] (-1)^(r+w) e^{-2 pi i w c / W}, written straight into the transposed
// k-space Z[W][H]; one batched column transform of W columns (rocFFT) finishes ALL images at once, and the select is
// implicit.  The adjoint runs backwards: column transform, then adj_m[r][w] = sum_{c in C_m} Zadj[c][r] e^{+2 pi i w c / W}
// evaluated inside the warp backward.  Per iteration this replaces 10 of 11 2-D transforms each way and the 8 MB
// warp-output / adjoint-input round trips by W^2 H complex multiply-adds (0.26 GFLOP at 320 x 320, whatever the masks).
// DFT_ROWS rows of one motion image per workgroup of the forward kernel.  The warp part is latency-bound (o -> tanh ->
// four image gathers per pixel), so fewer pixels per thread win: 320x320x10 isolated 0.0226 ms with 4 rows (5 pixels
// per thread), 0.0193 ms with 2 rows, 0.0202 ms with 1 (IMMOCO_DFT_ROWS in the diagnostics build).
template <int DFT_ROWS>
__global__ __launch_bounds__(256) void motion_warp_dft_kernel(const float2* __restrict__ img, const float2* __restrict__ o,
                                                              const float* __restrict__ xs, const float* __restrict__ ys,
                                                              int H, int W, const float2* __restrict__ tw_g,
                                                              const int32_t* __restrict__ cols,
                                                              const int32_t* __restrict__ off,
                                                              float2* __restrict__ t_out, float2* __restrict__ zt) {
  extern __shared__ float2 sm[];
  float2* v = sm;                      // [DFT_ROWS][W] warped, sign-modulated rows
  float2* tw = sm + DFT_ROWS * W;      // [W] e^{-2 pi i k / W}
  const int m = blockIdx.y, r0 = blockIdx.x * DFT_ROWS, tid = threadIdx.x;
  for (int idx = tid; idx < DFT_ROWS * W; idx += 256) {
    const int rr = idx / W, w = idx - rr * W, r = r0 + rr;
    float2 val = make_float2(0.f, 0.f);
    if (r < H) {
      const int64_t i = ((int64_t)m * H + r) * W + w;
      const float2 ov = o[i];
      const float2 t = make_float2(tanhf(ov.x), tanhf(ov.y));
      t_out[i] = t;
      const float2 sv = sample(img, make_taps(t.x + xs[w], t.y + ys[r], H, W), H, W);
      const float sg = ((r + w) & 1) ? -1.f : 1.f;
      val = make_float2(sv.x * sg, sv.y * sg);
    }
    v[idx] = val;
  }
  for (int k = tid; k < W; k += 256) tw[k] = tw_g[k];
  __syncthreads();
  const int c0 = off[m + 1], nc = off[m + 2] - c0;     // group m + 1 (0 is the unwarped image)
  const int lane = tid & 63, wave = tid >> 6;
  for (int oi = wave; oi < DFT_ROWS * nc; oi += 4) {   // one (row, column) output per wave pass
    const int rr = oi / nc, j = oi - rr * nc, r = r0 + rr;
    const int c = cols[c0 + j];
    const int step = (64 * c) % W;
    int k = (lane * c) % W;
    float ax = 0.f, ay = 0.f;
    for (int w = lane; w < W; w += 64) {
      const float2 x = v[rr * W + w], t = tw[k];
      ax = fmaf(x.x, t.x, fmaf(-x.y, t.y, ax));
      ay = fmaf(x.x, t.y, fmaf(x.y, t.x, ay));
      k += step;
      k -= k >= W ? W : 0;
    }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) {
      ax += __shfl_xor(ax, sft, 64);
      ay += __shfl_xor(ay, sft, 64);
    }
    if (lane == 0 && r < H) zt[(int64_t)c * H + r] = make_float2(ax, ay);
  }
}

int launch_motion_warp_dft(const float* image, const float* o, const float* xs, const float* ys, int nM, int H, int W,
                           const float* tw, const int32_t* cols, const int32_t* off, float* t_out, float* zt,
                           hipStream_t st) {
  if (nM == 0) return IMMOCO_OK;
  static const int rows_env = [] { const char* e = immoco_diag_env("IMMOCO_DFT_ROWS"); return e ? atoi(e) : 0; }();
  const int rows = (rows_env == 1 || rows_env == 2 || rows_env == 4) ? rows_env : 2;   // measured: 4 rows 0.0230, 2 rows 0.0193, 1 row 0.0202 ms
  dim3 grid((unsigned)cdiv(H, rows), (unsigned)nM);
  const size_t smem = (size_t)(rows + 1) * W * sizeof(float2);
  IMMOCO_REQUIRE(smem <= 64 * 1024, "motion_warp_dft: image width %d too large for the row tile", W);
#define IMMOCO_WDFT(R)                                                                                                  \
  motion_warp_dft_kernel<R><<<grid, 256, smem, st>>>((const float2*)image, (const float2*)o, xs, ys, H, W, (const float2*)tw, \
                                                     cols, off, (float2*)t_out, (float2*)zt)
  if (rows == 1) IMMOCO_WDFT(1);
  else if (rows == 2) IMMOCO_WDFT(2);
  else IMMOCO_WDFT(4);
#undef IMMOCO_WDFT
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// Backward: motion_warp_bwd_tiled_i64_kernel with the adjoint row DFT evaluated on the fly (16 x 16 pixel tile x a
// chunk of at most two motion groups).  zc[rr][jj] stages Zadj[c_j][r0 + rr] of 32 columns at a time.
constexpr int BWD_CC = 32;

template <int BWD_MPC>
__global__ __launch_bounds__(256) void motion_warp_bwd_dft_kernel(const float2* __restrict__ img,
                                                                  const float2* __restrict__ t_in,
                                                                  const float* __restrict__ xs,
                                                                  const float* __restrict__ ys,
                                                                  const float2* __restrict__ zt,
                                                                  const float2* __restrict__ tw_g,
                                                                  const int32_t* __restrict__ cols,
                                                                  const int32_t* __restrict__ off, int nM, int H, int W,
                                                                  int tiles_x, int tbits, float* __restrict__ dpl,
                                                                  float2* __restrict__ d_o) {
  __shared__ unsigned long long win[2 * WIN_MAX64];
  __shared__ float2 zc[16][BWD_CC];
  __shared__ int cl[BWD_CC];
  __shared__ int bb[4];
  __shared__ unsigned int amax_bits;
  extern __shared__ float2 tw[];   // [W] e^{-2 pi i k / W}; the adjoint uses the conjugate
  const int tid = threadIdx.x;
  const int tc = tid & 15, tr = tid >> 4;
  const int c = (blockIdx.x % tiles_x) * 16 + tc, r0 = (blockIdx.x / tiles_x) * 16, r = r0 + tr;
  const bool inside = c < W && r < H;
  const int m0 = blockIdx.y * BWD_MPC, m1 = min(nM, m0 + BWD_MPC);
  const int64_t P = (int64_t)H * W;
  if (tid == 0) {
    bb[0] = bb[1] = 1 << 30;
    bb[2] = bb[3] = -(1 << 30);
    amax_bits = 0u;
  }
  for (int k = tid; k < W; k += 256) tw[k] = tw_g[k];
  __syncthreads();
  // ---- adjoint row DFT: a[mi] = sum_{c' in C_m} Zadj[c'][r] e^{+2 pi i c c' / W}  (c = this thread's pixel column)
  float2 a[BWD_MPC];
#pragma unroll
  for (int mi = 0; mi < BWD_MPC; ++mi) {
    a[mi] = make_float2(0.f, 0.f);
    const int m = m0 + mi;
    if (m >= m1) continue;                                  // workgroup-uniform
    const int cb0 = off[m + 1], nc = off[m + 2] - cb0;
    for (int cb = 0; cb < nc; cb += BWD_CC) {
      const int ncc = min(BWD_CC, nc - cb);
      __syncthreads();                                      // the previous chunk's readers are done
      if (tid < ncc) cl[tid] = cols[cb0 + cb + tid];
      for (int e = tid; e < 16 * ncc; e += 256) {
        const int rr = e / ncc, jj = e - rr * ncc;
        const int cc = cols[cb0 + cb + jj];
        zc[rr][jj] = (r0 + rr) < H ? zt[(int64_t)cc * H + r0 + rr] : make_float2(0.f, 0.f);
      }
      __syncthreads();
      if (inside) {
        for (int jj = 0; jj < ncc; ++jj) {
          const float2 z = zc[tr][jj], t = tw[(c * cl[jj]) % W];
          // z * conj(t)
          a[mi].x = fmaf(z.x, t.x, fmaf(z.y, t.y, a[mi].x));
          a[mi].y = fmaf(z.y, t.x, fmaf(-z.x, t.y, a[mi].y));
        }
      }
    }
  }
  const float gx0 = inside ? xs[c] : 0.f, gy0 = inside ? ys[r] : 0.f;
  // ---- pass A: bounding box of the in-bounds taps, largest adjoint component
  int mnx = 1 << 30, mny = 1 << 30, mxx = -(1 << 30), mxy = -(1 << 30);
  float am = 0.f;
  if (inside) {
#pragma unroll
    for (int mi = 0; mi < BWD_MPC; ++mi) {
      const int m = m0 + mi;
      if (m >= m1) continue;
      const float2 t = t_in[((int64_t)m * H + r) * W + c];
      am = fmaxf(am, fmaxf(fabsf(a[mi].x), fabsf(a[mi].y)));
      const Taps tp = make_taps(t.x + gx0, t.y + gy0, H, W);
      const int ax = max(tp.x0, 0), bx = min(tp.x0 + 1, W - 1), ay = max(tp.y0, 0), by = min(tp.y0 + 1, H - 1);
      if (ax <= bx && ay <= by) {
        mnx = min(mnx, ax);
        mxx = max(mxx, bx);
        mny = min(mny, ay);
        mxy = max(mxy, by);
      }
    }
  }
#pragma unroll
  for (int sft = 32; sft > 0; sft >>= 1) {
    mnx = min(mnx, __shfl_xor(mnx, sft, 64));
    mny = min(mny, __shfl_xor(mny, sft, 64));
    mxx = max(mxx, __shfl_xor(mxx, sft, 64));
    mxy = max(mxy, __shfl_xor(mxy, sft, 64));
    am = fmaxf(am, __shfl_xor(am, sft, 64));
  }
  if ((tid & 63) == 0) {
    atomicMin(&bb[0], mnx);
    atomicMin(&bb[1], mny);
    atomicMax(&bb[2], mxx);
    atomicMax(&bb[3], mxy);
    atomicMax(&amax_bits, __float_as_uint(am));
  }
  __syncthreads();
  const int wx0 = bb[0], wy0 = bb[1];
  const int ww = bb[2] - bb[0] + 1, wh = bb[3] - bb[1] + 1;
  const bool any = bb[2] >= bb[0] && bb[3] >= bb[1];
  const unsigned int ab = amax_bits;
  const int ex = (int)(ab >> 23) - 126;
  const bool fixed_ok = ab >= 0x00800000u && ab < 0x7F800000u;
  const bool use_lds = any && fixed_ok && (int64_t)ww * wh <= WIN_MAX64;
  const int k2 = min(max(tbits - ex, -100), 100);
  const float scale = __uint_as_float((unsigned int)(k2 + 127) << 23);
  const int wn = use_lds ? ww * wh : 0;
  for (int i = tid; i < 2 * wn; i += 256) win[i] = 0ull;
  __syncthreads();
  // ---- pass B
  if (inside) {
#pragma unroll
    for (int mi = 0; mi < BWD_MPC; ++mi) {
      const int m = m0 + mi;
      if (m >= m1) continue;
      const int64_t i = ((int64_t)m * H + r) * W + c;
      const float2 t = t_in[i];
      const float s = ((r + c) & 1) ? -1.f : 1.f;
      const float2 go = make_float2(a[mi].x * s, a[mi].y * s);
      const Taps tp = make_taps(t.x + gx0, t.y + gy0, H, W);
      const float2 dg = sample_bwd(img, nullptr, tp, go, H, W);
      d_o[i] = make_float2(dg.x * (1.f - t.x * t.x), dg.y * (1.f - t.y * t.y));
      const float wgt[4] = {tp.nw, tp.ne, tp.sw, tp.se};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int xx = tp.x0 + (kk & 1), yy = tp.y0 + (kk >> 1);
        if (inb(yy, xx, H, W)) {
          const float vx = wgt[kk] * go.x, vy = wgt[kk] * go.y;
          if (use_lds) {
            const int li = (yy - wy0) * ww + (xx - wx0);
            atomicAdd(&win[li], (unsigned long long)__float2ll_rn(vx * scale));
            atomicAdd(&win[wn + li], (unsigned long long)__float2ll_rn(vy * scale));
          } else {
            unsafeAtomicAdd(dpl + (size_t)yy * W + xx, vx);
            unsafeAtomicAdd(dpl + P + (size_t)yy * W + xx, vy);
          }
        }
      }
    }
  }
  __syncthreads();
  const double inv = 1.0 / (double)scale;
  for (int i = tid; i < wn; i += 256) {
    const long long sre = (long long)win[i], sim = (long long)win[wn + i];
    const size_t g = (size_t)(wy0 + i / ww) * W + wx0 + i % ww;
    if (sre != 0) unsafeAtomicAdd(dpl + g, (float)((double)sre * inv));
    if (sim != 0) unsafeAtomicAdd(dpl + P + g, (float)((double)sim * inv));
  }
}

int launch_motion_warp_bwd_dft(const float* image, const float* t, const float* xs, const float* ys, const float* zt_adj,
                               const float* tw, const int32_t* cols, const int32_t* off, int nM, int H, int W,
                               float* dimage_planar, float* d_o, hipStream_t st) {
  if (nM == 0) return IMMOCO_OK;
  const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
  // motion groups per workgroup: 2 (0.0325 ms; 1 group: 0.0383 ms; IMMOCO_BWD_MPC in the diagnostics build)
  static const int mpc_env = [] { const char* e = immoco_diag_env("IMMOCO_BWD_MPC"); return e ? atoi(e) : 0; }();
  const int mpc = mpc_env == 1 ? 1 : 2;
  dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)cdiv(nM, mpc));
  const int tbits = 62 - 8 - 1 - 1;   // 256 pixels x <= 2 groups of < 2^(tbits + 1) each
  if (mpc == 1)
    motion_warp_bwd_dft_kernel<1><<<grid, 256, (size_t)W * sizeof(float2), st>>>(
        (const float2*)image, (const float2*)t, xs, ys, (const float2*)zt_adj, (const float2*)tw, cols, off, nM, H, W,
        tiles_x, tbits, dimage_planar, (float2*)d_o);
  else
    motion_warp_bwd_dft_kernel<2><<<grid, 256, (size_t)W * sizeof(float2), st>>>(
        (const float2*)image, (const float2*)t, xs, ys, (const float2*)zt_adj, (const float2*)tw, cols, off, nM, H, W,
        tiles_x, tbits, dimage_planar, (float2*)d_o);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// ---- motion simulator pieces (reference src/utils/motion_utils.py:165-195) ---------------------
// affine_grid(theta, align_corners=True) followed by grid_sample(bilinear, padding_mode="border",
// align_corners=False) of one complex image for n rigid movements.  ATen semantics: the
// unnormalised coordinate is clipped to [0, size-1] before the bilinear taps are taken.
__global__ __launch_bounds__(256) void affine_warp_border_kernel(const float2* __restrict__ img,
                                                                 const float* __restrict__ theta /*[n][2][3]*/,
                                                                 const float* __restrict__ xs,
                                                                 const float* __restrict__ ys, int64_t n_tot, int H,
                                                                 int W, float2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_tot) return;
  const int c = (int)(i % W), r = (int)((i / W) % H);
  const int m = (int)(i / ((int64_t)W * H));
  const float* t = theta + 6 * m;
  const float x = xs[c], y = ys[r];
  const float gx = x * t[0] + y * t[1] + t[2], gy = x * t[3] + y * t[4] + t[5];
  float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
  ix = fminf(fmaxf(ix, 0.f), (float)(W - 1));
  iy = fminf(fmaxf(iy, 0.f), (float)(H - 1));
  const float fx = floorf(ix), fy = floorf(iy);
  const int x0 = (int)fx, y0 = (int)fy;
  const float tx1 = ix - fx, tx0 = (fx + 1.f) - ix, ty1 = iy - fy, ty0 = (fy + 1.f) - iy;
  const float2 a = ld_or_zero(img, y0, x0, H, W), b = ld_or_zero(img, y0, x0 + 1, H, W);
  const float2 cc = ld_or_zero(img, y0 + 1, x0, H, W), d = ld_or_zero(img, y0 + 1, x0 + 1, H, W);
  const float nw = tx0 * ty0, ne = tx1 * ty0, sw = tx0 * ty1, se = tx1 * ty1;
  out[i] = make_float2(a.x * nw + b.x * ne + cc.x * sw + d.x * se, a.y * nw + b.y * ne + cc.y * sw + d.y * se);
}

// ksp[..., w0:w1] = ksp_m[m][..., w0:w1] applied for m = 0..n-1 in order (a later movement
// overwrites an earlier one where bands overlap), mask[:, w0:w1] = 1   (motion_utils.py:191-196)
__global__ __launch_bounds__(256) void band_replace_kernel(const float2* __restrict__ k0,
                                                           const float2* __restrict__ kall,
                                                           const int32_t* __restrict__ w0,
                                                           const int32_t* __restrict__ w1, int n, int H, int W,
                                                           float2* __restrict__ kout, int64_t* __restrict__ mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t P = (int64_t)H * W;
  if (i >= P) return;
  const int c = (int)(i % W);
  int sel = -1;
  for (int m = 0; m < n; ++m)
    if (c >= w0[m] && c < w1[m]) sel = m;
  kout[i] = sel >= 0 ? kall[(int64_t)sel * P + i] : k0[i];
  if (mask) mask[i] = sel >= 0 ? 1 : 0;
}

// ---- Autofocusing baseline (reference src/models/autofocusing.py:71-85): per-group affine grid
// (align_corners=True) + F.grid_sample(mode="bicubic", padding zeros, align_corners=False) of a
// per-group complex image; backward with respect to the affine matrices only (the images are
// constants of the method).  ATen cubic convolution, A = -0.75.
__device__ __forceinline__ void cubic_coeffs(float t, float (&c)[4]) {
  const float A = -0.75f;
  float x = t + 1.f;
  c[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  c[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  c[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
__device__ __forceinline__ void cubic_coeffs_grad(float t, float (&c)[4]) {
  const float A = -0.75f;
  float x = -1.f - t;
  c[0] = (-3.f * A * x - 10.f * A) * x - 8.f * A;
  x = -t;
  c[1] = (-3.f * (A + 2.f) * x - 2.f * (A + 3.f)) * x;
  x = 1.f - t;
  c[2] = (3.f * (A + 2.f) * x - 2.f * (A + 3.f)) * x;
  x = 2.f - t;
  c[3] = (3.f * A * x - 10.f * A) * x + 8.f * A;
}

// BWD = false: out[i] = sample ; BWD = true: dtheta[m][6] += d(sum Re(conj(dout) * out))/dtheta
template <bool BWD>
__global__ __launch_bounds__(256) void affine_bicubic_kernel(const float2* __restrict__ imgs /*[n][H][W]*/,
                                                             const float* __restrict__ theta /*[n][2][3]*/,
                                                             const float* __restrict__ xs,
                                                             const float* __restrict__ ys, int n, int H, int W,
                                                             const float2* __restrict__ dout,
                                                             float2* __restrict__ out, float* __restrict__ dtheta) {
  const int64_t P = (int64_t)H * W;
  const int m = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float g6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < P) {
    const int c = (int)(i % W), r = (int)(i / W);
    const float* t = theta + 6 * m;
    const float x = xs[c], y = ys[r];
    const float gx = x * t[0] + y * t[1] + t[2], gy = x * t[3] + y * t[4] + t[5];
    const float ix = ((gx + 1.f) * (float)W - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fminf(fmaxf(fx, -4.f), (float)W + 3.f), y0 = (int)fminf(fmaxf(fy, -4.f), (float)H + 3.f);
    float cx[4], cy[4];
    cubic_coeffs(ix - fx, cx);
    cubic_coeffs(iy - fy, cy);
    const float2* __restrict__ img = imgs + (int64_t)m * P;
    if (!BWD) {
      float2 acc = make_float2(0.f, 0.f);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float2 v = ld_or_zero(img, y0 - 1 + b, x0 - 1 + a, H, W);
          const float w = cx[a] * cy[b];
          acc.x += v.x * w;
          acc.y += v.y * w;
        }
      out[(int64_t)m * P + i] = acc;
    } else {
      float dx[4], dy[4];
      cubic_coeffs_grad(ix - fx, dx);
      cubic_coeffs_grad(iy - fy, dy);
      const float2 go = dout[(int64_t)m * P + i];
      float gix = 0.f, giy = 0.f;
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float2 v = ld_or_zero(img, y0 - 1 + b, x0 - 1 + a, H, W);
          const float vg = v.x * go.x + v.y * go.y;   // summed over the (re, im) channels
          gix -= vg * dx[a] * cy[b];
          giy -= vg * dy[b] * cx[a];
        }
      const float ggx = gix * 0.5f * (float)W, ggy = giy * 0.5f * (float)H;
      g6[0] = ggx * x;
      g6[1] = ggx * y;
      g6[2] = ggx;
      g6[3] = ggy * x;
      g6[4] = ggy * y;
      g6[5] = ggy;
    }
  }
  if (BWD) {
    __shared__ float red[4][6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float v = wave_sum(g6[k]);
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 6)
      unsafeAtomicAdd(dtheta + 6 * m + threadIdx.x,
                      red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
  }
}

int launch_warp_fwd(const float* image, const float* grids, int nM, int H, int W, float* out, hipStream_t st) {
  const int64_t n = (int64_t)nM * H * W;
  if (n == 0) return IMMOCO_OK;
  warp_fwd_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)image, (const float2*)grids, n, H, W,
                                                          (float2*)out);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_warp_bwd(const float* image, const float* grids, const float* dout, int nM, int H, int W,
                    float* dimage, float* dgrids, hipStream_t st) {
  const int64_t n = (int64_t)nM * H * W;
  if (n == 0) return IMMOCO_OK;
  warp_bwd_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)image, (const float2*)grids,
                                                          (const float2*)dout, n, H, W, dimage,
                                                          (float2*)dgrids);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_motion_warp_fwd(const float* image, const float* o, const float* xs, const float* ys, int nM,
                           int H, int W, float* t_out, float* fft_slots, hipStream_t st) {
  const int64_t n = (int64_t)nM * H * W;
  if (n == 0) return IMMOCO_OK;
  motion_warp_fwd_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)image, (const float2*)o, xs,
                                                                 ys, n, H, W, (float2*)t_out,
                                                                 (float2*)fft_slots);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_motion_warp_bwd(const float* image, const float* t, const float* xs, const float* ys,
                           const float* adj_slots, int nM, int H, int W, float* dimage, float* d_o,
                           hipStream_t st) {
  const int64_t n = (int64_t)nM * H * W;
  if (n == 0) return IMMOCO_OK;
  static const bool lane_merge_only = immoco_diag_env("IMMOCO_WARP_BWD") && strcmp(immoco_diag_env("IMMOCO_WARP_BWD"), "flat") == 0;
  if (!lane_merge_only) {
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    // enough workgroups to fill the chip: split the motion groups into chunks
    int chunks = 1;
    // (the kernel is latency-bound: 320x320x10 with 10 / 5 / 3 / 2 / 1 motion groups per workgroup takes 0.063 / 0.061 /
    // 0.056 / 0.054 / 0.061 ms - more, shorter workgroups until the per-workgroup window flush dominates.  A/B switch
    // (environment, read once): IMMOCO_WARP_BLOCKS = minimum number of workgroups.)
    static const int min_blocks = [] { const char* e = immoco_diag_env("IMMOCO_WARP_BLOCKS"); return e ? atoi(e) : 2000; }();
    while (chunks < nM && (int64_t)tiles_x * tiles_y * chunks < min_blocks) ++chunks;
    const int mpc = (nM + chunks - 1) / chunks;
    chunks = (nM + mpc - 1) / mpc;
    dim3 grid(tiles_x * tiles_y, chunks);
    // fixed-point LDS window (default; needs <= 2^9 contributions per window cell: 256 pixels x mpc groups);
    // IMMOCO_WARP_BWD=f32win (diagnostics build): the float-atomic window of rounds 2-3
    static const bool f32win = immoco_diag_env("IMMOCO_WARP_BWD") && strcmp(immoco_diag_env("IMMOCO_WARP_BWD"), "f32win") == 0;
    int lg = 0;
    while ((1 << lg) < mpc) ++lg;
    const int tbits = 62 - 8 - lg - 1;   // 256 * mpc contributions of < 2^(tbits + 1) each (|wgt * go| <= amax < 2^e)
    if (!f32win)
      motion_warp_bwd_tiled_i64_kernel<<<grid, 256, 0, st>>>((const float2*)image, (const float2*)t, xs, ys,
                                                             (const float2*)adj_slots, nM, H, W, tiles_x, mpc, tbits,
                                                             dimage, (float2*)d_o);
    else
      motion_warp_bwd_tiled_kernel<<<grid, 256, 0, st>>>((const float2*)image, (const float2*)t, xs, ys,
                                                         (const float2*)adj_slots, nM, H, W, tiles_x, mpc, dimage,
                                                         (float2*)d_o);
    IMMOCO_LAUNCH_CHECK();
    return IMMOCO_OK;
  }
  motion_warp_bwd_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)image, (const float2*)t, xs,
                                                                 ys, (const float2*)adj_slots, n, H, W,
                                                                 dimage, (float2*)d_o);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

}  // namespace immoco

using namespace immoco;

extern "C" int immoco_warp_fwd(const float* image, const float* grids, int32_t nM, int32_t H, int32_t W,
                               float* out, void* stream) {
  IMMOCO_REQUIRE(nM >= 0 && H > 0 && W > 0, "warp_fwd: bad shape nM=%d H=%d W=%d", nM, H, W);
  IMMOCO_REQUIRE(nM == 0 || (image && grids && out), "warp_fwd: NULL buffer");
  return launch_warp_fwd(image, grids, nM, H, W, out, as_stream(stream));
}

extern "C" int immoco_warp_bwd(const float* image, const float* grids, const float* dout, int32_t nM,
                               int32_t H, int32_t W, float* dimage, float* dgrids, void* stream) {
  IMMOCO_REQUIRE(nM >= 0 && H > 0 && W > 0, "warp_bwd: bad shape nM=%d H=%d W=%d", nM, H, W);
  IMMOCO_REQUIRE(nM == 0 || (image && grids && dout && dgrids), "warp_bwd: NULL buffer");
  return launch_warp_bwd(image, grids, dout, nM, H, W, dimage, dgrids, as_stream(stream));
}

extern "C" int immoco_affine_warp_border(const float* image, const float* theta, const float* xs, const float* ys,
                                         int32_t n, int32_t H, int32_t W, float* out, void* stream) {
  IMMOCO_REQUIRE(n >= 0 && H > 0 && W > 0, "affine_warp_border: bad shape n=%d H=%d W=%d", n, H, W);
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(image && theta && xs && ys && out, "affine_warp_border: NULL buffer");
  const int64_t tot = (int64_t)n * H * W;
  affine_warp_border_kernel<<<(unsigned)cdiv(tot, 256), 256, 0, as_stream(stream)>>>(
      (const float2*)image, theta, xs, ys, tot, H, W, (float2*)out);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

extern "C" int immoco_band_replace(const float* k0, const float* kall, const int32_t* w0, const int32_t* w1,
                                   int32_t n, int32_t H, int32_t W, float* kout, int64_t* mask, void* stream) {
  IMMOCO_REQUIRE(n >= 0 && H > 0 && W > 0 && k0 && kout, "band_replace: bad argument");
  IMMOCO_REQUIRE(n == 0 || (kall && w0 && w1), "band_replace: NULL buffer");
  band_replace_kernel<<<(unsigned)cdiv((int64_t)H * W, 256), 256, 0, as_stream(stream)>>>(
      (const float2*)k0, (const float2*)kall, w0, w1, n, H, W, (float2*)kout, mask);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

extern "C" int immoco_affine_bicubic_fwd(const float* images, const float* theta, const float* xs, const float* ys,
                                         int32_t n, int32_t H, int32_t W, float* out, void* stream) {
  IMMOCO_REQUIRE(n >= 0 && H > 0 && W > 0, "affine_bicubic_fwd: bad shape n=%d H=%d W=%d", n, H, W);
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(images && theta && xs && ys && out, "affine_bicubic_fwd: NULL buffer");
  dim3 grid((unsigned)cdiv((int64_t)H * W, 256), n);
  affine_bicubic_kernel<false><<<grid, 256, 0, as_stream(stream)>>>((const float2*)images, theta, xs, ys, n, H, W,
                                                                   nullptr, (float2*)out, nullptr);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

extern "C" int immoco_affine_bicubic_bwd(const float* images, const float* theta, const float* xs, const float* ys,
                                         const float* dout, int32_t n, int32_t H, int32_t W, float* dtheta,
                                         void* stream) {
  IMMOCO_REQUIRE(n >= 0 && H > 0 && W > 0, "affine_bicubic_bwd: bad shape n=%d H=%d W=%d", n, H, W);
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(images && theta && xs && ys && dout && dtheta, "affine_bicubic_bwd: NULL buffer");
  dim3 grid((unsigned)cdiv((int64_t)H * W, 256), n);
  affine_bicubic_kernel<true><<<grid, 256, 0, as_stream(stream)>>>((const float2*)images, theta, xs, ys, n, H, W,
                                                                  (const float2*)dout, nullptr, dtheta);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}
