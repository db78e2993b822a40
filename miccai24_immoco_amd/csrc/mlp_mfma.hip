// MFMA implementation of the bias-free MLP 32 -> HID -> 2 (forward and backward).
//
// Why: the fp32 VALU kernels (mlp.hip) spend 1.15 + 0.53 ms per iteration in the backward pass
// (rocprof, MI355X) - the weight-gradient outer products need a reduction over points that a
// lane-per-point mapping can only do through LDS broadcasts.  Every product of the MLP is a small
// dense GEMM with K = 32 or 64, so it belongs on the matrix cores.  gfx950's f32-input MFMA runs
// at the VALU rate (1/16 of bf16), so fp32 operands are split into THREE bf16 terms
// (x = x1 + x2 + x3, 8+8+8 mantissa bits) and each product is evaluated with the six MFMAs whose
// terms are >= 2^-24 relative: x1y1 + x1y2 + x2y1 + x1y3 + x2y2 + x3y1, accumulated in fp32.
// Result: fp32-equivalent accuracy at 16/6 of the fp32 matrix rate, and - more importantly - the
// reductions over points and over hidden units happen inside the MFMA K dimension.
//
// One wave owns a tile of 32 points.  v_mfma_f32_32x32x16_bf16 fragment maps (cdna guide §3):
//   A: lane l (r = l&31, h = l>>5) holds A[row r][k = 8h + i], i = 0..7
//   B: lane l holds B[k = 8h + i][col r]
//   D: reg g of lane l holds D[row (g&3) + 8(g>>2) + 4h][col r]
// A result tile X (rows in registers, column on the lane) feeds the next MFMA as B operand of
// k-step s with its registers 8s..8s+7 (k order 16s + 8(i>>2) + 4h + (i&3)); the other operand is
// loaded in that same k order.  Both layouts of the hidden tile are computed (MFMAs are cheap):
//   layout 1  pre [hidden][point] = W1 . enc^T      -> d enc^T = W1^T . dpre        (sum over hidden)
//   layout 2  pre'[point][hidden] = enc . W1^T      -> dW1^T   = enc^T . dpre'      (sum over points)
//                                                     dW2^T   = dout^T . h'        (sum over points)
// so no fp32 tile is ever transposed through LDS.  Pre-split weight fragments live in LDS.
#include "kernels.hpp"

namespace immoco {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

struct Frag3 {  // three bf16 terms of one 8-element fp32 fragment
  u32x4 h, m, l;
};

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
  bf16x2 v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float lo_f(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi_f(uint32_t p) { return __builtin_bit_cast(float, p & 0xFFFF0000u); }

// split 8 floats into 3 bf16 terms each
__device__ __forceinline__ Frag3 split3(const float (&x)[8]) {
  Frag3 f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = x[2 * q], b = x[2 * q + 1];
    const uint32_t ph = pack_bf16(a, b);
    const float ra = a - lo_f(ph), rb = b - hi_f(ph);
    const uint32_t pm = pack_bf16(ra, rb);
    const float qa = ra - lo_f(pm), qb = rb - hi_f(pm);
    const uint32_t pl = pack_bf16(qa, qb);
    f.h[q] = ph;
    f.m[q] = pm;
    f.l[q] = pl;
  }
  return f;
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0,
                                                 0, 0);
}

// acc += A . B with fp32-equivalent accuracy (six bf16 MFMAs, small terms first)
__device__ __forceinline__ void mfma6(f32x16& acc, const Frag3& a, const Frag3& b) {
  acc = mfma_bf16(a.l, b.h, acc);
  acc = mfma_bf16(a.m, b.m, acc);
  acc = mfma_bf16(a.h, b.l, acc);
  acc = mfma_bf16(a.m, b.h, acc);
  acc = mfma_bf16(a.h, b.m, acc);
  acc = mfma_bf16(a.h, b.h, acc);
}

// row index of D register g for lane half h
__device__ __forceinline__ int drow(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// tanh accurate to ~1e-7 relative without libm's cost: odd series below 0.25, exp form above
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = fabsf(x);
  const float x2 = x * x;
  const float ser = ax * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f + x2 * 0.021869488f))));
  const float e = __expf(2.f * ax);
  const float big = 1.f - __fdividef(2.f, e + 1.f);
  return copysignf(ax < 0.25f ? ser : big, x);
}

template <int ACT>
__device__ __forceinline__ float act_f(float pre) {
  return ACT == IMMOCO_ACT_RELU ? fmaxf(pre, 0.f) : tanh_fast(pre);
}
template <int ACT>
__device__ __forceinline__ float act_d(float pre, float hv) {
  return ACT == IMMOCO_ACT_RELU ? (pre > 0.f ? 1.f : 0.f) : 1.f - hv * hv;
}

// LDS image of pre-split weight fragments: frag index f, term t (0 h,1 m,2 l): [f][t][lane] u32x4
__device__ __forceinline__ void lds_store_frag(u32x4* base, int f, int lane, const Frag3& v) {
  base[(f * 3 + 0) * 64 + lane] = v.h;
  base[(f * 3 + 1) * 64 + lane] = v.m;
  base[(f * 3 + 2) * 64 + lane] = v.l;
}
__device__ __forceinline__ Frag3 lds_load_frag(const u32x4* base, int f, int lane) {
  Frag3 v;
  v.h = base[(f * 3 + 0) * 64 + lane];
  v.m = base[(f * 3 + 1) * 64 + lane];
  v.l = base[(f * 3 + 2) * 64 + lane];
  return v;
}

// fragment of the point tile: element i = in[point p][feature 16*ks + 8h + i] (0 beyond n).
// The loads are UNCONDITIONAL on a clamped address and masked afterwards: a load under a divergent
// `if` gets its own basic block and its own s_waitcnt, which serialises the whole tile prologue.
__device__ __forceinline__ void load_enc_frag(const float* in, int64_t ps, int64_t ls, int64_t p, int64_t n, int ks,
                                              int h, float (&x)[8]) {
  const int64_t pc = p < n ? p : n - 1;
  const float m = p < n ? 1.f : 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int level = 8 * ks + 4 * h + q;
    const float2 v = *reinterpret_cast<const float2*>(in + pc * ps + (int64_t)level * ls);
    x[2 * q] = v.x * m;
    x[2 * q + 1] = v.y * m;
  }
}

// ---------------------------------------------------------------------------------------------
// forward: out[p][0..1] = W2 . act(W1 . enc[p])
template <int HID, int ACT>
__global__ __launch_bounds__(256) void mlp_fwd_mfma_kernel(const float* __restrict__ in, int64_t ps, int64_t ls,
                                                           int64_t n, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, float* __restrict__ out,
                                                           int64_t n_tiles) {
  constexpr int NJT = HID / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* aw = reinterpret_cast<u32x4*>(smem);                          // [NJT*2][3][64]
  float* w2s = reinterpret_cast<float*>(aw + NJT * 2 * 3 * 64);         // [2][HID]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  // build the weight fragments once per block: wave w handles fragments w, w+4, ...
  for (int f = wave; f < NJT * 2; f += 4) {
    const int jt = f >> 1, ks = f & 1;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = w1[(jt * 32 + r) * 32 + ks * 16 + 8 * h + i];
    lds_store_frag(aw, f, lane, split3(x));
  }
  for (int i = threadIdx.x; i < 2 * HID; i += 256) w2s[i] = w2[i];
  __syncthreads();

  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  // software pipeline: the next tile's raw operands are in flight while this one is computed
  float nx[2][8];
  if (wave_id < n_tiles) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) load_enc_frag(in, ps, ls, wave_id * 32 + r, n, ks, h, nx[ks]);
  }
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const int64_t p = t * 32 + r;
    const bool valid = p < n;
    Frag3 eb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) eb[ks] = split3(nx[ks]);
    if (t + n_waves < n_tiles) {
      const int64_t pn = (t + n_waves) * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) load_enc_frag(in, ps, ls, pn, n, ks, h, nx[ks]);
    }
    float o0 = 0.f, o1 = 0.f;
#pragma unroll 2
    for (int jt = 0; jt < NJT; ++jt) {
      f32x16 pre = {0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) mfma6(pre, lds_load_frag(aw, jt * 2 + ks, lane), eb[ks]);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 wa = *reinterpret_cast<const float4*>(w2s + jt * 32 + 8 * a + 4 * h);
        const float4 wb = *reinterpret_cast<const float4*>(w2s + HID + jt * 32 + 8 * a + 4 * h);
        const float h0 = act_f<ACT>(pre[4 * a]), h1 = act_f<ACT>(pre[4 * a + 1]);
        const float h2 = act_f<ACT>(pre[4 * a + 2]), h3 = act_f<ACT>(pre[4 * a + 3]);
        o0 = fmaf(wa.x, h0, fmaf(wa.y, h1, fmaf(wa.z, h2, fmaf(wa.w, h3, o0))));
        o1 = fmaf(wb.x, h0, fmaf(wb.y, h1, fmaf(wb.z, h2, fmaf(wb.w, h3, o1))));
      }
    }
    o0 += __shfl_xor(o0, 32, 64);
    o1 += __shfl_xor(o1, 32, 64);
    if (valid && h == 0) *reinterpret_cast<float2*>(out + p * 2) = make_float2(o0, o1);
  }
}

// ---------------------------------------------------------------------------------------------
// backward
template <int HID, int ACT>
__global__ __launch_bounds__(256, HID == 64 ? 2 : 1) void mlp_bwd_mfma_kernel(const float* in /* may alias din */, int64_t ps, int64_t ls,
                                                           int64_t n, const float* __restrict__ w1,
                                                           const float* __restrict__ w2,
                                                           const float* __restrict__ dout, float* din,
                                                           float* __restrict__ dw1, float* __restrict__ dw2,
                                                           int64_t n_tiles, int64_t dout_plane) {
  constexpr int NJT = HID / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u32x4* aw = reinterpret_cast<u32x4*>(smem);                  // W1 fragments        [NJT*2][3][64]
  u32x4* awt = aw + NJT * 2 * 3 * 64;                          // W1^T fragments      [NJT*2][3][64]
  float* w2s = reinterpret_cast<float*>(awt + NJT * 2 * 3 * 64);  // [2][HID]
  float* dos_all = w2s + 2 * HID;                              // per wave [32][2] dout staging
  float* tr_all = dos_all + 4 * 64;                            // per wave [32][36] transpose tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  float* dos = dos_all + wave * 64;
  float* tr = tr_all + wave * 32 * 36;
  for (int f = wave; f < NJT * 2; f += 4) {
    const int jt = f >> 1, s = f & 1;
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = w1[(jt * 32 + r) * 32 + s * 16 + 8 * h + i];
    lds_store_frag(aw, f, lane, split3(x));
    // W1^T fragment in accumulator-k order: element i = W1[jt*32 + 16s + 8(i>>2) + 4h + (i&3)][k = r]
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = w1[(jt * 32 + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)) * 32 + r];
    lds_store_frag(awt, f, lane, split3(x));
  }
  for (int i = threadIdx.x; i < 2 * HID; i += 256) w2s[i] = w2[i];
  __syncthreads();

  f32x16 dw1t[NJT];  // dW1^T tiles: rows k (features), col = hidden jt*32 + r
  float dw2a[NJT][2];
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) {
    dw1t[jt] = (f32x16){0.f};
    dw2a[jt][0] = dw2a[jt][1] = 0.f;
  }

  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  // software pipeline: raw operands of the NEXT tile are loaded while the current one is computed
  float nx_e[2][8], nx_t[2][8];
  float2 nx_d = make_float2(0.f, 0.f);
  auto load_raw = [&](int64_t tt) {
    const int64_t q0 = tt * 32, q = q0 + r;
    const int64_t qc = q < n ? q : n - 1;
    const float mq = q < n ? 1.f : 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) load_enc_frag(in, ps, ls, q, n, ks, h, nx_e[ks]);
    if (dout_plane) {  // wave-uniform
      nx_d = make_float2(dout[qc] * mq, dout[dout_plane + qc] * mq);
    } else {
      const float2 dv = *reinterpret_cast<const float2*>(dout + qc * 2);
      nx_d = make_float2(dv.x * mq, dv.y * mq);
    }
    // enc^T fragments (rows = feature r, K = points in accumulator order)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t qq = q0 + 16 * s2 + 8 * (i >> 2) + 4 * h + (i & 3);
        const int64_t qqc = qq < n ? qq : n - 1;
        const float v = in[qqc * ps + (int64_t)(r >> 1) * ls + (r & 1)];
        nx_t[s2][i] = qq < n ? v : 0.f;
      }
  };
  if (wave_id < n_tiles) load_raw(wave_id);
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const int64_t p0 = t * 32, p = p0 + r;
    const bool valid = p < n;
    // ---- operands of this tile
    Frag3 eb[2], et[2], dt[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      eb[ks] = split3(nx_e[ks]);
      et[ks] = split3(nx_t[ks]);
    }
    const float2 d = nx_d;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // earlier readers of dos (previous tile) are done
    if (h == 0) *reinterpret_cast<float2*>(dos + 2 * r) = d;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // dos visible to the whole wave
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float x[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = 16 * s + 8 * (i >> 2) + 4 * h + (i & 3);
        x[i] = r < 2 ? dos[2 * q + r] : 0.f;
      }
      dt[s] = split3(x);
    }
    if (t + n_waves < n_tiles) load_raw(t + n_waves);
    f32x16 denc = {0.f};
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
      const Frag3 a0 = lds_load_frag(aw, jt * 2, lane), a1 = lds_load_frag(aw, jt * 2 + 1, lane);
      // ---- layout 1: rows = hidden, col = point
      f32x16 pre = {0.f};
      mfma6(pre, a0, eb[0]);
      mfma6(pre, a1, eb[1]);
      float dp[16], hv1[16];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 wa = *reinterpret_cast<const float4*>(w2s + jt * 32 + 8 * a + 4 * h);
        const float4 wb = *reinterpret_cast<const float4*>(w2s + HID + jt * 32 + 8 * a + 4 * h);
        const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float pv = pre[4 * a + b];
          const float hv = act_f<ACT>(pv);
          hv1[4 * a + b] = hv;
          dp[4 * a + b] = fmaf(was[b], d.x, wbs[b] * d.y) * act_d<ACT>(pv, hv);
        }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = dp[8 * s + i];
        mfma6(denc, lds_load_frag(awt, jt * 2 + s, lane), split3(x));
      }
      // ---- layout 2: rows = point, col = hidden.  h' is the transpose of h: it goes through a
      // per-wave LDS tile (16 ds_write_b32 + 4 ds_read_b128, both conflict-free with a 36-float
      // row) instead of being recomputed (12 MFMAs + 16 activations; tanh alone was 37 % of the
      // kernel's VALU instructions).  act' depends on h only (relu: h > 0; tanh: 1 - h^2).
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous readers of the tile are done
#pragma unroll
      for (int g = 0; g < 16; ++g) tr[drow(g, h) * 36 + r] = hv1[g];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const float w20 = w2s[jt * 32 + r], w21 = w2s[HID + jt * 32 + r];
      float hp[16];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 q4 = *reinterpret_cast<const float4*>(tr + r * 36 + 8 * a + 4 * h);
        hp[4 * a] = q4.x;
        hp[4 * a + 1] = q4.y;
        hp[4 * a + 2] = q4.z;
        hp[4 * a + 3] = q4.w;
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const float2 dq = *reinterpret_cast<const float2*>(dos + 2 * drow(g, h));
        const float hv = hp[g];
        const float dact = ACT == IMMOCO_ACT_RELU ? (hv > 0.f ? 1.f : 0.f) : 1.f - hv * hv;
        dp[g] = fmaf(w20, dq.x, w21 * dq.y) * dact;
      }
      f32x16 tmp = {0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = dp[8 * s + i];
        mfma6(dw1t[jt], et[s], split3(x));
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = hp[8 * s + i];
        mfma6(tmp, dt[s], split3(x));
      }
      dw2a[jt][0] += tmp[0];  // row o = 0 (lanes h = 0)
      dw2a[jt][1] += tmp[1];  // row o = 1
    }
    // ---- d enc: rows = feature (g&3) + 8(g>>2) + 4h, col = point
    if (valid) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int level = 4 * a + 2 * h;
        *reinterpret_cast<float2*>(din + p * ps + (int64_t)level * ls) = make_float2(denc[4 * a], denc[4 * a + 1]);
        *reinterpret_cast<float2*>(din + p * ps + (int64_t)(level + 1) * ls) =
            make_float2(denc[4 * a + 2], denc[4 * a + 3]);
      }
    }
  }
  if (wave_id >= n_tiles) return;  // this wave had no tile
  // ---- flush the weight gradients (once per wave): transpose dW1^T tiles through LDS so that every
  // atomic wave-instruction covers 256 contiguous bytes of dW1[j][k]
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < 16; ++g) tr[r * 33 + drow(g, h)] = dw1t[jt][g];  // [hidden r][feature k]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
      const int idx = it * 64 + lane;  // (hidden = idx>>5, feature = idx&31) of this 32x32 tile
      unsafeAtomicAdd(dw1 + (size_t)jt * 1024 + idx, tr[(idx >> 5) * 33 + (idx & 31)]);
    }
    if (h == 0) {
      unsafeAtomicAdd(dw2 + jt * 32 + r, dw2a[jt][0]);
      unsafeAtomicAdd(dw2 + HID + jt * 32 + r, dw2a[jt][1]);
    }
  }
}

static size_t fwd_smem(int hid) { return (size_t)(hid / 32) * 2 * 3 * 64 * 16 + (size_t)2 * hid * 4; }
static size_t bwd_smem(int hid) {
  return (size_t)(hid / 32) * 2 * 3 * 64 * 16 * 2 + (size_t)2 * hid * 4 + 4 * 64 * 4 + (size_t)4 * 32 * 36 * 4;
}

int launch_mlp_fwd_mfma(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                        const float* w1, const float* w2, float* out, hipStream_t st) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
  const int64_t n_tiles = cdiv(n, 32);
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 2048);
  const size_t sm = fwd_smem(cfg.n_hidden);
#define IMMOCO_FWD(H, A) mlp_fwd_mfma_kernel<H, A><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, out, n_tiles)
  if (cfg.n_hidden == 64 && cfg.activation == IMMOCO_ACT_TANH) IMMOCO_FWD(64, IMMOCO_ACT_TANH);
  else if (cfg.n_hidden == 64) IMMOCO_FWD(64, IMMOCO_ACT_RELU);
  else if (cfg.activation == IMMOCO_ACT_TANH) IMMOCO_FWD(256, IMMOCO_ACT_TANH);
  else IMMOCO_FWD(256, IMMOCO_ACT_RELU);
#undef IMMOCO_FWD
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

template <int HID, int ACT>
static int launch_bwd_t(const float* in, int64_t ps, int64_t ls, int64_t n, const float* w1, const float* w2,
                        const float* dout, float* din, float* dw1, float* dw2, hipStream_t st, int64_t dout_plane) {
  const int64_t n_tiles = cdiv(n, 32);
  // HID = 256 keeps 8 accumulator tiles per wave: one wave per SIMD (512-register budget)
  const int blocks_per_cu = HID == 64 ? 2 : 1;
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 256 * blocks_per_cu);
  const size_t sm = bwd_smem(HID);
  static bool attr_set = false;
  if (!attr_set) {
    IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_mfma_kernel<HID, ACT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    attr_set = true;
  }
  mlp_bwd_mfma_kernel<HID, ACT><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, n_tiles, dout_plane);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_mlp_bwd_mfma(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                        const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                        hipStream_t st, int64_t dout_plane) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
  if (cfg.n_hidden == 64 && cfg.activation == IMMOCO_ACT_TANH)
    return launch_bwd_t<64, IMMOCO_ACT_TANH>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  if (cfg.n_hidden == 64) return launch_bwd_t<64, IMMOCO_ACT_RELU>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  if (cfg.activation == IMMOCO_ACT_TANH)
    return launch_bwd_t<256, IMMOCO_ACT_TANH>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  return launch_bwd_t<256, IMMOCO_ACT_RELU>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
}

}  // namespace immoco
