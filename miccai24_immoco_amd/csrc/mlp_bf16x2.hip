// The bias-free MLP 32 -> HID -> 2 at (nearly) fp32 accuracy on the 16-bit matrix cores: every operand of a matrix
// product is split into TWO bf16 terms, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) - 16 significant bits with
// fp32's exponent range, so no scaling and no overflow - and a product A . B is three MFMAs,
//     A.lo * B.hi  +  A.hi * B.lo  +  A.hi * B.hi        (the lo * lo term is below 2^-17 of the product)
// accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (32 cycles for K = 16; the exact-fp32 v_mfma_f32_32x32x2_f32 of
// mlp_mfma.hip needs 64 cycles for K = 2: sixteen times the matrix-pipe time per product, five times after the
// three-fold split).  Relative error of a product term <= 2^-16.5 (4e-6 per operand), against 4.9e-4 of the single
// fp16 operand of mlp_f16.hip and 6e-8 of fp32; activations, their derivatives, dL/dhidden (two terms per hidden unit)
// and all accumulations are plain fp32, nothing is scaled, buffers stay fp32.  Kernel structure, operand maps, the
// accumulator-as-operand order and the transposing LDS reads are those of mlp_f16.hip (probe:
// tools/probe_mfma_f16.hip; the bf16 form has the same operand and result maps) with every operand in two halves.
//
// Selected by immoco_solver_cfg.mlp_bf16x2 / immoco_mlp_fwd_bf16x2 / immoco_mlp_bwd_bf16x2.  Replaces the network half
// of tinycudann.NetworkWithInputEncoding (reference src/models/immoco.py:11-25,60-65), which itself runs in fp16.
#include "kernels.hpp"

namespace immoco {

namespace {

typedef __bf16 bf2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf8v __attribute__((ext_vector_type(8)));
typedef __fp16 fh4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Op {        // 8 matrix elements as hi + lo bf16 terms
  uint4 hi, lo;
};

__device__ __forceinline__ f32x16 mfma_bf(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf8v*>(&a), *reinterpret_cast<const bf8v*>(&b),
                                                 c, 0, 0, 0);
}
// small terms first
__device__ __forceinline__ f32x16 mfma3(const Op& a, const Op& b, f32x16 c) {
  c = mfma_bf(a.lo, b.hi, c);
  c = mfma_bf(a.hi, b.lo, c);
  return mfma_bf(a.hi, b.hi, c);
}
__device__ __forceinline__ int drow16(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// two floats -> packed hi terms and packed lo terms (round to nearest even both times)
__device__ __forceinline__ void split2(float x, float y, uint32_t& hi, uint32_t& lo) {
  const f2v v = {x, y};
  const bf2v hb = __builtin_convertvector(v, bf2v);
  __builtin_memcpy(&hi, &hb, 4);
  const f2v r = {x - __uint_as_float(hi << 16), y - __uint_as_float(hi & 0xffff0000u)};
  const bf2v lb = __builtin_convertvector(r, bf2v);
  __builtin_memcpy(&lo, &lb, 4);
}
__device__ __forceinline__ Op split8(const float* v) {
  Op o;
  split2(v[0], v[1], o.hi.x, o.lo.x);
  split2(v[2], v[3], o.hi.y, o.lo.y);
  split2(v[4], v[5], o.hi.z, o.lo.z);
  split2(v[6], v[7], o.hi.w, o.lo.w);
  return o;
}

__device__ __forceinline__ float tanh_f(float x) {     // as in mlp_mfma.hip
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(ax * 2.885390082f);
  const float big = fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
  const float small = ax * fmaf(ax * ax, -0.33333334f, 1.f);
  return copysignf(ax < 0.04f ? small : big, x);
}
template <int ACT>
__device__ __forceinline__ float act_h(float pre) {
  return ACT == IMMOCO_ACT_RELU ? fmaxf(pre, 0.f) : tanh_f(pre);
}
template <int ACT>
__device__ __forceinline__ float act_dh(float hv) {
  return ACT == IMMOCO_ACT_RELU ? (hv > 0.f ? 1.f : 0.f) : fmaf(-hv, hv, 1.f);
}

constexpr int IMG_ROW = 72;                 // bytes per row of a [32][32] 16-bit image (see mlp_f16.hip's bank note)
constexpr int IMG_BYTES = 32 * IMG_ROW;

// W1 fragments in LDS (hi and lo arrays, one uint4 = 8 elements per entry):
//   AW [jt][s][lane]  elem i = W1[jt*32 + r][16s + 8h + i]
//   AWT[jt][s][lane]  elem i = W1[jt*32 + 16s + 8(i>>2) + 4h + (i&3)][k = r]
template <int HID>
__device__ __forceinline__ void build_w1_frags(const float* __restrict__ w1, uint4* aw_hi, uint4* aw_lo, uint4* awt_hi,
                                               uint4* awt_lo, int tid) {
  for (int c = tid; c < HID * 4; c += 256) {
    const int j = c >> 2, q = c & 3, s = q >> 1, h = q & 1, jt = j >> 5, rho = j & 31;
    const float4 a = reinterpret_cast<const float4*>(w1)[c * 2], b = reinterpret_cast<const float4*>(w1)[c * 2 + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const Op o = split8(v);
    aw_hi[(jt * 2 + s) * 64 + h * 32 + rho] = o.hi;
    aw_lo[(jt * 2 + s) * 64 + h * 32 + rho] = o.lo;
  }
  if (awt_hi) {
    for (int f = tid; f < HID * 4; f += 256) {
      const int lane = f & 63, s = (f >> 6) & 1, jt = f >> 7, k = lane & 31, h = lane >> 5;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = w1[(jt * 32 + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)) * 32 + k];
      const Op o = split8(v);
      awt_hi[f] = o.hi;
      awt_lo[f] = o.lo;
    }
  }
}
//   AW2[jt][s][lane]  elem i = W2[o = r][jt*32 + 16s + 8(i>>2) + 4h + (i&3)] for r < 2, else 0
template <int HID>
__device__ __forceinline__ void build_w2_frags(const float* __restrict__ w2, uint4* hi, uint4* lo, int tid) {
  for (int f = tid; f < HID * 4; f += 256) {
    const int lane = f & 63, s = (f >> 6) & 1, jt = f >> 7, r = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = r < 2 ? w2[r * HID + jt * 32 + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)] : 0.f;
    const Op o = split8(v);
    hi[f] = o.hi;
    lo[f] = o.lo;
  }
}

// enc of point p as the B operand of the first product: step s, elem i = enc[p][feature 16s + 8h + i]
struct EncRaw {
  float2 v[8];   // [4s + q] = level 8s + 4h + q
};
__device__ __forceinline__ void load_enc_raw(const float* in, int64_t ps, int64_t ls, int64_t p, int64_t n, int h,
                                             EncRaw& e) {
  const int64_t pc = p < n ? p : n - 1;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      e.v[4 * s + q] = *reinterpret_cast<const float2*>(in + pc * ps + (int64_t)(8 * s + 4 * h + q) * ls);
}
__device__ __forceinline__ void enc_frags(const EncRaw& e, bool valid, Op (&eb)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[2 * q] = valid ? e.v[4 * s + q].x : 0.f;
      v[2 * q + 1] = valid ? e.v[4 * s + q].y : 0.f;
    }
    eb[s] = split8(v);
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
template <int HID, int ACT>
__global__ __launch_bounds__(256) void mlp_fwd_bf16x2_kernel(const float* __restrict__ in, int64_t ps, int64_t ls,
                                                             int64_t n, const float* __restrict__ w1,
                                                             const float* __restrict__ w2, float* __restrict__ out,
                                                             int64_t n_tiles) {
  constexpr int NJT = HID / 32, NF = NJT * 2 * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* aw_hi = reinterpret_cast<uint4*>(smem);
  uint4* aw_lo = aw_hi + NF;
  uint4* aw2_hi = aw_lo + NF;
  uint4* aw2_lo = aw2_hi + NF;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  build_w1_frags<HID>(w1, aw_hi, aw_lo, nullptr, nullptr, threadIdx.x);
  build_w2_frags<HID>(w2, aw2_hi, aw2_lo, threadIdx.x);
  __syncthreads();
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  EncRaw nx;
  if (wave_id < n_tiles) load_enc_raw(in, ps, ls, wave_id * 32 + r, n, h, nx);
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const int64_t p = t * 32 + r;
    Op eb[2];
    enc_frags(nx, p < n, eb);
    if (t + n_waves < n_tiles) load_enc_raw(in, ps, ls, (t + n_waves) * 32 + r, n, h, nx);
    f32x16 o = {0.f};
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
      f32x16 pre = {0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Op a{aw_hi[(jt * 2 + s) * 64 + lane], aw_lo[(jt * 2 + s) * 64 + lane]};
        pre = mfma3(a, eb[s], pre);
      }
      float hv[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) hv[g] = act_h<ACT>(pre[g]);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Op a{aw2_hi[(jt * 2 + s) * 64 + lane], aw2_lo[(jt * 2 + s) * 64 + lane]};
        o = mfma3(a, split8(hv + 8 * s), o);
      }
    }
    if (p < n && h == 0) *reinterpret_cast<float2*>(out + p * 2) = make_float2(o[0], o[1]);
  }
}

// ---------------------------------------------------------------------------------------------
template <int HID, int ACT>
__global__ __launch_bounds__(256, HID == 64 ? 2 : 1) void mlp_bwd_bf16x2_kernel(
    const float* in /* may alias din */, int64_t ps, int64_t ls, int64_t n, const float* __restrict__ w1,
    const float* __restrict__ w2, const float* __restrict__ dout, float* din, float* __restrict__ dw1,
    float* __restrict__ dw2, int64_t n_tiles, int64_t dout_plane) {
  constexpr int NJT = HID / 32, NF = NJT * 2 * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* aw_hi = reinterpret_cast<uint4*>(smem);
  uint4* aw_lo = aw_hi + NF;
  uint4* awt_hi = aw_lo + NF;
  uint4* awt_lo = awt_hi + NF;
  float* w2s = reinterpret_cast<float*>(awt_lo + NF);              // [2][HID], exact fp32
  unsigned char* wv_all = reinterpret_cast<unsigned char*>(w2s + 2 * HID);
  constexpr int WAVE_BYTES = 6 * IMG_BYTES + 256;                  // (enc, dpre, h) x (hi, lo) images + dout tile hi / lo
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  unsigned char* wv = wv_all + wave * WAVE_BYTES;
  unsigned char* img_e = wv;                      // + IMG_BYTES: lo
  unsigned char* img_d = wv + 2 * IMG_BYTES;
  unsigned char* img_h = wv + 4 * IMG_BYTES;
  uint16_t* dm = reinterpret_cast<uint16_t*>(wv + 6 * IMG_BYTES);  // hi [2][32], lo [2][32]
  build_w1_frags<HID>(w1, aw_hi, aw_lo, awt_hi, awt_lo, threadIdx.x);
  for (int i = threadIdx.x; i < 2 * HID; i += 256) w2s[i] = w2[i];
  __syncthreads();

  // transposing reads (mlp_f16.hip): lane (column r, half h) receives rows 16s + 8h + {0..7} of column r
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int tr_off = (8 * h + tr_q) * IMG_ROW + 32 * ((lane >> 4) & 1) + 8 * tr_p;
  auto tr_read8 = [&](const unsigned char* img, int s) -> uint4 {
    const unsigned char* a = img + tr_off + s * 16 * IMG_ROW;
    const fh4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4v*)a);
    const fh4v hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4v*)(a + 4 * IMG_ROW));
    uint4 o;
    __builtin_memcpy(&o.x, &lo, 8);
    __builtin_memcpy(&o.z, &hi, 8);
    return o;
  };
  auto tr_op = [&](const unsigned char* img, int s) -> Op { return Op{tr_read8(img, s), tr_read8(img + IMG_BYTES, s)}; };

  f32x16 dw1t[NJT];
  f32x16 dw2acc = {0.f};
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) dw1t[jt] = (f32x16){0.f};

  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  EncRaw nx;
  float2 nx_d = make_float2(0.f, 0.f);
  auto load_raw = [&](int64_t tt) {
    const int64_t q = tt * 32 + r;
    const int64_t qc = q < n ? q : n - 1;
    const float mq = q < n ? 1.f : 0.f;
    load_enc_raw(in, ps, ls, q, n, h, nx);
    if (dout_plane) {  // wave-uniform
      nx_d = make_float2(dout[qc] * mq, dout[dout_plane + qc] * mq);
    } else {
      const float2 dv = *reinterpret_cast<const float2*>(dout + qc * 2);
      nx_d = make_float2(dv.x * mq, dv.y * mq);
    }
  };
  if (wave_id < n_tiles) load_raw(wave_id);
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const int64_t p = t * 32 + r;
    const bool valid = p < n;
    Op eb[2];
    enc_frags(nx, valid, eb);
    const float d0 = nx_d.x, d1 = nx_d.y;
    if (t + n_waves < n_tiles) load_raw(t + n_waves);
    // ---- stage the enc tile ([point][feature]) and the dout tile ([o][point]), hi and lo, for the transposed products
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      *reinterpret_cast<uint2*>(img_e + r * IMG_ROW + 32 * s + 16 * h) = make_uint2(eb[s].hi.x, eb[s].hi.y);
      *reinterpret_cast<uint2*>(img_e + r * IMG_ROW + 32 * s + 16 * h + 8) = make_uint2(eb[s].hi.z, eb[s].hi.w);
      *reinterpret_cast<uint2*>(img_e + IMG_BYTES + r * IMG_ROW + 32 * s + 16 * h) = make_uint2(eb[s].lo.x, eb[s].lo.y);
      *reinterpret_cast<uint2*>(img_e + IMG_BYTES + r * IMG_ROW + 32 * s + 16 * h + 8) = make_uint2(eb[s].lo.z, eb[s].lo.w);
    }
    if (h == 0) {
      uint32_t dh, dl;
      split2(d0, d1, dh, dl);
      dm[r] = (uint16_t)(dh & 0xffffu);
      dm[32 + r] = (uint16_t)(dh >> 16);
      dm[64 + r] = (uint16_t)(dl & 0xffffu);
      dm[96 + r] = (uint16_t)(dl >> 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    Op ea[2], da[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ea[s] = tr_op(img_e, s);
      da[s].hi = *reinterpret_cast<const uint4*>(dm + (r & 1) * 32 + 16 * s + 8 * h);
      da[s].lo = *reinterpret_cast<const uint4*>(dm + 64 + (r & 1) * 32 + 16 * s + 8 * h);
    }
    f32x16 denc = {0.f};
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
      // ---- L1: rows = hidden, col = point
      f32x16 pre = {0.f};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Op a{aw_hi[(jt * 2 + s) * 64 + lane], aw_lo[(jt * 2 + s) * 64 + lane]};
        pre = mfma3(a, eb[s], pre);
      }
      float hv[16], dp[16];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 wa = *reinterpret_cast<const float4*>(w2s + jt * 32 + 8 * a + 4 * h);
        const float4 wb = *reinterpret_cast<const float4*>(w2s + HID + jt * 32 + 8 * a + 4 * h);
        const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float hh = act_h<ACT>(pre[4 * a + b]);
          hv[4 * a + b] = hh;
          dp[4 * a + b] = fmaf(was[b], d0, wbs[b] * d1) * act_dh<ACT>(hh);
        }
      }
      Op dpo[2], hpo[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        dpo[s] = split8(dp + 8 * s);
        hpo[s] = split8(hv + 8 * s);
      }
      // d enc^T[k][p] += sum_j W1[j][k] dpre[j][p]
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Op a{awt_hi[(jt * 2 + s) * 64 + lane], awt_lo[(jt * 2 + s) * 64 + lane]};
        denc = mfma3(a, dpo[s], denc);
      }
      // ---- L2: transpose dpre and h through the per-wave images (registers 4a .. 4a+3 = hidden 8a + 4h + 0..3;
      // elements 2q, 2q+1 of a step's operand are one packed word)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int s = a >> 1;
        const int off = r * IMG_ROW + 2 * (8 * a + 4 * h);
        const uint4 &dh4 = dpo[s].hi, &dl4 = dpo[s].lo, &hh4 = hpo[s].hi, &hl4 = hpo[s].lo;
        const bool up = a & 1;   // words (x, y) or (z, w) of the step's operand
        *reinterpret_cast<uint2*>(img_d + off) = up ? make_uint2(dh4.z, dh4.w) : make_uint2(dh4.x, dh4.y);
        *reinterpret_cast<uint2*>(img_d + IMG_BYTES + off) = up ? make_uint2(dl4.z, dl4.w) : make_uint2(dl4.x, dl4.y);
        *reinterpret_cast<uint2*>(img_h + off) = up ? make_uint2(hh4.z, hh4.w) : make_uint2(hh4.x, hh4.y);
        *reinterpret_cast<uint2*>(img_h + IMG_BYTES + off) = up ? make_uint2(hl4.z, hl4.w) : make_uint2(hl4.x, hl4.y);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const bool mine = (r >> 1) == jt;
      const uint4 z4 = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const Op db = tr_op(img_d, s), hb = tr_op(img_h, s);
        dw1t[jt] = mfma3(ea[s], db, dw1t[jt]);                                        // dW1^T[k][j] += sum_p enc[p][k] dpre[p][j]
        const Op dam{mine ? da[s].hi : z4, mine ? da[s].lo : z4};
        dw2acc = mfma3(dam, hb, dw2acc);                                              // dW2[o][j] += sum_p dout[p][o] h[p][j]
      }
    }
    if (valid) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int level = 4 * a + 2 * h;
        *reinterpret_cast<float2*>(din + p * ps + (int64_t)level * ls) = make_float2(denc[4 * a], denc[4 * a + 1]);
        *reinterpret_cast<float2*>(din + p * ps + (int64_t)(level + 1) * ls) = make_float2(denc[4 * a + 2], denc[4 * a + 3]);
      }
    }
  }
  // ---- flush the weight gradients once per WORKGROUP (as in mlp_f16.hip)
  constexpr int TL = 33;
  constexpr int WAVE_F = WAVE_BYTES / 4;
  float* ft_all = reinterpret_cast<float*>(wv_all);
  static_assert(32 * TL * 4 <= WAVE_BYTES, "flush tile does not fit the per-wave LDS area");
  float* ft = ft_all + wave * WAVE_F;
  auto flush_tile = [&](const f32x16& acc, float* dst) {
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 16; ++g) ft[r * TL + drow16(g, h)] = acc[g];   // parked transposed: [hidden][feature]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = (4 * wave + k) * 64 + lane;
      const int off = (idx >> 5) * TL + (idx & 31);
      const float v = (ft_all[off] + ft_all[WAVE_F + off]) + (ft_all[2 * WAVE_F + off] + ft_all[3 * WAVE_F + off]);
      unsafeAtomicAdd(dst + idx, v);
    }
  };
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) flush_tile(dw1t[jt], dw1 + (size_t)jt * 1024);
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 16; ++g) ft[drow16(g, h) * TL + r] = dw2acc[g];
  __syncthreads();
  for (int idx = threadIdx.x; idx < 2 * NJT * 32; idx += 256) {
    const int row = idx >> 5, col = idx & 31, off = row * TL + col;
    const float v = (ft_all[off] + ft_all[WAVE_F + off]) + (ft_all[2 * WAVE_F + off] + ft_all[3 * WAVE_F + off]);
    unsafeAtomicAdd(dw2 + (row & 1) * HID + (row >> 1) * 32 + col, v);
  }
}

static size_t bx_fwd_smem(int hid) { return (size_t)(hid / 32) * 2 * 64 * 16 * 4; }
static size_t bx_bwd_smem(int hid) {
  return (size_t)(hid / 32) * 2 * 64 * 16 * 4 + (size_t)2 * hid * 4 + (size_t)4 * (6 * IMG_BYTES + 256);
}

int launch_mlp_fwd_bf16x2(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                          const float* w1, const float* w2, float* out, hipStream_t st) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
  const int64_t n_tiles = cdiv(n, 32);
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 512);
  const size_t sm = bx_fwd_smem(cfg.n_hidden);
#define IMMOCO_FWD(H, A)                                                                                                  \
  do {                                                                                                                    \
    static bool attr = false;                                                                                             \
    if (!attr) {                                                                                                          \
      IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_fwd_bf16x2_kernel<H, A>),                   \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));                        \
      attr = true;                                                                                                        \
    }                                                                                                                     \
    mlp_fwd_bf16x2_kernel<H, A><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, out, n_tiles);                             \
  } while (0)
  if (cfg.n_hidden == 64 && cfg.activation == IMMOCO_ACT_TANH) IMMOCO_FWD(64, IMMOCO_ACT_TANH);
  else if (cfg.n_hidden == 64) IMMOCO_FWD(64, IMMOCO_ACT_RELU);
  else if (cfg.activation == IMMOCO_ACT_TANH) IMMOCO_FWD(256, IMMOCO_ACT_TANH);
  else IMMOCO_FWD(256, IMMOCO_ACT_RELU);
#undef IMMOCO_FWD
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

template <int HID, int ACT>
static int launch_bwd_bx_t(const float* in, int64_t ps, int64_t ls, int64_t n, const float* w1, const float* w2,
                           const float* dout, float* din, float* dw1, float* dw2, hipStream_t st, int64_t dout_plane) {
  const int64_t n_tiles = cdiv(n, 32);
  const int blocks_per_cu = HID == 64 ? 2 : 1;
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 256 * blocks_per_cu);
  const size_t sm = bx_bwd_smem(HID);
  static bool attr_set = false;
  if (!attr_set) {
    IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_bf16x2_kernel<HID, ACT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    attr_set = true;
  }
  mlp_bwd_bf16x2_kernel<HID, ACT><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, n_tiles, dout_plane);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_mlp_bwd_bf16x2(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                          const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                          hipStream_t st, int64_t dout_plane) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
  if (cfg.n_hidden == 64 && cfg.activation == IMMOCO_ACT_TANH)
    return launch_bwd_bx_t<64, IMMOCO_ACT_TANH>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  if (cfg.n_hidden == 64)
    return launch_bwd_bx_t<64, IMMOCO_ACT_RELU>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  if (cfg.activation == IMMOCO_ACT_TANH)
    return launch_bwd_bx_t<256, IMMOCO_ACT_TANH>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  return launch_bwd_bx_t<256, IMMOCO_ACT_RELU>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
}

}  // namespace immoco

using namespace immoco;

extern "C" int immoco_mlp_fwd_bf16x2(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                                     int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                                     float* out, void* stream) {
  int rc = check_mlp_cfg(cfg);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (in && w1 && w2 && out)), "mlp_fwd_bf16x2: NULL buffer");
  return launch_mlp_fwd_bf16x2(*cfg, in, in_point_stride, in_level_stride, n, w1, w2, out, as_stream(stream));
}

extern "C" int immoco_mlp_bwd_bf16x2(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                                     int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                                     const float* dout, float* din, float* dw1, float* dw2, void* stream) {
  int rc = check_mlp_cfg(cfg);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (in && w1 && w2 && dout && din && dw1 && dw2)), "mlp_bwd_bf16x2: NULL buffer");
  return launch_mlp_bwd_bf16x2(*cfg, in, in_point_stride, in_level_stride, n, w1, w2, dout, din, dw1, dw2,
                               as_stream(stream), 0);
}
