// The bias-free MLP 32 -> HID -> 2 in tiny-cuda-nn's OWN operand precision: fp16 operands, fp32 accumulation on
// gfx950's v_mfma_f32_32x32x16_f16 (32 cycles for 16x the K of the exact-fp32 v_mfma_f32_32x32x2_f32's 64).
//
// The reference instantiates both networks with `__half` precision (tinycudann FullyFusedMLP / CutlassMLP, call
// sites /root/reference/src/models/immoco.py:11-25,60-65; torch binding: loss_scale = 128).  This
// mode (`immoco_solver_cfg.mlp_fp16`, `immoco_mlp_fwd_half` / `immoco_mlp_bwd_half`) rounds exactly the operands
// tcnn rounds - the encoding, W1, the hidden activations, W2, dL/dout * loss_scale, dL/dpre - and keeps everything
// tcnn keeps wider or equal: products accumulate in fp32 (tcnn's fully fused kernel accumulates the hidden layer in
// fp16), activations are evaluated in fp32, outputs / dL/denc / weight gradients leave in fp32.  The oracle states
// the same arithmetic (oracle/immoco_oracle.py:_MLPHalf).
//
// One wave owns a tile of 32 points.  Operand maps of v_mfma_f32_32x32x16_f16 (lane l, r = l & 31, h = l >> 5;
// checked with exact integer data by tools/probe_mfma_f16.hip):
//   A: 8 halves A[row r][k = 8h + i]     B: 8 halves B[k = 8h + i][col r]     D: reg g = D[(g&3) + 8(g>>2) + 4h][r]
// Two layouts of the hidden tile:
//   L1  X[hidden j][point p] = W1 . enc^T (rows in registers, point on the lane).  Everything that sums over the
//       hidden units takes X straight from the accumulator as the B operand of the next MFMA (fp16 pairs of
//       registers 8s .. 8s+7 are k-step s; the other operand is stored in that permuted k order):
//         out^T [o][p] = W2 . h            d enc^T [k][p] = W1^T . dpre
//   L2  everything that sums over the POINTS (the lane index of L1) needs the transpose: each lane packs its
//       dpre / h registers 4a .. 4a+3 (four consecutive hidden units) into 8 bytes and stores them at
//       [point][hidden 8a + 4h] of a per-wave [32 points][32 hidden] fp16 image; ds_read_b64_tr_b16 (gfx950's
//       transposing LDS read: a 16-lane group reads a 4 x 16 block, lane i receives column i) hands the image back
//       with the hidden unit on the lane and four points per register pair - the MFMA operand layout - so the
//       transposed tiles cost no VALU work at all:
//         dW1^T [k][j] = enc^T . dpre'     dW2^T [2 jt + o][j] = dout^T . h'
//       (the encoding tile goes through the same kind of image; dW2 of all hidden tiles accumulates in ONE
//       accumulator: hidden tile jt's dout^T sits in rows 2 jt, 2 jt + 1 of the A operand, zero elsewhere).
// Weight fragments live in LDS (built once per workgroup, converted to fp16 there); weight gradients stay in
// accumulators over all of a wave's tiles and are flushed once per workgroup.
#include "kernels.hpp"

namespace immoco {

namespace {

typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef __fp16 fh4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mfma16(const h8v& a, const h8v& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int drow16(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// two floats -> two halves, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ h2v pk(float a, float b) {
  const f2v x = {a, b};
  return __builtin_convertvector(x, h2v);
}
__device__ __forceinline__ h8v pk8(const float* v) {
  const h2v a = pk(v[0], v[1]), b = pk(v[2], v[3]), c = pk(v[4], v[5]), d = pk(v[6], v[7]);
  return (h8v){a[0], a[1], b[0], b[1], c[0], c[1], d[0], d[1]};
}
__device__ __forceinline__ float rh(float x) { return (float)(_Float16)x; }   // round through fp16

// tanh as in mlp_mfma.hip (relative accuracy 3e-6, far inside fp16's 4.9e-4)
__device__ __forceinline__ float tanh_f(float x) {
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

constexpr int IMG_ROW = 72;                 // bytes per row of a [32][32] fp16 image (64 + 8: see the bank note below)
constexpr int IMG_BYTES = 32 * IMG_ROW;     // 2304
// Bank note: ds_write_b64 by lane (point p, half h) at p * 72 + 2 * (8a + 4h): 18 dwords per row -> lanes p and
// p + 16 share banks (2-way, hidden under the store's own cycles); ds_read_b64_tr_b16 of a 32-lane half reads four
// rows of 64 bytes 72 bytes apart: 6 of 64 banks are hit twice.  64-byte rows would be 16-way on the stores.

// W1 fragments in LDS, all fp16:
//   AW [jt][s][lane]  elem i = W1[jt*32 + r][16s + 8h + i]                      (A of L1's pre; 8 consecutive floats)
//   AWT[jt][s][lane]  elem i = W1[jt*32 + 16s + 8(i>>2) + 4h + (i&3)][k = r]    (A of d enc^T, accumulator k order)
template <int HID>
__device__ __forceinline__ void build_w1_frags(const float* __restrict__ w1, h8v* aw, h8v* awt, int tid) {
  for (int c = tid; c < HID * 4; c += 256) {
    const int j = c >> 2, q = c & 3, s = q >> 1, h = q & 1, jt = j >> 5, rho = j & 31;
    const float4 a = reinterpret_cast<const float4*>(w1)[c * 2], b = reinterpret_cast<const float4*>(w1)[c * 2 + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    aw[(jt * 2 + s) * 64 + h * 32 + rho] = pk8(v);
  }
  if (awt) {
    for (int f = tid; f < HID * 4; f += 256) {
      const int lane = f & 63, s = (f >> 6) & 1, jt = f >> 7, k = lane & 31, h = lane >> 5;
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = w1[(jt * 32 + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)) * 32 + k];
      awt[f] = pk8(v);
    }
  }
}
// the AW fragments of `njl` hidden tiles starting at tile jt0 only (split backward, MODE 2): aw_local[jl][s][lane]
__device__ __forceinline__ void build_w1_frags_range(const float* __restrict__ w1, h8v* aw_local, int jt0, int njl, int tid) {
  for (int c = tid; c < njl * 32 * 4; c += 256) {
    const int jloc = c >> 2, q = c & 3, s = q >> 1, h = q & 1, jl = jloc >> 5, rho = jloc & 31;
    const int cg = (jt0 * 32 + jloc) * 4 + q;   // index of the 8-float chunk in W1
    const float4 a = reinterpret_cast<const float4*>(w1)[cg * 2], b = reinterpret_cast<const float4*>(w1)[cg * 2 + 1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    aw_local[(jl * 2 + s) * 64 + h * 32 + rho] = pk8(v);
  }
}
//   AW2[jt][s][lane]  elem i = W2[o = r][jt*32 + 16s + 8(i>>2) + 4h + (i&3)] for r < 2, else 0   (A of out^T)
template <int HID>
__device__ __forceinline__ void build_w2_frags(const float* __restrict__ w2, h8v* aw2, int tid) {
  for (int f = tid; f < HID * 4; f += 256) {
    const int lane = f & 63, s = (f >> 6) & 1, jt = f >> 7, r = lane & 31, h = lane >> 5;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      v[i] = r < 2 ? w2[r * HID + jt * 32 + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)] : 0.f;
    aw2[f] = pk8(v);
  }
}

// enc of point p as the B operand of L1: step s, elem i = enc[p][feature 16s + 8h + i] = level 8s + 4h + (i>>1).
// EH = false: the encoding is stored in fp32 (a float2 per point and level; strides in floats) and rounded here;
// EH = true: it is stored as packed halves (one 4-byte word per point and level; strides in 4-byte words, the
// solver's layout with cfg.mlp_fp16: tiny-cuda-nn's encoding output is fp16 too) and the four words of a k-step
// ARE the operand.
template <bool EH>
struct EncRaw {
  float2 v[EH ? 1 : 8];      // [4s + q] = level 8s + 4h + q
  uint32_t w[EH ? 8 : 1];
};
template <bool EH>
__device__ __forceinline__ void load_enc_raw(const float* in, int64_t ps, int64_t ls, int64_t p, int64_t n, int h,
                                             EncRaw<EH>& e) {
  const int64_t pc = p < n ? p : n - 1;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (EH) e.w[EH ? 4 * s + q : 0] = reinterpret_cast<const uint32_t*>(in)[pc * ps + (int64_t)(8 * s + 4 * h + q) * ls];
      else e.v[EH ? 0 : 4 * s + q] = *reinterpret_cast<const float2*>(in + pc * ps + (int64_t)(8 * s + 4 * h + q) * ls);
    }
}
template <bool EH>
__device__ __forceinline__ void enc_frags(const EncRaw<EH>& e, bool valid, h8v (&eb)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (EH) {
      uint4 q4;
      q4.x = valid ? e.w[EH ? 4 * s : 0] : 0u;
      q4.y = valid ? e.w[EH ? 4 * s + 1 : 0] : 0u;
      q4.z = valid ? e.w[EH ? 4 * s + 2 : 0] : 0u;
      q4.w = valid ? e.w[EH ? 4 * s + 3 : 0] : 0u;
      eb[s] = *reinterpret_cast<const h8v*>(&q4);
    } else {
      float v[8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v[2 * q] = valid ? e.v[EH ? 0 : 4 * s + q].x : 0.f;
        v[2 * q + 1] = valid ? e.v[EH ? 0 : 4 * s + q].y : 0.f;
      }
      eb[s] = pk8(v);
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// forward: out[p][0..1] = W2h . half(act(W1h . half(enc[p])))
template <int HID, int ACT, bool EH>
__global__ __launch_bounds__(256) void mlp_fwd_f16_kernel(const float* __restrict__ in, int64_t ps, int64_t ls,
                                                          int64_t n, const float* __restrict__ w1,
                                                          const float* __restrict__ w2, float* __restrict__ out,
                                                          int64_t n_tiles) {
  constexpr int NJT = HID / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  h8v* aw = reinterpret_cast<h8v*>(smem);      // [NJT][2][64]
  h8v* aw2 = aw + NJT * 2 * 64;                // [NJT][2][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  build_w1_frags<HID>(w1, aw, nullptr, threadIdx.x);
  build_w2_frags<HID>(w2, aw2, threadIdx.x);
  __syncthreads();
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  EncRaw<EH> nx;
  if (wave_id < n_tiles) load_enc_raw<EH>(in, ps, ls, wave_id * 32 + r, n, h, nx);
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const int64_t p = t * 32 + r;
    h8v eb[2];
    enc_frags<EH>(nx, p < n, eb);
    if (t + n_waves < n_tiles) load_enc_raw<EH>(in, ps, ls, (t + n_waves) * 32 + r, n, h, nx);
    f32x16 o = {0.f};
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
      f32x16 pre = {0.f};
      pre = mfma16(aw[(jt * 2) * 64 + lane], eb[0], pre);
      pre = mfma16(aw[(jt * 2 + 1) * 64 + lane], eb[1], pre);
      float hv[16];
#pragma unroll
      for (int g = 0; g < 16; ++g) hv[g] = act_h<ACT>(pre[g]);
      o = mfma16(aw2[(jt * 2) * 64 + lane], pk8(hv), o);
      o = mfma16(aw2[(jt * 2 + 1) * 64 + lane], pk8(hv + 8), o);
    }
    // rows o = 0, 1 of out^T: registers 0, 1 of the lanes h = 0
    if (p < n && h == 0) *reinterpret_cast<float2*>(out + p * 2) = make_float2(o[0], o[1]);
  }
}

// ---------------------------------------------------------------------------------------------
// backward.  dout is scaled by `scale` (tcnn's loss scale) before it is rounded to fp16; d enc and the weight
// gradients are unscaled in fp32 on the way out.
// MODE 0: the whole backward in one kernel.  MODE 1 / 2 (round 4, the 256-wide net in the solver): the same arithmetic as
// two kernels that fit BESIDE the motion grid's encode backward instead of one that needs a CU to itself -
//   MODE 1: d enc only (no transposed products, no weight-gradient accumulators: 34 KB of LDS);
//   MODE 2: dW1 / dW2 only, for the NJW hidden tiles blockIdx.y selects (grid.y = NJT / NJW): two accumulators instead
//           of eight.  Every value is computed by the same instructions in the same order as in MODE 0.
constexpr int F16_NJW = 2;
template <int HID, int ACT, bool EH, int MODE = 0>
__global__ __launch_bounds__(256, MODE == 0 ? (HID == 64 ? 2 : 1) : 3) void mlp_bwd_f16_kernel(
    const float* in /* may alias din (MODE 0) */, int64_t ps, int64_t ls, int64_t n, const float* __restrict__ w1,
    const float* __restrict__ w2, const float* __restrict__ dout, float* din, float* __restrict__ dw1,
    float* __restrict__ dw2, int64_t n_tiles, int64_t dout_plane, float scale, const float* __restrict__ dout2 = nullptr) {
  constexpr int NJT = HID / 32;
  constexpr int NJL = MODE == 2 ? F16_NJW : NJT;                   // hidden tiles this workgroup works on
  constexpr int UNR = MODE == 1 ? 2 : NJL;                         // unroll factor of the loop over them
  const int jt0 = MODE == 2 ? (int)blockIdx.y * F16_NJW : 0;       // first of them
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // MODE 2 keeps only its own hidden tiles' W1 fragments and no W1^T fragments (34 KB instead of 62: the kernel has to
  // fit beside the encode backward's workgroups)
  h8v* aw = reinterpret_cast<h8v*>(smem);                          // [NJT][2][64]   (MODE 2: [NJL][2][64], local tile index)
  h8v* awt = aw + (MODE == 2 ? NJL : NJT) * 2 * 64;                // [NJT][2][64]   (MODE 2: none)
  float* w2s = reinterpret_cast<float*>(awt + (MODE == 2 ? 0 : NJT) * 2 * 64);   // [2][HID], fp16-rounded values
  unsigned char* wv_all = reinterpret_cast<unsigned char*>(w2s + 2 * HID);
  constexpr int WAVE_BYTES = 3 * IMG_BYTES + 128;                  // three images + the dout tile [2][32] fp16
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  unsigned char* wv = wv_all + wave * WAVE_BYTES;
  unsigned char* img_e = wv;                   // enc tile   [point][feature]
  unsigned char* img_d = wv + IMG_BYTES;       // dpre tile  [point][hidden of the current jt]
  unsigned char* img_h = wv + 2 * IMG_BYTES;   // h tile     [point][hidden of the current jt]
  _Float16* dm = reinterpret_cast<_Float16*>(wv + 3 * IMG_BYTES);  // [2][32]
  if (MODE == 2) build_w1_frags_range(w1, aw, jt0, NJL, threadIdx.x);
  else build_w1_frags<HID>(w1, aw, awt, threadIdx.x);
  for (int i = threadIdx.x; i < 2 * HID; i += 256) w2s[i] = rh(w2[i]);
  __syncthreads();

  // transposing reads: lane (column c = r, half h) receives rows 16s + 8h + {0..3} (first read) and + {4..7}
  // (second) of column r: lane 4q + p' of its 16-lane group supplies row q, columns 4p' .. 4p'+3 of the block
  const int tr_q = (lane & 15) >> 2, tr_p = lane & 3;
  const int tr_off = (8 * h + tr_q) * IMG_ROW + 32 * ((lane >> 4) & 1) + 8 * tr_p;
  auto tr_read8 = [&](const unsigned char* img, int s) -> h8v {
    const unsigned char* a = img + tr_off + s * 16 * IMG_ROW;
    const fh4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4v*)a);
    const fh4v hi =
        __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4v*)(a + 4 * IMG_ROW));
    return (h8v){(_Float16)lo[0], (_Float16)lo[1], (_Float16)lo[2], (_Float16)lo[3],
                 (_Float16)hi[0], (_Float16)hi[1], (_Float16)hi[2], (_Float16)hi[3]};
  };

  f32x16 dw1t[MODE == 1 ? 1 : NJL];   // dW1^T tiles (x scale): rows k (features), col = hidden (jt0 + jl)*32 + r
  f32x16 dw2acc = {0.f};  // dW2 (x scale): row 2 jt + o, col = hidden r of tile jt
#pragma unroll
  for (int jl = 0; jl < (MODE == 1 ? 1 : NJL); ++jl) dw1t[jl] = (f32x16){0.f};

  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  EncRaw<EH> nx;
  float2 nx_d = make_float2(0.f, 0.f);
  auto load_raw = [&](int64_t tt) {
    const int64_t q = tt * 32 + r;
    const int64_t qc = q < n ? q : n - 1;
    const float mq = q < n ? 1.f : 0.f;
    load_enc_raw<EH>(in, ps, ls, q, n, h, nx);
    if (dout_plane) {  // wave-uniform; dout2: a second planar addend (the warp backward's share of dL/dimage)
      nx_d = make_float2(dout[qc], dout[dout_plane + qc]);
      if (MODE != 0 && dout2) nx_d = make_float2(dout2[qc] + nx_d.x, dout2[dout_plane + qc] + nx_d.y);
      nx_d = make_float2(nx_d.x * mq, nx_d.y * mq);
    } else {
      const float2 dv = *reinterpret_cast<const float2*>(dout + qc * 2);
      nx_d = make_float2(dv.x * mq, dv.y * mq);
    }
  };
  if (wave_id < n_tiles) load_raw(wave_id);
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const int64_t p = t * 32 + r;
    const bool valid = p < n;
    h8v eb[2];
    enc_frags<EH>(nx, valid, eb);
    const h2v dpk = pk(nx_d.x * scale, nx_d.y * scale);
    const float d0 = (float)dpk[0], d1 = (float)dpk[1];
    if (t + n_waves < n_tiles) load_raw(t + n_waves);
    // ---- stage the enc tile ([point][feature], 16 bytes per step) and the dout tile ([o][point]) for the
    // transposed products
    h8v ea[2], da[2];
    if (MODE != 1) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous tile's readers are done
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint4 q4 = *reinterpret_cast<const uint4*>(&eb[s]);
        *reinterpret_cast<uint2*>(img_e + r * IMG_ROW + 32 * s + 16 * h) = make_uint2(q4.x, q4.y);
        *reinterpret_cast<uint2*>(img_e + r * IMG_ROW + 32 * s + 16 * h + 8) = make_uint2(q4.z, q4.w);
      }
      if (h == 0) {
        dm[r] = dpk[0];
        dm[32 + r] = dpk[1];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      // A operands of the transposed products (kept for all hidden tiles):
      //   enc^T: row = feature r, k = points 16s + 8h + i;   dout^T: row o = r & 1, same k
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        ea[s] = tr_read8(img_e, s);
        da[s] = *reinterpret_cast<const h8v*>(dm + (r & 1) * 32 + 16 * s + 8 * h);
      }
    }
    f32x16 denc = {0.f};
    // (MODE 1 keeps no per-tile register arrays, so it need not be unrolled 8 times: fully unrolled the compiler hoists
    // every tile's fragment loads and spills 1.3 KB per lane at this kernel's 168-register budget)
#pragma unroll UNR
    for (int jl = 0; jl < NJL; ++jl) {
      const int jt = jt0 + jl;
      // ---- L1: rows = hidden, col = point
      f32x16 pre = {0.f};
      const int ja = MODE == 2 ? jl : jt;   // index of the tile's fragments in LDS
      pre = mfma16(aw[(ja * 2) * 64 + lane], eb[0], pre);
      pre = mfma16(aw[(ja * 2 + 1) * 64 + lane], eb[1], pre);
      h2v hp[8], dp[8];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 wa = *reinterpret_cast<const float4*>(w2s + jt * 32 + 8 * a + 4 * h);
        const float4 wb = *reinterpret_cast<const float4*>(w2s + HID + jt * 32 + 8 * a + 4 * h);
        const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
          const int g = 4 * a + 2 * b2;
          const h2v hh = pk(act_h<ACT>(pre[g]), act_h<ACT>(pre[g + 1]));
          const float h0 = (float)hh[0], h1 = (float)hh[1];   // the STORED fp16 activation drives act'
          const float p0 = fmaf(was[2 * b2], d0, wbs[2 * b2] * d1) * act_dh<ACT>(h0);
          const float p1 = fmaf(was[2 * b2 + 1], d0, wbs[2 * b2 + 1] * d1) * act_dh<ACT>(h1);
          hp[2 * a + b2] = hh;
          dp[2 * a + b2] = pk(p0, p1);
        }
      }
      // d enc^T[k][p] += sum_j W1[j][k] dpre[j][p]: dpre from the registers (accumulator k order)
      if (MODE != 2) {
        const h8v b0 = {dp[0][0], dp[0][1], dp[1][0], dp[1][1], dp[2][0], dp[2][1], dp[3][0], dp[3][1]};
        const h8v b1 = {dp[4][0], dp[4][1], dp[5][0], dp[5][1], dp[6][0], dp[6][1], dp[7][0], dp[7][1]};
        denc = mfma16(awt[(jt * 2) * 64 + lane], b0, denc);
        denc = mfma16(awt[(jt * 2 + 1) * 64 + lane], b1, denc);
      }
      if (MODE == 1) continue;
      // ---- L2: transpose dpre and h through the per-wave images: registers 4a .. 4a+3 = hidden 8a + 4h + (0..3)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous hidden tile's transposed reads are done
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const h4v d4 = {dp[2 * a][0], dp[2 * a][1], dp[2 * a + 1][0], dp[2 * a + 1][1]};
        const h4v h4 = {hp[2 * a][0], hp[2 * a][1], hp[2 * a + 1][0], hp[2 * a + 1][1]};
        *reinterpret_cast<h4v*>(img_d + r * IMG_ROW + 2 * (8 * a + 4 * h)) = d4;
        *reinterpret_cast<h4v*>(img_h + r * IMG_ROW + 2 * (8 * a + 4 * h)) = h4;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const bool mine = (r >> 1) == jt;   // dW2 rows 2 jt, 2 jt + 1 belong to this hidden tile
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const h8v db = tr_read8(img_d, s), hb = tr_read8(img_h, s);
        const h8v zero = {0, 0, 0, 0, 0, 0, 0, 0};
        dw1t[MODE == 1 ? 0 : jl] = mfma16(ea[s], db, dw1t[MODE == 1 ? 0 : jl]);   // dW1^T[k][j] += sum_p enc[p][k] dpre[p][j]
        dw2acc = mfma16(mine ? da[s] : zero, hb, dw2acc);       // dW2[o][j]   += sum_p dout[p][o] h[p][j]
      }
    }
    // ---- d enc: rows = feature (g&3) + 8(g>>2) + 4h, col = point
    if (MODE != 2 && valid) {
      const float inv = 1.f / scale;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int level = 4 * a + 2 * h;
        if (EH) {
          // packed halves, STILL scaled by the loss scale (tcnn hands dL/d enc to the grid backward in fp16 and
          // scaled; the consumer - csr_bwd_kernel - unscales in fp32 after the gather)
          const h2v lo = pk(denc[4 * a], denc[4 * a + 1]), hi = pk(denc[4 * a + 2], denc[4 * a + 3]);
          uint32_t* dw = reinterpret_cast<uint32_t*>(din);
          dw[p * ps + (int64_t)level * ls] = *reinterpret_cast<const uint32_t*>(&lo);
          dw[p * ps + (int64_t)(level + 1) * ls] = *reinterpret_cast<const uint32_t*>(&hi);
        } else {
          *reinterpret_cast<float2*>(din + p * ps + (int64_t)level * ls) = make_float2(denc[4 * a] * inv, denc[4 * a + 1] * inv);
          *reinterpret_cast<float2*>(din + p * ps + (int64_t)(level + 1) * ls) =
              make_float2(denc[4 * a + 2] * inv, denc[4 * a + 3] * inv);
        }
      }
    }
  }
  if (MODE == 1) return;
  // ---- flush the weight gradients once per WORKGROUP: the four waves park their tiles in their (now free) LDS
  // areas, every wave sums a quarter of the tile over the four copies and adds it with contiguous atomics
  const float inv = 1.f / scale;
  constexpr int TL = 33;                                   // floats per row of the flush tiles
  constexpr int WAVE_F = WAVE_BYTES / 4;
  float* ft_all = reinterpret_cast<float*>(wv_all);
  static_assert(32 * TL * 4 <= WAVE_BYTES, "flush tile does not fit the per-wave LDS area");
  float* ft = ft_all + wave * WAVE_F;
  // transpose: tile element (row, col) is parked at [col][row], so that the flat index runs over (col, row) -
  // (hidden, feature) for a dW1^T tile, the layout of dW1 itself
  auto flush_tile = [&](const f32x16& acc, float* dst, bool transpose, int n_valid) {
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      if (transpose) ft[r * TL + drow16(g, h)] = acc[g];
      else ft[drow16(g, h) * TL + r] = acc[g];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = (4 * wave + k) * 64 + lane;          // (idx >> 5, idx & 31) of the parked tile
      const int off = (idx >> 5) * TL + (idx & 31);
      const float v = ((ft_all[off] + ft_all[WAVE_F + off]) + (ft_all[2 * WAVE_F + off] + ft_all[3 * WAVE_F + off])) * inv;
      if (idx < n_valid) unsafeAtomicAdd(dst + idx, v);
    }
  };
#pragma unroll
  for (int jl = 0; jl < (MODE == 1 ? 1 : NJL); ++jl) flush_tile(dw1t[jl], dw1 + (size_t)(jt0 + jl) * 1024, true, 1024);
  // dW2 accumulator: row 2 jt + o, col = hidden r  ->  parked [row][col]; rows >= 2 NJT are zero.  dW2 is
  // [o][HID]: two passes (o = 0, 1) over a de-interleaved view would need another tile; the 2 NJT x 32 values are
  // few, so they go out with one atomic each from the flat view
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 16; ++g) ft[drow16(g, h) * TL + r] = dw2acc[g];
  __syncthreads();
  for (int idx0 = threadIdx.x; idx0 < 2 * NJL * 32; idx0 += 256) {
    const int idx = idx0 + 2 * jt0 * 32;   // rows 2 jt0 .. 2 (jt0 + NJL) - 1 belong to this workgroup's hidden tiles
    const int row = idx >> 5, col = idx & 31, off = row * TL + col;
    const float v = ((ft_all[off] + ft_all[WAVE_F + off]) + (ft_all[2 * WAVE_F + off] + ft_all[3 * WAVE_F + off])) * inv;
    unsafeAtomicAdd(dw2 + (row & 1) * HID + (row >> 1) * 32 + col, v);
  }
}

static size_t f16_fwd_smem(int hid) { return (size_t)(hid / 32) * 2 * 64 * 16 * 2; }
static size_t f16_bwd_smem(int hid) {
  return (size_t)(hid / 32) * 2 * 64 * 16 * 2 + (size_t)2 * hid * 4 + (size_t)4 * (3 * IMG_BYTES + 128);
}

int launch_mlp_fwd_f16(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                       const float* w1, const float* w2, float* out, hipStream_t st, bool enc_half) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(enc_half || ((ps % 2) == 0 && (ls % 2) == 0), "mlp input strides must be even");
  const int64_t n_tiles = cdiv(n, 32);
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 512);
  const size_t sm = f16_fwd_smem(cfg.n_hidden);
#define IMMOCO_FWD(H, A)                                                                             \
  do {                                                                                               \
    if (enc_half) mlp_fwd_f16_kernel<H, A, true><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, out, n_tiles); \
    else mlp_fwd_f16_kernel<H, A, false><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, out, n_tiles);         \
  } while (0)
  if (cfg.n_hidden == 64 && cfg.activation == IMMOCO_ACT_TANH) IMMOCO_FWD(64, IMMOCO_ACT_TANH);
  else if (cfg.n_hidden == 64) IMMOCO_FWD(64, IMMOCO_ACT_RELU);
  else if (cfg.activation == IMMOCO_ACT_TANH) IMMOCO_FWD(256, IMMOCO_ACT_TANH);
  else IMMOCO_FWD(256, IMMOCO_ACT_RELU);
#undef IMMOCO_FWD
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

template <int HID, int ACT, bool EH>
static int launch_bwd_f16_t(const float* in, int64_t ps, int64_t ls, int64_t n, const float* w1, const float* w2,
                            const float* dout, float* din, float* dw1, float* dw2, hipStream_t st,
                            int64_t dout_plane, float scale) {
  const int64_t n_tiles = cdiv(n, 32);
  const int blocks_per_cu = HID == 64 ? 2 : 1;
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 256 * blocks_per_cu);
  const size_t sm = f16_bwd_smem(HID);
  static bool attr_set = false;
  if (!attr_set) {
    IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_f16_kernel<HID, ACT, EH>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    attr_set = true;
  }
  mlp_bwd_f16_kernel<HID, ACT, EH><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, n_tiles,
                                                          dout_plane, scale);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// split backward of the 256-wide net: part 1 = d enc (din must not alias in), part 2 = dW1 / dW2
template <int ACT, bool EH, int MODE>
static int launch_bwd_f16_split_t(const float* in, int64_t ps, int64_t ls, int64_t n, const float* w1, const float* w2,
                                  const float* dout, float* din, float* dw1, float* dw2, hipStream_t st,
                                  int64_t dout_plane, float scale, const float* dout2) {
  constexpr int HID = 256, NJT = HID / 32;
  const int64_t n_tiles = cdiv(n, 32);
  // MODE 1: W1 and W1^T fragments + W2 (34 KB); MODE 2: its own tiles' W1 fragments + W2 + the per-wave images (34 KB)
  const size_t sm = MODE == 1 ? (size_t)NJT * 2 * 64 * 16 * 2 + (size_t)2 * HID * 4
                              : (size_t)F16_NJW * 2 * 64 * 16 + (size_t)2 * HID * 4 + (size_t)4 * (3 * IMG_BYTES + 128);
  static bool attr_set = false;
  if (!attr_set) {
    IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_f16_kernel<HID, ACT, EH, MODE>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    attr_set = true;
  }
  const dim3 grid((unsigned)std::min<int64_t>(cdiv(n_tiles, 4), MODE == 1 ? 512 : 256), MODE == 2 ? NJT / F16_NJW : 1);
  mlp_bwd_f16_kernel<HID, ACT, EH, MODE><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, n_tiles,
                                                                dout_plane, scale, dout2);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_mlp_bwd_f16_split(const immoco_mlp_cfg& cfg, int part, const float* in, int64_t ps, int64_t ls, int64_t n,
                             const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                             hipStream_t st, int64_t dout_plane, float scale, bool enc_half, const float* dout2) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(dout2 == nullptr || dout_plane != 0, "mlp_bwd_f16_split: a second dout addend needs the planar layout");
  IMMOCO_REQUIRE(cfg.n_hidden == 256 && (part == 1 || part == 2), "mlp_bwd_f16_split: 256-wide net, part 1 or 2");
  IMMOCO_REQUIRE(part == 2 || in != din, "mlp_bwd_f16_split: din must not alias in");
  IMMOCO_REQUIRE(enc_half || ((ps % 2) == 0 && (ls % 2) == 0), "mlp input strides must be even");
#define IMMOCO_SPLIT(A, M)                                                                                                 \
  return enc_half ? launch_bwd_f16_split_t<A, true, M>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane, scale, dout2)  \
                  : launch_bwd_f16_split_t<A, false, M>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane, scale, dout2)
  if (cfg.activation == IMMOCO_ACT_TANH) {
    if (part == 1) IMMOCO_SPLIT(IMMOCO_ACT_TANH, 1);
    IMMOCO_SPLIT(IMMOCO_ACT_TANH, 2);
  }
  if (part == 1) IMMOCO_SPLIT(IMMOCO_ACT_RELU, 1);
  IMMOCO_SPLIT(IMMOCO_ACT_RELU, 2);
#undef IMMOCO_SPLIT
}

int launch_mlp_bwd_f16(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                       const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                       hipStream_t st, int64_t dout_plane, float scale, bool enc_half) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(enc_half || ((ps % 2) == 0 && (ls % 2) == 0), "mlp input strides must be even");
  IMMOCO_REQUIRE(scale > 0.f, "mlp_bwd_half: loss scale must be positive");
#define IMMOCO_BWD(H, A)                                                                                          \
  return enc_half ? launch_bwd_f16_t<H, A, true>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane, scale) \
                  : launch_bwd_f16_t<H, A, false>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane, scale)
  if (cfg.n_hidden == 64 && cfg.activation == IMMOCO_ACT_TANH) IMMOCO_BWD(64, IMMOCO_ACT_TANH);
  if (cfg.n_hidden == 64) IMMOCO_BWD(64, IMMOCO_ACT_RELU);
  if (cfg.activation == IMMOCO_ACT_TANH) IMMOCO_BWD(256, IMMOCO_ACT_TANH);
  IMMOCO_BWD(256, IMMOCO_ACT_RELU);
#undef IMMOCO_BWD
}

}  // namespace immoco

using namespace immoco;

extern "C" int immoco_mlp_fwd_half(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                                   int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                                   float* out, void* stream) {
  int rc = check_mlp_cfg(cfg);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (in && w1 && w2 && out)), "mlp_fwd_half: NULL buffer");
  return launch_mlp_fwd_f16(*cfg, in, in_point_stride, in_level_stride, n, w1, w2, out, as_stream(stream), false);
}

extern "C" int immoco_mlp_bwd_half(const immoco_mlp_cfg* cfg, const float* in, int64_t in_point_stride,
                                   int64_t in_level_stride, int64_t n, const float* w1, const float* w2,
                                   const float* dout, float loss_scale, float* din, float* dw1, float* dw2,
                                   void* stream) {
  int rc = check_mlp_cfg(cfg);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (in && w1 && w2 && dout && din && dw1 && dw2)), "mlp_bwd_half: NULL buffer");
  return launch_mlp_bwd_f16(*cfg, in, in_point_stride, in_level_stride, n, w1, w2, dout, din, dw1, dw2,
                            as_stream(stream), 0, loss_scale, false);
}
