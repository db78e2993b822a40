// Matrix-core implementation of the bias-free MLP 32 -> HID -> 2 (forward and backward) with
// gfx950's f32-input MFMA, v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain, no
// conversion of the operands) at the fp32 vector rate.
//
// Why MFMA although its fp32 rate equals the VALU's: every product of this MLP is a small dense
// GEMM (K = 32 features, 32 hidden units or 32 points), and three of them reduce over the lane
// dimension of a lane-per-point layout (dW1, dW2 over points; d enc over hidden units).  The fp32
// VALU kernels (mlp.hip) did those reductions through LDS broadcasts: 1.15 + 0.53 ms per iteration
// (rocprof, MI355X).  A first matrix-core version split fp32 operands into three bf16 terms (six
// v_mfma_f32_32x32x16_bf16 per product, fp32-equivalent): 0.245 + 0.141 ms backward, 0.082 + 0.024
// forward, VALU-issue bound (rocprof: 2400 VALU instructions per 32-point tile, a third of them
// operand splitting).  This f32-MFMA version needs no splitting and is bit-exact fp32:
// 0.237 + 0.143 ms backward, 0.101 + 0.036 ms forward - the same speed within 5 %, so the exact
// one is kept.  Both are latency/issue bound at 1-2 waves per SIMD, far from the MFMA pipe.
//
// One wave owns a tile of 32 points.  Fragment maps of v_mfma_f32_32x32x2_f32 (cdna guide §3),
// lane l, r = l & 31, h = l >> 5:
//   A: one float A[row r][k = h]      B: one float B[k = h][col r]
//   D: reg g holds D[row (g&3) + 8(g>>2) + 4h][col r]
// A result tile X (rows in registers, column on the lane) is the B operand of a following product
// that sums over X's rows WITHOUT any data movement: k-step g takes register g, i.e. the row pair
// (drow(g,0), drow(g,1)); the A operand is simply loaded in that k order.
//   layout 1  pre [hidden][point] = W1 . enc^T          -> d enc^T = W1^T . dpre      (sum over hidden)
//   layout 2  h'  [point][hidden] = transpose(h) via LDS -> dW1^T  = enc^T . dpre'    (sum over points)
//                                                          dW2^T  = dout^T . h'      (sum over points)
// Weight fragments live in LDS (built once per workgroup); weight gradients stay in accumulator
// registers across a wave's tiles and are flushed once with 256-byte contiguous atomics.
#include "kernels.hpp"

namespace immoco {

typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ f32x16 mfma32(float a, float b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// row index of D register g for lane half h
__device__ __forceinline__ int drow(int g, int h) { return (g & 3) + 8 * (g >> 2) + 4 * h; }

// tanh in ten instructions: 1 - 2/(exp(2|x|) + 1) on v_exp_f32 / v_rcp_f32 (absolute error ~1e-7,
// i.e. relative <= 3e-6 for |x| >= 0.04) and x - x^3/3 below 0.04 (relative error < 4e-7).
// Keeping the RELATIVE accuracy for tiny arguments matters: the motion field starts at ~1e-3 and
// an absolute-only tanh (six instructions) shifted the loss of iteration 5 by 1.3e-3 against the
// oracle (2e-4 with this form); a 17-instruction series form was 40 % of the forward VALU work.
//
// Round 4, built and NOT shipped (-DIMMOCO_DIAG_POLY_TANH; `make diag EXTRA=-DIMMOCO_DIAG_POLY_TANH`): the exp form loses
// relative accuracy below ~0.3 (1 - 2/(e+1) cancels: rms 8e-7 in [0.04, 0.1), 2.6e-7 in [0.1, 0.3) - where the motion
// net's hidden units live), and the weight gradients of a nearly converged motion net are residuals of cancelling sums over
// 1 M points: on some late states that 1e-6 comes out as 1.5e-4 ... 4.6e-4 (rel. L2) in the motion gradient against the
// oracle where the VALU kernels with libm tanhf have 3e-6 (tools/diag_tf_slice.py).  The variant below - x (1 + u (c0 + c1 u
// + c2 u^2 + c3 u^3)), u = x^2, for |x| < 0.5, a Chebyshev-node least-squares fit, relative error <= 1e-7 (rms 3e-8, the
// rounding floor); thirteen instructions, the same 0.2784 slices/s - makes HIP's single steps coincide with the DEVICE
// oracle's: both then differ from the CPU oracle by the same 3.59e-5 / 3.63e-5 (K = 5, 320x320) and 1.350e-5 / 1.354e-5
// (K = 60, 96x96) in the motion gradient, and 78 of 9.45 M Adam updates move by more than 1e-3 lr against the CPU oracle
// (gpurun_out -> DESIGN.md 2.5).  Every ensemble of DESIGN.md 2.4 was drawn with the ten-instruction form, so that is what
// ships; switching needs the cells redrawn.
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = fabsf(x);
  const float e = __builtin_amdgcn_exp2f(ax * 2.885390082f);  // exp(2|x|) = 2^(2|x| log2 e)
  const float big = fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
#ifdef IMMOCO_DIAG_POLY_TANH
  const float u = ax * ax;
  const float p = fmaf(fmaf(fmaf(0.017999219f, u, -0.053391054f), u, 0.13330513f), u, -0.3333331f);
  const float small = fmaf(ax * u, p, ax);
  return copysignf(ax < 0.5f ? small : big, x);
#else
  const float small = ax * fmaf(ax * ax, -0.33333334f, 1.f);
  return copysignf(ax < 0.04f ? small : big, x);
#endif
}

template <int ACT>
__device__ __forceinline__ float act_f(float pre) {
#ifdef IMMOCO_DIAG_LIBM_TANH   // diagnostics build only (tools/README.md): libm tanhf instead of tanh_fast
  return ACT == IMMOCO_ACT_RELU ? fmaxf(pre, 0.f) : tanhf(pre);
#else
  return ACT == IMMOCO_ACT_RELU ? fmaxf(pre, 0.f) : tanh_fast(pre);
#endif
}
// derivative from the activation VALUE (relu: h > 0 <=> pre > 0; tanh: 1 - h^2)
template <int ACT>
__device__ __forceinline__ float act_d(float hv) {
  return ACT == IMMOCO_ACT_RELU ? (hv > 0.f ? 1.f : 0.f) : 1.f - hv * hv;
}

constexpr int TLD = 36;  // row length of the per-wave 32x32 transpose tiles (conflict-free b32 writes / b128 reads)

// B fragments of the point tile: eb[s] = in[point p][feature 2s + h] (0 beyond n).  Unconditional
// loads on a clamped address (a load under a divergent `if` gets its own basic block and wait).
__device__ __forceinline__ void load_enc_b(const float* in, int64_t ps, int64_t ls, int64_t p, int64_t n, int h,
                                           float (&eb)[16]) {
  const int64_t pc = p < n ? p : n - 1;
  const float m = p < n ? 1.f : 0.f;
#pragma unroll
  for (int s = 0; s < 16; ++s) eb[s] = in[pc * ps + (int64_t)s * ls + h] * m;
}

// LDS weight fragments: AW[jt][s4][lane] float4 = W1[jt*32 + r][2*(4*s4+i) + h], i = 0..3
//                       AWT[jt][g4][lane] float4 = W1[jt*32 + drow(4*g4+i, h)][r]
// W1 is read with coalesced 16-byte loads and every element is SCATTERED to its fragment positions in LDS (the
// gather formulation - fragment entry by fragment entry, 64 different cache lines per load instruction - cost the
// wide backward ~13 us per workgroup, a tenth of its run time).  Element W1[j][k], j = jt*32 + rho:
//   AW : entry (jt*4 + s4)*64 + (h*32 + rho),  component i,  with k = 2*(4*s4 + i) + h
//   AWT: entry (jt*4 + g4)*64 + (h'*32 + k),   component i', with rho = drow(4*g4 + i', h'), i.e. h' = (rho>>2)&1,
//        g4 = rho>>3, i' = rho&3
template <int HID>
__device__ __forceinline__ void build_weight_frags(const float* __restrict__ w1, float4* aw, float4* awt, int tid) {
  float* awf = reinterpret_cast<float*>(aw);
  float* awtf = reinterpret_cast<float*>(awt);
  for (int e4 = tid; e4 < HID * 32 / 4; e4 += 256) {
    const float4 q = reinterpret_cast<const float4*>(w1)[e4];   // W1[j][k0 .. k0+3]
    const float v[4] = {q.x, q.y, q.z, q.w};
    const int j = e4 >> 3, k0 = (e4 & 7) * 4;
    const int jt = j >> 5, rho = j & 31;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = k0 + c, sidx = k >> 1, h = k & 1;
      awf[(((jt * 4 + (sidx >> 2)) * 64) + h * 32 + rho) * 4 + (sidx & 3)] = v[c];
      if (awt) awtf[(((jt * 4 + (rho >> 3)) * 64) + ((rho >> 2) & 1) * 32 + k) * 4 + (rho & 3)] = v[c];
    }
  }
}

// pre[hidden][point] tile of one jt
__device__ __forceinline__ f32x16 pre_tile(const float4* aw, int jt, int lane, const float (&eb)[16]) {
  f32x16 pre = {0.f};
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    const float4 a = aw[(jt * 4 + s4) * 64 + lane];
    pre = mfma32(a.x, eb[4 * s4], pre);
    pre = mfma32(a.y, eb[4 * s4 + 1], pre);
    pre = mfma32(a.z, eb[4 * s4 + 2], pre);
    pre = mfma32(a.w, eb[4 * s4 + 3], pre);
  }
  return pre;
}

// ---------------------------------------------------------------------------------------------
// forward: out[p][0..1] = W2 . act(W1 . enc[p])
// (Forcing 4 waves/SIMD with a register cap spills and is slower: 0.133 vs 0.101 ms; getting there without
// spills - one hidden tile live at a time, 90 registers, 1024 workgroups - changes nothing: 0.078 vs 0.080 ms.)
template <int HID, int ACT>
__global__ __launch_bounds__(256) void mlp_fwd_mfma_kernel(const float* __restrict__ in, int64_t ps, int64_t ls,
                                                           int64_t n, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, float* __restrict__ out,
                                                           int64_t n_tiles) {
  constexpr int NJT = HID / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* aw = reinterpret_cast<float4*>(smem);                 // [NJT][4][64]
  float* w2s = reinterpret_cast<float*>(aw + NJT * 4 * 64);     // [2][HID]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  build_weight_frags<HID>(w1, aw, nullptr, threadIdx.x);
  for (int i = threadIdx.x; i < 2 * HID; i += 256) w2s[i] = w2[i];
  __syncthreads();

  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  float nx[16];  // software pipeline: the next tile's operands are in flight while this one is computed
  if (wave_id < n_tiles) load_enc_b(in, ps, ls, wave_id * 32 + r, n, h, nx);
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const int64_t p = t * 32 + r;
    float eb[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) eb[s] = nx[s];
    if (t + n_waves < n_tiles) load_enc_b(in, ps, ls, (t + n_waves) * 32 + r, n, h, nx);
    float o0 = 0.f, o1 = 0.f;
#pragma unroll 2
    for (int jt = 0; jt < NJT; ++jt) {
      const f32x16 pre = pre_tile(aw, jt, lane, eb);
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
    if (p < n && h == 0) *reinterpret_cast<float2*>(out + p * 2) = make_float2(o0, o1);
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// Measured dead end (round 2): the wide net (HID = 256: 448 registers, 105 KB of LDS, 1 wave/SIMD - a workgroup
// excludes every other kernel from its CU, and beside the motion grid's encode backward it takes 0.52 ms instead of
// 0.14) split into two hidden-layer halves per point tile (grid.y = 2, 248 registers, 71 KB, the halves ADD their
// d enc into a zeroed buffer): 0.155 ms alone instead of 0.148, still 0.47 ms beside the encode backward (whose four
// 33 KB workgroups per CU leave no 71 KB hole), and the 13 MB memset costs the image chain another 0.06 ms:
// graph iteration 1.353 instead of 1.346 ms.  Not kept.
// RAW (diagnostics build only, IMMOCO_MLP_RAWLOAD=1): the next tile's loads carry no arithmetic (see load_raw)
template <int HID, int ACT, bool RAW = false>
__global__ __launch_bounds__(256, HID == 64 ? 2 : 1) void mlp_bwd_mfma_kernel(const float* in /* may alias din */, int64_t ps, int64_t ls,
                                                           int64_t n, const float* __restrict__ w1,
                                                           const float* __restrict__ w2,
                                                           const float* __restrict__ dout, float* din,
                                                           float* __restrict__ dw1, float* __restrict__ dw2,
                                                           int64_t n_tiles, int64_t dout_plane) {
  constexpr int NJT = HID / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* aw = reinterpret_cast<float4*>(smem);                  // W1 fragments    [NJT][4][64]
  float4* awt = aw + NJT * 4 * 64;                               // W1^T fragments  [NJT][4][64]
  float* w2s = reinterpret_cast<float*>(awt + NJT * 4 * 64);     // [2][HID]
  float* dos_all = w2s + 2 * HID;                                // per wave [32][2] dout staging
  float* tr_all = dos_all + 4 * 64;                              // per wave two [32][TLD] transpose tiles
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  float* dos = dos_all + wave * 64;
  float* tr = tr_all + wave * 2 * 32 * TLD;   // h tile (and the final dW1 flush)
  float* te = tr + 32 * TLD;                  // enc tile
  build_weight_frags<HID>(w1, aw, awt, threadIdx.x);
  for (int i = threadIdx.x; i < 2 * HID; i += 256) w2s[i] = w2[i];
  __syncthreads();

  // dW2: the narrow net (HID = 64, the motion INR on the critical path) keeps per-LANE partial sums
  // dw2l[jt][g][o] += h[hidden drow(g,h)][point r] * dout[point r][o] over all of the wave's tiles and
  // reduces them over the lanes once at the end (32 FMAs per tile instead of a 16-MFMA chain whose
  // 32x32 result has two useful rows); the wide net has no registers for that (NJT*32 of them) and
  // multiplies dout^T . h' on the matrix core.
  constexpr bool LANE_DW2 = HID == 64;
  f32x16 dw1t[NJT];  // dW1^T tiles: rows k (features), col = hidden jt*32 + r
  float dw2a[NJT][2];
  float dw2l[LANE_DW2 ? NJT : 1][16][2];
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) {
    dw1t[jt] = (f32x16){0.f};
    dw2a[jt][0] = dw2a[jt][1] = 0.f;
  }
#pragma unroll
  for (int jt = 0; jt < (LANE_DW2 ? NJT : 1); ++jt)
#pragma unroll
    for (int g = 0; g < 16; ++g) dw2l[jt][g][0] = dw2l[jt][g][1] = 0.f;

  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  float nx[16];
  float2 nx_d = make_float2(0.f, 0.f);
  // (load_enc_b multiplies every value by the tail mask, so the compiler waits for these "prefetch" loads on the spot -
  // s_waitcnt vmcnt(16) ... vmcnt(0) directly behind them in the ISA.  RAW: no arithmetic here; only dout needs the mask -
  // a point beyond n then has dpre = 0 and its clamped, finite encoding contributes nothing - applied one tile later.)
  float mnx = 1.f;
  auto load_raw = [&](int64_t tt) {
    const int64_t q = tt * 32 + r;
    const int64_t qc = q < n ? q : n - 1;
    const float mq = q < n ? 1.f : 0.f;
    if (RAW) {
      mnx = mq;
#pragma unroll
      for (int s = 0; s < 16; ++s) nx[s] = in[qc * ps + (int64_t)s * ls + h];
      if (dout_plane) nx_d = make_float2(dout[qc], dout[dout_plane + qc]);
      else nx_d = *reinterpret_cast<const float2*>(dout + qc * 2);
      return;
    }
    load_enc_b(in, ps, ls, q, n, h, nx);
    if (dout_plane) {  // wave-uniform
      nx_d = make_float2(dout[qc] * mq, dout[dout_plane + qc] * mq);
    } else {
      const float2 dv = *reinterpret_cast<const float2*>(dout + qc * 2);
      nx_d = make_float2(dv.x * mq, dv.y * mq);
    }
  };
  if (wave_id < n_tiles) load_raw(wave_id);
#ifdef IMMOCO_DIAG
  // RAW also staggers the two workgroups of a CU by about half a tile (the second half of the grid sleeps 8128 cycles once):
  // SIMD partners that run the same program fall into lockstep - both in their matrix phase, both in their memory phase
  // (MI355X_MICROARCH.md, "two waves that run the same program") - which is what the ablations of this kernel look like
  if (RAW && blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(127);
#endif
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const int64_t p = t * 32 + r;
    const bool valid = p < n;
    float eb[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) eb[s] = nx[s];
    const float2 d = RAW ? make_float2(nx_d.x * mnx, nx_d.y * mnx) : nx_d;
    if (t + n_waves < n_tiles) load_raw(t + n_waves);
    // ---- stage dout and the enc tile (rows = feature k = 2s + h, cols = point) in LDS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // earlier readers (previous tile) are done
    if (h == 0) *reinterpret_cast<float2*>(dos + 2 * r) = d;
#pragma unroll
    for (int s = 0; s < 16; ++s) te[(2 * s + h) * TLD + r] = eb[s];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // A fragments in accumulator-k order: et[g] = enc[point drow(g,h)][feature r], dt[g] = dout[point drow(g,h)][r]
    float et[16];
    if (!LANE_DW2) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 q4 = *reinterpret_cast<const float4*>(te + r * TLD + 8 * a + 4 * h);
        et[4 * a] = q4.x;
        et[4 * a + 1] = q4.y;
        et[4 * a + 2] = q4.z;
        et[4 * a + 3] = q4.w;
      }
    }
    f32x16 denc = {0.f};
#pragma unroll
    for (int jt = 0; jt < NJT; ++jt) {
      // ---- layout 1: rows = hidden, col = point
      const f32x16 pre = pre_tile(aw, jt, lane, eb);
      float dp[16], hv[16];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 wa = *reinterpret_cast<const float4*>(w2s + jt * 32 + 8 * a + 4 * h);
        const float4 wb = *reinterpret_cast<const float4*>(w2s + HID + jt * 32 + 8 * a + 4 * h);
        const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const float hh = act_f<ACT>(pre[4 * a + b]);
          hv[4 * a + b] = hh;
          dp[4 * a + b] = fmaf(was[b], d.x, wbs[b] * d.y) * act_d<ACT>(hh);
          if (LANE_DW2) {
            dw2l[LANE_DW2 ? jt : 0][4 * a + b][0] = fmaf(hh, d.x, dw2l[LANE_DW2 ? jt : 0][4 * a + b][0]);
            dw2l[LANE_DW2 ? jt : 0][4 * a + b][1] = fmaf(hh, d.y, dw2l[LANE_DW2 ? jt : 0][4 * a + b][1]);
          }
        }
      }
      // d enc^T[k][p] += sum_j W1[j][k] dpre[j][p]
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 a = awt[(jt * 4 + g4) * 64 + lane];
        denc = mfma32(a.x, dp[4 * g4], denc);
        denc = mfma32(a.y, dp[4 * g4 + 1], denc);
        denc = mfma32(a.z, dp[4 * g4 + 2], denc);
        denc = mfma32(a.w, dp[4 * g4 + 3], denc);
      }
      // ---- layout 2: transpose through the per-wave LDS tile
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous readers of the tile are done
      if (LANE_DW2) {
        // dpre' = transpose(dpre) is all that is left to do in this layout
#pragma unroll
        for (int g = 0; g < 16; ++g) tr[drow(g, h) * TLD + r] = dp[g];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // dW1^T[k][j] += sum_p enc[p][k] dpre'[p][j]; both fragments come straight from the LDS tiles
        // (no 16-register copies: the kernel sits at the 256-register budget of 2 waves/SIMD)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float4 qd = *reinterpret_cast<const float4*>(tr + r * TLD + 8 * a + 4 * h);
          const float4 qe = *reinterpret_cast<const float4*>(te + r * TLD + 8 * a + 4 * h);
          dw1t[jt] = mfma32(qe.x, qd.x, dw1t[jt]);
          dw1t[jt] = mfma32(qe.y, qd.y, dw1t[jt]);
          dw1t[jt] = mfma32(qe.z, qd.z, dw1t[jt]);
          dw1t[jt] = mfma32(qe.w, qd.w, dw1t[jt]);
        }
      } else {
        // h' = transpose(h); dpre' is recomputed from it
#pragma unroll
        for (int g = 0; g < 16; ++g) tr[drow(g, h) * TLD + r] = hv[g];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float w20 = w2s[jt * 32 + r], w21 = w2s[HID + jt * 32 + r];
        float hp[16];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const float4 q4 = *reinterpret_cast<const float4*>(tr + r * TLD + 8 * a + 4 * h);
          hp[4 * a] = q4.x;
          hp[4 * a + 1] = q4.y;
          hp[4 * a + 2] = q4.z;
          hp[4 * a + 3] = q4.w;
        }
        f32x16 tmp = {0.f};
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          // dout of row point drow(g,h): re-read from LDS (broadcast) instead of holding 48 registers
          const float2 dq = *reinterpret_cast<const float2*>(dos + 2 * drow(g, h));
          const float dpt = fmaf(w20, dq.x, w21 * dq.y) * act_d<ACT>(hp[g]);
          const float dtg = r == 0 ? dq.x : (r == 1 ? dq.y : 0.f);
          dw1t[jt] = mfma32(et[g], dpt, dw1t[jt]);  // dW1^T[k][j] += sum_p enc[p][k] dpre'[p][j]
          tmp = mfma32(dtg, hp[g], tmp);            // dW2^T[o][j] += sum_p dout[p][o] h'[p][j]
        }
        dw2a[jt][0] += tmp[0];  // row o = 0 (lanes h = 0)
        dw2a[jt][1] += tmp[1];  // row o = 1
      }
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
  const bool has_tile = wave_id < n_tiles;  // a wave without tiles holds zeros and only helps with the reduction
  // ---- flush the weight gradients (once per WORKGROUP): the four waves transpose their dW1^T tiles into their LDS
  // regions, every wave sums a quarter of the 32x32 tile over the four regions and adds it with 256-byte contiguous
  // atomics - a quarter of the atomics of a per-wave flush (the wide net's 1024 waves sent 33 MB of them per launch
  // at a 32 KB target: rocprofv3 WRITE_SIZE 46 MB for 13 MB of d enc)
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) {
    __syncthreads();   // the tiles are free (main loop / previous round)
#pragma unroll
    for (int g = 0; g < 16; ++g) tr[r * TLD + drow(g, h)] = dw1t[jt][g];  // [hidden r][feature k]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = (4 * wave + k) * 64 + lane;  // (hidden = idx>>5, feature = idx&31) of this 32x32 tile
      const int off = (idx >> 5) * TLD + (idx & 31);
      const float v = (tr_all[off] + tr_all[2 * 32 * TLD + off]) + (tr_all[4 * 32 * TLD + off] + tr_all[6 * 32 * TLD + off]);
      unsafeAtomicAdd(dw1 + (size_t)jt * 1024 + idx, v);
    }
    if (!has_tile) continue;
    if (LANE_DW2) {
      // sum the per-lane partials over the 32 points of each lane half, stage the 2 x 32 sums in LDS and
      // flush them with two 128-byte atomics (one atomic per (hidden, output) pair - 64 two-lane
      // instructions per wave on the same four cache lines - cost 0.3 ms of serialised L2 atomics)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        float v0 = dw2l[LANE_DW2 ? jt : 0][g][0], v1 = dw2l[LANE_DW2 ? jt : 0][g][1];
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) {
          v0 += __shfl_xor(v0, m, 64);
          v1 += __shfl_xor(v1, m, 64);
        }
        if (r == 0) {
          dos[drow(g, h)] = v0;
          dos[32 + drow(g, h)] = v1;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (h == 0) {
        unsafeAtomicAdd(dw2 + jt * 32 + r, dos[r]);
        unsafeAtomicAdd(dw2 + HID + jt * 32 + r, dos[32 + r]);
      }
    } else if (h == 0) {
      unsafeAtomicAdd(dw2 + jt * 32 + r, dw2a[jt][0]);
      unsafeAtomicAdd(dw2 + HID + jt * 32 + r, dw2a[jt][1]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// EXPERIMENT (diagnostics library only: IMMOCO_MLP_BWD64=pipe | pipe_partials | pipe_<ablation>; written and measured with the
// last GPU minutes of round 4, profiles/r04_narrow_mlp_bwd_ablations.txt; the shipped library does not contain it): the narrow
// backward with its matrix-core chains and its VALU work INTERLEAVED inside the wave.
// Why it was tried: per 32-point tile the kernel above issues 96 v_mfma_f32_32x32x2_f32 (6144 cycles of the SIMD's matrix pipe:
// 0.080 ms for 32 000 tiles on 1024 SIMDs at 2.4 GHz) and ~570 VALU instructions incl. 64 quarter-rate ones, and takes 0.176 ms;
// its ISA is strictly phase after phase - 16 dependent MFMAs, `s_nop 14`, the tanh block, 16 MFMAs, the transpose, 16 MFMAs.
// Here the two hidden tiles are software-pipelined: while the matrix pipe runs a chain of one tile the VALU does the activation
// work of the other,
//     pre0 | pre1 + V(pre0)a | d enc(dp0) + V(pre0)b | dW1_0 + V(pre1)a | d enc(dp1) + V(pre1)b | dW1_1 + next loads, stores
// (V(.)a / b = the two halves of a hidden tile's activation work) with `sched_group_barrier` pinning "1 MFMA, then 7 VALU" in the
// four mixed stages (`hipcc -S`: 242 registers, no scratch, every MFMA of those stages followed by ~7 VALU instructions).
// Arithmetic, operand order and the order of every accumulation are those of mlp_bwd_mfma_kernel<64, ACT>.
// MEASURED: correct (3e-7 against float64) and NOT faster - 0.197 against 0.191 ms in the same harness.  The kernel is not
// issue-bound: with ReLU instead of tanh the shipped kernel takes the same time; without any MFMA this one takes 0.100 ms,
// without MFMA and activation work 0.097, i.e. the kernel is the SUM of its matrix-core time (0.096 ms, the full rate) and
// of a memory / LDS skeleton that is 2.3x slower than a plain copy of the same bytes, and the two do not overlap.  The
// partial-sum flush (ABL = 12 below: no atomics, deterministic, more accurate) saves 7 us; raw prefetch loads 3 %; ONE wave per
// SIMD (IMMOCO_MLP_GRID=256) is 4 % slower than two.  Reading: the clock the chip holds under fp32 matrix load, not a pipe.
#ifdef IMMOCO_DIAG   // compiled into libimmoco_hip_diag.so only
#define IMMOCO_CB() asm volatile("" ::: "memory")
// writes that OTHER lanes of the wave read next: a real wait (3 per tile, ~100 cycles each), as in the kernel above
#define IMMOCO_LDS_DONE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// ABL (timing-only ablations, wrong results): 1 = no dW1 chains, 2 = no d enc chains (a third of the MFMAs each),
// 3 = no weight-gradient flush at the end, 4 = no MFMA at all (loads, LDS traffic, VALU work and stores stay),
// 5 / 6 = the flush spread over 4 / 16 copies of dW1 / dW2 (2048 / 1024 floats apart: the CALLER must have allocated them -
// tools/check_pipe_bwd.py does; timing of the atomics' contention only), 7 = prologue and flush only (no tile loop),
// 8 = as 4 and no activation work either (loads, LDS staging / transposes, stores, flush), 9 = 7 without the atomics,
// 10 = 7 without the weight-fragment build, 11 = empty kernel (harness offset).
// ABL = 12 is NOT an ablation: the weight gradients leave the workgroup as plain stores of per-workgroup (dW1) / per-wave (dW2)
// partial sums into a scratch buffer (dw1 / dw2 point there), and mlp_dw_reduce_kernel sums them in a fixed order - no global
// atomics, and a deterministic result.
template <int ACT, int ABL = 0>
__global__ __launch_bounds__(256, 2) void mlp_bwd64_pipe_kernel(const float* in /* may alias din */, int64_t ps, int64_t ls,
                                                                int64_t n, const float* __restrict__ w1,
                                                                const float* __restrict__ w2,
                                                                const float* __restrict__ dout, float* din,
                                                                float* __restrict__ dw1, float* __restrict__ dw2,
                                                                int64_t n_tiles, int64_t dout_plane) {
  constexpr int HID = 64, NJT = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* aw = reinterpret_cast<float4*>(smem);                  // W1 fragments    [NJT][4][64]
  float4* awt = aw + NJT * 4 * 64;                               // W1^T fragments  [NJT][4][64]
  float* w2s = reinterpret_cast<float*>(awt + NJT * 4 * 64);     // [2][HID]
  float* dos_all = w2s + 2 * HID;                                // per wave 64 floats (epilogue scratch)
  float* tr_all = dos_all + 4 * 64;                              // per wave three [32][TLD] tiles: dp0', dp1', enc
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  float* dos = dos_all + wave * 64;
  float* tr0 = tr_all + wave * 3 * 32 * TLD;
  float* tr1 = tr0 + 32 * TLD;
  float* te = tr1 + 32 * TLD;
  if (ABL == 11) return;
  if (ABL != 10) build_weight_frags<HID>(w1, aw, awt, threadIdx.x);
  for (int i = threadIdx.x; i < 2 * HID; i += 256) w2s[i] = w2[i];
  __syncthreads();

  f32x16 dw1t0 = {0.f}, dw1t1 = {0.f};
  float dw2l[NJT][16][2];
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt)
#pragma unroll
    for (int g = 0; g < 16; ++g) dw2l[jt][g][0] = dw2l[jt][g][1] = 0.f;

  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  float eb[16];
  float2 dnx = make_float2(0.f, 0.f);
  // RAW loads: no arithmetic on the loaded values here.  load_enc_b multiplies every value by the tail mask right after its
  // load, which makes the wave wait for the "prefetch" on the spot (the ISA of the kernel above shows s_waitcnt vmcnt(16) ...
  // vmcnt(0) directly behind the 18 loads: the full memory latency of every tile is exposed).  Only dout needs the mask (a
  // point beyond n then has dpre = 0, so its clamped - finite - encoding contributes nothing), and it is applied one tile
  // later, where the values are needed anyway.
  float mnx = 0.f;
  auto load_tile = [&](int64_t tt) {   // straight into eb (dead after the two pre chains of the current tile)
    const int64_t q = tt * 32 + r;
    const int64_t qc = q < n ? q : n - 1;
    mnx = q < n ? 1.f : 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) eb[s] = in[qc * ps + (int64_t)s * ls + h];
    if (dout_plane) {  // wave-uniform
      dnx = make_float2(dout[qc], dout[dout_plane + qc]);
    } else {
      dnx = *reinterpret_cast<const float2*>(dout + qc * 2);
    }
  };
  // activation work of one hidden tile: h = act(pre), dpre = (W2^T dout) act'(h), per-lane dW2 partial sums
  auto valu = [&](const f32x16& pre, const int jt, const float2 d, float (&dp)[16], const int a0, const int a1) {
#pragma unroll
    for (int a = a0; a < a1; ++a) {
      const float4 wa = *reinterpret_cast<const float4*>(w2s + jt * 32 + 8 * a + 4 * h);
      const float4 wb = *reinterpret_cast<const float4*>(w2s + HID + jt * 32 + 8 * a + 4 * h);
      const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (ABL == 8) {   // timing ablation: no activation work
          dp[4 * a + b] = pre[4 * a + b] + was[b];
          dw2l[jt][4 * a + b][0] += wbs[b];
          continue;
        }
        const float hh = act_f<ACT>(pre[4 * a + b]);
        dp[4 * a + b] = fmaf(was[b], d.x, wbs[b] * d.y) * act_d<ACT>(hh);
        dw2l[jt][4 * a + b][0] = fmaf(hh, d.x, dw2l[jt][4 * a + b][0]);
        dw2l[jt][4 * a + b][1] = fmaf(hh, d.y, dw2l[jt][4 * a + b][1]);
      }
    }
  };
  auto denc_chain = [&](const int jt, const float (&dp)[16], f32x16& denc) {   // d enc^T[k][p] += sum_j W1[j][k] dpre[j][p]
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 a = awt[(jt * 4 + g4) * 64 + lane];
      denc = mfma32(a.x, dp[4 * g4], denc);
      denc = mfma32(a.y, dp[4 * g4 + 1], denc);
      denc = mfma32(a.z, dp[4 * g4 + 2], denc);
      denc = mfma32(a.w, dp[4 * g4 + 3], denc);
    }
  };
  auto dw1_chain = [&](const float* tr, f32x16& acc) {   // dW1^T[k][j] += sum_p enc[p][k] dpre'[p][j]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 qd = *reinterpret_cast<const float4*>(tr + r * TLD + 8 * a + 4 * h);
      const float4 qe = *reinterpret_cast<const float4*>(te + r * TLD + 8 * a + 4 * h);
      acc = mfma32(qe.x, qd.x, acc);
      acc = mfma32(qe.y, qd.y, acc);
      acc = mfma32(qe.z, qd.z, acc);
      acc = mfma32(qe.w, qd.w, acc);
    }
  };

  constexpr bool NO_LOOP = ABL == 7 || ABL == 9 || ABL == 10;
  if (!NO_LOOP && wave_id < n_tiles) load_tile(wave_id);
  for (int64_t t = wave_id; t < (NO_LOOP ? 0 : n_tiles); t += n_waves) {
    const int64_t p = t * 32 + r;
    const bool valid = p < n;
    const float2 d = make_float2(dnx.x * mnx, dnx.y * mnx);
    // ---- stage the enc tile (rows = feature k = 2s + h, cols = point); earlier readers of `te` (the previous tile's dW1
    // chains) precede these writes in the wave's in-order LDS queue
    IMMOCO_CB();
#pragma unroll
    for (int s = 0; s < 16; ++s) te[(2 * s + h) * TLD + r] = eb[s];
    IMMOCO_LDS_DONE();
    // Six stages of 16 MFMAs; sched_barrier(0) = nothing moves across, so every stage is its own scheduling region, and in
    // the four mixed ones sched_group_barrier pins "1 MFMA, then 13 VALU instructions" (half a hidden tile's activation work
    // is ~215 instructions incl. 16 quarter-rate ones: ~76 cycles of VALU per 64-cycle MFMA).
#define IMMOCO_MIX()                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {   \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  \
    __builtin_amdgcn_sched_group_barrier(0x002, 7, 0);  \
  }                                                     \
  __builtin_amdgcn_sched_barrier(0)
    // ---- 1: pre0
    f32x16 pre0, pre1;
    if (ABL == 4 || ABL == 8) {
#pragma unroll
      for (int g = 0; g < 16; ++g) { pre0[g] = eb[g]; pre1[g] = eb[15 - g]; }
    } else {
      pre0 = pre_tile(aw, 0, lane, eb);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2: pre1 | first half of the activation work of pre0
    if (ABL != 4 && ABL != 8) pre1 = pre_tile(aw, 1, lane, eb);
    float dp0[16], dp1[16];
    valu(pre0, 0, d, dp0, 0, 2);
    IMMOCO_MIX();
    if (ABL == 12 && t + n_waves < n_tiles) load_tile(t + n_waves);   // eb is dead from here: four stages of latency hiding
    __builtin_amdgcn_sched_barrier(0);
    // ---- 3: d enc chain of hidden tile 0 | second half of pre0's activation work (k-steps 8..15 take dp0[8..15] just in time)
    f32x16 denc = {0.f};
    valu(pre0, 0, d, dp0, 2, 4);
    if (ABL != 2 && ABL != 4 && ABL != 8) denc_chain(0, dp0, denc);
    IMMOCO_MIX();
    // ---- 4: dW1 chain of hidden tile 0 (dpre' through the wave's LDS tile) | first half of pre1's activation work
#pragma unroll
    for (int g = 0; g < 16; ++g) tr0[drow(g, h) * TLD + r] = dp0[g];
    IMMOCO_LDS_DONE();
    if (ABL != 1 && ABL != 4 && ABL != 8) dw1_chain(tr0, dw1t0);
    valu(pre1, 1, d, dp1, 0, 2);
    IMMOCO_MIX();
    // ---- 5: d enc chain of hidden tile 1 | second half of pre1's activation work
    valu(pre1, 1, d, dp1, 2, 4);
    if (ABL != 2 && ABL != 4 && ABL != 8) denc_chain(1, dp1, denc);
    IMMOCO_MIX();
    // ---- 6: dW1 chain of hidden tile 1 | the next tile's loads (eb is dead since stage 2) and the d enc stores
#pragma unroll
    for (int g = 0; g < 16; ++g) tr1[drow(g, h) * TLD + r] = dp1[g];
    IMMOCO_LDS_DONE();
    if (ABL != 12 && t + n_waves < n_tiles) load_tile(t + n_waves);
    if (ABL != 1 && ABL != 4 && ABL != 8) dw1_chain(tr1, dw1t1);
    if (ABL == 4 || ABL == 8) {   // keep the LDS transposes alive without MFMAs
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        dw1t0[g] += tr0[r * TLD + g + 16 * h] + te[r * TLD + g];
        dw1t1[g] += tr1[r * TLD + g + 16 * h];
        denc[g] = dp0[g] + dp1[g];
      }
    }
#undef IMMOCO_MIX
    if (valid) {   // rows = feature (g&3) + 8(g>>2) + 4h, col = point
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int level = 4 * a + 2 * h;
        *reinterpret_cast<float2*>(din + p * ps + (int64_t)level * ls) = make_float2(denc[4 * a], denc[4 * a + 1]);
        *reinterpret_cast<float2*>(din + p * ps + (int64_t)(level + 1) * ls) =
            make_float2(denc[4 * a + 2], denc[4 * a + 3]);
      }
    }
  }
  const bool has_tile = wave_id < n_tiles;
  if (ABL == 3) return;
  if (ABL == 12) {
    dw1 += (size_t)blockIdx.x * 2048;
    dw2 += (size_t)(blockIdx.x * 4 + wave) * 128;
  }
  if (ABL == 5 || ABL == 6) {
    const int c = blockIdx.x & (ABL == 5 ? 3 : 15);
    dw1 += (size_t)c * 2048;
    dw2 += (size_t)c * 1024;
  }
  // ---- flush (as in mlp_bwd_mfma_kernel<64, ACT>; tile stride 3 per wave here)
#pragma unroll
  for (int jt = 0; jt < NJT; ++jt) {
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 16; ++g) tr0[r * TLD + drow(g, h)] = jt == 0 ? dw1t0[g] : dw1t1[g];  // [hidden r][feature k]
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = (4 * wave + k) * 64 + lane;
      const int off = (idx >> 5) * TLD + (idx & 31);
      const float v = (tr_all[off] + tr_all[3 * 32 * TLD + off]) + (tr_all[6 * 32 * TLD + off] + tr_all[9 * 32 * TLD + off]);
      if (ABL == 9 || ABL == 12) dw1[(size_t)jt * 1024 + idx] = v;
      else unsafeAtomicAdd(dw1 + (size_t)jt * 1024 + idx, v);
    }
    if (!has_tile && ABL != 12) continue;   // (12: a wave without tiles stores its zeros)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      float v0 = dw2l[jt][g][0], v1 = dw2l[jt][g][1];
#pragma unroll
      for (int m = 16; m >= 1; m >>= 1) {
        v0 += __shfl_xor(v0, m, 64);
        v1 += __shfl_xor(v1, m, 64);
      }
      if (r == 0) {
        dos[drow(g, h)] = v0;
        dos[32 + drow(g, h)] = v1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (h == 0) {
      if (ABL == 9 || ABL == 12) {
        dw2[jt * 32 + r] = dos[r];
        dw2[HID + jt * 32 + r] = dos[32 + r];
      } else {
        unsafeAtomicAdd(dw2 + jt * 32 + r, dos[r]);
        unsafeAtomicAdd(dw2 + HID + jt * 32 + r, dos[32 + r]);
      }
    }
  }
}

// dw1[o] += sum_b p1[b][o] (o < 2048, b < nb), dw2[o] += sum_w p2[w][o] (o < 128, w < 4 nb).  Workgroups 0..127: 16 outputs
// of dW1 x 16 partial groups; workgroups 128..159: 4 outputs of dW2 x 64 partial groups (four times as many partials per
// output).  Every thread sums its group's partials in index order, the group sums are added in group order: deterministic.
__global__ __launch_bounds__(256) void mlp_dw_reduce_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int nb,
                                                            float* __restrict__ dw1, float* __restrict__ dw2) {
  __shared__ float red[64][17];
  const bool first = blockIdx.x < 128;
  const int no = first ? 16 : 4, ng = first ? 16 : 64;                 // outputs per workgroup, partial groups
  const int ol = threadIdx.x % no, q = threadIdx.x / no;
  const int o = first ? blockIdx.x * 16 + ol : (blockIdx.x - 128) * 4 + ol;
  const float* src = first ? p1 + o : p2 + o;
  const int stride = first ? 2048 : 128, count = first ? nb : 4 * nb;
  float s0 = 0.f;
#pragma unroll 8
  for (int b = q; b < count; b += ng) s0 += src[(size_t)b * stride];
  red[q][ol] = s0;
  __syncthreads();
  if (q == 0) {
    float t = 0.f;
    for (int k = 0; k < ng; ++k) t += red[k][ol];
    if (first) dw1[o] += t;
    else dw2[o] += t;
  }
}
#undef IMMOCO_CB
#undef IMMOCO_LDS_DONE
#endif  // IMMOCO_DIAG

// ---------------------------------------------------------------------------------------------
// Split backward of the WIDE net (round 4): two kernels that each fit BESIDE the motion grid's encode backward
// (csr_bwd_kernel<3>: 126 registers, 3 workgroups of 45 KB of LDS per CU) instead of one 448-register / 105 KB
// kernel that needs a CU to itself and therefore starves there (0.50 ms beside the gather against 0.11 ms alone:
// profiles/r03_iteration_timeline_f32.txt; the image chain ended 60-90 us after the motion chain).
//   mlp_bwd_denc_kernel  d enc = W1^T . ((W2^T dout) * act'(pre))   -> what the image grid's encode backward waits for
//                        (one wave per 32-point tile, as above; W1 / W1^T fragments in LDS: 66 KB, < 128 registers)
//   mlp_bwd_dw_kernel    dW1, dW2: a wave owns ONE tile of 32 hidden units (grid.y = 2 halves x 4 waves) for ALL of the
//                        workgroup's point tiles; pre' = enc . W1^T is computed with the point on the register index
//                        and the hidden unit on the lane (layout 2 directly: operands swapped), so dpre' IS the B operand
//                        of dW1^T = enc^T . dpre' with no transpose; dW2 = per-lane partial sums.  No weight fragments
//                        in LDS (the wave's 32 x 32 slice of W1 lives in 16 registers): 20 KB of LDS.
// Both recompute pre (128 MFMAs per tile each): together 512 MFMAs per tile - what the fused kernel spends too (its
// dW2 goes through a 16-MFMA chain per hidden tile).  Same arithmetic per product as the fused kernel: d enc is
// bit-identical, dW1 / dW2 differ in the order of the sum over points.
template <int HID, int ACT>
__global__ __launch_bounds__(256, 4) void mlp_bwd_denc_kernel(const float* __restrict__ in, int64_t ps, int64_t ls,
                                                              int64_t n, const float* __restrict__ w1,
                                                              const float* __restrict__ w2,
                                                              const float* __restrict__ dout, float* __restrict__ din,
                                                              int64_t n_tiles, int64_t dout_plane,
                                                              const float* __restrict__ dout2) {
  constexpr int NJT = HID / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float4* aw = reinterpret_cast<float4*>(smem);                  // W1 fragments    [NJT][4][64]
  float4* awt = aw + NJT * 4 * 64;                               // W1^T fragments  [NJT][4][64]
  float* w2s = reinterpret_cast<float*>(awt + NJT * 4 * 64);     // [2][HID]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  build_weight_frags<HID>(w1, aw, awt, threadIdx.x);
  for (int i = threadIdx.x; i < 2 * HID; i += 256) w2s[i] = w2[i];
  __syncthreads();
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  for (int64_t t = wave_id; t < n_tiles; t += n_waves) {
    const int64_t p = t * 32 + r;
    const bool valid = p < n;
    const int64_t pc = valid ? p : n - 1;
    const float mq = valid ? 1.f : 0.f;
    float eb[16];
    load_enc_b(in, ps, ls, p, n, h, eb);
    float2 d;
    if (dout_plane) {  // wave-uniform; dout2: a second planar addend (the warp backward's share of dL/dimage)
      d = make_float2(dout[pc], dout[dout_plane + pc]);
      if (dout2) d = make_float2(dout2[pc] + d.x, dout2[dout_plane + pc] + d.y);
      d = make_float2(d.x * mq, d.y * mq);
    } else {
      const float2 dv = *reinterpret_cast<const float2*>(dout + pc * 2);
      d = make_float2(dv.x * mq, dv.y * mq);
    }
    f32x16 denc = {0.f};
#pragma unroll 2
    for (int jt = 0; jt < NJT; ++jt) {
      const f32x16 pre = pre_tile(aw, jt, lane, eb);
      float dp[16];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 wa = *reinterpret_cast<const float4*>(w2s + jt * 32 + 8 * a + 4 * h);
        const float4 wb = *reinterpret_cast<const float4*>(w2s + HID + jt * 32 + 8 * a + 4 * h);
        const float was[4] = {wa.x, wa.y, wa.z, wa.w}, wbs[4] = {wb.x, wb.y, wb.z, wb.w};
#pragma unroll
        for (int b = 0; b < 4; ++b)
          dp[4 * a + b] = fmaf(was[b], d.x, wbs[b] * d.y) * act_d<ACT>(act_f<ACT>(pre[4 * a + b]));
      }
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float4 a = awt[(jt * 4 + g4) * 64 + lane];
        denc = mfma32(a.x, dp[4 * g4], denc);
        denc = mfma32(a.y, dp[4 * g4 + 1], denc);
        denc = mfma32(a.z, dp[4 * g4 + 2], denc);
        denc = mfma32(a.w, dp[4 * g4 + 3], denc);
      }
    }
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
}

template <int ACT>
__global__ __launch_bounds__(256, 4) void mlp_bwd_dw_kernel(const float* __restrict__ in, int64_t ps, int64_t ls,
                                                            int64_t n, const float* __restrict__ w1,
                                                            const float* __restrict__ w2,
                                                            const float* __restrict__ dout, float* __restrict__ dw1,
                                                            float* __restrict__ dw2, int64_t n_tiles,
                                                            int64_t dout_plane, const float* __restrict__ dout2) {
  constexpr int HID = 256;
  __shared__ __attribute__((aligned(16))) float te_all[4 * 32 * TLD];   // per wave: enc tile, rows = feature, cols = point
  __shared__ __attribute__((aligned(16))) float dos_all[4 * 64];        // per wave: dout [32 points][2]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  float* te = te_all + wave * 32 * TLD;
  float* dos = dos_all + wave * 64;
  // One hidden tile of 32 units per wave (two such waves' worth of state - 64 units - spills at 128 registers):
  // blockIdx.y selects the half of the hidden layer, the wave its tile; jt = 4 blockIdx.y + wave.
  const int jt = 4 * (int)blockIdx.y + wave;
  const int j = jt * 32 + r;
  // B fragments of pre' = enc . W1^T for hidden unit j: bw[s] = W1[j][2 s + h]
  float bw[16];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {   // 8 consecutive features per pair of 16-byte loads
    const float4 q0 = *reinterpret_cast<const float4*>(w1 + j * 32 + 8 * s4);
    const float4 q1 = *reinterpret_cast<const float4*>(w1 + j * 32 + 8 * s4 + 4);
    bw[4 * s4] = h ? q0.y : q0.x;
    bw[4 * s4 + 1] = h ? q0.w : q0.z;
    bw[4 * s4 + 2] = h ? q1.y : q1.x;
    bw[4 * s4 + 3] = h ? q1.w : q1.z;
  }
  const float w20 = w2[j], w21 = w2[HID + j];
  f32x16 dw1t = {0.f};   // dW1^T tile: dw1t[g] = dW1[j][k = drow(g, h)]
  float dw2l0 = 0.f, dw2l1 = 0.f;
  // (no software prefetch of the next tile: with it the kernel spills at the 128 registers that let it share a SIMD
  // with three encode-backward waves; four workgroups per CU hide the load latency instead)
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {   // all four waves walk the SAME tiles
    const int64_t q = t * 32 + r;
    const int64_t qc = q < n ? q : n - 1;
    const float mq = q < n ? 1.f : 0.f;
    float eb[16];
    load_enc_b(in, ps, ls, q, n, h, eb);
    float2 d;
    if (dout_plane) {  // wave-uniform
      d = make_float2(dout[qc], dout[dout_plane + qc]);
      if (dout2) d = make_float2(dout2[qc] + d.x, dout2[dout_plane + qc] + d.y);
      d = make_float2(d.x * mq, d.y * mq);
    } else {
      const float2 dv = *reinterpret_cast<const float2*>(dout + qc * 2);
      d = make_float2(dv.x * mq, dv.y * mq);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous tile's readers are done (per-wave tiles)
    if (h == 0) *reinterpret_cast<float2*>(dos + 2 * r) = d;
#pragma unroll
    for (int s = 0; s < 16; ++s) te[(2 * s + h) * TLD + r] = eb[s];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // pre'[point drow(g, h)][hidden r]: A = enc (row = point r, k = feature 2 s + h), B = W1^T (k, col = hidden r)
    f32x16 pp = {0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s) pp = mfma32(eb[s], bw[s], pp);
#pragma unroll
    for (int g = 0; g < 16; ++g) {
      const float2 dq = *reinterpret_cast<const float2*>(dos + 2 * drow(g, h));   // dout of point drow(g, h): broadcast
      const float hh = act_f<ACT>(pp[g]);
      dw2l0 = fmaf(hh, dq.x, dw2l0);
      dw2l1 = fmaf(hh, dq.y, dw2l1);
      pp[g] = fmaf(w20, dq.x, w21 * dq.y) * act_d<ACT>(hh);               // dpre'
    }
    // dW1^T[k][j] += sum_p enc[p][k] dpre'[p][j]: A = enc^T in accumulator-k order (enc[point drow(g,h)][feature r])
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float4 q4 = *reinterpret_cast<const float4*>(te + r * TLD + 8 * a + 4 * h);
      dw1t = mfma32(q4.x, pp[4 * a], dw1t);
      dw1t = mfma32(q4.y, pp[4 * a + 1], dw1t);
      dw1t = mfma32(q4.z, pp[4 * a + 2], dw1t);
      dw1t = mfma32(q4.w, pp[4 * a + 3], dw1t);
    }
  }
  // ---- flush: the wave's [32 hidden][32 features] tile of dW1 through its LDS tile -> 256-byte contiguous atomics
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int g = 0; g < 16; ++g) te[r * TLD + drow(g, h)] = dw1t[g];   // [hidden r][feature k]
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int idx = k * 64 + lane;   // (hidden = idx >> 5, feature = idx & 31) of this 32 x 32 tile
    unsafeAtomicAdd(dw1 + (size_t)jt * 1024 + idx, te[(idx >> 5) * TLD + (idx & 31)]);
  }
  dw2l0 += __shfl_xor(dw2l0, 32, 64);
  dw2l1 += __shfl_xor(dw2l1, 32, 64);
  if (h == 0) {
    unsafeAtomicAdd(dw2 + j, dw2l0);
    unsafeAtomicAdd(dw2 + HID + j, dw2l1);
  }
}

static size_t denc_smem(int hid) { return (size_t)(hid / 32) * 4 * 64 * 16 * 2 + (size_t)2 * hid * 4; }

// d enc only (din may NOT alias in: the dW kernel still needs the encoding)
int launch_mlp_bwd_denc(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n, const float* w1,
                        const float* w2, const float* dout, float* din, hipStream_t st, int64_t dout_plane,
                        const float* dout2) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
  IMMOCO_REQUIRE(dout2 == nullptr || dout_plane != 0, "mlp_bwd_denc: a second dout addend needs the planar layout");
  IMMOCO_REQUIRE(cfg.n_hidden == 256 && in != din, "mlp_bwd_denc: the split backward is built for the 256-wide net, out of place");
  const int64_t n_tiles = cdiv(n, 32);
  // 2 workgroups per CU at most (66 KB of LDS each); a workgroup's fragment build is amortised over its tiles
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 512);
  const size_t sm = denc_smem(256);
#define IMMOCO_DENC(A)                                                                                               \
  do {                                                                                                               \
    static bool attr_set = false;                                                                                    \
    if (!attr_set) {                                                                                                 \
      IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_denc_kernel<256, A>),              \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));                   \
      attr_set = true;                                                                                               \
    }                                                                                                                \
    mlp_bwd_denc_kernel<256, A><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, dout, din, n_tiles, dout_plane, dout2); \
  } while (0)
  if (cfg.activation == IMMOCO_ACT_TANH) IMMOCO_DENC(IMMOCO_ACT_TANH);
  else IMMOCO_DENC(IMMOCO_ACT_RELU);
#undef IMMOCO_DENC
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// dW1 += ..., dW2 += ... (atomics into the caller's gradient buffers, like the fused kernel)
int launch_mlp_bwd_dw(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n, const float* w1,
                      const float* w2, const float* dout, float* dw1, float* dw2, hipStream_t st, int64_t dout_plane,
                      const float* dout2) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
  IMMOCO_REQUIRE(dout2 == nullptr || dout_plane != 0, "mlp_bwd_dw: a second dout addend needs the planar layout");
  IMMOCO_REQUIRE(cfg.n_hidden == 256, "mlp_bwd_dw: the split backward is built for the 256-wide net");
  const int64_t n_tiles = cdiv(n, 32);
  // grid.y: the two halves of the hidden layer; every workgroup flushes 16 KB of atomics once
  const dim3 grid((unsigned)std::min<int64_t>(n_tiles, 384), 2);
  if (cfg.activation == IMMOCO_ACT_TANH)
    mlp_bwd_dw_kernel<IMMOCO_ACT_TANH><<<grid, 256, 0, st>>>(in, ps, ls, n, w1, w2, dout, dw1, dw2, n_tiles, dout_plane, dout2);
  else
    mlp_bwd_dw_kernel<IMMOCO_ACT_RELU><<<grid, 256, 0, st>>>(in, ps, ls, n, w1, w2, dout, dw1, dw2, n_tiles, dout_plane, dout2);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

static size_t fwd_smem(int hid) { return (size_t)(hid / 32) * 4 * 64 * 16 + (size_t)2 * hid * 4; }
static size_t bwd_smem(int hid) {
  return (size_t)(hid / 32) * 4 * 64 * 16 * 2 + (size_t)2 * hid * 4 + 4 * 64 * 4 + (size_t)4 * 2 * 32 * TLD * 4;
}

int launch_mlp_fwd_mfma(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                        const float* w1, const float* w2, float* out, hipStream_t st) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
  const int64_t n_tiles = cdiv(n, 32);
  // persistent grid (2 workgroups per CU): the per-workgroup weight-fragment build is amortised
  // over ~16 tiles per wave instead of ~4
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 512);
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
  const int blocks_per_cu = HID == 64 ? 2 : 1;
  unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 256 * blocks_per_cu);
#ifdef IMMOCO_DIAG   // IMMOCO_MLP_GRID=<n>: cap the persistent grid (256 = one workgroup per CU, one wave per SIMD)
  static const int grid_cap = [] { const char* e = immoco_diag_env("IMMOCO_MLP_GRID"); return e ? atoi(e) : 0; }();
  if (grid_cap > 0) grid = std::min<unsigned>(grid, (unsigned)grid_cap);
#endif
  const size_t sm = bwd_smem(HID);
  static bool attr_set = false;
  if (!attr_set) {
    IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_mfma_kernel<HID, ACT>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    attr_set = true;
  }
#ifdef IMMOCO_DIAG
  static const bool raw = [] { const char* e = immoco_diag_env("IMMOCO_MLP_RAWLOAD"); return e && atoi(e) != 0; }();
  if (raw && HID == 64) {
    IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd_mfma_kernel<HID, ACT, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
    mlp_bwd_mfma_kernel<HID, ACT, true><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, n_tiles, dout_plane);
    IMMOCO_LAUNCH_CHECK();
    return IMMOCO_OK;
  }
#endif
  mlp_bwd_mfma_kernel<HID, ACT><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, n_tiles, dout_plane);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_mlp_bwd_mfma(const immoco_mlp_cfg& cfg, const float* in, int64_t ps, int64_t ls, int64_t n,
                        const float* w1, const float* w2, const float* dout, float* din, float* dw1, float* dw2,
                        hipStream_t st, int64_t dout_plane) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "mlp input strides must be even");
#ifdef IMMOCO_DIAG
  // diagnostics build only: IMMOCO_MLP_BWD64=pipe selects the software-pipelined narrow backward (experiment, see above)
  static const int pipe64 = [] {
    const char* e = immoco_diag_env("IMMOCO_MLP_BWD64");
    return !e ? 0 : strcmp(e, "pipe") == 0 ? 1 : strcmp(e, "pipe_nodw1") == 0 ? 2 : strcmp(e, "pipe_nodenc") == 0 ? 3
           : strcmp(e, "pipe_noflush") == 0 ? 4 : strcmp(e, "pipe_nomfma") == 0 ? 5
           : strcmp(e, "pipe_spread4") == 0 ? 6 : strcmp(e, "pipe_spread16") == 0 ? 7
           : strcmp(e, "pipe_fixed") == 0 ? 8 : strcmp(e, "pipe_memonly") == 0 ? 9
           : strcmp(e, "pipe_fixed_noatomic") == 0 ? 10 : strcmp(e, "pipe_fixed_nobuild") == 0 ? 11 : strcmp(e, "pipe_empty") == 0 ? 12
           : strcmp(e, "pipe_partials") == 0 ? 13 : 0;
  }();
  if (pipe64 && cfg.n_hidden == 64) {
    const int64_t n_tiles = cdiv(n, 32);
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n_tiles, 4), 512);
    const size_t sm = (size_t)2 * 4 * 64 * 16 * 2 + (size_t)2 * 64 * 4 + 4 * 64 * 4 + (size_t)4 * 3 * 32 * TLD * 4;
#define IMMOCO_PIPE(A, B)                                                                                              \
  do {                                                                                                                 \
    IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd64_pipe_kernel<A, B>),                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));                        \
    mlp_bwd64_pipe_kernel<A, B><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, n_tiles, dout_plane); \
  } while (0)
    if (pipe64 == 13 && cfg.activation == IMMOCO_ACT_TANH) {   // partial sums + fixed-order reduction instead of atomics
      static float* scratch = nullptr;   // [512][2048] + [2048][128] floats (diagnostics build: never freed)
      if (!scratch) IMMOCO_CHECK_HIP(hipMalloc((void**)&scratch, ((size_t)512 * 2048 + (size_t)2048 * 128) * sizeof(float)));
      float* p1 = scratch;
      float* p2 = scratch + (size_t)512 * 2048;
      IMMOCO_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&mlp_bwd64_pipe_kernel<IMMOCO_ACT_TANH, 12>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
      mlp_bwd64_pipe_kernel<IMMOCO_ACT_TANH, 12><<<grid, 256, sm, st>>>(in, ps, ls, n, w1, w2, dout, din, p1, p2, n_tiles, dout_plane);
      mlp_dw_reduce_kernel<<<128 + 32, 256, 0, st>>>(p1, p2, (int)grid, dw1, dw2);
      IMMOCO_LAUNCH_CHECK();
      return IMMOCO_OK;
    }
    if (cfg.activation != IMMOCO_ACT_TANH) IMMOCO_PIPE(IMMOCO_ACT_RELU, 0);
    else if (pipe64 == 2) IMMOCO_PIPE(IMMOCO_ACT_TANH, 1);
    else if (pipe64 == 3) IMMOCO_PIPE(IMMOCO_ACT_TANH, 2);
    else if (pipe64 == 4) IMMOCO_PIPE(IMMOCO_ACT_TANH, 3);
    else if (pipe64 == 5) IMMOCO_PIPE(IMMOCO_ACT_TANH, 4);
    else if (pipe64 == 6) IMMOCO_PIPE(IMMOCO_ACT_TANH, 5);
    else if (pipe64 == 7) IMMOCO_PIPE(IMMOCO_ACT_TANH, 6);
    else if (pipe64 == 8) IMMOCO_PIPE(IMMOCO_ACT_TANH, 7);
    else if (pipe64 == 9) IMMOCO_PIPE(IMMOCO_ACT_TANH, 8);
    else if (pipe64 == 10) IMMOCO_PIPE(IMMOCO_ACT_TANH, 9);
    else if (pipe64 == 11) IMMOCO_PIPE(IMMOCO_ACT_TANH, 10);
    else if (pipe64 == 12) IMMOCO_PIPE(IMMOCO_ACT_TANH, 11);
    else IMMOCO_PIPE(IMMOCO_ACT_TANH, 0);
#undef IMMOCO_PIPE
    IMMOCO_LAUNCH_CHECK();
    return IMMOCO_OK;
  }
#endif
  if (cfg.n_hidden == 64 && cfg.activation == IMMOCO_ACT_TANH)
    return launch_bwd_t<64, IMMOCO_ACT_TANH>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  if (cfg.n_hidden == 64)
    return launch_bwd_t<64, IMMOCO_ACT_RELU>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  if (cfg.activation == IMMOCO_ACT_TANH)
    return launch_bwd_t<256, IMMOCO_ACT_TANH>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
  return launch_bwd_t<256, IMMOCO_ACT_RELU>(in, ps, ls, n, w1, w2, dout, din, dw1, dw2, st, dout_plane);
}

}  // namespace immoco
