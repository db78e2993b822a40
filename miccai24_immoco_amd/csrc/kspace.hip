// k-space side of the forward operator: centred 2-D FFTs through rocFFT (hipFFT
// API), line select, data-consistency loss, gradient-entropy loss, k-space
// normalisation.  Reference: src/utils/data_utils.py:29-34 (FFT/IFFT),
// src/models/immoco.py:109-111 (line select), :137-141 (normalisation),
// :170-172 (losses), src/utils/losses.py:20-40 (gradient entropy).
//
// Centred FFT without the two roll copies: for even N,
//   fftshift(F(ifftshift(x)))[k] = (-1)^(k+N/2) * F{(-1)^n x[n]}[k]
// per dimension, so both shifts fold into +-1 sign multiplications that are
// fused into the producer/consumer kernels.  Odd sizes take explicit rolls.
#include <hipfft/hipfft.h>

#include <map>
#include <mutex>
#include <tuple>

#include "kernels.hpp"

namespace immoco {

// ---- plan cache -------------------------------------------------------------
namespace {
std::mutex g_plan_mu;
std::map<std::tuple<int, int, int, int>, hipfftHandle> g_plans;  // (device, batch, H, W)

int get_plan(int batch, int H, int W, hipfftHandle* out) {
  int dev = 0;
  IMMOCO_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_plan_mu);
  auto key = std::make_tuple(dev, batch, H, W);
  auto it = g_plans.find(key);
  if (it != g_plans.end()) {
    *out = it->second;
    return IMMOCO_OK;
  }
  hipfftHandle plan;
  int dims[2] = {H, W};
  hipfftResult r = hipfftPlanMany(&plan, 2, dims, nullptr, 1, H * W, nullptr, 1, H * W, HIPFFT_C2C, batch);
  if (r != HIPFFT_SUCCESS) {
    set_error("hipfftPlanMany(batch=%d, %dx%d) failed: %d", batch, H, W, (int)r);
    return IMMOCO_E_FFT;
  }
  g_plans[key] = plan;
  *out = plan;
  return IMMOCO_OK;
}
}  // namespace

int fft_exec_inplace(float* buf, int batch, int H, int W, bool inverse, hipStream_t st) {
  hipfftHandle plan;
  int rc = get_plan(batch, H, W, &plan);
  if (rc) return rc;
  hipfftResult r = hipfftSetStream(plan, st);
  if (r == HIPFFT_SUCCESS)
    r = hipfftExecC2C(plan, (hipfftComplex*)buf, (hipfftComplex*)buf, inverse ? HIPFFT_BACKWARD : HIPFFT_FORWARD);
  if (r != HIPFFT_SUCCESS) {
    set_error("hipfftExecC2C failed: %d", (int)r);
    return IMMOCO_E_FFT;
  }
  return IMMOCO_OK;
}

// ---- 2-D transform as two 1-D passes around a TRANSPOSED k-space layout (solver path) ---------------
// rocFFT runs a batched 320x320 transform as row FFT, transpose, row FFT, transpose (4 kernels, 35 us for
// batch 11).  The solver only touches k-space in one kernel between the forward and the adjoint
// transform, so it can live with k-space stored as [kx = W][b][ky = H]: pass 1 transforms the rows of
// [b][H][W] and writes them with stride B*H (one strided-output kernel), pass 2 transforms the now
// contiguous H axis in place - 2 kernels, 18 us, bit-identical values; the adjoint runs them backwards.
namespace {
struct SplitPlans {
  hipfftHandle rows_out_t;  // along W: [B*H][W] -> [W][B*H]
  hipfftHandle cols;        // along H, contiguous, batch W*B, in place
  hipfftHandle rows_in_t;   // along W: [W][B*H] -> [B*H][W]
};
std::map<std::tuple<int, int, int, int>, SplitPlans> g_split;  // (device, batch, H, W)

int get_split_plans(int B, int H, int W, SplitPlans* out) {
  int dev = 0;
  IMMOCO_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_plan_mu);
  auto key = std::make_tuple(dev, B, H, W);
  auto it = g_split.find(key);
  if (it != g_split.end()) {
    *out = it->second;
    return IMMOCO_OK;
  }
  SplitPlans sp{};
  int nw[1] = {W}, nh[1] = {H};
  hipfftResult r = hipfftPlanMany(&sp.rows_out_t, 1, nw, nw, 1, W, nw, B * H, 1, HIPFFT_C2C, B * H);
  if (r == HIPFFT_SUCCESS) r = hipfftPlanMany(&sp.cols, 1, nh, nh, 1, H, nh, 1, H, HIPFFT_C2C, W * B);
  if (r == HIPFFT_SUCCESS) r = hipfftPlanMany(&sp.rows_in_t, 1, nw, nw, B * H, 1, nw, 1, W, HIPFFT_C2C, B * H);
  if (r != HIPFFT_SUCCESS) {
    set_error("hipfftPlanMany(split, batch=%d, %dx%d) failed: %d", B, H, W, (int)r);
    return IMMOCO_E_FFT;
  }
  g_split[key] = sp;
  *out = sp;
  return IMMOCO_OK;
}

// Plans are cached process-wide and shared by all solver handles (several same-shape slices may be in flight
// from different host threads, ctypes releases the GIL): SetStream + Exec on a shared handle is one critical
// section, otherwise another thread's SetStream could land between the two and the transform would be
// enqueued - or captured - on the wrong stream.  (The 1-D plans used here need no rocFFT work buffer.)
std::mutex g_exec_mu;
int exec_c2c(hipfftHandle plan, float* in, float* out, int dir, hipStream_t st) {
  std::lock_guard<std::mutex> lk(g_exec_mu);
  hipfftResult r = hipfftSetStream(plan, st);
  if (r == HIPFFT_SUCCESS) r = hipfftExecC2C(plan, (hipfftComplex*)in, (hipfftComplex*)out, dir);
  if (r != HIPFFT_SUCCESS) {
    set_error("hipfftExecC2C failed: %d", (int)r);
    return IMMOCO_E_FFT;
  }
  return IMMOCO_OK;
}
}  // namespace

// forward: in [B][H][W] -> out_t [W][B][H] (raw transform, no shifts)
int fft_fwd_to_transposed(float* in, float* out_t, int B, int H, int W, hipStream_t st) {
  SplitPlans sp;
  int rc = get_split_plans(B, H, W, &sp);
  if (rc) return rc;
  if ((rc = exec_c2c(sp.rows_out_t, in, out_t, HIPFFT_FORWARD, st))) return rc;
  return exec_c2c(sp.cols, out_t, out_t, HIPFFT_FORWARD, st);
}

// adjoint (unnormalised inverse): in_t [W][B][H] (overwritten) -> out [B][H][W]
int fft_adj_from_transposed(float* in_t, float* out, int B, int H, int W, hipStream_t st) {
  SplitPlans sp;
  int rc = get_split_plans(B, H, W, &sp);
  if (rc) return rc;
  if ((rc = exec_c2c(sp.cols, in_t, in_t, HIPFFT_BACKWARD, st))) return rc;
  return exec_c2c(sp.rows_in_t, in_t, out, HIPFFT_BACKWARD, st);
}

// ---- pieces of the same split transform for ONE image (pruned path of the solver, warp.hip: the motion images never
// go through a full 2-D transform there; only the unwarped image's rows and the W selected k-space columns do)
int fft_rows_fwd_to_t(float* in, float* out_t, int H, int W, hipStream_t st) {       // [H][W] -> [W][H], along W
  SplitPlans sp;
  int rc = get_split_plans(1, H, W, &sp);
  if (rc) return rc;
  return exec_c2c(sp.rows_out_t, in, out_t, HIPFFT_FORWARD, st);
}
int fft_cols_inplace_t(float* zt, int H, int W, bool inverse, hipStream_t st) {      // [W][H], along H, in place
  SplitPlans sp;
  int rc = get_split_plans(1, H, W, &sp);
  if (rc) return rc;
  return exec_c2c(sp.cols, zt, zt, inverse ? HIPFFT_BACKWARD : HIPFFT_FORWARD, st);
}
int fft_rows_adj_from_t(float* in_t, float* out, int H, int W, hipStream_t st) {     // [W][H] -> [H][W], along W, unnormalised inverse
  SplitPlans sp;
  int rc = get_split_plans(1, H, W, &sp);
  if (rc) return rc;
  return exec_c2c(sp.rows_in_t, in_t, out, HIPFFT_BACKWARD, st);
}

// Column lists of the line masks (immoco.py:109-111: k-space column c comes from image g(c), 0 = the unwarped one):
// cols[off[g] .. off[g + 1]) = the columns of group g in increasing order, g = 0 .. nM.  One workgroup; W and nM are
// a few hundred at most.
__global__ __launch_bounds__(256) void build_col_lists_kernel(const int32_t* __restrict__ col_group, int nM, int W,
                                                              int32_t* __restrict__ cols, int32_t* __restrict__ off) {
  __shared__ int cnt[256], start[257];
  const int g = threadIdx.x;   // one thread per group (nM <= 255, checked by the launcher)
  int n = 0;
  if (g <= nM)
    for (int c = 0; c < W; ++c) {
      int v = col_group[c];
      v = v < 0 ? 0 : (v > nM ? 0 : v);
      n += v == g;
    }
  cnt[g] = n;
  __syncthreads();
  if (g == 0) {
    int run = 0;
    for (int k = 0; k <= nM; ++k) {
      start[k] = run;
      run += cnt[k];
    }
    start[nM + 1] = run;
  }
  __syncthreads();
  if (g <= nM + 1) off[g] = start[g];
  if (g <= nM) {
    int w = start[g];
    for (int c = 0; c < W; ++c) {
      int v = col_group[c];
      v = v < 0 ? 0 : (v > nM ? 0 : v);
      if (v == g) cols[w++] = c;
    }
  }
}

int launch_build_col_lists(const int32_t* col_group, int nM, int W, int32_t* cols, int32_t* off, hipStream_t st) {
  IMMOCO_REQUIRE(nM >= 0 && nM <= 254, "column lists: at most 254 motion groups (got %d)", nM);
  build_col_lists_kernel<<<1, 256, 0, st>>>(col_group, nM, W, cols, off);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// out_t[c][r] = g(c) == 0 ? in_t[c][r] : 0   (adjoint seed of the unwarped image: its own columns only)
__global__ __launch_bounds__(256) void keep_group0_cols_kernel(const float2* __restrict__ in_t,
                                                               const int32_t* __restrict__ col_group, int nM, int H,
                                                               int W, float2* __restrict__ out_t) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // = c * H + r
  if (i >= (int64_t)H * W) return;
  int g = col_group[(int)(i / H)];
  g = g < 0 ? 0 : (g > nM ? 0 : g);
  out_t[i] = g == 0 ? in_t[i] : make_float2(0.f, 0.f);
}

int launch_keep_group0_cols(const float* in_t, const int32_t* col_group, int nM, int H, int W, float* out_t,
                            hipStream_t st) {
  keep_group0_cols_kernel<<<(unsigned)cdiv((int64_t)H * W, 256), 256, 0, st>>>((const float2*)in_t, col_group, nM, H, W,
                                                                              (float2*)out_t);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// out[c][r] = in[r][c]  (complex, once per solve: the measured k-space into / the result out of the layout above)
__global__ __launch_bounds__(256) void transpose_c64_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                            int H, int W) {
  __shared__ float2 tile[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + tx, r = blockIdx.y * 16 + ty;
  if (r < H && c < W) tile[ty][tx] = in[(int64_t)r * W + c];
  __syncthreads();
  const int r2 = blockIdx.y * 16 + tx, c2 = blockIdx.x * 16 + ty;
  if (r2 < H && c2 < W) out[(int64_t)c2 * H + r2] = tile[tx][ty];
}

int launch_transpose_c64(const float* in, float* out, int H, int W, hipStream_t st) {
  dim3 grid((unsigned)cdiv(W, 16), (unsigned)cdiv(H, 16));
  transpose_c64_kernel<<<grid, 256, 0, st>>>((const float2*)in, (float2*)out, H, W);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// out[b][(r+sr)%H][(c+sc)%W] = in[b][r][c] * mul * (sign ? (-1)^(r+c) : 1)
__global__ __launch_bounds__(256) void roll_scale_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                         int64_t n, int H, int W, int sr, int sc, float mul,
                                                         int checker) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % W), r = (int)((i / W) % H);
  const int64_t b = i / ((int64_t)H * W);
  float s = mul;
  if (checker && ((r + c) & 1)) s = -s;
  const float2 v = in[i];
  const int r2 = (r + sr) % H, c2 = (c + sc) % W;
  out[(b * H + r2) * W + c2] = make_float2(v.x * s, v.y * s);
}

static int launch_roll(const float* in, float* out, int64_t n, int H, int W, int sr, int sc, float mul,
                       int checker, hipStream_t st) {
  roll_scale_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)in, (float2*)out, n, H, W, sr, sc,
                                                            mul, checker);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

namespace {
std::mutex g_tmp_mu;
std::map<int, std::pair<void*, size_t>> g_tmp;  // per-device scratch for the odd-size path
int get_tmp(size_t bytes, void** out) {
  int dev = 0;
  IMMOCO_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_tmp_mu);
  auto& e = g_tmp[dev];
  if (e.second < bytes) {
    if (e.first) IMMOCO_CHECK_HIP(hipFree(e.first));
    e = {nullptr, 0};
    IMMOCO_CHECK_HIP(hipMalloc(&e.first, bytes));
    e.second = bytes;
  }
  *out = e.first;
  return IMMOCO_OK;
}
}  // namespace

int fft2c(const float* in, float* out, int batch, int H, int W, int mode, hipStream_t st) {
  const int64_t n = (int64_t)batch * H * W;
  if (n == 0) return IMMOCO_OK;
  // mode 0: FFT; 1: IFFT; 2: adjoint of FFT (FFT's shifts around the unnormalised inverse transform);
  // 3: adjoint of IFFT (IFFT's shifts around the forward transform, / (H*W)).  For even sizes both shifts are
  // the same roll and 2 / 3 coincide with IFFT*(HW) / FFT/(HW); for odd sizes they do not.
  const bool inverse = mode == 1 || mode == 2;
  const float norm = (mode == 1 || mode == 3) ? 1.0f / ((float)H * (float)W) : 1.0f;
  int rc;
  if ((H % 2) == 0 && (W % 2) == 0) {
    // even sizes: checkerboard signs replace both rolls
    if ((rc = launch_roll(in, out, n, H, W, 0, 0, 1.0f, 1, st))) return rc;
    if ((rc = fft_exec_inplace(out, batch, H, W, inverse, st))) return rc;
    const float gs = (((H / 2) + (W / 2)) & 1) ? -norm : norm;
    return launch_roll(out, out, n, H, W, 0, 0, gs, 1, st);
  }
  // generic sizes.  FFT: pre = ifftshift (roll by -(N/2)), post = fftshift (roll by N/2);
  // IFFT/adjoint: pre = fftshift, post = ifftshift.
  void* tmp = nullptr;
  if ((rc = get_tmp((size_t)n * 8, &tmp))) return rc;
  // (mode 2, the adjoint of mode 0, keeps mode 0's shifts around the inverse transform.)
  const int hr = H / 2, hc = W / 2;
  const bool ifft_shifts = mode == 1 || mode == 3;
  const int pre_r = ifft_shifts ? hr : (H - hr) % H, pre_c = ifft_shifts ? hc : (W - hc) % W;
  const int post_r = ifft_shifts ? (H - hr) % H : hr, post_c = ifft_shifts ? (W - hc) % W : hc;
  if ((rc = launch_roll(in, (float*)tmp, n, H, W, pre_r, pre_c, 1.0f, 0, st))) return rc;
  if ((rc = fft_exec_inplace((float*)tmp, batch, H, W, inverse, st))) return rc;
  return launch_roll((const float*)tmp, out, n, H, W, post_r, post_c, norm, 0, st);
}

// ---- line select -------------------------------------------------------------
__global__ __launch_bounds__(256) void kspace_select_kernel(const float2* __restrict__ kall,
                                                            const int32_t* __restrict__ col_group, int nM,
                                                            int H, int W, float2* __restrict__ kout) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const int c = (int)(i % W);
  int g = col_group[c];
  g = g < 0 ? 0 : (g > nM ? 0 : g);
  kout[i] = kall[(int64_t)g * H * W + i];
}

int launch_kspace_select(const float* kall, const int32_t* col_group, int nM, int H, int W, float* kout,
                         hipStream_t st) {
  kspace_select_kernel<<<(unsigned)cdiv((int64_t)H * W, 256), 256, 0, st>>>((const float2*)kall, col_group, nM,
                                                                            H, W, (float2*)kout);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// block-level sum of one float per thread (256 threads) -> thread 0 returns the total
__device__ __forceinline__ float block_sum_256(float v) {
  __shared__ float red[4];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  if (threadIdx.x == 0) t = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return t;
}

// ---- DC loss -----------------------------------------------------------------
__global__ __launch_bounds__(256) void dc_loss_kernel(const float2* __restrict__ k, const float2* __restrict__ kin,
                                                      int64_t n, float inv_2n, float inv_n,
                                                      float* __restrict__ loss, float2* __restrict__ dk) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float part = 0.f;
  if (i < n) {
    const float2 a = k[i], b = kin[i];
    const float dr = a.x - b.x, di = a.y - b.y;
    part = dr * dr + di * di;
    if (dk) dk[i] = make_float2(dr * inv_n, di * inv_n);
  }
  const float tot = block_sum_256(part);
  if (threadIdx.x == 0 && loss) unsafeAtomicAdd(loss, tot * inv_2n);
}

int launch_dc_loss(const float* k, const float* kin, int H, int W, float* loss, float* dk, hipStream_t st) {
  const int64_t n = (int64_t)H * W;
  dc_loss_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)k, (const float2*)kin, n,
                                                         1.0f / (2.0f * (float)n), 1.0f / (float)n, loss,
                                                         (float2*)dk);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// ---- gradient entropy --------------------------------------------------------
// g(r,c) = |x[r,c]-x[r,c+1]| (0 on the last column) + |x[r,c]-x[r+1,c]| (0 on the last row)
// loss = -sum g log(g + 1e-24).  Gather form of the gradient (no atomics):
// G[r,c] = L'(r,c)(sgn a_rc + sgn b_rc) - L'(r,c-1) sgn a_{r,c-1} - L'(r-1,c) sgn b_{r-1,c}
// with a = x[r,c]-x[r,c+1], b = x[r,c]-x[r+1,c], sgn z = z/|z| (0 at 0), L'(g) = -(log(g+eps) + g/(g+eps)).
struct GeCell {
  float2 sa, sb;  // unit directions of a and b (0 when the difference is 0 / padded)
  float g;
};

__device__ __forceinline__ GeCell ge_cell(const float2* __restrict__ x, int r, int c, int H, int W) {
  GeCell o;
  const float2 v = x[(size_t)r * W + c];
  float ga = 0.f, gb = 0.f;
  o.sa = o.sb = make_float2(0.f, 0.f);
  if (c + 1 < W) {
    const float2 u = x[(size_t)r * W + c + 1];
    const float ar = v.x - u.x, ai = v.y - u.y;
    ga = hypotf(ar, ai);  // torch.abs of a complex uses hypot
    if (ga > 0.f) o.sa = make_float2(ar / ga, ai / ga);
  }
  if (r + 1 < H) {
    const float2 u = x[(size_t)(r + 1) * W + c];
    const float br = v.x - u.x, bi = v.y - u.y;
    gb = hypotf(br, bi);
    if (gb > 0.f) o.sb = make_float2(br / gb, bi / gb);
  }
  o.g = ga + gb;
  return o;
}

__device__ __forceinline__ float ge_dl(float g) {
  const float ge = g + 1e-24f;
  return -(logf(ge) + g / ge);
}

// MODE 0: op-level (weight passed by value; dimage accumulates)
// MODE 1: solver (weight = lambda_sched[*iter]; dimage = sign*adj0 + weight*dGE written; loss_hist[*iter])
// MODE 2: solver, pruned path: the warp backward ran BEFORE this kernel and added its share into the planar buffer
//         `dwarp`; dimage = dwarp + sign*adj0 + weight*dGE, and dwarp is cleared for the next iteration
template <int MODE>
__global__ __launch_bounds__(256) void ge_loss_kernel(const float2* __restrict__ x, int H, int W, float weight,
                                                      const float* __restrict__ lambda_sched,
                                                      const int32_t* __restrict__ iter_dev,
                                                      const float2* __restrict__ adj0, float* __restrict__ loss,
                                                      float2* __restrict__ dimage, float* __restrict__ dwarp = nullptr) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = (int64_t)H * W;
  int it = 0;
  if (MODE >= 1) {
    it = *iter_dev;
    weight = lambda_sched[it];
  }
  float part = 0.f;
  if (i < n) {
    const int c = (int)(i % W), r = (int)(i / W);
    const GeCell me = ge_cell(x, r, c, H, W);
    part = -me.g * logf(me.g + 1e-24f);
    if (dimage) {
      const float dl = ge_dl(me.g);
      float gx = dl * (me.sa.x + me.sb.x), gy = dl * (me.sa.y + me.sb.y);
      if (c > 0) {
        const GeCell lf = ge_cell(x, r, c - 1, H, W);
        const float d = ge_dl(lf.g);
        gx -= d * lf.sa.x;
        gy -= d * lf.sa.y;
      }
      if (r > 0) {
        const GeCell up = ge_cell(x, r - 1, c, H, W);
        const float d = ge_dl(up.g);
        gx -= d * up.sb.x;
        gy -= d * up.sb.y;
      }
      if (MODE == 0) {
        float2 o = dimage[i];
        o.x += weight * gx;
        o.y += weight * gy;
        dimage[i] = o;
      } else {
        // solver: PLANAR gradient image (re plane, im plane), see motion_warp_bwd_kernel
        const float s = ((r + c) & 1) ? -1.f : 1.f;
        const float2 a = adj0[i];
        float* pl = reinterpret_cast<float*>(dimage);
        if (MODE == 2) {
          pl[i] = dwarp[i] + (a.x * s + weight * gx);
          pl[n + i] = dwarp[n + i] + (a.y * s + weight * gy);
          dwarp[i] = 0.f;
          dwarp[n + i] = 0.f;
        } else {
          pl[i] = a.x * s + weight * gx;
          pl[n + i] = a.y * s + weight * gy;
        }
      }
    }
  }
  const float tot = block_sum_256(part);
  if (threadIdx.x == 0 && loss) unsafeAtomicAdd(loss + it, tot * weight);
}

int launch_ge_loss(const float* image, int H, int W, float weight, const float* /*weight_dev*/, float* loss,
                   float* dimage, hipStream_t st) {
  const int64_t n = (int64_t)H * W;
  ge_loss_kernel<0><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)image, H, W, weight, nullptr, nullptr,
                                                            nullptr, loss, (float2*)dimage);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_image_grad_init(const float* image, const float* adj_slot0, int H, int W, const float* lambda_sched,
                           const int32_t* iter_dev, float* loss_hist, float* dimage, hipStream_t st) {
  const int64_t n = (int64_t)H * W;
  ge_loss_kernel<1><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)image, H, W, 0.f, lambda_sched,
                                                            iter_dev, (const float2*)adj_slot0, loss_hist,
                                                            (float2*)dimage);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_image_grad_init_after_warp(const float* image, const float* adj_slot0, int H, int W, const float* lambda_sched,
                                      const int32_t* iter_dev, float* loss_hist, float* dimage, float* dwarp,
                                      hipStream_t st) {
  const int64_t n = (int64_t)H * W;
  ge_loss_kernel<2><<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)image, H, W, 0.f, lambda_sched, iter_dev,
                                                            (const float2*)adj_slot0, loss_hist, (float2*)dimage, dwarp);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// ---- solver-fused pieces -----------------------------------------------------
__global__ __launch_bounds__(256) void image_to_slot_kernel(const float2* __restrict__ img, int64_t n, int W,
                                                            float2* __restrict__ slot0) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % W), r = (int)(i / W);
  const float s = ((r + c) & 1) ? -1.f : 1.f;
  const float2 v = img[i];
  slot0[i] = make_float2(v.x * s, v.y * s);
}

int launch_image_to_slot(const float* image, int H, int W, float* slot0, hipStream_t st) {
  const int64_t n = (int64_t)H * W;
  image_to_slot_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)image, n, W, (float2*)slot0);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// After the forward FFT of all nM+1 slots: K[r,c] = sout(r,c) * slot[g(c)][r,c]  (immoco.py:109-111);
// DC residual and loss (immoco.py:170); then every slot is overwritten with the adjoint seed
// sout * (K - kin)/(HW) on its own columns and 0 elsewhere, ready for the backward FFT.
__global__ __launch_bounds__(256) void select_dc_seed_kernel(float2* __restrict__ fftbuf,
                                                             const int32_t* __restrict__ col_group,
                                                             const float2* __restrict__ kin, int nM, int H, int W,
                                                             float gsign, float2* __restrict__ kout,
                                                             float* __restrict__ loss_hist,
                                                             const int32_t* __restrict__ iter_dev) {
  const int64_t n = (int64_t)H * W;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float part = 0.f;
  if (i < n) {
    const int c = (int)(i % W), r = (int)(i / W);
    int g = col_group[c];
    g = g < 0 ? 0 : (g > nM ? 0 : g);
    const float s = (((r + c) & 1) ? -1.f : 1.f) * gsign;
    const float2 v = fftbuf[(int64_t)g * n + i];
    const float2 K = make_float2(v.x * s, v.y * s);
    if (kout) kout[i] = K;
    const float2 kk = kin[i];
    const float dr = K.x - kk.x, di = K.y - kk.y;
    part = dr * dr + di * di;
    const float inv_n = 1.0f / (float)n;
    const float2 seed = make_float2(dr * inv_n * s, di * inv_n * s);
    for (int b = 0; b <= nM; ++b) fftbuf[(int64_t)b * n + i] = (b == g) ? seed : make_float2(0.f, 0.f);
  }
  const float tot = block_sum_256(part);
  if (threadIdx.x == 0 && loss_hist) unsafeAtomicAdd(loss_hist + *iter_dev, tot / (2.0f * (float)n));
}

// The same on the transposed k-space layout of fft_fwd_to_transposed: fft_t [W][nM+1][H]; kin_t / kout_t [W][H].
__global__ __launch_bounds__(256) void select_dc_seed_t_kernel(float2* __restrict__ fft_t,
                                                               const int32_t* __restrict__ col_group,
                                                               const float2* __restrict__ kin_t, int nM, int H, int W,
                                                               float gsign, float2* __restrict__ kout_t,
                                                               float* __restrict__ loss_hist,
                                                               const int32_t* __restrict__ iter_dev) {
  const int64_t n = (int64_t)H * W;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // = c * H + r
  const int B = nM + 1;
  float part = 0.f;
  if (i < n) {
    const int r = (int)(i % H), c = (int)(i / H);
    int g = col_group[c];
    g = g < 0 ? 0 : (g > nM ? 0 : g);
    const float s = (((r + c) & 1) ? -1.f : 1.f) * gsign;
    float2* col = fft_t + (int64_t)c * B * H + r;
    const float2 v = col[(int64_t)g * H];
    const float2 K = make_float2(v.x * s, v.y * s);
    if (kout_t) kout_t[i] = K;
    const float2 kk = kin_t[i];
    const float dr = K.x - kk.x, di = K.y - kk.y;
    part = dr * dr + di * di;
    const float inv_n = 1.0f / (float)n;
    const float2 seed = make_float2(dr * inv_n * s, di * inv_n * s);
    for (int b = 0; b <= nM; ++b) col[(int64_t)b * H] = (b == g) ? seed : make_float2(0.f, 0.f);
  }
  const float tot = block_sum_256(part);
  if (threadIdx.x == 0 && loss_hist) unsafeAtomicAdd(loss_hist + *iter_dev, tot / (2.0f * (float)n));
}

int launch_select_dc_seed_t(float* fft_t, const int32_t* col_group, const float* kin_t, int nM, int H, int W,
                            float* kout_t, float* loss_hist, const int32_t* iter_dev, hipStream_t st) {
  const int64_t n = (int64_t)H * W;
  const float gsign = (((H / 2) + (W / 2)) & 1) ? -1.f : 1.f;
  select_dc_seed_t_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((float2*)fft_t, col_group, (const float2*)kin_t,
                                                                  nM, H, W, gsign, (float2*)kout_t, loss_hist,
                                                                  iter_dev);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_select_dc_seed(float* fftbuf, const int32_t* col_group, const float* kin, int nM, int H, int W,
                          float* kout, float* loss_hist, const int32_t* iter_dev, hipStream_t st) {
  const int64_t n = (int64_t)H * W;
  const float gsign = (((H / 2) + (W / 2)) & 1) ? -1.f : 1.f;
  select_dc_seed_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((float2*)fftbuf, col_group, (const float2*)kin,
                                                                nM, H, W, gsign, (float2*)kout, loss_hist,
                                                                iter_dev);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// ---- normalisation -----------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_kernel(const float2* __restrict__ k, int64_t n,
                                                     unsigned int* __restrict__ out_bits) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float2 v = k[i];
    m = fmaxf(m, hypotf(v.x, v.y));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  // non-negative floats order like their bit patterns
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void scale_by_max_kernel(const float2* __restrict__ k, int64_t n, float target,
                                                           const float* __restrict__ scale,
                                                           float2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float s = *scale;
  const float2 v = k[i];
  // torch: kspace_corr.div(scale).mul(16000)  (complex / real scalar, then * real scalar)
  out[i] = make_float2((v.x / s) * target, (v.y / s) * target);
}

int launch_normalize(const float* k, int64_t n, float target, float* out, float* scale_out, hipStream_t st) {
  IMMOCO_CHECK_HIP(hipMemsetAsync(scale_out, 0, sizeof(float), st));
  const unsigned grid = (unsigned)std::min<int64_t>(cdiv(n, 256), 1024);
  absmax_kernel<<<grid, 256, 0, st>>>((const float2*)k, n, (unsigned int*)scale_out);
  IMMOCO_LAUNCH_CHECK();
  scale_by_max_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>((const float2*)k, n, target, scale_out,
                                                              (float2*)out);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

}  // namespace immoco

using namespace immoco;

extern "C" int immoco_fft2c(const float* in, float* out, int32_t batch, int32_t H, int32_t W, int32_t mode,
                            void* stream) {
  IMMOCO_REQUIRE(batch >= 0 && H > 0 && W > 0, "fft2c: bad shape batch=%d H=%d W=%d", batch, H, W);
  IMMOCO_REQUIRE(mode >= 0 && mode <= 3, "fft2c: bad mode %d", mode);
  IMMOCO_REQUIRE(batch == 0 || (in && out), "fft2c: NULL buffer");
  return fft2c(in, out, batch, H, W, mode, as_stream(stream));
}

extern "C" int immoco_kspace_select(const float* kall, const int32_t* col_group, int32_t nM, int32_t H,
                                    int32_t W, float* kout, void* stream) {
  IMMOCO_REQUIRE(nM >= 0 && H > 0 && W > 0 && kall && col_group && kout, "kspace_select: bad argument");
  return launch_kspace_select(kall, col_group, nM, H, W, kout, as_stream(stream));
}

extern "C" int immoco_dc_loss(const float* k, const float* kin, int32_t H, int32_t W, float* loss, float* dk,
                              void* stream) {
  IMMOCO_REQUIRE(H > 0 && W > 0 && k && kin, "dc_loss: bad argument");
  return launch_dc_loss(k, kin, H, W, loss, dk, as_stream(stream));
}

extern "C" int immoco_ge_loss(const float* image, int32_t H, int32_t W, float weight, float* loss,
                              float* dimage, void* stream) {
  IMMOCO_REQUIRE(H > 0 && W > 0 && image, "ge_loss: bad argument");
  return launch_ge_loss(image, H, W, weight, nullptr, loss, dimage, as_stream(stream));
}

extern "C" int immoco_normalize_kspace(const float* k, int64_t n_complex, float target, float* out,
                                       float* scale_out, void* stream) {
  IMMOCO_REQUIRE(n_complex > 0 && k && out && scale_out, "normalize_kspace: bad argument");
  return launch_normalize(k, n_complex, target, out, scale_out, as_stream(stream));
}
