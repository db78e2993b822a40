// Multiresolution hash-grid encoding, forward and backward (tiny-cuda-nn grid.h
// semantics restated from SURVEY Appendix A; replaces the encoding half of
// tinycudann.NetworkWithInputEncoding, reference src/models/immoco.py:60-65,85,93).
//
// Layout: one thread per (point, level); blockIdx.y = level, so that all
// workgroups in flight touch one or two 4 MB level slices of the table, which
// stay resident in every XCD's 4 MB L2.  Coordinates either come from a generic
// [n, D] array (op-level API) or from per-axis lattices (solver: the reference
// always samples linspace(-1,1,.) lattices, immoco.py:48-53,72-80).
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"

namespace immoco {

int build_levels(const immoco_grid_cfg* cfg, Levels* out) {
  IMMOCO_REQUIRE(cfg != nullptr, "grid cfg is NULL");
  IMMOCO_REQUIRE(cfg->dims == 2 || cfg->dims == 3, "grid dims must be 2 or 3 (got %d)", cfg->dims);
  IMMOCO_REQUIRE(cfg->n_levels >= 1 && cfg->n_levels <= IMMOCO_MAX_LEVELS, "n_levels %d out of range",
                 cfg->n_levels);
  IMMOCO_REQUIRE(cfg->n_features == 2, "n_features_per_level must be 2 (got %d)", cfg->n_features);
  IMMOCO_REQUIRE(cfg->log2_hashmap_size >= 3 && cfg->log2_hashmap_size <= 28, "log2_hashmap_size %d",
                 cfg->log2_hashmap_size);
  IMMOCO_REQUIRE(cfg->base_resolution >= 1 && cfg->per_level_scale >= 1.0f, "bad resolution config");
  memset(out, 0, sizeof(*out));
  out->n_levels = cfg->n_levels;
  out->dims = cfg->dims;
  const float log2_pls = log2f(cfg->per_level_scale);
  uint64_t off = 0;
  for (int l = 0; l < cfg->n_levels; ++l) {
    // grid_scale(): exp2f(level * log2_per_level_scale) * base_resolution - 1
    float scale = exp2f((float)l * log2_pls) * (float)cfg->base_resolution - 1.0f;
    uint32_t res = (uint32_t)ceilf(scale) + 1u;  // grid_resolution()
    double dense = pow((double)res, cfg->dims);
    uint64_t n = dense > (double)0x7FFFFFFFu ? 0x7FFFFFFFull : (uint64_t)dense;
    n = (n + 7) / 8 * 8;
    uint64_t cap = 1ull << cfg->log2_hashmap_size;
    if (n > cap) n = cap;
    // grid_index(): stride after the dense walk, in uint32 like upstream - the product WRAPS.  For the
    // reference's config (base 16, scale 2: power-of-two resolutions) res*res wraps to exactly 0 from
    // res = 2^16 (level 12) on, so `hashmap_size < stride` is false there and levels 12-15 of both grids
    // use the dense wrapped index (c0 + c1*res) % size, not the prime hash (matches device grid_index()).
    uint32_t stride = 1;
    for (int d = 0; d < cfg->dims; ++d) {
      if (stride > n) break;
      stride *= res;
    }
    out->scale[l] = scale;
    out->res[l] = res;
    out->size[l] = (uint32_t)n;
    out->offset[l] = (uint32_t)off;
    if (n < stride) out->hashed |= 1u << l;
    if ((n & (n - 1)) == 0) out->pow2 |= 1u << l;
    off += n;
    IMMOCO_REQUIRE(off < 0x7FFFFFFFull, "hash table too large");
  }
  out->offset[cfg->n_levels] = (uint32_t)off;
  return IMMOCO_OK;
}

// ---------------------------------------------------------------------------
// Lattice layouts (checked by the launchers): D = 3: (m, row, col), strides (n1*n2, n2, 1); D = 2: dim 0 = column
// (stride 1), dim 1 = row (stride n0).  The point index is decomposed with fast_div (no 64-bit divisions).
template <int D, bool LAT>
__device__ __forceinline__ void load_coords(const float* __restrict__ coords, const Lattice& lat,
                                            int64_t p, float (&x)[D]) {
  if (LAT) {
    const uint32_t q = (uint32_t)p;
    if (D == 3) {
      const uint32_t i0 = fast_div(q, (uint32_t)lat.stride[0]);
      const uint32_t rem = q - i0 * (uint32_t)lat.stride[0];
      const uint32_t i1 = fast_div(rem, (uint32_t)lat.stride[1]);
      x[0] = lat.axis[0][i0];
      x[1] = lat.axis[1][i1];
      if (D > 2) x[2] = lat.axis[2][rem - i1 * (uint32_t)lat.stride[1]];
    } else {
      const uint32_t i1 = fast_div(q, (uint32_t)lat.stride[1]);
      x[0] = lat.axis[0][q - i1 * (uint32_t)lat.stride[1]];
      x[1] = lat.axis[1][i1];
    }
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) x[d] = coords[p * D + d];
  }
}

static int check_lattice(const Levels& lv, const Lattice& lat, int64_t n) {
  IMMOCO_REQUIRE(n < (1ll << 32), "lattice of %lld points: too large for 32-bit point indices", (long long)n);
  if (lv.dims == 3)
    IMMOCO_REQUIRE(lat.stride[2] == 1 && lat.stride[1] == lat.n[2] && lat.stride[0] == lat.n[1] * lat.n[2] &&
                       (int64_t)lat.n[0] * lat.n[1] * lat.n[2] == n,
                   "3-D lattice must be (m, row, col) row-major");
  else
    IMMOCO_REQUIRE(lat.stride[0] == 1 && lat.stride[1] == lat.n[0] && (int64_t)lat.n[0] * lat.n[1] == n,
                   "2-D lattice must be (x = col, y = row) row-major");
  return IMMOCO_OK;
}

// A/B switch (environment, read once): IMMOCO_ENC_STORE = plain | sc1 | nt (default) - see store_enc().
// Motion-grid forward at 320x320x10 on one box: plain 0.2667, sc1 0.2618, nt 0.2565 ms.
static int enc_store_sc1() {
  static const int v = [] {
    const char* e = immoco_diag_env("IMMOCO_ENC_STORE");
    return (e && strcmp(e, "plain") == 0) ? 0 : (e && strcmp(e, "sc1") == 0) ? 1 : 2;
  }();
  return v;
}

// mode 0: plain store; 1: agent-scope relaxed store = `global_store ... sc1` (write through, line dropped from L2);
// 2: non-temporal store (`nt`)
// mode & 16: the encoding is stored as packed halves (round to nearest even), one 4-byte word per (point, level) -
// tiny-cuda-nn's encoding output precision, the input precision of the fp16 MLP kernels (mlp_f16.hip); `dst` then
// counts 4-byte words
__device__ __forceinline__ void store_enc(float* dst, float2 e, int mode) {
  if (mode & 16) {
    typedef _Float16 h2s __attribute__((ext_vector_type(2)));
    typedef float f2s __attribute__((ext_vector_type(2)));
    const f2s x = {e.x, e.y};
    const h2s hv = __builtin_convertvector(x, h2s);
    uint32_t u;
    __builtin_memcpy(&u, &hv, 4);
    if ((mode & 15) == 2) __builtin_nontemporal_store(u, reinterpret_cast<uint32_t*>(dst));
    else *reinterpret_cast<uint32_t*>(dst) = u;
    return;
  }
  if (mode == 1) {
    union { float2 f; uint64_t u; } cv;
    cv.f = e;
    __hip_atomic_store(reinterpret_cast<uint64_t*>(dst), cv.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if (mode == 2) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 v;
    v.x = e.x;
    v.y = e.y;
    __builtin_nontemporal_store(v, reinterpret_cast<f32x2*>(dst));
  } else {
    *reinterpret_cast<float2*>(dst) = e;
  }
}

// TAB = float2: fp32 table; TAB = __half2: fp16 shadow of the table ("fp16 hash-grid features",
// BASELINE config 5 / what tiny-cuda-nn itself gathers); interpolation is accumulated in fp32 either way.
__device__ __forceinline__ float2 tab_to_f2(float2 v) { return v; }
__device__ __forceinline__ float2 tab_to_f2(__half2 v) { return __half22float2(v); }
// one load of an aligned slot pair (16 bytes of an fp32 table, 8 of an fp16 one)
__device__ __forceinline__ void load_pair(const float2* __restrict__ p, float2& lo, float2& hi) {
  const float4 q = *reinterpret_cast<const float4*>(p);
  lo = make_float2(q.x, q.y);
  hi = make_float2(q.z, q.w);
}
__device__ __forceinline__ void load_pair(const __half2* __restrict__ p, float2& lo, float2& hi) {
  const uint2 q = *reinterpret_cast<const uint2*>(p);
  union { uint32_t u; __half2 h; } a, b;
  a.u = q.x;
  b.u = q.y;
  lo = __half22float2(a.h);
  hi = __half22float2(b.h);
}

// One (point, level) of the encoding: gathers the 2^D corners and interpolates.
// The two corners that differ in dimension 0 sit in ONE aligned 16-byte pair of the table whenever
// their indices differ only in bit 0 (dense levels with an even cell; hashed levels with an even
// dim-0 cell, because dim 0 carries the prime 1): one dwordx4 load then serves both.  For the motion
// grid dim 0 is the motion group, constant over a wave, so the choice is wave-uniform: ~27 % fewer
// divergent loads on the TA-bound gather (rocprof: SQ_WAIT_INST_ANY 72 % of wave cycles before).
// Measured alternatives: `nt` loads 1.30 ms, `sc1` (L1-bypass) loads 0.49 ms, plain loads 0.43 ms,
// plain + pair merge 0.36 ms.
template <int D, typename TAB>
__device__ __forceinline__ float2 encode_point_level(const Levels& lv, int l, const float (&x)[D],
                                                     const TAB* __restrict__ table) {
  const float scale = lv.scale[l];
  const uint32_t size = lv.size[l], res = lv.res[l];
  const bool hashed = (lv.hashed >> l) & 1u, pow2 = (lv.pow2 >> l) & 1u;
  const TAB* __restrict__ tab = table + lv.offset[l];
  uint32_t cell[D];
  float fr[D];
#pragma unroll
  for (int d = 0; d < D; ++d) pos_fract(x[d], scale, cell[d], fr[d]);
  constexpr int NCORN = 1 << D;
  uint32_t idx[NCORN];
  float wgt[NCORN];
#pragma unroll
  for (int corner = 0; corner < NCORN; ++corner) {
    uint32_t c[D];
    float w = 1.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const bool hi = (corner >> d) & 1;
      c[d] = cell[d] + (hi ? 1u : 0u);
      w = mul_nc(w, hi ? fr[d] : sub_nc(1.0f, fr[d]));
    }
    idx[corner] = grid_index<D>(c, size, res, hashed, pow2);
    wgt[corner] = w;
  }
  // ALL gathers of the point are issued back to back, in straight-line code, before the first one is consumed.
  // (Round 1 decided "merged pair or two loads" per pair inside a per-lane branch: the compiler then has to
  // drain the loads (s_waitcnt vmcnt(0)) inside each branch, so a lane never had more than two gathers in
  // flight - found in the ISA in round 2.)  Whether the two dim-0 corners of EVERY pair form an aligned slot pair
  // is decided once per wave (a scalar branch): for the motion lattice dim 0 is the motion group, constant over
  // a wave.
  float2 v[NCORN];
  bool merge = D == 3;
#pragma unroll
  for (int pair = 0; pair < NCORN / 2; ++pair) merge = merge && ((idx[2 * pair] ^ idx[2 * pair + 1]) == 1u);
  if (D == 3 && __all(merge)) {
    float2 lo[NCORN / 2], hi[NCORN / 2];
#pragma unroll
    for (int pair = 0; pair < NCORN / 2; ++pair) load_pair(tab + (idx[2 * pair] & ~1u), lo[pair], hi[pair]);
#pragma unroll
    for (int pair = 0; pair < NCORN / 2; ++pair) {
      const bool odd = idx[2 * pair] & 1u;
      v[2 * pair] = odd ? hi[pair] : lo[pair];
      v[2 * pair + 1] = odd ? lo[pair] : hi[pair];
    }
  } else {
    TAB t[NCORN];
#pragma unroll
    for (int corner = 0; corner < NCORN; ++corner) t[corner] = tab[idx[corner]];
#pragma unroll
    for (int corner = 0; corner < NCORN; ++corner) v[corner] = tab_to_f2(t[corner]);
  }
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int corner = 0; corner < (1 << D); ++corner) {
    // separate mul/add (no contraction): bit-identical to the fp32 oracle
    const float t0 = mul_nc(v[corner].x, wgt[corner]), t1 = mul_nc(v[corner].y, wgt[corner]);
    a0 = corner == 0 ? t0 : add_nc(a0, t0);
    a1 = corner == 0 ? t1 : add_nc(a1, t1);
  }
  return make_float2(a0, a1);
}

// Measured dead end (round 2): for an ODD wave-uniform dim-0 cell the two dim-0 corners are not an aligned
// pair, but they share a 128-byte line; putting them into adjacent lanes of ONE load (lane 2i low, 2i+1 high
// corner, two rounds of 32 points, products exchanged by DPP, bit-identical) halves the distinct lines per
// instruction - and changed nothing (0.278 -> 0.291 ms): the second corner's load already merges into the
// first one's outstanding L1 miss, so a pair costs one L2->L1 fill either way.  The forward is bound by 4 line
// fills per (point, hashed level).
template <int D, bool LAT, typename TAB = float2>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(Levels lv, const float* __restrict__ coords,
                                                           Lattice lat, int64_t n,
                                                           const TAB* __restrict__ table,
                                                           float* __restrict__ enc, int64_t ps, int64_t ls,
                                                           int store_sc1) {
  const int l = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  float x[D];
  load_coords<D, LAT>(coords, lat, p, x);
  const float2 e = encode_point_level<D, TAB>(lv, l, x, table);
  // The encoding streams out (131 MB per launch at 320x320x10) while the gathers want the 4 MB level slice of
  // the table to STAY in the XCD's 4 MB L2: a non-temporal store (`global_store ... nt`) does not keep the line
  // (round 1 used the write-through `sc1` flavour of MI355X_MICROARCH.md for the same reason; nt is 2 % better).
  store_enc(enc + p * ps + (int64_t)l * ls, e, store_sc1);
}

// Large 3-D lattices: K points per thread, three phases - indices and weights of all K points, then ALL their
// gathers back to back, then the interpolation.  Ablations of the one-point kernel at 320x320x10 (0.261 ms): index
// arithmetic alone 0.070 ms, no store 0.239, every level gathering from ONE 4 MB slice (no compulsory misses)
// 0.242, both 0.216 - the pieces add up instead of overlapping, because a wave lives for one (point, level):
// compute, load, wait, combine, exit.  Same arithmetic per point: bit-identical.
// (Keeping the wave alive instead - a sequential loop over 2 / 4 / 8 blocks of points per thread, same registers -
// is SLOWER: 0.274 / 0.317 / 0.434 ms.  Stores and loads share vmcnt on gfx9 and return in order, so the next
// item's gathers wait behind the write-through store of the previous one; fresh waves do not.)
// Occupancy: the one-point kernel (42 registers, 8 waves/SIMD) only loses when workgroups per CU are capped with
// unused LDS - 8 per CU 0.2555 ms, 6: 0.263, 5: 0.270, 4: 0.2825, 3: 0.301 - unlike the encode backward (csr.hip).
template <typename TAB, int K>
__global__ __launch_bounds__(256) void hashgrid_fwd_lat3_kernel(Levels lv, Lattice lat, int64_t n,
                                                                const TAB* __restrict__ table,
                                                                float* __restrict__ enc, int64_t ps, int64_t ls,
                                                                int store_sc1) {
  constexpr int NCORN = 8;
  const int l = blockIdx.y;
  const float scale = lv.scale[l];
  const uint32_t size = lv.size[l], res = lv.res[l];
  const bool hashed = (lv.hashed >> l) & 1u, pow2 = (lv.pow2 >> l) & 1u;
  const TAB* __restrict__ tab = table + lv.offset[l];
  int64_t p[K];
  uint32_t idx[K][NCORN];
  float wgt[K][NCORN];
  bool merge = true;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    p[k] = ((int64_t)blockIdx.x * K + k) * 256 + threadIdx.x;
    const int64_t pc = p[k] < n ? p[k] : n - 1;   // tail lanes recompute the last point (never stored)
    float x[3];
    load_coords<3, true>(nullptr, lat, pc, x);
    uint32_t cell[3];
    float fr[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) pos_fract(x[d], scale, cell[d], fr[d]);
#pragma unroll
    for (int corner = 0; corner < NCORN; ++corner) {
      uint32_t c[3];
      float w = 1.0f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const bool hi = (corner >> d) & 1;
        c[d] = cell[d] + (hi ? 1u : 0u);
        w = mul_nc(w, hi ? fr[d] : sub_nc(1.0f, fr[d]));
      }
      idx[k][corner] = grid_index<3>(c, size, res, hashed, pow2);
      wgt[k][corner] = w;
    }
#pragma unroll
    for (int pair = 0; pair < NCORN / 2; ++pair) merge = merge && ((idx[k][2 * pair] ^ idx[k][2 * pair + 1]) == 1u);
  }
  float2 v[K][NCORN];
  if (__all(merge)) {   // every dim-0 corner pair of every point of the wave is an aligned slot pair: 16-byte loads
    float2 lo[K][NCORN / 2], hi[K][NCORN / 2];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int pair = 0; pair < NCORN / 2; ++pair) load_pair(tab + (idx[k][2 * pair] & ~1u), lo[k][pair], hi[k][pair]);
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int pair = 0; pair < NCORN / 2; ++pair) {
        const bool odd = idx[k][2 * pair] & 1u;
        v[k][2 * pair] = odd ? hi[k][pair] : lo[k][pair];
        v[k][2 * pair + 1] = odd ? lo[k][pair] : hi[k][pair];
      }
  } else {
    TAB t[K][NCORN];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int corner = 0; corner < NCORN; ++corner) t[k][corner] = tab[idx[k][corner]];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int corner = 0; corner < NCORN; ++corner) v[k][corner] = tab_to_f2(t[k][corner]);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int corner = 0; corner < NCORN; ++corner) {
      const float t0 = mul_nc(v[k][corner].x, wgt[k][corner]), t1 = mul_nc(v[k][corner].y, wgt[k][corner]);
      a0 = corner == 0 ? t0 : add_nc(a0, t0);
      a1 = corner == 0 ? t1 : add_nc(a1, t1);
    }
    if (p[k] < n) {
      float* dst = enc + p[k] * ps + (int64_t)l * ls;
      store_enc(dst, make_float2(a0, a1), store_sc1);
    }
  }
}

// Round 4: a software-PIPELINED variant of the one-point kernel.  A thread walks K points of one level (256 apart); the
// gathers of point k + 1 are issued BEFORE point k's are consumed and its encoding is stored, so a wave always has one
// point's gathers in flight while it interpolates and stores the previous one - the round-2 loops (compute, load, wait,
// combine, store, next) drained the memory pipe at every store (vector-memory results return in issue order) and were
// slower than fresh waves.  Whether the dim-0 corner pairs are aligned slot pairs (one 16-byte load per pair) is
// decided once per WORKGROUP from its first and last point (dim 0 is the motion group: constant over the workgroup's
// contiguous points unless it straddles two groups), which keeps each path straight-line code.  Same arithmetic per
// point: bit-identical.
// MEASURED (MI355X, 320x320x10, isolated): one-point kernel 0.2528 ms; K = 2: 0.2528, K = 4: 0.2697, K = 8: 0.3393 ms (the ISA
// does keep point k + 1's eight gathers in flight while point k is consumed: s_waitcnt vmcnt(15) ... (8)).  Longer-lived
// waves lose even when pipelined - a dead end like the round-2 loops; the kernel stays behind IMMOCO_FWD_PIPE in the
// diagnostics build only.
// (Specialised for power-of-two level sizes - the reference's configuration - and lattices below 2^24 points, so that
// the per-point code has no branches at all: scalar branches at every corner would put a wait at every join.)
__device__ __forceinline__ uint32_t div_small(uint32_t n, uint32_t d, float inv_d) {   // n / d for n, d < 2^24
  uint32_t q = (uint32_t)((float)n * inv_d);
  q -= (q * d > n) ? 1u : 0u;
  q += ((q + 1u) * d <= n) ? 1u : 0u;
  return q;
}

// The lattice axes are staged in LDS (`ax`: axis 0, then 1, then 2): a coordinate load from global memory would share
// the vector-memory counter with the gathers - its wait would drain the previous point's gathers and undo the pipeline.
template <int K, bool MERGED, bool HASHED, bool HALF>
__device__ __forceinline__ void fwd_pipe_body(const Levels& lv, const Lattice& lat, const float* ax, int64_t n, int l,
                                              const float2* __restrict__ tab, float* __restrict__ enc, int64_t ps,
                                              int64_t ls, int64_t p0) {
  constexpr int NC = 8;
  const float scale = lv.scale[l];
  const uint32_t mask = lv.size[l] - 1u, res = lv.res[l];
  // dense levels: idx = c0 + c1 res + c2 res^2 with the uint32 stride walk of grid_index() (a stride that has wrapped
  // to 0 or exceeds the level size drops the remaining dimensions)
  uint32_t st1 = 0u, st2 = 0u;
  {
    uint32_t stride = 1u;
    if (stride <= lv.size[l]) stride *= res;               // after dim 0
    if (stride <= lv.size[l]) { st1 = stride; stride *= res; }
    if (stride <= lv.size[l]) st2 = stride;
  }
  const uint32_t s0 = (uint32_t)lat.stride[0], s1 = (uint32_t)lat.stride[1];
  const float inv_s0 = 1.0f / (float)s0, inv_s1 = 1.0f / (float)s1;
  uint32_t idx[2][NC];
  float wgt[2][NC];
  float4 q4[2][NC / 2];   // MERGED: one aligned slot pair per dim-0 corner pair
  float2 q2[2][NC];       // else: one slot per corner
  auto prepare = [&](int buf, int64_t p) {
    const uint32_t q = (uint32_t)(p < n ? p : n - 1);   // tail lanes recompute the last point (never stored)
    const uint32_t i0 = div_small(q, s0, inv_s0), rem = q - i0 * s0, i1 = div_small(rem, s1, inv_s1);
    const float x[3] = {ax[i0], ax[lat.n[0] + i1], ax[lat.n[0] + lat.n[1] + (rem - i1 * s1)]};
    uint32_t cell[3];
    float fr[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) pos_fract(x[d], scale, cell[d], fr[d]);
#pragma unroll
    for (int corner = 0; corner < NC; ++corner) {
      uint32_t c[3];
      float w = 1.0f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const bool hi = (corner >> d) & 1;
        c[d] = cell[d] + (hi ? 1u : 0u);
        w = mul_nc(w, hi ? fr[d] : sub_nc(1.0f, fr[d]));
      }
      const uint32_t raw = HASHED ? (c[0] ^ (c[1] * IMMOCO_PRIME1) ^ (c[2] * IMMOCO_PRIME2)) : (c[0] + c[1] * st1 + c[2] * st2);
      idx[buf][corner] = raw & mask;
      wgt[buf][corner] = w;
    }
    if (MERGED) {
#pragma unroll
      for (int pair = 0; pair < NC / 2; ++pair)
        q4[buf][pair] = *reinterpret_cast<const float4*>(tab + (idx[buf][2 * pair] & ~1u));
    } else {
#pragma unroll
      for (int corner = 0; corner < NC; ++corner) q2[buf][corner] = tab[idx[buf][corner]];
    }
  };
  auto finish = [&](int buf, int64_t p) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int corner = 0; corner < NC; ++corner) {
      float2 v;
      if (MERGED) {
        const float4 q = q4[buf][corner >> 1];
        const bool odd = idx[buf][corner & ~1] & 1u;          // the pair's low corner sits in the odd slot
        const bool take_hi = ((corner & 1) != 0) != odd;       // low corner: lo unless odd; high corner: hi unless odd
        v = take_hi ? make_float2(q.z, q.w) : make_float2(q.x, q.y);
      } else {
        v = q2[buf][corner];
      }
      const float t0 = mul_nc(v.x, wgt[buf][corner]), t1 = mul_nc(v.y, wgt[buf][corner]);
      a0 = corner == 0 ? t0 : add_nc(a0, t0);
      a1 = corner == 0 ? t1 : add_nc(a1, t1);
    }
    if (p < n) store_enc(enc + p * ps + (int64_t)l * ls, make_float2(a0, a1), HALF ? (16 | 2) : 2);   // non-temporal
  };
  prepare(0, p0);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (k + 1 < K) prepare((k + 1) & 1, p0 + (int64_t)(k + 1) * 256);
    finish(k & 1, p0 + (int64_t)k * 256);
  }
}

template <int K, bool HALF>
__global__ __launch_bounds__(256) void hashgrid_fwd_pipe_kernel(Levels lv, Lattice lat, int64_t n,
                                                                const float2* __restrict__ table,
                                                                float* __restrict__ enc, int64_t ps, int64_t ls) {
  extern __shared__ float ax[];   // the three lattice axes
  const int l = blockIdx.y;
  const int n01 = lat.n[0] + lat.n[1], n012 = n01 + lat.n[2];
  for (int i = threadIdx.x; i < n012; i += 256)
    ax[i] = i < lat.n[0] ? lat.axis[0][i] : (i < n01 ? lat.axis[1][i - lat.n[0]] : lat.axis[2][i - n01]);
  __syncthreads();
  const float2* __restrict__ tab = table + lv.offset[l];
  const int64_t first = (int64_t)blockIdx.x * K * 256, p0 = first + threadIdx.x;
  const bool hashed = (lv.hashed >> l) & 1u;
  // aligned-pair test for the workgroup's first and last point: the two dim-0 corners differ in bit 0 only <=> the
  // dim-0 cell is even (dense levels: stride 1 for dim 0; hashed levels: prime 1 for dim 0; sizes are powers of two)
  auto even_cell = [&](int64_t p) {
    const uint32_t q = (uint32_t)(p < n ? p : n - 1);
    const uint32_t i0 = q / (uint32_t)lat.stride[0];
    uint32_t cell;
    float fr;
    pos_fract(ax[i0], lv.scale[l], cell, fr);
    return (cell & 1u) == 0u;
  };
  const bool merged = lv.size[l] >= 2u && even_cell(first) && even_cell(first + (int64_t)K * 256 - 1);   // workgroup-uniform
  if (merged) {
    if (hashed) fwd_pipe_body<K, true, true, HALF>(lv, lat, ax, n, l, tab, enc, ps, ls, p0);
    else fwd_pipe_body<K, true, false, HALF>(lv, lat, ax, n, l, tab, enc, ps, ls, p0);
  } else {
    if (hashed) fwd_pipe_body<K, false, true, HALF>(lv, lat, ax, n, l, tab, enc, ps, ls, p0);
    else fwd_pipe_body<K, false, false, HALF>(lv, lat, ax, n, l, tab, enc, ps, ls, p0);
  }
}

// Points per thread in the lattice kernel.  Measured (MI355X): 320x320x10 (1.0 M points): K = 1 0.262, K = 2 0.276,
// K = 4 0.313 ms - the occupancy lost to the extra registers (8 -> 5 -> 3 waves/SIMD) costs more than the overlap
// brings; 640x640x20 (8.2 M points): K = 1 1.77, K = 2 1.62 ms.  So: 2 from 4 M points on, else the one-point
// kernel.  A/B switch (environment, read once): IMMOCO_FWD_K = 1 | 2 | 4.
static int fwd_points_per_thread(int64_t n) {
  static const int forced = [] {
    const char* e = immoco_diag_env("IMMOCO_FWD_K");
    const int k = e ? atoi(e) : 0;
    return (k == 1 || k == 2 || k == 4) ? k : 0;
  }();
  return forced ? forced : (n >= (4ll << 20) ? 2 : 1);
}

template <typename TAB>
static bool launch_fwd_lat3(const Levels& lv, const Lattice& lat, int64_t n, const TAB* t, float* enc, int64_t ps,
                            int64_t ls, hipStream_t st, int mode) {
  const int K = fwd_points_per_thread(n);
  if (lv.dims != 3 || K == 1 || n < 256 * 64) return false;
  dim3 grid((unsigned)cdiv(n, 256 * K), lv.n_levels);
  if (K == 2) hashgrid_fwd_lat3_kernel<TAB, 2><<<grid, 256, 0, st>>>(lv, lat, n, t, enc, ps, ls, mode);
  else hashgrid_fwd_lat3_kernel<TAB, 4><<<grid, 256, 0, st>>>(lv, lat, n, t, enc, ps, ls, mode);
  return true;
}

// Measured dead end (round 2): an XCD-scheduled launch - workgroup w runs on XCD w % 8, XCD x encodes ALL points
// of "its" hashed level (the first eight hashed levels, one each, so that a 4 MB table slice is fetched by ONE L2
// instead of eight) plus an eighth of the points of every other level - cut the compulsory L2 misses (3.6 M of
// the 43.2 M L1->L2 requests per launch, rocprofv3 TCP_TCC_READ_REQ / TCC_MISS) and changed nothing: 0.279 vs
// 0.274 ms.  Neither L2 misses nor distinct lines per instruction (lane-paired corners, above) are what the
// forward waits for; it runs at 157 G L2 requests/s against the probe's 259.
// Measured dead end: a "level sweep" launch (one workgroup keeps 512 points and walks the 16 levels, meant
// to keep a single 4 MB level slice live per XCD L2) ran at 0.81 ms instead of 0.36 ms: workgroups drift
// apart and soon touch 3-4 level slices at once, and immoco_probe_gather shows the price - random 16-byte
// gathers sustain 259 G requests/s inside a 4 MB footprint, 120 G/s inside 8 MB, 80 G/s inside 16 MB.
// The (points, level) grid below keeps at most two slices live.

// v1 backward: one float atomic per (corner, feature).
template <int D, bool LAT>
__global__ __launch_bounds__(256) void hashgrid_bwd_atomic_kernel(Levels lv, const float* __restrict__ coords,
                                                                  Lattice lat, int64_t n,
                                                                  const float* __restrict__ denc, int64_t ps,
                                                                  int64_t ls, float* __restrict__ dtable) {
  const int l = blockIdx.y;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const float2 g = *reinterpret_cast<const float2*>(denc + p * ps + (int64_t)l * ls);
  if (g.x == 0.f && g.y == 0.f) return;
  float x[D];
  load_coords<D, LAT>(coords, lat, p, x);
  const float scale = lv.scale[l];
  const uint32_t size = lv.size[l], res = lv.res[l];
  const bool hashed = (lv.hashed >> l) & 1u, pow2 = (lv.pow2 >> l) & 1u;
  float* __restrict__ tab = dtable + (size_t)lv.offset[l] * 2;
  uint32_t cell[D];
  float fr[D];
#pragma unroll
  for (int d = 0; d < D; ++d) pos_fract(x[d], scale, cell[d], fr[d]);
#pragma unroll
  for (int corner = 0; corner < (1 << D); ++corner) {
    uint32_t c[D];
    float w = 1.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const bool hi = (corner >> d) & 1;
      c[d] = cell[d] + (hi ? 1u : 0u);
      w *= hi ? fr[d] : 1.0f - fr[d];
    }
    const uint32_t idx = grid_index<D>(c, size, res, hashed, pow2);
    unsafeAtomicAdd(tab + (size_t)idx * 2, w * g.x);
    unsafeAtomicAdd(tab + (size_t)idx * 2 + 1, w * g.y);
  }
}

int launch_hashgrid_fwd(const Levels& lv, const float* coords, const Lattice* lat, int64_t n,
                        const float* table, float* enc, int64_t ps, int64_t ls, hipStream_t st, bool half_out) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(half_out || ((ps % 2) == 0 && (ls % 2) == 0), "encoding strides must be even (float2 stores)");
  dim3 grid((unsigned)cdiv(n, 256), lv.n_levels), block(256);
  Lattice L{};
  if (lat) {
    L = *lat;
    int rc = check_lattice(lv, L, n);
    if (rc) return rc;
  }
  const int mode = enc_store_sc1() | (half_out ? 16 : 0);
  const float2* t = reinterpret_cast<const float2*>(table);
  // software-pipelined walk over K points per thread (3-D lattices; A/B switch IMMOCO_FWD_PIPE = 0 | 2 | 4 | 8)
  static const int pipe_env = [] { const char* e = immoco_diag_env("IMMOCO_FWD_PIPE"); return e ? atoi(e) : -1; }();
  const int pipe = pipe_env >= 0 ? pipe_env : 0;
  const bool all_pow2 = lv.pow2 == ((lv.n_levels >= 32 ? 0u : (1u << lv.n_levels)) - 1u);
  if (lat && lv.dims == 3 && (pipe == 2 || pipe == 4 || pipe == 8) && n >= 256 * 64 && n < (1 << 24) && all_pow2) {
    dim3 g((unsigned)cdiv(n, 256 * pipe), lv.n_levels);
    const size_t sm = (size_t)(L.n[0] + L.n[1] + L.n[2]) * sizeof(float);
    IMMOCO_REQUIRE(sm <= 48 * 1024, "hashgrid_fwd: lattice axes too long for the LDS copy");
#define IMMOCO_PIPE(KK)                                                                                    \
  do {                                                                                                     \
    if (half_out) hashgrid_fwd_pipe_kernel<KK, true><<<g, 256, sm, st>>>(lv, L, n, t, enc, ps, ls);        \
    else hashgrid_fwd_pipe_kernel<KK, false><<<g, 256, sm, st>>>(lv, L, n, t, enc, ps, ls);                \
  } while (0)
    if (pipe == 2) IMMOCO_PIPE(2);
    else if (pipe == 4) IMMOCO_PIPE(4);
    else IMMOCO_PIPE(8);
#undef IMMOCO_PIPE
    IMMOCO_LAUNCH_CHECK();
    return IMMOCO_OK;
  }
  if (lat && launch_fwd_lat3<float2>(lv, L, n, t, enc, ps, ls, st, mode)) {
    IMMOCO_LAUNCH_CHECK();
    return IMMOCO_OK;
  }
  if (lv.dims == 2) {
    if (lat) hashgrid_fwd_kernel<2, true><<<grid, block, 0, st>>>(lv, coords, L, n, t, enc, ps, ls, mode);
    else hashgrid_fwd_kernel<2, false><<<grid, block, 0, st>>>(lv, coords, L, n, t, enc, ps, ls, mode);
  } else {
    if (lat) hashgrid_fwd_kernel<3, true><<<grid, block, 0, st>>>(lv, coords, L, n, t, enc, ps, ls, mode);
    else hashgrid_fwd_kernel<3, false><<<grid, block, 0, st>>>(lv, coords, L, n, t, enc, ps, ls, mode);
  }
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// forward from the fp16 shadow table (lattice mode only: the solver's path)
int launch_hashgrid_fwd_half(const Levels& lv, const Lattice& lat, int64_t n, const void* table_half2, float* enc,
                             int64_t ps, int64_t ls, hipStream_t st, bool half_out) {
  if (n == 0) return IMMOCO_OK;
  {
    int rc = check_lattice(lv, lat, n);
    if (rc) return rc;
  }
  dim3 grid((unsigned)cdiv(n, 256), lv.n_levels), block(256);
  const int mode = enc_store_sc1() | (half_out ? 16 : 0);
  const __half2* t = reinterpret_cast<const __half2*>(table_half2);
  if (launch_fwd_lat3<__half2>(lv, lat, n, t, enc, ps, ls, st, mode)) {
    IMMOCO_LAUNCH_CHECK();
    return IMMOCO_OK;
  }
  if (lv.dims == 2) hashgrid_fwd_kernel<2, true, __half2><<<grid, block, 0, st>>>(lv, nullptr, lat, n, t, enc, ps, ls, mode);
  else hashgrid_fwd_kernel<3, true, __half2><<<grid, block, 0, st>>>(lv, nullptr, lat, n, t, enc, ps, ls, mode);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

__global__ __launch_bounds__(256) void f32_to_half_kernel(const float* __restrict__ in, __half* __restrict__ out,
                                                          int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = __float2half_rn(in[i]);
}

int launch_f32_to_half(const float* in, void* out_half, int64_t n, hipStream_t st) {
  if (n == 0) return IMMOCO_OK;
  f32_to_half_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>(in, reinterpret_cast<__half*>(out_half), n);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

int launch_hashgrid_bwd(const Levels& lv, const float* coords, const Lattice* lat, int64_t n,
                        const float* denc, int64_t ps, int64_t ls, float* dtable, hipStream_t st) {
  if (n == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE((ps % 2) == 0 && (ls % 2) == 0, "encoding strides must be even (float2 loads)");
  dim3 grid((unsigned)cdiv(n, 256), lv.n_levels), block(256);
  Lattice L{};
  if (lat) L = *lat;
  if (lv.dims == 2) {
    if (lat) hashgrid_bwd_atomic_kernel<2, true><<<grid, block, 0, st>>>(lv, coords, L, n, denc, ps, ls, dtable);
    else hashgrid_bwd_atomic_kernel<2, false><<<grid, block, 0, st>>>(lv, coords, L, n, denc, ps, ls, dtable);
  } else {
    if (lat) hashgrid_bwd_atomic_kernel<3, true><<<grid, block, 0, st>>>(lv, coords, L, n, denc, ps, ls, dtable);
    else hashgrid_bwd_atomic_kernel<3, false><<<grid, block, 0, st>>>(lv, coords, L, n, denc, ps, ls, dtable);
  }
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

// ---------------------------------------------------------------------------
// parameter init (shared bit-exactly with oracle.uniform_init)
__global__ __launch_bounds__(256) void init_uniform_kernel(float* __restrict__ out, int64_t n, uint32_t key,
                                                           float lo, float span) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t h = pcg_hash((uint32_t)i + key);
  const float u = (float)(h >> 8) * 5.9604644775390625e-08f;  // 2^-24
  out[i] = add_nc(lo, mul_nc(u, span));
}

int launch_init_uniform(float* out, int64_t n, uint32_t seed, uint32_t stream_id, float lo, float hi,
                        hipStream_t st) {
  if (n == 0) return IMMOCO_OK;
  const uint32_t key = pcg_hash(seed ^ (stream_id * 0x9E3779B9u));
  init_uniform_kernel<<<(unsigned)cdiv(n, 256), 256, 0, st>>>(out, n, key, lo, hi - lo);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

}  // namespace immoco

using namespace immoco;

extern "C" int immoco_grid_geometry_query(const immoco_grid_cfg* cfg, immoco_grid_geometry* out) {
  Levels lv;
  int rc = build_levels(cfg, &lv);
  if (rc) return rc;
  IMMOCO_REQUIRE(out != nullptr, "geometry out is NULL");
  memset(out, 0, sizeof(*out));
  for (int l = 0; l < lv.n_levels; ++l) {
    out->offset[l] = lv.offset[l];
    out->resolution[l] = lv.res[l];
    out->size[l] = lv.size[l];
    out->scale[l] = lv.scale[l];
    out->hashed[l] = (lv.hashed >> l) & 1u;
  }
  out->offset[lv.n_levels] = lv.offset[lv.n_levels];
  return IMMOCO_OK;
}

extern "C" int immoco_hashgrid_fwd(const immoco_grid_cfg* cfg, const float* coords, int64_t n,
                                   const float* table, float* enc, int64_t enc_point_stride,
                                   int64_t enc_level_stride, void* stream) {
  Levels lv;
  int rc = build_levels(cfg, &lv);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (coords && table && enc)), "hashgrid_fwd: NULL buffer");
  return launch_hashgrid_fwd(lv, coords, nullptr, n, table, enc, enc_point_stride, enc_level_stride,
                             as_stream(stream));
}

extern "C" int immoco_hashgrid_fwd_lattice(const immoco_grid_cfg* cfg, int32_t nM, int32_t H, int32_t W,
                                           const float* ax0, const float* ax1, const float* ax2, const float* table,
                                           float* enc, int64_t enc_point_stride, int64_t enc_level_stride,
                                           void* stream) {
  Levels lv;
  int rc = build_levels(cfg, &lv);
  if (rc) return rc;
  IMMOCO_REQUIRE(nM >= 1 && H >= 1 && W >= 1, "hashgrid_fwd_lattice: bad lattice %dx%dx%d", nM, H, W);
  IMMOCO_REQUIRE(ax0 && ax1 && (lv.dims == 2 || ax2) && table && enc, "hashgrid_fwd_lattice: NULL buffer");
  IMMOCO_REQUIRE(lv.dims == 3 || nM == 1, "hashgrid_fwd_lattice: a 2-D grid takes nM = 1");
  Lattice lat{};
  if (lv.dims == 3) {  // (m, row, col)
    lat.axis[0] = ax0; lat.n[0] = nM; lat.stride[0] = H * W;
    lat.axis[1] = ax1; lat.n[1] = H;  lat.stride[1] = W;
    lat.axis[2] = ax2; lat.n[2] = W;  lat.stride[2] = 1;
  } else {             // (x = col, y = row)
    lat.axis[0] = ax0; lat.n[0] = W; lat.stride[0] = 1;
    lat.axis[1] = ax1; lat.n[1] = H; lat.stride[1] = W;
    lat.axis[2] = ax0; lat.n[2] = 1; lat.stride[2] = 1;
  }
  return launch_hashgrid_fwd(lv, nullptr, &lat, (int64_t)nM * H * W, table, enc, enc_point_stride,
                             enc_level_stride, as_stream(stream));
}

extern "C" int immoco_hashgrid_fwd_f16(const immoco_grid_cfg* cfg, const float* coords, int64_t n,
                                       const void* table_f16, float* enc, int64_t enc_point_stride,
                                       int64_t enc_level_stride, void* stream) {
  Levels lv;
  int rc = build_levels(cfg, &lv);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (coords && table_f16 && enc)), "hashgrid_fwd_f16: NULL buffer");
  IMMOCO_REQUIRE((enc_point_stride % 2) == 0 && (enc_level_stride % 2) == 0,
                 "encoding strides must be even (float2 stores)");
  if (n == 0) return IMMOCO_OK;
  dim3 grid((unsigned)cdiv(n, 256), lv.n_levels), block(256);
  const __half2* t = reinterpret_cast<const __half2*>(table_f16);
  Lattice none{};
  if (lv.dims == 2)
    hashgrid_fwd_kernel<2, false, __half2><<<grid, block, 0, as_stream(stream)>>>(lv, coords, none, n, t, enc,
                                                                                 enc_point_stride, enc_level_stride, enc_store_sc1());
  else
    hashgrid_fwd_kernel<3, false, __half2><<<grid, block, 0, as_stream(stream)>>>(lv, coords, none, n, t, enc,
                                                                                 enc_point_stride, enc_level_stride, enc_store_sc1());
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

extern "C" int immoco_hashgrid_bwd(const immoco_grid_cfg* cfg, const float* coords, int64_t n,
                                   const float* denc, int64_t enc_point_stride, int64_t enc_level_stride,
                                   float* dtable, void* stream) {
  Levels lv;
  int rc = build_levels(cfg, &lv);
  if (rc) return rc;
  IMMOCO_REQUIRE(n >= 0 && (n == 0 || (coords && denc && dtable)), "hashgrid_bwd: NULL buffer");
  return launch_hashgrid_bwd(lv, coords, nullptr, n, denc, enc_point_stride, enc_level_stride, dtable,
                             as_stream(stream));
}

extern "C" int immoco_init_params(const immoco_grid_cfg* grid, const immoco_mlp_cfg* mlp, uint32_t seed,
                                  float* params, void* stream) {
  Levels lv;
  int rc = build_levels(grid, &lv);
  if (rc) return rc;
  IMMOCO_REQUIRE(mlp && params, "init_params: NULL argument");
  const int64_t n_w1 = (int64_t)mlp->n_hidden * mlp->n_in;
  const int64_t n_w2 = (int64_t)mlp->n_out_padded * mlp->n_hidden;
  const int64_t n_tab = (int64_t)lv.offset[lv.n_levels] * 2;
  const float b1 = (float)sqrt(6.0 / (double)(mlp->n_in + mlp->n_hidden));
  const float b2 = (float)sqrt(6.0 / (double)(mlp->n_hidden + mlp->n_out_padded));
  hipStream_t st = as_stream(stream);
  if ((rc = launch_init_uniform(params, n_w1, seed, 1, -b1, b1, st))) return rc;
  if ((rc = launch_init_uniform(params + n_w1, n_w2, seed, 2, -b2, b2, st))) return rc;
  return launch_init_uniform(params + n_w1 + n_w2, n_tab, seed, 3, -1e-4f, 1e-4f, st);
}
