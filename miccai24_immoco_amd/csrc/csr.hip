// Hash-grid backward WITHOUT a global atomic scatter.
//
// The reference always queries its INRs on the same lattice (immoco.py:48-53,72-80),
// so the (point, corner) -> table-slot map of tiny-cuda-nn's encoding is a constant
// of the slice shape.  Measured on MI355X the naive backward (one float atomic per
// corner and feature, 2 x 131 M per iteration at 320x320 / 10 groups) runs at the
// memory-side atomic rate and takes 12.9 ms per iteration (77 % of the step).
//
// Plan (once per solver): a transposed index ("CSR by slot"): for every table slot the list of its
// contributions, slot-sorted, built on the GPU (count -> hipCUB exclusive scan -> fill), plus a
// host-built list of work items.  An entry is 8 bytes: {point (relative to its part) << 11 | slot
// (relative to its 2048-slot block), interpolation weight fp32}; a "twin" entry (sign bit of the weight
// set) stands for BOTH dim-0 corners of its point whenever they fall into one slot block: it is filed under the
// low corner's slot and carries the weight of the other dimensions; the dim-0 fraction and the partner slot
// (slot + 1 on dense levels, slot ^ (c0 ^ (c0+1)) on hashed ones) come from per-level tables.  Earlier versions stored 4 bytes
// (corner | col | row | m) and recomputed slot and weight per entry from LDS axis tables: an
// ablation on MI355X showed that this decode/hash/weight ALU work was 70 % of the kernel
// (0.63 ms with, 0.45 ms without the gathers), so the plan now pays 4 more bytes of (otherwise
// idle) HBM stream per entry to remove it.
//
// Backward (every iteration): one workgroup per work item; every thread owns 16 CONSECUTIVE
// entries (eight coalesced 16-byte loads from a transposed chunk layout, next chunk prefetched),
// gathers dL/denc (8 B, level slice
// L2/MALL resident), sums equal-slot runs in registers and issues one LDS atomic per run into a
// 2048-slot LDS tile; the tile is written out with plain stores (items that share a slot block -
// only on the coarse dense levels - flush with contiguous atomics instead).
//
// L2 locality: the gather's working set is one level slice of dL/denc (8 B x points = 8 MB at
// 320x320x10), twice an XCD's 4 MB L2.  The plan splits the points into `n_parts` contiguous
// ranges, sorts entries by (level, part, slot) and orders the work items so that workgroup index i
// (which lands on XCD i % 8) only touches part (i % 8) % n_parts: every XCD keeps a 8/n_parts MB
// slice hot.  Each part writes its own partial gradient table; the Adam kernel sums them.
#include <hipcub/hipcub.hpp>

#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#include "kernels.hpp"

namespace immoco {

constexpr int SLOT_BITS = 11;
constexpr int SLOTS_PER_ITEM = 1 << SLOT_BITS;  // 2048: aligned slot blocks
constexpr int ENTRIES_PER_ITEM = 32768;

// lattice index of dimension d for point fields (m, r, c):
//   D == 3 (motion INR, make_grids order):  dim0 = m, dim1 = row, dim2 = col
//   D == 2 (image INR, identy_grid order):  dim0 = col (x), dim1 = row (y)
template <int D>
__device__ __forceinline__ void entry_dims(uint32_t m, uint32_t r, uint32_t c, uint32_t (&i)[D]) {
  if (D == 3) {
    i[0] = m;
    i[1] = r;
    if (D > 2) i[2] = c;
  } else {
    i[0] = c;
    i[1] = r;
  }
}

template <int D>
struct AxisPtrs {
  const float* a[3];
  int32_t n[3];
};

// ---- plan build ------------------------------------------------------------------
// one thread per (point, level): count / fill all 2^D corners.
template <int D, bool FILL>
__global__ __launch_bounds__(256) void csr_count_fill_kernel(Levels lv, AxisPtrs<D> ax, int nM, int H, int W,
                                                             int n_parts, int64_t part_size, bool pair_merge,
                                                             uint32_t* __restrict__ counts_or_cursor,
                                                             const uint32_t* __restrict__ offs,
                                                             uint2* __restrict__ entries) {
  const int l = blockIdx.y;
  const int64_t n = (int64_t)nM * H * W;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const uint32_t c = (uint32_t)(p % W), r = (uint32_t)((p / W) % H), m = (uint32_t)(p / ((int64_t)W * H));
  uint32_t li[D];
  entry_dims<D>(m, r, c, li);
  const float scale = lv.scale[l];
  const uint32_t size = lv.size[l], res = lv.res[l];
  const bool hashed = (lv.hashed >> l) & 1u, pow2 = (lv.pow2 >> l) & 1u;
  uint32_t cell[D];
  float fr[D];
#pragma unroll
  for (int d = 0; d < D; ++d) pos_fract(ax.a[d][li[d]], scale, cell[d], fr[d]);
  const uint32_t part = (uint32_t)(p / part_size);
  const uint32_t p_rel = (uint32_t)(p - (int64_t)part * part_size);
  constexpr int NPAIR = 1 << (D - 1);
  uint32_t idx[NPAIR][2];
  float w[NPAIR][2], wrest[NPAIR];  // wrest: product of the factors of dimensions >= 1 (shared by the two dim-0 corners)
  bool live[NPAIR];
#pragma unroll
  for (int pair = 0; pair < NPAIR; ++pair) {
#pragma unroll
    for (int b0 = 0; b0 < 2; ++b0) {
      const int corner = 2 * pair + b0;
      uint32_t cc[D];
      float wc = 1.0f, wr = 1.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const bool hi = (corner >> d) & 1;
        cc[d] = cell[d] + (hi ? 1u : 0u);
        const float f = hi ? fr[d] : 1.0f - fr[d];
        wc *= f;
        if (d > 0) wr *= f;
      }
      idx[pair][b0] = grid_index<D>(cc, size, res, hashed, pow2);
      w[pair][b0] = wc;
      wrest[pair] = wr;
    }
    live[pair] = true;
  }
  // Corners of ONE point that land in the same slot are folded into one entry (weights summed).  This is
  // what makes the wrapped-stride levels cheap (levels 12-15 of the reference's grids, hashgrid.hip
  // build_levels: their index ignores the trailing dimensions, so 2 or 4 corner pairs coincide); on
  // ordinary levels nothing coincides and the entries are unchanged.
#pragma unroll
  for (int a = 1; a < NPAIR; ++a) {
#pragma unroll
    for (int b = 0; b < a; ++b) {
      if (live[a] && live[b] && idx[a][0] == idx[b][0] && idx[a][1] == idx[b][1]) {
        w[b][0] += w[a][0];
        w[b][1] += w[a][1];
        wrest[b] += wrest[a];
        live[a] = false;
      }
    }
  }
  const uint32_t kbase = lv.offset[l] * (uint32_t)n_parts + part * size;  // counters ordered by (level, part, slot)
#pragma unroll
  for (int pair = 0; pair < NPAIR; ++pair) {
    if (!live[pair]) continue;
    if (pair_merge && (idx[pair][0] ^ idx[pair][1]) == 1u) {
      // twin entry: both dim-0 corners of this point land in ONE aligned slot pair (see hashgrid.hip).
      // Stored once at the even slot: slot LSB = 1 when the even slot belongs to the HIGH dim-0 corner,
      // weight = -(product of the other dimensions' factors); the dim-0 fraction comes from a per-level
      // table at run time.  One dL/denc gather then serves two slots.
      // (Round 2 tried the general rule - any dim-0 pair inside one slot block, partner at slot ^ (c0 ^ (c0+1))
      // on hashed levels: 21-35 % fewer entries and gathers (71 M / 59 M instead of 90 M).  With the SAME kernel
      // the time follows the entry count (0.335 ms for 69 M entries in a timing-only experiment), but the far
      // partner's share has to go somewhere: scattered LDS float atomics cost 3 clocks PER LANE on this chip
      // (tools/bench_lds_atomic.hip): 0.58 / 0.48 / 0.43 ms; per-class partner sums carried with the run and
      // stored into extra LDS tiles (no atomics) need 160 VGPRs and 43 KB of LDS: 3 waves/SIMD, 0.447 ms, and
      // 4.3 instead of 3.5 ms at 640x640x20.  Neither beats the aligned-pair rule's 0.449 ms.)
      const uint32_t even = idx[pair][0] & ~1u, swap = idx[pair][0] & 1u;
      const uint32_t key = kbase + even;
      const uint32_t pos = atomicAdd(counts_or_cursor + key, 1u);
      if (FILL)
        entries[offs[key] + pos] = make_uint2((p_rel << SLOT_BITS) | (even & (SLOTS_PER_ITEM - 1)) | swap,
                                              __float_as_uint(wrest[pair]) | 0x80000000u);
    } else {
#pragma unroll
      for (int b0 = 0; b0 < 2; ++b0) {
        const uint32_t key = kbase + idx[pair][b0];
        const uint32_t pos = atomicAdd(counts_or_cursor + key, 1u);
        if (FILL)
          entries[offs[key] + pos] =
              make_uint2((p_rel << SLOT_BITS) | (idx[pair][b0] & (SLOTS_PER_ITEM - 1)), __float_as_uint(w[pair][b0]));
      }
    }
  }
}

// dim-0 interpolation fraction per (level, dim-0 lattice index): the run-time half of a twin entry's weights
__global__ void csr_f0_kernel(Levels lv, const float* __restrict__ ax0, int n0, float* __restrict__ f0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lv.n_levels * n0) return;
  uint32_t cell;
  float fr;
  pos_fract(ax0[i % n0], lv.scale[i / n0], cell, fr);
  f0[i] = fr;
}

struct BwdItem {
  uint32_t e0, e1;       // range in the slot-sorted build array (plan build only)
  uint32_t s0, ns;       // first global slot (block aligned within its level), number of slots
  uint32_t level;
  uint32_t part_shared;  // part | (shared << 16) | (first << 17): shared = the slot block also belongs to other items
                         // of the same part; first = no earlier item (of any round) writes this (table, slot block)
  uint32_t pe0, n_wc;    // start (in entries, multiple of 1024) and wave-chunk count in the final array
};

// Final entry layout: an item owns n_wc "wave chunks" of 1024 entries.  Inside a wave chunk the
// entries are stored transposed, [j = 0..7][lane = 0..63] x 16 bytes, so that a wave's eight loads
// are each 1 KiB contiguous (coalesced) while lane `lane` still receives the 16 CONSECUTIVE sorted
// entries 16*lane .. 16*lane+15 of the chunk.  (Thread-contiguous 128-byte reads were measured
// 25 % slower: each of the 8 loads touches 64 different lines.)  Chunks are padded with entries
// {point 0, slot of the last entry, weight 0}, which add 0 to the last run and need no masking.
constexpr int WAVE_CHUNK = 1024;

__global__ __launch_bounds__(256) void csr_permute_kernel(const BwdItem* __restrict__ items,
                                                          const uint2* __restrict__ sorted, uint2* __restrict__ out) {
  const BwdItem it = items[blockIdx.x];
  const uint32_t n_e = it.e1 - it.e0, n_pad = it.n_wc * WAVE_CHUNK;
  for (uint32_t s = threadIdx.x; s < n_pad; s += 256) {
    // padding: point 0, weight 0, the slot of the last real entry - it EXTENDS the last run (the backward
    // stores run sums with plain LDS stores, so a stray run on another slot would overwrite that slot's sum)
    const uint2 v = s < n_e ? sorted[it.e0 + s] : make_uint2(sorted[it.e1 - 1].x & (SLOTS_PER_ITEM - 1), 0u);
    const uint32_t wc = s / WAVE_CHUNK, lane = (s % WAVE_CHUNK) / 16, k = s % 16;
    out[(size_t)it.pe0 + (size_t)wc * WAVE_CHUNK + ((k >> 1) * 64 + lane) * 2 + (k & 1)] = v;
  }
}

// ---- backward ----------------------------------------------------------------------
// History (MI355X, 320x320x10): v1 lane-per-entry LDS atomics 1.86 ms; v2 LDS-staged tiles 0.87 ms
// (latency-bound, SQ_WAIT_ANY 61 %); v3 per-thread runs of 16 entries, 4-byte entries decoded on
// the fly 0.63 ms; v4 precomputed 8-byte entries, thread-contiguous loads 0.79 ms; v5 (this kernel)
// the same in the transposed wave-chunk layout 0.59 ms; v6 twin entries (the two dim-0 corners of a point
// that fall into one aligned slot pair share ONE entry and ONE gather: 19.5 % fewer entries) 0.54 ms
// isolated, 0.65 instead of 0.78 ms beside the image chain.  The same idea cost time on v3 (0.61 ms), where
// entries were decoded on the fly and the kernel was ALU-bound; twin corners merely stored adjacently
// gained nothing (0.59 ms on v5).  rocprof: TA busy 94 %, TA_ADDR_STALLED_BY_TC 82 %, TCP_PENDING_STALL 280 M cycles
// -> both gather kernels sit at the L1 miss-concurrency limit (~175 G L2 requests/s chip-wide).
constexpr int EPT = 16;                  // entries per thread per chunk

// v7 (round 2): NO scattered LDS float atomics.  tools/bench_lds_atomic.hip: ds_add_f32 costs ~3 clocks per
// ACTIVE LANE per CU (193 clocks for a full wave instruction, whatever the addresses; ds_add_u32: 8), so the
// "one LDS atomic per run" of v3-v6 - about 0.75 lane-atomics per entry - was 2.2 of the kernel's 2.9 clocks
// per entry, not the gathers.  Entries are slot-sorted and a lane owns 16 CONSECUTIVE ones, so a run (all
// entries of one slot, or of one slot pair with twin entries) is a contiguous range of the chunk:
//   * a run that lies inside one lane is written with a plain LDS store;
//   * a run that spans two lanes: the right lane hands its head partial sum to the left lane through a
//     DPP/permute shift, the left lane (which holds the run's first entry) adds it and stores;
//   * only what is left - lanes that lie wholly inside a longer run, and lane 0 of a chunk (its left
//     neighbour is another wave) - uses float atomics, into a SECOND tile, so that they never race with
//     the plain stores; the result is the sum of the two tiles.
// Exactly one lane holds a run's first entry, so every slot gets at most one plain store.
// Ablation (round 2, 320x320x10): with the divergent dL/denc gathers replaced by one shared line the kernel
// still takes 0.23 of its 0.455 ms - the 724 MB entry stream (3.2 TB/s) plus ~40 VALU instructions per entry -
// and the gathers ADD their 0.22 ms instead of hiding under it: 61.6 M L1->L2 requests per launch (rocprofv3
// TCP_TCC_READ_REQ) at 137 G/s, half of what immoco_probe_gather sustains with nothing else in flight.  (Issuing
// the gathers ahead of the next chunk's prefetch - vector-memory results return in issue order - is undone by
// the compiler, which sinks the last two gathers below the prefetch loads again; no change in the time.)
// The entry stream is read ONCE: non-temporal loads (`global_load_dwordx4 ... nt`) keep it from displacing the
// dL/denc window in the XCD's L2 - 0.482 -> 0.464 ms (4 parts), 0.462 -> 0.448 ms (8 parts, but Adam then reads
// eight partial tables: +0.04 ms).  The gradient tiles are written once and read by Adam much later: non-temporal
// stores too.  Same box, 4 parts: plain 0.4749, nt loads 0.4524, nt stores 0.4615, both 0.4474 ms.
// DH: dL/d enc is stored as packed halves (one 4-byte word per point and level, still multiplied by the fp16 MLP
// backward's loss scale: mlp_f16.hip); the gather is 4 bytes, the sums are unscaled (out_scale) on the way out.
template <int DIMS, bool PAIR, bool DH = false>  // DIMS names the instantiation (2: image grid, 3: motion grid) in profiles
__global__ __launch_bounds__(256) void csr_bwd_kernel(int64_t n_points, int64_t part_size,
                                                      const BwdItem* __restrict__ items,
                                                      const uint2* __restrict__ entries,
                                                      const float2* __restrict__ denc /*[L][n]*/,
                                                      float* __restrict__ dtable, int64_t part_stride,
                                                      int n_tables, int zeroed, const float* __restrict__ f0tab,
                                                      int n0, uint32_t hw, float inv_hw, int nt, float out_scale) {
  constexpr int NS = PAIR ? 4 : 2;                     // sums per run: (even.x, even.y, odd.x, odd.y) or (x, y)
  __shared__ __attribute__((aligned(16))) float accA[2 * SLOTS_PER_ITEM];  // plain stores: one per run
  __shared__ __attribute__((aligned(16))) float accB[2 * SLOTS_PER_ITEM];  // float atomics: the leftovers
  __shared__ float f0s[PAIR ? 256 : 1];
  const BwdItem it = items[blockIdx.x];
  if (it.n_wc == 0) return;  // padding item of the XCD interleave
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * (int)it.ns; i += 256) {
    accA[i] = 0.f;
    accB[i] = 0.f;
  }
  if (PAIR && tid < n0) f0s[tid] = f0tab[it.level * n0 + tid];
  __syncthreads();
  const uint32_t part = it.part_shared & 0xFFFFu;
  const float2* __restrict__ dl = denc + (int64_t)it.level * n_points + (int64_t)part * part_size;
  const __half2* __restrict__ dlh =
      reinterpret_cast<const __half2*>(denc) + (int64_t)it.level * n_points + (int64_t)part * part_size;
  const uint32_t p_off = (uint32_t)((int64_t)part * part_size);
  const int lane = tid & 63, wave = tid >> 6;
  const uint4* __restrict__ e4 = reinterpret_cast<const uint4*>(entries + it.pe0);
  uint4 q[EPT / 2];
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  auto ld = [&](size_t i) {
    if (nt & 1) {
      const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(e4 + i));
      return make_uint4(v.x, v.y, v.z, v.w);
    }
    return e4[i];
  };
  // nt & 4: TIMING-ONLY ablation (wrong results): only HALF of the entry stream is loaded - what 4-byte entries
  // would stream - and the missing entries are stand-ins made from the loaded ones with another point (same slot,
  // another cache line), so that gathers and arithmetic stay what they are.  The upper bound of what a 4-byte entry
  // format can gain BEFORE its decode work is added.
  const bool half_stream = (nt & 4) != 0;
  const uint32_t psz = (uint32_t)part_size;
  auto fill = [&](size_t base) {
#pragma unroll
    for (int j = 0; j < EPT / 2; ++j) {
      if (half_stream && j >= EPT / 4) {
        uint4 v = q[j - EPT / 4];
        const uint32_t a = v.x ^ (1u << (SLOT_BITS + 7)), b = v.z ^ (1u << (SLOT_BITS + 7));
        v.x = (a >> SLOT_BITS) < psz ? a : v.x;
        v.z = (b >> SLOT_BITS) < psz ? b : v.z;
        q[j] = v;
      } else {
        q[j] = ld(base + j * 64 + lane);
      }
    }
  };
  if ((uint32_t)wave < it.n_wc) fill((size_t)wave * (WAVE_CHUNK / 2));
  auto store_run = [&](uint32_t unit, const float (&v)[NS]) {   // unit: slot (NS = 2) or slot pair (NS = 4)
    if (PAIR) *reinterpret_cast<float4*>(&accA[4 * unit]) = make_float4(v[0], v[1], v[2], v[3]);
    else *reinterpret_cast<float2*>(&accA[2 * unit]) = make_float2(v[0], v[1]);
  };
  auto atomic_run = [&](uint32_t unit, const float (&v)[NS]) {
#pragma unroll
    for (int j = 0; j < NS; ++j) atomicAdd(&accB[NS * unit + j], v[j]);
  };
  for (uint32_t wc = wave; wc < it.n_wc; wc += 4) {
    uint32_t key[EPT];
    float wt[EPT];
#pragma unroll
    for (int j = 0; j < EPT / 2; ++j) {
      key[2 * j] = q[j].x;
      wt[2 * j] = __uint_as_float(q[j].y);
      key[2 * j + 1] = q[j].z;
      wt[2 * j + 1] = __uint_as_float(q[j].w);
    }
    if (wc + 4 < it.n_wc) fill((size_t)(wc + 4) * (WAVE_CHUNK / 2));  // prefetch this wave's next chunk
    float2 g[EPT];
    if (DH) {
      __half2 gh[EPT];
#pragma unroll
      for (int k = 0; k < EPT; ++k) gh[k] = dlh[key[k] >> SLOT_BITS];
#pragma unroll
      for (int k = 0; k < EPT; ++k) g[k] = __half22float2(gh[k]);
    } else {
#pragma unroll
      for (int k = 0; k < EPT; ++k) g[k] = dl[key[k] >> SLOT_BITS];
    }
    // run-length accumulate; the head run (the one that contains entry 0) is kept aside, runs in the
    // middle of the lane are stored at once, the tail run stays in s[]
    const uint32_t first = (key[0] & (SLOTS_PER_ITEM - 1)) >> (PAIR ? 1 : 0);
    uint32_t cur = first;
    float s[NS], hs[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) s[j] = hs[j] = 0.f;
    bool single = true;  // the lane has seen one run only so far
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const uint32_t loc = key[k] & (SLOTS_PER_ITEM - 1);
      const uint32_t unit = PAIR ? loc >> 1 : loc;
      float c[NS];
      if (PAIR) {
        const bool twin = __float_as_uint(wt[k]) >> 31, odd = loc & 1u;
        const float w = fabsf(wt[k]);
        // dim-0 lattice index of the point (exact: p < 2^24, one correction step each way)
        const uint32_t pnt = p_off + (key[k] >> SLOT_BITS);
        uint32_t i0 = (uint32_t)((float)pnt * inv_hw);
        i0 -= (i0 * hw > pnt) ? 1u : 0u;
        i0 += ((i0 + 1u) * hw <= pnt) ? 1u : 0u;
        const float f = f0s[twin ? i0 : 0u];
        // twin: the even slot takes the low dim-0 corner (1 - f) unless the swap bit (odd) is set
        const float fe = odd ? f : 1.f - f;
        const float we = twin ? w * fe : (odd ? 0.f : w);
        const float wo = twin ? w * (1.f - fe) : (odd ? w : 0.f);
        c[0] = we * g[k].x;
        c[1] = we * g[k].y;
        c[2] = wo * g[k].x;
        c[3] = wo * g[k].y;
      } else {
        c[0] = wt[k] * g[k].x;
        c[1] = wt[k] * g[k].y;
      }
      if (unit != cur) {
        if (single) {
#pragma unroll
          for (int j = 0; j < NS; ++j) hs[j] = s[j];
          single = false;
        } else {
          store_run(cur, s);
        }
        cur = unit;
#pragma unroll
        for (int j = 0; j < NS; ++j) s[j] = 0.f;
      }
#pragma unroll
      for (int j = 0; j < NS; ++j) s[j] += c[j];
    }
    // does the lane's first run continue the previous lane's last one?  (lane 0: unknown -> treated as open)
    const uint32_t prev_cur = __shfl_up(cur, 1, 64);
    const bool open_left = lane == 0 || prev_cur == first;
    // a lane with >= 2 runs hands an open head to its left neighbour, which holds that run's first entry
    // or is itself wholly inside it (then it goes on to the atomics below)
    const bool send = open_left && !single && lane != 0;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const float r = __shfl_down(send ? hs[j] : 0.f, 1, 64);
      s[j] += lane == 63 ? 0.f : r;
    }
    if (single) {
      if (open_left) atomic_run(cur, s);
      else store_run(cur, s);
    } else {
      store_run(cur, s);            // the tail run began in this lane
      if (!open_left) store_run(first, hs);
      else if (lane == 0) atomic_run(first, hs);
    }
  }
  __syncthreads();
  float* __restrict__ out = dtable + (size_t)(part % (uint32_t)n_tables) * part_stride + (size_t)it.s0 * 2;
  if ((it.part_shared >> 16) & 1u) {
    for (int i = tid; i < 2 * (int)it.ns; i += 256) {
      const float v = (accA[i] + accB[i]) * (DH ? out_scale : 1.f);
      if (v != 0.f) unsafeAtomicAdd(out + i, v);
    }
  } else if (zeroed && ((it.part_shared >> 17) & 1u)) {
    // exclusive owner of these (part, slot) pairs: no atomics.  Solver mode (`zeroed`): the FIRST writer of a
    // (table, slot block) over all rounds OVERWRITES the tile every iteration (zeros included), so nobody has
    // to clear it (Adam's fused zero_grad skips these ranges: 16 B/param less HBM traffic); later writers
    // (other parts of later rounds) and op-level mode accumulate.  The first writer is decided per ITEM at
    // plan build, not per round: a block whose only items sit in a later round (the wrapped-stride levels
    // index by the m cell, so different rounds hit different blocks) would otherwise accumulate for ever.
    if (nt & 2) {
      for (int i = tid; i < 2 * (int)it.ns; i += 256)
        __builtin_nontemporal_store((accA[i] + accB[i]) * (DH ? out_scale : 1.f), out + i);
    } else {
      for (int i = tid; i < 2 * (int)it.ns; i += 256) out[i] = (accA[i] + accB[i]) * (DH ? out_scale : 1.f);
    }
  } else {
    for (int i = tid; i < 2 * (int)it.ns; i += 256) {
      const float v = (accA[i] + accB[i]) * (DH ? out_scale : 1.f);
      if (v != 0.f) out[i] += v;
    }
  }
}

// ---- host side -------------------------------------------------------------------------
struct CsrPlan {
  int dims = 0;
  Levels lv{};
  int nM = 0, H = 0, W = 0;
  int n_parts = 1;               // point ranges
  int n_tables = 1;              // partial gradient tables = parts per ROUND; part q writes table q % n_tables
  std::vector<std::pair<uint32_t, uint32_t>> rounds;  // (first item, item count) of every round = one launch
  int64_t part_size = 0;
  const float* axes[3] = {nullptr, nullptr, nullptr};
  int32_t axn[3] = {0, 0, 0};
  uint2* entries = nullptr;
  BwdItem* items = nullptr;
  uint32_t n_items = 0;
  uint64_t n_entries = 0;
  int64_t bytes = 0;
  // slot blocks that receive any gradient, merged over the parts: {first slot, n slots | shared << 31}.
  // shared: some item flushes into the block with atomics, so the consumer (Adam) has to clear it.  Blocks
  // that are not listed never receive a gradient: their Adam update is exactly zero and is skipped.
  uint2* touched = nullptr;
  uint32_t n_touched = 0;
  bool pair_merge = false;       // twin entries present (3-D grids with <= 256 dim-0 lattice values)
  float* f0tab = nullptr;        // [n_levels][axn[0]] dim-0 fractions for the twin entries
};

void csr_plan_free(CsrPlan* p) {
  if (!p) return;
  if (p->entries) hipFree(p->entries);
  if (p->items) hipFree(p->items);
  if (p->f0tab) hipFree(p->f0tab);
  if (p->touched) hipFree(p->touched);
  delete p;
}

template <int D>
static int csr_build_t(CsrPlan* pl, hipStream_t st) {
  const Levels& lv = pl->lv;
  const int NP = pl->n_parts;
  const uint32_t n_slots = lv.offset[lv.n_levels];
  const size_t n_cnt = (size_t)n_slots * NP;
  const int64_t n = (int64_t)pl->nM * pl->H * pl->W;
  AxisPtrs<D> ax{};
  for (int d = 0; d < D; ++d) {
    ax.a[d] = pl->axes[d];
    ax.n[d] = pl->axn[d];
  }
  uint32_t *counts = nullptr, *offs = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  IMMOCO_CHECK_HIP(hipMalloc((void**)&counts, (n_cnt + 1) * 4));
  IMMOCO_CHECK_HIP(hipMalloc((void**)&offs, (n_cnt + 1) * 4));
  IMMOCO_CHECK_HIP(hipMemsetAsync(counts, 0, (n_cnt + 1) * 4, st));
  dim3 grid((unsigned)cdiv(n, 256), lv.n_levels);
  // twin entries: 3-D lattice whose dim 0 is the slowest axis (m), small enough for the LDS table and
  // for the float-assisted index division of the kernel
  pl->pair_merge = D == 3 && pl->axn[0] <= 256 && n < (1 << 24) && !immoco_diag_env("IMMOCO_CSR_NO_TWIN");
  if (pl->pair_merge) {
    const int nf = lv.n_levels * pl->axn[0];
    IMMOCO_CHECK_HIP(hipMalloc((void**)&pl->f0tab, (size_t)nf * sizeof(float)));
    csr_f0_kernel<<<cdiv(nf, 256), 256, 0, st>>>(lv, pl->axes[0], pl->axn[0], pl->f0tab);
    IMMOCO_LAUNCH_CHECK();
  }
  csr_count_fill_kernel<D, false><<<grid, 256, 0, st>>>(lv, ax, pl->nM, pl->H, pl->W, NP, pl->part_size, pl->pair_merge, counts,
                                                       nullptr, nullptr);
  IMMOCO_LAUNCH_CHECK();
  IMMOCO_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, counts, offs, (int)n_cnt + 1, st));
  IMMOCO_CHECK_HIP(hipMalloc(&tmp, tmp_bytes));
  IMMOCO_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, counts, offs, (int)n_cnt + 1, st));
  std::vector<uint32_t> h_offs(n_cnt + 1);
  IMMOCO_CHECK_HIP(hipMemcpyAsync(h_offs.data(), offs, h_offs.size() * 4, hipMemcpyDeviceToHost, st));
  IMMOCO_CHECK_HIP(hipStreamSynchronize(st));
  pl->n_entries = h_offs[n_cnt];
  // (twin entries and same-slot folding only ever remove entries)
  IMMOCO_REQUIRE(pl->n_entries <= (uint64_t)n * lv.n_levels * (1u << D), "csr plan: entry count mismatch");
  uint2* sorted = nullptr;  // slot-sorted build array, permuted into the final layout below
  IMMOCO_CHECK_HIP(hipMalloc((void**)&sorted, (size_t)pl->n_entries * 8));
  IMMOCO_CHECK_HIP(hipMemsetAsync(counts, 0, (n_cnt + 1) * 4, st));
  csr_count_fill_kernel<D, true><<<grid, 256, 0, st>>>(lv, ax, pl->nM, pl->H, pl->W, NP, pl->part_size, pl->pair_merge, counts, offs,
                                                      sorted);
  IMMOCO_LAUNCH_CHECK();
  // Work items per part: one aligned block of SLOTS_PER_ITEM slots of one (level, part); a block
  // with more than ENTRIES_PER_ITEM entries (coarse dense levels) is split by entries over several
  // items, which then flush with atomics ("shared").
  std::vector<std::vector<BwdItem>> per_part(NP);
  std::map<uint32_t, uint32_t> touched;  // first slot of a block -> n slots | shared << 31
  for (int l = 0; l < lv.n_levels; ++l) {
    for (int q = 0; q < NP; ++q) {
      const size_t cb = (size_t)lv.offset[l] * NP + (size_t)q * lv.size[l];  // counter index of slot 0
      for (uint32_t s = 0; s < lv.size[l]; s += SLOTS_PER_ITEM) {
        const uint32_t ns = std::min<uint32_t>(SLOTS_PER_ITEM, lv.size[l] - s);
        const uint32_t e0 = h_offs[cb + s], e1 = h_offs[cb + s + ns];
        if (e1 == e0) continue;
        uint32_t& tb = touched[lv.offset[l] + s];
        tb |= ns;
        if (e1 - e0 > (uint32_t)ENTRIES_PER_ITEM) tb |= 0x80000000u;
        if (e1 - e0 <= (uint32_t)ENTRIES_PER_ITEM) {
          per_part[q].push_back({e0, e1, lv.offset[l] + s, ns, (uint32_t)l, (uint32_t)q, 0u, 0u});
        } else {
          for (uint32_t e = e0; e < e1; e += ENTRIES_PER_ITEM)
            per_part[q].push_back({e, std::min<uint32_t>(e + ENTRIES_PER_ITEM, e1), lv.offset[l] + s, ns, (uint32_t)l,
                                   (uint32_t)q | (1u << 16), 0u, 0u});
        }
      }
    }
  }
  {
    // The consumer (Adam) visits the listed slot ranges only.  A touched 2048-slot block is listed whole unless at
    // most half of its 32-slot granules receive anything - tiny-cuda-nn's wrapped-stride levels 12-15 of a 2-D grid
    // spread a few thousand distinct slots evenly over all 256 blocks of the level (image grid at 320x320: 10, 5, 2.5
    // and 1.25 slots per block) - then only the occupied granules are listed, as runs.  (Twin entries are counted on
    // the even slot of their pair; a granule is even-aligned, so it holds both.)
    constexpr uint32_t GRAN = 32;
    std::vector<uint2> tv;
    tv.reserve(touched.size());
    for (int l = 0; l < lv.n_levels; ++l) {
      for (uint32_t sb = 0; sb < lv.size[l]; sb += SLOTS_PER_ITEM) {
        const auto itb = touched.find(lv.offset[l] + sb);
        if (itb == touched.end()) continue;
        const uint32_t ns = itb->second & 0x7FFFFFFFu, flag = itb->second & 0x80000000u;
        const uint32_t n_gran = (ns + GRAN - 1) / GRAN;
        std::vector<char> occ(n_gran, 0);
        uint32_t k = 0;
        for (uint32_t g = 0; g < n_gran; ++g) {
          const uint32_t a = sb + g * GRAN, b = sb + std::min(ns, (g + 1) * GRAN);
          for (int q = 0; q < NP && !occ[g]; ++q) {
            const size_t cb = (size_t)lv.offset[l] * NP + (size_t)q * lv.size[l];
            occ[g] = h_offs[cb + b] != h_offs[cb + a];
          }
          k += occ[g];
        }
        if (2 * k > n_gran || (ns % GRAN) != 0) {
          tv.push_back(make_uint2(lv.offset[l] + sb, ns | flag));
          continue;
        }
        for (uint32_t g = 0; g < n_gran;) {
          if (!occ[g]) {
            ++g;
            continue;
          }
          uint32_t e = g;
          while (e < n_gran && occ[e]) ++e;
          tv.push_back(make_uint2(lv.offset[l] + sb + g * GRAN, ((e - g) * GRAN) | flag));
          g = e;
        }
      }
    }
    pl->n_touched = (uint32_t)tv.size();
    IMMOCO_CHECK_HIP(hipMalloc((void**)&pl->touched, std::max<size_t>(1, tv.size()) * sizeof(uint2)));
    IMMOCO_CHECK_HIP(hipMemcpyAsync(pl->touched, tv.data(), tv.size() * sizeof(uint2), hipMemcpyHostToDevice, st));
    IMMOCO_CHECK_HIP(hipStreamSynchronize(st));  // tv goes out of scope
  }
  // XCD-aware interleave: workgroup i runs on XCD i % 8 (observed round-robin dispatch; a different
  // placement only costs speed).  XCD x serves part x % NP; the 8/NP XCDs of a part alternate over
  // that part's item list.  NP must divide 8.  (Part-major order - all XCDs on one part at a time, which
  // would allow pipelining the MLP backward of part k+1 under the encode backward of part k - costs
  // 0.68 ms instead of 0.59 ms: every XCD then pulls every part's dL/denc slices through its L2.)
  // Rounds: more than 8 parts (a level slice of dL/denc per part has to stay near 2 MB, an XCD's L2 share) or
  // fewer tables than parts (op-level plans: ONE table) run as n_parts / n_tables launches of n_tables parts
  // each; the first overwrites its tiles (solver mode), the following ones add to them - a (table, slot block)
  // pair has one owner per launch and launches are stream-ordered.
  const int NT = pl->n_tables, R = NP / NT;
  std::vector<BwdItem> items;
  for (int r = 0; r < R; ++r) {
    const size_t begin = items.size();
    if (NT == 1) {
      items.insert(items.end(), per_part[r].begin(), per_part[r].end());
    } else {
      const int xcds_per_part = 8 / NT;
      size_t steps = 0;
      for (int j = 0; j < NT; ++j)
        steps = std::max(steps, (per_part[r * NT + j].size() + xcds_per_part - 1) / xcds_per_part);
      items.resize(begin + steps * 8, BwdItem{0, 0, 0, 0, 0, 0, 0, 0});
      for (int x = 0; x < 8; ++x) {
        const int q = r * NT + x % NT, lane = x / NT;
        for (size_t k = 0; k < steps; ++k) {
          const size_t j = k * xcds_per_part + lane;
          if (j < per_part[q].size()) items[begin + k * 8 + x] = per_part[q][j];
        }
      }
    }
    pl->rounds.emplace_back((uint32_t)begin, (uint32_t)(items.size() - begin));
  }
  // first writer of every (table, slot block), in launch order (rounds are stream-ordered; inside a round a
  // (table, block) pair has one part, and a block that is split over several items is `shared`: atomics into a
  // tile that the consumer clears)
  {
    std::map<uint64_t, char> seen;
    for (BwdItem& it : items) {
      if (it.e1 == it.e0) continue;  // padding item
      const uint64_t key = ((uint64_t)((it.part_shared & 0xFFFFu) % (uint32_t)NT) << 32) | it.s0;
      if (seen.emplace(key, 1).second) it.part_shared |= 1u << 17;
    }
  }
  // final (padded, transposed) positions
  uint64_t total = 0;
  for (BwdItem& it : items) {
    it.n_wc = (it.e1 - it.e0 + WAVE_CHUNK - 1) / WAVE_CHUNK;
    it.pe0 = (uint32_t)total;
    total += (uint64_t)it.n_wc * WAVE_CHUNK;
  }
  IMMOCO_REQUIRE(total < 0xFFFFFF00ull, "csr plan: padded entry count exceeds uint32");
  pl->n_items = (uint32_t)items.size();
  IMMOCO_CHECK_HIP(hipMalloc((void**)&pl->items, std::max<size_t>(1, items.size()) * sizeof(BwdItem)));
  IMMOCO_CHECK_HIP(hipMemcpyAsync(pl->items, items.data(), items.size() * sizeof(BwdItem), hipMemcpyHostToDevice, st));
  IMMOCO_CHECK_HIP(hipMalloc((void**)&pl->entries, std::max<uint64_t>(total, 1) * 8));
  if (pl->n_items) {
    csr_permute_kernel<<<pl->n_items, 256, 0, st>>>(pl->items, sorted, pl->entries);
    IMMOCO_LAUNCH_CHECK();
  }
  IMMOCO_CHECK_HIP(hipStreamSynchronize(st));
  pl->bytes = (int64_t)total * 8 + (int64_t)items.size() * sizeof(BwdItem);
  hipFree(sorted);
  hipFree(counts);
  hipFree(offs);
  hipFree(tmp);
  return IMMOCO_OK;
}

// axes: device pointers, only read while the plan is built.
int csr_plan_build(const Levels& lv, int nM, int H, int W, const float* const* axes, const int32_t* axn,
                   int n_parts, int n_tables, CsrPlan** out, hipStream_t st) {
  IMMOCO_REQUIRE(n_parts >= 1 && n_parts <= 256 && (n_parts & (n_parts - 1)) == 0, "csr plan: n_parts must be a power of two <= 256");
  IMMOCO_REQUIRE((n_tables == 1 || n_tables == 2 || n_tables == 4 || n_tables == 8) && n_tables <= n_parts,
                 "csr plan: n_tables must divide 8 and not exceed n_parts");
  const int64_t n = (int64_t)nM * H * W;
  const int64_t part_size = cdiv(n, n_parts);
  IMMOCO_REQUIRE(part_size <= (1ll << (32 - SLOT_BITS)),
                 "csr plan: %lld points per part exceed the %d-bit entry field (use more parts)", (long long)part_size,
                 32 - SLOT_BITS);
  IMMOCO_REQUIRE((uint64_t)n * lv.n_levels * (1u << lv.dims) < 0xFFFFFF00ull, "csr plan: too many entries for uint32");
  CsrPlan* pl = new CsrPlan();
  pl->dims = lv.dims;
  pl->lv = lv;
  pl->nM = nM;
  pl->H = H;
  pl->W = W;
  pl->n_parts = n_parts;
  pl->n_tables = n_tables;
  pl->part_size = part_size;
  for (int d = 0; d < lv.dims; ++d) {
    pl->axes[d] = axes[d];
    pl->axn[d] = axn[d];
  }
  int rc = lv.dims == 3 ? csr_build_t<3>(pl, st) : csr_build_t<2>(pl, st);
  if (rc) {
    csr_plan_free(pl);
    return rc;
  }
  *out = pl;
  return IMMOCO_OK;
}

int64_t csr_plan_bytes(const CsrPlan* p) { return p ? p->bytes : 0; }
int64_t csr_plan_entries(const CsrPlan* p) { return p ? (int64_t)p->n_entries : 0; }
int csr_plan_parts(const CsrPlan* p) { return p ? p->n_parts : 1; }
int csr_plan_tables(const CsrPlan* p) { return p ? p->n_tables : 1; }
// parts for a lattice of n points: one level slice of dL/denc (8 B per point in fp32, 4 B as packed halves) per
// part ~ 2 MB
int csr_auto_parts(int64_t n_points, int bytes_per_point) {
  const int64_t slice = n_points * bytes_per_point;
  int parts = 1;
  while (parts < 256 && slice > (int64_t)parts * (2 << 20)) parts *= 2;
  return parts;
}
const uint2* csr_plan_touched(const CsrPlan* p, uint32_t* n) {
  *n = p ? p->n_touched : 0;
  return p ? p->touched : nullptr;
}

// dtable: n_parts partial tables, `part_stride` floats apart.  zeroed != 0: the caller guarantees
// that the buffers hold zeros or stale values of the same plan (plain stores); otherwise the
// results are accumulated.
int launch_csr_bwd(const CsrPlan* pl, const float* denc_level_major, float* dtable, int64_t part_stride, int zeroed,
                   hipStream_t st, bool denc_half, float out_scale) {
  if (!pl || pl->n_items == 0) return IMMOCO_OK;
  const int64_t n = (int64_t)pl->nM * pl->H * pl->W;
  const uint32_t hw = (uint32_t)pl->H * (uint32_t)pl->W;
  // A/B switch (environment, read once): IMMOCO_CSR_STREAM = plain | 1 (nt entry loads) | 2 (nt tile stores) | 3 (both,
  // the default)
  static const int nt = [] {
    const char* e = immoco_diag_env("IMMOCO_CSR_STREAM");
    return (e && strcmp(e, "plain") == 0) ? 0 : (e && atoi(e) > 0 ? (atoi(e) & 7) : 3);   // 7 = 3 + the half-stream ablation
  }();
  // Three workgroups per CU instead of the four that 33 KB of LDS would allow (12 KB of unused dynamic LDS): fewer
  // (part, level) windows in flight per XCD L2.  Isolated kernel, 4 -> 3 per CU: 320x320x10 0.448 -> 0.429 ms (2 per
  // CU: 0.438; graph iteration 1.337 -> 1.322), 320x320x20 0.837 -> 0.800, 256x256x8 0.252 -> 0.241, 160x160x10
  // unchanged; plans that run as several rounds (launches of 8 parts) lose instead - 480x480x10 1.25 -> 1.33,
  // 640x640x20 3.54 -> 3.65 - and keep four.  A/B switch (environment, read once): IMMOCO_CSR_PAD_LDS=<bytes>.
  static const int pad_env = [] { const char* e = immoco_diag_env("IMMOCO_CSR_PAD_LDS"); return e ? atoi(e) : -1; }();
  const int pad_lds = pad_env >= 0 ? pad_env : (pl->rounds.size() == 1 ? 12288 : 0);
  for (size_t r = 0; r < pl->rounds.size(); ++r) {
    const uint32_t first = pl->rounds[r].first, cnt = pl->rounds[r].second;
    if (cnt == 0) continue;
#define IMMOCO_CSR_BWD(D, PAIR, DH)                                                                             \
  csr_bwd_kernel<D, PAIR, DH><<<cnt, 256, pad_lds, st>>>(n, pl->part_size, pl->items + first, pl->entries,             \
                                                           (const float2*)denc_level_major, dtable, part_stride, \
                                                           pl->n_tables, zeroed, pl->f0tab, pl->axn[0], hw,      \
                                                           1.0f / (float)hw, nt, out_scale)
    if (denc_half) {
      if (pl->dims == 3 && pl->pair_merge) IMMOCO_CSR_BWD(3, true, true);
      else if (pl->dims == 3) IMMOCO_CSR_BWD(3, false, true);
      else IMMOCO_CSR_BWD(2, false, true);
    } else {
      if (pl->dims == 3 && pl->pair_merge) IMMOCO_CSR_BWD(3, true, false);
      else if (pl->dims == 3) IMMOCO_CSR_BWD(3, false, false);
      else IMMOCO_CSR_BWD(2, false, false);
    }
#undef IMMOCO_CSR_BWD
  }
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

}  // namespace immoco

using namespace immoco;

struct immoco_grid_plan {
  CsrPlan* plan;
};

extern "C" int immoco_grid_plan_create(const immoco_grid_cfg* cfg, int32_t nM, int32_t H, int32_t W,
                                       const float* ax0, const float* ax1, const float* ax2,
                                       immoco_grid_plan_t* out, void* stream) {
  Levels lv;
  int rc = build_levels(cfg, &lv);
  if (rc) return rc;
  IMMOCO_REQUIRE(out && ax0 && ax1 && (lv.dims == 2 || ax2), "grid_plan_create: NULL argument");
  IMMOCO_REQUIRE(nM >= 1 && H >= 1 && W >= 1, "grid_plan_create: bad lattice %dx%dx%d", nM, H, W);
  IMMOCO_REQUIRE(lv.dims == 3 || nM == 1, "grid_plan_create: a 2-D grid takes nM = 1");
  const float* axes[3] = {ax0, ax1, ax2};
  int32_t axn[3];
  if (lv.dims == 3) {
    axn[0] = nM; axn[1] = H; axn[2] = W;   // (m, row, col)
  } else {
    axn[0] = W; axn[1] = H; axn[2] = 0;    // (x = col, y = row)
  }
  CsrPlan* pl = nullptr;
  // one output table; large lattices are walked in rounds of ~2 MB of dL/denc per level (csr_auto_parts)
  if ((rc = csr_plan_build(lv, nM, H, W, axes, axn, csr_auto_parts((int64_t)nM * H * W), 1, &pl, as_stream(stream))))
    return rc;
  *out = new immoco_grid_plan{pl};
  return IMMOCO_OK;
}

extern "C" int immoco_grid_plan_destroy(immoco_grid_plan_t p) {
  if (!p) return IMMOCO_OK;
  csr_plan_free(p->plan);
  delete p;
  return IMMOCO_OK;
}

extern "C" int64_t immoco_grid_plan_bytes(immoco_grid_plan_t p) { return p ? csr_plan_bytes(p->plan) : 0; }

extern "C" int immoco_grid_plan_bwd(immoco_grid_plan_t p, const float* denc_level_major, float* dtable,
                                    void* stream) {
  IMMOCO_REQUIRE(p && denc_level_major && dtable, "grid_plan_bwd: NULL argument");
  return launch_csr_bwd(p->plan, denc_level_major, dtable, 0, 0, as_stream(stream), false, 1.f);
}
