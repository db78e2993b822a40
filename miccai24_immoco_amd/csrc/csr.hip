// Hash-grid backward WITHOUT a global atomic scatter.
//
// The reference always queries its INRs on the same lattice (immoco.py:48-53,72-80),
// so the (point, corner) -> table-slot map of tiny-cuda-nn's encoding is a constant
// of the slice shape.  Measured on MI355X the naive backward (one float atomic per
// corner and feature, 2 x 131 M per iteration at 320x320 / 10 groups) runs at the
// memory-side atomic rate and takes 12.9 ms per iteration (77 % of the step).
//
// Plan (once per solver): a transposed index ("CSR by slot"): for every table slot the
// list of (m, row, col, corner) contributions, 4 bytes each, slot-sorted, built on the
// GPU (count -> exclusive scan -> fill), plus a host-built list of work items that cover
// <= SLOTS_PER_ITEM consecutive slots and <= ENTRIES_PER_ITEM entries each.
//
// Backward (every iteration): one workgroup per work item streams its entries
// (coalesced 4-B reads), recomputes slot + interpolation weight from per-level axis
// tables held in LDS, gathers dL/denc (8 B, level slice is L2/MALL resident),
// accumulates into an LDS tile of the gradient table (ds_add_f32), and flushes the
// tile with contiguous float atomics (the fast 256-B shape; slots that straddle work
// items are thereby summed correctly).
//
// L2 locality: the gather's working set is one level slice of dL/denc (8 B x points = 8 MB at
// 320x320x10), twice an XCD's 4 MB L2, so v1 ran at Infinity-Cache speed (1.24 ms).  The plan
// therefore splits the points into `n_parts` contiguous ranges, sorts entries by (level, part,
// slot) and orders the work items so that workgroup index i (which lands on XCD i % 8) only
// touches part (i % 8) % n_parts: every XCD keeps a 8/n_parts MB slice hot.  Each part writes its
// own partial gradient table (plain stores; a (part, slot) pair belongs to exactly one work item
// unless the slot is oversized); the Adam kernel sums the partial tables while reading them.
#include <hipcub/hipcub.hpp>

#include <vector>

#include "kernels.hpp"

namespace immoco {

constexpr int SLOTS_PER_ITEM = 2048;
constexpr int ENTRIES_PER_ITEM = 32768;

// entry packing: corner [0,3) | col [3,13) | row [13,23) | m [23,28)
__device__ __forceinline__ uint32_t pack_entry(uint32_t m, uint32_t r, uint32_t c, uint32_t corner) {
  return corner | (c << 3) | (r << 13) | (m << 23);
}

// lattice index of dimension d for entry fields (m, r, c):
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
                                                             int n_parts, int64_t part_size,
                                                             uint32_t* __restrict__ counts_or_cursor,
                                                             const uint32_t* __restrict__ offs,
                                                             uint32_t* __restrict__ entries) {
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
#pragma unroll
  for (int corner = 0; corner < (1 << D); ++corner) {
    uint32_t cc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) cc[d] = cell[d] + ((corner >> d) & 1);
    // counter index ordered by (level, part, slot)
    const uint32_t part = (uint32_t)(p / part_size);
    const uint32_t slot = lv.offset[l] * (uint32_t)n_parts + part * size + grid_index<D>(cc, size, res, hashed, pow2);
    const uint32_t pos = atomicAdd(counts_or_cursor + slot, 1u);
    if (FILL) entries[offs[slot] + pos] = pack_entry(m, r, c, (uint32_t)corner);
  }
}

struct BwdItem {
  uint32_t e0, e1;    // entry range
  uint32_t s0, ns;    // first global slot, number of slots covered
  uint32_t level;
  uint32_t part_shared;  // part | (shared << 16): shared = the slot range also belongs to other items
};

// ---- backward ----------------------------------------------------------------------
// Entries are slot-sorted, so neighbouring entries mostly share a slot (runs of ~16 on the hashed
// levels of the motion grid).  v1 fed them lane-by-lane into LDS atomics, which serialise on the
// shared address (1.86 ms per iteration).  v2 staged 4096-entry tiles through LDS with two
// barriers per tile and was latency-bound (rocprof: SQ_WAIT_ANY 61 % of wave cycles, 0.87 ms).
// v3 (this kernel): every thread owns 16 CONSECUTIVE entries = four aligned 16-byte loads straight
// from global memory (no staging, no barrier inside the loop; the next chunk is prefetched while
// the current one is processed), sums equal-slot runs in registers and issues one LDS atomic per
// run, so lanes of one wave-instruction hit different slots.
constexpr int EPT = 16;                  // entries per thread per chunk
constexpr int CHUNK_ENTRIES = 256 * EPT;  // 4096

template <int D>
__global__ __launch_bounds__(256) void csr_bwd_kernel(Levels lv, AxisPtrs<D> ax, int H, int W, int64_t n_points,
                                                      const BwdItem* __restrict__ items,
                                                      const uint32_t* __restrict__ entries,
                                                      const float2* __restrict__ denc /*[L][n]*/,
                                                      float* __restrict__ dtable, int64_t part_stride,
                                                      int zeroed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* acc = reinterpret_cast<float*>(smem);  // [2*SLOTS_PER_ITEM]
  // per-axis tables indexed by ENTRY FIELD: col, row, m
  const int n_col = ax.n[D == 3 ? 2 : 0], n_row = ax.n[1], n_m = D == 3 ? ax.n[0] : 0;
  uint32_t* c_col = reinterpret_cast<uint32_t*>(acc + 2 * SLOTS_PER_ITEM);
  uint32_t* c_row = c_col + n_col;
  uint32_t* c_m = c_row + n_row;
  float* f_col = reinterpret_cast<float*>(c_m + n_m);
  float* f_row = f_col + n_col;
  float* f_m = f_row + n_row;

  const BwdItem it = items[blockIdx.x];
  if (it.e0 >= it.e1) return;  // padding item of the XCD interleave
  const int l = (int)it.level;
  const float scale = lv.scale[l];
  const uint32_t size = lv.size[l], res = lv.res[l];
  const bool hashed = (lv.hashed >> l) & 1u, pow2 = (lv.pow2 >> l) & 1u;
  const uint32_t slot_base = it.s0 - lv.offset[l];
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * (int)it.ns; i += 256) acc[i] = 0.f;
  for (int i = tid; i < n_col; i += 256) pos_fract(ax.a[D == 3 ? 2 : 0][i], scale, c_col[i], f_col[i]);
  for (int i = tid; i < n_row; i += 256) pos_fract(ax.a[1][i], scale, c_row[i], f_row[i]);
  if (D == 3)
    for (int i = tid; i < n_m; i += 256) pos_fract(ax.a[0][i], scale, c_m[i], f_m[i]);
  __syncthreads();
  const float2* __restrict__ dl = denc + (int64_t)l * n_points;
  // 16-byte aligned walk: start at e0 rounded down to a multiple of 4 entries and mask the strays
  // (the entries allocation is padded, see csr_build_t)
  const uint32_t a0 = it.e0 & ~3u;
  const uint4* __restrict__ e4 = reinterpret_cast<const uint4*>(entries);
  uint32_t base = a0 + (uint32_t)tid * EPT;
  uint4 q[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) q[j] = base + 4u * j < it.e1 ? e4[(base >> 2) + j] : make_uint4(0, 0, 0, 0);
  for (uint32_t cb = a0; cb < it.e1; cb += CHUNK_ENTRIES) {
    uint32_t u[EPT];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u[4 * j] = q[j].x;
      u[4 * j + 1] = q[j].y;
      u[4 * j + 2] = q[j].z;
      u[4 * j + 3] = q[j].w;
    }
    const uint32_t my = base;
    base += CHUNK_ENTRIES;
    if (cb + CHUNK_ENTRIES < it.e1) {  // prefetch the next chunk
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = base + 4u * j < it.e1 ? e4[(base >> 2) + j] : make_uint4(0, 0, 0, 0);
    }
    uint32_t loc[EPT];
    float wt[EPT];
    float2 g[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const uint32_t idx = my + k;
      const bool ok = idx >= it.e0 && idx < it.e1;
      const uint32_t uu = ok ? u[k] : 0u;
      const uint32_t corner = uu & 7u, c = (uu >> 3) & 1023u, r = (uu >> 13) & 1023u, m = (uu >> 23) & 31u;
      uint32_t cc[D];
      float w;
      if (D == 3) {
        const uint32_t b0 = corner & 1u, b1 = (corner >> 1) & 1u, b2 = (corner >> 2) & 1u;
        cc[0] = c_m[m] + b0;
        cc[1] = c_row[r] + b1;
        if (D > 2) cc[2] = c_col[c] + b2;
        const float f0 = f_m[m], f1 = f_row[r], f2 = f_col[c];
        w = (b0 ? f0 : 1.f - f0) * (b1 ? f1 : 1.f - f1) * (b2 ? f2 : 1.f - f2);
      } else {
        const uint32_t b0 = corner & 1u, b1 = (corner >> 1) & 1u;
        cc[0] = c_col[c] + b0;
        cc[1] = c_row[r] + b1;
        const float f0 = f_col[c], f1 = f_row[r];
        w = (b0 ? f0 : 1.f - f0) * (b1 ? f1 : 1.f - f1);
      }
      loc[k] = ok ? grid_index<D>(cc, size, res, hashed, pow2) - slot_base : 0xFFFFFFFFu;
      wt[k] = ok ? w : 0.f;
      g[k] = dl[((int64_t)m * H + r) * W + c];
    }
    // run-length accumulate
    uint32_t cur = loc[0];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (loc[k] != cur) {
        if (cur < it.ns) {
          atomicAdd(&acc[2 * cur], s0);
          atomicAdd(&acc[2 * cur + 1], s1);
        }
        cur = loc[k];
        s0 = s1 = 0.f;
      }
      s0 = fmaf(wt[k], g[k].x, s0);
      s1 = fmaf(wt[k], g[k].y, s1);
    }
    if (cur < it.ns) {  // (masked entries carry 0xFFFFFFFF and are dropped)
      atomicAdd(&acc[2 * cur], s0);
      atomicAdd(&acc[2 * cur + 1], s1);
    }
  }
  __syncthreads();
  float* __restrict__ out = dtable + (size_t)(it.part_shared & 0xFFFFu) * part_stride + (size_t)it.s0 * 2;
  if (it.part_shared >> 16) {
    for (int i = tid; i < 2 * (int)it.ns; i += 256) {
      const float v = acc[i];
      if (v != 0.f) unsafeAtomicAdd(out + i, v);
    }
  } else {
    // exclusive owner of these (part, slot) pairs: no atomics.  Solver mode (`zeroed`): the tile is
    // OVERWRITTEN every iteration (zeros included), so nobody has to clear it (Adam's fused
    // zero_grad skips these ranges: 16 B/param less HBM traffic); op-level mode accumulates.
    if (zeroed) {
      for (int i = tid; i < 2 * (int)it.ns; i += 256) out[i] = acc[i];
    } else {
      for (int i = tid; i < 2 * (int)it.ns; i += 256) {
        const float v = acc[i];
        if (v != 0.f) out[i] += v;
      }
    }
  }
}

static size_t csr_bwd_smem(int n_col, int n_row, int n_m) {
  return (size_t)2 * SLOTS_PER_ITEM * 4 + (size_t)(n_col + n_row + n_m) * 8;
}

// ---- host side -------------------------------------------------------------------------
struct CsrPlan {
  int dims = 0;
  Levels lv{};
  int nM = 0, H = 0, W = 0;
  int n_parts = 1;
  int64_t part_size = 0;
  const float* axes[3] = {nullptr, nullptr, nullptr};
  int32_t axn[3] = {0, 0, 0};
  uint32_t* entries = nullptr;
  BwdItem* items = nullptr;
  uint32_t n_items = 0;
  uint64_t n_entries = 0;
  int64_t bytes = 0;
  uint32_t shared_slot_end = 0;  // slots < this may belong to "shared" (atomic-flush) items
};

void csr_plan_free(CsrPlan* p) {
  if (!p) return;
  if (p->entries) hipFree(p->entries);
  if (p->items) hipFree(p->items);
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
  csr_count_fill_kernel<D, false><<<grid, 256, 0, st>>>(lv, ax, pl->nM, pl->H, pl->W, NP, pl->part_size, counts,
                                                       nullptr, nullptr);
  IMMOCO_LAUNCH_CHECK();
  IMMOCO_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, counts, offs, (int)n_cnt + 1, st));
  IMMOCO_CHECK_HIP(hipMalloc(&tmp, tmp_bytes));
  IMMOCO_CHECK_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, counts, offs, (int)n_cnt + 1, st));
  std::vector<uint32_t> h_offs(n_cnt + 1);
  IMMOCO_CHECK_HIP(hipMemcpyAsync(h_offs.data(), offs, h_offs.size() * 4, hipMemcpyDeviceToHost, st));
  IMMOCO_CHECK_HIP(hipStreamSynchronize(st));
  pl->n_entries = h_offs[n_cnt];
  IMMOCO_REQUIRE(pl->n_entries == (uint64_t)n * lv.n_levels * (1u << D), "csr plan: entry count mismatch");
  IMMOCO_CHECK_HIP(hipMalloc((void**)&pl->entries, (size_t)pl->n_entries * 4 + 64));  // + slack for aligned 16-B reads
  IMMOCO_CHECK_HIP(hipMemsetAsync(counts, 0, (n_cnt + 1) * 4, st));
  csr_count_fill_kernel<D, true><<<grid, 256, 0, st>>>(lv, ax, pl->nM, pl->H, pl->W, NP, pl->part_size, counts, offs,
                                                      pl->entries);
  IMMOCO_LAUNCH_CHECK();
  // Work items per part: consecutive slots of one (level, part), bounded in slots and entries; a
  // slot with more entries than ENTRIES_PER_ITEM (coarse dense levels) is split over several
  // items, which then flush with atomics ("shared").
  std::vector<std::vector<BwdItem>> per_part(NP);
  for (int l = 0; l < lv.n_levels; ++l) {
    for (int q = 0; q < NP; ++q) {
      const size_t cb = (size_t)lv.offset[l] * NP + (size_t)q * lv.size[l];  // counter index of slot 0
      uint32_t s = 0;
      const uint32_t s_end = lv.size[l];
      while (s < s_end) {
        const uint32_t e0 = h_offs[cb + s];
        uint32_t s1 = s;
        while (s1 < s_end && (s1 - s) < (uint32_t)SLOTS_PER_ITEM && (h_offs[cb + s1 + 1] - e0) <= (uint32_t)ENTRIES_PER_ITEM)
          ++s1;
        if (s1 == s) {
          const uint32_t e_end = h_offs[cb + s + 1];
          for (uint32_t e = e0; e < e_end; e += ENTRIES_PER_ITEM)
            per_part[q].push_back({e, std::min<uint32_t>(e + ENTRIES_PER_ITEM, e_end), lv.offset[l] + s, 1u,
                                   (uint32_t)l, (uint32_t)q | (1u << 16)});
          pl->shared_slot_end = std::max(pl->shared_slot_end, lv.offset[l] + s + 1);
          s = s + 1;
        } else {
          if (h_offs[cb + s1] > e0)
            per_part[q].push_back({e0, h_offs[cb + s1], lv.offset[l] + s, s1 - s, (uint32_t)l, (uint32_t)q});
          s = s1;
        }
      }
    }
  }
  // XCD-aware interleave: workgroup i runs on XCD i % 8 (observed round-robin dispatch; a different
  // placement only costs speed).  XCD x serves part x % NP; the 8/NP XCDs of a part alternate over
  // that part's item list.  NP must divide 8.
  std::vector<BwdItem> items;
  if (NP == 1) {
    items = per_part[0];
  } else {
    const int xcds_per_part = 8 / NP;
    size_t rounds = 0;
    for (int q = 0; q < NP; ++q) rounds = std::max(rounds, (per_part[q].size() + xcds_per_part - 1) / xcds_per_part);
    items.assign(rounds * 8, BwdItem{0, 0, 0, 0, 0, 0});
    for (int x = 0; x < 8; ++x) {
      const int q = x % NP, lane = x / NP;
      for (size_t k = 0; k < rounds; ++k) {
        const size_t j = k * xcds_per_part + lane;
        if (j < per_part[q].size()) items[k * 8 + x] = per_part[q][j];
      }
    }
  }
  pl->n_items = (uint32_t)items.size();
  IMMOCO_CHECK_HIP(hipMalloc((void**)&pl->items, std::max<size_t>(1, items.size()) * sizeof(BwdItem)));
  IMMOCO_CHECK_HIP(hipMemcpyAsync(pl->items, items.data(), items.size() * sizeof(BwdItem), hipMemcpyHostToDevice, st));
  IMMOCO_CHECK_HIP(hipStreamSynchronize(st));
  pl->bytes = (int64_t)pl->n_entries * 4 + (int64_t)items.size() * sizeof(BwdItem);
  hipFree(counts);
  hipFree(offs);
  hipFree(tmp);
  return IMMOCO_OK;
}

// axes: device pointers that must stay valid and constant for the plan's lifetime.
int csr_plan_build(const Levels& lv, int nM, int H, int W, const float* const* axes, const int32_t* axn,
                   int n_parts, CsrPlan** out, hipStream_t st) {
  IMMOCO_REQUIRE(W <= 1024 && H <= 1024 && nM <= 32, "csr plan: lattice %dx%dx%d exceeds the entry packing", nM, H, W);
  IMMOCO_REQUIRE(n_parts == 1 || n_parts == 2 || n_parts == 4 || n_parts == 8, "csr plan: n_parts must divide 8");
  CsrPlan* pl = new CsrPlan();
  pl->dims = lv.dims;
  pl->lv = lv;
  pl->nM = nM;
  pl->H = H;
  pl->W = W;
  pl->n_parts = n_parts;
  pl->part_size = cdiv((int64_t)nM * H * W, n_parts);
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
int csr_plan_parts(const CsrPlan* p) { return p ? p->n_parts : 1; }
uint32_t csr_plan_shared_slot_end(const CsrPlan* p) { return p ? p->shared_slot_end : 0; }

// dtable: n_parts partial tables, `part_stride` floats apart.  zeroed != 0: the caller guarantees
// that the buffers hold zeros (plain stores); otherwise the results are accumulated.
int launch_csr_bwd(const CsrPlan* pl, const float* denc_level_major, float* dtable, int64_t part_stride, int zeroed,
                   hipStream_t st) {
  if (!pl || pl->n_items == 0) return IMMOCO_OK;
  const int64_t n = (int64_t)pl->nM * pl->H * pl->W;
  if (pl->dims == 3) {
    AxisPtrs<3> ax{};
    for (int d = 0; d < 3; ++d) {
      ax.a[d] = pl->axes[d];
      ax.n[d] = pl->axn[d];
    }
    csr_bwd_kernel<3><<<pl->n_items, 256, csr_bwd_smem(pl->W, pl->H, pl->nM), st>>>(
        pl->lv, ax, pl->H, pl->W, n, pl->items, pl->entries, (const float2*)denc_level_major, dtable, part_stride,
        zeroed);
  } else {
    AxisPtrs<2> ax{};
    for (int d = 0; d < 2; ++d) {
      ax.a[d] = pl->axes[d];
      ax.n[d] = pl->axn[d];
    }
    csr_bwd_kernel<2><<<pl->n_items, 256, csr_bwd_smem(pl->W, pl->H, 0), st>>>(
        pl->lv, ax, pl->H, pl->W, n, pl->items, pl->entries, (const float2*)denc_level_major, dtable, part_stride,
        zeroed);
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
  if ((rc = csr_plan_build(lv, nM, H, W, axes, axn, 1, &pl, as_stream(stream)))) return rc;
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
  return launch_csr_bwd(p->plan, denc_level_major, dtable, 0, 0, as_stream(stream));
}
