// Access-pattern probes: what the chip sustains for the gather shape of the hash-grid kernels.
//
// The hash-grid encode (csrc/hashgrid.hip) and its transposed backward (csrc/csr.hip) are not
// bounded by HBM bytes but by the rate of divergent, cache-line-granular requests (DESIGN.md
// §4.1).  The guide's peaks (HBM 8 TB/s, MFMA) do not price that, so bench.py reports next to the
// HBM roofline the *measured* ceiling of the same request shape: every lane issues independent
// 16-byte (or 8-byte) loads at pseudo-random aligned offsets inside a `footprint`-byte table, LOADS
// per lane with 4 in flight, nothing else.  footprint = 4 MB is one fp32 level slice (what a level's
// workgroups share in an XCD's L2), 57 MB the whole motion table.
#include "common.hpp"
#include "kernels.hpp"

namespace immoco {

template <typename V>
__global__ __launch_bounds__(256) void probe_gather_kernel(const V* __restrict__ table, uint32_t mask,
                                                           int loads, float* __restrict__ out) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  uint32_t s = pcg_hash(tid * 2654435761u + 12345u);
  float acc = 0.f;
  for (int i = 0; i < loads; i += 4) {
    uint32_t j[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s = s * 747796405u + 2891336453u;
      j[u] = ((s >> 9) ^ s) & mask;
    }
    V v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = table[j[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += v[u].x;
  }
  out[tid] = acc;
}

// Diagnostic (environment IMMOCO_PROBE_PATTERN=<cells per pixel>, loads_per_lane is then 4): the address pattern of
// one fine hashed level of the motion grid - lane = pixel column, the four (y, z) corner pairs of the tcnn hash.
// Measured: the same 240-268 G loads/s as pseudo-random addresses (4 MB / 2 MB footprint), so the hash pattern itself
// (L2 channel conflicts) is not what the encode forward loses against the probe.
template <typename V>
__global__ __launch_bounds__(256) void probe_hash_kernel(const V* __restrict__ table, uint32_t mask, float cpp,
                                                         float* __restrict__ out) {
  const uint32_t tid = blockIdx.x * 256u + threadIdx.x;
  const uint32_t z = tid % 320u, y = (tid / 320u) % 320u, m = tid / 102400u;
  const uint32_t cz = (uint32_t)((float)z * cpp), cy = (uint32_t)((float)y * cpp), cx = m * 7919u;
  uint32_t j[4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
    j[u] = (cx ^ ((cy + (u & 1)) * 2654435761u) ^ ((cz + (u >> 1)) * 805459861u)) & mask;
  V v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = table[j[u]];
  out[tid] = v[0].x + v[1].x + v[2].x + v[3].x;
}

}  // namespace immoco

using namespace immoco;

extern "C" int immoco_probe_gather(int64_t footprint_bytes, int32_t bytes_per_load, int64_t n_lanes,
                                   int32_t loads_per_lane, int32_t repeats, void* stream, float* ms_out) {
  hipStream_t st = as_stream(stream);
  IMMOCO_REQUIRE(ms_out != nullptr, "ms_out is NULL");
  IMMOCO_REQUIRE(bytes_per_load == 8 || bytes_per_load == 16, "bytes_per_load must be 8 or 16");
  IMMOCO_REQUIRE(footprint_bytes >= 4096 && (footprint_bytes & (footprint_bytes - 1)) == 0,
                 "footprint must be a power of two >= 4096");
  IMMOCO_REQUIRE(n_lanes > 0 && n_lanes % 256 == 0 && n_lanes <= (int64_t)1 << 30, "n_lanes must be a multiple of 256");
  IMMOCO_REQUIRE(loads_per_lane > 0 && loads_per_lane % 4 == 0, "loads_per_lane must be a multiple of 4");
  IMMOCO_REQUIRE(repeats > 0, "repeats must be positive");
  void* table = nullptr;
  float* out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  IMMOCO_CHECK_HIP(hipMalloc(&table, (size_t)footprint_bytes));
  IMMOCO_CHECK_HIP(hipMalloc(&out, (size_t)n_lanes * sizeof(float)));
  IMMOCO_CHECK_HIP(hipMemsetAsync(table, 0, (size_t)footprint_bytes, st));
  IMMOCO_CHECK_HIP(hipEventCreate(&e0));
  IMMOCO_CHECK_HIP(hipEventCreate(&e1));
  const uint32_t mask = (uint32_t)(footprint_bytes / bytes_per_load) - 1u;
  const unsigned grid = (unsigned)(n_lanes / 256);
  for (int r = 0; r <= repeats; ++r) {        // r == 0: warm-up
    if (r == 1) IMMOCO_CHECK_HIP(hipEventRecord(e0, st));
    static const char* pat = immoco_diag_env("IMMOCO_PROBE_PATTERN");
    if (pat && bytes_per_load == 16)
      probe_hash_kernel<float4><<<grid, 256, 0, st>>>(reinterpret_cast<const float4*>(table), mask, (float)atof(pat), out);
    else if (pat)
      probe_hash_kernel<float2><<<grid, 256, 0, st>>>(reinterpret_cast<const float2*>(table), mask, (float)atof(pat), out);
    else if (bytes_per_load == 16)
      probe_gather_kernel<float4><<<grid, 256, 0, st>>>(reinterpret_cast<const float4*>(table), mask, loads_per_lane, out);
    else
      probe_gather_kernel<float2><<<grid, 256, 0, st>>>(reinterpret_cast<const float2*>(table), mask, loads_per_lane, out);
  }
  IMMOCO_LAUNCH_CHECK();
  IMMOCO_CHECK_HIP(hipEventRecord(e1, st));
  IMMOCO_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  IMMOCO_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / (float)repeats;
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(table);
  hipFree(out);
  return IMMOCO_OK;
}
