// Line-select masks: run-length labelling of corrupted phase-encode lines
// (reference src/utils/motion_utils.py:56-109), integer / bit-exact.
// A True line i gets label 1 + #{j < i : v[j] && !v[j+1]} (falling edges strictly
// before it); False lines get 0.  n is a few hundred, so one workgroup with an
// LDS prefix scan does it.
#include "kernels.hpp"

namespace immoco {

__global__ __launch_bounds__(1024) void extract_groups_kernel(const uint8_t* __restrict__ lines, int n,
                                                              int32_t* __restrict__ col_group,
                                                              int32_t* __restrict__ n_groups) {
  extern __shared__ int32_t scan[];  // n ints
  // falling edge flag at j: v[j] && (j == n-1 ? 0 : !v[j+1])
  for (int j = threadIdx.x; j < n; j += blockDim.x)
    scan[j] = (lines[j] != 0 && j != n - 1 && lines[j + 1] == 0) ? 1 : 0;
  __syncthreads();
  // inclusive Hillis-Steele scan over n (<= a few thousand) with double stepping
  for (int off = 1; off < n; off <<= 1) {
    int32_t add[8];
    int cnt = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) add[cnt++] = j >= off ? scan[j - off] : 0;
    __syncthreads();
    cnt = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) scan[j] += add[cnt++];
    __syncthreads();
  }
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const int32_t before = j > 0 ? scan[j - 1] : 0;  // falling edges strictly before j
    col_group[j] = lines[j] != 0 ? before + 1 : 0;
  }
  if (threadIdx.x == 0) {
    // number of runs = falling edges + (last line True ? 1 : 0)
    n_groups[0] = scan[n - 1] + (lines[n - 1] != 0 ? 1 : 0);
  }
}

int launch_extract_groups(const uint8_t* lines, int n, int32_t* col_group, int32_t* n_groups, hipStream_t st) {
  IMMOCO_REQUIRE(n >= 1 && n <= 8192, "extract_movement_groups: n=%d out of range [1,8192]", n);
  extract_groups_kernel<<<1, 1024, (size_t)n * sizeof(int32_t), st>>>(lines, n, col_group, n_groups);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

__global__ __launch_bounds__(256) void groups_to_matrix_kernel(const int32_t* __restrict__ col_group, int64_t total,
                                                               int n, int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  out[i] = (int64_t)col_group[i % n];
}

__global__ __launch_bounds__(256) void groups_to_masks_kernel(const int32_t* __restrict__ col_group, int64_t total,
                                                              int64_t per_mask, int n,
                                                              int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int g = (int)(i / per_mask) + 1;
  out[i] = col_group[i % n] == g ? 1 : 0;
}

__global__ __launch_bounds__(256) void masks_to_groups_kernel(const int64_t* __restrict__ masks, int nM,
                                                              int64_t per_mask, int n,
                                                              int32_t* __restrict__ col_group) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  int g = 0;
  for (int m = 0; m < nM; ++m)
    if (masks[(int64_t)m * per_mask + c] != 0) g = m + 1;  // row 0 of each mask
  col_group[c] = g;
}

}  // namespace immoco

using namespace immoco;

extern "C" int immoco_extract_movement_groups(const uint8_t* lines, int32_t n, int32_t* col_group,
                                              int32_t* n_groups, void* stream) {
  IMMOCO_REQUIRE(lines && col_group && n_groups, "extract_movement_groups: NULL buffer");
  return launch_extract_groups(lines, n, col_group, n_groups, as_stream(stream));
}

extern "C" int immoco_groups_to_matrix(const int32_t* col_group, int32_t rows, int32_t n, int64_t* groups,
                                       void* stream) {
  IMMOCO_REQUIRE(col_group && groups && rows >= 1 && n >= 1, "groups_to_matrix: bad argument");
  const int64_t total = (int64_t)rows * n;
  groups_to_matrix_kernel<<<(unsigned)cdiv(total, 256), 256, 0, as_stream(stream)>>>(col_group, total, n, groups);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

extern "C" int immoco_groups_to_masks(const int32_t* col_group, int32_t n_groups, int32_t rows, int32_t n,
                                      int64_t* masks, void* stream) {
  IMMOCO_REQUIRE(col_group && rows >= 1 && n >= 1 && n_groups >= 0, "groups_to_masks: bad argument");
  if (n_groups == 0) return IMMOCO_OK;
  IMMOCO_REQUIRE(masks != nullptr, "groups_to_masks: NULL masks");
  const int64_t per = (int64_t)rows * n, total = per * n_groups;
  groups_to_masks_kernel<<<(unsigned)cdiv(total, 256), 256, 0, as_stream(stream)>>>(col_group, total, per, n, masks);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}

extern "C" int immoco_masks_to_groups(const int64_t* masks, int32_t nM, int32_t rows, int32_t n,
                                      int32_t* col_group, void* stream) {
  IMMOCO_REQUIRE(col_group && rows >= 1 && n >= 1 && nM >= 0 && (nM == 0 || masks), "masks_to_groups: bad argument");
  masks_to_groups_kernel<<<(unsigned)cdiv(n, 256), 256, 0, as_stream(stream)>>>(masks, nM, (int64_t)rows * n, n,
                                                                            col_group);
  IMMOCO_LAUNCH_CHECK();
  return IMMOCO_OK;
}
