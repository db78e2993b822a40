// Micro-benchmark (GPU box): throughput of ds_add_f32 (LDS float atomic add, no return) per CU as a function
// of address pattern and active lanes.  Build: hipcc -O3 --offload-arch=gfx950 tools/bench_lds_atomic.hip -o tools/_bin/bench_lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t pcg(uint32_t x) {
  uint32_t s = x * 747796405u + 2891336453u;
  uint32_t w = ((s >> ((s >> 28u) + 4u)) ^ s) * 277803737u;
  return (w >> 22u) ^ w;
}
// mode 0: lane -> distinct consecutive words; 1: random word in 4096; 2: all lanes same word;
// 3: random, only lanes < active; 4: plain LDS store (no atomic) random; 5: ds_add_u32 random
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int reps, int active) {
  __shared__ float acc[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) acc[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  uint32_t h = pcg(blockIdx.x * 256 + threadIdx.x);
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      h = h * 1664525u + 1013904223u;
      uint32_t a = MODE == 0 ? (uint32_t)(threadIdx.x + 256 * (u & 7)) : MODE == 2 ? (uint32_t)(u * 17) : (h >> 20);
      a &= 4095u;
      if (MODE == 3) { if (lane < active) atomicAdd(&acc[a], 1.0f); }
      else if (MODE == 4) acc[a] = 1.0f;
      else if (MODE == 5) atomicAdd(reinterpret_cast<uint32_t*>(&acc[a]), 1u);
      else atomicAdd(&acc[a], 1.0f);
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = acc[threadIdx.x];
}
template <int MODE>
void run(const char* name, int active = 64) {
  const int blocks = 256 * 8, reps = 512;
  float* out;
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, reps, active);
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, reps, active);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double winstr = (double)blocks * 4 * reps * 8;            // wave instructions
  const double clk_per_cu = ms * 1e-3 * 2.4e9;                    // nominal 2.4 GHz
  printf("%-34s %8.3f ms  %7.1f clk per wave-instr per CU  (%.2f clk per active lane)\n", name, ms,
         clk_per_cu / (winstr / 256.0), clk_per_cu / (winstr / 256.0) / active);
  hipFree(out);
}
int main() {
  run<0>("ds_add_f32 conflict-free");
  run<1>("ds_add_f32 random of 4096 words");
  run<2>("ds_add_f32 all lanes one word");
  run<3>("ds_add_f32 random, 8 active lanes", 8);
  run<3>("ds_add_f32 random, 1 active lane", 1);
  run<3>("ds_add_f32 random, 32 active lanes", 32);
  run<4>("ds_write_b32 random");
  run<5>("ds_add_u32 random");
  return 0;
}
