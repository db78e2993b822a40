// Probe (MI355X): operand maps the fp16 MLP kernels (csrc/mlp_f16.hip) rely on, checked with exact integer data.
//   1. v_mfma_f32_32x32x16_f16: A[row r][k = 8h + i], B[k = 8h + i][col r], D reg g = D[(g&3) + 8(g>>2) + 4h][r]
//   2. ds_read_b64_tr_b16: lane 4q+p of a 16-lane group supplies the address of row q, columns 4p..4p+3 of a
//      4 x 16 block; lane i of the group receives column i, row q in element q
//   3. an accumulator tile X (rows in registers) converted to fp16 as the B operand of the next MFMA (Y = A.X):
//      element i of lane half h of k-step s is row 16s + 8(i>>2) + 4h + (i&3) of X
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma_f16.hip -o tools/_bin/probe_mfma_f16 && tools/_bin/probe_mfma_f16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __fp16 fh4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k_mfma(const float* A /*[32][16]*/, const float* B /*[16][32]*/, float* D /*[32][32]*/) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  h8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)A[r * 16 + 8 * h + i];
    b[i] = (_Float16)B[(8 * h + i) * 32 + r];
  }
  f32x16 c = {0.f};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int g = 0; g < 16; ++g) D[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = c[g];
}

// T [32 rows][32 cols] fp16 image, 64-byte rows.  out[lane][4]: block rows r0..r0+3 with r0 = 4*(lane>>5),
// columns 16*((lane>>4)&1) + (lane&15)
__global__ void k_tr(const float* T, float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 t[32 * 32];
  for (int i = threadIdx.x; i < 1024; i += 64) t[i] = (_Float16)T[i];
  __syncthreads();
  const int l = threadIdx.x, grp = l >> 4, li = l & 15, q = li >> 2, p = li & 3;
  const int r0 = 4 * (l >> 5), c0 = 16 * (grp & 1);
  const _Float16* addr = t + (r0 + q) * 32 + c0 + 4 * p;
  fh4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fh4*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)v[e];
}

// Y[32][32] = A2[32][32] . X[32][32] with X = A1 . B1 computed by a first MFMA and fed from the accumulator
__global__ void k_chain(const float* A1 /*[32][16]*/, const float* B1 /*[16][32]*/, const float* A2 /*[32][32]*/, float* Y) {
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  h8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)A1[r * 16 + 8 * h + i];
    b[i] = (_Float16)B1[(8 * h + i) * 32 + r];
  }
  f32x16 x = {0.f};
  x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, x, 0, 0, 0);
  f32x16 y = {0.f};
  for (int s = 0; s < 2; ++s) {
    h8 xb, a2;
    for (int i = 0; i < 8; ++i) {
      xb[i] = (_Float16)x[8 * s + i];
      a2[i] = (_Float16)A2[r * 32 + 16 * s + 8 * (i >> 2) + 4 * h + (i & 3)];
    }
    y = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, xb, y, 0, 0, 0);
  }
  for (int g = 0; g < 16; ++g) Y[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = y[g];
}

// 4. fp16 subnormals: does v_cvt_pk_f16_f32 produce them, does the MFMA consume them?  x = 2^-20 (a subnormal
//    half), y = 2^10: out[0] = bits of half(x), out[1] = (float)half(x), out[2] = D[0][0] of A = x, B = y (k = 0)
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_denorm(float x, float y, float* out) {
  const f2 v = {x, x};
  const h2 hv = __builtin_convertvector(v, h2);
  const int l = threadIdx.x;
  h8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  if (l < 32) { a[0] = hv[0]; b[0] = (_Float16)y; }
  f32x16 c = {0.f};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (l == 0) {
    const _Float16 h0 = hv[0];
    unsigned short bits;
    __builtin_memcpy(&bits, &h0, 2);
    out[0] = (float)bits;
    out[1] = (float)hv[0];
    out[2] = c[0];
    out[3] = (float)hv[0] * y;   // VALU product of the converted value
  }
}

int main() {
  std::vector<float> A(512), B(512), D(1024), T(1024), O(256), A2(1024), Y(1024);
  for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7 + 3) % 11 - 5); B[i] = (float)((i * 5 + 1) % 13 - 6); }
  for (int i = 0; i < 1024; ++i) { T[i] = (float)i; A2[i] = (float)((i * 3 + 2) % 7 - 3); }
  float *dA, *dB, *dD, *dT, *dO, *dA2, *dY;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096); hipMalloc(&dT, 4096); hipMalloc(&dO, 1024);
  hipMalloc(&dA2, 4096); hipMalloc(&dY, 4096);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  hipMemcpy(dT, T.data(), 4096, hipMemcpyHostToDevice); hipMemcpy(dA2, A2.data(), 4096, hipMemcpyHostToDevice);
  k_mfma<<<1, 64>>>(dA, dB, dD);
  k_tr<<<1, 64>>>(dT, dO);
  k_chain<<<1, 64>>>(dA, dB, dA2, dY);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
  hipMemcpy(Y.data(), dY, 4096, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0, bad3 = 0;
  std::vector<float> X(1024);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 32 + j];
    X[i * 32 + j] = s; bad1 += D[i * 32 + j] != s;
  }
  for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) {
    const int r0 = 4 * (l >> 5), c = 16 * ((l >> 4) & 1) + (l & 15);
    bad2 += O[l * 4 + e] != T[(r0 + e) * 32 + c];
  }
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    float s = 0; for (int k = 0; k < 32; ++k) s += A2[i * 32 + k] * X[k * 32 + j];
    bad3 += Y[i * 32 + j] != s;
  }
  {
    float* dd; hipMalloc(&dd, 16); float hh[4];
    k_denorm<<<1, 64>>>(9.5367431640625e-07f, 1024.f, dd);
    hipMemcpy(hh, dd, 16, hipMemcpyDeviceToHost);
    printf("fp16 subnormal 2^-20: cvt bits 0x%04x (expect 0x0010), back to float %g (expect 9.53674e-07), MFMA x*1024 = %g, VALU %g (expect 0.000976562)\n",
           (unsigned)hh[0], hh[1], hh[2], hh[3]);
  }
  printf("mfma_f16 operand map: %s (%d bad)\nds_read_b64_tr_b16 map: %s (%d bad)\naccumulator as B operand: %s (%d bad)\n",
         bad1 ? "FAIL" : "PASS", bad1, bad2 ? "FAIL" : "PASS", bad2, bad3 ? "FAIL" : "PASS", bad3);
  if (bad2) for (int l = 0; l < 20; ++l) printf("lane %d: %g %g %g %g\n", l, O[l*4], O[l*4+1], O[l*4+2], O[l*4+3]);
  return bad1 + bad2 + bad3 ? 1 : 0;
}
