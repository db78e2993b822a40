/* CPU ORACLE helper - TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of the interpolation half of tiny-cuda-nn's grid encoding (kernel_grid /
 * kernel_grid_backward of include/tiny-cuda-nn/encodings/grid.h upstream; un-vendored dependency of the
 * reference, call sites /root/reference/src/models/immoco.py:60-65,85,93) for the
 * fixed lattices the reference queries.  The integer half (cell, hash / dense index, corner weights) is
 * computed once per lattice by oracle/immoco_oracle.py:HashGridPlan with numpy; this file only evaluates
 *
 *     enc[n][l][f]  = sum_c  table[idx[l][n][c]][f] * w[l][n][c]          (corner order, fp32, mul then add)
 *     dtable[i][f] += w[l][n][c] * denc[n][l][f]   for i = idx[l][n][c]   (the transpose of the above)
 *
 * which is exactly what the torch expression in HashGridPlan.encode() and its autograd compute; the torch
 * path stays available (OracleINR(..., backend="torch")) and tests/test_oracle_hashgrid.py checks this file
 * against it (forward bit-exact, backward to rounding).  It exists because the torch expression spends 75 %
 * of an oracle iteration in index/index_put/mul temporaries (7.3 s per C2 iteration on 4 cores), which made
 * full-size oracle records cost hours.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg (through oracle/immoco_oracle.py) may
 * load it.  Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 */
#include <stdint.h>
#include <stddef.h>

/* table [E][2], idx [L][N][C] (absolute entry index), w [L][N][C], enc [N][L][2] */
void hg_encode_fwd(const float* table, const int32_t* idx, const float* w, int64_t N, int32_t L, int32_t C,
                   float* enc) {
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < N; ++n) {
    for (int32_t l = 0; l < L; ++l) {
      const int32_t* ii = idx + ((int64_t)l * N + n) * C;
      const float* ww = w + ((int64_t)l * N + n) * C;
      float a0 = 0.f, a1 = 0.f;
      for (int32_t c = 0; c < C; ++c) {
        const float t0 = table[2 * (int64_t)ii[c]] * ww[c];
        const float t1 = table[2 * (int64_t)ii[c] + 1] * ww[c];
        a0 = c == 0 ? t0 : a0 + t0;
        a1 = c == 0 ? t1 : a1 + t1;
      }
      enc[(n * L + l) * 2] = a0;
      enc[(n * L + l) * 2 + 1] = a1;
    }
  }
}

/* dtable [E][2] (accumulated into), denc [N][L][2].  Levels own disjoint ranges of the table, so one thread
 * per level is race-free and the summation order is fixed by `order`:
 *   0: points ascending; 1: points descending; k >= 2: blocks of 4096 points visited with stride k (a
 *   different but equally valid fp32 summation order - used to draw independent trajectories of the
 *   chaotic optimisation, like the nondeterministic atomics of tiny-cuda-nn / torch's index_put do). */
void hg_encode_bwd(const float* denc, const int32_t* idx, const float* w, int64_t N, int32_t L, int32_t C,
                   float* dtable, int32_t order) {
  const int64_t BL = 4096;
  const int64_t nb = (N + BL - 1) / BL;
  int64_t m = nb;  /* order >= 2: smallest m >= nb coprime with the stride, so b = bi*k mod m is a bijection */
  if (order >= 2) {
    for (;;) {
      int64_t x = m, y = order;
      while (y) { const int64_t t = x % y; x = y; y = t; }
      if (x == 1) break;
      ++m;
    }
  }
#pragma omp parallel for schedule(dynamic, 1)
  for (int32_t l = 0; l < L; ++l) {
    for (int64_t bi = 0; bi < m; ++bi) {
      const int64_t b = order == 0 ? bi : order == 1 ? nb - 1 - bi : (bi * (int64_t)order) % m;
      if (b >= nb) continue;
      const int64_t n0 = b * BL, n1 = n0 + BL < N ? n0 + BL : N;
      for (int64_t nn = n0; nn < n1; ++nn) {
        const int64_t n = order == 1 ? n1 - 1 - (nn - n0) : nn;
        const int32_t* ii = idx + ((int64_t)l * N + n) * C;
        const float* ww = w + ((int64_t)l * N + n) * C;
        const float d0 = denc[(n * L + l) * 2], d1 = denc[(n * L + l) * 2 + 1];
        for (int32_t c = 0; c < C; ++c) {
          dtable[2 * (int64_t)ii[c]] += ww[c] * d0;
          dtable[2 * (int64_t)ii[c] + 1] += ww[c] * d1;
        }
      }
    }
  }
}

int32_t hg_oracle_version(void) { return 1; }
