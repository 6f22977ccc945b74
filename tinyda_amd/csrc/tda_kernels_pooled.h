// Pooled sample moments (extension: one AdaptiveMetropolis covariance for all chains and GPUs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tda_kernels_mh.h"

namespace tda {

// ------------------------------------------------------------------------------------------------
// Pooled sample moments of recorded states (extension, not in tinyDA: one AdaptiveMetropolis covariance shared by
// every chain on every GPU).  out = [count, sum x (d), sum x x^T (d x d)] over rows [0, nrows) of a row-major
// [nrows][d] matrix; two deterministic stages (fixed 512-row chunks, ordered accumulation), so the result is the
// same however the rows are later all-reduced across ranks.
// ------------------------------------------------------------------------------------------------
constexpr int MOM_CHUNK = 512;
template <int DPAD>
__global__ void __launch_bounds__(64) k_moments_partial(const double* __restrict__ x, int64_t nrows, int d,
                                                        double* __restrict__ partial) {
  __shared__ double s_x[DPAD];
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t lo = b * MOM_CHUNK, hi = lo + MOM_CHUNK < nrows ? lo + MOM_CHUNK : nrows;
  double s1 = 0.0, S[DPAD];
#pragma unroll
  for (int i = 0; i < DPAD; ++i) S[i] = 0.0;
  for (int64_t r = lo; r < hi; ++r) {
    const double xj = lane < d ? x[(size_t)r * d + lane] : 0.0;
    __syncthreads();
    if (lane < DPAD) s_x[lane] = xj;
    __syncthreads();
    s1 += xj;
#pragma unroll
    for (int i = 0; i < DPAD; ++i) S[i] = fma(s_x[i], xj, S[i]);
  }
  double* o = partial + (size_t)b * (DPAD + DPAD * DPAD);
  if (lane < DPAD) {
    o[lane] = s1;
#pragma unroll
    for (int i = 0; i < DPAD; ++i) o[DPAD + (size_t)i * DPAD + lane] = S[i];
  }
}

template <int DPAD>
__global__ void __launch_bounds__(64) k_moments_final(const double* __restrict__ partial, int64_t nb, int64_t nrows, int d,
                                                      double* __restrict__ out) {
  const int lane = threadIdx.x;
  const int row = blockIdx.x;  // 0: sum x, 1 + i: row i of sum x x^T
  if (lane >= d || row > d) return;
  double acc = 0.0;
  for (int64_t b = 0; b < nb; ++b) {
    const double* o = partial + (size_t)b * (DPAD + DPAD * DPAD);
    acc += row == 0 ? o[lane] : o[DPAD + (size_t)(row - 1) * DPAD + lane];
  }
  if (row == 0) {
    out[1 + lane] = acc;
    if (lane == 0) out[0] = (double)nrows;
  } else {
    out[1 + d + (size_t)(row - 1) * d + lane] = acc;
  }
}

}  // namespace tda
