// Cholesky kernel experiments for tools/chol_microbench.hip (not part of the library).
//   k_chol_x<DPAD, RS, WPE>: k_chol with (RS) the pivot column scaled by a Newton-refined reciprocal square root instead of
//                            sqrt + division, and (WPE) a waves-per-SIMD target for the register allocator
//   k_chol_2t<DPAD>:         two lanes per row (columns split even / odd), i.e. two waves per chain, half the registers
#pragma once
#include "tda_kernels_mh.h"

namespace tda {

template <int DPAD, int RS, int WPE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_chol_x(const CholArgs a) {
  constexpr int T = am_tile_rows<DPAD>();
  constexpr int NTL = am_tiles<DPAD>();
  constexpr int CHOL_WAVES = 4;
  constexpr int CW = DPAD >= 16 ? DPAD : 16;
  __shared__ __attribute__((aligned(16))) double s_tile[CHOL_WAVES][16 * 17];
  __shared__ __attribute__((aligned(16))) double s_col[CHOL_WAVES][CW];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t c = (int64_t)blockIdx.x * CHOL_WAVES + w;
  if (c >= a.N) return;
  const bool lj = lane < a.d;
  const int li = lane < DPAD ? lane : DPAD - 1;
  double A[DPAD];
#pragma unroll
  for (int j = 0; j < DPAD; ++j) A[j] = 0.0;
  const double* sg = a.am_sigma + (size_t)c * NTL * 256;
  double* st = s_tile[w];
#pragma unroll
  for (int ti = 0; ti < T; ++ti) {
#pragma unroll
    for (int tj = 0; tj <= ti; ++tj) {
      const int tile = ti * (ti + 1) / 2 + tj;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) st[(r * 4 + (lane >> 4)) * 17 + (lane & 15)] = sg[tile * 256 + r * 64 + lane];
      __builtin_amdgcn_wave_barrier();
      if ((li >> 4) == ti && lane < DPAD) {
#pragma unroll
        for (int cc = 0; cc < 16; ++cc)
          if (16 * tj + cc < DPAD) A[16 * tj + cc] = st[(li & 15) * 17 + cc];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < DPAD; ++j)
    if (!(lj && j < a.d && j <= li)) A[j] = (j == li) ? 1.0 : 0.0;
  double* sc = s_col[w];
  bool ok = true;
#pragma unroll
  for (int k = 0; k < DPAD; ++k) {
    const double dkk = bcast_lane<DPAD>(A[k], k);
    ok = ok && (dkk > 0.0);
    double lik;
    if (RS) {
      double r = __builtin_amdgcn_rsq(dkk);            // ~2^-26 relative
      const double h = 0.5 * dkk;
      r = fma(r, fma(-h * r, r, 0.5), r);              // Newton: r <- r + r (1/2 - h r^2)
      r = fma(r, fma(-h * r, r, 0.5), r);
      lik = A[k] * r;                                   // row k: d / sqrt(d) = sqrt(d)
    } else {
      const double lkk = sqrt(dkk);
      lik = (li == k) ? lkk : A[k] / lkk;
    }
    A[k] = lik;
    __builtin_amdgcn_wave_barrier();
    if (lane < DPAD) sc[lane] = lik;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = k + 1; j < DPAD; ++j) A[j] = fma(-lik, sc[j], A[j]);
  }
  if (ok) {
    if (lane < DPAD) {
#pragma unroll
      for (int k = 0; k < DPAD; ++k) {
        const double v = (lj && k < a.d && li >= k) ? A[k] : 0.0;
        a.Lk[((size_t)c * DPAD + k) * DPAD + lane] = v;
      }
    }
  } else if (lane == 0) {
    atomicOr(&a.flags[c], 1);
  }
}

// two lanes per row: thread t of a 128-thread group: row = t & 63, half = t >> 6 owns columns j with (j & 1) == half
template <int DPAD, int RS>
__global__ void __launch_bounds__(256) k_chol_2t(const CholArgs a) {
  static_assert(DPAD == 64, "experiment at DPAD = 64 only");
  constexpr int NTL = am_tiles<DPAD>();
  constexpr int H = DPAD / 2;
  __shared__ __attribute__((aligned(16))) double s_col[2][2][DPAD];  // [chain in block][double buffer][row]
  const int t = threadIdx.x & 127, g = threadIdx.x >> 7;
  const int row = t & 63, half = t >> 6;
  const int64_t c = (int64_t)blockIdx.x * 2 + g;
  const bool live = c < a.N;
  const int64_t cc = live ? c : a.N - 1;
  const bool lj = row < a.d;
  double A[H];  // A[q] = column 2 q + half of row `row`
  const double* sg = a.am_sigma + (size_t)cc * NTL * 256;
#pragma unroll
  for (int q = 0; q < H; ++q) {
    const int j = 2 * q + half;
    double v = (j == row) ? 1.0 : 0.0;
    if (lj && j < a.d && j <= row) v = sg[am_sigma_offset(row, j)];
    A[q] = v;
  }
  bool ok = true;
#pragma unroll
  for (int k = 0; k < DPAD; ++k) {
    double* sc = s_col[g][k & 1];
    if ((k & 1) == half) {  // the half that owns column k scales it and publishes it
      const double dkk = bcast_lane<DPAD>(A[k >> 1], k);
      ok = ok && (dkk > 0.0);
      double lik;
      if (RS) {
        double r = __builtin_amdgcn_rsq(dkk);
        const double h = 0.5 * dkk;
        r = fma(r, fma(-h * r, r, 0.5), r);
        r = fma(r, fma(-h * r, r, 0.5), r);
        lik = A[k >> 1] * r;
      } else {
        const double lkk = sqrt(dkk);
        lik = (row == k) ? lkk : A[k >> 1] / lkk;
      }
      A[k >> 1] = lik;
      sc[row] = lik;
    }
    __syncthreads();
    const double lik = sc[row];
#pragma unroll
    for (int q = 0; q < H; ++q) {
      const int j = 2 * q + half;  // compile-time parity unknown: predicate on j > k
      if (2 * q + 1 > k) {         // q's columns 2q, 2q+1: at least one may be > k
        const double ljk = sc[2 * q + half];
        if (j > k) A[q] = fma(-lik, ljk, A[q]);
      }
    }
  }
  // flags / stores
  if (!ok && live && row == 0) atomicOr(&a.flags[c], 1);
  __shared__ int s_bad[2];
  if (threadIdx.x < 2) s_bad[threadIdx.x] = 0;
  __syncthreads();
  if (!ok) s_bad[g] = 1;
  __syncthreads();
  if (live && !s_bad[g]) {
#pragma unroll
    for (int q = 0; q < H; ++q) {
      const int k = 2 * q + half;
      const double v = (lj && k < a.d && row >= k) ? A[q] : 0.0;
      a.Lk[((size_t)c * DPAD + k) * DPAD + row] = v;
    }
  }
}

// Round 2: the same factorisation as a real loop over the columns -- the row shifts by one register per column (the update of
// element j lands in register j - 1), so the active column is always register 0; four bodies of decreasing width.  450
// instructions instead of 12 000, columns staged in LDS (packed lower triangle) and written out at the end.  SKIP bit 0: no
// loads of Sigma (synthetic rows), bit 1: no stores of L -- wrong results, to see what a launch spends on HBM.
template <int DPAD>
__host__ __device__ constexpr int chol_lds_doubles() {
  int n = 0;
  for (int k = 0; k < DPAD; ++k) n += (DPAD - k + 1) & ~1;
  return n + 16;
}
template <int DPAD, int W>
__device__ __forceinline__ void chol_phase(double (&A)[DPAD], double* __restrict__ s_L, int& off, int k0, int lane, int li, bool& ok) {
  constexpr int NK = W < 16 ? W : 16;
#pragma unroll 1
  for (int kk = 0; kk < NK; ++kk) {
    const int k = k0 + kk;
    const double dkk = bcast_lane<DPAD>(A[0], k);
    ok = ok && (dkk > 0.0);
    const double lkk = sqrt(dkk);
    const double lik = (li == k) ? lkk : A[0] / lkk;
    if (lane >= k && lane < DPAD) s_L[off + lane - k] = lik;
    __syncthreads();
    const double2* __restrict__ col = reinterpret_cast<const double2*>(s_L + off);
#pragma unroll
    for (int p = 0; p < W / 2; ++p) {
      const double2 l2 = col[p];
      if (p > 0) A[2 * p - 1] = fma(-lik, l2.x, A[2 * p]);
      A[2 * p] = fma(-lik, l2.y, A[2 * p + 1 < DPAD ? 2 * p + 1 : 2 * p]);
    }
    off += (DPAD - k + 1) & ~1;
  }
}
template <int DPAD, int SKIP>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2))) k_chol_loop(const CholArgs a) {
  constexpr int NTL = am_tiles<DPAD>();
  __shared__ __attribute__((aligned(16))) double s_L[chol_lds_doubles<DPAD>()];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const bool lj = lane < a.d;
  const int li = lane < DPAD ? lane : DPAD - 1;
  double A[DPAD];
#pragma unroll
  for (int j = 0; j < DPAD; ++j) {
    double v = (j == li) ? 1.0 : 0.0;
    if (!(SKIP & 1)) {
      if (lj && j < a.d && j <= li) v = a.am_sigma[(size_t)c * NTL * 256 + am_sigma_offset(li, j)];
    } else if (j == li) {
      v = 2.0 + 1e-3 * (double)(c & 7);
    }
    A[j] = v;
  }
  bool ok = true;
  int off = 0;
  chol_phase<DPAD, DPAD>(A, s_L, off, 0, lane, li, ok);
  if constexpr (DPAD > 16) chol_phase<DPAD, DPAD - 16>(A, s_L, off, 16, lane, li, ok);
  if constexpr (DPAD > 32) chol_phase<DPAD, DPAD - 32>(A, s_L, off, 32, lane, li, ok);
  if constexpr (DPAD > 48) chol_phase<DPAD, DPAD - 48>(A, s_L, off, 48, lane, li, ok);
  if (ok) {
    if (lane < DPAD) {
      off = 0;
      double acc = 0.0;
#pragma unroll 4
      for (int k = 0; k < DPAD; ++k) {
        const double v = (lj && k < a.d && lane >= k) ? s_L[off + lane - k] : 0.0;
        if (!(SKIP & 2)) a.Lk[((size_t)c * DPAD + k) * DPAD + lane] = v;
        else acc += v;
        off += (DPAD - k + 1) & ~1;
      }
      if ((SKIP & 2) && acc == 12345.678) a.Lk[c] = acc;
    }
  } else if (lane == 0) {
    atomicOr(&a.flags[c], 1);
  }
}

}  // namespace tda
