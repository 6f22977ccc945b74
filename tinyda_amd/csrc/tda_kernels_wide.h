// Single-level chains with 65 .. 128 parameters (round 5, VERDICT r4 item 4: "lift d <= 64 to d <= 128").
//
// The step kernel is the same template (k_mh_steps<128, 4>: 32 k-slices per 16-row block, one 512-register wave per SIMD), as are the
// generators (k_rng<128>, k_rng_direct<128>, k_rng_uniforms).  What does not stretch to 128 is everything that kept a 64 x 64 matrix
// in one wave's registers lane = row / lane = parameter: the AdaptiveMetropolis moment recursion (k_adapt: Sigma AND t mu mu^T as
// 2 x 10 tiles), the row-per-lane Cholesky (k_chol: lane = row), the product INC = Z L^T with the whole factor as B fragments
// (k_apply: 64 x 16 doubles per lane at 128).  Their 128-parameter forms, on the tile machinery of the dense error model
// (tda_kernels_aemr.h):
//   k_wide_adapt   RecursiveSampleMoments.update (utils.py:113-124) per recorded state in the reference's order of operations, Sigma as
//                  its 36 UPPER 16 x 16 tiles in the MFMA C/D layout in 288 registers (aemr_ut order: the layout k_aem_refresh factors);
//                  the global scaling adaptation (proposal.py:228-245) rides along as in k_adapt;
//   k_aem_refresh<8, 1> with AemRefreshArgs::wide set: C <- Sigma (proposal.py:509-510) = the blocked left-looking Cholesky of the
//                  error model, no 1e-9 rule, the factors of the diagonal tiles stored beside their inverses, a pivot <= 0 flags the
//                  chain and leaves its previous factor in place (two factor buffers per chain, a selector says which one is current);
//   k_wide_apply   INC = Z L^T for a block of steps on the matrix cores, the B fragments read straight from the factor's tiles (row
//                  16 p + 4 r + hi of a tile is k = 4 kk + hi with kk = 4 p + r: a finished tile IS the fragment, as k_chol_apply_blk
//                  notes for 64 parameters), eight tile columns in two halves of four, the zero tiles below the block diagonal skipped.
// Layouts per chain: factor buffer b in {0, 1}: [b][NP][36 tiles][4][64] in the factor form's tile order (tile (i, p), p < i: U_pi at
// aemr_lt(i, p); the diagonal slots hold the inverses, unused here) and Ud [b][NP][8][4][64] (U_pp, upper triangular, zero below);
// sel[NP]: the current buffer.  A SHARED factor (GaussianRandomWalk, CrankNicolson under a non-identity prior): one chain's worth, stride 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tda_kernels_aemr.h"
#include "tda_kernels_mh.h"

namespace tda {

constexpr int WIDE_T = 8;                       // tile rows at 128 parameters
constexpr int WIDE_NT = aemr_tiles(WIDE_T);      // 36
constexpr size_t WIDE_FACTOR_DOUBLES = (size_t)WIDE_NT * 256;  // factor-form tiles of one chain
constexpr size_t WIDE_UD_DOUBLES = (size_t)WIDE_T * 256;       // its diagonal tiles U_pp

// ---- AdaptiveMetropolis moments (utils.py:113-124), the reference's operation order; Sigma: upper tiles (aemr_ut) in registers ----
//   mu'    = (1 / (t + 1)) (t mu + x)
//   Sigma' = (t - 1) / t Sigma + sd / t ((t (mu_i mu_j) - (t + 1) (mu'_i mu'_j)) + x_i x_j [+ eps on the diagonal])
// (k_adapt carries (t + 1) mu' mu'^T over as the next state's t mu mu^T -- the same operands, the same rounding -- in 80 more
// registers; at 36 tiles there is no room for a second matrix, so the products are formed again: the same values.)
// Two waves per chain, 18 tiles each (tile rows {0, 3, 4, 7} and {1, 2, 5, 6}: 8 + 5 + 4 + 1 = 7 + 6 + 3 + 2): 144 registers of Sigma
// per wave, two waves per SIMD.  (First version: one wave with all 36 tiles -- 288 registers plus operands, 240 bytes of scratch
// at 512 registers, the single-wave issue rate: 2.2 ms per 100 states of 4096 chains.)
template <int T, int R0, int R1, int R2, int R3>
__device__ __forceinline__ void wide_adapt_rows(const AdaptArgs& a, const int64_t c, const int lane, const int tid, double* s_x, double* s_m,
                                                double* s_p) {
  constexpr int NT = aemr_tiles(T), W = 16 * T;
  constexpr int ROWS[4] = {R0, R1, R2, R3};
  constexpr int NMINE = (T - R0) + (T - R1) + (T - R2) + (T - R3);
  const int lc = lane & 15, hi = lane >> 4;
  // (descriptor + one lane offset + compile-time scalar offsets: 64-bit addresses per tile row would stay alive across the state loop)
  const __amdgpu_buffer_rsrc_t srs = aemr_rsrc(a.am_sigma + (size_t)c * NT * 256);
  double Sg[NMINE][4];
  {
    int k = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = ROWS[q]; i < T; ++i, ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) Sg[k][r] = aemr_ld(srs, lane * 8, (aemr_ut(T, ROWS[q], i) * 4 + r) * 512);
  }
  double mu = a.am_mu[c * W + tid];  // thread tid of the 2 x 64 owns parameter tid
  for (int s = 0; s < a.S; ++s) {
    const double t = (double)(a.t_base + s + 1), t1 = t + 1.0;  // recursor.t before this update
    const double c_inv = 1.0 / t1, ca = (t - 1.0) / t, cb = a.sd / t;
    const double x = tid < a.d ? a.rec_params[((size_t)s * a.N + c) * a.d + tid] : 0.0;
    const double mup = c_inv * (t * mu + x);
    __syncthreads();  // the previous state's operand reads are done
    s_x[tid] = x;
    s_m[tid] = mu;
    s_p[tid] = mup;
    __syncthreads();
    int k = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = ROWS[q];
      double xr[4], mr[4], pr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xr[r] = s_x[16 * p + hi + 4 * r];
        mr[r] = s_m[16 * p + hi + 4 * r];
        pr[r] = s_p[16 * p + hi + 4 * r];
      }
#pragma unroll
      for (int i = p; i < T; ++i, ++k) {
        const double xc = s_x[16 * i + lc], mc = s_m[16 * i + lc], pc = s_p[16 * i + lc];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          double M = (t * (mr[r] * mc) - t1 * (pr[r] * pc)) + xr[r] * xc;
          if (p == i) M += (hi + 4 * r == lc) ? a.eps : 0.0;  // + eps on the diagonal, + 0 beside it (padded dimensions collect eps too; nothing reads them)
          Sg[k][r] = ca * Sg[k][r] + cb * M;
        }
        __builtin_amdgcn_sched_barrier(0);  // (a tile at a time: hoisted ahead, the operand reads of a whole state spill Sigma)
      }
    }
    mu = mup;
  }
  a.am_mu[c * W + tid] = mu;  // (thread tid of the two waves owns parameter tid)
  {
    int k = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = ROWS[q]; i < T; ++i, ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) aemr_st(Sg[k][r], srs, lane * 8, (aemr_ut(T, ROWS[q], i) * 4 + r) * 512);
  }
}

template <int T>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) k_wide_adapt(const AdaptArgs a) {
  static_assert(T == 8, "the tile rows are dealt to the two waves for eight tile rows");
  constexpr int W = 16 * T;
  __shared__ double s_x[W], s_m[W], s_p[W];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  if (a.do_am) {
    if (wave == 0) wide_adapt_rows<T, 0, 3, 4, 7>(a, c, lane, tid, s_x, s_m, s_p);
    else wide_adapt_rows<T, 1, 2, 5, 6>(a, c, lane, tid, s_x, s_m, s_p);
  }
  if (wave == 0) adapt_scaling(a, c, lane);
}

// ---- INC = Z L^T for a block of steps, one wave per chain ----
struct WideApplyArgs {
  int64_t NP;
  int S;
  const double* fac;     // [2][NPf][36][4][64] factor-form tiles (NPf = NP, or 1 for a shared factor)
  const double* ud;      // [2][NPf][8][4][64] diagonal tiles U_pp
  const int32_t* sel;    // [NP] current buffer of every chain (null: buffer 0)
  int64_t chain_stride;  // 1 = per-chain factors, 0 = one shared factor
  int64_t NPf;           // chains in a factor buffer (the stride between the two buffers)
  const double* zf;      // [groups][NP][32][64] MFMA A fragments of the normals (k_rng<128>)
  double* inc;           // [S][NP][128]
};

template <int T>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) k_wide_apply(const WideApplyArgs a) {
  constexpr int KK = 4 * T, DP = 16 * T, TH = T / 2;
  const int lane = threadIdx.x, lc = lane & 15, hi = lane >> 4;
  const int64_t c = blockIdx.x;
  const int64_t cf = c * a.chain_stride;
  const int b = a.sel ? a.sel[cf] : 0;
  const double* __restrict__ F = a.fac + ((size_t)b * a.NPf + cf) * WIDE_FACTOR_DOUBLES + lane;
  const double* __restrict__ D = a.ud + ((size_t)b * a.NPf + cf) * WIDE_UD_DOUBLES + lane;
  const int ng = (a.S + 15) / 16;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    // B fragments of tile columns tj = TH half .. TH half + TH - 1: Lf[tj][kk] = U[4 kk + hi][16 tj + lc] = tile (p = kk >> 2, tj), register kk & 3
    double Lf[TH][KK];
#pragma unroll
    for (int j = 0; j < TH; ++j) {
      const int tj = TH * half + j;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int p = kk >> 2, r = kk & 3;
        double v = 0.0;
        if (p < tj) v = F[(size_t)(aemr_lt(tj, p) * 4 + r) * 64];
        else if (p == tj) v = D[(size_t)(p * 4 + r) * 64];
        Lf[j][kk] = v;
      }
    }
    double zf[KK];
    {
      const double* __restrict__ src = a.zf + (size_t)c * KK * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) zf[kk] = kk < 4 * (TH * half + TH) ? src[kk * 64] : 0.0;
    }
    for (int g = 0; g < ng; ++g) {
      double zn[KK];
      {  // the next group's fragments fly during this group's matrix instructions (clamped: the last group re-reads itself)
        const int gn = g + 1 < ng ? g + 1 : g;
        const double* __restrict__ src = a.zf + ((size_t)gn * a.NP + c) * KK * 64 + lane;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) zn[kk] = kk < 4 * (TH * half + TH) ? src[kk * 64] : 0.0;
      }
      double4_t acc[TH];
#pragma unroll
      for (int j = 0; j < TH; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int j = 0; j < TH; ++j)
          if ((kk >> 2) <= TH * half + j) acc[j] = mfma_f64(zf[kk], Lf[j][kk], acc[j]);  // (tiles below the block diagonal are zero: skipped)
#pragma unroll
      for (int j = 0; j < TH; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int so = g * 16 + hi + 4 * r;
          if (so < a.S) a.inc[((size_t)so * a.NP + c) * DP + 16 * (TH * half + j) + lc] = acc[j][r];
        }
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) zf[kk] = zn[kk];
    }
  }
}

// ---- recorded normals (replay mode) -> the fragment layout k_wide_apply reads (what k_rng<128> writes from Philox) ----
struct WideReplayArgs {
  int64_t N, NP;
  int d, S;
  const double* z;   // [S][N][d]
  double* zf;        // [groups][NP][32][64]
};
template <int T>
__global__ void __launch_bounds__(64) k_wide_replay_frags(const WideReplayArgs a) {
  constexpr int KK = 4 * T;
  const int lane = threadIdx.x, lc = lane & 15, hi = lane >> 4;
  const int64_t c = blockIdx.x;
  const int g = blockIdx.y, s = g * 16 + lc;
  double* __restrict__ dst = a.zf + ((size_t)g * a.NP + c) * KK * 64;
  for (int kk = 0; kk < KK; ++kk) {
    const int dim = 4 * kk + hi;  // fragment kk, lane (lc = step in group, hi): z[step][4 kk + hi]
    dst[kk * 64 + lane] = (s < a.S && c < a.N && dim < a.d) ? a.z[((size_t)s * a.N + c) * a.d + dim] : 0.0;
  }
}

// accept uniforms of a block and their logs, recorded (replay) or from the engine's stream: what k_propose does per step inline
__global__ void __launch_bounds__(256) k_wide_uniforms(const ProposeArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)a.S * a.NP) return;
  const int64_t s = i / a.NP, c = i % a.NP;
  double u = 0.5;
  if (c < a.N) {
    u = a.u_replay ? a.u_replay[(size_t)s * a.N + c] : accept_uniform(a.seed, (uint32_t)(a.chain_offset + c), (uint32_t)(a.step0 + s), 0u);
    if (a.u_export) a.u_export[(size_t)s * a.N + c] = u;
  }
  a.u[i] = u;
  if (a.logu) a.logu[i] = log(u);
}

}  // namespace tda
