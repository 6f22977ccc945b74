// HIP kernels of the many-chain MH engine for gfx950 (MI355X).  No CUDA / multi-backend paths.
//
// Pipeline per block of S <= period steps (proposal distribution is constant inside a block, because
// tinyDA's proposals only change at adapt-count multiples of `period`, proposal.py:234,509):
//
//   k_propose   wave per chain : Philox normals z_s, increments inc_s = L z_s (L = chol C, per chain for
//                                AdaptiveMetropolis), accept uniforms u_s            -> HBM [S][N][D]
//   k_mh_steps  workgroup = 16 chains x 4 waves, S fused steps:
//                                theta' = theta + scaling * inc_s  (pCN: sqrt(1-b^2) theta + b inc_s)
//                                F = A theta' on fp64 MFMA (v_mfma_f64_16x16x4), observations split over
//                                the 4 waves, A fragments streamed from L2, residual + weighted SSE fused
//                                in the MFMA epilogue, prior, log alpha, accept, coalesced record write
//   k_adapt     wave per chain : RecursiveSampleMoments catch-up over the S recorded states in the
//                                reference's exact elementwise arithmetic (utils.py:113-122), symmetric half
//                                only (circulant fold), global scaling adaptation at period boundaries
//   k_chol      wave per chain : C <- Sigma swap, Cholesky in LDS (only at period boundaries with t >= t0)
//
// Chains never interact, so there is no inter-workgroup communication anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tda_philox.h"

namespace tda {

typedef double double4_t __attribute__((ext_vector_type(4)));

enum : int { MODE_STEP = 0, MODE_EVAL = 1 };
enum : int { PRIOR_DIAG = 0, PRIOR_DENSE = 1, PRIOR_STANDARD = 2 };  // STANDARD = N(0, I): no constants to load

// One level's linear-Gaussian posterior pieces, device pointers.
struct LevelDev {
  const double* Apk;   // packed MFMA fragments [ncb][KS/2][64 lanes][2]: A[cb*16+(l&15)][4*(2*k2+e)+(l>>4)]
  const double* ytil;  // [m_pad] data - b   (zero padded)
  const double* w;     // [m_pad] 1/diag(noise) for TDA_NOISE_DIAG, else nullptr
  const double* Ppk;   // TDA_NOISE_DENSE: Sigma^-1 as MFMA fragments [ncb][m_pad/8][64 lanes][2] (rows = o', k = o)
  int ncb;             // m_pad / 16
  int m_pad;
  int noise_kind;
  double var;  // isotropic variance
};

struct PriorDev {
  const double* mean;  // [DPAD]
  const double* pinv;  // [DPAD] 1/var_j (PRIOR_DIAG), zero padded
  const double* Wpk;   // PRIOR_DENSE: packed fragments of the whitening matrix W = chol(cov)^-1
  const double* wmu;   // PRIOR_DENSE: W mean, [ncb*16]
  int ncb;
  int kind;
  double logconst;  // d*log(2 pi) + log det cov
};

struct StepArgs {
  LevelDev lv;
  PriorDev pr;
  int64_t N;        // real chains
  int64_t NP;       // padded to 16
  int d;            // real dim
  int S;            // steps in this launch
  int mode;         // MODE_STEP / MODE_EVAL
  int prop_kind;    // tda_proposal_kind
  // chain state (engine-internal, padded layouts)
  double* theta;    // [NP][DPAD]
  double* lp;       // [NP]
  double* ll;       // [NP]
  const double* scaling;  // [NP]
  int32_t* acc_count;     // [NP] accepted since last adaptation boundary
  // block inputs
  const double* inc;  // [S][NP][DPAD]
  const double* u;    // [S][NP]
  const double* logu; // [S][NP] log(u), produced by k_propose off the critical path (may be null)
  // records, layout of tda_outputs (may be null)
  double* rec_params;
  double* rec_stats;
  uint8_t* rec_acc;
};

struct ProposeArgs {
  int64_t N, NP;
  int64_t chain_offset;
  int d;
  int S;
  int64_t step0;          // global step index of s = 0
  uint64_t seed;
  const double* Lk;       // [NP or 1][DPAD][DPAD] k-major: Lk[c][k][j] = L[j][k]
  int64_t L_stride;       // DPAD*DPAD or 0 when shared
  double* inc;            // [S][NP][DPAD]
  double* u;              // [S][NP]
  double* logu;           // [S][NP] (may be null)
  const double* z_replay; // [.][N][d] at step0 (may be null)
  const double* u_replay; // [.][N]
  double* z_export;       // same layout (may be null)
  double* u_export;
};

struct AdaptArgs {
  int64_t N, NP;
  int d;
  int S;
  int64_t t_base;  // proposal.t before this block
  int do_am;       // update RecursiveSampleMoments
  int boundary;    // (t_base + S) % period == 0
  int do_scale;    // adaptive scaling at boundary
  int do_swap;     // AM: t >= t0 at boundary -> C <- Sigma
  int period;
  double gamma_pow;  // gamma ** -k  (proposal.py:240)
  double sd, eps;
  const double* rec_params;  // [S][N][d] states recorded by k_mh_steps
  double* am_mu;             // [NP][DPAD]
  double* am_sigma;          // [NP][DPAD/2+1][DPAD] circulant fold: [s][l] = Sigma[l][(l+s) mod DPAD]
  double* scaling;           // [NP]
  int32_t* acc_count;        // [NP]
  int32_t* flags;            // [NP]
  const uint8_t* ring;       // multi-level: recent entries of the base proposal's accepted list, [ring_P][NP]
  int ring_P;                // ring capacity (>= period + levels)
  int64_t ring_hi;           // absolute list position just after the boundary base step's own flag: adapt()
                             // runs before the upper level of that step appends its alignment entry
};

typedef unsigned uint2_t __attribute__((ext_vector_type(2)));

// v + (lane ^ 16) + (lane ^ 32) + (lane ^ 48): the reduction over the four 16-lane rows that hold one chain's partial
// sums in the MFMA C/D layout.  v_permlane16_swap / v_permlane32_swap (gfx950) instead of two dependent ds_bpermute
// round trips; same grouping ((r0 + r1) + (r2 + r3)) as the shuffle form, so results are bit-identical.
__device__ __forceinline__ double sum_rows(double v) {
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  uint2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  uint2_t b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double s = __hiloint2double(b.x, a.x) + __hiloint2double(b.y, a.y);
  lo = __double2loint(s);
  hi = __double2hiint(s);
  a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b.x, a.x) + __hiloint2double(b.y, a.y);
}

__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// MFMA A-operand fragments of one 16-row block: K2 16-byte loads per lane, unconditional (the block index is
// clamped, out-of-range blocks are simply not accumulated) so that hipcc emits plain global_load_dwordx4
// and counted vmcnt waits instead of one branch per load.
template <int DPAD>
__device__ __forceinline__ void frag_load(const double2* __restrict__ base, int cb, int ncb,
                                          double2 (&f)[DPAD / 8]) {
  const int cbc = cb < ncb ? cb : ncb - 1;
  const double2* __restrict__ p = base + (size_t)cbc * (DPAD / 8) * 64;
#pragma unroll
  for (int k = 0; k < DPAD / 8; ++k) f[k] = p[k * 64];
}

// One pair of 16-row blocks: 2 x KS MFMAs on two accumulators, then the fused epilogue
// sum_r w_o (F_o - ytil_o)^2 over the rows this lane holds ((l >> 4) + 4 r, C/D layout of the f64 MFMA).
template <int DPAD, int MODE>
__device__ __forceinline__ double pair_sse(const double2 (&f0)[DPAD / 8], const double2 (&f1)[DPAD / 8],
                                           const double (&th)[DPAD / 4], const double* __restrict__ s_y,
                                           double* __restrict__ s_w, int cb0, int cb1, bool v1, int hi) {
  constexpr bool HAS_W = MODE == 1;
  double4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < DPAD / 8; ++k) {
    a0 = mfma_f64(f0[k].x, th[2 * k], a0);
    a1 = mfma_f64(f1[k].x, th[2 * k], a1);
    a0 = mfma_f64(f0[k].y, th[2 * k + 1], a0);
    a1 = mfma_f64(f1[k].y, th[2 * k + 1], a1);
  }
  double sse = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = cb0 * 16 + hi + 4 * r;
    const double res = a0[r] - s_y[o];
    if (MODE == 2) {
      s_w[o] = res;  // s_w = this lane's residual row (chain l & 15) of the LDS tile
    } else {
      double sq = res * res;
      if (HAS_W) sq *= s_w[o];
      sse += sq;
    }
  }
  const int ob1 = v1 ? cb1 : cb0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = ob1 * 16 + hi + 4 * r;
    const double res = a1[r] - s_y[o];
    if (MODE == 2) {
      if (v1) s_w[o] = res;
    } else {
      double sq = res * res;
      if (HAS_W) sq *= s_w[o];
      sse += v1 ? sq : 0.0;
    }
  }
  return sse;
}

// Sum over this wave's observation blocks (wave, wave+4, wave+8, ...) of w_o (A theta' - ytil)_o^2 for the
// 16 chains of the tile.  Software pipeline with two explicit register sets: while pair P is in the matrix
// pipe (2 x KS x 64 cycles), the fragments of pair P+1 are in flight from L2.  The sched_barriers keep hipcc
// from sinking the loads below the MFMAs that precede them in program order.
// fa0 / fa1 must hold blocks `wave` and `wave + NW` on entry (issued by the caller ahead of its barrier);
// NW = waves sharing the tile (observation blocks are dealt round-robin over them).
template <int DPAD, int MODE, int NW = 4>
__device__ __forceinline__ double level_sse_partial(const double* __restrict__ Apk, int ncb,
                                                    const double* __restrict__ s_y,
                                                    double* __restrict__ s_w,
                                                    const double (&th)[DPAD / 4], int wave, int lane,
                                                    double2 (&fa0)[DPAD / 8], double2 (&fa1)[DPAD / 8]) {
  constexpr int K2 = DPAD / 8;
  const int hi = lane >> 4;
  const double2* __restrict__ base = reinterpret_cast<const double2*>(Apk) + lane;
  double sse = 0.0;
  double2 fb0[K2], fb1[K2];
  for (int cb = wave; cb < ncb; cb += 4 * NW) {
    frag_load<DPAD>(base, cb + 2 * NW, ncb, fb0);
    frag_load<DPAD>(base, cb + 3 * NW, ncb, fb1);
    __builtin_amdgcn_sched_barrier(0);
    sse += pair_sse<DPAD, MODE>(fa0, fa1, th, s_y, s_w, cb, cb + NW, cb + NW < ncb, hi);
    __builtin_amdgcn_sched_barrier(0);
    frag_load<DPAD>(base, cb + 4 * NW, ncb, fa0);
    frag_load<DPAD>(base, cb + 5 * NW, ncb, fa1);
    __builtin_amdgcn_sched_barrier(0);
    if (cb + 2 * NW < ncb)
      sse += pair_sse<DPAD, MODE>(fb0, fb1, th, s_y, s_w, cb + 2 * NW, cb + 3 * NW, cb + 3 * NW < ncb, hi);
    __builtin_amdgcn_sched_barrier(0);
  }
  return sse;
}

// Single-block variant of the pipeline for the 8-wave tile (two waves per SIMD, 256 registers each): one
// accumulator chain per block (a dependent f64 MFMA chain issues at full rate), two fragment sets of 32 VGPRs.
// The second wave of the SIMD covers this wave's epilogue and waits.  fa holds block `wave` on entry.
template <int DPAD, int MODE>
__device__ __forceinline__ double block_sse(const double2 (&f)[DPAD / 8], const double (&th)[DPAD / 4],
                                            const double* __restrict__ s_y, double* __restrict__ s_w, int cb, int hi) {
  double4_t a0 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < DPAD / 8; ++k) {
    a0 = mfma_f64(f[k].x, th[2 * k], a0);
    a0 = mfma_f64(f[k].y, th[2 * k + 1], a0);
  }
  double sse = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = cb * 16 + hi + 4 * r;
    const double res = a0[r] - s_y[o];
    if (MODE == 2) {
      s_w[o] = res;
    } else {
      double sq = res * res;
      if (MODE == 1) sq *= s_w[o];
      sse += sq;
    }
  }
  return sse;
}

// frag_load with wrap-around: past the wave's last block it fetches the wave's FIRST block again, i.e. the fragments
// the next MH step starts with, so their L2 latency hides behind the serial end-of-step phase.
template <int DPAD>
__device__ __forceinline__ void frag_load_wrap(const double2* __restrict__ base, int cb, int ncb, int first,
                                               double2 (&f)[DPAD / 8]) {
  const int cbc = cb < ncb ? cb : first;
  const double2* __restrict__ p = base + (size_t)cbc * (DPAD / 8) * 64;
#pragma unroll
  for (int k = 0; k < DPAD / 8; ++k) f[k] = p[k * 64];
}

template <int DPAD, int MODE, int NW>
__device__ __forceinline__ double level_sse_single(const double* __restrict__ Apk, int ncb,
                                                   const double* __restrict__ s_y, double* __restrict__ s_w,
                                                   const double (&th)[DPAD / 4], int wave, int lane,
                                                   double2 (&fa)[DPAD / 8]) {
  const int hi = lane >> 4;
  const double2* __restrict__ base = reinterpret_cast<const double2*>(Apk) + lane;
  const int first = wave < ncb ? wave : ncb - 1;
  double sse = 0.0;
  double2 fb[DPAD / 8];
  bool next_in_fb = false;  // where the next step's first block ended up
  for (int cb = wave; cb < ncb; cb += 2 * NW) {
    frag_load_wrap<DPAD>(base, cb + NW, ncb, first, fb);
    __builtin_amdgcn_sched_barrier(0);
    sse += block_sse<DPAD, MODE>(fa, th, s_y, s_w, cb, hi);
    __builtin_amdgcn_sched_barrier(0);
    if (cb + NW < ncb) {
      frag_load_wrap<DPAD>(base, cb + 2 * NW, ncb, first, fa);
      __builtin_amdgcn_sched_barrier(0);
      sse += block_sse<DPAD, MODE>(fb, th, s_y, s_w, cb + NW, hi);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      next_in_fb = true;  // odd number of blocks: fb already holds the wrapped-around first block
    }
  }
  if (next_in_fb) {
#pragma unroll
    for (int k = 0; k < DPAD / 8; ++k) fa[k] = fb[k];
  }
  return sse;  // fa now holds block `first` again, ready for the next step
}

// r^T Sigma^-1 r for the 16 chains of a tile, residual tile s_R[chain][o] (row stride RS doubles) in LDS,
// DefaultGaussianLogLike.loglike (tinyDA/distributions.py:295-298).  The D layout of the f64 MFMA (row = (l>>4)+4r)
// is also its B-operand layout, so the residuals feed the second GEMM straight from LDS with ds_read_b64.
// Sigma^-1 is symmetric: only 16x16 blocks on or below the diagonal are multiplied, off-diagonal blocks count twice.
// The (block row, k-group) work list of a wave is flattened so that the 8 fragment loads of the next item are in
// flight from L2 / Infinity Cache while the current item's up to 16 MFMAs execute.
__device__ __forceinline__ void dq_load(const double2* __restrict__ base, int K2tot, int cbp, int g0, int ncb,
                                        double2 (&f)[8]) {
  const int cb = cbp < ncb ? cbp : ncb - 1;
  const int kend = 2 * (cb + 1);
  const double2* __restrict__ row = base + (size_t)cb * K2tot * 64;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k2 = g0 + j < kend ? g0 + j : kend - 1;
    f[j] = row[(size_t)k2 * 64];
  }
}

__device__ __forceinline__ void dq_compute(const double2 (&f)[8], int cbp, int g0, const double* __restrict__ rrow,
                                           int hi, double4_t& aoff, double4_t& adiag) {
  const int kend = 2 * (cbp + 1), kdiag = 2 * cbp;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k2 = g0 + j;
    if (k2 < kend) {
      const double b0 = rrow[8 * k2 + hi], b1 = rrow[8 * k2 + 4 + hi];
      if (k2 < kdiag) {
        aoff = mfma_f64(f[j].x, b0, aoff);
        aoff = mfma_f64(f[j].y, b1, aoff);
      } else {
        adiag = mfma_f64(f[j].x, b0, adiag);
        adiag = mfma_f64(f[j].y, b1, adiag);
      }
    }
  }
}

template <int NW = 4>
__device__ __forceinline__ double dense_quadform(const double* __restrict__ Ppk, int ncb, int m_pad,
                                                 const double* __restrict__ s_R, int RS, int wave, int lane) {
  const int lc = lane & 15, hi = lane >> 4;
  const int K2tot = m_pad / 8;
  const double2* __restrict__ base = reinterpret_cast<const double2*>(Ppk) + lane;
  const double* __restrict__ rrow = s_R + lc * RS;
  double s = 0.0;
  double2 fa[8], fb[8];
  int cbp = wave, g0 = 0;
  double4_t aoff = {0.0, 0.0, 0.0, 0.0}, adiag = {0.0, 0.0, 0.0, 0.0};
  dq_load(base, K2tot, cbp, g0, ncb, fa);
  while (cbp < ncb) {
    // ---- phase A: compute from fa while fb loads ----
    int ncbp = cbp, ng0 = g0 + 8;
    if (ng0 >= 2 * (cbp + 1)) {
      ncbp = cbp + NW;
      ng0 = 0;
    }
    dq_load(base, K2tot, ncbp, ng0, ncb, fb);
    __builtin_amdgcn_sched_barrier(0);
    dq_compute(fa, cbp, g0, rrow, hi, aoff, adiag);
    if (ncbp != cbp) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s += rrow[cbp * 16 + hi + 4 * r] * (2.0 * aoff[r] + adiag[r]);
      aoff = double4_t{0.0, 0.0, 0.0, 0.0};
      adiag = double4_t{0.0, 0.0, 0.0, 0.0};
    }
    __builtin_amdgcn_sched_barrier(0);
    cbp = ncbp;
    g0 = ng0;
    if (cbp >= ncb) break;
    // ---- phase B: compute from fb while fa loads ----
    ncbp = cbp;
    ng0 = g0 + 8;
    if (ng0 >= 2 * (cbp + 1)) {
      ncbp = cbp + NW;
      ng0 = 0;
    }
    dq_load(base, K2tot, ncbp, ng0, ncb, fa);
    __builtin_amdgcn_sched_barrier(0);
    dq_compute(fb, cbp, g0, rrow, hi, aoff, adiag);
    if (ncbp != cbp) {
#pragma unroll
      for (int r = 0; r < 4; ++r) s += rrow[cbp * 16 + hi + 4 * r] * (2.0 * aoff[r] + adiag[r]);
      aoff = double4_t{0.0, 0.0, 0.0, 0.0};
      adiag = double4_t{0.0, 0.0, 0.0, 0.0};
    }
    __builtin_amdgcn_sched_barrier(0);
    cbp = ncbp;
    g0 = ng0;
  }
  return s;
}

template <int DPAD>
__host__ __device__ constexpr int steps_lds_doubles(int m_pad, bool diag, int prior_rows) {
  return 16 * (DPAD + 2) + 64 + 64 + m_pad + (diag ? m_pad : 0) + prior_rows;
}

// ------------------------------------------------------------------------------------------------
// S fused Metropolis-Hastings steps for one tile of 16 chains  (Chain.sample, tinyDA/chain.py:95-125)
// NW waves share the tile (4 = one wave per SIMD with up to 512 registers, 8 = two per SIMD with 256):
// the observation blocks of the forward model are dealt round-robin over the waves.
// ------------------------------------------------------------------------------------------------
template <int DPAD, int NW>
__global__ void __launch_bounds__(64 * NW, NW / 4) k_mh_steps(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int KS = DPAD / 4;
  constexpr int LDP = DPAD + 2;  // row stride: conflict-free ds_read_b64 fragment gather
  constexpr int TPC = 4 * NW;    // threads per chain in the thread-mapped phases
  constexpr int EPT = DPAD >= TPC ? DPAD / TPC : 1;
  constexpr int QACT = DPAD / EPT;

  const bool diag = a.lv.noise_kind == 1;
  const bool dense = a.lv.noise_kind == 2;
  const bool prior_dense = a.pr.kind == PRIOR_DENSE;
  const int RS = a.lv.m_pad + 2;  // residual tile row stride (dense noise)
  double* s_prop = smem;
  double* s_red = s_prop + 16 * LDP;   // [NW][16]
  double* s_redp = s_red + 16 * NW;    // [NW][16]
  double* s_pm = s_redp + 16 * NW;     // prior mean  [DPAD]
  double* s_pinv = s_pm + DPAD;        // prior 1/var [DPAD]
  double* s_y = s_pinv + DPAD;
  double* s_w = s_y + a.lv.m_pad;
  double* s_py = s_w + (diag ? a.lv.m_pad : 0);
  double* s_R = s_py + (prior_dense ? a.pr.ncb * 16 : 0);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = blockIdx.x;
  const int c = tid / TPC, q = tid % TPC;    // thread-mapped (chain, element group)
  const int lc = lane & 15, hi = lane >> 4;  // lane-mapped chain / k sub-index
  const int64_t gct = tile * 16 + c;
  const int64_t gcl = tile * 16 + lc;
  const bool active = q < QACT;
  constexpr int NT = 64 * NW;

  for (int i = tid; i < a.lv.m_pad; i += NT) {
    s_y[i] = a.lv.ytil[i];
    if (diag) s_w[i] = a.lv.w[i];
  }
  if (prior_dense)
    for (int i = tid; i < a.pr.ncb * 16; i += NT) s_py[i] = a.pr.wmu[i];
  for (int i = tid; i < DPAD; i += NT) {
    s_pm[i] = a.pr.mean[i];
    s_pinv[i] = prior_dense ? 0.0 : a.pr.pinv[i];
  }

  double cur[EPT], prp[EPT], xin[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    cur[e] = active ? a.theta[gct * DPAD + q * EPT + e] : 0.0;
    xin[e] = 0.0;
  }
  double lp = a.lp[gcl], ll = a.ll[gcl];
  const double scal_t = a.scaling[gct];
  const double keep_t = a.prop_kind == 1 ? sqrt(1.0 - scal_t * scal_t) : 1.0;  // proposal.py:351-352
  int nacc = 0;
  const bool is_eval = a.mode == MODE_EVAL;
  const bool is_pcn = a.prop_kind == 1;
  const double2* fbase = reinterpret_cast<const double2*>(a.lv.Apk) + lane;
  const double2* pbase = reinterpret_cast<const double2*>(a.pr.Wpk) + lane;
  constexpr bool PAIRS = NW == 4;  // 4 waves: pairs of blocks, 4 fragment sets; 8 waves: single blocks, 2 sets
  double2 f0[KS / 2], f1[PAIRS ? KS / 2 : 1];
  double unext = 0.5, lunext = 0.0;
  const bool has_logu = a.logu != nullptr;
  const bool prior_std = a.pr.kind == PRIOR_STANDARD;
  if (!is_eval) {
    if (active) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) xin[e] = a.inc[(size_t)gct * DPAD + q * EPT + e];
    }
    unext = a.u[gcl];
    if (has_logu) lunext = a.logu[gcl];
  }
  __syncthreads();

  if constexpr (!PAIRS) frag_load<DPAD>(fbase, wave, a.lv.ncb, f0);  // later steps: prefetched by the previous step
  for (int s = 0; s < a.S; ++s) {
    // first fragment block(s) of this step: independent of theta', issued ahead of the barrier
    if constexpr (PAIRS) {
      frag_load<DPAD>(fbase, wave, a.lv.ncb, f0);
      frag_load<DPAD>(fbase, wave + NW, a.lv.ncb, f1);
    }
    // ---- proposal: theta' (proposal.py:249-251 / :351-355) ----
    if (active) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        if (is_eval) {
          prp[e] = cur[e];
        } else {
          const double sx = scal_t * xin[e];
          prp[e] = is_pcn ? keep_t * cur[e] + sx : cur[e] + sx;
        }
        s_prop[c * LDP + q * EPT + e] = prp[e];
      }
    }
    const double u = unext, lu = lunext;
    if (!is_eval && s + 1 < a.S) {  // next step's increment and uniform fly during the MFMA phase
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) xin[e] = a.inc[((size_t)(s + 1) * a.NP + gct) * DPAD + q * EPT + e];
      }
      unext = a.u[(size_t)(s + 1) * a.NP + gcl];
      if (has_logu) lunext = a.logu[(size_t)(s + 1) * a.NP + gcl];
    }
    __syncthreads();

    // ---- gather theta' into MFMA B-operand fragments ----
    double th[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) th[kk] = s_prop[lc * LDP + 4 * kk + hi];

    // ---- prior: scipy MVN logpdf (posterior.py:92) ----
    double maha = 0.0;
    if (prior_std) {
      double p = 0.0;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) p += th[kk] * th[kk];
      p = sum_rows(p);
      maha = p;
    } else if (!prior_dense) {
      double p = 0.0;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const double dv = th[kk] - s_pm[4 * kk + hi];
        p += dv * dv * s_pinv[4 * kk + hi];
      }
      p = sum_rows(p);
      maha = p;
    } else {
      double p;
      if constexpr (PAIRS) {
        double2 p0[KS / 2], p1[KS / 2];
        frag_load<DPAD>(pbase, wave, a.pr.ncb, p0);
        frag_load<DPAD>(pbase, wave + NW, a.pr.ncb, p1);
        p = level_sse_partial<DPAD, 0, NW>(a.pr.Wpk, a.pr.ncb, s_py, nullptr, th, wave, lane, p0, p1);
      } else {
        double2 p0[KS / 2];
        frag_load<DPAD>(pbase, wave, a.pr.ncb, p0);
        p = level_sse_single<DPAD, 0, NW>(a.pr.Wpk, a.pr.ncb, s_py, nullptr, th, wave, lane, p0);
      }
      p = sum_rows(p);
      if (lane < 16) s_redp[wave * 16 + lane] = p;
    }

    // ---- forward model + Gaussian log-likelihood (posterior.py:95-108, distributions.py:295-326) ----
    double sse;
    if constexpr (PAIRS) {
      if (dense) {
        // residuals -> LDS tile, then r^T Sigma^-1 r on the matrix cores (distributions.py:295-298)
        (void)level_sse_partial<DPAD, 2, NW>(a.lv.Apk, a.lv.ncb, s_y, s_R + (lane & 15) * RS, th, wave, lane, f0, f1);
        __syncthreads();
        sse = dense_quadform<NW>(a.lv.Ppk, a.lv.ncb, a.lv.m_pad, s_R, RS, wave, lane);
      } else {
        sse = diag ? level_sse_partial<DPAD, 1, NW>(a.lv.Apk, a.lv.ncb, s_y, s_w, th, wave, lane, f0, f1)
                   : level_sse_partial<DPAD, 0, NW>(a.lv.Apk, a.lv.ncb, s_y, nullptr, th, wave, lane, f0, f1);
      }
    } else {  // the host launches the 8-wave tile for isotropic / diagonal noise only
      sse = diag ? level_sse_single<DPAD, 1, NW>(a.lv.Apk, a.lv.ncb, s_y, s_w, th, wave, lane, f0)
                 : level_sse_single<DPAD, 0, NW>(a.lv.Apk, a.lv.ncb, s_y, nullptr, th, wave, lane, f0);
    }
    sse = sum_rows(sse);
    if (lane < 16) s_red[wave * 16 + lane] = sse;
    __syncthreads();

    double tot = s_red[lc];
#pragma unroll
    for (int w = 1; w < NW; ++w) tot += s_red[w * 16 + lc];
    if (prior_dense) {
      maha = s_redp[lc];
#pragma unroll
      for (int w = 1; w < NW; ++w) maha += s_redp[w * 16 + lc];
    }
    const double ll_n = (diag || dense) ? -0.5 * tot : -0.5 * tot / a.lv.var;
    const double lp_n = -0.5 * (a.pr.logconst + maha);
    const double post_n = lp_n + ll_n;  // link.py:48

    // ---- Metropolis test (proposal.py:253-258, :357-362; chain.py:112) ----
    // The reference tests u < exp(delta).  exp is monotone, so away from the knife edge log(u) < delta decides the
    // same way without a transcendental on the critical path; within 1e-9 of the edge (probability ~1e-9 per step)
    // the reference form itself is evaluated.
    bool acc;
    if (is_eval) {
      acc = true;
    } else {
      const double delta = is_pcn ? ll_n - ll : post_n - (lp + ll);
      if (has_logu && (fabs(lu - delta) > 1e-9 || delta != delta)) {
        acc = (post_n == post_n) && (lu < delta);
      } else {
        double alpha = exp(delta);
        if (post_n != post_n) alpha = 0.0;
        acc = u < alpha;
      }
    }
    if (acc) {
      lp = lp_n;
      ll = ll_n;
    }
    nacc += acc ? 1 : 0;

    if (!is_eval && wave == 0 && lane < 16 && gcl < a.N) {
      const size_t r = (size_t)s * a.N + gcl;
      if (a.rec_stats) {
        a.rec_stats[r * 3 + 0] = lp;
        a.rec_stats[r * 3 + 1] = ll;
        a.rec_stats[r * 3 + 2] = lp + ll;
      }
      if (a.rec_acc) a.rec_acc[r] = acc ? 1 : 0;
    }

    // ---- state update + coalesced parameter record ----
    const int accf = __shfl(acc ? 1 : 0, c);
    if (active) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        cur[e] = accf ? prp[e] : cur[e];
        const int j = q * EPT + e;
        if (!is_eval && a.rec_params && gct < a.N && j < a.d)
          a.rec_params[((size_t)s * a.N + gct) * a.d + j] = cur[e];
      }
    }
  }

  if (active) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) a.theta[gct * DPAD + q * EPT + e] = cur[e];
  }
  if (wave == 0 && lane < 16) {
    a.lp[gcl] = lp;
    a.ll[gcl] = ll;
    if (!is_eval && a.acc_count) a.acc_count[gcl] += nacc;
  }
}

// ------------------------------------------------------------------------------------------------
// Proposal increments for a block of steps: one wave per chain.
//   np.random.multivariate_normal(0, C) (proposal.py:249-251) as L z with L = chol(C), z from Philox.
// Lane j owns row j of L in registers; Box-Muller pairs of 64/(DPAD/2) steps are generated per pass.
// inc_j = sum_k fma(L[j][k], z[k]) in ascending k.
// ------------------------------------------------------------------------------------------------
template <int DPAD>
__global__ void __launch_bounds__(64) k_propose(const ProposeArgs a) {
  constexpr int HP = DPAD / 2;     // Box-Muller pairs per step
  constexpr int SPP = 64 / HP;     // steps per pass
  __shared__ double s_z[SPP * DPAD];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  const bool real_chain = c < a.N;
  const uint32_t gc = (uint32_t)(a.chain_offset + c);

  double Lrow[DPAD];
#pragma unroll
  for (int k = 0; k < DPAD; ++k)
    Lrow[k] = lane < DPAD ? a.Lk[(size_t)c * a.L_stride + (size_t)k * DPAD + lane] : 0.0;

  const int sp = lane / HP, p = lane % HP;
  for (int s0 = 0; s0 < a.S; s0 += SPP) {
    const int s = s0 + sp;
    double z0 = 0.0, z1 = 0.0;
    if (s < a.S && real_chain && 2 * p < a.d) {
      if (a.z_replay) {
        const size_t o = ((size_t)s * a.N + c) * a.d + 2 * p;
        z0 = a.z_replay[o];
        z1 = (2 * p + 1 < a.d) ? a.z_replay[o + 1] : 0.0;
      } else {
        normal_pair(a.seed, gc, (uint32_t)(a.step0 + s), STREAM_PROPOSAL, (uint32_t)p, z0, z1);
        if (2 * p + 1 >= a.d) z1 = 0.0;
      }
      if (a.z_export) {
        const size_t o = ((size_t)s * a.N + c) * a.d + 2 * p;
        a.z_export[o] = z0;
        if (2 * p + 1 < a.d) a.z_export[o + 1] = z1;
      }
    }
    s_z[sp * DPAD + 2 * p] = z0;
    s_z[sp * DPAD + 2 * p + 1] = z1;
    __syncthreads();
    double accv[SPP];
#pragma unroll
    for (int i = 0; i < SPP; ++i) accv[i] = 0.0;
#pragma unroll
    for (int k = 0; k < DPAD; ++k) {
#pragma unroll
      for (int i = 0; i < SPP; ++i) accv[i] = fma(Lrow[k], s_z[i * DPAD + k], accv[i]);
    }
    if (lane < DPAD) {
#pragma unroll
      for (int i = 0; i < SPP; ++i)
        if (s0 + i < a.S) a.inc[((size_t)(s0 + i) * a.NP + c) * DPAD + lane] = accv[i];
    }
    __syncthreads();
  }
  // accept uniforms (chain.py:112)
  for (int s = lane; s < a.S; s += 64) {
    double u = 0.5;
    if (real_chain) {
      u = a.u_replay ? a.u_replay[(size_t)s * a.N + c]
                     : accept_uniform(a.seed, gc, (uint32_t)(a.step0 + s), 0u);
      if (a.u_export) a.u_export[(size_t)s * a.N + c] = u;
    }
    a.u[(size_t)s * a.NP + c] = u;
    if (a.logu) a.logu[(size_t)s * a.NP + c] = log(u);
  }
}

// ------------------------------------------------------------------------------------------------
// Adaptation for a block: one wave per chain.
//   RecursiveSampleMoments.update (utils.py:113-124) for each recorded state, elementwise, unfused:
//     mu' = (1/(t+1)) (t mu + x)
//     Sigma' = (t-1)/t Sigma + sd/t ( t mu mu^T - (t+1) mu' mu'^T + x x^T + eps I )
//   global scaling (proposal.py:234-243).
// Sigma is symmetric and every product commutes bitwise, so only one of (i,j)/(j,i) is carried, in a
// circulant fold: lane l, slot s holds Sigma[l][(l+s) mod D], s = 0..D/2.  The "row" operand is the lane's
// own value and the "column" operand a rotation read from LDS with consecutive addresses (conflict free),
// so a step costs (D/2+1) x (3 ds_read_b64 + 10 fp64 VALU ops) instead of D x (3 broadcasts + 10 ops).
// This file is compiled with -ffp-contract=off so the products and sums round exactly like NumPy's.
// ------------------------------------------------------------------------------------------------
template <int DPAD>
__global__ void __launch_bounds__(64) k_adapt(const AdaptArgs a) {
  constexpr int NS = DPAD / 2 + 1;
  // x, mu, mu' each stored twice ([j] and [j + DPAD]) so that the rotated operand of slot s is a plain ds_read_b64
  // at immediate offset s from the lane's own base: consecutive lanes hit consecutive banks (conflict free) and no
  // per-slot address arithmetic is needed
  __shared__ __attribute__((aligned(16))) double s_vec[3 * 2 * DPAD];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const bool lj = lane < a.d;
  const bool lp = lane < DPAD;

  if (a.do_am) {
    double Sg[NS];
    double mu = lp ? a.am_mu[c * DPAD + lane] : 0.0;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) Sg[sl] = lp ? a.am_sigma[((size_t)c * NS + sl) * DPAD + lane] : 0.0;
    double xn = lj ? a.rec_params[(size_t)c * a.d + lane] : 0.0;
    for (int s = 0; s < a.S; ++s) {
      const double x = xn;
      if (s + 1 < a.S) xn = lj ? a.rec_params[((size_t)(s + 1) * a.N + c) * a.d + lane] : 0.0;
      const double t = (double)(a.t_base + s + 1);  // recursor.t before this update
      const double mup = (1.0 / (t + 1.0)) * (t * mu + x);
      const double ca = (t - 1.0) / t, cb = a.sd / t;
      const double t1 = t + 1.0;
      __syncthreads();  // previous step's rotation reads are done
      if (lp) {
        s_vec[lane] = x;
        s_vec[lane + DPAD] = x;
        s_vec[2 * DPAD + lane] = mu;
        s_vec[2 * DPAD + lane + DPAD] = mu;
        s_vec[4 * DPAD + lane] = mup;
        s_vec[4 * DPAD + lane + DPAD] = mup;
      }
      __syncthreads();
      const double* __restrict__ rot = s_vec + (lane < DPAD ? lane : 0);
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) {
        const double xj = rot[sl], mj = rot[2 * DPAD + sl], mpj = rot[4 * DPAD + sl];
        double M = (t * (mu * mj) - t1 * (mup * mpj)) + x * xj;
        if (sl == 0) M = lj ? M + a.eps : M;
        Sg[sl] = ca * Sg[sl] + cb * M;
      }
      mu = mup;
    }
    if (lp) {
      a.am_mu[c * DPAD + lane] = mu;
#pragma unroll
      for (int sl = 0; sl < NS; ++sl) a.am_sigma[((size_t)c * NS + sl) * DPAD + lane] = Sg[sl];
    }
  }

  if (!a.boundary) return;
  if (a.do_scale && lane == 0) {
    int hits = 0;
    if (a.ring) {
      for (int i = 1; i <= a.period; ++i) hits += a.ring[(size_t)((a.ring_hi - i) % a.ring_P) * a.NP + c];
    } else {
      hits = a.acc_count[c];
    }
    const double rate = (double)hits / (double)a.period;  // np.mean(accepted[-period:])
    a.scaling[c] = exp(log(a.scaling[c]) + a.gamma_pow * (rate - 0.24));
  }
  if (lane == 0) a.acc_count[c] = 0;
}

// ------------------------------------------------------------------------------------------------
// C <- Sigma (proposal.py:509-510) and its Cholesky factor, one wave per chain, matrix in LDS,
// left-looking by columns with lane i = row i, sequential fma chain per element.
// ------------------------------------------------------------------------------------------------
struct CholArgs {
  int64_t N;
  int d;
  const double* am_sigma;  // folded [NP][DPAD/2+1][DPAD]
  double* Lk;              // [NP][DPAD][DPAD] k-major
  int32_t* flags;
};

template <int DPAD>
__device__ __forceinline__ double bcast_lane(double v, int src) {  // wave-uniform broadcast of lane `src`'s value
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Right-looking Cholesky with lane i holding row i of the (padded) matrix in registers; the k / j loops are fully
// unrolled so every register index is static and L[j][k] reaches the other lanes through v_readlane: no LDS, no
// barriers.  Element (i, j) receives the subtractions fma(-L[i][k], L[j][k], .) for k = 0..j-1 in ascending order,
// the same sequence as a left-looking dot product.
template <int DPAD>
__global__ void __launch_bounds__(64) k_chol(const CholArgs a) {
  constexpr int NS = DPAD / 2 + 1;
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const bool lj = lane < a.d;
  const int li = lane < DPAD ? lane : DPAD - 1;
  double A[DPAD];
  // row `lane` of Sigma (columns j <= lane) from the circulant fold; padded rows / columns = identity
#pragma unroll
  for (int j = 0; j < DPAD; ++j) {
    double v = (j == li) ? 1.0 : 0.0;
    if (lj && j < a.d && j <= li) {
      const int sl = li - j;
      v = sl <= DPAD / 2 ? a.am_sigma[((size_t)c * NS + sl) * DPAD + j]
                         : a.am_sigma[((size_t)c * NS + (DPAD - sl)) * DPAD + li];
    }
    A[j] = v;
  }
  bool ok = true;
#pragma unroll
  for (int k = 0; k < DPAD; ++k) {
    const double dkk = bcast_lane<DPAD>(A[k], k);
    ok = ok && (dkk > 0.0);
    const double lkk = sqrt(dkk);
    const double lik = (li == k) ? lkk : A[k] / lkk;  // L[i][k] for i >= k (garbage above the diagonal, never read)
    A[k] = lik;
#pragma unroll
    for (int j = k + 1; j < DPAD; ++j) {
      const double ljk = bcast_lane<DPAD>(lik, j);
      A[j] = fma(-lik, ljk, A[j]);
    }
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

// ------------------------------------------------------------------------------------------------
// Multi-level engine: Delayed Acceptance (tinyDA/chain.py:325-444, 475-483) and MLDA
// (chain.py:680-737, proposal.py:1502-1624) as ONE iterative state machine over base-level steps.
//
// All chains run the same schedule (subchain lengths are fixed), so control flow is uniform:
//   for each base step:   level-0 MH step (as k_mh_steps)
//     while the subchain of level k just completed (cnt[k] == sl[k]):  level k+1 acts:
//        y = state of level k (DA with randomize_subchain_length: the state after step `pick`)
//        skip-eval rule: chains whose level-k subchain accepted nothing record a rejection (chain.py:357-364)
//        alpha = exp(pi_{k+1}(y) - pi_{k+1}(x) + pi_k(x_start) - pi_k(y))
//        accept: level k+1 takes y.   reject: every level below reverts to theta_{k+1} with the
//        log-densities it had there.  (Invariant: after a step of level q, all levels j < q sit at theta_q;
//        S[j][q] caches level j's log-prior / log-like at theta_q.  This is what align_chain's identity
//        search (proposal.py:1469-1493) and the coarse re-append (chain.py:360-362, 394-396) amount to.)
// The accept flag of every upper-level step is also appended to the base proposal's `accepted` window
// (chain.py:363,389,397; proposal.py:1486), kept as a ring of the last `period` entries.
// ------------------------------------------------------------------------------------------------
constexpr int MAXLEV = 4;
constexpr int AEM_MP = 64;  // row stride of the per-chain error-model vectors / matrices in HBM
enum : uint32_t { STREAM_INDEX = 3 };

struct MLArgs {
  LevelDev lv[MAXLEV];
  int lds_y[MAXLEV];  // offset (doubles) of ytil / w of level k inside the staging region
  int lds_w[MAXLEV];
  int lds_total;      // doubles in the staging region
  PriorDev pr;
  int64_t N, NP;
  int d, S, prop_kind, nlev, randomize;
  int sl[MAXLEV];        // sl[k]: steps of level k per step of level k+1
  int cnt[MAXLEV];       // position inside the running subchain of level k at launch
  int64_t done[MAXLEV];  // local steps of level k completed before this launch (RNG step of level k)
  uint64_t seed;
  int64_t chain_offset;
  double* theta;    // [nlev][NP][DPAD]
  double* lp;       // [nlev][NP]
  double* ll;       // [nlev][NP]
  double* Sst;      // [npairs][2][NP], pair (j,q) at q(q-1)/2 + j
  int32_t* anyacc;  // [nlev][NP]
  double* ysnap;    // [NP][DPAD + 2] promoted coarse state of the running DA subchain
  int32_t* pick;    // [NP]
  const double* scaling;
  uint8_t* ring;    // [P][NP]
  int ring_P;
  int64_t ring_pos;
  const double* inc;  // [S][NP][DPAD]
  const double* u0;   // [S][NP]
  const double* u_rep[MAXLEV];  // replay uniforms of level k >= 1, row 0 = step done[k]; null -> Philox
  const double* ridx_rep;       // replay promoted index (DA), row 0 = fine iteration done[1]
  double* rec_params[MAXLEV];   // row 0 = first local step of level k in this launch
  double* rec_stats[MAXLEV];
  uint8_t* rec_acc[MAXLEV];
  // adaptive error model (host-sequenced mode): the kernel only advances level 0, whose likelihood is the
  // bias-corrected dense Gaussian of AdaptiveGaussianLogLike (distributions.py:404-425) with per-chain state
  int cascade;             // 1: upper levels act inside the kernel; 0: the host launches k_aem_action between blocks
  int aem_on;
  int aem_mp;              // padded output dimension (<= 64)
  const double* aem_bias;  // [NP][AEM_MP]          total bias of level 0
  const double* aem_P;     // [NP][AEM_MP][AEM_MP]  (Sigma_e + Sigma_bias)^-1 of level 0
  int64_t* sid;            // [nlev][NP] identity of the parameter vector each level currently holds
};

__device__ __forceinline__ constexpr int pair_index(int j, int q) { return q * (q - 1) / 2 + j; }

template <int DPAD, int NLEV>
__global__ void __launch_bounds__(256, 1) k_ml_steps(const MLArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int KS = DPAD / 4;
  constexpr int LDP = DPAD + 2;
  constexpr int EPT = DPAD >= 16 ? DPAD / 16 : 1;
  constexpr int QACT = DPAD / EPT;
  constexpr int NPAIR = NLEV * (NLEV - 1) / 2;

  const bool prior_dense = a.pr.kind == PRIOR_DENSE;
  double* s_prop = smem;
  double* s_red = s_prop + 16 * LDP;
  double* s_redp = s_red + 64;
  double* s_stage = s_redp + 64;           // ytil / w of every level
  double* s_py = s_stage + a.lds_total;    // dense prior: W mu
  double* s_R = s_py + (prior_dense ? a.pr.ncb * 16 : 0);  // AEM: residual tile [16][aem_mp + 2], then [16] ll slots
  const int RSa = a.aem_mp + 2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = blockIdx.x;
  const int c = tid >> 4, q_ = tid & 15;
  const int lc = lane & 15, hi = lane >> 4;
  const int64_t gct = tile * 16 + c;
  const int64_t gcl = tile * 16 + lc;
  const bool active = q_ < QACT;
  const uint32_t gchain = (uint32_t)(a.chain_offset + gcl);

#pragma unroll
  for (int k = 0; k < NLEV; ++k) {
    for (int i = tid; i < a.lv[k].m_pad; i += 256) {
      s_stage[a.lds_y[k] + i] = a.lv[k].ytil[i];
      if (a.lv[k].noise_kind == 1) s_stage[a.lds_w[k] + i] = a.lv[k].w[i];
    }
  }
  if (prior_dense)
    for (int i = tid; i < a.pr.ncb * 16; i += 256) s_py[i] = a.pr.wmu[i];

  double pm[KS], pinv[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    pm[kk] = a.pr.mean[4 * kk + hi];
    pinv[kk] = prior_dense ? 0.0 : a.pr.pinv[4 * kk + hi];
  }

  // ---- per-chain state: thread-mapped parameter slices, lane-mapped scalars ----
  double cur[NLEV][EPT], snp[EPT], prp[EPT], xin[EPT];
  double lp[NLEV], ll[NLEV], Slp[NPAIR > 0 ? NPAIR : 1], Sll[NPAIR > 0 ? NPAIR : 1];
  int anyacc[NLEV];
#pragma unroll
  for (int k = 0; k < NLEV; ++k) {
#pragma unroll
    for (int e = 0; e < EPT; ++e)
      cur[k][e] = active ? a.theta[((size_t)k * a.NP + gct) * DPAD + q_ * EPT + e] : 0.0;
    lp[k] = a.lp[(size_t)k * a.NP + gcl];
    ll[k] = a.ll[(size_t)k * a.NP + gcl];
    anyacc[k] = a.anyacc[(size_t)k * a.NP + gcl];
  }
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) {
    Slp[p] = a.Sst[((size_t)p * 2 + 0) * a.NP + gcl];
    Sll[p] = a.Sst[((size_t)p * 2 + 1) * a.NP + gcl];
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    snp[e] = active ? a.ysnap[gct * LDP + q_ * EPT + e] : 0.0;
    xin[e] = active ? a.inc[(size_t)gct * DPAD + q_ * EPT + e] : 0.0;
  }
  double snap_lp = a.ysnap[gcl * LDP + DPAD], snap_ll = a.ysnap[gcl * LDP + DPAD + 1];
  int pick = a.pick[gcl];
  const double scal_t = a.scaling[gct];
  const bool is_pcn = a.prop_kind == 1;
  const double keep_t = is_pcn ? sqrt(1.0 - scal_t * scal_t) : 1.0;
  double unext = a.u0[gcl];

  int cnt[NLEV];
  int64_t stepno[NLEV];  // local step index (global, for RNG) of the NEXT step of level k
  int nrec[NLEV];        // records written by this launch per level
#pragma unroll
  for (int k = 0; k < NLEV; ++k) {
    cnt[k] = a.cnt[k];
    stepno[k] = a.done[k];
    nrec[k] = 0;
  }
  int64_t ringpos = a.ring_pos;
  const double2* fbase = reinterpret_cast<const double2*>(a.lv[0].Apk) + lane;
  double2 f0[KS / 2], f1[KS / 2];
  __syncthreads();

  // evaluate level `k` at the state currently in s_prop (all 4 waves); returns (lp_n, ll_n) lane-mapped
  auto evaluate = [&](int k, double2 (&g0)[KS / 2], double2 (&g1)[KS / 2], double& lp_n, double& ll_n) {
    double th[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) th[kk] = s_prop[lc * LDP + 4 * kk + hi];
    double maha = 0.0;
    if (!prior_dense) {
      double p = 0.0;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const double dv = th[kk] - pm[kk];
        p += dv * dv * pinv[kk];
      }
      p = sum_rows(p);
      maha = p;
    } else {
      const double2* pbase = reinterpret_cast<const double2*>(a.pr.Wpk) + lane;
      double2 p0[KS / 2], p1[KS / 2];
      frag_load<DPAD>(pbase, wave, a.pr.ncb, p0);
      frag_load<DPAD>(pbase, wave + 4, a.pr.ncb, p1);
      double p = level_sse_partial<DPAD, 0>(a.pr.Wpk, a.pr.ncb, s_py, nullptr, th, wave, lane, p0, p1);
      p = sum_rows(p);
      if (lane < 16) s_redp[wave * 16 + lane] = p;
    }
    const LevelDev& L = a.lv[k];
    const bool dg = L.noise_kind == 1;
    if (a.aem_on && k == 0) {
      // residual tile, then per chain  -1/2 (F + bias - y)^T P (F + bias - y)  with that chain's bias and P
      (void)level_sse_partial<DPAD, 2>(L.Apk, L.ncb, s_stage + a.lds_y[k], s_R + lc * RSa, th, wave, lane, g0, g1);
      __syncthreads();
      const int MP = a.aem_mp;
      for (int cc = wave; cc < 16; cc += 4) {
        const int64_t gc = tile * 16 + cc;
        double* rrow = s_R + cc * RSa;
        double rb = 0.0;
        if (lane < MP) rb = rrow[lane] + a.aem_bias[gc * AEM_MP + lane];
        __builtin_amdgcn_wave_barrier();
        if (lane < MP) rrow[lane] = rb;
        __builtin_amdgcn_wave_barrier();
        double sacc = 0.0;
        if (lane < MP) {
          const double* Pc = a.aem_P + (size_t)gc * AEM_MP * AEM_MP + lane;
          for (int o = 0; o < MP; ++o) sacc = fma(Pc[(size_t)o * AEM_MP], rrow[o], sacc);
          sacc *= rb;
        }
        for (int off = 32; off >= 1; off >>= 1) sacc += __shfl_xor(sacc, off);
        if (lane == 0) s_R[16 * RSa + cc] = -0.5 * sacc;
      }
      if (prior_dense && lane < 16) {}  // (s_redp already written above)
      __syncthreads();
      ll_n = s_R[16 * RSa + lc];
      if (prior_dense) maha = ((s_redp[lc] + s_redp[16 + lc]) + s_redp[32 + lc]) + s_redp[48 + lc];
      lp_n = -0.5 * (a.pr.logconst + maha);
      return;
    }
    double sse = dg ? level_sse_partial<DPAD, 1>(L.Apk, L.ncb, s_stage + a.lds_y[k], s_stage + a.lds_w[k], th, wave, lane, g0, g1)
                    : level_sse_partial<DPAD, 0>(L.Apk, L.ncb, s_stage + a.lds_y[k], nullptr, th, wave, lane, g0, g1);
    sse = sum_rows(sse);
    if (lane < 16) s_red[wave * 16 + lane] = sse;
    __syncthreads();
    const double tot = ((s_red[lc] + s_red[16 + lc]) + s_red[32 + lc]) + s_red[48 + lc];
    if (prior_dense) maha = ((s_redp[lc] + s_redp[16 + lc]) + s_redp[32 + lc]) + s_redp[48 + lc];
    ll_n = dg ? -0.5 * tot : -0.5 * tot / L.var;
    lp_n = -0.5 * (a.pr.logconst + maha);
  };

  for (int s = 0; s < a.S; ++s) {
    // ================= level 0: one Metropolis-Hastings step =================
    frag_load<DPAD>(fbase, wave, a.lv[0].ncb, f0);
    frag_load<DPAD>(fbase, wave + 4, a.lv[0].ncb, f1);
    if (a.randomize && cnt[0] == 0) {  // DA: draw the promoted index of the subchain that starts now
      const int L0 = a.sl[0];
      if (a.ridx_rep) {
        const double r = a.ridx_rep[(size_t)(stepno[1] - a.done[1]) * a.N + (gcl < a.N ? gcl : 0)];
        pick = (r != r) ? L0 - 1 : (int)r + L0;  // reference index in [-L, -1] (chain.py:525-527)
      } else {
        const u32x4 r = philox4x32_10(u32x4{0u, (uint32_t)stepno[1], gchain, STREAM_INDEX}, (uint32_t)a.seed,
                                      (uint32_t)(a.seed >> 32));
        pick = (int)(((uint64_t)r.x * (uint64_t)L0) >> 32);
      }
    }
    if (active) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const double sx = scal_t * xin[e];
        prp[e] = is_pcn ? keep_t * cur[0][e] + sx : cur[0][e] + sx;
        s_prop[c * LDP + q_ * EPT + e] = prp[e];
      }
    }
    const double u = unext;
    if (s + 1 < a.S) {
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) xin[e] = a.inc[((size_t)(s + 1) * a.NP + gct) * DPAD + q_ * EPT + e];
      }
      unext = a.u0[(size_t)(s + 1) * a.NP + gcl];
    }
    __syncthreads();
    double lp_n, ll_n;
    evaluate(0, f0, f1, lp_n, ll_n);
    const double post_n = lp_n + ll_n;
    double alpha = is_pcn ? exp(ll_n - ll[0]) : exp(post_n - (lp[0] + ll[0]));
    if (post_n != post_n) alpha = 0.0;
    const bool acc0 = u < alpha;
    if (acc0) {
      lp[0] = lp_n;
      ll[0] = ll_n;
    }
    anyacc[0] |= acc0 ? 1 : 0;
    if (a.sid && acc0 && wave == 0 && lane < 16) a.sid[gcl] = stepno[0] + 1;  // a new parameter vector was created
    {
      const int accf = __shfl(acc0 ? 1 : 0, c);
      const bool take = a.randomize && cnt[0] == pick;
      const int takef = __shfl(take ? 1 : 0, c);
      if (take) {
        snap_lp = lp[0];
        snap_ll = ll[0];
      }
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          cur[0][e] = accf ? prp[e] : cur[0][e];
          if (takef) snp[e] = cur[0][e];
          const int j = q_ * EPT + e;
          if (a.rec_params[0] && gct < a.N && j < a.d)
            a.rec_params[0][((size_t)nrec[0] * a.N + gct) * a.d + j] = cur[0][e];
        }
      }
    }
    if (wave == 0 && lane < 16) {
      if (gcl < a.N) {
        const size_t r = (size_t)nrec[0] * a.N + gcl;
        if (a.rec_stats[0]) {
          a.rec_stats[0][r * 3 + 0] = lp[0];
          a.rec_stats[0][r * 3 + 1] = ll[0];
          a.rec_stats[0][r * 3 + 2] = lp[0] + ll[0];
        }
        if (a.rec_acc[0]) a.rec_acc[0][r] = acc0 ? 1 : 0;
      }
      a.ring[(size_t)(ringpos % a.ring_P) * a.NP + gcl] = acc0 ? 1 : 0;
    }
    ringpos += 1;
    nrec[0] += 1;
    stepno[0] += 1;
    cnt[0] += 1;

    // ================= upper levels whose subchain just completed =================
#pragma unroll
    for (int k = 0; k < NLEV - 1; ++k) {
      if (!a.cascade || cnt[k] != a.sl[k]) break;
      const int q = k + 1;
      const bool use_snap = (a.randomize != 0) && k == 0;
      // y -> LDS for the fragment gather
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) s_prop[c * LDP + q_ * EPT + e] = use_snap ? snp[e] : cur[k][e];
      }
      const double2* gb = reinterpret_cast<const double2*>(a.lv[q].Apk) + lane;
      double2 g0[KS / 2], g1[KS / 2];
      frag_load<DPAD>(gb, wave, a.lv[q].ncb, g0);
      frag_load<DPAD>(gb, wave + 4, a.lv[q].ncb, g1);
      __syncthreads();
      double lpq, llq;
      evaluate(q, g0, g1, lpq, llq);
      const double y_lp = use_snap ? snap_lp : lp[k], y_ll = use_snap ? snap_ll : ll[k];
      const int pkq = pair_index(k, q);
      double uq;
      if (a.u_rep[q])
        uq = a.u_rep[q][(size_t)(stepno[q] - a.done[q]) * a.N + (gcl < a.N ? gcl : 0)];
      else
        uq = accept_uniform(a.seed, gchain, (uint32_t)stepno[q], (uint32_t)q);
      const double alq = exp(((lpq + llq) - (lp[q] + ll[q])) + (Slp[pkq] + Sll[pkq]) - (y_lp + y_ll));
      const bool accq = (anyacc[k] != 0) && (uq < alq);
      const int accf = __shfl(accq ? 1 : 0, c);
      // parameters: accept -> level q (and level k, if a promoted intermediate state) take y;
      //             reject -> all levels below q return to theta_q
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const double yv = use_snap ? snp[e] : cur[k][e];
          if (accf) {
            cur[q][e] = yv;
            cur[k][e] = yv;
          } else {
#pragma unroll
            for (int j = 0; j < q; ++j) cur[j][e] = cur[q][e];
          }
        }
      }
      if (accq) {
        lp[q] = lpq;
        ll[q] = llq;
        lp[k] = y_lp;
        ll[k] = y_ll;
      } else {
#pragma unroll
        for (int j = 0; j < q; ++j) {
          lp[j] = Slp[pair_index(j, q)];
          ll[j] = Sll[pair_index(j, q)];
        }
      }
#pragma unroll
      for (int j = 0; j < q; ++j) {
#pragma unroll
        for (int q2 = j + 1; q2 <= q; ++q2) {
          Slp[pair_index(j, q2)] = lp[j];
          Sll[pair_index(j, q2)] = ll[j];
        }
      }
      anyacc[k] = 0;
      if (q < NLEV - 1) anyacc[q] |= accq ? 1 : 0;
      // records of level q and the alignment entry in the base proposal's accepted window
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const int j = q_ * EPT + e;
          if (a.rec_params[q] && gct < a.N && j < a.d)
            a.rec_params[q][((size_t)nrec[q] * a.N + gct) * a.d + j] = cur[q][e];
        }
      }
      if (wave == 0 && lane < 16) {
        if (gcl < a.N) {
          const size_t r = (size_t)nrec[q] * a.N + gcl;
          if (a.rec_stats[q]) {
            a.rec_stats[q][r * 3 + 0] = lp[q];
            a.rec_stats[q][r * 3 + 1] = ll[q];
            a.rec_stats[q][r * 3 + 2] = lp[q] + ll[q];
          }
          if (a.rec_acc[q]) a.rec_acc[q][r] = accq ? 1 : 0;
        }
        a.ring[(size_t)(ringpos % a.ring_P) * a.NP + gcl] = accq ? 1 : 0;
      }
      ringpos += 1;
      nrec[q] += 1;
      stepno[q] += 1;
      cnt[k] = 0;
      cnt[q] += 1;
    }
  }

  // ---- write the state back ----
#pragma unroll
  for (int k = 0; k < NLEV; ++k) {
    if (active) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) a.theta[((size_t)k * a.NP + gct) * DPAD + q_ * EPT + e] = cur[k][e];
    }
    if (wave == 0 && lane < 16) {
      a.lp[(size_t)k * a.NP + gcl] = lp[k];
      a.ll[(size_t)k * a.NP + gcl] = ll[k];
      a.anyacc[(size_t)k * a.NP + gcl] = anyacc[k];
    }
  }
  if (wave == 0 && lane < 16) {
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      a.Sst[((size_t)p * 2 + 0) * a.NP + gcl] = Slp[p];
      a.Sst[((size_t)p * 2 + 1) * a.NP + gcl] = Sll[p];
    }
    a.ysnap[gcl * LDP + DPAD] = snap_lp;
    a.ysnap[gcl * LDP + DPAD + 1] = snap_ll;
    a.pick[gcl] = pick;
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) a.ysnap[gct * LDP + q_ * EPT + e] = snp[e];
  }
}

// ------------------------------------------------------------------------------------------------
// Adaptive error model (Cui et al. 2019): one step of level q >= 1 for every chain, one wave per chain, followed by
// the error-model update of level q-1.  Used in the host-sequenced mode (MLArgs::cascade = 0): the tile kernel
// advances the base level, this kernel performs what DAChain.sample (chain.py:353-402, 446-523) / MLDA.make_mlda_proposal
// (proposal.py:1515-1578) / MLDAChain.sample (chain.py:711-765) do once the subchain below has finished.
// Sizes are "parity sizes": output dimension m <= 64 (lane = observation), per-chain m x m matrices in HBM; the
// reference itself re-inverts an m x m matrix per chain per step (distributions.py:402).
// ------------------------------------------------------------------------------------------------
struct AemArgs {
  int64_t N, NP, chain_offset;
  int d, DP, m, MP, nlev, q;
  int is_da, dependent, prop_kind;
  uint64_t seed;
  int64_t step;            // index of this level-q step (RNG / replay row)
  const double* A[MAXLEV];     // row-major [m][d]
  const double* ytil[MAXLEV];  // y - b, [MP]   (residual r = A theta - ytil = F - y)
  const double* data[MAXLEV];  // y, [MP]       (model output F = r + y)
  const double* cov[MAXLEV];   // adaptive levels: Sigma_e [MP][MP]
  double var_finest;
  const double* pr_mean;   // [DP]
  const double* pr_W;      // [d][d] whitening (L^-1 of the prior covariance), row-major
  double pr_logdet;
  double* theta;   // [nlev][NP][DP]
  double* lp;      // [nlev][NP]
  double* ll;
  double* Sst;     // [npairs][2][NP]
  int32_t* anyacc; // [nlev][NP]
  int64_t* sid;    // [nlev][NP]
  double* bias_tot[MAXLEV];  // [NP][MP]     adaptive levels
  double* cov_inv[MAXLEV];   // [NP][MP][MP]
  double* b_mu[MAXLEV];      // trackers of levels >= 1: [NP][MP]
  double* b_sig[MAXLEV];     // [NP][MP][MP]
  double* mdiff[MAXLEV];     // [NP][MP]
  int64_t b_t;               // recursion counter of level q's tracker before this update
  const double* scaling;     // [NP] (pCN beta for the state-dependent q terms)
  const double* u_rep;       // [N] replay uniform of this step (NaN = none drawn) or null
  uint8_t* ring;
  int ring_P;
  int64_t ring_pos;
  double* rec_params;  // row of this step, [N][d] (may be null)
  double* rec_stats;
  uint8_t* rec_acc;
};

__global__ void __launch_bounds__(64) k_aem_action(const AemArgs a) {
  constexpr int LDM = AEM_MP + 1;
  __shared__ double s_M[AEM_MP * LDM];
  __shared__ double s_v[4 * AEM_MP];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const int q = a.q, k = a.q - 1, nl = a.nlev, MP = a.MP, d = a.d;
  const bool lo = lane < a.m, lj = lane < d;
  auto TH = [&](int lev) { return a.theta + ((size_t)lev * a.NP + c) * a.DP; };
  auto bsum = [&](double v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
  };
  // F_lev(theta)[lane] - ytil_lev[lane]  (theta given through LDS vector s_v[0..d))
  auto resid = [&](int lev) {
    double f = 0.0;
    if (lo) {
      const double* Ar = a.A[lev] + (size_t)lane * d;
      for (int j = 0; j < d; ++j) f = fma(Ar[j], s_v[j], f);
      f -= a.ytil[lev][lane];
    }
    return f;
  };
  // -1/2 r^T P r with chain c's inverse of adaptive level lev; r given per lane (already bias corrected)
  auto quad = [&](int lev, double r) {
    __syncthreads();
    s_v[AEM_MP + lane] = lo ? r : 0.0;
    __syncthreads();
    double s = 0.0;
    if (lo) {
      const double* Pc = a.cov_inv[lev] + (size_t)c * MP * MP + lane;
      for (int o = 0; o < a.m; ++o) s = fma(Pc[(size_t)o * MP], s_v[AEM_MP + o], s);
      s *= r;
    }
    return -0.5 * bsum(s);
  };
  auto loglike_of = [&](int lev, double r0) {  // r0 = F - ytil without bias
    if (lev == nl - 1) return -0.5 * bsum(lo ? r0 * r0 : 0.0) / a.var_finest;
    return quad(lev, lo ? r0 + a.bias_tot[lev][c * MP + lane] : 0.0);
  };

  // ---------------- the level-q decision ----------------
  const double yj = lj ? TH(k)[lane] : 0.0, xj = lj ? TH(q)[lane] : 0.0;
  const double y_lp = a.lp[(size_t)k * a.NP + c], y_ll = a.ll[(size_t)k * a.NP + c];
  const double x_lp = a.lp[(size_t)q * a.NP + c], x_ll = a.ll[(size_t)q * a.NP + c];
  const int pkq = pair_index(k, q);
  const double st_lp = a.Sst[((size_t)pkq * 2 + 0) * a.NP + c], st_ll = a.Sst[((size_t)pkq * 2 + 1) * a.NP + c];
  const bool any = a.anyacc[(size_t)k * a.NP + c] != 0;
  __syncthreads();
  if (lane < AEM_MP) s_v[lane] = yj;
  __syncthreads();
  const double rq_y = resid(q);                   // F_q(y) - ytil_q
  const double rk_y = a.dependent ? resid(k) : 0.0;  // F_k(y) - ytil_k
  const double lpn = y_lp;  // same prior, same parameters (posterior.py:92)
  const double lln = loglike_of(q, rq_y);
  double alpha;
  if (a.dependent) {  // chain.py:446-473
    // bias at the proposal and the coarse density of the subchain start under it
    const double bias_next = (rq_y + a.data[q][lane < MP ? lane : 0]) - (rk_y + a.data[k][lane < MP ? lane : 0]);
    __syncthreads();
    if (lane < AEM_MP) s_v[lane] = xj;  // subchain start = the fine state
    __syncthreads();
    const double rk_x = resid(k);
    const double ll_b = quad(k, lo ? rk_x + bias_next : 0.0);
    double q_xy = 0.0, q_yx = 0.0;
    if (a.prop_kind == 1) {  // pCN transition densities (proposal.py:364-369) between the fine links
      const double beta = a.scaling[c], kp = sqrt(1.0 - beta * beta);
      for (int dir = 0; dir < 2; ++dir) {
        __syncthreads();
        if (lane < AEM_MP) s_v[2 * AEM_MP + lane] = dir == 0 ? yj - kp * xj : xj - kp * yj;
        __syncthreads();
        double w = 0.0;
        if (lj) {
          const double* Wr = a.pr_W + (size_t)lane * d;
          for (int j = 0; j <= lane; ++j) w = fma(Wr[j], s_v[2 * AEM_MP + j], w);
        }
        const double maha = bsum(lj ? w * w : 0.0) / (beta * beta);
        const double v = -0.5 * (d * 1.8378770664093453 + a.pr_logdet + d * log(beta * beta) + maha);
        if (dir == 0) q_xy = v; else q_yx = v;
      }
    }
    const double n1 = (lpn + lln) + q_yx, n2 = (st_lp + ll_b) + q_xy;
    const double d1 = (x_lp + x_ll) + q_xy, d2 = (y_lp + y_ll) + q_yx;
    alpha = exp((n1 < n2 ? n1 : n2) - (d1 < d2 ? d1 : d2));
  } else {
    alpha = exp(((lpn + lln) - (x_lp + x_ll)) + (st_lp + st_ll) - (y_lp + y_ll));  // chain.py:475-483, proposal.py:1615-1624
  }
  double u;
  if (a.u_rep) u = a.u_rep[c];
  else u = accept_uniform(a.seed, (uint32_t)(a.chain_offset + c), (uint32_t)a.step, (uint32_t)q);
  const bool acc = any && (u < alpha);

  // ---------------- alignment (chain.py:357-398; proposal.py:1469-1493) ----------------
  if (acc) {
    if (lane < a.DP) TH(q)[lane] = lj ? yj : 0.0;
  } else {
    for (int j = 0; j < q; ++j)
      if (lane < a.DP) TH(j)[lane] = lj ? xj : 0.0;
  }
  __syncthreads();
  if (lane == 0) {
    if (acc) {
      a.lp[(size_t)q * a.NP + c] = lpn;
      a.ll[(size_t)q * a.NP + c] = lln;
      a.sid[(size_t)q * a.NP + c] = a.sid[(size_t)k * a.NP + c];
    } else {
      for (int j = 0; j < q; ++j) {
        const int p = pair_index(j, q);
        a.lp[(size_t)j * a.NP + c] = a.Sst[((size_t)p * 2 + 0) * a.NP + c];
        a.ll[(size_t)j * a.NP + c] = a.Sst[((size_t)p * 2 + 1) * a.NP + c];
        a.sid[(size_t)j * a.NP + c] = a.sid[(size_t)q * a.NP + c];
      }
    }
    for (int j = 0; j < q; ++j)
      for (int q2 = j + 1; q2 <= q; ++q2) {
        const int p = pair_index(j, q2);
        a.Sst[((size_t)p * 2 + 0) * a.NP + c] = a.lp[(size_t)j * a.NP + c];
        a.Sst[((size_t)p * 2 + 1) * a.NP + c] = a.ll[(size_t)j * a.NP + c];
      }
    a.anyacc[(size_t)k * a.NP + c] = 0;
    if (q < nl - 1) a.anyacc[(size_t)q * a.NP + c] |= acc ? 1 : 0;
    a.ring[(size_t)(a.ring_pos % a.ring_P) * a.NP + c] = acc ? 1 : 0;
    if (a.rec_stats) {
      const double l1 = a.lp[(size_t)q * a.NP + c], l2 = a.ll[(size_t)q * a.NP + c];
      a.rec_stats[c * 3 + 0] = l1;
      a.rec_stats[c * 3 + 1] = l2;
      a.rec_stats[c * 3 + 2] = l1 + l2;
    }
    if (a.rec_acc) a.rec_acc[c] = acc ? 1 : 0;
  }
  if (a.rec_params && lj) a.rec_params[c * d + lane] = acc ? yj : xj;
  __syncthreads();

  // ---------------- error model update (chain.py:485-523, :739-765; proposal.py:1547-1578) ----------------
  const double cj = acc ? yj : xj;  // theta_q = theta_k now
  __syncthreads();
  if (lane < AEM_MP) s_v[lane] = cj;
  __syncthreads();
  const double rq = resid(q), rk = resid(k);
  const double diff_new = lo ? (rq + a.data[q][lane]) - (rk + a.data[k][lane]) : 0.0;
  double* md = a.mdiff[q] + c * MP;
  double* Sg = a.b_sig[q] + (size_t)c * MP * MP;
  const double t = (double)a.b_t;
  double xupd;  // the sample fed to the running moments
  if (a.dependent) {
    xupd = lo ? (rq + a.data[q][lane]) - ((rk + a.data[k][lane]) + md[lane]) : 0.0;  // chain.py:505-507
    if (lo) md[lane] = diff_new;
    __syncthreads();
    if (lane < AEM_MP) s_v[AEM_MP + lane] = xupd;
    __syncthreads();
    if (lo)
      for (int i = 0; i < a.m; ++i) {  // utils.py:199  Sigma <- (t-1)/t Sigma + 1/t x x^T
        const double xi = s_v[AEM_MP + i];
        Sg[(size_t)i * MP + lane] = (t - 1.0) / t * Sg[(size_t)i * MP + lane] + 1.0 / t * (xi * xupd);
      }
  } else {
    const double dm = (a.is_da || acc) ? diff_new : (lo ? md[lane] : 0.0);  // MLDA refreshes the difference on accept only
    if (lo) md[lane] = dm;
    double* mu = a.b_mu[q] + c * MP;
    const double mu_o = lo ? mu[lane] : 0.0;
    const double mu_n = (1.0 / (t + 1.0)) * (t * mu_o + dm);  // utils.py:113-122 with sd = 1, eps = 0
    __syncthreads();
    if (lane < AEM_MP) {
      s_v[AEM_MP + lane] = dm;
      s_v[2 * AEM_MP + lane] = mu_o;
      s_v[3 * AEM_MP + lane] = mu_n;
    }
    __syncthreads();
    if (lo) {
      const double ca = (t - 1.0) / t, cb = 1.0 / t;
      for (int i = 0; i < a.m; ++i) {
        const double M = (t * (s_v[2 * AEM_MP + i] * mu_o) - (t + 1.0) * (s_v[3 * AEM_MP + i] * mu_n)) + s_v[AEM_MP + i] * dm;
        Sg[(size_t)i * MP + lane] = ca * Sg[(size_t)i * MP + lane] + cb * M;
      }
      mu[lane] = mu_n;
    }
  }
  __threadfence_block();
  __syncthreads();
  // total bias of level k: state-dependent = the last difference; otherwise sums over the trackers of levels >= q
  double bt = 0.0;
  if (lo) {
    if (a.dependent) bt = md[lane];
    else
      for (int p = q; p < nl; ++p) bt += a.b_mu[p][c * MP + lane];
    a.bias_tot[k][c * MP + lane] = bt;
  }
  // Sigma_e + Sigma_bias into LDS (row i = lane), and the 1e-9 rule of distributions.py:399-402
  bool big = false;
  if (lo) {
    for (int j = 0; j < a.m; ++j) {
      // symmetric matrices: element (lane, j) is read as (j, lane), the entry this very lane wrote above
      double sb = 0.0;
      if (a.dependent) sb = Sg[(size_t)j * MP + lane];
      else
        for (int p = q; p < nl; ++p) sb += a.b_sig[p][(size_t)c * MP * MP + (size_t)j * MP + lane];
      big = big || !(sb < 1e-9);
      s_M[lane * LDM + j] = a.cov[k][(size_t)j * MP + lane] + sb;
    }
  }
  const bool refresh = __ballot(big) != 0ull;
  __syncthreads();
  if (refresh) {
    // inverse through the Cholesky factor: M = L L^T, W = L^-1, P = W^T W
    for (int kk = 0; kk < a.m; ++kk) {
      double sacc = 0.0;
      if (lane >= kk && lo) {
        sacc = s_M[lane * LDM + kk];
        for (int p = 0; p < kk; ++p) sacc = fma(-s_M[lane * LDM + p], s_M[kk * LDM + p], sacc);
      }
      const double lkk = sqrt(__shfl(sacc, kk));
      if (lane >= kk && lo) s_M[lane * LDM + kk] = lane == kk ? lkk : sacc / lkk;
      __syncthreads();
    }
    // W = L^-1 : lane = column j, forward substitution down the rows; stored in the upper triangle region via a second pass
    double Wc[AEM_MP];
#pragma unroll 1
    for (int i = 0; i < a.m; ++i) {
      double v = 0.0;
      if (lo && i >= lane) {
        if (i == lane) v = 1.0 / s_M[i * LDM + i];
        else {
          double sacc = 0.0;
          for (int p = lane; p < i; ++p) sacc = fma(s_M[i * LDM + p], Wc[p], sacc);
          v = -sacc / s_M[i * LDM + i];
        }
      }
      Wc[i] = v;
    }
    __syncthreads();
    // s_M <- W (row i, column j = lane)
    for (int i = 0; i < a.m; ++i)
      if (lo) s_M[i * LDM + lane] = Wc[i];
    __syncthreads();
    if (lo) {
      double* Pc = a.cov_inv[k] + (size_t)c * MP * MP;
      for (int i = 0; i < a.m; ++i) {  // P[i][lane] = sum_r W[r][i] W[r][lane]
        double sacc = 0.0;
        const int r0 = i > lane ? i : lane;
        for (int r = r0; r < a.m; ++r) sacc = fma(s_M[r * LDM + i], s_M[r * LDM + lane], sacc);
        Pc[(size_t)i * MP + lane] = sacc;
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  // update_link of level k's latest link (posterior.py:112-134)
  const double llk = quad(k, lo ? rk + bt : 0.0);
  if (lane == 0) {
    a.ll[(size_t)k * a.NP + c] = llk;
    const int64_t idk = a.sid[(size_t)k * a.NP + c];
    for (int q2 = q; q2 < nl; ++q2)
      if (a.sid[(size_t)q2 * a.NP + c] == idk) a.Sst[((size_t)pair_index(k, q2) * 2 + 1) * a.NP + c] = llk;
  }
}

// ------------------------------------------------------------------------------------------------
// DREAM(Z)  (tinyDA/proposal.py:608-852) for single-level chains.
//   k_dreamz_draw   wave per chain: everything make_proposal draws that does not depend on the chain state
//                   (archive row indices, crossover index, subspace mask, (1+e) gamma, eps) for a block of steps
//   k_dreamz_steps  16-chain tile: theta' = theta + mask ((1+e) gamma (sum Z_r1 - sum Z_r2) + eps) with the rows
//                   gathered from the chain's archive in HBM, evaluation (linear model on MFMA, or the
//                   Rosenbrock chain on VALU), accept, record, archive append (proposal.py:794)
//   k_dreamz_adapt  wave per chain: archive column sums catch-up, global scaling, pCR update (proposal.py:797-809)
// RNG stream 4 (block = what, step, chain):  block i < delta : r1 = (x0*M)>>32, r2 = (x1*(M-1))>>32, r2 += r2>=r1
//   block delta : mCR by inverse cdf of u53(x0,x1) over pCR, forced index = (x2*d)>>32
//   block delta+1+j : subspace uniform u53(x0,x1) and e-uniform u53(x2,x3) of parameter j;  eps_j from stream 0.
// ------------------------------------------------------------------------------------------------
enum : uint32_t { STREAM_DREAM = 4 };
constexpr int MAX_NCR = 8;
constexpr int MAX_DELTA = 4;

struct DreamDrawArgs {
  int64_t N, NP, chain_offset;
  int d, S, delta, nCR;
  int64_t step0;   // proposal.t at s = 0
  int64_t M_base;  // archive rows visible at s = 0
  int grow;        // 1: archive grows by one row per step inside the block (per-chain DREAMZ); 0: frozen (shared DREAM)
  uint64_t seed;
  double b, b_star;
  const double* scaling;  // [NP]
  const double* pCR;      // [NP][MAX_NCR]
  double* coef;           // [S][NP][DPAD]  mask * (1+e) * gamma
  double* epsm;           // [S][NP][DPAD]  mask * eps
  int32_t* ridx;          // [S][NP][2*MAX_DELTA]
  double* u;              // [S][NP]
  int32_t* mcr_last;      // [NP] crossover index of the block's last step (proposal.py:801)
  // replay (all may be null) at step0: r [.][N][delta][2] int32, mcr [.][N] int32, sub_u/e_u/eps_n [.][N][d], forced [.][N] int32, u [.][N]
  const int32_t* r_rep;
  const int32_t* mcr_rep;
  const double* sub_rep;
  const int32_t* forced_rep;
  const double* e_rep;
  const double* eps_rep;
  const double* u_rep;
  double* eps_export;  // [.][N][d] standard normals actually used (null = off)
  double* u_export;
};

template <int DPAD>
__global__ void __launch_bounds__(64) k_dreamz_draw(const DreamDrawArgs a) {
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  const bool real_chain = c < a.N;
  const uint32_t gc = (uint32_t)(a.chain_offset + c);
  const bool lj = lane < a.d;
  const double scaling = a.scaling[c];
  double cdf[MAX_NCR];
  {
    double run = 0.0;
#pragma unroll
    for (int k = 0; k < MAX_NCR; ++k) {
      run += k < a.nCR ? a.pCR[c * MAX_NCR + k] : 0.0;
      cdf[k] = run;
    }
  }
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  int mcr = 0;
  for (int s = 0; s < a.S; ++s) {
    const uint32_t step = (uint32_t)(a.step0 + s);
    const int64_t M = a.M_base + (a.grow ? s : 0);
    const size_t row = (size_t)s * a.N + c;  // replay / export row (real chains only)
    // ---- archive row pairs (proposal.py:823-826) ----
    if (lane < a.delta) {
      int r1, r2;
      if (a.r_rep && real_chain) {
        r1 = a.r_rep[(row * a.delta + lane) * 2 + 0];
        r2 = a.r_rep[(row * a.delta + lane) * 2 + 1];
      } else {
        const u32x4 x = philox4x32_10(u32x4{(uint32_t)lane, step, gc, STREAM_DREAM}, k0, k1);
        r1 = (int)(((uint64_t)x.x * (uint64_t)M) >> 32);
        r2 = (int)(((uint64_t)x.y * (uint64_t)(M - 1)) >> 32);
        r2 += r2 >= r1 ? 1 : 0;
      }
      a.ridx[((size_t)s * a.NP + c) * (2 * MAX_DELTA) + 2 * lane + 0] = r1;
      a.ridx[((size_t)s * a.NP + c) * (2 * MAX_DELTA) + 2 * lane + 1] = r2;
    }
    // ---- crossover index and the index forced when the subspace is empty (proposal.py:829-839) ----
    int forced;
    {
      const u32x4 x = philox4x32_10(u32x4{(uint32_t)a.delta, step, gc, STREAM_DREAM}, k0, k1);
      if (a.mcr_rep && real_chain) {
        mcr = a.mcr_rep[row];
        forced = a.forced_rep[row];
      } else {
        const double uu = u53(x.x, x.y);
        mcr = a.nCR - 1;
        for (int k = a.nCR - 1; k >= 0; --k)
          if (cdf[k] > uu) mcr = k;
        forced = (int)(((uint64_t)x.z * (uint64_t)a.d) >> 32);
      }
    }
    const double CR = (double)(mcr + 1) / (double)a.nCR;
    // ---- per-parameter draws ----
    double su = 2.0, eu = 0.5, en = 0.0;
    if (lj) {
      if (a.sub_rep && real_chain) {
        su = a.sub_rep[row * a.d + lane];
        eu = a.e_rep[row * a.d + lane];
        en = a.eps_rep[row * a.d + lane];
      } else {
        const u32x4 x = philox4x32_10(u32x4{(uint32_t)(a.delta + 1 + lane), step, gc, STREAM_DREAM}, k0, k1);
        su = u53(x.x, x.y);
        eu = u53(x.z, x.w);
        double z0, z1;
        normal_pair(a.seed, gc, step, STREAM_PROPOSAL, (uint32_t)(lane >> 1), z0, z1);
        en = (lane & 1) ? z1 : z0;
      }
      if (a.eps_export && real_chain) a.eps_export[row * a.d + lane] = en;
    }
    bool ind = lj && (su < CR);
    const unsigned long long bal = __ballot(ind);
    int dsub = __popcll(bal);
    if (dsub == 0) {  // proposal.py:838-839
      ind = lane == forced;
      dsub = 1;
    }
    const double gam = scaling * 2.38 / sqrt((double)(2 * a.delta * dsub));  // proposal.py:842-844
    const double e = -a.b + (a.b - (-a.b)) * eu;
    const double eps = 0.0 + a.b_star * en;
    if (lane < DPAD) {
      a.coef[((size_t)s * a.NP + c) * DPAD + lane] = ind ? (1.0 + e) * gam : 0.0;
      a.epsm[((size_t)s * a.NP + c) * DPAD + lane] = ind ? eps : 0.0;
    }
    if (lane == 0) {
      double u = 0.5;
      if (real_chain) {
        u = a.u_rep ? a.u_rep[row] : accept_uniform(a.seed, gc, step, 0u);
        if (a.u_export) a.u_export[row] = u;
      }
      a.u[(size_t)s * a.NP + c] = u;
    }
  }
  if (lane == 0) a.mcr_last[c] = mcr;
}

enum : int { MODEL_LINEAR = 0, MODEL_ROSENBROCK = 1 };

struct DreamStepArgs {
  LevelDev lv;
  PriorDev pr;
  int model;        // MODEL_*
  double ros_a, ros_b, ros_data;
  int64_t N, NP;
  int d, S, delta;
  int64_t M_base;      // archive rows at s = 0
  int shared;          // 1: one archive for all chains (frozen inside the block), 0: per chain (grows every step)
  int64_t cap;         // rows per archive
  double* arch;        // per chain [NP][cap][DPAD] / shared [cap][DPAD]
  double* theta;       // [NP][DPAD]
  double* theta_prev;  // [NP][DPAD] state before the block's last step (jumping distance, proposal.py:800)
  double* lp;
  double* ll;
  int32_t* acc_count;
  const double* coef;
  const double* epsm;
  const int32_t* ridx;
  const double* u;
  double* rec_params;
  double* rec_stats;
  uint8_t* rec_acc;
  double* blk_states;  // [S][NP][DPAD] states of this block (shared mode: appended to the archive afterwards)
};

template <int DPAD>
__global__ void __launch_bounds__(256, 1) k_dreamz_steps(const DreamStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int KS = DPAD / 4;
  constexpr int LDP = DPAD + 2;
  constexpr int EPT = DPAD >= 16 ? DPAD / 16 : 1;
  constexpr int QACT = DPAD / EPT;
  const bool diag = a.lv.noise_kind == 1;
  const bool prior_dense = a.pr.kind == PRIOR_DENSE;
  const bool linear = a.model == MODEL_LINEAR;
  double* s_prop = smem;
  double* s_red = s_prop + 16 * LDP;
  double* s_redp = s_red + 64;
  double* s_y = s_redp + 64;
  double* s_w = s_y + (linear ? a.lv.m_pad : 0);
  double* s_py = s_w + ((linear && diag) ? a.lv.m_pad : 0);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = blockIdx.x;
  const int c = tid >> 4, q_ = tid & 15;
  const int lc = lane & 15, hi = lane >> 4;
  const int64_t gct = tile * 16 + c;
  const int64_t gcl = tile * 16 + lc;
  const bool active = q_ < QACT;
  if (linear) {
    for (int i = tid; i < a.lv.m_pad; i += 256) {
      s_y[i] = a.lv.ytil[i];
      if (diag) s_w[i] = a.lv.w[i];
    }
  }
  if (prior_dense)
    for (int i = tid; i < a.pr.ncb * 16; i += 256) s_py[i] = a.pr.wmu[i];
  double pm[KS], pinv[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    pm[kk] = a.pr.mean[4 * kk + hi];
    pinv[kk] = prior_dense ? 0.0 : a.pr.pinv[4 * kk + hi];
  }
  double cur[EPT], prp[EPT], prev[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    cur[e] = active ? a.theta[gct * DPAD + q_ * EPT + e] : 0.0;
    prev[e] = cur[e];
  }
  double lp = a.lp[gcl], ll = a.ll[gcl];
  int nacc = 0;
  double* arch_c = a.shared ? a.arch : a.arch + (size_t)gct * a.cap * DPAD;
  const double2* fbase = reinterpret_cast<const double2*>(a.lv.Apk) + lane;
  __syncthreads();

  for (int s = 0; s < a.S; ++s) {
    double2 f0[KS / 2], f1[KS / 2];
    if (linear) {
      frag_load<DPAD>(fbase, wave, a.lv.ncb, f0);
      frag_load<DPAD>(fbase, wave + 4, a.lv.ncb, f1);
    }
    // ---- proposal (proposal.py:850-852) ----
    if (active) {
      double z1[EPT], z2[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) z1[e] = z2[e] = 0.0;
      for (int i = 0; i < a.delta; ++i) {
        const int r1 = a.ridx[((size_t)s * a.NP + gct) * (2 * MAX_DELTA) + 2 * i + 0];
        const int r2 = a.ridx[((size_t)s * a.NP + gct) * (2 * MAX_DELTA) + 2 * i + 1];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          z1[e] += arch_c[(size_t)r1 * DPAD + q_ * EPT + e];
          z2[e] += arch_c[(size_t)r2 * DPAD + q_ * EPT + e];
        }
      }
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const size_t o = ((size_t)s * a.NP + gct) * DPAD + q_ * EPT + e;
        const double jump = a.coef[o] * (z1[e] - z2[e]) + a.epsm[o];
        prp[e] = cur[e] + jump;
        s_prop[c * LDP + q_ * EPT + e] = prp[e];
      }
    }
    const double u = a.u[(size_t)s * a.NP + gcl];
    __syncthreads();
    // ---- prior ----
    double th[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) th[kk] = s_prop[lc * LDP + 4 * kk + hi];
    double maha = 0.0;
    if (!prior_dense) {
      double p = 0.0;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const double dv = th[kk] - pm[kk];
        p += dv * dv * pinv[kk];
      }
      p = sum_rows(p);
      maha = p;
    } else {
      const double2* pbase = reinterpret_cast<const double2*>(a.pr.Wpk) + lane;
      double2 p0[KS / 2], p1[KS / 2];
      frag_load<DPAD>(pbase, wave, a.pr.ncb, p0);
      frag_load<DPAD>(pbase, wave + 4, a.pr.ncb, p1);
      double p = level_sse_partial<DPAD, 0>(a.pr.Wpk, a.pr.ncb, s_py, nullptr, th, wave, lane, p0, p1);
      p = sum_rows(p);
      if (lane < 16) s_redp[wave * 16 + lane] = p;
    }
    // ---- likelihood ----
    double ll_n;
    if (linear) {
      double sse = diag ? level_sse_partial<DPAD, 1>(a.lv.Apk, a.lv.ncb, s_y, s_w, th, wave, lane, f0, f1)
                        : level_sse_partial<DPAD, 0>(a.lv.Apk, a.lv.ncb, s_y, nullptr, th, wave, lane, f0, f1);
      sse = sum_rows(sse);
      if (lane < 16) s_red[wave * 16 + lane] = sse;
      __syncthreads();
      const double tot = ((s_red[lc] + s_red[16 + lc]) + s_red[32 + lc]) + s_red[48 + lc];
      ll_n = diag ? -0.5 * tot : -0.5 * tot / a.lv.var;
    } else {
      // Rosenbrock chain: f = sum_i (a - x_i)^2 + b (x_{i+1} - x_i^2)^2 ; loglike = -0.5 (f - data)^2 / var
      double f = 0.0;
      for (int i = 0; i + 1 < a.d; ++i) {
        const double x0 = s_prop[lc * LDP + i], x1 = s_prop[lc * LDP + i + 1];
        const double t0 = a.ros_a - x0, t1 = x1 - x0 * x0;
        f += t0 * t0 + a.ros_b * (t1 * t1);
      }
      const double r = f - a.ros_data;
      ll_n = -0.5 * (r * r) / a.lv.var;
      __syncthreads();
    }
    if (prior_dense) maha = ((s_redp[lc] + s_redp[16 + lc]) + s_redp[32 + lc]) + s_redp[48 + lc];
    const double lp_n = -0.5 * (a.pr.logconst + maha);
    const double post_n = lp_n + ll_n;
    double alpha = exp(post_n - (lp + ll));
    if (post_n != post_n) alpha = 0.0;
    const bool acc = u < alpha;
    if (acc) {
      lp = lp_n;
      ll = ll_n;
    }
    nacc += acc ? 1 : 0;
    if (wave == 0 && lane < 16 && gcl < a.N) {
      const size_t r = (size_t)s * a.N + gcl;
      if (a.rec_stats) {
        a.rec_stats[r * 3 + 0] = lp;
        a.rec_stats[r * 3 + 1] = ll;
        a.rec_stats[r * 3 + 2] = lp + ll;
      }
      if (a.rec_acc) a.rec_acc[r] = acc ? 1 : 0;
    }
    const int accf = __shfl(acc ? 1 : 0, c);
    if (active) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        if (s == a.S - 1) prev[e] = cur[e];
        cur[e] = accf ? prp[e] : cur[e];
        const int j = q_ * EPT + e;
        if (a.rec_params && gct < a.N && j < a.d) a.rec_params[((size_t)s * a.N + gct) * a.d + j] = cur[e];
        // archive append (proposal.py:794): per-chain archives see it at once, the shared one after the block
        if (!a.shared) arch_c[(size_t)(a.M_base + s) * DPAD + j] = cur[e];
        if (a.blk_states) a.blk_states[((size_t)s * a.NP + gct) * DPAD + j] = cur[e];
      }
    }
  }
  if (active) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      a.theta[gct * DPAD + q_ * EPT + e] = cur[e];
      a.theta_prev[gct * DPAD + q_ * EPT + e] = prev[e];
    }
  }
  if (wave == 0 && lane < 16) {
    a.lp[gcl] = lp;
    a.ll[gcl] = ll;
    a.acc_count[gcl] += nacc;
  }
}

struct DreamAdaptArgs {
  int64_t N, NP;
  int d, nCR, period;
  int boundary, do_scale;
  double gamma_pow;
  int shared;
  int64_t cap;
  int64_t row0, nrows;     // archive rows appended since the last catch-up
  int64_t M_total;         // archive size after them
  const double* arch;      // per chain [NP][cap][DPAD] / shared [cap][DPAD]
  double* zsum;            // [NP or 1][DPAD] column sums of the archive
  double* zsq;             // [NP or 1][DPAD] column sums of squares
  const double* partial;   // shared archive: per-chunk column sums [npart][2][DPAD] from k_colsum_partial (or null)
  int64_t npart;
  const double* theta;
  const double* theta_prev;
  const int32_t* mcr_last;
  double* pCR;             // [NP][MAX_NCR]
  double* LCR;             // [NP][MAX_NCR]
  double* DeltaCR;         // [NP][MAX_NCR]
  double* scaling;
  int32_t* acc_count;
};

// column sums / sums of squares of rows [row0 + 256 b, row0 + 256 (b+1)) of a row-major [.][DPAD] matrix
constexpr int COLSUM_CHUNK = 256;
template <int DPAD>
__global__ void __launch_bounds__(64) k_colsum_partial(const double* __restrict__ m, int64_t row0, int64_t nrows,
                                                       double* __restrict__ partial) {
  const int lane = threadIdx.x;
  if (lane >= DPAD) return;
  const int64_t b = blockIdx.x;
  const int64_t lo = b * COLSUM_CHUNK, hi = lo + COLSUM_CHUNK < nrows ? lo + COLSUM_CHUNK : nrows;
  double zs = 0.0, zq = 0.0;
#pragma unroll 8
  for (int64_t r = lo; r < hi; ++r) {
    const double z = m[(size_t)(row0 + r) * DPAD + lane];
    zs += z;
    zq += z * z;
  }
  partial[((size_t)b * 2 + 0) * DPAD + lane] = zs;
  partial[((size_t)b * 2 + 1) * DPAD + lane] = zq;
}

template <int DPAD>
__global__ void __launch_bounds__(64) k_dreamz_adapt(const DreamAdaptArgs a) {
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const bool lj = lane < a.d;
  const double* arch_c = a.shared ? a.arch : a.arch + (size_t)c * a.cap * DPAD;
  const size_t so = a.shared ? 0 : (size_t)c * DPAD;
  // archive column sums: per-chain archives are caught up by their own wave; for the shared archive the host
  // first runs a one-wave launch (N = 1, nrows > 0) and then the per-chain launch with nrows = 0.
  double zs = 0.0, zq = 0.0;
  if (lane < DPAD) {
    zs = a.zsum[so + lane];
    zq = a.zsq[so + lane];
    if (a.partial) {  // ordered accumulation of the chunk sums: deterministic for a given append
      for (int64_t b = 0; b < a.npart; ++b) {
        zs += a.partial[((size_t)b * 2 + 0) * DPAD + lane];
        zq += a.partial[((size_t)b * 2 + 1) * DPAD + lane];
      }
    } else {
      for (int64_t r = 0; r < a.nrows; ++r) {
        const double z = arch_c[(size_t)(a.row0 + r) * DPAD + lane];
        zs += z;
        zq += z * z;
      }
    }
    if (a.nrows > 0) {
      a.zsum[so + lane] = zs;
      a.zsq[so + lane] = zq;
    }
  }
  if (!a.boundary) return;
  if (a.do_scale) {
    if (lane == 0) {
      const double rate = (double)a.acc_count[c] / (double)a.period;
      a.scaling[c] = exp(log(a.scaling[c]) + a.gamma_pow * (rate - 0.24));
    }
    // crossover probabilities (proposal.py:797-809)
    const double Mt = (double)a.M_total;
    const double mean = zs / Mt;
    const double var = zq / Mt - mean * mean;  // np.var(Z, axis=0)
    const double jd = lj ? a.theta[c * DPAD + lane] - a.theta_prev[c * DPAD + lane] : 0.0;
    double term = lj ? jd * jd / var : 0.0;
    for (int off = 32; off >= 1; off >>= 1) term += __shfl_xor(term, off);
    if (lane == 0) {
      const int m = a.mcr_last[c];
      a.DeltaCR[c * MAX_NCR + m] += term;
      a.LCR[c * MAX_NCR + m] += 1.0;
      bool all = true;
      double tot = 0.0, mn[MAX_NCR];
      for (int k = 0; k < a.nCR; ++k) {
        all = all && a.LCR[c * MAX_NCR + k] > 0.0;
        mn[k] = a.DeltaCR[c * MAX_NCR + k] / (a.LCR[c * MAX_NCR + k] > 0.0 ? a.LCR[c * MAX_NCR + k] : 1.0);
        tot += mn[k];
      }
      if (all)
        for (int k = 0; k < a.nCR; ++k) a.pCR[c * MAX_NCR + k] = mn[k] / tot;
    }
  }
  if (lane == 0) a.acc_count[c] = 0;
}

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
