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
//                                reference's exact elementwise arithmetic (utils.py:113-122), global
//                                scaling adaptation, C <- Sigma swap + in-LDS Cholesky at period boundaries
//
// Chains never interact, so there is no inter-workgroup communication anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tda_philox.h"

namespace tda {

typedef double double4_t __attribute__((ext_vector_type(4)));

enum : int { MODE_STEP = 0, MODE_EVAL = 1 };
enum : int { PRIOR_DIAG = 0, PRIOR_DENSE = 1 };

// One level's linear-Gaussian posterior pieces, device pointers.
struct LevelDev {
  const double* Apk;   // packed MFMA fragments [ncb][KS/2][64 lanes][2]: A[cb*16+(l&15)][4*(2*k2+e)+(l>>4)]
  const double* ytil;  // [m_pad] data - b   (zero padded)
  const double* w;     // [m_pad] 1/diag(noise) for TDA_NOISE_DIAG, else nullptr
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
  double* am_sigma;          // [NP][DPAD][DPAD]
  double* Lk;                // [NP][DPAD][DPAD]
  double* scaling;           // [NP]
  int32_t* acc_count;        // [NP]
  int32_t* flags;            // [NP]
};

__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Sum over this wave's observation blocks of w_o (A theta' - ytil)_o^2 for the 16 chains of the tile.
// Lane l holds chain (l & 15) and rows (l >> 4) + 4 r of every 16-row block (f64 MFMA C/D layout).
// Two independent accumulators are interleaved and the next pair of fragments is prefetched from L2
// while the current pair is in the matrix pipe.
template <int DPAD>
__device__ __forceinline__ double level_sse_partial(const double* __restrict__ Apk, int ncb,
                                                    const double* __restrict__ s_y,
                                                    const double* __restrict__ s_w,
                                                    const double (&th)[DPAD / 4], int wave, int lane) {
  constexpr int KS = DPAD / 4, K2 = KS / 2;
  const int hi = lane >> 4;
  const double2* __restrict__ base = reinterpret_cast<const double2*>(Apk) + lane;
  double sse = 0.0;
  double2 f0[K2], f1[K2], g0[K2], g1[K2];
  int cb0 = wave, cb1 = wave + 4;
  const double2 zero2 = make_double2(0.0, 0.0);
#pragma unroll
  for (int k = 0; k < K2; ++k) {
    f0[k] = cb0 < ncb ? base[((size_t)cb0 * K2 + k) * 64] : zero2;
    f1[k] = cb1 < ncb ? base[((size_t)cb1 * K2 + k) * 64] : zero2;
  }
  while (cb0 < ncb) {
    const int nb0 = cb0 + 8, nb1 = cb1 + 8;
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      g0[k] = nb0 < ncb ? base[((size_t)nb0 * K2 + k) * 64] : zero2;
      g1[k] = nb1 < ncb ? base[((size_t)nb1 * K2 + k) * 64] : zero2;
    }
    double4_t a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      a0 = mfma_f64(f0[k].x, th[2 * k], a0);
      a1 = mfma_f64(f1[k].x, th[2 * k], a1);
      a0 = mfma_f64(f0[k].y, th[2 * k + 1], a0);
      a1 = mfma_f64(f1[k].y, th[2 * k + 1], a1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int o = cb0 * 16 + hi + 4 * r;
      const double res = a0[r] - s_y[o];
      double sq = res * res;
      if (s_w) sq *= s_w[o];
      sse += sq;
    }
    if (cb1 < ncb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = cb1 * 16 + hi + 4 * r;
        const double res = a1[r] - s_y[o];
        double sq = res * res;
        if (s_w) sq *= s_w[o];
        sse += sq;
      }
    }
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      f0[k] = g0[k];
      f1[k] = g1[k];
    }
    cb0 = nb0;
    cb1 = nb1;
  }
  return sse;
}

template <int DPAD>
__host__ __device__ constexpr int steps_lds_doubles(int m_pad, bool diag, int prior_rows) {
  return 16 * (DPAD + 2) + 64 + 64 + m_pad + (diag ? m_pad : 0) + prior_rows;
}

// ------------------------------------------------------------------------------------------------
// S fused Metropolis-Hastings steps for one tile of 16 chains  (Chain.sample, tinyDA/chain.py:95-125)
// ------------------------------------------------------------------------------------------------
template <int DPAD>
__global__ void __launch_bounds__(256, 1) k_mh_steps(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int KS = DPAD / 4;
  constexpr int LDP = DPAD + 2;  // row stride: conflict-free ds_read_b64 fragment gather
  constexpr int EPT = DPAD >= 16 ? DPAD / 16 : 1;
  constexpr int QACT = DPAD / EPT;

  const bool diag = a.lv.noise_kind == 1;
  const bool prior_dense = a.pr.kind == PRIOR_DENSE;
  double* s_prop = smem;
  double* s_red = s_prop + 16 * LDP;
  double* s_redp = s_red + 64;
  double* s_y = s_redp + 64;
  double* s_w = s_y + a.lv.m_pad;
  double* s_py = s_w + (diag ? a.lv.m_pad : 0);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = blockIdx.x;
  const int c = tid >> 4, q = tid & 15;  // thread-mapped (chain, element group)
  const int lc = lane & 15, hi = lane >> 4;  // lane-mapped chain / k sub-index
  const int64_t gct = tile * 16 + c;
  const int64_t gcl = tile * 16 + lc;
  const bool active = q < QACT;

  for (int i = tid; i < a.lv.m_pad; i += 256) {
    s_y[i] = a.lv.ytil[i];
    if (diag) s_w[i] = a.lv.w[i];
  }
  if (prior_dense)
    for (int i = tid; i < a.pr.ncb * 16; i += 256) s_py[i] = a.pr.wmu[i];

  // per-lane prior constants for parameters j = 4 kk + hi
  double pm[KS], pinv[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) {
    pm[kk] = a.pr.mean[4 * kk + hi];
    pinv[kk] = prior_dense ? 0.0 : a.pr.pinv[4 * kk + hi];
  }

  double cur[EPT], prp[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) cur[e] = active ? a.theta[gct * DPAD + q * EPT + e] : 0.0;
  double lp = a.lp[gcl], ll = a.ll[gcl];
  const double scal_t = a.scaling[gct];
  const double keep_t = a.prop_kind == 1 ? sqrt(1.0 - scal_t * scal_t) : 1.0;  // proposal.py:351-352
  int nacc = 0;
  const bool is_eval = a.mode == MODE_EVAL;
  const bool is_pcn = a.prop_kind == 1;
  const double* s_w_or_null = diag ? s_w : nullptr;
  __syncthreads();

  for (int s = 0; s < a.S; ++s) {
    // ---- proposal: theta' (proposal.py:249-251 / :351-355) ----
    if (active) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        if (is_eval) {
          prp[e] = cur[e];
        } else {
          const double x = a.inc[((size_t)s * a.NP + gct) * DPAD + q * EPT + e];
          const double sx = scal_t * x;
          prp[e] = is_pcn ? keep_t * cur[e] + sx : cur[e] + sx;
        }
        s_prop[c * LDP + q * EPT + e] = prp[e];
      }
    }
    __syncthreads();

    // ---- gather theta' into MFMA B-operand fragments ----
    double th[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) th[kk] = s_prop[lc * LDP + 4 * kk + hi];

    // ---- prior: scipy MVN logpdf (posterior.py:92) ----
    double maha = 0.0;
    if (!prior_dense) {
      double p = 0.0;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const double dv = th[kk] - pm[kk];
        p += dv * dv * pinv[kk];
      }
      p += __shfl_xor(p, 16);
      p += __shfl_xor(p, 32);
      maha = p;
    } else {
      double p = level_sse_partial<DPAD>(a.pr.Wpk, a.pr.ncb, s_py, nullptr, th, wave, lane);
      p += __shfl_xor(p, 16);
      p += __shfl_xor(p, 32);
      if (lane < 16) s_redp[wave * 16 + lane] = p;
    }

    // ---- forward model + Gaussian log-likelihood (posterior.py:95-108, distributions.py:310-326) ----
    double sse = level_sse_partial<DPAD>(a.lv.Apk, a.lv.ncb, s_y, s_w_or_null, th, wave, lane);
    sse += __shfl_xor(sse, 16);
    sse += __shfl_xor(sse, 32);
    if (lane < 16) s_red[wave * 16 + lane] = sse;
    __syncthreads();

    const double tot = ((s_red[lc] + s_red[16 + lc]) + s_red[32 + lc]) + s_red[48 + lc];
    if (prior_dense) maha = ((s_redp[lc] + s_redp[16 + lc]) + s_redp[32 + lc]) + s_redp[48 + lc];
    const double ll_n = diag ? -0.5 * tot : -0.5 * tot / a.lv.var;
    const double lp_n = -0.5 * (a.pr.logconst + maha);
    const double post_n = lp_n + ll_n;  // link.py:48

    // ---- Metropolis test (proposal.py:253-258, :357-362; chain.py:112) ----
    bool acc;
    if (is_eval) {
      acc = true;
    } else {
      double alpha = is_pcn ? exp(ll_n - ll) : exp(post_n - (lp + ll));
      if (post_n != post_n) alpha = 0.0;
      const double u = a.u[(size_t)s * a.NP + gcl];
      acc = u < alpha;
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
  }
}

// ------------------------------------------------------------------------------------------------
// Adaptation for a block: one wave per chain, lane j owns column j of Sigma in registers.
//   RecursiveSampleMoments.update (utils.py:113-124) for each recorded state, elementwise, unfused:
//     mu' = (1/(t+1)) (t mu + x)
//     Sigma' = (t-1)/t Sigma + sd/t ( t mu mu^T - (t+1) mu' mu'^T + x x^T + eps I )
//   global scaling (proposal.py:234-243), C <- Sigma (proposal.py:509-510) + Cholesky in LDS.
// This file is compiled with -ffp-contract=off so the products and sums round exactly like NumPy's.
// ------------------------------------------------------------------------------------------------
template <int DPAD>
__global__ void __launch_bounds__(64) k_adapt(const AdaptArgs a) {
  constexpr int LDM = DPAD + 1;
  __shared__ double s_vec[3 * DPAD];
  __shared__ double s_M[DPAD * LDM];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const bool lj = lane < a.d;

  double Sg[DPAD];
  if (a.do_am) {
    double mu = lane < DPAD ? a.am_mu[c * DPAD + lane] : 0.0;
#pragma unroll
    for (int i = 0; i < DPAD; ++i)
      Sg[i] = lane < DPAD ? a.am_sigma[((size_t)c * DPAD + i) * DPAD + lane] : 0.0;
    for (int s = 0; s < a.S; ++s) {
      const double x = lj ? a.rec_params[((size_t)s * a.N + c) * a.d + lane] : 0.0;
      const double t = (double)(a.t_base + s + 1);  // recursor.t before this update
      const double mup = (1.0 / (t + 1.0)) * (t * mu + x);
      const double ca = (t - 1.0) / t, cb = a.sd / t;
      if (lane < DPAD) {
        s_vec[lane] = x;
        s_vec[DPAD + lane] = mu;
        s_vec[2 * DPAD + lane] = mup;
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < DPAD; ++i) {
        const double xi = s_vec[i], mi = s_vec[DPAD + i], mpi = s_vec[2 * DPAD + i];
        double M = (t * (mi * mu) - (t + 1.0) * (mpi * mup)) + xi * x;
        if (i == lane) M = M + a.eps;
        Sg[i] = ca * Sg[i] + cb * M;
      }
      mu = mup;
      __syncthreads();
    }
    if (lane < DPAD) {
      a.am_mu[c * DPAD + lane] = mu;
#pragma unroll
      for (int i = 0; i < DPAD; ++i) a.am_sigma[((size_t)c * DPAD + i) * DPAD + lane] = Sg[i];
    }
  }

  if (!a.boundary) return;

  if (a.do_scale && lane == 0) {
    const double rate = (double)a.acc_count[c] / (double)a.period;  // np.mean(accepted[-period:])
    a.scaling[c] = exp(log(a.scaling[c]) + a.gamma_pow * (rate - 0.24));
  }
  if (lane == 0) a.acc_count[c] = 0;

  if (a.do_am && a.do_swap) {
    // C <- Sigma; factor in LDS, left-looking, lane i = row i.
#pragma unroll
    for (int i = 0; i < DPAD; ++i)
      if (lane < DPAD) s_M[i * LDM + lane] = Sg[i];
    __syncthreads();
    bool ok = true;
    for (int k = 0; k < a.d; ++k) {
      double sacc = 0.0;
      if (lane >= k && lj) {
        sacc = s_M[lane * LDM + k];
        for (int p = 0; p < k; ++p) sacc = fma(-s_M[lane * LDM + p], s_M[k * LDM + p], sacc);
      }
      const double dkk = __shfl(sacc, k);
      if (!(dkk > 0.0)) {
        ok = false;
        break;
      }
      const double lkk = sqrt(dkk);
      if (lane >= k && lj) s_M[lane * LDM + k] = (lane == k) ? lkk : sacc / lkk;
      __syncthreads();
    }
    if (ok) {
      if (lane < DPAD) {
        for (int k = 0; k < DPAD; ++k) {
          const double v = (lj && k < a.d && lane >= k) ? s_M[lane * LDM + k] : 0.0;
          a.Lk[((size_t)c * DPAD + k) * DPAD + lane] = v;
        }
      }
    } else if (lane == 0) {
      atomicOr(&a.flags[c], 1);
    }
  }
}

}  // namespace tda
