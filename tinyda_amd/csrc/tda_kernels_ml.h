// Multi-level tile kernel (Delayed Acceptance, MLDA) and the adaptive-error-model action kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tda_kernels_mh.h"
#include "tda_kernels_aemr.h"

namespace tda {

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
#ifndef TDA_MAXLEV
#define TDA_MAXLEV 6
#endif
constexpr int MAXLEV = TDA_MAXLEV;  // levels of a hierarchy (0.5: six; the generic level kernel with five / six levels spills 51 / 95 registers at 64 parameters)
constexpr int AEM_MAXLEV = 4;       // ... under an error model (dense: AEMR_MAXSUM trackers are summed; diagonal: per-level register arrays), with a dense
                                    // observation covariance, and above 64 parameters
constexpr int AEM_MP_MAX = 256;  // error-model output dimension limit; per-chain vectors / matrices in HBM have row stride 64, 128 or 256
constexpr int AEM_MP_MAX_EXT = 256;  // ... of hierarchies sequenced by the host (callback / source-defined levels, DREAM(Z) below a hierarchy): the same since k_ext_aem_*<256>
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
  int aem_on;              // 1: dense error model, 2: diagonal error model (level 0's likelihood under chain c's bias)
  int aem_mp;              // output dimension padded to 16 (dense: <= 128)
  int aem_ld;              // row stride of the per-chain error-model state: dense 64 or 128, diagonal m
  const double* aem_bias;  // [NP][aem_ld] (diagonal: [N][m])  total bias of level 0
  const double* aem_P;     // dense: [NP][tiles][4][64] lower tiles of V = L^-1, (Sigma_e + Sigma_bias) = L L^T, of level 0 (tda_kernels_aemr.h); diagonal: [N][m] inverse variances
  int64_t* sid;            // [nlev][NP] identity of the parameter vector each level currently holds
  const double* logu0;     // [S][NP] log of the base-level uniforms (k_propose / k_rng, off the critical path); may be null
};

__device__ __forceinline__ constexpr int pair_index(int j, int q) { return q * (q - 1) / 2 + j; }

// NW = waves sharing the tile: 4 (one per SIMD, up to 512 registers, pairs of observation blocks in flight) or 8 (two per
// SIMD with 256 registers each, the single-block pipeline of the 8-wave single-level tile; 17-30 % faster at every m,
// tools/waves_vs_m.py) where the level count leaves the registers for it.
// DENSE: the instance for hierarchies with a dense observation covariance on some level (a template parameter: as a run-time branch
// of `evaluate` it cost the other instances 39-95 more spilled registers at 64 parameters)
template <int DPAD, int NLEV, int NW = 4, bool DENSE = false>
__global__ void __launch_bounds__(64 * NW, NW / 4) k_ml_steps(const MLArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int TPC = 4 * NW;  // threads per chain in the thread-mapped phases
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int KS = DPAD / 4;
  constexpr int LDP = DPAD + 2;
  constexpr int EPT = DPAD >= TPC ? DPAD / TPC : 1;
  constexpr int QACT = DPAD / EPT;
  constexpr int NPAIR = NLEV * (NLEV - 1) / 2;

  const bool prior_dense = a.pr.kind == PRIOR_DENSE;
  // what the levels above the base hold -- states (thread-mapped), densities and flags (lane-mapped, one copy per wave) -- waits in
  // LDS between level actions: as registers across the base steps it is 60 of them at four levels, in a kernel that spills
  constexpr int UP_T = (NLEV - 1) * EPT;                        // doubles per thread
  constexpr int UP_L = 3 * (NLEV - 1) + 2 * NPAIR;              // doubles per lane-mapped chain and wave
  double* s_upt = smem;                                          // [UP_T][NT]
  double* s_upl = s_upt + UP_T * NT;                             // [NW][UP_L][16]
  double* s_prop = s_upl + NW * UP_L * 16;
  double* s_red = s_prop + 16 * LDP;
  double* s_redp = s_red + 16 * NW;
  double* s_stage = s_redp + 16 * NW;      // ytil / w of every level
  double* s_py = s_stage + a.lds_total;    // dense prior: W mu
  double* s_R = s_py + (prior_dense ? a.pr.ncb * 16 : 0);  // AEM: residual tile [16][aem_mp + 2], then [16] ll slots
  const int RSa = a.aem_mp + 2;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = blockIdx.x;
  const int c = tid / TPC, q_ = tid % TPC;
  const int lc = lane & 15, hi = lane >> 4;
  const int64_t gct = tile * 16 + c;
  const int64_t gcl = tile * 16 + lc;
  const bool active = q_ < QACT;
  const uint32_t gchain = (uint32_t)(a.chain_offset + gcl);

#pragma unroll
  for (int k = 0; k < NLEV; ++k) {
    if (k > 0 && !a.cascade) break;  // host-sequenced level actions: only the base level is evaluated here (and staged: the
                                     // residual tile of the error models needs the room)
    for (int i = tid; i < a.lv[k].m_pad; i += NT) {
      s_stage[a.lds_y[k] + i] = a.lv[k].ytil[i];
      if (a.lv[k].noise_kind == 1) s_stage[a.lds_w[k] + i] = a.lv[k].w[i];
    }
  }
  if (prior_dense)
    for (int i = tid; i < a.pr.ncb * 16; i += NT) s_py[i] = a.pr.wmu[i];

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

  auto stash_upper = [&]() {
    if constexpr (NLEV > 1) {
#pragma unroll
      for (int k = 1; k < NLEV; ++k)
#pragma unroll
        for (int e = 0; e < EPT; ++e) s_upt[((k - 1) * EPT + e) * NT + tid] = cur[k][e];
      double* __restrict__ ul = s_upl + wave * (UP_L * 16) + lc;
#pragma unroll
      for (int k = 1; k < NLEV; ++k) {
        ul[(3 * (k - 1) + 0) * 16] = lp[k];
        ul[(3 * (k - 1) + 1) * 16] = ll[k];
        ul[(3 * (k - 1) + 2) * 16] = (double)anyacc[k];
      }
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        ul[(3 * (NLEV - 1) + 2 * p) * 16] = Slp[p];
        ul[(3 * (NLEV - 1) + 2 * p + 1) * 16] = Sll[p];
      }
    }
  };
  auto fetch_upper = [&]() {
    if constexpr (NLEV > 1) {
#pragma unroll
      for (int k = 1; k < NLEV; ++k)
#pragma unroll
        for (int e = 0; e < EPT; ++e) cur[k][e] = s_upt[((k - 1) * EPT + e) * NT + tid];
      const double* __restrict__ ul = s_upl + wave * (UP_L * 16) + lc;
#pragma unroll
      for (int k = 1; k < NLEV; ++k) {
        lp[k] = ul[(3 * (k - 1) + 0) * 16];
        ll[k] = ul[(3 * (k - 1) + 1) * 16];
        anyacc[k] = (int)ul[(3 * (k - 1) + 2) * 16];
      }
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        Slp[p] = ul[(3 * (NLEV - 1) + 2 * p) * 16];
        Sll[p] = ul[(3 * (NLEV - 1) + 2 * p + 1) * 16];
      }
    }
  };
  stash_upper();
  int cnt[NLEV];
  int64_t stepno[NLEV];  // local step index (global, for RNG) of the NEXT step of level k
  int nrec[NLEV];        // records written by this launch per level
#pragma unroll
  for (int k = 0; k < NLEV; ++k) {
    cnt[k] = a.cnt[k];
    stepno[k] = a.done[k];
    nrec[k] = 0;
  }
  // slot of the next entry in the accept-flag ring, kept incrementally: the 64-bit remainder is ~150 instructions, and wave 0
  // alone would pay them in every step while seven waves wait at the barrier
  int ringidx = (int)(a.ring_pos % a.ring_P);
  const FragSrc fbase = frag_src(a.lv[0].Apk, lane);
  constexpr bool PAIRS = NW == 4 && DPAD <= 64;  // (128 parameters: two blocks of 16 fragments and the prior in registers would be 256 of them before any state)
  double2 f0[KS / 2], f1[PAIRS ? KS / 2 : 1];
  __syncthreads();

  // evaluate level `k` at the state currently in s_prop (all 4 waves); returns (lp_n, ll_n) lane-mapped
  auto evaluate = [&](int k, double2 (&g0)[KS / 2], double2 (&g1)[PAIRS ? KS / 2 : 1], double& lp_n, double& ll_n) {
    double th[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) th[kk] = s_prop[lc * LDP + 4 * kk + hi];
    double maha = 0.0;
    if (!prior_dense) {
      double p = 0.0;
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        const double dv = th[kk] - (PAIRS ? pm[kk] : a.pr.mean[4 * kk + hi]);
        p += dv * dv * (PAIRS ? pinv[kk] : a.pr.pinv[4 * kk + hi]);
      }
      if (a.pr.lo) {  // uniform components: zero density outside their support (rare path, bounds read through L1)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          if (th[kk] < a.pr.lo[4 * kk + hi] || th[kk] > a.pr.hi[4 * kk + hi]) p = INFINITY;
      }
      p = sum_rows(p);
      maha = p;
    } else {
      const FragSrc pbase = frag_src(a.pr.Wpk, lane);
      double p;
      if constexpr (PAIRS) {
        double2 p0[KS / 2], p1[KS / 2];
        frag_load<DPAD>(pbase, wave, a.pr.ncb, p0);
        frag_load<DPAD>(pbase, wave + 4, a.pr.ncb, p1);
        p = level_sse_partial<DPAD, 0>(a.pr.Wpk, a.pr.ncb, s_py, nullptr, th, wave, lane, p0, p1);
      } else {
        double2 p0[KS / 2];
        frag_load_buf<DPAD>(pbase, wave < a.pr.ncb ? wave : a.pr.ncb - 1, p0);
        p = level_sse_single<DPAD, 0, NW>(a.pr.Wpk, a.pr.ncb, s_py, nullptr, th, wave, lane, p0);
      }
      p = sum_rows(p);
      if (lane < 16) s_redp[wave * 16 + lane] = p;
    }
    const LevelDev& L = a.lv[k];
    const bool dg = L.noise_kind == 1;
    if (a.aem_on && k == 0) {
      // residual tile, then per chain  -1/2 (F + bias - y)^T P (F + bias - y)  with that chain's bias and P
      if constexpr (PAIRS) (void)level_sse_partial<DPAD, 2>(L.Apk, L.ncb, s_stage + a.lds_y[k], s_R + lc * RSa, th, wave, lane, g0, g1);
      else (void)level_sse_single<DPAD, 2, NW>(L.Apk, L.ncb, s_stage + a.lds_y[k], s_R + lc * RSa, th, wave, lane, g0);
      __syncthreads();
      const int MP = a.aem_mp, LD = a.aem_ld;
      if (a.aem_on == 2) {
        // diagonal error model (tda_kernels_aemd.h): -1/2 sum_o w_o (F_o - y_o + b_o)^2 with chain c's bias and inverse
        // variances, [N][m] each (LD = m); any m the residual tile holds
        for (int cc = wave; cc < 16; cc += NW) {
          const int64_t gc = tile * 16 + cc;
          const double* rrow = s_R + cc * RSa;
          double sacc = 0.0;
          if (gc < a.N) {
            // eight outputs per lane in flight (the loop is bound by the latency of the two loads per output, not by their volume)
            const double* __restrict__ bc = a.aem_bias + gc * LD;
            const double* __restrict__ wc = a.aem_P + gc * LD;
            int o = lane;
            for (; o + 7 * 64 < LD; o += 8 * 64) {
              double bv[8], wv[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                bv[u] = bc[o + 64 * u];
                wv[u] = wc[o + 64 * u];
              }
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const double r = rrow[o + 64 * u] + bv[u];
                sacc += wv[u] * (r * r);
              }
            }
            for (; o < LD; o += 64) {
              const double r = rrow[o] + bc[o];
              sacc += wc[o] * (r * r);
            }
          }
          for (int off = 32; off >= 1; off >>= 1) sacc += __shfl_xor(sacc, off);
          if (lane == 0) s_R[16 * RSa + cc] = -0.5 * sacc;
        }
        __syncthreads();
        ll_n = s_R[16 * RSa + lc];
        if (prior_dense) {
          maha = s_redp[lc];
#pragma unroll
          for (int w = 1; w < NW; ++w) maha += s_redp[w * 16 + lc];
        }
        lp_n = -0.5 * (a.pr.logconst + maha);
        return;
      }
      for (int cc = wave; cc < 16; cc += NW) {  // lane = observation (and observation + 64, + 128, + 192 beyond 64 outputs)
        const int64_t gc = tile * 16 + cc;
        double* rrow = s_R + cc * RSa;
        for (int o = lane; o < MP; o += 64) rrow[o] += a.aem_bias[gc * LD + o];  // (a lane's own entries: no exchange until the solve)
        __builtin_amdgcn_wave_barrier();
        // -1/2 |V r|^2 from the chain's lower tiles of V = L^-1 (72 KB per chain at 128 outputs, 512-byte rows; the rows of
        // blocks beyond the outputs are identity and r is not defined there: only the block rows of the outputs are read)
        // (one instance for every row stride: the substitution's loops are run-time loops over the block rows of the outputs)
        const double qv = aem_quad_factor_inplace<8>(a.aem_P + (size_t)gc * aemr_v_doubles(LD), rrow, lane, MP >> 4);
        if (lane == 0) s_R[16 * RSa + cc] = qv;
      }
      if (prior_dense && lane < 16) {}  // (s_redp already written above)
      __syncthreads();
      ll_n = s_R[16 * RSa + lc];
      if (prior_dense) {
        maha = s_redp[lc];
#pragma unroll
        for (int w = 1; w < NW; ++w) maha += s_redp[w * 16 + lc];
      }
      lp_n = -0.5 * (a.pr.logconst + maha);
      return;
    }
    if (DENSE && L.noise_kind == 2) {
      // dense observation covariance (DefaultGaussianLogLike, distributions.py:246-301) at any level of the hierarchy (round 4):
      // residual tile to LDS, then r^T Sigma^-1 r for the tile's 16 chains on the matrix cores, as the single-level kernel does
      const int RSd = L.m_pad + 2;
      if constexpr (PAIRS) (void)level_sse_partial<DPAD, 2>(L.Apk, L.ncb, s_stage + a.lds_y[k], s_R + lc * RSd, th, wave, lane, g0, g1);
      else (void)level_sse_single<DPAD, 2, NW>(L.Apk, L.ncb, s_stage + a.lds_y[k], s_R + lc * RSd, th, wave, lane, g0);
      __syncthreads();
      double qs = dense_quadform<NW>(L.Ppk, L.ncb, L.m_pad, s_R, RSd, wave, lane);
      qs = sum_rows(qs);
      if (lane < 16) s_red[wave * 16 + lane] = qs;
      __syncthreads();
      double tot = s_red[lc];
#pragma unroll
      for (int w = 1; w < NW; ++w) tot += s_red[w * 16 + lc];
      if (prior_dense) {
        maha = s_redp[lc];
#pragma unroll
        for (int w = 1; w < NW; ++w) maha += s_redp[w * 16 + lc];
      }
      ll_n = -0.5 * tot;
      lp_n = -0.5 * (a.pr.logconst + maha);
      __syncthreads();  // s_red / the residual tile are free again
      return;
    }
    double sse;
    if constexpr (PAIRS) {
      sse = dg ? level_sse_partial<DPAD, 1>(L.Apk, L.ncb, s_stage + a.lds_y[k], s_stage + a.lds_w[k], th, wave, lane, g0, g1)
               : level_sse_partial<DPAD, 0>(L.Apk, L.ncb, s_stage + a.lds_y[k], nullptr, th, wave, lane, g0, g1);
    } else {
      sse = dg ? level_sse_single<DPAD, 1, NW>(L.Apk, L.ncb, s_stage + a.lds_y[k], s_stage + a.lds_w[k], th, wave, lane, g0)
               : level_sse_single<DPAD, 0, NW>(L.Apk, L.ncb, s_stage + a.lds_y[k], nullptr, th, wave, lane, g0);
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
    ll_n = dg ? -0.5 * tot : -0.5 * tot / L.var;
    lp_n = -0.5 * (a.pr.logconst + maha);
  };

  for (int s = 0; s < a.S; ++s) {
    // ================= level 0: one Metropolis-Hastings step =================
    if constexpr (PAIRS) {
      frag_load<DPAD>(fbase, wave, a.lv[0].ncb, f0);
      frag_load<DPAD>(fbase, wave + 4, a.lv[0].ncb, f1);
    } else {
      frag_load_buf<DPAD>(fbase, wave < a.lv[0].ncb ? wave : a.lv[0].ncb - 1, f0);
    }
    if (a.randomize && cnt[0] == 0) {  // DA: draw the promoted index of the subchain that starts now
      const int L0 = a.sl[0];
      if (a.ridx_rep) {
        const double r = a.ridx_rep[(size_t)(stepno[NLEV > 1 ? 1 : 0] - a.done[1]) * a.N + (gcl < a.N ? gcl : 0)];  // (NLEV = 1 is never launched with randomised subchains)
        pick = (r != r) ? L0 - 1 : (int)r + L0;  // reference index in [-L, -1] (chain.py:525-527)
      } else {
        const u32x4 r = philox4x32_10(u32x4{0u, (uint32_t)stepno[NLEV > 1 ? 1 : 0], gchain, STREAM_INDEX}, (uint32_t)a.seed,
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
      a.ring[(size_t)ringidx * a.NP + gcl] = acc0 ? 1 : 0;
    }
    ringidx = ringidx + 1 == a.ring_P ? 0 : ringidx + 1;
    nrec[0] += 1;
    stepno[0] += 1;
    cnt[0] += 1;

    // ================= upper levels whose subchain just completed =================
    if (NLEV > 1 && a.cascade && cnt[0] == a.sl[0]) {
    fetch_upper();
#pragma unroll
    for (int k = 0; k < NLEV - 1; ++k) {
      if (cnt[k] != a.sl[k]) break;
      const int q = k + 1;
      const bool use_snap = (a.randomize != 0) && k == 0;
      // y -> LDS for the fragment gather
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) s_prop[c * LDP + q_ * EPT + e] = use_snap ? snp[e] : cur[k][e];
      }
      const FragSrc gb = frag_src(a.lv[q].Apk, lane);
      double2 g0[KS / 2], g1[PAIRS ? KS / 2 : 1];
      if constexpr (PAIRS) {
        frag_load<DPAD>(gb, wave, a.lv[q].ncb, g0);
        frag_load<DPAD>(gb, wave + 4, a.lv[q].ncb, g1);
      } else {
        frag_load_buf<DPAD>(gb, wave < a.lv[q].ncb ? wave : a.lv[q].ncb - 1, g0);
      }
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
        a.ring[(size_t)ringidx * a.NP + gcl] = accq ? 1 : 0;
      }
      ringidx = ringidx + 1 == a.ring_P ? 0 : ringidx + 1;
      nrec[q] += 1;
      stepno[q] += 1;
      cnt[k] = 0;
      cnt[q] += 1;
    }
    stash_upper();
    }
  }

  // ---- write the state back ----
  fetch_upper();
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
// Delayed Acceptance, two levels, on the 8-wave tile (DAChain.sample, chain.py:342-444, acceptance :475-483), for the
// shape that matters for throughput (BASELINE config 3): a SMALL coarse model (m0 <= 256) stepped many times per fine
// evaluation, fixed subchain length, iso / diag noise, diagonal prior.  At m0 = 256 a coarse step of the generic kernel
// is 4 100 cycles of matrix work inside a 9 000-cycle step (tools/steps_microbench.hip): proposal -> LDS -> barrier ->
// fragment gather -> prior -> MFMA -> reduction -> barrier -> decision is one dependent chain.  This kernel breaks it:
//   * the coarse forward model is linear, A theta' = keep A theta + A (s inc): every wave keeps the model output F of
//     its <= 2 observation blocks for the current state in registers, and a step only multiplies the INCREMENT, which does
//     not depend on the previous decision -- the MFMAs of step s + 1 are issued BEFORE the barrier of step s, the matrix
//     pipe works while step s is reduced and decided;
//   * the coarse operator itself is register resident (2 blocks x 32 registers per wave; nothing is streamed per step);
//   * increments are staged two steps ahead into LDS in fragment order by the threads that also keep the chain state;
//     the prior of theta' is a thread-mapped partial sum reduced in the shadow of the MFMAs;
//   * F is re-derived from theta by a direct product at every launch, so rounding cannot accumulate beyond one block; the
//     outputs at the states the upper levels hold wait in LDS and return with the state when a level step is rejected
//     (re-deriving them after every level action was 7 000 cycles of barriers, LDS round trips and a cold MFMA chain);
//   * the fine level is evaluated directly (streamed fragments, two waves per SIMD) every sl[0] coarse steps: skip rule,
//     two-stage acceptance with the densities kept from the subchain start, alignment, records, accept-flag window.
// Same MLArgs, records and RNG contract as k_ml_steps<DPAD, 2>; log-densities agree with it to rounding (the linear update
// and the summation order of the prior differ in the last bits), decisions are the same.
// ------------------------------------------------------------------------------------------------
template <int DPAD>
__host__ __device__ constexpr int da_lds_doubles(int stage_total) {
  return 16 * (DPAD + 2) + 2 * 64 * (DPAD / 4 + 2) + 2 * 16 * 8 + 2 * 16 + 4 * DPAD + 2 * 16 + 8 * 2 * 2 * 16 + 6 * 16 + 64 * (DPAD / 4 + 2) + 16 * 512 + stage_total;
}

#ifdef TDA_DA_TRACE
__device__ long long g_da_trace[128 * 8 * 8];  // debug builds only: [step][wave][stamp] cycle stamps of tile 0 (tools/da_trace.py)
#define DA_STAMP(i) \
  if (blockIdx.x == 0 && lane == 0 && s < 128) g_da_trace[((size_t)s * 8 + wave) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define DA_STAMP(i)
#endif

// RB = 16-row blocks of the coarse operator per wave (1: m0 <= 128, 2: m0 <= 256); PCN: CrankNicolson proposals; NZ0: noise of the
// coarse level: 0 isotropic, 1 diagonal, 2 the diagonal error model (per-chain bias and inverse variances, MLArgs::aem_on == 2: the
// launch is one base subchain, the host sequences the level actions, MLArgs::cascade == 0)
// on the coarse level -- template parameters, because as run-time flags they cost a select per model output and step in the
// vector section that decides when the SIMD's other wave may start its burst
// The kernel proper (tda_kernels_da_body.inc), twice: as the compiler allocates it (up to 256 registers per wave, two waves per SIMD:
// the tile owns the CU), and held to 224 registers so that one wave of the generator (64 registers) is co-resident on every SIMD and
// the next block's draws run under this block's steps (run_multilevel).  amdgpu_num_vgpr takes a literal, hence two entry points
// and not a template argument -- and a textual include and not a __device__ function: with the arguments behind a reference the
// register allocation of every instance changes (C3's from 227 to 238).  On gfx90a and later the backend doubles the number
// (112 -> a budget of 224).
template <int DPAD, int RB, bool PCN, int NZ0, int NLEV = 2>
__global__ void __launch_bounds__(512, 2) k_da_steps(const MLArgs a) {
#include "tda_kernels_da_body.inc"
}
template <int DPAD, int RB, bool PCN, int NZ0, int NLEV = 2>
__global__ void __launch_bounds__(512, 2) __attribute__((amdgpu_num_vgpr(112))) k_da_steps_r224(const MLArgs a) {
#include "tda_kernels_da_body.inc"
}

// ------------------------------------------------------------------------------------------------
// Model outputs of a LINEAR level inside a host-sequenced hierarchy (callback / source-defined levels beside it, DREAMZ at the
// base, the diagonal error model): F[N][m] = prop[N][d] A^T + b for all chains on the matrix cores.  One workgroup of four
// waves per 16-chain tile; the chains' parameters are gathered once into MFMA B fragments, the observation blocks are dealt
// over the waves, A comes as the packed fragments the fused kernels use (two register sets, next block in flight), so every
// fragment read serves 16 chains.  (The first version -- one wave per chain, every wave streaming all of A -- moved
// N m d 8 bytes through L2 per launch: 256 MB at 4096 chains, m = 128, 2 GB at m = 1024.)
// ------------------------------------------------------------------------------------------------
template <int DPAD>
__device__ __forceinline__ void linear_outputs_tile(long long N, int d, int m, const double* __restrict__ Apk, int ncb,
                                                    const double* __restrict__ bvec, const double* __restrict__ prop, int ldp,
                                                    double* __restrict__ F, int ldf, long long tile) {  // prop: rows of ldp doubles ([N][d] or a state array); bvec may be null

  constexpr int KS = DPAD / 4, K2 = DPAD / 8, NWV = 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lc = lane & 15, hi = lane >> 4;
  const long long c = tile * 16 + lc;
  double th[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) th[kk] = (c < N && 4 * kk + hi < d) ? prop[c * ldp + 4 * kk + hi] : 0.0;
  const FragSrc src = frag_src(Apk, lane);
  double2 fa[K2], fb[K2];
  if (wave < ncb) frag_load_buf<DPAD>(src, wave, fa);
  for (int cb = wave; cb < ncb; cb += 2 * NWV) {
    const bool more = cb + NWV < ncb;
    frag_load_buf<DPAD>(src, more ? cb + NWV : cb, fb);
    __builtin_amdgcn_sched_barrier(0);
    {
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < K2; ++k) {
        acc = mfma_f64(fa[k].x, th[2 * k], acc);
        acc = mfma_f64(fa[k].y, th[2 * k + 1], acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = cb * 16 + hi + 4 * r;
        if (c < N && o < m) F[c * ldf + o] = acc[r] + (bvec ? bvec[o] : 0.0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!more) break;
    frag_load_buf<DPAD>(src, cb + 2 * NWV < ncb ? cb + 2 * NWV : cb, fa);
    __builtin_amdgcn_sched_barrier(0);
    {
      double4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < K2; ++k) {
        acc = mfma_f64(fb[k].x, th[2 * k], acc);
        acc = mfma_f64(fb[k].y, th[2 * k + 1], acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = (cb + NWV) * 16 + hi + 4 * r;
        if (c < N && o < m) F[c * ldf + o] = acc[r] + (bvec ? bvec[o] : 0.0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int DPAD>
__global__ void __launch_bounds__(256) k_linear_outputs(long long N, int d, int m, const double* __restrict__ Apk, int ncb,
                                                        const double* __restrict__ bvec, const double* __restrict__ prop, int ldp,
                                                        double* __restrict__ F) {
  linear_outputs_tile<DPAD>(N, d, m, Apk, ncb, bvec, prop, ldp, F, m, (long long)blockIdx.x);
}
// several products in one launch (blockIdx.y picks the item): the model outputs k_aem_action needs -- levels q and q - 1 at the
// states of levels q - 1 and q -- for every chain on the matrix cores, an operator fragment serving a 16-chain tile, instead of
// three matrix-vector products per chain inside that kernel (each chain read the whole operator from L2: 0.8 GB per launch at
// 4096 chains x 128 outputs, the larger part of the kernel's time)
struct LinMultiArgs {
  long long N;
  int d, m, ldp, ldf, n;
  const double* Apk[4];
  int ncb[4];
  const double* prop[4];
  double* F[4];
};
template <int DPAD>
__global__ void __launch_bounds__(256) k_linear_outputs_multi(const LinMultiArgs a) {
  const int w = blockIdx.y;
  linear_outputs_tile<DPAD>(a.N, a.d, a.m, a.Apk[w], a.ncb[w], nullptr, a.prop[w], a.ldp, a.F[w], a.ldf, (long long)blockIdx.x);
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
  const double* Fpre[4];       // [NP][MP] each, or null: A_q theta_{q-1}, A_q theta_q, A_{q-1} theta_{q-1}, A_{q-1} theta_q (k_linear_outputs_multi)
  const double* A[MAXLEV];     // row-major [m][d]
  const double* ytil[MAXLEV];  // y - b, [MP]   (residual r = A theta - ytil = F - y)
  const double* data[MAXLEV];  // y, [MP]       (model output F = r + y)
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
  double* cov_inv[MAXLEV];   // [NP][tiles][4][64] lower tiles of V = L^-1 (tda_kernels_aemr.h)
  double* b_mu[MAXLEV];      // trackers of levels >= 1: [NP][MP]
  double* mdiff[MAXLEV];     // [NP][MP]
  int64_t b_t;               // recursion counter of level q's tracker before this update
  double* rvec;              // [NP][MP] out: bias-corrected residual of level q-1's latest link (k_aem_refresh ends with its update_link)
  double* upd;               // [NP][3][MP] out: vectors of the tracker's covariance update (dm, mu, mu'; state-dependent: x), applied by k_aem_refresh
  const double* scaling;     // [NP] (pCN beta for the state-dependent q terms)
  const double* u_rep;       // [N] replay uniform of this step (NaN = none drawn) or null
  uint8_t* ring;
  int ring_P;
  int64_t ring_pos;
  double* rec_params;  // row of this step, [N][d] (may be null)
  double* rec_stats;
  uint8_t* rec_acc;
};

template <int MPT>
__global__ void __launch_bounds__(MPT) k_aem_action(const AemArgs a) {
  constexpr int NW = MPT / 64;  // thread = observation: one wave up to 64 outputs, two up to 128, four up to 256
  static_assert(NW == 1 || NW == 2 || NW == 4, "k_aem_action: 64, 128 or 256 outputs");
  __shared__ double s_v[4 * MPT];
  __shared__ double s_x[8];  // exchange slots between the waves
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const int q = a.q, k = a.q - 1, nl = a.nlev, MP = a.MP, d = a.d;
  const bool lo = lane < a.m, lj = lane < d;
  auto TH = [&](int lev) { return a.theta + ((size_t)lev * a.NP + c) * a.DP; };
  auto bsum = [&](double v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if constexpr (NW > 1) {
      __syncthreads();
      if ((lane & 63) == 0) s_x[lane >> 6] = v;
      __syncthreads();
      v = s_x[0] + s_x[1];
      if constexpr (NW == 4) v += s_x[2] + s_x[3];
    }
    return v;
  };
  // F_lev(theta)[lane] - ytil_lev[lane]  (theta given through LDS vector s_v[0..d))
  const bool pre = a.Fpre[0] != nullptr;
  auto resid_pre = [&](int which, int lev) {  // the same from the products of k_linear_outputs_multi
    return lo ? a.Fpre[which][c * MP + lane] - a.ytil[lev][lane] : 0.0;
  };
  auto resid = [&](int lev) {
    double f = 0.0;
    if (lo) {
      const double* __restrict__ Ac = a.A[lev] + lane;  // column-major [d][MPT]
#pragma unroll 8
      for (int j = 0; j < d; ++j) f = fma(Ac[(size_t)j * MPT], s_v[j], f);
      f -= a.ytil[lev][lane];
    }
    return f;
  };
  // -1/2 r^T (Sigma_e + Sigma_bias)^-1 r = -1/2 |L^-1 r|^2 from chain c's factor of adaptive level lev (tda_kernels_aemr.h: factor
  // form, 16 x 16 tiles in the MFMA C/D layout) by blocked forward substitution; r given per lane (already bias corrected).
  // (the substitution is one wave's dependent chain: wave 0 runs it, a second wave of a 128-output workgroup contributes zero; the first
  // tiles of the acting level's factor are requested HERE, ahead of the decision's own loads)
  AemQuadStream<MPT / 16> qs;
  const bool quad_q = a.q < a.nlev - 1, wave0 = (lane >> 6) == 0;
  if (quad_q && wave0) aem_quad_request(qs, a.cov_inv[a.q] + (size_t)c * aemr_v_doubles(MPT), lane & 63);
  auto quad = [&](int lev, double r, bool requested) {
    __syncthreads();
    s_v[MPT + lane] = lo ? r : 0.0;
    __syncthreads();
    double s = 0.0;
    if (wave0) {
      if (!requested) aem_quad_request(qs, a.cov_inv[lev] + (size_t)c * aemr_v_doubles(MPT), lane & 63);
      s = sum_rows(aem_quad_factor_stream(qs, s_v + MPT, lane & 63));
    }
    if constexpr (NW > 1) {
      __syncthreads();
      if ((lane & 63) == 0) s_x[lane >> 6] = s;
      __syncthreads();
      s = s_x[0] + s_x[1];
      if constexpr (NW == 4) s += s_x[2] + s_x[3];
    }
    return -0.5 * s;
  };
  auto loglike_of = [&](int lev, double r0) {  // r0 = F - ytil without bias
    if (lev == nl - 1) return -0.5 * bsum(lo ? r0 * r0 : 0.0) / a.var_finest;
    return quad(lev, lo ? r0 + a.bias_tot[lev][c * MP + lane] : 0.0, lev == a.q);
  };

  // ---------------- the level-q decision ----------------
  const double yj = lj ? TH(k)[lane] : 0.0, xj = lj ? TH(q)[lane] : 0.0;
  const double y_lp = a.lp[(size_t)k * a.NP + c], y_ll = a.ll[(size_t)k * a.NP + c];
  const double x_lp = a.lp[(size_t)q * a.NP + c], x_ll = a.ll[(size_t)q * a.NP + c];
  const int pkq = pair_index(k, q);
  const double st_lp = a.Sst[((size_t)pkq * 2 + 0) * a.NP + c], st_ll = a.Sst[((size_t)pkq * 2 + 1) * a.NP + c];
  const bool any = a.anyacc[(size_t)k * a.NP + c] != 0;
  __syncthreads();
  s_v[lane] = yj;
  __syncthreads();
  const double rq_y = pre ? resid_pre(0, q) : resid(q);                          // F_q(y) - ytil_q
  const double rk_y = a.dependent ? (pre ? resid_pre(2, k) : resid(k)) : 0.0;  // F_k(y) - ytil_k
  const double lpn = y_lp;  // same prior, same parameters (posterior.py:92)
  const double lln = loglike_of(q, rq_y);
  double alpha;
  if (a.dependent) {  // chain.py:446-473
    // bias at the proposal and the coarse density of the subchain start under it
    const double bias_next = (rq_y + a.data[q][lane < MP ? lane : 0]) - (rk_y + a.data[k][lane < MP ? lane : 0]);
    __syncthreads();
    s_v[lane] = xj;  // subchain start = the fine state
    __syncthreads();
    const double rk_x = pre ? resid_pre(3, k) : resid(k);
    const double ll_b = quad(k, lo ? rk_x + bias_next : 0.0, false);
    double q_xy = 0.0, q_yx = 0.0;
    if (a.prop_kind == 1) {  // pCN transition densities (proposal.py:364-369) between the fine links
      const double beta = a.scaling[c], kp = sqrt(1.0 - beta * beta);
      for (int dir = 0; dir < 2; ++dir) {
        __syncthreads();
        s_v[2 * MPT + lane] = dir == 0 ? yj - kp * xj : xj - kp * yj;
        __syncthreads();
        double w = 0.0;
        if (lj) {
          const double* Wr = a.pr_W + (size_t)lane * d;
          for (int j = 0; j <= lane; ++j) w = fma(Wr[j], s_v[2 * MPT + j], w);
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
  s_v[lane] = cj;
  __syncthreads();
  // (the state is y = theta_k after an acceptance, x = theta_q after a rejection: both were multiplied out ahead of the decision)
  const double rq = pre ? (acc ? rq_y : resid_pre(1, q)) : resid(q);
  const double rk = pre ? resid_pre(acc ? 2 : 3, k) : resid(k);
  const double diff_new = lo ? (rq + a.data[q][lane]) - (rk + a.data[k][lane]) : 0.0;
  double* md = a.mdiff[q] + c * MP;
  const double t = (double)a.b_t;
  // The covariance of the tracker (utils.py:117-122 / :199) is updated by k_aem_refresh, which reads every tile of it anyway
  // (round 4: the read-modify-write of the full m x m matrix here was 1 GB per launch at 4096 chains x 128 outputs, and the
  // refresh kernel read the result straight back): this kernel leaves the vectors of the update, a.upd[c][3][MP].
  double* up = a.upd + (size_t)c * 3 * MP;
  if (a.dependent) {
    const double xupd = lo ? (rq + a.data[q][lane]) - ((rk + a.data[k][lane]) + md[lane]) : 0.0;  // chain.py:505-507
    if (lo) md[lane] = diff_new;
    up[lane] = xupd;  // Sigma <- (t-1)/t Sigma + 1/t x x^T: the general update with mu = mu' = 0
    up[MP + lane] = 0.0;
    up[2 * MP + lane] = 0.0;
  } else {
    const double dm = (a.is_da || acc) ? diff_new : (lo ? md[lane] : 0.0);  // MLDA refreshes the difference on accept only
    if (lo) md[lane] = dm;
    double* mu = a.b_mu[q] + c * MP;
    const double mu_o = lo ? mu[lane] : 0.0;
    const double mu_n = (1.0 / (t + 1.0)) * (t * mu_o + dm);  // utils.py:113-122 with sd = 1, eps = 0
    up[lane] = lo ? dm : 0.0;
    up[MP + lane] = mu_o;
    up[2 * MP + lane] = lo ? mu_n : 0.0;
    if (lo) mu[lane] = mu_n;
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
  // update_link of level k's latest link (posterior.py:112-134) happens at the end of k_aem_refresh, under the factor it has just
  // computed: this kernel leaves the bias-corrected residual F_k(theta_k) - y_k + bias_k (zero in the padding)
  a.rvec[c * MP + lane] = lo ? rk + bt : 0.0;
}

// ------------------------------------------------------------------------------------------------
// The state-independent error model for hierarchies whose models live outside the engine's kernels (batched host
// callbacks, source-defined models, linear levels beside them).  Same algorithm as k_aem_action, but a model output is
// never recomputed from a matrix: every level keeps the output of its current link (Fcur), every pair (j, q) the output
// of level j at theta_q (Fst, the companion of the densities in Sst), and the host hands in level q's fresh evaluation
// at theta_{q-1} (Fnew): decision, alignment, tracker update, total bias, output bookkeeping; then k_aem_refresh (the new factor
// and update_link of level q - 1).  k_ext_aem_accept is the base-level step under the bias-corrected likelihood.
// ------------------------------------------------------------------------------------------------
struct ExtAemArgs {
  int64_t N, NP, chain_offset;
  int d, DP, m, MP, nlev, q, is_da;
  int dependent, prop_kind;    // state-dependent error model (DA only, chain.py:446-473, :501-523); pCN needs the q terms
  const double* pr_W;          // [d][d] whitening matrix of the prior, row-major
  double pr_logdet;
  const double* scaling;       // [NP] pCN beta
  uint64_t seed;
  int64_t step;
  const double* Fnew;          // [N][m] level q at theta_{q-1}
  const double* data[MAXLEV];  // [MP]
  double var_finest;
  double* Fcur[MAXLEV];        // [NP][MP]
  double* Fst;                 // [npairs][NP][MP]
  double* theta;               // [nlev][NP][DP]
  double* lp;
  double* ll;
  double* Sst;
  int32_t* anyacc;
  int64_t* sid;
  double* bias_tot[MAXLEV];
  double* cov_inv[MAXLEV];     // [NP][tiles][4][64] lower tiles of V = L^-1 (tda_kernels_aemr.h)
  double* b_mu[MAXLEV];
  double* mdiff[MAXLEV];
  int64_t b_t;
  double* rvec;                // [NP][MP] out: bias-corrected residual of level q-1's latest link
  double* upd;                 // [NP][3][MP] out: vectors of the tracker's covariance update (k_aem_action)
  const double* u_rep;
  uint8_t* ring;
  int ring_P;
  int64_t ring_pos;
  double* rec_params;
  double* rec_stats;
  uint8_t* rec_acc;
};

template <int MPT>
__global__ void __launch_bounds__(MPT) k_ext_aem_action(const ExtAemArgs a) {
  constexpr int NW = MPT / 64;
  __shared__ double s_v[4 * MPT];
  __shared__ double s_x[8];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const int q = a.q, k = a.q - 1, nl = a.nlev, MP = a.MP, d = a.d;
  const bool lo = lane < a.m, lj = lane < d;
  auto TH = [&](int lev) { return a.theta + ((size_t)lev * a.NP + c) * a.DP; };
  auto FS = [&](int j, int qq) { return a.Fst + ((size_t)pair_index(j, qq) * a.NP + c) * MP; };
  auto bsum = [&](double v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if constexpr (NW > 1) {
      __syncthreads();
      if ((lane & 63) == 0) s_x[lane >> 6] = v;
      __syncthreads();
      v = s_x[0] + s_x[1];
      if constexpr (NW == 4) v += s_x[2] + s_x[3];
    }
    return v;
  };
  auto quad = [&](int lev, double r) {  // -1/2 |V r|^2 with chain c's factor V = L^-1 of adaptive level lev (as in k_aem_action)
    __syncthreads();
    s_v[MPT + lane] = lo ? r : 0.0;
    __syncthreads();
    const double* __restrict__ Vc = a.cov_inv[lev] + (size_t)c * aemr_v_doubles(MPT);
    // (the substitution is one wave's dependent chain: wave 0 runs it, a second wave of a 128-output workgroup contributes zero)
    double s = (lane >> 6) == 0 ? sum_rows(aem_quad_factor<MPT / 16>(Vc, s_v + MPT, lane & 63)) : 0.0;
    if constexpr (NW > 1) {
      __syncthreads();
      if ((lane & 63) == 0) s_x[lane >> 6] = s;
      __syncthreads();
      s = s_x[0] + s_x[1];
      if constexpr (NW == 4) s += s_x[2] + s_x[3];
    }
    return -0.5 * s;
  };
  auto loglike_of = [&](int lev, double r0) {  // r0 = F - data without bias
    if (lev == nl - 1) return -0.5 * bsum(lo ? r0 * r0 : 0.0) / a.var_finest;
    return quad(lev, lo ? r0 + a.bias_tot[lev][c * MP + lane] : 0.0);
  };

  // ---------------- the level-q decision (chain.py:475-483, proposal.py:1615-1624) ----------------
  const double yj = lj ? TH(k)[lane] : 0.0, xj = lj ? TH(q)[lane] : 0.0;
  const double y_lp = a.lp[(size_t)k * a.NP + c], y_ll = a.ll[(size_t)k * a.NP + c];
  const double x_lp = a.lp[(size_t)q * a.NP + c], x_ll = a.ll[(size_t)q * a.NP + c];
  const int pkq = pair_index(k, q);
  const double st_lp = a.Sst[((size_t)pkq * 2 + 0) * a.NP + c], st_ll = a.Sst[((size_t)pkq * 2 + 1) * a.NP + c];
  const bool any = a.anyacc[(size_t)k * a.NP + c] != 0;
  const double fnew = lo ? a.Fnew[(size_t)c * a.m + lane] : 0.0;
  const double lpn = y_lp;
  const double lln = loglike_of(q, lo ? fnew - a.data[q][lane] : 0.0);
  double alpha;
  if (a.dependent) {  // chain.py:446-473: the bias at the proposal, the coarse density of the subchain start under it
    const double bias_next = lo ? fnew - a.Fcur[k][c * MP + lane] : 0.0;
    const double rk_x = lo ? FS(k, q)[lane] - a.data[k][lane] : 0.0;
    const double ll_b = quad(k, lo ? rk_x + bias_next : 0.0);
    double q_xy = 0.0, q_yx = 0.0;
    if (a.prop_kind == 1) {  // pCN transition densities (proposal.py:364-369) between the fine links
      const double beta = a.scaling[c], kp = sqrt(1.0 - beta * beta);
      for (int dir = 0; dir < 2; ++dir) {
        __syncthreads();
        s_v[2 * MPT + lane] = dir == 0 ? yj - kp * xj : xj - kp * yj;
        __syncthreads();
        double w = 0.0;
        if (lj) {
          const double* Wr = a.pr_W + (size_t)lane * d;
          for (int j = 0; j <= lane; ++j) w = fma(Wr[j], s_v[2 * MPT + j], w);
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
    alpha = exp(((lpn + lln) - (x_lp + x_ll)) + (st_lp + st_ll) - (y_lp + y_ll));
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
    if (a.ring) a.ring[(size_t)(a.ring_pos % a.ring_P) * a.NP + c] = acc ? 1 : 0;
    if (a.rec_stats) {
      const double l1 = a.lp[(size_t)q * a.NP + c], l2 = a.ll[(size_t)q * a.NP + c];
      a.rec_stats[c * 3 + 0] = l1;
      a.rec_stats[c * 3 + 1] = l2;
      a.rec_stats[c * 3 + 2] = l1 + l2;
    }
    if (a.rec_acc) a.rec_acc[c] = acc ? 1 : 0;
  }
  if (a.rec_params && lj) a.rec_params[c * d + lane] = acc ? yj : xj;

  // ---------------- model outputs of the aligned links ----------------
  const double fq_cur = lo ? (acc ? fnew : a.Fcur[q][c * MP + lane]) : 0.0;
  const double fk_cur = lo ? (acc ? a.Fcur[k][c * MP + lane] : FS(k, q)[lane]) : 0.0;
  if (lo) {
    if (acc) {
      a.Fcur[q][c * MP + lane] = fnew;
    } else {
      for (int j = 0; j < q; ++j) a.Fcur[j][c * MP + lane] = FS(j, q)[lane];
    }
    for (int j = 0; j < q; ++j) {
      const double fj = a.Fcur[j][c * MP + lane];  // this very lane wrote / owns the entry
      for (int q2 = j + 1; q2 <= q; ++q2) FS(j, q2)[lane] = fj;
    }
  }

  // ---------------- error model update (chain.py:485-499, :739-753; proposal.py:1547-1578; utils.py:113-122) ----------------
  const double diff_new = fq_cur - fk_cur;
  double* md = a.mdiff[q] + c * MP;
  const double t = (double)a.b_t;
  double* up = a.upd + (size_t)c * 3 * MP;  // vectors of the tracker's covariance update, applied by k_aem_refresh (see k_aem_action)
  if (a.dependent) {  // chain.py:501-523; utils.py:199: zero-mean moments of the change of the difference
    const double xupd = lo ? fq_cur - (fk_cur + md[lane]) : 0.0;
    if (lo) md[lane] = diff_new;
    up[lane] = xupd;
    up[MP + lane] = 0.0;
    up[2 * MP + lane] = 0.0;
    if (lo) a.bias_tot[k][c * MP + lane] = diff_new;  // the bias of the coarse level is the last difference
    a.rvec[c * MP + lane] = lo ? (fk_cur - a.data[k][lane]) + diff_new : 0.0;  // for the update_link at the end of k_aem_refresh
    return;
  }
  const double dm = (a.is_da || acc) ? diff_new : (lo ? md[lane] : 0.0);  // MLDA refreshes the difference on accept only
  if (lo) md[lane] = dm;
  double* mu = a.b_mu[q] + c * MP;
  const double mu_o = lo ? mu[lane] : 0.0;
  const double mu_n = (1.0 / (t + 1.0)) * (t * mu_o + dm);
  up[lane] = lo ? dm : 0.0;
  up[MP + lane] = mu_o;
  up[2 * MP + lane] = lo ? mu_n : 0.0;
  if (lo) mu[lane] = mu_n;
  __threadfence_block();
  __syncthreads();
  double bt = 0.0;
  if (lo) {  // total bias of level k: sum over the trackers of the levels above
    for (int p = q; p < nl; ++p) bt += a.b_mu[p][c * MP + lane];
    a.bias_tot[k][c * MP + lane] = bt;
  }
  a.rvec[c * MP + lane] = lo ? (fk_cur - a.data[k][lane]) + bt : 0.0;  // for the update_link at the end of k_aem_refresh
}

// base-level step of such a hierarchy: like k_ext_accept, with the bias-corrected dense likelihood of
// AdaptiveGaussianLogLike (distributions.py:404-425) under chain c's bias and inverse; keeps the output of the current link
struct ExtAemAcceptArgs {
  int64_t N, NP;
  int d, DP, m, MP, s, prop_kind;
  double* theta;  // level 0: [NP][DP]
  double* lp;
  double* ll;
  const double* u;     // [S][NP]
  const double* prop;  // [N][d]
  const double* F;     // [N][m]
  const double* data;  // [MP]
  const double* bias;  // [NP][MP]
  const double* P;     // [NP][tiles][4][64] lower tiles of V = L^-1 (tda_kernels_aemr.h)
  double* Fcur;        // [NP][MP]
  const double* pr_mean;
  const double* pr_pinv;
  const double* pr_lo;  // support bounds of uniform prior components, or null
  const double* pr_hi;
  double logconst;
  int32_t* anyacc;
  int64_t* sid;
  int64_t sid_value;  // identity given to the parameter vector this step creates
  uint8_t* ring;
  int ring_P;
  int64_t ring_pos;
  double* rec_params;
  double* rec_stats;
  uint8_t* rec_acc;
};

template <int MPT>
__global__ void __launch_bounds__(MPT) k_ext_aem_accept(const ExtAemAcceptArgs a) {
  constexpr int NW = MPT / 64;
  __shared__ double s_r[MPT];
  __shared__ double s_x[8];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const int MP = a.MP;
  const bool lo = lane < a.m, lj = lane < a.d;
  auto bsum = [&](double v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if constexpr (NW > 1) {
      __syncthreads();
      if ((lane & 63) == 0) s_x[lane >> 6] = v;
      __syncthreads();
      v = s_x[0] + s_x[1];
      if constexpr (NW == 4) v += s_x[2] + s_x[3];
    }
    return v;
  };
  const double f = lo ? a.F[(size_t)c * a.m + lane] : 0.0;
  const double r = lo ? (f + a.bias[c * MP + lane]) - a.data[lane] : 0.0;
  s_r[lane] = r;
  __syncthreads();
  double ll_n;
  {
    const double* __restrict__ Vc = a.P + (size_t)c * aemr_v_doubles(MPT);
    double sq = (lane >> 6) == 0 ? sum_rows(aem_quad_factor<MPT / 16>(Vc, s_r, lane & 63)) : 0.0;
    if constexpr (NW > 1) {
      __syncthreads();
      if ((lane & 63) == 0) s_x[lane >> 6] = sq;
      __syncthreads();
      sq = s_x[0] + s_x[1];
      if constexpr (NW == 4) sq += s_x[2] + s_x[3];
    }
    ll_n = -0.5 * sq;
  }
  const double prp = lj ? a.prop[c * a.d + lane] : 0.0;
  double pj = 0.0;
  if (lj) {
    const double dv = prp - a.pr_mean[lane];
    pj = dv * dv * a.pr_pinv[lane];
    if (a.pr_lo && (prp < a.pr_lo[lane] || prp > a.pr_hi[lane])) pj = __builtin_inf();  // uniform prior components
  }
  const double maha = bsum(pj);
  const double lp_n = -0.5 * (a.logconst + maha);
  const double post_n = lp_n + ll_n;
  double lp = a.lp[c], ll = a.ll[c];
  const double delta = a.prop_kind == 1 ? ll_n - ll : post_n - (lp + ll);
  double alpha = exp(delta);
  if (post_n != post_n) alpha = 0.0;
  const bool acc = a.u[(size_t)a.s * a.NP + c] < alpha;
  double cur = lj ? a.theta[c * a.DP + lane] : 0.0;
  if (acc) {
    lp = lp_n;
    ll = ll_n;
    cur = prp;
    if (lj) a.theta[c * a.DP + lane] = cur;
    if (lo) a.Fcur[c * MP + lane] = f;
    if (lane == 0) {
      a.lp[c] = lp;
      a.ll[c] = ll;
      a.anyacc[c] = 1;
      a.sid[c] = a.sid_value;
    }
  }
  const size_t rr = (size_t)a.s * a.N + c;
  if (lane == 0) {
    if (a.ring) a.ring[(size_t)(a.ring_pos % a.ring_P) * a.NP + c] = acc ? 1 : 0;
    if (a.rec_stats) {
      a.rec_stats[rr * 3 + 0] = lp;
      a.rec_stats[rr * 3 + 1] = ll;
      a.rec_stats[rr * 3 + 2] = lp + ll;
    }
    if (a.rec_acc) a.rec_acc[rr] = acc ? 1 : 0;
  }
  if (a.rec_params && lj) a.rec_params[rr * a.d + lane] = cur;
}

}  // namespace tda
