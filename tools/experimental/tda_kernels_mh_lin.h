// EXPERIMENTAL: fused MH steps with the RESIDUAL as chain state (tools/steps_microbench.hip, kernel 3).
//
// The forward model is linear, so A theta' - ytil = (A theta - ytil) + A (s inc): with the residual r of the current
// state kept per chain (LDS, 8 waves x 8 blocks x 4 x 64 doubles = 128 KiB for m = 1024), the MFMAs of a step only need
// the scaled increment as B operand -- which does not depend on the previous step's decision.  The first block of step
// s + 1 is issued right after the barrier of step s and runs while the partial sums are reduced, the decision is taken
// and the accepted residuals are written back; the matrix pipe idles for ~400 cycles per step instead of ~3 000.
// r is re-derived from theta at every launch (one extra MFMA pass per <= 128 steps), so rounding cannot accumulate
// beyond a block.  GRW / AM proposals, iso / diag noise, diagonal priors, m <= 1024.
#pragma once
#include "tda_kernels_mh_frag.h"

namespace tda {

template <int DPAD>
__host__ __device__ constexpr int steps_lin_lds_doubles(int m_pad, bool diag) {
  return 2 * 2 * 16 * 8 + 2 * 2 * 16 + 2 * DPAD + 2 * 64 * (DPAD / 4 + 2) + m_pad + (diag ? m_pad : 0) + 8 * 8 * 256;
}

template <int DPAD, bool DIAG, int PRI>
__global__ void __launch_bounds__(512, 2) k_mh_steps_lin(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int KS = DPAD / 4, K2 = DPAD / 8;
  constexpr int NW = 8, NT = 64 * NW, MAXB = 8;
  constexpr int RSX = KS + 2;
  constexpr int EPT = 16 * DPAD >= NT ? 16 * DPAD / NT : 1;
  double* s_red = smem;                  // [2][NW][16]  partial sums of squares
  double* s_redp = s_red + 2 * 16 * NW;  // [2][NW][16]  partial prior sums (each wave owns two of the 16 theta fragments)
  double* s_u = s_redp + 2 * 16 * NW;    // [2][2][16]
  double* s_pm = s_u + 2 * 2 * 16;
  double* s_pinv = s_pm + DPAD;
  double* s_sx = s_pinv + DPAD;          // [2][64][RSX] scaled increments in fragment order
  double* s_y = s_sx + 2 * 64 * RSX;
  double* s_w = s_y + a.lv.m_pad;
  double* s_r = s_w + (DIAG ? a.lv.m_pad : 0);  // [NW][MAXB][4][64] residuals of the current states

  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, hi = lane >> 4;
  const int64_t tile0 = (int64_t)blockIdx.x * 16;
  const int64_t gcl = tile0 + lc;
  const int ncb = a.lv.ncb;
  const int nbw = wave < ncb ? (ncb - wave + NW - 1) / NW : 0;  // blocks wave, wave + NW, ... (at most MAXB)

  for (int i = tid; i < a.lv.m_pad; i += NT) {
    s_y[i] = a.lv.ytil[i];
    if (DIAG) s_w[i] = a.lv.w[i];
  }
  for (int i = tid; i < DPAD; i += NT) {
    s_pm[i] = a.pr.mean[i];
    s_pinv[i] = a.pr.pinv[i];
  }
  const bool has_logu = a.logu != nullptr;
  // Each wave carries only two of the 16 fragments of theta (kk = wave and wave + 8: dims 4 kk + hi of chain lc): the
  // matrix phase needs the increments, not theta; prior and records are split over the waves the same way.
  const int kk0 = wave, kk1 = wave + NW;
  const bool own0 = kk0 < KS, own1 = kk1 < KS;
  const double* __restrict__ th_row = a.theta + gcl * DPAD + hi;
  double cur0 = own0 ? th_row[4 * kk0] : 0.0, cur1 = own1 ? th_row[4 * kk1] : 0.0;
  double lp = a.lp[gcl], ll = a.ll[gcl];
  int nacc = 0;

  // ---- thread-mapped staging of the increments (see k_mh_steps_frag) ----
  const int e0 = tid * EPT;
  const bool st_on = e0 < 16 * DPAD;
  const int st_c = st_on ? e0 / DPAD : 0, st_j = st_on ? e0 % DPAD : 0;
  const double st_scal = a.scaling[tile0 + st_c];
  const double* __restrict__ st_src = a.inc + (size_t)(tile0 + st_c) * DPAD + st_j;
  const size_t inc_step = (size_t)a.NP * DPAD;
  int st_dst[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) st_dst[e] = (((st_j + e) & 3) * 16 + st_c) * RSX + ((st_j + e) >> 2);
  const bool uw = wave == 0 && lane < 16;
  double st_x[EPT], st_u = 0.5, st_lu = 0.0;
  auto stage_load = [&](int s) {
    if (st_on) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) st_x[e] = st_src[(size_t)s * inc_step + e];
    }
    if (uw) {
      st_u = a.u[(size_t)s * a.NP + gcl];
      if (has_logu) st_lu = a.logu[(size_t)s * a.NP + gcl];
    }
  };
  auto stage_store = [&](int s) {
    double* __restrict__ dst = s_sx + (s & 1) * 64 * RSX;
    if (st_on) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) dst[st_dst[e]] = st_scal * st_x[e];
    }
    if (uw) {
      s_u[(s & 1) * 32 + lane] = st_u;
      s_u[(s & 1) * 32 + 16 + lane] = st_lu;
    }
  };
  double2 sx[K2];
  double u, lu, sx0, sx1;  // sx0 / sx1: the increments of the two fragments this wave owns
  auto frag_fetch = [&](int s, double2 (&dst)[K2], double& o0, double& o1) {
    const double* __restrict__ rowd = s_sx + (s & 1) * 64 * RSX + lane * RSX;
    const double2* __restrict__ row = reinterpret_cast<const double2*>(rowd);
#pragma unroll
    for (int k = 0; k < K2; ++k) dst[k] = row[k];
    o0 = own0 ? rowd[kk0] : 0.0;
    o1 = own1 ? rowd[kk1] : 0.0;
  };

  // ---- records (deferred into the next step's matrix phase, see k_mh_steps_frag) ----
  bool acc_prev = false;
  const bool rec_lane = gcl < a.N;
  const bool rec_p0 = a.rec_params != nullptr && rec_lane && own0 && 4 * kk0 + hi < a.d;
  const bool rec_p1 = a.rec_params != nullptr && rec_lane && own1 && 4 * kk1 + hi < a.d;
  double* __restrict__ const rec_p = a.rec_params + (size_t)gcl * a.d + hi + 4 * wave;
  const size_t rec_p_step = (size_t)a.N * a.d;
  const bool rec_s = wave == 1 && lane < 16 && rec_lane;
  auto write_records = [&](int s) {
    if (rec_s) {
      const size_t r = (size_t)s * a.N + gcl;
      if (a.rec_stats) {
        a.rec_stats[r * 3 + 0] = lp;
        a.rec_stats[r * 3 + 1] = ll;
        a.rec_stats[r * 3 + 2] = lp + ll;
      }
      if (a.rec_acc) a.rec_acc[r] = acc_prev ? 1 : 0;
    }
    double* __restrict__ row = rec_p + (size_t)s * rec_p_step;
    if (rec_p0) row[0] = cur0;
    if (rec_p1) row[4 * NW] = cur1;
  };

  // ---- A-fragment access ----
  // Fragments through buffer loads: resource descriptor (4 SGPRs) + scalar block offset + one 32-bit lane offset.  With
  // per-lane 64-bit pointers hipcc precomputes an address pair per block half and keeps ~80 registers of addresses
  // alive across the step loop.
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  const __amdgpu_buffer_rsrc_t arsrc =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(a.lv.Apk), 0, 0x7fffffff, 0x00020000);
  const int lane_off = lane * 16;
  const int first = wave < ncb ? wave : ncb - 1;
  auto ldA = [&](double2 (&f)[K2], int i) {  // this wave's i-th block (wraps to its first block past the end)
    const int cb = wave + i * NW;
    const int soff = (cb < ncb ? cb : first) * (K2 * 1024);
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(arsrc, lane_off + (k & 3) * 1024, soff + (k >> 2) * 4096, 0);
      f[k] = *reinterpret_cast<const double2*>(&v);
    }
  };
  auto chain = [&](const double2 (&f)[K2], const double2 (&b)[K2]) {
    double4_t g = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      g = mfma_f64(f[k].x, b[k].x, g);
      g = mfma_f64(f[k].y, b[k].y, g);
    }
    return g;
  };
  double* const my_r = s_r + (size_t)wave * MAXB * 256 + lane;  // element (i, r) at my_r[(4 i + r) 64]
  double2 fa[K2], fb[K2];

  stage_load(0);
  __syncthreads();  // s_y staged
  // ---- residuals of the current states: r = A theta - ytil for this wave's blocks ----
  {
    double2 cb2[K2];
#pragma unroll
    for (int k = 0; k < K2; ++k) cb2[k] = double2{th_row[8 * k], th_row[8 * k + 4]};
    ldA(fa, 0);
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
      if (i < nbw) {
        if (i & 1) ldA(fa, i + 1); else ldA(fb, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        const double4_t g = (i & 1) ? chain(fb, cb2) : chain(fa, cb2);
#pragma unroll
        for (int r = 0; r < 4; ++r) my_r[(4 * i + r) * 64] = g[r] - s_y[(wave + i * NW) * 16 + hi + 4 * r];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  stage_store(0);
  __syncthreads();
  frag_fetch(0, sx, sx0, sx1);
  u = s_u[lc];
  lu = s_u[16 + lc];
  ldA(fa, 0);
  ldA(fb, 1);
  __builtin_amdgcn_sched_barrier(0);
  double4_t X = chain(fa, sx);  // block 0 of step 0

#ifdef TDA_STEP_TRACE
  const bool tracing = a.trace != nullptr && blockIdx.x == 0;
  long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define TDA_STAMP(i) \
  if (tracing) stamp[i] = (long long)__builtin_amdgcn_s_memtime()
#else
#define TDA_STAMP(i)
#endif
  for (int s = 0; s < a.S; ++s) {
    TDA_STAMP(0);
    const bool more = s + 1 < a.S;
    if (more) stage_load(s + 1);
    if (s > 0) write_records(s - 1);
    // this wave's share of the prior of theta' = theta + sx (scipy MVN logpdf, posterior.py:92)
    double pp = 0.0;
    {
      const double t0 = cur0 + sx0, t1 = cur1 + sx1;
      if (PRI == 2) {
        const double d0 = t0 - s_pm[(4 * kk0 + hi) & (DPAD - 1)], d1 = t1 - s_pm[(4 * kk1 + hi) & (DPAD - 1)];
        pp = own0 ? d0 * d0 * s_pinv[(4 * kk0 + hi) & (DPAD - 1)] : 0.0;
        pp += own1 ? d1 * d1 * s_pinv[(4 * kk1 + hi) & (DPAD - 1)] : 0.0;
      } else {
        pp = t0 * t0;
        pp += t1 * t1;
      }
    }
    // ---- matrix phase: blocks 1 .. nbw-1 (block 0 was issued behind the previous barrier), epilogue one block behind ----
    // eight statically named accumulators (an array indexed through a lambda ends up in scratch memory)
    double4_t G0 = X, G1, G2, G3, G4, G5, G6, G7;
    G1 = G2 = G3 = G4 = G5 = G6 = G7 = double4_t{0.0, 0.0, 0.0, 0.0};
    double sse = 0.0;
    if (nbw <= 1) ldA(fa, 0);  // (a wave with a single block: its fragments again for the next step)
#define TDA_EPI(I, GI)                                                   \
  {                                                                      \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                      \
      const double rp = my_r[(4 * (I) + r) * 64] + GI[r];                \
      GI[r] = rp;                                                        \
      double sq = rp * rp;                                               \
      if (DIAG) sq *= s_w[(wave + (I) * NW) * 16 + hi + 4 * r];          \
      sse += sq;                                                         \
    }                                                                    \
  }
#define TDA_BLK(I, GI, GPREV)                                            \
  if ((I) < nbw) {                                                       \
    if ((I) + 1 < nbw) {                                                 \
      if ((I) & 1) ldA(fa, (I) + 1); else ldA(fb, (I) + 1);              \
    }                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                   \
    GI = ((I) & 1) ? chain(fb, sx) : chain(fa, sx);                      \
    if ((I) + 1 == nbw) ldA(fa, 0);                                      \
    TDA_EPI((I) - 1, GPREV)                                              \
    __builtin_amdgcn_sched_barrier(0);                                   \
  } else if ((I) == nbw) {                                               \
    TDA_EPI((I) - 1, GPREV)                                              \
  }
    TDA_BLK(1, G1, G0)
    TDA_BLK(2, G2, G1)
    TDA_BLK(3, G3, G2)
    TDA_BLK(4, G4, G3)
    TDA_BLK(5, G5, G4)
    TDA_BLK(6, G6, G5)
    TDA_BLK(7, G7, G6)
    TDA_STAMP(2);
    if (nbw == 8) TDA_EPI(7, G7)
#undef TDA_BLK
    TDA_STAMP(3);
    if (more) stage_store(s + 1);
    pp = sum_rows(pp);
    sse = sum_rows(sse);
    double* s_red_s = s_red + (s & 1) * 16 * NW;
    double* s_redp_s = s_redp + (s & 1) * 16 * NW;
    if (lane < 16) {
      s_red_s[wave * 16 + lane] = sse;
      s_redp_s[wave * 16 + lane] = pp;
    }
    TDA_STAMP(1);
    __syncthreads();
    TDA_STAMP(4);
    const double u_s = u, lu_s = lu;
    double2 sxn[K2];
    double nx0 = 0.0, nx1 = 0.0;
    if (more) {  // block 0 of step s + 1 runs while this step is decided
      frag_fetch(s + 1, sxn, nx0, nx1);
      u = s_u[((s + 1) & 1) * 32 + lc];
      lu = s_u[((s + 1) & 1) * 32 + 16 + lc];
      X = chain(fa, sxn);
    }
    TDA_STAMP(5);
    double tot = s_red_s[lc], maha = s_redp_s[lc];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      tot += s_red_s[w * 16 + lc];
      maha += s_redp_s[w * 16 + lc];
    }
    const double ll_n = DIAG ? -0.5 * tot : -0.5 * tot / a.lv.var;
    const double lp_n = -0.5 * (a.pr.logconst + maha);
    const double post_n = lp_n + ll_n;
    bool acc;
    {
      const double delta = post_n - (lp + ll);
      if (has_logu && (fabs(lu_s - delta) > 1e-9 || delta != delta)) acc = (post_n == post_n) && (lu_s < delta);
      else acc = accept_exact(u_s, delta, post_n);
    }
    if (acc) {
      lp = lp_n;
      ll = ll_n;
#define TDA_WB(I, GI)                                                                      \
  if ((I) < nbw) {                                                                         \
    _Pragma("unroll") for (int r = 0; r < 4; ++r) my_r[(4 * (I) + r) * 64] = GI[r];        \
  }
      TDA_WB(0, G0) TDA_WB(1, G1) TDA_WB(2, G2) TDA_WB(3, G3) TDA_WB(4, G4) TDA_WB(5, G5) TDA_WB(6, G6) TDA_WB(7, G7)
#undef TDA_WB
    }
    TDA_STAMP(6);
    nacc += acc ? 1 : 0;
    acc_prev = acc;
#undef TDA_EPI
    // proposal.py:249-251: theta' = theta + scaling * inc, kept if accepted
    cur0 = acc ? cur0 + sx0 : cur0;
    cur1 = acc ? cur1 + sx1 : cur1;
    sx0 = nx0;
    sx1 = nx1;
#pragma unroll
    for (int k = 0; k < K2; ++k) sx[k] = sxn[k];
    if (more) ldA(fb, 1);  // block 1 of the next step (the accumulators are free again)
    TDA_STAMP(7);
#ifdef TDA_STEP_TRACE
    if (tracing && lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a.trace[((size_t)s * NW + wave) * 8 + i] = stamp[i];
    }
#endif
  }
#undef TDA_STAMP
  if (a.S > 0) write_records(a.S - 1);

  {
    double* __restrict__ row = a.theta + gcl * DPAD + hi;
    if (own0) row[4 * kk0] = cur0;
    if (own1) row[4 * kk1] = cur1;
  }
  if (wave == 0 && lane < 16) {
    a.lp[gcl] = lp;
    a.ll[gcl] = ll;
    if (a.acc_count) a.acc_count[gcl] += nacc;
  }
}

}  // namespace tda
