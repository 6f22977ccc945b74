// EXPERIMENTAL variants of the fused MH-step kernel, kept for tools/steps_microbench.hip only (not built into the library).
//
//   k_mh_steps_frag<D, DIAG, PRI, 8>  chain state held in MFMA B-operand layout by every wave, increments staged through
//                                     LDS in fragment order, ONE barrier per step, records written from inside the next
//                                     step's matrix phase
//   k_mh_steps_frag<D, DIAG, PRI, 4>  the same with one wave per SIMD and a software-pipelined matrix phase (epilogue of
//                                     block i-1 and the loads of block i+2 issued between the MFMAs of block i)
//
// Findings on MI355X (C2a shape): all three kernels land within 3 % of each other
// (2.55-2.75 ns/eval).  Removing the second barrier, the LDS transpose and the cross-lane broadcast from the serial
// end of a step does not shorten the step; what bounds it is the matrix phase itself: two waves per SIMD reach
// ~1190 cycles per 16-MFMA block (1024 ideal) because the older wave wins every arbitration and the younger one runs
// its last blocks alone; one software-pipelined wave per SIMD reaches only ~1330.
#pragma once
#include "tda_kernels_mh.h"

namespace tda {

// ------------------------------------------------------------------------------------------------
// 8-wave tile with the chain state kept in MFMA B-operand layout (iso / diag noise; the hot kernel of C2a).
// Every wave holds theta of the tile's 16 chains as 16 fragments (lane (lc, hi) <-> theta[lc][4 kk + hi]) and the
// next step's increments in the same layout, prefetched from HBM/L2 during the matrix phase.  After the single
// barrier of a step every lane knows the decision of its own chain (lc), so the proposal of the next step is two
// VALU ops per fragment away: no LDS transpose, no second barrier, no cross-lane broadcast on the critical path.
// `tools/step_trace.py` shows the serial phase of the thread-mapped kernel above at ~3100 of 22 800 cycles per step.
// ------------------------------------------------------------------------------------------------
template <int DPAD, int MODE>
__device__ __forceinline__ double4_t block_mfma(const double2 (&f)[DPAD / 8], const double (&th)[DPAD / 4]) {
  double4_t a0 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < DPAD / 8; ++k) {
    a0 = mfma_f64(f[k].x, th[2 * k], a0);
    a0 = mfma_f64(f[k].y, th[2 * k + 1], a0);
  }
  return a0;
}
template <int MODE>
__device__ __forceinline__ double block_epilogue(const double4_t a0, const double* __restrict__ s_y,
                                                 const double* __restrict__ s_w, int cb, int hi) {
  double sse = 0.0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int o = cb * 16 + hi + 4 * r;
    const double res = a0[r] - s_y[o];
    double sq = res * res;
    if (MODE == 1) sq *= s_w[o];
    sse += sq;
  }
  return sse;
}

// First block of a step in the fragment-state kernel: theta' = cur + sx (pCN: keep cur + sx) is formed fragment by
// fragment between the MFMAs that consume it -- the scaled increments sx arrive from LDS while the first MFMAs already
// run -- and the diagonal-prior Mahalanobis sum (PRI: 1 = N(0, I), 2 = general diagonal) rides along on the VALU.
template <int DPAD, int PRI>
__device__ __forceinline__ double4_t first_block(const double2 (&f)[DPAD / 8], const double (&cur)[DPAD / 4],
                                                 const double2 (&sx)[DPAD / 8], double (&th)[DPAD / 4], bool is_pcn,
                                                 double keep, const double* __restrict__ s_pm,
                                                 const double* __restrict__ s_pinv, int hi, double& prior_partial) {
  constexpr int K2 = DPAD / 8;
  // proposal.py:249-251 / :351-355.  keep1 = 1 for the random walk: 1.0 * cur is exact, so one code path serves both.
  const double keep1 = is_pcn ? keep : 1.0;
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  double p = 0.0;
  double t0 = keep1 * cur[0] + sx[0].x, t1 = keep1 * cur[1] + sx[0].y;
#pragma unroll
  for (int k = 0; k < K2; ++k) {
    th[2 * k] = t0;
    th[2 * k + 1] = t1;
    acc = mfma_f64(f[k].x, t0, acc);
    // the next pair of fragments and this pair's prior terms are formed while the MFMA above runs
    double n0 = 0.0, n1 = 0.0;
    if (k + 1 < K2) {
      n0 = keep1 * cur[2 * k + 2] + sx[k + 1].x;
      n1 = keep1 * cur[2 * k + 3] + sx[k + 1].y;
    }
    if (PRI == 2) {
      const double d0 = t0 - s_pm[8 * k + hi], d1 = t1 - s_pm[8 * k + 4 + hi];
      p += d0 * d0 * s_pinv[8 * k + hi];  // ascending kk within a lane, as in k_mh_steps
      p += d1 * d1 * s_pinv[8 * k + 4 + hi];
    } else {
      p += t0 * t0;
      p += t1 * t1;
    }
    acc = mfma_f64(f[k].y, t1, acc);
    __builtin_amdgcn_sched_barrier(0);
    t0 = n0;
    t1 = n1;
  }
  prior_partial = p;
  return acc;
}

// The forward-model pipeline of level_sse_single with the first pair of blocks peeled: the first block builds theta'
// (see first_block) and is computed unconditionally -- a wave without any block (fewer than NW blocks) still needs
// theta' and the prior, because every wave takes the accept decision for itself.
template <int DPAD, int MODE, int NW, int PRI, class Mid>
__device__ __forceinline__ double level_sse_frag(const double* __restrict__ Apk, int ncb,
                                                 const double* __restrict__ s_y, const double* __restrict__ s_w,
                                                 const double (&cur)[DPAD / 4], const double2 (&sx)[DPAD / 8],
                                                 double (&th)[DPAD / 4], bool is_pcn, double keep, int wave, int lane,
                                                 double2 (&fa)[DPAD / 8], const double* __restrict__ s_pm,
                                                 const double* __restrict__ s_pinv, double& prior_partial, Mid&& mid,
                                                 long long* stamp = nullptr) {
  const int hi = lane >> 4;
  const FragSrc base = frag_src(Apk, lane);
  const int first = wave < ncb ? wave : ncb - 1;
  double sse = 0.0;
  double2 fb[DPAD / 8];
  bool next_in_fb = false;
  frag_load_buf<DPAD>(base, (wave + NW) < ncb ? (wave + NW) : first, fb);
  __builtin_amdgcn_sched_barrier(0);
#ifdef TDA_STEP_TRACE
  if (stamp) stamp[2] = (long long)__builtin_amdgcn_s_memtime();
#endif
  {
    const double4_t acc = first_block<DPAD, PRI>(fa, cur, sx, th, is_pcn, keep, s_pm, s_pinv, hi, prior_partial);
#ifdef TDA_STEP_TRACE
    if (stamp) stamp[3] = (long long)__builtin_amdgcn_s_memtime();
#endif
    mid();  // the caller's prefetch of the next step's inputs: issued behind the first 2 x KS MFMAs
    const double e = block_epilogue<MODE>(acc, s_y, s_w, first, hi);
    sse += wave < ncb ? e : 0.0;
  }
  __builtin_amdgcn_sched_barrier(0);
  if (wave + NW < ncb) {
    frag_load_buf<DPAD>(base, (wave + 2 * NW) < ncb ? (wave + 2 * NW) : first, fa);
    __builtin_amdgcn_sched_barrier(0);
    const double4_t acc = block_mfma<DPAD, MODE>(fb, th);
    sse += block_epilogue<MODE>(acc, s_y, s_w, wave + NW, hi);
    __builtin_amdgcn_sched_barrier(0);
  } else {
    next_in_fb = true;
  }
  for (int cb = wave + 2 * NW; cb < ncb; cb += 2 * NW) {
#ifdef TDA_PRIO_SWITCH
    // The SIMD's older wave wins the MFMA arbitration, so the younger one (waves NW/2..) would finish ~2-4k cycles
    // later and then run alone; it takes the priority over for the last part of the phase instead.
    if (wave >= NW / 2 && cb >= wave + (TDA_PRIO_SWITCH) * 2 * NW && cb < wave + (TDA_PRIO_SWITCH + 1) * 2 * NW)
      __builtin_amdgcn_s_setprio(3);
#endif
    frag_load_buf<DPAD>(base, (cb + NW) < ncb ? (cb + NW) : first, fb);
    __builtin_amdgcn_sched_barrier(0);
    {
      const double4_t acc = block_mfma<DPAD, MODE>(fa, th);
      sse += block_epilogue<MODE>(acc, s_y, s_w, cb, hi);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (cb + NW < ncb) {
      frag_load_buf<DPAD>(base, (cb + 2 * NW) < ncb ? (cb + 2 * NW) : first, fa);
      __builtin_amdgcn_sched_barrier(0);
      const double4_t acc = block_mfma<DPAD, MODE>(fb, th);
      sse += block_epilogue<MODE>(acc, s_y, s_w, cb + NW, hi);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      next_in_fb = true;
    }
  }
#ifdef TDA_PRIO_SWITCH
  __builtin_amdgcn_s_setprio(0);
#endif
  if (next_in_fb) {
#pragma unroll
    for (int k = 0; k < DPAD / 8; ++k) fa[k] = fb[k];
  }
  return sse;  // fa holds block `first` again, ready for the next step
}

// Gapless variant of the forward-model pipeline for ONE wave per SIMD (4-wave tile, up to 512 registers): the epilogue
// of block i-1 (LDS reads of the data, residual, square, accumulate) is issued between the MFMAs of block i, which run
// on a second accumulator, and the fragments of block i+2 are requested as soon as block i has been issued (three
// fragment sets, three accumulators, loop unrolled by three).  A wave therefore keeps the matrix pipe busy by itself --
// with two waves per SIMD the older one wins every arbitration, the younger one only fills the ~500-cycle gaps
// behind each of its chains and then finishes the step alone (tools/steps_microbench.hip -DTDA_STEP_TRACE).
// On exit f0/f1/f2 are being refilled with this wave's blocks 0/1/2 for the next step.
// One wave alone cannot issue an f64 MFMA that accumulates into the result of the previous one back to back (measured:
// ~83 instead of 64 cycles per instruction), so a block's k-range is split over two accumulators that alternate.
struct acc2_t {
  double4_t a, b;
};

template <int DPAD, int MODE, bool REFILL>
__device__ __forceinline__ double chain_with_epilogue(const double2 (&f)[DPAD / 8], const double (&th)[DPAD / 4],
                                                      acc2_t& acc, const acc2_t prev, int cb_prev,
                                                      const double* __restrict__ s_y, const double* __restrict__ s_w,
                                                      int hi, double2 (&refill)[DPAD / 8], const FragSrc& fsrc,
                                                      int src_cb) {
  constexpr int K2 = DPAD / 8;
  double y[4], w[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    y[r] = s_y[cb_prev * 16 + hi + 4 * r];
    if (MODE == 1) w[r] = s_w[cb_prev * 16 + hi + 4 * r];
  }
  double sse = 0.0;
  acc.a = double4_t{0.0, 0.0, 0.0, 0.0};
  acc.b = double4_t{0.0, 0.0, 0.0, 0.0};
  auto piece = [&](int r) {
    const double res = (prev.a[r] + prev.b[r]) - y[r];
    double sq = res * res;
    if (MODE == 1) sq *= w[r];
    sse += sq;
  };
#pragma unroll
  for (int k = 0; k < K2; ++k) {
    acc.a = mfma_f64(f[k].x, th[2 * k], acc.a);
    if (REFILL) {  // one 16-byte load per MFMA pair: the wave never sits in a burst of issues
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(fsrc.rsrc, fsrc.lane_off + (k & 3) * 1024, src_cb * (DPAD / 8) * 1024 + (k >> 2) * 4096, 0);
      refill[k] = *reinterpret_cast<const double2*>(&v);
    }
    if (k >= 1 && k - 1 < 4) piece(k - 1);
    acc.b = mfma_f64(f[k].y, th[2 * k + 1], acc.b);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = (K2 >= 1 ? K2 - 1 : 0); r < 4; ++r) piece(r);
  return sse;
}

template <int DPAD, int MODE, int NW, int PRI, class Mid>
__device__ __forceinline__ double level_sse_pipe(const double* __restrict__ Apk, int ncb,
                                                 const double* __restrict__ s_y, const double* __restrict__ s_w,
                                                 const double (&cur)[DPAD / 4], const double2 (&sx)[DPAD / 8],
                                                 double (&th)[DPAD / 4], bool is_pcn, double keep, int wave, int lane,
                                                 double2 (&f0)[DPAD / 8], double2 (&f1)[DPAD / 8], double2 (&f2)[DPAD / 8],
                                                 const double* __restrict__ s_pm, const double* __restrict__ s_pinv,
                                                 double& prior_partial, Mid&& mid, long long* stamp = nullptr) {
  const int hi = lane >> 4;
  const FragSrc base = frag_src(Apk, lane);
  const int nb = wave < ncb ? (ncb - wave + NW - 1) / NW : 0;  // blocks wave, wave + NW, ... of this wave
  double sse = 0.0;
  acc2_t a0, a1, a2;
#ifdef TDA_STEP_TRACE
  if (stamp) stamp[2] = (long long)__builtin_amdgcn_s_memtime();
#endif
  a0.a = first_block<DPAD, PRI>(f0, cur, sx, th, is_pcn, keep, s_pm, s_pinv, hi, prior_partial);
  a0.b = double4_t{0.0, 0.0, 0.0, 0.0};
#ifdef TDA_STEP_TRACE
  if (stamp) stamp[3] = (long long)__builtin_amdgcn_s_memtime();
#endif
  mid();  // the caller's requests for the next step's inputs and the previous step's records
  __builtin_amdgcn_sched_barrier(0);
  // chain j runs on set j % 3 and accumulates into a{j % 3}; the epilogue of chain j - 1 rides along.  Full triples in
  // a plain counted loop (a loop with exits in the middle makes hipcc fall back to near-zero vmcnt waits at its
  // header, i.e. no prefetch), then up to two leftover chains and the last epilogue.
  auto src_of = [&](int c) {  // this wave's c-th block, clamped into the matrix
    const int cb = wave + c * NW;
    return cb < ncb ? cb : ncb - 1;
  };
  int j = 1;
  for (; j + 2 < nb; j += 3) {
    // chain j also refills the set chain j - 1 has just released with block j + 2
    sse += chain_with_epilogue<DPAD, MODE, true>(f1, th, a1, a0, wave + (j - 1) * NW, s_y, s_w, hi, f0, base, src_of(j + 2));
    sse += chain_with_epilogue<DPAD, MODE, true>(f2, th, a2, a1, wave + j * NW, s_y, s_w, hi, f1, base, src_of(j + 3));
    sse += chain_with_epilogue<DPAD, MODE, true>(f0, th, a0, a2, wave + (j + 1) * NW, s_y, s_w, hi, f2, base, src_of(j + 4));
  }
  const int rem = nb - j;  // chains left: 0, 1 or 2 (negative: this wave has no block at all)
  if (rem >= 1) {
    sse += chain_with_epilogue<DPAD, MODE, false>(f1, th, a1, a0, wave + (j - 1) * NW, s_y, s_w, hi, f0, base, 0);
    if (rem >= 2) {
      sse += chain_with_epilogue<DPAD, MODE, false>(f2, th, a2, a1, wave + j * NW, s_y, s_w, hi, f0, base, 0);
      sse += block_epilogue<MODE>(a2.a + a2.b, s_y, s_w, wave + (j + 1) * NW, hi);
    } else {
      sse += block_epilogue<MODE>(a1.a + a1.b, s_y, s_w, wave + j * NW, hi);
    }
  } else if (rem == 0) {
    sse += block_epilogue<MODE>(a0.a + a0.b, s_y, s_w, wave + (j - 1) * NW, hi);
  }
  // next step's first three blocks (independent of theta'): in flight during the serial end of the step
  frag_load<DPAD>(base, wave, ncb, f0);
  frag_load<DPAD>(base, wave + NW, ncb, f1);
  frag_load<DPAD>(base, wave + 2 * NW, ncb, f2);
  return sse;
}

// v[OFF + w] for a wave-uniform w in [0, NW): a select chain over compile-time indices (a run-time index would send the
// register array to scratch)
template <int KS, int OFF, int NW, int W = NW - 1>
__device__ __forceinline__ double pick_frag(const double (&v)[KS], int w) {
  if constexpr (W == 0) {
    return v[OFF < KS ? OFF : 0];
  } else {
    const double r = pick_frag<KS, OFF, NW, W - 1>(v, w);
    if constexpr (OFF + W < KS) return w == W ? v[OFF + W] : r;
    else return r;
  }
}

template <int DPAD>
__host__ __device__ constexpr int steps_frag_lds_doubles(int m_pad, bool diag) {
  return 2 * 16 * 8 + 2 * 2 * 16 + 2 * DPAD + 2 * 64 * (DPAD / 4 + 2) + m_pad + (diag ? m_pad : 0);
}

// Noise kind (DIAG) and prior kind (PRI: 1 = N(0, I), 2 = general diagonal) are template parameters: with them as
// run-time flags hipcc spills 70-140 registers of the 256 this tile can have.  Dense priors, dense noise and
// single evaluations run on k_mh_steps above.
//
// Data movement per step: the tile's increments [16][DPAD] are fetched once, coalesced, by the threads of the
// workgroup (thread-mapped), scaled, and parked in LDS transposed into fragment order (row = lane, 16 contiguous
// values, row stride KS + 2 doubles: conflict-free ds_read_b128); after the step's barrier every wave pulls its own
// copy.  Fetching fragments straight from HBM/L2 instead costs 18 strided loads per wave and step, and the texture
// addresser -- already ~40 % busy streaming A -- then stretches the matrix phase by a quarter (measured).
template <int DPAD, bool DIAG, int PRI, int NW = 8>
__global__ void __launch_bounds__(64 * NW, NW / 4) k_mh_steps_frag(const StepArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  constexpr int KS = DPAD / 4;
  constexpr int NT = 64 * NW;
  constexpr int RSX = KS + 2;                      // row stride of the increment tile
  constexpr int EPT = 16 * DPAD >= NT ? 16 * DPAD / NT : 1;  // staged elements per thread (contiguous)
  double* s_red = smem;                    // [2][NW][16]  double-buffered by step parity (one barrier per step)
  double* s_u = s_red + 2 * 16 * NW;       // [2][2][16]   accept uniforms and their logs
  double* s_pm = s_u + 2 * 2 * 16;         // prior mean  [DPAD]
  double* s_pinv = s_pm + DPAD;            // prior 1/var [DPAD]
  double* s_sx = s_pinv + DPAD;            // [2][64][RSX] scaled increments in fragment order
  double* s_y = s_sx + 2 * 64 * RSX;
  double* s_w = s_y + a.lv.m_pad;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lc = lane & 15, hi = lane >> 4;
  const int64_t tile0 = (int64_t)blockIdx.x * 16;
  const int64_t gcl = tile0 + lc;

  for (int i = tid; i < a.lv.m_pad; i += NT) {
    s_y[i] = a.lv.ytil[i];
    if (DIAG) s_w[i] = a.lv.w[i];
  }
  for (int i = tid; i < DPAD; i += NT) {
    s_pm[i] = a.pr.mean[i];
    s_pinv[i] = a.pr.pinv[i];
  }

  const bool is_pcn = a.prop_kind == 1;
  const bool has_logu = a.logu != nullptr;
  double cur[KS];
  {
    const double* __restrict__ th_row = a.theta + gcl * DPAD + hi;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) cur[kk] = th_row[4 * kk];
  }
  double lp = a.lp[gcl], ll = a.ll[gcl];
  const double keep = is_pcn ? sqrt(1.0 - a.scaling[gcl] * a.scaling[gcl]) : 1.0;  // proposal.py:351-352
  int nacc = 0;

  // ---- thread-mapped staging of the increments: element e0 .. e0 + EPT - 1 of the tile's [16][DPAD] block ----
  const int e0 = tid * EPT;
  const bool st_on = e0 < 16 * DPAD;
  const int st_c = st_on ? e0 / DPAD : 0, st_j = st_on ? e0 % DPAD : 0;
  const double st_scal = a.scaling[tile0 + st_c];
  const double* __restrict__ st_src = a.inc + (size_t)(tile0 + st_c) * DPAD + st_j;
  const size_t inc_step = (size_t)a.NP * DPAD;
  // element j of chain c lives in row (j & 3) * 16 + c (= the lane that owns it), column j >> 2
  int st_dst[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) st_dst[e] = (((st_j + e) & 3) * 16 + st_c) * RSX + ((st_j + e) >> 2);
  const bool uw = wave == 0 && lane < 16;  // the lanes that fetch the tile's accept uniforms
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
  double2 sx[KS / 2];
  double u, lu;
  auto frag_fetch = [&](int s) {  // this lane's row of the increment tile: KS contiguous doubles
    const double2* __restrict__ row = reinterpret_cast<const double2*>(s_sx + (s & 1) * 64 * RSX + lane * RSX);
#pragma unroll
    for (int k = 0; k < KS / 2; ++k) sx[k] = row[k];
    u = s_u[(s & 1) * 32 + lc];
    lu = s_u[(s & 1) * 32 + 16 + lc];
  };
  // Records of step s (stats by wave 0, the parameter row split over the waves: two fragments each at d = 64) are
  // written from inside the matrix phase of step s + 1 -- state, lp and ll do not change before that step's decision.
  // Issued right after the decision, their write acknowledgements would be waited for at the top of the next step:
  // vmcnt retires in order, and the fragments prefetched for that step are older than the stores.
  // Wave w writes fragments w and w + NW of every chain (two 8-byte stores per lane at d = 64); the fragment is picked
  // with a select chain on the wave id -- sixteen predicated branches cost ~1500 cycles of issue time per step (traced).
  bool acc_prev = false;
  const bool rec_lane = gcl < a.N;
  constexpr int NFW = (KS + NW - 1) / NW;  // fragments per wave and chain: w, w + NW, ...
  bool rec_pj[NFW];
#pragma unroll
  for (int j = 0; j < NFW; ++j)
    rec_pj[j] = a.rec_params != nullptr && rec_lane && wave + j * NW < KS && 4 * (wave + j * NW) + hi < a.d;
  double* __restrict__ const rec_p = a.rec_params + (size_t)gcl * a.d + hi + 4 * wave;
  const size_t rec_p_step = (size_t)a.N * a.d;
  const bool rec_s = wave == 1 && lane < 16 && rec_lane;  // (wave 0 already stages the uniforms)
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
    if constexpr (NFW >= 1) { const double v = pick_frag<KS, 0, NW>(cur, wave); if (rec_pj[0]) row[0] = v; }
    if constexpr (NFW >= 2) { const double v = pick_frag<KS, NW, NW>(cur, wave); if (rec_pj[1]) row[4 * NW] = v; }
    if constexpr (NFW >= 3) { const double v = pick_frag<KS, 2 * NW, NW>(cur, wave); if (rec_pj[2]) row[8 * NW] = v; }
    if constexpr (NFW >= 4) { const double v = pick_frag<KS, 3 * NW, NW>(cur, wave); if (rec_pj[3]) row[12 * NW] = v; }
  };
  stage_load(0);
  stage_store(0);
  const FragSrc fbase = frag_src(a.lv.Apk, lane);
  double2 f0[KS / 2], f1[NW == 4 ? KS / 2 : 1], f2[NW == 4 ? KS / 2 : 1];
  frag_load<DPAD>(fbase, wave, a.lv.ncb, f0);  // later steps: prefetched by the previous step
  if constexpr (NW == 4) {
    frag_load<DPAD>(fbase, wave + NW, a.lv.ncb, f1);
    frag_load<DPAD>(fbase, wave + 2 * NW, a.lv.ncb, f2);
  }
  __syncthreads();
  frag_fetch(0);
  // debug trace (see k_mh_steps): the stamps stay in scalar registers and are written out once per step, so that
  // reading them back (s_memtime returns through lgkmcnt) adds no waits inside the phases being measured
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
    // ---- proposal, forward model + Gaussian log-likelihood (posterior.py:95-108, distributions.py:295-326), prior
    //      (scipy MVN logpdf, posterior.py:92); the next step's inputs are requested from inside the matrix phase ----
    double th[KS];
    double pp = 0.0;
    const bool more = s + 1 < a.S;
    auto prefetch = [&]() {
      if (more) stage_load(s + 1);
      if (s > 0) write_records(s - 1);
      TDA_STAMP(1);
    };
    double sse;
    if constexpr (NW == 4) {
      sse = level_sse_pipe<DPAD, DIAG ? 1 : 0, NW, PRI>(a.lv.Apk, a.lv.ncb, s_y, s_w, cur, sx, th, is_pcn, keep, wave, lane,
                                                       f0, f1, f2, s_pm, s_pinv, pp, prefetch
#ifdef TDA_STEP_TRACE
                                                       , tracing ? stamp : nullptr
#endif
      );
    } else {
      sse = level_sse_frag<DPAD, DIAG ? 1 : 0, NW, PRI>(a.lv.Apk, a.lv.ncb, s_y, s_w, cur, sx, th, is_pcn, keep, wave, lane,
                                                       f0, s_pm, s_pinv, pp, prefetch);
    }
    TDA_STAMP(4);
#ifndef TDA_EXP_NOSTAGE
    if (more) stage_store(s + 1);
#endif
    const double maha = sum_rows(pp);
    sse = sum_rows(sse);
    double* s_red_s = s_red + (s & 1) * 16 * NW;
    if (lane < 16) s_red_s[wave * 16 + lane] = sse;
    TDA_STAMP(5);
#ifndef TDA_EXP_NOSYNC
    __syncthreads();
#endif
    TDA_STAMP(6);

#ifndef TDA_EXP_NOSYNC
    double tot = s_red_s[lc];
#pragma unroll
    for (int w = 1; w < NW; ++w) tot += s_red_s[w * 16 + lc];
#else
    double tot = sse * 8.0;
#endif
    const double u_s = u, lu_s = lu;
#if !defined(TDA_EXP_FETCH_LATE) && !defined(TDA_EXP_NOSTAGE)
    if (more) frag_fetch(s + 1);  // in flight during the decision
#endif
    const double ll_n = DIAG ? -0.5 * tot : -0.5 * tot / a.lv.var;
    const double lp_n = -0.5 * (a.pr.logconst + maha);
    const double post_n = lp_n + ll_n;  // link.py:48

    // ---- Metropolis test (proposal.py:253-258, :357-362; chain.py:112), see k_mh_steps ----
    bool acc;
    {
      const double delta = is_pcn ? ll_n - ll : post_n - (lp + ll);
      if (has_logu && (fabs(lu_s - delta) > 1e-9 || delta != delta)) {
        acc = (post_n == post_n) && (lu_s < delta);
      } else {
        acc = accept_exact(u_s, delta, post_n);
      }
    }
    if (acc) {
      lp = lp_n;
      ll = ll_n;
    }
    nacc += acc ? 1 : 0;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) cur[kk] = acc ? th[kk] : cur[kk];

    acc_prev = acc;
#ifdef TDA_EXP_FETCH_LATE
    if (more) frag_fetch(s + 1);
#endif
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

  if (wave == 0) {
    double* __restrict__ row = a.theta + gcl * DPAD + hi;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) row[4 * kk] = cur[kk];
    if (lane < 16) {
      a.lp[gcl] = lp;
      a.ll[gcl] = ll;
      if (a.acc_count) a.acc_count[gcl] += nacc;
    }
  }
}


}  // namespace tda
