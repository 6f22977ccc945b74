// Single-level Metropolis-Hastings steps for LINEAR forward models with additive proposals: the step kernel whose matrix work
// does not wait for the previous decision.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tda_kernels_mh.h"

namespace tda {

typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// k_mh_lin: S fused Metropolis-Hastings steps (Chain.sample, tinyDA/chain.py:95-125) for one tile of 16 chains, eight waves,
// for the shape of BASELINE config 2a: a linear model of 256 < m <= 1024 observations, isotropic / diagonal noise, diagonal
// prior, theta' = theta + s * inc (GaussianRandomWalk / AdaptiveMetropolis, proposal.py:249-251).
//
// k_mh_steps runs  decision(s) -> theta' -> LDS -> barrier -> fragment gather -> 16 384 cycles of MFMA -> reduction -> barrier ->
// decision(s + 1)  as one dependent chain: ~4 000 of its ~22 400 cycles per step are latency no second wave can fill, because
// the matrix work of step s + 1 needs theta' and theta' needs decision(s).  For a linear model it does not:
//     A theta' - y = (A theta - y) + A (s inc),
// and the increment is known before the step starts.  So
//   * the residual R = A theta - y of the CURRENT state of every chain is tile state: 16 chains x m doubles = 128 KiB of LDS at
//     m = 1024, each wave's own blocks in the MFMA C/D layout of its lanes (re-derived from theta by a direct product at every
//     launch, so rounding cannot accumulate beyond one block of steps);
//   * G_s = A (s inc_s) is what the matrix cores compute, 16 MFMAs per 16-observation block as before -- but the burst of step
//     s + 1 is issued BEFORE the barrier of step s and runs while step s is reduced and decided;
//   * the epilogue of step s adds: R' = R + G_s, sse = sum w R'^2.  Whether R becomes R' is known one step later, so G_s stays
//     in registers until the epilogue of step s + 1 applies it (R += G_s for the chains that accepted) on its way through R:
//     two accumulator sets alternate (2 x 64 registers per wave at m = 1024);
//   * the burst runs k-outermost over groups of four blocks: for each pair of k-steps the operator fragments of the group are
//     fetched (next pair in flight) and every accumulator of the group receives one MFMA -- four independent chains instead
//     of one chain of 16 per block, and the increment fragment is two registers instead of 32.
// The fp64 matrix instruction and fp64 vector arithmetic share the SIMD's fp64 lanes, so per SIMD and
// step the two waves' bursts (2 x 8 192 cycles) and epilogues (2 x ~1 000) add up; everything else hides under the partner's
// burst.
// Same StepArgs, records and RNG contract as k_mh_steps<DPAD, 8>; log-densities agree with it to rounding (linear update,
// summation order of the prior), decisions are the same.
// ------------------------------------------------------------------------------------------------
template <int DPAD, int NB>
__host__ __device__ constexpr int mhlin_lds_doubles(bool diag, int m_pad) {
  return 8 * NB * 256 + 3 * 64 * (DPAD / 4 + 2) + 2 * 16 * 8 + 2 * 16 + 2 * DPAD + (diag ? m_pad : 0);
}

// NB = 16-row operator blocks per wave (4: m <= 512, 8: m <= 1024); DIAG: diagonal noise.  SKIP (measurements only; the library
// instantiates 0): 1 = no residual traffic in the epilogue, 2 = no bursts.
template <int DPAD, int NB, bool DIAG, int SKIP = 0>
__global__ void __launch_bounds__(512, 2) k_mh_lin(const StepArgs a) {
  constexpr int NW = 8, NT = 64 * NW, TPC = 4 * NW;
  constexpr int KS = DPAD / 4, K2 = DPAD / 8, LDP = DPAD + 2, RSX = KS + 2;
  constexpr int EPT = DPAD / TPC;
  static_assert(DPAD % TPC == 0 && EPT == 2, "a thread keeps two consecutive parameters (one 16-byte access)");
  static_assert(16 * LDP <= 3 * 64 * RSX, "the state tile of the direct evaluation aliases the increment tiles");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* s_R = smem;                       // [NW][NB][2][64][2]: residuals of the current states, C/D layout per wave and block
  double* s_inc = s_R + NW * NB * 256;      // [3][64][RSX] scaled increments in fragment order: the tile of step s is read until
                                            // the second half of its burst (behind the barrier of step s - 1), so three rotate
  double* s_prop = s_inc;                   // [16][LDP] states in natural order (direct evaluation at launch only)
  double* s_red = s_inc + 3 * 64 * RSX;     // [2][NW][16]
  double* s_pri = s_red + 2 * 16 * NW;      // [2][16] prior quadratic form of theta'
  double* s_pm = s_pri + 2 * 16;
  double* s_pinv = s_pm + DPAD;
  double* s_w = s_pinv + DPAD;              // [m_pad] (diagonal noise)

  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = blockIdx.x;
  const int c = tid / TPC, q_ = tid % TPC;
  const int lc = lane & 15, hi = lane >> 4;
  const int gct = tile * 16 + c;   // thread-mapped chain (the two parameters 2 q_, 2 q_ + 1 of it)
  const int gcl = tile * 16 + lc;  // lane-mapped chain (densities, decision)

  if (DIAG)
    for (int i = tid; i < a.lv.m_pad; i += NT) s_w[i] = a.lv.w[i];
  for (int i = tid; i < DPAD; i += NT) {
    s_pm[i] = a.pr.mean[i];
    s_pinv[i] = a.pr.pinv[i];
  }
  const bool prior_std = a.pr.kind == PRIOR_STANDARD;

  // Everything a step reads or writes in global memory goes through buffer descriptors: a 32-bit per-thread offset that never
  // changes plus a scalar offset per step, instead of a 64-bit address per array and thread (the kernel needs the registers:
  // a single spilled value reloaded inside the step loop makes the wave wait for ALL its outstanding fragment loads)
  auto rsrc_of = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, 0x7fffffff, 0x00020000); };
  const __amdgpu_buffer_rsrc_t r_inc = rsrc_of(a.inc), r_u = rsrc_of(a.u), r_lu = rsrc_of(a.logu);  // (log u: required by the launcher)
  const __amdgpu_buffer_rsrc_t r_par = rsrc_of(a.rec_params ? (const void*)a.rec_params : (const void*)a.u);
  const __amdgpu_buffer_rsrc_t r_sta = rsrc_of(a.rec_stats ? (const void*)a.rec_stats : (const void*)a.u);
  const __amdgpu_buffer_rsrc_t r_acc = rsrc_of(a.rec_acc ? (const void*)a.rec_acc : (const void*)a.u);
  const bool want_par = a.rec_params != nullptr, want_sta = a.rec_stats != nullptr,
             want_acc = a.rec_acc != nullptr;
  const int vo_inc = (gct * DPAD + q_ * EPT) * 8, so_inc = (int)a.NP * DPAD * 8;  // per step: one [NP][DPAD] slab
  const int vo_u = gcl * 8, so_u = (int)a.NP * 8;
  const int vo_par = (gct * a.d + q_ * EPT) * 8, so_par = (int)a.N * a.d * 8;
  const bool par0 = gct < a.N && q_ * EPT < a.d, par1 = gct < a.N && q_ * EPT + 1 < a.d;
  const int vo_sta = gcl * 24, so_sta = (int)a.N * 24, so_acc = (int)a.N;
  const bool rec_lane = wave == 0 && lane < 16 && gcl < a.N;

  double cur[EPT], prp[EPT];
  {
    const double2 t = *reinterpret_cast<const double2*>(a.theta + (size_t)gct * DPAD + q_ * EPT);
    cur[0] = t.x;
    cur[1] = t.y;
  }
  double lp = a.lp[gcl], ll = a.ll[gcl];
  const double scal_t = a.scaling[gct];
  int nacc = 0;
  const int ncb = a.lv.ncb;
  bool has_b[NB];
  int cbi[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    has_b[i] = wave + i * NW < ncb;
    cbi[i] = has_b[i] ? wave + i * NW : ncb - 1;
  }

  // where this thread's two elements of an increment go in the fragment-ordered tile: row = fragment lane (dim & 3) * 16 + chain
  int st_dst[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int j = q_ * EPT + e;
    st_dst[e] = ((j & 3) * 16 + c) * RSX + (j >> 2);
  }
  auto inc_load = [&](int s) {  // raw increment of step s (no use here, so no wait here)
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r_inc, vo_inc, s * so_inc, 0);
    return *reinterpret_cast<const double2*>(&v);
  };
  auto inc_store = [&](int s, const double2 x) {
    double* __restrict__ dst = s_inc + (s % 3) * 64 * RSX;
    dst[st_dst[0]] = scal_t * x.x;
    dst[st_dst[1]] = scal_t * x.y;
  };

  // ---- operator fragments: k-pair k2 of block cb sits at cb * K2 * 1024 + k2 * 1024 + lane * 16 bytes of the packed operator ----
  const FragSrc src = frag_src(a.lv.Apk, lane);
  // The burst runs in two halves of HB = NB / 2 blocks (the barrier and the decision of the step sit between them), k-pairs
  // outermost inside a half: HB independent accumulator chains, two fragment sets of HB double2 (the next k-pair is requested
  // before the current one is used; a third set does not fit beside the two accumulator sets)
  constexpr int HB = NB / 2;
  static_assert(NB % 2 == 0 && K2 % 2 == 0, "two halves; k-pairs two at a time");
  auto load_pair = [&](int g, int k2, double2 (&f)[HB]) {
#pragma unroll
    for (int i = 0; i < HB; ++i) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(src.rsrc, src.lane_off, (cbi[g * HB + i] * K2 + k2) * 1024, 0);
      f[i] = *reinterpret_cast<const double2*>(&v);
    }
  };
  auto mfma_pair = [&](const double2 (&f)[HB], const double2 b, double4_t (&G)[NB], int g) {
#pragma unroll
    for (int i = 0; i < HB; ++i) G[g * HB + i] = mfma_f64(f[i].x, b.x, G[g * HB + i]);
#pragma unroll
    for (int i = 0; i < HB; ++i) G[g * HB + i] = mfma_f64(f[i].y, b.y, G[g * HB + i]);
  };
  // G[half g] = A x for that half of this wave's blocks, x as double2 fragments per k-pair from `row` (LDS)
  auto burst_half = [&](const double2* __restrict__ row, double4_t (&G)[NB], int g) {
    double2 f0[HB], f1[HB];
    load_pair(g, 0, f0);
#pragma unroll
    for (int i = 0; i < HB; ++i) G[g * HB + i] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k2 = 0; k2 < K2; k2 += 2) {
      load_pair(g, k2 + 1, f1);
      mfma_pair(f0, row[k2], G, g);
      if (k2 + 2 < K2) load_pair(g, k2 + 2, f0);
      mfma_pair(f1, row[k2 + 1], G, g);
    }
  };
  double* const r_base = s_R + ((size_t)wave * NB * 2 * 64 + lane) * 2;  // slot (i, rh) = r_base + (2 i + rh) * 128

  // ---- R = A theta - y for the states the launch starts from (a direct product: the same burst with theta as the operand) ----
  double4_t Ga[NB], Gb[NB];
  s_prop[c * LDP + q_ * EPT] = cur[0];
  s_prop[c * LDP + q_ * EPT + 1] = cur[1];
  __syncthreads();  // state tile, prior constants, weights
  {
    // the rows of s_prop in fragment order: element k-step kk of chain lc = s_prop[lc][4 kk + hi]
    double2 xrow[K2];
#pragma unroll
    for (int k = 0; k < K2; ++k) xrow[k] = double2{s_prop[lc * LDP + 8 * k + hi], s_prop[lc * LDP + 8 * k + 4 + hi]};
    burst_half(xrow, Ga, 0);
    burst_half(xrow, Ga, 1);
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        const int ob = cbi[i] * 16 + hi + 8 * rh;
        *reinterpret_cast<double2*>(r_base + (2 * i + rh) * 128) = double2{Ga[i][2 * rh] - a.lv.ytil[ob], Ga[i][2 * rh + 1] - a.lv.ytil[ob + 4]};
      }
  }
  __syncthreads();  // the state tile is free again: the increment tiles live there from here on

  // ---- pipeline prologue: increments of steps 0 and 1 staged, A (s inc_0) issued, increment of step 2 in flight ----
  double2 xc = double2{0.0, 0.0}, xd = double2{0.0, 0.0};  // raw increments of steps s + 2 (arrived, staged this step) and s + 3
  inc_store(0, inc_load(0));
  if (a.S > 1) inc_store(1, inc_load(1));
  if (a.S > 2) xc = inc_load(2);
  double lunext;  // log u of the next decision; u itself is only fetched within 1e-9 of the knife edge
  {
    const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(r_lu, vo_u, 0, 0);
    lunext = *reinterpret_cast<const double*>(&w);
  }
  __syncthreads();
  auto inc_row = [&](int s) { return reinterpret_cast<const double2*>(s_inc + (s % 3) * 64 * RSX + lane * RSX); };
  burst_half(inc_row(0), Ga, 0);
  burst_half(inc_row(0), Ga, 1);

  // record of the last decided step (state after the decision, its densities, the flag): off the chain decision -> next proposal
  bool rec_pending = false, acc_prev = false;
  int rec_row = 0;
  auto flush_record = [&]() {
    if (!rec_pending) return;
    rec_pending = false;
    if (want_par) {
      if (par0) __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2_t*>(&cur[0]), r_par, vo_par, rec_row * so_par, 0);
      if (par1) __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2_t*>(&cur[1]), r_par, vo_par + 8, rec_row * so_par, 0);
    }
    if (rec_lane) {
      if (want_sta) {
        const double post = lp + ll;
        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2_t*>(&lp), r_sta, vo_sta, rec_row * so_sta, 0);
        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2_t*>(&ll), r_sta, vo_sta + 8, rec_row * so_sta, 0);
        __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const u32x2_t*>(&post), r_sta, vo_sta + 16, rec_row * so_sta, 0);
      }
      if (want_acc) __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(acc_prev ? 1 : 0), r_acc, gcl, rec_row * so_acc, 0);
    }
  };

  // one step; Gc = A (s inc_s) (issued a step ago), Gp = the same of step s - 1, applied here to the chains that accepted it and
  // then overwritten by the burst of step s + 1
  auto one_step = [&](const int s, double4_t (&Gc)[NB], double4_t (&Gp)[NB]) {
    // theta' and its prior (thread-mapped: the 32 threads of a chain are consecutive lanes of one wave); the scaled increment
    // comes back from the tile it was staged into two steps ago
    double pp = 0.0;
    {
      const double* __restrict__ tile_s = s_inc + (s % 3) * 64 * RSX;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        prp[e] = cur[e] + tile_s[st_dst[e]];
        if (prior_std) {
          pp += prp[e] * prp[e];
        } else {
          const double dv = prp[e] - s_pm[q_ * EPT + e];
          pp += dv * dv * s_pinv[q_ * EPT + e];
          if (a.pr.lo && (prp[e] < a.pr.lo[q_ * EPT + e] || prp[e] > a.pr.hi[q_ * EPT + e])) pp = INFINITY;
        }
      }
    }
    // epilogue: R of the current state (the previous step's move applied on the way), residual of theta', weighted squares;
    // two blocks at a time (all of R requested at once would hold 64 registers beside the two accumulator sets)
    double sse = 0.0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
      for (int rh = 0; rh < 2; ++rh) {
        double2* slot = reinterpret_cast<double2*>(r_base + (2 * i + rh) * 128);
        const double2 rv = (SKIP & 1) ? double2{0.1, 0.2} : *slot;
        const double r0 = acc_prev ? rv.x + Gp[i][2 * rh] : rv.x;
        const double r1 = acc_prev ? rv.y + Gp[i][2 * rh + 1] : rv.y;
        if (!(SKIP & 1)) *slot = double2{r0, r1};
        const double p0 = r0 + Gc[i][2 * rh], p1 = r1 + Gc[i][2 * rh + 1];
        double q0 = p0 * p0, q1 = p1 * p1;
        if (DIAG) {
          const int ob = cbi[i] * 16 + hi + 8 * rh;
          q0 *= s_w[ob];
          q1 *= s_w[ob + 4];
        }
        sse += has_b[i] ? q0 + q1 : 0.0;
      }
      if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);
    }
    sse = sum_rows(sse);
    if (lane < 16) s_red[(s & 1) * 16 * NW + wave * 16 + lane] = sse;
    // the next step's products do not depend on this step's decision: the first half is issued now ...
    __builtin_amdgcn_sched_barrier(0);
    const bool more = !(SKIP & 2) && s + 1 < a.S;
    if (more) burst_half(inc_row(s + 1), Gp, 0);
    __builtin_amdgcn_sched_barrier(0);
    flush_record();  // (step s - 1)
    if (s + 3 < a.S) xd = inc_load(s + 3);  // staged one step from now
    double lu_nx = 0.0;
    if (s + 1 < a.S) {
      const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(r_lu, vo_u, (s + 1) * so_u, 0);
      lu_nx = *reinterpret_cast<const double*>(&w);
    }
    pp = sum_half_wave(pp);
    if (q_ == 0) s_pri[(s & 1) * 16 + c] = pp;
    if (s + 2 < a.S) inc_store(s + 2, xc);  // loaded a step ago; its tile last held step s - 1's increments, read for the last
                                            // time behind the barrier of step s - 2
    __syncthreads();

    // ... the second half behind the barrier, in a different order on the two waves of a SIMD: waves 0-3 decide first and
    // multiply then, waves 4-7 multiply first -- a wave cannot leave its own burst (MFMA issue blocks), so the latency chain
    // of one wave's decision (LDS, the accept test, the shuffle) runs under the OTHER wave's matrix work
    bool acc = false;
    auto decide = [&]() {
      double part[NW];  // all partial sums requested before the first is used: one LDS latency, not eight
#pragma unroll
      for (int w = 0; w < NW; ++w) part[w] = s_red[(s & 1) * 16 * NW + w * 16 + lc];
      const double maha = s_pri[(s & 1) * 16 + lc];
      __builtin_amdgcn_sched_barrier(0);
      double tot = part[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) tot += part[w];
      const double ll_n = DIAG ? -0.5 * tot : -0.5 * tot / a.lv.var;
      const double lp_n = -0.5 * (a.pr.logconst + maha);
      const double post_n = lp_n + ll_n;
      const double delta = post_n - (lp + ll);
      // (the exact form inline: a function call here would confine everything that lives across it -- both accumulator sets --
      // to the callee-saved half of the register file)
      if (fabs(lunext - delta) > 1e-9 || delta != delta) {
        acc = (post_n == post_n) && (lunext < delta);
      } else {
        const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(r_u, vo_u, s * so_u, 0);
        double alpha = exp(delta);
        if (post_n != post_n) alpha = 0.0;
        acc = *reinterpret_cast<const double*>(&v) < alpha;
      }
      if (acc) {
        lp = lp_n;
        ll = ll_n;
      }
      nacc += acc ? 1 : 0;
      const int accf = __shfl(acc ? 1 : 0, c);
#pragma unroll
      for (int e = 0; e < EPT; ++e) cur[e] = accf ? prp[e] : cur[e];
    };
    if (wave < NW / 2) {
      decide();
      __builtin_amdgcn_sched_barrier(0);
      if (more) burst_half(inc_row(s + 1), Gp, 1);
    } else {
      if (more) burst_half(inc_row(s + 1), Gp, 1);
      __builtin_amdgcn_sched_barrier(0);
      decide();
    }
    lunext = lu_nx;
    acc_prev = acc;
    rec_pending = true;  // the record of this step is written behind the next step's products
    rec_row = s;
    xc = xd;
  };

  {
    int s = 0;
    for (; s + 1 < a.S; s += 2) {
      one_step(s, Ga, Gb);
      one_step(s + 1, Gb, Ga);
    }
    if (s < a.S) one_step(s, Ga, Gb);
  }
  flush_record();

  *reinterpret_cast<double2*>(a.theta + (size_t)gct * DPAD + q_ * EPT) = double2{cur[0], cur[1]};
  if (wave == 0 && lane < 16) {
    a.lp[gcl] = lp;
    a.ll[gcl] = ll;
    if (a.acc_count) a.acc_count[gcl] += nacc;
  }
}

}  // namespace tda
