// HIP kernels of the many-chain MH engine for gfx950 (MI355X).  No CUDA / multi-backend paths.
//
// Pipeline per block of S <= period steps (the proposal distribution is constant inside a block, because tinyDA's
// proposals only change at adapt-count multiples of `period`, proposal.py:234,509):
//
//   k_rng        wave per (chain, 16 steps): Philox normals z_s as MFMA A fragments, accept uniforms u_s and log u_s.
//                State independent, so block b+1's draws run on a second stream UNDER block b's k_mh_steps
//                (64 registers: shares the SIMDs with the step kernel's 2 x 224)
//   k_apply      wave per chain: INC = Z L^T on the matrix cores (L = chol C as B fragments in registers)  -> HBM [S][N][D]
//                (k_propose fuses k_rng + k_apply for recorded variates and the multi-level / DREAM paths)
//   k_mh_steps   workgroup = 16 chains x 8 waves (two per SIMD), S fused steps:
//                theta' = theta + scaling * inc_s  (pCN: sqrt(1-b^2) theta + b inc_s; independence: mu_q + inc_s),
//                F = A theta' on fp64 MFMA (v_mfma_f64_16x16x4), observation blocks dealt over the waves, A fragments
//                streamed from L2 with buffer loads, residual + weighted SSE fused in the MFMA epilogue, prior,
//                log alpha, accept, coalesced record write (4-wave variant: dense noise, residual tile in LDS)
//   k_adapt      wave per chain: RecursiveSampleMoments catch-up over the S recorded states (utils.py:113-122) in the
//                reference's exact arithmetic, Sigma as 16x16 tiles in registers; global scaling adaptation at period
//                boundaries.  k_adapt_block: the same covariance as one rank-S SYRK on the matrix cores (opt-in)
//   k_chol       wave per chain: C <- Sigma swap, register-resident Cholesky (only at period boundaries with t >= t0)
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
  double logconst;  // d*log(2 pi) + log det cov  (JointPrior: + 2 log(width) per uniform component)
  const double* lo;  // JointPrior with uniform components (distributions.py:8-100): support [lo_j, hi_j] per parameter
  const double* hi;  // (+-inf for normal components), null when unbounded; outside the support log-prior = -inf
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
  // independence sampler (prop_kind 4): theta' = q_mean + inc, alpha = exp(post' - post + lq - qz)
  const double* q_mean;  // [DPAD]
  const double* qz;      // [S][NP] log q(theta') up to q's constant: -|z|^2 / 2
  double* lq;            // [NP]    the same for the current state
  // operator-weighted pCN (prop_kind 5): theta' = S theta + inc
  const double* SopT;    // [DPAD][DPAD] transposed state operator: SopT[j][i] = S[i][j]
  // MALA (prop_kind 6): grad log post = cvec - H theta with H (symmetric) in SopT; theta' = theta + s^2/2 grad + s inc
  const double* cvec;    // [DPAD]
  double* grad;          // [NP][DPAD] gradient at the current state (chain state, like theta)
  // records, layout of tda_outputs (may be null)
  double* rec_params;
  double* rec_stats;
  uint8_t* rec_acc;
  long long* trace;  // debug builds (-DTDA_STEP_TRACE, tools/steps_microbench.hip): [S][waves][8] cycle stamps of tile 0, then 4 clock stamps
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
  double* qz;             // [S][NP] -|z_s|^2 / 2 for the independence sampler (may be null)
  const double* z_replay; // [.][N][d] at step0 (may be null)
  const double* u_replay; // [.][N]
  double* z_export;       // same layout (may be null)
  double* u_export;
  // 65 .. 128 parameters only (tda_kernels_wide.h: fragments of the normals, then k_wide_apply): a fragment buffer, the diagonal
  // tiles, the current-buffer selector, chains per factor buffer
  double* zf_tmp;
  const double* ud;
  const int32_t* sel;
  int64_t NPf;
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
  int block_moments;  // AM: 0 = the reference's elementwise covariance recursion, 1 = one rank-S update per block
  int period;
  double gamma_pow;  // gamma ** -k  (proposal.py:240)
  double sd, eps;
  double alpha_star;  // target acceptance of the scaling adaptation: 0.24 (proposal.py:169), MALA 0.57 (:899)
  const double* rec_params;  // [S][N][d] states recorded by k_mh_steps
  double* am_mu;             // [NP][DPAD]
  double* am_sigma;          // [NP][am_tiles][4][64]: lower 16x16 tiles in MFMA C/D layout (see k_adapt)
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

// Sum over the 32 consecutive lanes [32 k, 32 k + 32) that hold one chain in the thread-mapped phases of the 8-wave tile,
// result in every lane: one v_permlane16_swap stage (rows 16 apart) and four DPP row rotations inside the 16-lane row,
// instead of five dependent ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum_half_wave(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const uint2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const uint2_t b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  double s = __hiloint2double(b.x, a.x) + __hiloint2double(b.y, a.y);  // v + v(lane ^ 16), as in sum_rows
  s += dpp_move<0x128>(s);  // row_ror:8
  s += dpp_move<0x124>(s);  // row_ror:4
  s += dpp_move<0x122>(s);  // row_ror:2
  s += dpp_move<0x121>(s);  // row_ror:1
  return s;
}

__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Fragment source for buffer loads: resource descriptor (4 SGPRs) + scalar block offset + one 32-bit lane offset.
// With per-lane 64-bit pointers hipcc keeps an address pair per block half alive across the step loop (tens of
// registers); the descriptor form needs one VGPR.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
struct FragSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  int lane_off;
};
__device__ __forceinline__ FragSrc frag_src(const double* packed, int lane) {
  return FragSrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(packed), 0, 0x7fffffff, 0x00020000), lane * 16};
}
template <int DPAD>
__device__ __forceinline__ void frag_load_buf(const FragSrc& src, int cb, double2 (&f)[DPAD / 8]) {
  const int soff = cb * (DPAD / 8) * 1024;
#pragma unroll
  for (int k = 0; k < DPAD / 8; ++k) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(src.rsrc, src.lane_off + (k & 3) * 1024, soff + (k >> 2) * 4096, 0);
    f[k] = *reinterpret_cast<const double2*>(&v);
  }
}

// MFMA A-operand fragments of one 16-row block: K2 16-byte buffer loads per lane, unconditional (the block index is
// clamped, out-of-range blocks are simply not accumulated) so that hipcc emits counted vmcnt waits instead of one
// branch per load.
template <int DPAD>
__device__ __forceinline__ void frag_load(const FragSrc& src, int cb, int ncb, double2 (&f)[DPAD / 8]) {
  frag_load_buf<DPAD>(src, cb < ncb ? cb : ncb - 1, f);
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
  const FragSrc base = frag_src(Apk, lane);
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

// The same with BOTH fragment buffers supplied by the caller (k_da_steps lends the registers of its coarse operator), and --
// FRESH -- with the chains' parameters (the B operand) read from LDS for every block instead of 32 registers held across the
// sum: the level actions of the multi-level kernels run at the kernel's register peak.  The parameters are staged in
// fragment order, th_frag = tile + lane * (DPAD / 4 + 2): one 16-byte read serves two MFMAs (k-slices 2 k, 2 k + 1).
template <int DPAD, int MODE, bool FRESH>
__device__ __forceinline__ double block_sse_frag(const double2 (&f)[DPAD / 8], const double2* th_frag, const double* __restrict__ s_y,
                                                 double* __restrict__ s_w, int cb, int hi) {
  double4_t a0 = {0.0, 0.0, 0.0, 0.0};
  // th_frag points into LDS.  FRESH: a new read per block (hoisted out of the block loop these are the 32 registers again), forced
  // by passing the pointer through an empty asm -- as an LDS (address space 3, 32-bit) pointer.  Round 5: rounds 3-4 passed the
  // GENERIC pointer through the asm; the compiler no longer knew it was LDS and read the states with flat_load_dwordx4, each
  // followed by `s_waitcnt vmcnt(0) lgkmcnt(0)` (a flat access may resolve to either memory): eight full drains of the fragment
  // prefetch per observation block in every level action of the three-level instances (tools/loop_audit.py found them).
  typedef double lds_d2 __attribute__((ext_vector_type(2)));  // (a builtin vector: HIP's double2 is a class, whose copy cannot bind an LDS reference)
  typedef const lds_d2 __attribute__((address_space(3)))* lds_frag_ptr;
  lds_frag_ptr th_lds = (lds_frag_ptr)th_frag;
  if (FRESH) asm volatile("" : "+v"(th_lds));
#pragma unroll
  for (int k = 0; k < DPAD / 8; ++k) {
    const lds_d2 b = th_lds[k];
    a0 = mfma_f64(f[k].x, b.x, a0);
    a0 = mfma_f64(f[k].y, b.y, a0);
  }
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
template <int DPAD, int MODE, int NW, bool FRESH>
__device__ __forceinline__ double level_sse_frag(const double* __restrict__ Apk, int ncb, const double* __restrict__ s_y,
                                                 double* __restrict__ s_w, const double2* th_frag, int wave, int lane,
                                                 double2 (&fa)[DPAD / 8], double2 (&fb)[DPAD / 8]) {
  const int hi = lane >> 4;
  const FragSrc src = frag_src(Apk, lane);
  const int first = wave < ncb ? wave : ncb - 1;
  double sse = 0.0;
  for (int cb = wave; cb < ncb; cb += 2 * NW) {  // fa holds block cb
    frag_load_buf<DPAD>(src, cb + NW < ncb ? cb + NW : first, fb);
    __builtin_amdgcn_sched_barrier(0);
    sse += block_sse_frag<DPAD, MODE, FRESH>(fa, th_frag, s_y, s_w, cb, hi);
    __builtin_amdgcn_sched_barrier(0);
    if (cb + NW < ncb) {
      frag_load_buf<DPAD>(src, cb + 2 * NW < ncb ? cb + 2 * NW : first, fa);
      __builtin_amdgcn_sched_barrier(0);
      sse += block_sse_frag<DPAD, MODE, FRESH>(fb, th_frag, s_y, s_w, cb + NW, hi);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  return sse;  // (the buffers hold nothing the caller may rely on)
}

template <int DPAD, int MODE, int NW>
__device__ __forceinline__ double level_sse_single(const double* __restrict__ Apk, int ncb,
                                                   const double* __restrict__ s_y, double* __restrict__ s_w,
                                                   const double (&th)[DPAD / 4], int wave, int lane,
                                                   double2 (&fa)[DPAD / 8]) {
  const int hi = lane >> 4;
  const FragSrc src = frag_src(Apk, lane);
  const int first = wave < ncb ? wave : ncb - 1;
  double sse = 0.0;
  double2 fb[DPAD / 8];
  bool next_in_fb = false;  // where the next step's first block ended up
  for (int cb = wave; cb < ncb; cb += 2 * NW) {
    frag_load_buf<DPAD>(src, cb + NW < ncb ? cb + NW : first, fb);
    __builtin_amdgcn_sched_barrier(0);
    sse += block_sse<DPAD, MODE>(fa, th, s_y, s_w, cb, hi);
    __builtin_amdgcn_sched_barrier(0);
    if (cb + NW < ncb) {
      frag_load_buf<DPAD>(src, cb + 2 * NW < ncb ? cb + 2 * NW : first, fa);
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

// u < exp(delta) (chain.py:112), out of line: it is only needed within 1e-9 of the knife edge, and inlining exp's
// polynomial costs the hot loop 16 registers -- with it the 8-wave tile allocates 2 x 240 of a SIMD's 512 registers,
// without it 2 x 224, which leaves room for a 64-register kernel (k_rng) to share the SIMD
__device__ __attribute__((noinline)) bool accept_exact(double u, double delta, double post_n) {
  double alpha = exp(delta);
  if (post_n != post_n) alpha = 0.0;
  return u < alpha;
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
// IND = IndependenceSampler proposals (a template parameter: as a run-time flag it costs the random-walk path 16 registers
// and 3 %)
// PX = 1: OperatorWeightedCrankNicolson proposals (proposal.py:592-598): theta' = S theta + inc needs the whole current
// state of a chain in every thread of its group, so the variant keeps a current-state tile and S^T in LDS and pays one
// more barrier per step; a template parameter for the same reason as IND.
// PX = 2: MALA (proposal.py:945-984).  The target is linear-Gaussian, so grad log post(theta) = c - H theta with the
// d x d matrix H = Sigma_prior^-1 + A^T Sigma_e^-1 A built once by the host: the gradient at the proposal is one more
// d x d product from LDS (the operator sits where S^T does) instead of a second pass over the observations; the
// gradient at the current state is chain state.  The two transition densities reach the lane-mapped accept test
// through 32 LDS slots.
// PX = 3: OperatorWeightedCrankNicolson with PER-CHAIN operators (adaptive scaling, proposal.py:582-590).  For a symmetric
// B = V diag(lambda) V^T the operators of a chain with scaling s are functions of the spectrum,
//   sqrtm(I - s B) = V diag(sqrt(1 - s lambda)) V^T,   sqrtm(s B) = V diag(sqrt(s lambda)) V^T   (real parts: negative arguments -> 0),
// so the chain state is also carried in eigen-coordinates e = V^T theta: e' = a . e + b . w with w = V^T chol(C_prior) z (the
// increment block, shared factor) and theta' = V e' -- ONE d x d product per step from LDS, like PX = 1, and no matrix square
// root per chain and period.  SopT = [V row-major | V^T row-major], cvec = lambda.
template <int DPAD, int NW, bool IND = false, int PX = 0>
__global__ void __launch_bounds__(64 * NW, NW / 4) k_mh_steps(const StepArgs a) {
  constexpr bool OW = PX == 1, MA = PX == 2, OS = PX == 3;
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
  double* s_cur = s_R + (dense ? 16 * RS : 0);  // OW: current states [16][LDP]; MALA: [0..32) the transition densities
  double* s_S = s_cur + 16 * LDP;               // OW: S^T [DPAD][DPAD]; MALA: H; OS: V [DPAD][DPAD], then V^T [DPAD][DPAD]

  // the step kernel is the critical path: kernels that share its SIMDs (k_rng on the second stream) only get the
  // issue slots it leaves empty
  __builtin_amdgcn_s_setprio(3);
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
  const bool is_pcn = a.prop_kind == 1 || OW || OS;  // OperatorWeightedCrankNicolson inherits pCN's likelihood-ratio acceptance
  constexpr bool is_ind = IND;  // IndependenceSampler (proposal.py:65-129)
  double qm[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) qm[e] = (is_ind && active) ? a.q_mean[q * EPT + e] : 0.0;
  double lq = is_ind ? a.lq[gcl] : 0.0, qznext = 0.0;
  const FragSrc fbase = frag_src(a.lv.Apk, lane);
  const FragSrc pbase = frag_src(a.pr.Wpk, lane);
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
    if (is_ind) qznext = a.qz[gcl];
  }
  double gcur[EPT], gprp[EPT], cv[EPT];  // MALA: gradient at the current state / at the proposal, constant term
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    gcur[e] = (MA && active) ? a.grad[gct * DPAD + q * EPT + e] : 0.0;
    cv[e] = (MA && active) ? a.cvec[q * EPT + e] : 0.0;
    gprp[e] = 0.0;
  }
  const double sig_t = scal_t, half_s2 = 0.5 * sig_t * sig_t;  // thread-mapped chain: 0.5 * scaling**2 (proposal.py:953)
  const double sig_l = MA ? a.scaling[gcl] : 1.0;              // lane-mapped chain
  if constexpr (OW || MA || OS) {
    for (int i = tid; i < (OS ? 2 : 1) * DPAD * DPAD; i += NT) s_S[i] = a.SopT[i];
    if ((OW || OS) && active) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) s_cur[c * LDP + q * EPT + e] = cur[e];
    }
  }
  __syncthreads();
  double ecur[EPT], eprp[EPT], oa[EPT], ob[EPT];  // OS: state / proposal in eigen-coordinates, this chain's jump coefficients
#pragma unroll
  for (int e = 0; e < EPT; ++e) ecur[e] = eprp[e] = oa[e] = ob[e] = 0.0;
  if constexpr (OS) {
    if (active) {
#pragma unroll 4
      for (int j = 0; j < DPAD; ++j) {  // e = V^T theta: e[i] = sum_j V[j][i] theta[j] (consecutive threads, consecutive i)
        const double tj = s_cur[c * LDP + j];
#pragma unroll
        for (int e = 0; e < EPT; ++e) ecur[e] = fma(s_S[j * DPAD + q * EPT + e], tj, ecur[e]);
      }
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const double sl = scal_t * a.cvec[q * EPT + e];
        oa[e] = sqrt(fmax(0.0, 1.0 - sl));
        ob[e] = sqrt(fmax(0.0, sl));
      }
    }
    __syncthreads();  // s_cur is the exchange tile of the step loop from here on
  }

  if constexpr (!PAIRS) frag_load_buf<DPAD>(frag_src(a.lv.Apk, lane), wave < a.lv.ncb ? wave : a.lv.ncb - 1, f0);  // later steps: prefetched by the previous step
  // cycle stamps for tools/steps_microbench.hip: compiled in only on request, because even a never-taken s_memtime
  // makes hipcc fall back to lgkmcnt(0) waits in the hot loop
#ifdef TDA_STEP_TRACE
  const bool tracing = a.trace != nullptr && blockIdx.x == 0 && lane == 0;
#define TDA_STAMP(i) \
  if (tracing) a.trace[((size_t)s * NW + wave) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime()
  // the clock the chip holds under this kernel: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime) over the step loop
  if (tracing && wave == 0) {
    a.trace[(size_t)a.S * NW * 8 + 0] = (long long)__builtin_amdgcn_s_memtime();
    a.trace[(size_t)a.S * NW * 8 + 1] = (long long)__builtin_amdgcn_s_memrealtime();
  }
#else
#define TDA_STAMP(i)
#endif
  for (int s = 0; s < a.S; ++s) {
    TDA_STAMP(0);
    // first fragment block(s) of this step: independent of theta', issued ahead of the barrier
    if constexpr (PAIRS) {
      frag_load<DPAD>(fbase, wave, a.lv.ncb, f0);
      frag_load<DPAD>(fbase, wave + NW, a.lv.ncb, f1);
    }
    // ---- proposal: theta' (proposal.py:249-251 / :351-355 / :592-598) ----
    if constexpr (OW) {
      if (s > 0) __syncthreads();  // the accepted states of the previous step are in s_cur
      if (active) {
        double st[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) st[e] = 0.0;
        for (int j = 0; j < DPAD; ++j) {  // (S theta)[i] = sum_j S^T[j][i] theta[j]: consecutive threads, consecutive i
          const double tj = s_cur[c * LDP + j];
#pragma unroll
          for (int e = 0; e < EPT; ++e) st[e] = fma(s_S[j * DPAD + q * EPT + e], tj, st[e]);
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          prp[e] = st[e] + xin[e];
          s_prop[c * LDP + q * EPT + e] = prp[e];
        }
      }
    } else if constexpr (OS) {
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          eprp[e] = oa[e] * ecur[e] + ob[e] * xin[e];
          s_cur[c * LDP + q * EPT + e] = eprp[e];
        }
      }
      __syncthreads();
      if (active) {
        double st[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) st[e] = 0.0;
#pragma unroll 4
        for (int i = 0; i < DPAD; ++i) {  // theta'[k] = sum_i V[k][i] e'[i], read from V^T: consecutive threads, consecutive k
          const double ei = s_cur[c * LDP + i];
#pragma unroll
          for (int e = 0; e < EPT; ++e) st[e] = fma(s_S[DPAD * DPAD + i * DPAD + q * EPT + e], ei, st[e]);
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          prp[e] = st[e];
          s_prop[c * LDP + q * EPT + e] = prp[e];
        }
      }
    } else if constexpr (MA) {
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          prp[e] = (cur[e] + half_s2 * gcur[e]) + sig_t * xin[e];  // proposal.py:951-956
          s_prop[c * LDP + q * EPT + e] = prp[e];
        }
      }
    } else if (active) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        if (is_eval) {
          prp[e] = cur[e];
        } else {
          const double sx = scal_t * xin[e];
          prp[e] = is_ind ? qm[e] + xin[e] : (is_pcn ? keep_t * cur[e] + sx : cur[e] + sx);
        }
        s_prop[c * LDP + q * EPT + e] = prp[e];
      }
    }
    const double u = unext, lu = lunext, qzs = qznext;
    if (!is_eval && s + 1 < a.S) {  // next step's increment and uniform fly during the MFMA phase
      if (is_ind) qznext = a.qz[(size_t)(s + 1) * a.NP + gcl];
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) xin[e] = a.inc[((size_t)(s + 1) * a.NP + gct) * DPAD + q * EPT + e];
      }
      unext = a.u[(size_t)(s + 1) * a.NP + gcl];
      if (has_logu) lunext = a.logu[(size_t)(s + 1) * a.NP + gcl];
    }
    TDA_STAMP(1);
    __syncthreads();
    TDA_STAMP(2);

    if constexpr (MA) {
      // gradient at the proposal and the two transition densities (proposal.py:958-984), thread-mapped:
      //   q_x_y = -|theta - theta' - s^2/2 grad'|^2 / (2 s^2),  q_y_x = -|theta' - theta - s^2/2 grad|^2 / (2 s^2)
      double qa = 0.0, qb = 0.0;
      if (active) {
        double hg[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) hg[e] = 0.0;
        for (int j = 0; j < DPAD; ++j) {
          const double tj = s_prop[c * LDP + j];
#pragma unroll
          for (int e = 0; e < EPT; ++e) hg[e] = fma(s_S[j * DPAD + q * EPT + e], tj, hg[e]);
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          gprp[e] = cv[e] - hg[e];
          const double da = (cur[e] - prp[e]) - half_s2 * gprp[e];
          const double db = (prp[e] - cur[e]) - half_s2 * gcur[e];
          qa += da * da;
          qb += db * db;
        }
      }
#pragma unroll
      for (int off = TPC / 2; off >= 1; off >>= 1) {  // the TPC threads of a chain are consecutive lanes of one wave
        qa += __shfl_xor(qa, off);
        qb += __shfl_xor(qb, off);
      }
      if (q == 0) {
        s_cur[c] = qa;
        s_cur[16 + c] = qb;
      }
    }

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
      if (a.pr.lo) {  // uniform components: zero density outside their support (rare path, bounds read through L1)
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          if (th[kk] < a.pr.lo[4 * kk + hi] || th[kk] > a.pr.hi[4 * kk + hi]) p = INFINITY;
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
        frag_load_buf<DPAD>(frag_src(a.pr.Wpk, lane), wave < a.pr.ncb ? wave : a.pr.ncb - 1, p0);
        p = level_sse_single<DPAD, 0, NW>(a.pr.Wpk, a.pr.ncb, s_py, nullptr, th, wave, lane, p0);
      }
      p = sum_rows(p);
      if (lane < 16) s_redp[wave * 16 + lane] = p;
    }

    // ---- forward model + Gaussian log-likelihood (posterior.py:95-108, distributions.py:295-326) ----
    TDA_STAMP(3);
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
    TDA_STAMP(4);
    sse = sum_rows(sse);
    if (lane < 16) s_red[wave * 16 + lane] = sse;
    TDA_STAMP(5);
    __syncthreads();
    TDA_STAMP(6);

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
      double delta = is_pcn ? ll_n - ll : (is_ind ? ((post_n - (lp + ll)) + lq) - qzs : post_n - (lp + ll));
      if constexpr (MA) {
        const double kq = -0.5 / (sig_l * sig_l);
        delta = (delta + kq * s_cur[lc]) - kq * s_cur[16 + lc];  // exp(post' - post + q_x_y - q_y_x)
      }
      if (has_logu && (fabs(lu - delta) > 1e-9 || delta != delta)) {
        acc = (post_n == post_n) && (lu < delta);
      } else {
        acc = accept_exact(u, delta, post_n);
      }
    }
    if (acc) {
      lp = lp_n;
      ll = ll_n;
      lq = qzs;
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
        if constexpr (OS) ecur[e] = accf ? eprp[e] : ecur[e];
        if constexpr (OW) s_cur[c * LDP + q * EPT + e] = cur[e];
        if constexpr (MA) gcur[e] = accf ? gprp[e] : gcur[e];
        const int j = q * EPT + e;
        if (!is_eval && a.rec_params && gct < a.N && j < a.d)
          a.rec_params[((size_t)s * a.N + gct) * a.d + j] = cur[e];
      }
    }
    TDA_STAMP(7);
  }
#undef TDA_STAMP
#ifdef TDA_STEP_TRACE
  if (tracing && wave == 0) {
    a.trace[(size_t)a.S * NW * 8 + 2] = (long long)__builtin_amdgcn_s_memtime();
    a.trace[(size_t)a.S * NW * 8 + 3] = (long long)__builtin_amdgcn_s_memrealtime();
  }
#endif

  if (active) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      a.theta[gct * DPAD + q * EPT + e] = cur[e];
      if constexpr (MA) a.grad[gct * DPAD + q * EPT + e] = gcur[e];
    }
  }
  if (wave == 0 && lane < 16) {
    a.lp[gcl] = lp;
    a.ll[gcl] = ll;
    if (is_ind && !is_eval) a.lq[gcl] = lq;
    if (!is_eval && a.acc_count) a.acc_count[gcl] += nacc;
  }
}

// MALA: gradient of the log-posterior at the initial states, grad = c - H theta (thread = (chain, component))
__global__ void k_mala_grad0(int64_t NP, int DP, const double* __restrict__ H, const double* __restrict__ cvec,
                             const double* __restrict__ theta, double* __restrict__ grad) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= NP * DP) return;
  const int64_t c = t / DP;
  const int i = (int)(t % DP);
  double hg = 0.0;
  for (int j = 0; j < DP; ++j) hg = fma(H[(size_t)j * DP + i], theta[c * DP + j], hg);
  grad[t] = cvec[i] - hg;
}

// ------------------------------------------------------------------------------------------------
// Proposal increments for a block of steps: one wave per chain.
//   np.random.multivariate_normal(0, C) (proposal.py:249-251) as L z with L = chol(C), z from Philox.
// INC[S][d] = Z[S][d] L^T is a per-chain GEMM and runs on the matrix cores, 16 steps at a time: lane (lc, hi) owns
// step lc of the group and draws the Box-Muller pair p = 4 q + hi (dims 2p, 2p + 1, the RNG contract) for
// q = 0 .. d/8 - 1; a 16 x d tile in LDS turns pairs into MFMA A fragments (z[step lc][4 kk + hi]); L sits in
// registers as B fragments (L[16 tj + lc][4 kk + hi]) for the whole block.  The normals are the bound (~240 VALU
// instructions per pair); the products ride on the otherwise idle matrix pipe, and two waves per SIMD overlap one
// wave's MFMAs with the other's RNG.  The earlier form (lane j = row j of L, 64 broadcast FMAs per step from LDS,
// one wave per SIMD) spent 0.73 ns/eval.
// ------------------------------------------------------------------------------------------------
// out of line: inlined eight times into the unrolled pair loop, the polynomial constants of log / sincospi push the
// kernel past the 256 registers that two waves per SIMD allow
__device__ __attribute__((noinline)) double2 normal_pair_call(uint64_t seed, uint32_t chain, uint32_t step, uint32_t p) {
  double z0, z1;
  normal_pair(seed, chain, step, STREAM_PROPOSAL, p, z0, z1);
  return double2{z0, z1};
}

template <int DPAD>
__global__ void __launch_bounds__(64, 2) k_propose(const ProposeArgs a) {
  constexpr int KK = DPAD / 4;                  // k-steps of the 16x16x4 MFMA
  constexpr int QN = DPAD / 8;                  // Box-Muller pairs per lane and 16-step group
  constexpr int TJ = DPAD >= 16 ? DPAD / 16 : 1;  // 16-column output tiles
  constexpr int RS = DPAD + 2;
  __shared__ __attribute__((aligned(16))) double s_z[16 * RS];
  const int lane = threadIdx.x;
  const int lc = lane & 15, hi = lane >> 4;
  const int64_t c = blockIdx.x;
  const bool real_chain = c < a.N;
  const uint32_t gc = (uint32_t)(a.chain_offset + c);

  double Lf[TJ][KK];  // B fragments: B[k = hi][j = lc] of tile tj, k-step kk  <->  L[16 tj + lc][4 kk + hi]
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      Lf[tj][kk] = 16 * tj + lc < DPAD ? a.Lk[(size_t)c * a.L_stride + (size_t)(4 * kk + hi) * DPAD + 16 * tj + lc] : 0.0;

  for (int s0 = 0; s0 < a.S; s0 += 16) {
    const int s = s0 + lc;  // this lane's step
    double4_t acc[TJ];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) acc[tj] = double4_t{0.0, 0.0, 0.0, 0.0};
    double zz = 0.0;
#pragma unroll
    for (int q = 0; q < QN; ++q) {
      const int p = 4 * q + hi;  // pair index: dims 2p, 2p + 1
      double z0 = 0.0, z1 = 0.0;
      if (s < a.S && real_chain && 2 * p < a.d) {
        if (a.z_replay) {
          const size_t o = ((size_t)s * a.N + c) * a.d + 2 * p;
          z0 = a.z_replay[o];
          z1 = (2 * p + 1 < a.d) ? a.z_replay[o + 1] : 0.0;
        } else {
          const double2 zz = normal_pair_call(a.seed, gc, (uint32_t)(a.step0 + s), (uint32_t)p);
          z0 = zz.x;
          z1 = (2 * p + 1 >= a.d) ? 0.0 : zz.y;
        }
        if (a.z_export) {
          const size_t o = ((size_t)s * a.N + c) * a.d + 2 * p;
          a.z_export[o] = z0;
          if (2 * p + 1 < a.d) a.z_export[o + 1] = z1;
        }
      }
      zz += z0 * z0;
      zz += z1 * z1;
      *reinterpret_cast<double2*>(&s_z[lc * RS + 2 * p]) = double2{z0, z1};
      __syncthreads();
      const double za = s_z[lc * RS + 8 * q + hi], zb = s_z[lc * RS + 8 * q + 4 + hi];  // k-steps 2q, 2q + 1
      // all tiles with za, then all with zb: back-to-back MFMAs never wait on their own accumulator
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) acc[tj] = mfma_f64(za, Lf[tj][2 * q], acc[tj]);
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) acc[tj] = mfma_f64(zb, Lf[tj][2 * q + 1], acc[tj]);
    }
    if (a.qz) {  // the four lanes of a step hold a quarter of its pairs each
      zz = sum_rows(zz);
      if (hi == 0 && s < a.S) a.qz[(size_t)s * a.NP + c] = -0.5 * zz;
    }
    // D layout: step hi + 4 r of the group, column 16 tj + lc
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int so = s0 + hi + 4 * r;
        if (so < a.S && 16 * tj + lc < DPAD) a.inc[((size_t)so * a.NP + c) * DPAD + 16 * tj + lc] = acc[tj][r];
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
// The same work split in two, for runs on the engine's own Philox stream:
//   k_rng    the normals of a block (state independent, so they can be drawn while the PREVIOUS block's k_mh_steps
//            runs: at 62 registers this kernel shares the SIMDs with the step kernel's 2 x 224 and uses the VALU slots
//            the matrix phase leaves idle -- tools/overlap_probe.hip: 0.10 ms alone, 0 ms extra next to the steps),
//            stored as the MFMA A fragments k_apply consumes: Zf[group][chain][kk][lane], 512-byte rows;
//            also the accept uniforms and their logs
//   k_apply  INC = Z L^T for one chain per wave (after the Cholesky swap of the block boundary)
// ------------------------------------------------------------------------------------------------
struct RngArgs {
  int64_t N, NP;
  int64_t chain_offset;
  int d;
  int S;
  int64_t step0;
  uint64_t seed;
  double* zf;        // [groups][NP][DPAD/4][64]
  double* u;         // [S][NP]
  double* logu;      // [S][NP]
  double* qz;        // [S][NP] -|z_s|^2 / 2 (may be null)
  double* z_export;  // [S][N][d] (may be null)
  double* u_export;
};

template <int DPAD>
__global__ void __launch_bounds__(64, 8) k_rng(const RngArgs a) {
  constexpr int KK = DPAD / 4, QN = DPAD / 8;
  const int lane = threadIdx.x, lc = lane & 15, hi = lane >> 4;
  const int64_t c = blockIdx.x;
  const int g = blockIdx.y;
  const int s = g * 16 + lc;
  const bool real_chain = c < a.N;
  const uint32_t gc = (uint32_t)(a.chain_offset + c);
  double* __restrict__ dst = a.zf + ((size_t)g * a.NP + c) * KK * 64;
  double zz = 0.0;
#pragma unroll 1
  for (int q = 0; q < QN; ++q) {
    const int p = 4 * q + hi;  // pair index: dims 2p, 2p + 1
    double z0 = 0.0, z1 = 0.0;
    if (s < a.S && real_chain && 2 * p < a.d) {
      normal_pair(a.seed, gc, (uint32_t)(a.step0 + s), STREAM_PROPOSAL, (uint32_t)p, z0, z1);
      if (2 * p + 1 >= a.d) z1 = 0.0;
      if (a.z_export) {
        const size_t o = ((size_t)s * a.N + c) * a.d + 2 * p;
        a.z_export[o] = z0;
        if (2 * p + 1 < a.d) a.z_export[o + 1] = z1;
      }
    }
    // dim 2p = 4 kk + h with kk = 2q + (hi >> 1), h = 2 (hi & 1): fragment lane lc + 16 h (and + 16 for dim 2p + 1)
    const int kk = 2 * q + (hi >> 1), h = 2 * (hi & 1);
    dst[kk * 64 + lc + 16 * h] = z0;
    dst[kk * 64 + lc + 16 * h + 16] = z1;
    zz += z0 * z0;
    zz += z1 * z1;
  }
  if (a.qz) {
    zz = sum_rows(zz);
    if (hi == 0 && s < a.S) a.qz[(size_t)s * a.NP + c] = -0.5 * zz;
  }
}

// Proposal increments when the proposal factor is the identity (CrankNicolson under a standard-normal prior, a random walk with
// C = I): INC = Z, written straight in the [S][NP][DPAD] layout the step kernels read -- no fragments, no product.  A wave takes
// 16 steps of one chain; lane = (step-in-pass, Box-Muller pair): the DPAD / 2 lanes of a step write its DPAD doubles as one
// contiguous row.  Same counters as k_propose / k_rng (block = pair, step, chain), so the increments are bitwise the ones
// k_propose computes with L = I (its products with the zeros of L add exact zeros).
template <int DPAD>
__global__ void __launch_bounds__(64, 8) k_rng_direct(const RngArgs a, double* __restrict__ inc) {
  constexpr int HP = DPAD / 2;   // pairs per step
  constexpr int SPW = 64 / HP;   // steps per pass of the wave
  const int lane = threadIdx.x, p = lane % HP, so = lane / HP;
  const int64_t c = blockIdx.x;
  const int s_hi = (int)blockIdx.y * 16 + 16 < a.S ? (int)blockIdx.y * 16 + 16 : a.S;
  const bool real_chain = c < a.N;
  const uint32_t gc = (uint32_t)(a.chain_offset + c);
  for (int s = (int)blockIdx.y * 16 + so; s < s_hi; s += SPW) {
    double z0 = 0.0, z1 = 0.0;
    if (real_chain && 2 * p < a.d) {
      normal_pair(a.seed, gc, (uint32_t)(a.step0 + s), STREAM_PROPOSAL, (uint32_t)p, z0, z1);
      if (2 * p + 1 >= a.d) z1 = 0.0;
      if (a.z_export) {
        const size_t o = ((size_t)s * a.N + c) * a.d + 2 * p;
        a.z_export[o] = z0;
        if (2 * p + 1 < a.d) a.z_export[o + 1] = z1;
      }
    }
    *reinterpret_cast<double2*>(inc + ((size_t)s * a.NP + c) * DPAD + 2 * p) = double2{z0, z1};
  }
}

// accept uniforms of a block (chain.py:112) and their logs: one thread per (step, chain)
__global__ void __launch_bounds__(256, 2) k_rng_uniforms(const RngArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)a.S * a.NP) return;
  const int64_t s = i / a.NP, c = i % a.NP;
  double u = 0.5;
  if (c < a.N) {
    u = accept_uniform(a.seed, (uint32_t)(a.chain_offset + c), (uint32_t)(a.step0 + s), 0u);
    if (a.u_export) a.u_export[(size_t)s * a.N + c] = u;
  }
  a.u[i] = u;
  a.logu[i] = log(u);
}

struct ApplyArgs {
  int64_t NP;
  int S;
  const double* Lk;   // [NP or 1][DPAD][DPAD] k-major (65 .. 128 parameters: the factor tiles of tda_kernels_wide.h, [2][NPf][36][4][64])
  int64_t L_stride;   // doubles between the factors of two chains (0: one shared factor)
  const double* zf;   // [groups][NP][DPAD/4][64]
  double* inc;        // [S][NP][DPAD]
  // 65 .. 128 parameters only (tda_kernels_wide.h): the diagonal tiles, the current-buffer selector, chains per factor buffer
  const double* ud;
  const int32_t* sel;
  int64_t NPf;
};

template <int DPAD>
__device__ __forceinline__ void apply_chain(const ApplyArgs& a, const int64_t c) {
  constexpr int KK = DPAD / 4;
  constexpr int TJ = DPAD >= 16 ? DPAD / 16 : 1;
  const int lane = threadIdx.x, lc = lane & 15, hi = lane >> 4;
  double Lf[TJ][KK];  // B fragments, see k_propose
#pragma unroll
  for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
      Lf[tj][kk] = 16 * tj + lc < DPAD ? a.Lk[(size_t)c * a.L_stride + (size_t)(4 * kk + hi) * DPAD + 16 * tj + lc] : 0.0;
  const int ng = (a.S + 15) / 16;
  double zf[KK], zn[KK];
  {
    const double* __restrict__ src = a.zf + (size_t)c * KK * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) zf[kk] = src[kk * 64];
  }
  for (int g = 0; g < ng; ++g) {
    {  // next group's fragments fly during this group's MFMAs (clamped: the last group re-reads itself)
      const int gn = g + 1 < ng ? g + 1 : g;
      const double* __restrict__ src = a.zf + ((size_t)gn * a.NP + c) * KK * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) zn[kk] = src[kk * 64];
    }
    double4_t acc[TJ];
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) acc[tj] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int tj = 0; tj < TJ; ++tj) acc[tj] = mfma_f64(zf[kk], Lf[tj][kk], acc[tj]);
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int so = g * 16 + hi + 4 * r;
        if (so < a.S && 16 * tj + lc < DPAD) a.inc[((size_t)so * a.NP + c) * DPAD + 16 * tj + lc] = acc[tj][r];
      }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) zf[kk] = zn[kk];
  }
}

template <int DPAD>
__global__ void __launch_bounds__(64, 2) k_apply(const ApplyArgs a) {
  apply_chain<DPAD>(a, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Adaptation for a block: one wave per chain.
//   RecursiveSampleMoments.update (utils.py:113-124) for each recorded state x (t = recursor.t before the update):
//     mu'    = (1/(t+1)) (t mu + x)
//     Sigma' = (t-1)/t Sigma + sd/t ( t mu mu^T - (t+1) mu' mu'^T + x x^T + eps I )
//   global scaling (proposal.py:234-243).
// The bracket equals t/(t+1) (x - mu)(x - mu)^T identically, so over a block of S updates (t_1 .. t_S)
//     Sigma_S = (t_1 - 1)/t_S Sigma_0 + sum_s w_s delta_s delta_s^T + (S sd eps / t_S) I,
//     delta_s = x_s - mu_{s-1},   w_s = sd t_s / ((t_s + 1) t_S)
// (the factors (t-1)/t telescope).  mu follows the reference recursion literally (bit-identical); the rank-S update
// is a 64 x S x 64 SYRK per chain and runs on the matrix cores: rows sqrt(w_s) delta_s are staged in LDS in chunks of
// AM_CH steps and every 16 x 16 tile on or below the diagonal accumulates 4 steps per v_mfma_f64_16x16x4_f64.
// Against the reference's elementwise recursion this differs by rounding only -- and by less than the reference's own
// error, whose form cancels t mu mu^T against (t+1) mu' mu'^T (1e-11 .. 1e-8 relative when |mu| >> spread).  The
// elementwise form costs 330 non-fusable fp64 VALU operations per state and was 23 % of the pipeline (1.07 ns/eval).
// Sigma is stored per chain as the 16 x 16 tiles on or below the diagonal in the MFMA C/D register layout
// ([tile][r][lane] <-> row 16 ti + (lane >> 4) + 4 r, column 16 tj + (lane & 15); tile = ti (ti + 1) / 2 + tj), so its
// read-modify-write is four coalesced 512-byte accesses per tile.
// ------------------------------------------------------------------------------------------------
constexpr int AM_CH = 20;  // steps per LDS chunk (multiple of 4)

template <int DPAD>
__host__ __device__ constexpr int am_tile_rows() {
  return DPAD >= 16 ? DPAD / 16 : 1;
}
template <int DPAD>
__host__ __device__ constexpr int am_tiles() {
  return am_tile_rows<DPAD>() * (am_tile_rows<DPAD>() + 1) / 2;
}
// offset (in doubles, within one chain's block of am_tiles * 256) of Sigma[i][j], i >= j
__host__ __device__ inline int am_sigma_offset(int i, int j) {
  const int ti = i >> 4, tj = j >> 4, ri = i & 15, cj = j & 15;
  return (((ti * (ti + 1) / 2 + tj) * 4 + (ri >> 2)) * 64) + (ri & 3) * 16 + cj;
}

// global scaling adaptation at a period boundary (proposal.py:234-243) and the reset of the acceptance counter
__device__ __forceinline__ void adapt_scaling(const AdaptArgs& a, int64_t c, int lane) {
  if (!a.boundary) return;
  if (a.do_scale && lane == 0) {
    int hits = 0;
    if (a.ring) {
      for (int i = 1; i <= a.period; ++i) hits += a.ring[(size_t)((a.ring_hi - i) % a.ring_P) * a.NP + c];
    } else {
      hits = a.acc_count[c];
    }
    const double rate = (double)hits / (double)a.period;  // np.mean(accepted[-period:])
    a.scaling[c] = exp(log(a.scaling[c]) + a.gamma_pow * (rate - a.alpha_star));
  }
  if (lane == 0) a.acc_count[c] = 0;
}

// wave-uniform broadcast of lane `src`'s value (src uniform)
__device__ __forceinline__ double bcast_lane64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// Reference-form recursion (the default).  Every element follows RecursiveSampleMoments.update operation for operation
// (this file is compiled with -ffp-contract=off, so products and sums round exactly like NumPy's).  Sigma lives in
// registers in the tile layout described above: lane (lc, hi) holds, for each tile (ti, tj) on or below the diagonal,
// the 4 elements (16 ti + hi + 4 r, 16 tj + lc).  Per state the three vectors x, mu, mu' go to LDS twice: in natural
// order (column operands: one ds_read_b64 per tile column) and permuted so that a lane's four row indices are
// contiguous (row operands: two ds_read_b128 per tile row); mu mu^T is carried over from the previous step's
// (t + 1) mu' mu'^T.  24 LDS reads + 320 fp64 operations per state; the circulant fold used earlier needed 99 + 330 and ran
// at the speed of the 36 + 400 tile version: the bound is the VALU.  Round 2 confirmed it from the other side: staging the
// operands of step s + 1 under step s (no store -> load round trip in front of a step) and reading the row operands one tile
// row ahead changed nothing (347 -> 349-359 us per 100 states of 4096 chains); the 382 vector instructions of a state take
// 2 100 cycles of a SIMD with its two resident waves, 5.5 per instruction, where tools/valu_rate_probe.hip gets 4.1 for the
// same 5 mul : 3 add mix on 16 registers -- the rest is the operand traffic of 230 live registers, not latency.  Also tried late in round 2,
// same-box A/B on the headline run: pinning this state's record before the next one's load is issued (the compiler waits with
// vmcnt(0) at the first use, which otherwise covers the load just issued) -- 8.85 against 8.89 ms per 20 periods, no change:
// the partner wave's arithmetic already covers that latency; the wave-scope fence instead of __syncthreads() -- no change.
// At DPAD <= 16 the two together took 82 -> 66-71 us per 100 states of 4096 chains (the fixed ~50 instructions per state
// dominate there), at DPAD = 32 nothing (150 us: four resident waves x 125 instructions, the vector unit 56 % busy); not
// adopted, no benchmark configuration adapts in fewer than 64 dimensions.
// the moment recursion of one chain by its wave; Sigma is left in Sg (and stored): k_adapt, and the first half of k_adapt_chol_apply
// HALF: -1 = every tile row; 0 / 1 (64 parameters, k_adapt_split): the tile rows {0, 3} / {1, 2} -- five of the ten tiles each -- for two
// waves that share a chain: 80 instead of 160 registers of Sigma and t mu mu^T, four waves per SIMD instead of two
template <int HALF>
__host__ __device__ constexpr bool am_row_mine(int ti) {
  return HALF < 0 || (HALF == 0 ? (ti == 0 || ti == 3) : (ti == 1 || ti == 2));
}
template <int DPAD, int HALF = -1>
__device__ __forceinline__ void adapt_am_chain(const AdaptArgs& a, const int64_t c, const int lane, double (&Sg)[am_tiles<DPAD>()][4]) {
  constexpr int T = am_tile_rows<DPAD>();
  constexpr int NTL = am_tiles<DPAD>();
  constexpr int W = 16 * T;
  __shared__ __attribute__((aligned(16))) double s_nat[2 * W];  // x, mu' by dimension
  __shared__ __attribute__((aligned(16))) double s_prm[2 * W];  // the same, dimension 16 ti + h + 4 r at 16 ti + 4 h + r
  const bool lj = lane < a.d;
  const bool lp = lane < DPAD;
  const int lc = lane & 15, hi = lane >> 4;
  {
    double* __restrict__ sig = a.am_sigma + (size_t)c * NTL * 256;
    double TM[NTL][4];  // t mu mu^T of the current mean, element (16 ti + hi + 4 r, 16 tj + lc), like Sigma in Sg
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
      for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (am_row_mine<HALF>(ti)) Sg[ti * (ti + 1) / 2 + tj][r] = sig[((ti * (ti + 1) / 2 + tj) * 4 + r) * 64 + lane];
    double mu = lp ? a.am_mu[c * DPAD + lane] : 0.0;
    const int ppos = (lane & ~15) | ((lane & 3) << 2) | ((lane >> 2) & 3);
    // (t + 1) (mu'_s mu'_s^T) of step s IS t (mu_{s+1} mu_{s+1}^T) of step s + 1 (same operands, same rounding): it is
    // carried in registers instead of being multiplied out again, and mu itself never goes through LDS after this prologue
    if (lane < W) {
      s_nat[lane] = mu;
      s_prm[ppos] = mu;
    }
    __syncthreads();
    {
      double mc[T];
#pragma unroll
      for (int tj = 0; tj < T; ++tj) mc[tj] = s_nat[16 * tj + lc];
#pragma unroll
      for (int ti = 0; ti < T; ++ti) {
        if (!am_row_mine<HALF>(ti)) continue;
        const double2* __restrict__ q = reinterpret_cast<const double2*>(s_prm + 16 * ti + 4 * hi);
        const double2 m01 = q[0], m23 = q[1];
        const double mr[4] = {m01.x, m01.y, m23.x, m23.y};
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
          for (int r = 0; r < 4; ++r) TM[ti * (ti + 1) / 2 + tj][r] = (double)(a.t_base + 1) * (mr[r] * mc[tj]);
      }
    }
    // the step coefficients depend on t only: lane l works out those of step 64 k + l (three fp64 divisions, ~45
    // instructions that every lane would otherwise repeat in every step) and the loop reads them back with v_readlane
    double c_inv = 0.0, c_a = 0.0, c_b = 0.0;
    double xn = lj ? a.rec_params[(size_t)c * a.d + lane] : 0.0;
    // the eps I term as an operand instead of a select: element r of a diagonal tile is on the diagonal iff hi + 4 r == lc
    // (padded dimensions collect eps too; nothing reads them)
    double epsd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) epsd[r] = (hi + 4 * r == lc) ? a.eps : 0.0;
    for (int s = 0; s < a.S; ++s) {
      if ((s & 63) == 0) {
        const double tl = (double)(a.t_base + s + lane + 1);
        c_inv = 1.0 / (tl + 1.0);
        c_a = (tl - 1.0) / tl;
        c_b = a.sd / tl;
      }
      const double x = xn;
      if (s + 1 < a.S) xn = lj ? a.rec_params[((size_t)(s + 1) * a.N + c) * a.d + lane] : 0.0;
      const double t = (double)(a.t_base + s + 1);  // recursor.t before this update
      const double mup = bcast_lane64(c_inv, s & 63) * (t * mu + x);
      const double ca = bcast_lane64(c_a, s & 63), cb = bcast_lane64(c_b, s & 63);
      const double t1 = t + 1.0;
      __syncthreads();  // previous step's operand reads are done
      if (lane < W) {
        s_nat[lane] = x;
        s_nat[W + lane] = mup;
        s_prm[ppos] = x;
        s_prm[W + ppos] = mup;
      }
      __syncthreads();
      double xc[T], pc[T];
#pragma unroll
      for (int tj = 0; tj < T; ++tj) {
        xc[tj] = s_nat[16 * tj + lc];
        pc[tj] = s_nat[W + 16 * tj + lc];
      }
#pragma unroll
      for (int ti = 0; ti < T; ++ti) {
        if (!am_row_mine<HALF>(ti)) continue;
        double xr[4], pr[4];
        {
          const double2* __restrict__ q = reinterpret_cast<const double2*>(s_prm + 16 * ti + 4 * hi);
          const double2 x01 = q[0], x23 = q[1], p01 = q[W / 2], p23 = q[W / 2 + 1];
          xr[0] = x01.x; xr[1] = x01.y; xr[2] = x23.x; xr[3] = x23.y;
          pr[0] = p01.x; pr[1] = p01.y; pr[2] = p23.x; pr[3] = p23.y;
        }
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
          const int idx = ti * (ti + 1) / 2 + tj;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double tp = t1 * (pr[r] * pc[tj]);
            double M = (TM[idx][r] - tp) + xr[r] * xc[tj];
            if (ti == tj) M += epsd[r];  // + eps on the diagonal, + 0 beside it
            Sg[idx][r] = ca * Sg[idx][r] + cb * M;
            TM[idx][r] = tp;
          }
        }
      }
      mu = mup;
    }
    if (lp && HALF <= 0) a.am_mu[c * DPAD + lane] = mu;  // (both halves carry the mean; one stores it)
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
      for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (am_row_mine<HALF>(ti)) sig[((ti * (ti + 1) / 2 + tj) * 4 + r) * 64 + lane] = Sg[ti * (ti + 1) / 2 + tj][r];
  }
}

// The same recursion at 64 parameters with the four DIAGONAL tiles kept as circulant slots instead of full 16 x 16 tiles (round 5).
// A diagonal tile is symmetric -- bitwise: x_i x_j = x_j x_i, and so for every product of the recursion --, so 120 of its 256
// elements are computed twice.  Here lane l = 16 b + lc holds, of diagonal block b, the nine pairs {lc, (lc + k) mod 16}, k = 0 .. 8
// (every unordered pair once, those at distance 8 twice): 9 instead of 16 registers of Sigma and of t mu mu^T, 72 instead of 128
// operations per state; in all 264 instead of 320.  The column operand of a slot is the lane's own x / mu' (a register), its row
// operand one of eight consecutive doubles of a block stored twice in LDS (four ds_read2_b64 per vector); the six tiles below the
// diagonal keep the tile layout and their operand reads (the column operands now come from the same doubled array).  Memory layout
// of Sigma unchanged; same operations per element in the same order: the same bits (the "+ 0" that the tile version adds to the
// off-diagonal elements of a diagonal tile is not added: it can change the sign of a zero only).  NEED_TILES: leave the diagonal
// tiles in Sg in the tile layout (k_adapt_chol_apply factorises from registers), converted through LDS once per block.
#ifndef CIRC_SCHED
#define CIRC_SCHED
#endif
// one ds_read_b64 (2 LDS cycles per wave: MI355X_MICROARCH.md, LDS), kept from being paired into a ds_read2_b64 (8 cycles)
__device__ __forceinline__ double lds_b64(const double* p) {
#ifdef CIRC_TIMING_NOLDS
  return (double)(int)(size_t)p;  // timing-only build: no read
#else
  return *(const volatile __attribute__((address_space(3))) double*)p;
#endif
}
template <bool NEED_TILES>
__device__ __forceinline__ void adapt_am_chain_c64_store(const AdaptArgs& a, const int64_t c, const int lane, const double mu, double (&Sg)[10][4],
                                                         const double (&Dg)[9]);
template <bool NEED_TILES>
__device__ __forceinline__ void adapt_am_chain_c64(const AdaptArgs& a, const int64_t c, const int lane, double (&Sg)[10][4]) {
  constexpr int W = 64;
  // x, mu': block b (16 dimensions) at BS b .. BS b + 15 and again at + 16; BS = 48: the two blocks of a 32-lane ds_read_b64 group on disjoint banks
  constexpr int BS = 48, VS = 4 * BS;
  __shared__ __attribute__((aligned(16))) double s_dbl[2 * VS];
  __shared__ __attribute__((aligned(16))) double s_prm[2 * W];    // x, mu': dimension 16 ti + h + 4 r at 16 ti + 4 h + r
  const bool lj = lane < a.d;
  const int lc = lane & 15, hi = lane >> 4;
  double* __restrict__ sig = a.am_sigma + (size_t)c * 10 * 256;
  // t mu mu^T of the current mean in TWO sets used alternately (the loop takes two states per pass): a state reads one and
  // writes (t + 1) mu' mu'^T into the other, so that no register is copied (with one set: 34 v_mov_b64 of 316 instructions per state)
  double TM[2][10][4];  // (only the six tiles below the diagonal are used)
  double Dg[9], TD[2][9];
  int doff[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const int i = 16 * hi + ((lc + k) & 15), j = 16 * hi + lc;
    doff[k] = i >= j ? am_sigma_offset(i, j) : am_sigma_offset(j, i);
  }
#pragma unroll
  for (int ti = 1; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < ti; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) Sg[ti * (ti + 1) / 2 + tj][r] = sig[((ti * (ti + 1) / 2 + tj) * 4 + r) * 64 + lane];
#pragma unroll
  for (int k = 0; k < 9; ++k) Dg[k] = sig[doff[k]];
  double mu = a.am_mu[c * 64 + lane];
  const int ppos = (lane & ~15) | ((lane & 3) << 2) | ((lane >> 2) & 3);
  const int dpos = BS * hi + lc;
  s_dbl[dpos] = mu;
  s_dbl[dpos + 16] = mu;
  s_prm[ppos] = mu;
  __syncthreads();
  {
    double mc[3];
#pragma unroll
    for (int tj = 0; tj < 3; ++tj) mc[tj] = s_dbl[BS * tj + lc];
#pragma unroll
    for (int ti = 1; ti < 4; ++ti) {
      const double2* __restrict__ q = reinterpret_cast<const double2*>(s_prm + 16 * ti + 4 * hi);
      const double2 m01 = q[0], m23 = q[1];
      const double mr[4] = {m01.x, m01.y, m23.x, m23.y};
#pragma unroll
      for (int tj = 0; tj < ti; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) TM[0][ti * (ti + 1) / 2 + tj][r] = (double)(a.t_base + 1) * (mr[r] * mc[tj]);
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) TD[0][k] = (double)(a.t_base + 1) * (s_dbl[dpos + k] * mu);
  }
  double c_inv = 0.0, c_a = 0.0, c_b = 0.0;
  double xn = lj ? a.rec_params[(size_t)c * a.d + lane] : 0.0;
  const double eps = a.eps;
  auto state = [&](const int s, const double (&TMi)[10][4], double (&TMo)[10][4], const double (&TDi)[9], double (&TDo)[9]) {
    if ((s & 63) == 0) {
      const double tl = (double)(a.t_base + s + 1) + (double)lane;  // (integers below 2^53: the sum is exact)
      c_inv = 1.0 / (tl + 1.0);
      c_a = (tl - 1.0) / tl;
      c_b = a.sd / tl;
    }
    const double x = xn;
    if (s + 1 < a.S) xn = lj ? a.rec_params[((size_t)(s + 1) * a.N + c) * a.d + lane] : 0.0;
    const double t = (double)(a.t_base + s + 1);  // recursor.t before this update
    const double mup = bcast_lane64(c_inv, s & 63) * (t * mu + x);
    const double ca = bcast_lane64(c_a, s & 63), cb = bcast_lane64(c_b, s & 63);
    const double t1 = t + 1.0;
    __syncthreads();  // previous step's operand reads are done
    s_dbl[dpos] = x;
    s_dbl[dpos + 16] = x;
    s_dbl[VS + dpos] = mup;
    s_dbl[VS + dpos + 16] = mup;
    s_prm[ppos] = x;
    s_prm[W + ppos] = mup;
    __syncthreads();
    // every operand of the state requested first: the diagonal slots' (slot k = the pair (16 b + (lc + k) mod 16, 16 b + lc)), the
    // column operands and the three tile rows'
    double xd[9], pd[9], xc[3], pc[3], xr[4][4], pr[4][4];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      xd[k] = k == 0 ? x : lds_b64(s_dbl + dpos + k);
      pd[k] = k == 0 ? mup : lds_b64(s_dbl + VS + dpos + k);
    }
#pragma unroll
    for (int tj = 0; tj < 3; ++tj) {
      xc[tj] = lds_b64(s_dbl + BS * tj + lc);
      pc[tj] = lds_b64(s_dbl + VS + BS * tj + lc);
    }
#pragma unroll
    for (int ti = 1; ti < 4; ++ti) {
      const double2* __restrict__ q = reinterpret_cast<const double2*>(s_prm + 16 * ti + 4 * hi);
#ifdef CIRC_TIMING_NOLDS
      const double2 x01 = {x, mup}, x23 = {mup, x}, p01 = {x + 1.0, mup}, p23 = {x, mup + 1.0};
#else
      const double2 x01 = q[0], x23 = q[1], p01 = q[W / 2], p23 = q[W / 2 + 1];
#endif
      xr[ti][0] = x01.x; xr[ti][1] = x01.y; xr[ti][2] = x23.x; xr[ti][3] = x23.y;
      pr[ti][0] = p01.x; pr[ti][1] = p01.y; pr[ti][2] = p23.x; pr[ti][3] = p23.y;
    }
    // (t - 1)/t Sigma needs no operand: its 33 multiplications stand between the requests and the first wait (the scheduling barrier
    // keeps the compiler from sinking them to their uses)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      Dg[k] = ca * Dg[k];
      asm volatile("" : "+v"(Dg[k]));  // (an ordered no-op: the product exists before the barrier below)
    }
#pragma unroll
    for (int ti = 1; ti < 4; ++ti)
#pragma unroll
      for (int tj = 0; tj < ti; ++tj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Sg[ti * (ti + 1) / 2 + tj][r] = ca * Sg[ti * (ti + 1) / 2 + tj][r];
          asm volatile("" : "+v"(Sg[ti * (ti + 1) / 2 + tj][r]));
        }
    __builtin_amdgcn_sched_barrier(0);
    {
      double q[9], w[9], m[9];
#pragma unroll
      for (int g = 0; g < 9; g += 3) {  // three slots operation by operation
#pragma unroll
        for (int k = g; k < g + 3; ++k) q[k] = pd[k] * mup;
#pragma unroll
        for (int k = g; k < g + 3; ++k) w[k] = xd[k] * x;
#pragma unroll
        for (int k = g; k < g + 3; ++k) q[k] = t1 * q[k];
#pragma unroll
        for (int k = g; k < g + 3; ++k) m[k] = TDi[k] - q[k];
#pragma unroll
        for (int k = g; k < g + 3; ++k) TDo[k] = q[k];
#pragma unroll
        for (int k = g; k < g + 3; ++k) m[k] = m[k] + w[k];
        if (g == 0) m[0] += eps;
#pragma unroll
        for (int k = g; k < g + 3; ++k) m[k] = cb * m[k];
#pragma unroll
        for (int k = g; k < g + 3; ++k) Dg[k] = Dg[k] + m[k];
      }
    }
#pragma unroll
    for (int ti = 1; ti < 4; ++ti) {
#pragma unroll
      for (int tj = 0; tj < ti; ++tj) {
        const int idx = ti * (ti + 1) / 2 + tj;
        // the four elements of a tile operation by operation (independent instructions back to back, not four dependent chains)
        double q[4], m[4], w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = pr[ti][r] * pc[tj];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = xr[ti][r] * xc[tj];
#pragma unroll
        for (int r = 0; r < 4; ++r) q[r] = t1 * q[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = TMi[idx][r] - q[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) TMo[idx][r] = q[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = m[r] + w[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] = cb * m[r];
#pragma unroll
        for (int r = 0; r < 4; ++r) Sg[idx][r] = Sg[idx][r] + m[r];
      }
    }
    mu = mup;
  };
  for (int s = 0; s < a.S; s += 2) {
    state(s, TM[0], TM[1], TD[0], TD[1]);
    if (s + 1 < a.S) state(s + 1, TM[1], TM[0], TD[1], TD[0]);
  }
  // what follows the loop derives its addresses from a lane index the compiler cannot identify with `lane`: kept live across the
  // loop for these few stores they cost registers the loop does not have (7 / 15 spilled)
  int lane2 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  asm volatile("" : "+v"(lane2));
  __builtin_assume(lane2 >= 0 && lane2 < 64);
  return adapt_am_chain_c64_store<NEED_TILES>(a, c, lane2, mu, Sg, Dg);
}

template <bool NEED_TILES>
__device__ __forceinline__ void adapt_am_chain_c64_store(const AdaptArgs& a, const int64_t c, const int lane, const double mu, double (&Sg)[10][4],
                                                         const double (&Dg)[9]) {
  __shared__ __attribute__((aligned(16))) double s_cv[NEED_TILES ? 4 * 256 : 2];
  const int lc = lane & 15, hi = lane >> 4;
  double* __restrict__ sig = a.am_sigma + (size_t)c * 10 * 256;
  a.am_mu[c * 64 + lane] = mu;
#pragma unroll
  for (int ti = 1; ti < 4; ++ti)
#pragma unroll
    for (int tj = 0; tj < ti; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) sig[((ti * (ti + 1) / 2 + tj) * 4 + r) * 64 + lane] = Sg[ti * (ti + 1) / 2 + tj][r];
  if constexpr (NEED_TILES) {
    // the diagonal tiles in the tile layout: both mirror images through LDS, [b][row][column]; stored from there
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int i = (lc + k) & 15;
      s_cv[hi * 256 + i * 16 + lc] = Dg[k];
      s_cv[hi * 256 + lc * 16 + i] = Dg[k];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = s_cv[p * 256 + (hi + 4 * r) * 16 + lc];
        Sg[p * (p + 1) / 2 + p][r] = v;
        sig[((p * (p + 1) / 2 + p) * 4 + r) * 64 + lane] = v;
      }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) {  // both mirror images of a diagonal tile (its readers take the whole tile); distance 8: two lanes store the same bits
      const int i = (lc + k) & 15, tile = (hi * (hi + 1) / 2 + hi) * 256;
      sig[tile + (i >> 2) * 64 + (i & 3) * 16 + lc] = Dg[k];
      sig[tile + (lc >> 2) * 64 + (lc & 3) * 16 + i] = Dg[k];
    }
  }
}

template <int DPAD, bool CIRC = false>
__global__ void __launch_bounds__(64, CIRC ? 2 : 1) k_adapt(const AdaptArgs a) {
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  if (a.do_am) {
    double Sg[am_tiles<DPAD>()][4];
    if constexpr (CIRC) {
      static_assert(DPAD == 64, "circulant diagonal blocks: four tile rows");
      adapt_am_chain_c64<false>(a, c, lane, Sg);
    } else {
      adapt_am_chain<DPAD>(a, c, lane, Sg);
    }
  }
  adapt_scaling(a, c, lane);
}

// AdaptiveMetropolis(block_moments=True): the covariance as one rank-S update per block (see above)
// The same recursion with the ten tiles of a 64-parameter chain dealt to TWO waves (workgroups 2 c and 2 c + 1: tile rows {0, 3} and
// {1, 2}), each carrying the mean itself: 5 tiles of Sigma and of t mu mu^T per wave, at most 128 registers, FOUR waves per SIMD.  The
// loop is bound by vector issue (434 vector instructions per state); tools/valu_rate_probe.hip prices that mix at 3.95
// cycles per instruction with two waves per SIMD and 2.87 with four.  Same operations per element, same results bit for bit.
template <int DPAD>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 3))) k_adapt_split(const AdaptArgs a) {
  static_assert(DPAD == 64, "four tile rows");
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x >> 1;
  if (c >= a.N) return;
  if (blockIdx.x & 1) {
    double Sg[am_tiles<DPAD>()][4];
    adapt_am_chain<DPAD, 1>(a, c, lane, Sg);
  } else {
    double Sg[am_tiles<DPAD>()][4];
    adapt_am_chain<DPAD, 0>(a, c, lane, Sg);
    adapt_scaling(a, c, lane);
  }
}

template <int DPAD>
__global__ void __launch_bounds__(64) k_adapt_block(const AdaptArgs a) {
  constexpr int T = am_tile_rows<DPAD>();
  constexpr int NTL = am_tiles<DPAD>();
  constexpr int W = 16 * T;   // staged row width (>= DPAD)
  constexpr int RS = W + 2;   // row stride in LDS
  __shared__ __attribute__((aligned(16))) double s_d[AM_CH * RS];
  __shared__ double s_coef[3 * AM_CH];
  const int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  if (c >= a.N) return;
  const bool lj = lane < a.d;
  const int lc = lane & 15, hi = lane >> 4;

  if (a.do_am) {
    double4_t acc[NTL];
#pragma unroll
    for (int i = 0; i < NTL; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    double mu = lane < DPAD ? a.am_mu[c * DPAD + lane] : 0.0;
    const double tS = (double)(a.t_base + a.S);
    for (int s0 = 0; s0 < a.S; s0 += AM_CH) {
      const int ns = a.S - s0 < AM_CH ? a.S - s0 : AM_CH;
      double xs[AM_CH];  // the chunk's states, all requested before the sequential recursion consumes them
#pragma unroll
      for (int i = 0; i < AM_CH; ++i)
        xs[i] = (lj && i < ns) ? a.rec_params[((size_t)(s0 + i) * a.N + c) * a.d + lane] : 0.0;
      // the step coefficients depend on t only: lane i works out those of step s0 + i (two divisions and a square
      // root) while the loads fly, so that the sequential recursion below is five VALU operations per state
      {
        const double t = (double)(a.t_base + s0 + lane + 1);  // recursor.t before the update of step s0 + lane
        if (lane < AM_CH) {
          s_coef[3 * lane + 0] = sqrt(a.sd * t / ((t + 1.0) * tS));
          s_coef[3 * lane + 1] = 1.0 / (t + 1.0);
          s_coef[3 * lane + 2] = t;
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < AM_CH; ++i) {
        double dv = 0.0;
        if (i < ns) {
          dv = s_coef[3 * i + 0] * (xs[i] - mu);
          mu = s_coef[3 * i + 1] * (s_coef[3 * i + 2] * mu + xs[i]);
        }
        if (lane < W) s_d[i * RS + lane] = dv;
      }
      __syncthreads();
      const int nq = (ns + 3) >> 2;
      for (int kq = 0; kq < nq; ++kq) {
        double v[T];
#pragma unroll
        for (int t = 0; t < T; ++t) v[t] = s_d[(4 * kq + hi) * RS + 16 * t + lc];
#pragma unroll
        for (int ti = 0; ti < T; ++ti)
#pragma unroll
          for (int tj = 0; tj <= ti; ++tj) acc[ti * (ti + 1) / 2 + tj] = mfma_f64(v[ti], v[tj], acc[ti * (ti + 1) / 2 + tj]);
      }
      __syncthreads();
    }
    const double P = (double)a.t_base / tS;
    const double e = (double)a.S * a.sd * a.eps / tS;
    double* __restrict__ sig = a.am_sigma + (size_t)c * NTL * 256;
#pragma unroll
    for (int ti = 0; ti < T; ++ti)
#pragma unroll
      for (int tj = 0; tj <= ti; ++tj) {
        const int idx = ti * (ti + 1) / 2 + tj;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int off = (idx * 4 + r) * 64 + lane;
          double v = P * sig[off] + acc[idx][r];
          if (ti == tj && hi + 4 * r == lc && 16 * ti + lc < a.d) v += e;
          sig[off] = v;
        }
      }
    if (lane < DPAD) a.am_mu[c * DPAD + lane] = mu;
  }

  adapt_scaling(a, c, lane);
}

// ------------------------------------------------------------------------------------------------
// C <- Sigma (proposal.py:509-510) and its Cholesky factor, one wave per chain, matrix in LDS,
// left-looking by columns with lane i = row i, sequential fma chain per element.
// ------------------------------------------------------------------------------------------------
struct CholArgs {
  int64_t N;
  int d;
  const double* am_sigma;  // [NP][am_tiles][4][64] (see k_adapt)
  double* Lk;              // [NP][DPAD][DPAD] k-major
  int32_t* flags;
  // 65 .. 128 parameters only (tda_kernels_wide.h: the swap runs on k_aem_refresh): the diagonal tiles, the buffer selector, the
  // "Sigma_e" of that kernel (zero inside the d parameters, identity in the padding), chains per buffer
  double* ud;
  int32_t* sel;
  const double* pad_cov;
  int64_t NP;
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
__device__ __forceinline__ void chol_chain(const CholArgs& a, const int64_t c) {
  constexpr int NTL = am_tiles<DPAD>();
  const int lane = threadIdx.x;
  const bool lj = lane < a.d;
  const int li = lane < DPAD ? lane : DPAD - 1;
  double A[DPAD];
  // row `lane` of Sigma (columns j <= lane); padded rows / columns = identity
#pragma unroll
  for (int j = 0; j < DPAD; ++j) {
    double v = (j == li) ? 1.0 : 0.0;
    if (lj && j < a.d && j <= li) v = a.am_sigma[(size_t)c * NTL * 256 + am_sigma_offset(li, j)];
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

template <int DPAD>
__global__ void __launch_bounds__(64) k_chol(const CholArgs a) {
  if ((int64_t)blockIdx.x >= a.N) return;
  chol_chain<DPAD>(a, blockIdx.x);
}

// C <- Sigma at a swap boundary AND the next block's increments INC = Z L^T in one launch, one wave per chain: the factor a
// wave has just stored is read back as MFMA B fragments by the same wave (an L2 hit: nothing else touches that chain's factor;
// the stores are complete and the wave's L1 view invalidated before the first load).  Against k_chol followed by k_apply this
// saves a kernel boundary and the 134 MB the second kernel read from HBM, and it lets the two phases overlap ACROSS waves: the
// Cholesky is bound by its chain of pivots (v_readlane -> sqrt -> divide, two waves per SIMD), the product by HBM (the
// normals in, the increments out), and a wave in one phase no longer waits for every other wave to finish the other.
template <int DPAD>
__global__ void __launch_bounds__(64, 2) k_chol_apply(const CholArgs ca, const ApplyArgs ap) {
  const int64_t c = blockIdx.x;
  if (c < ca.N) {
    chol_chain<DPAD>(ca, c);
    // The wave reads back what it has just written itself: the stores only have to be complete (the vector L1 is write-through,
    // and nothing on this CU has read these lines since the launch invalidated it).  Workgroup scope does exactly that; an
    // AGENT-scope release / acquire pair writes back and invalidates the XCD's whole L2 per wave (the L2s of the eight XCDs are not
    // coherent with each other) and made the launch 75 % slower than the two kernels it replaces.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  apply_chain<DPAD>(ap, c);  // (padding chains c >= N: their factor is whatever init left there, as for k_apply)
}


// ------------------------------------------------------------------------------------------------
// Blocked covariance swap + increments for 64 parameters (round 3).  The unblocked kernel above spends its time on operand
// delivery: 2016 column updates per chain, each one fma fed by TWO v_readlane (9 000 vector instructions per chain, bound by the
// pivot chain at two waves per SIMD).  Here the matrix stays in the layout k_adapt keeps it in -- 16 x 16 tiles in the MFMA C/D
// register layout, as the UPPER factor (tile (p, i), p <= i, holds rows 16 p + hi + 4 r, column 16 i + lc of U = L^T):
//   * a block row of 16 pivots is eliminated across all its tiles at once: the pivot row reaches the other lanes of its column
//     with one ds_bpermute per tile, the four multipliers of a lane's own rows with four more -- no per-element readlane;
//   * the trailing update of the block rows below runs on the matrix cores: tile(q, i) -= U_pq^T U_pi is
//     mfma(-tile(p, q)[r], tile(p, i)[r], .) over the four registers r, both operands ARE the C/D registers of the block row
//     (the accumulator layout of the fp64 MFMA is its operand layout for X^T Y products);
//   * and the finished tiles ARE the B fragments of INC = Z L^T (row 16 p + 4 r + hi <-> k = 4 kk + hi with kk = 4 p + r), so the
//     increments of the next block are multiplied out of the registers that hold the factor: no read-back at all, and the
//     16 - 10 = 6 all-zero tiles below the block diagonal are skipped (40 instead of 64 MFMAs per group of 16 steps).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ constexpr int up_tile(int p, int i) { return p * 4 - p * (p - 1) / 2 + (i - p); }  // p <= i < 4

__device__ __forceinline__ double lane_pick(double v, int src) { return __shfl(v, src); }

// The factorisation and the increments from the tiles in G (upper tiles of Sigma for chains c < ca.N; padding chains and chains
// whose Sigma is not positive definite take the factor in memory).  APPLY = false: the swap alone (multilevel drivers, whose
// increments come from the fused k_propose).
template <int DPAD, bool APPLY>
__device__ __forceinline__ void chol_apply_tiles(const CholArgs& ca, const ApplyArgs& ap, const int64_t c, double (&G)[10][4], const int lane_in = -1) {
  const int lane = lane_in < 0 ? (int)threadIdx.x : lane_in, lc = lane & 15, hi = lane >> 4;
  bool have = false;
  if (c < ca.N) {
    const int d = ca.d;
    bool ok = true;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
#pragma unroll
      for (int kl = 0; kl < 16; ++kl) {
        constexpr int dummy = 0;
        (void)dummy;
        const int kh = kl & 3, kr = kl >> 2;
        const double dkk = bcast_lane64(G[up_tile(p, p)][kr], kh * 16 + kl);
        ok = ok && (dkk > 0.0);
        // 1 / sqrt(pivot): the hardware estimate (2^-26) and one third-order correction y (1 + e / 2 + 3 e^2 / 8), e = 1 - d y^2 --
        // eight instructions where sqrt followed by a division takes thirty, per pivot and on every lane; the result is within
        // a few units in the last place, which is what a factor of a sample covariance needs (the swap is compared at 1e-9)
        const double y0 = __builtin_amdgcn_rsq(dkk);
        const double e0 = fma(-dkk * y0, y0, 1.0);
        const double inv = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
        const double fac = (hi == kh) ? inv : 1.0;  // the pivot row lives in the lanes with hi == kh
#pragma unroll
        for (int i = p; i < 4; ++i) G[up_tile(p, i)][kr] *= fac;
        // multipliers of this lane's rows 16 p + hi + 4 r' > k: U[k][16 p + hi + 4 r'], from the lane that holds that column
        double ucol[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) ucol[rr] = rr >= kr ? lane_pick(G[up_tile(p, p)][kr], kh * 16 + hi + 4 * rr) : 0.0;
#pragma unroll
        for (int i = p; i < 4; ++i) {
          const double urow = lane_pick(G[up_tile(p, i)][kr], kh * 16 + lc);  // U[k][16 i + lc] for every row group of this column
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            if (rr < kr) continue;  // rows 16 p + hi + 4 rr <= k: finished
            const double upd = fma(-ucol[rr], urow, G[up_tile(p, i)][rr]);
            G[up_tile(p, i)][rr] = (rr > kr || hi > kh) ? upd : G[up_tile(p, i)][rr];
          }
        }
      }
      // block rows below: tile(q, i) -= U_pq^T U_pi on the matrix cores
#pragma unroll
      for (int q = p + 1; q < 4; ++q)
#pragma unroll
        for (int i = q; i < 4; ++i) {
          double4_t acc = {G[up_tile(q, i)][0], G[up_tile(q, i)][1], G[up_tile(q, i)][2], G[up_tile(q, i)][3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = mfma_f64(-G[up_tile(p, q)][r], G[up_tile(p, i)][r], acc);
#pragma unroll
          for (int r = 0; r < 4; ++r) G[up_tile(q, i)][r] = acc[r];
        }
    }
    if (ok) {
      have = true;
      // U as the factor in use: zero below the diagonal and in the padding, stored k-major (Lk[k][j] = L[j][k] = U[k][j])
      double* __restrict__ Lk = ca.Lk + (size_t)c * DPAD * DPAD;
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * p + hi + 4 * r, col = 16 * i + lc;
            double v = 0.0;
            if (i >= p) {
              v = (col >= row && row < d && col < d) ? G[up_tile(p, i < p ? p : i)][r] : 0.0;
              G[up_tile(p, i < p ? p : i)][r] = v;
            }
            Lk[(size_t)row * DPAD + col] = v;
          }
    } else if (lane == 0) {
      atomicOr(&ca.flags[c], 1);
    }
  }
  if constexpr (!APPLY) return;
  // ---- increments of the next block: INC[S][64] = Z[S][64] L^T, 16 steps per group ----
  constexpr int KK = DPAD / 4;
  if (!have) {  // not positive definite (flagged) or a padding chain: the factor in memory, tile by tile in the same register layout
    const double* __restrict__ Lk = ap.Lk + (size_t)c * ap.L_stride;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int i = p; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) G[up_tile(p, i)][r] = Lk[(size_t)(16 * p + hi + 4 * r) * DPAD + 16 * i + lc];
  }
  const int ng = (ap.S + 15) / 16;
  double zf[KK], zn[KK];
  {
    const double* __restrict__ src = ap.zf + (size_t)c * KK * 64 + lane;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) zf[kk] = src[kk * 64];
  }
  for (int g = 0; g < ng; ++g) {
    {  // next group's fragments fly during this group's MFMAs (clamped: the last group re-reads itself)
      const int gn = g + 1 < ng ? g + 1 : g;
      const double* __restrict__ src = ap.zf + ((size_t)gn * ap.NP + c) * KK * 64 + lane;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) zn[kk] = src[kk * 64];
    }
    double4_t acc[4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) acc[tj] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int tj = p; tj < 4; ++tj) acc[tj] = mfma_f64(zf[4 * p + r], G[up_tile(p, tj)][r], acc[tj]);  // (tiles below the block diagonal are zero)
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int so = g * 16 + hi + 4 * r;
        if (so < ap.S) ap.inc[((size_t)so * ap.NP + c) * DPAD + 16 * tj + lc] = acc[tj][r];
      }
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) zf[kk] = zn[kk];
  }
}

template <int DPAD, bool APPLY = true>
__global__ void __launch_bounds__(64, 2) k_chol_apply_blk(const CholArgs ca, const ApplyArgs ap) {
  static_assert(DPAD == 64, "the blocked swap is written for four 16-column panels");
  constexpr int NTL = am_tiles<DPAD>();
  const int lane = threadIdx.x, lc = lane & 15, hi = lane >> 4;
  const int64_t c = blockIdx.x;
  double G[10][4];  // upper tiles of Sigma, then of U
  if (c < ca.N) {
    const double* __restrict__ sig = ca.am_sigma + (size_t)c * NTL * 256;
    const int d = ca.d;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int i = p; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * p + hi + 4 * r, col = 16 * i + lc;
          // stored: the tiles on or below the diagonal, [tile(ti, tj)][r][lane] = Sigma[16 ti + hi + 4 r][16 tj + lc]; an upper
          // tile is the transpose of its mirror image (Sigma is symmetric: bitwise, k_adapt computes x_i x_j = x_j x_i)
          const int off = (p == i) ? ((p * (p + 1) / 2 + p) * 4 + r) * 64 + lane : am_sigma_offset(col, row);
          double v = sig[off];
          if (row >= d || col >= d) v = (row == col) ? 1.0 : 0.0;  // padded rows / columns: identity
          G[up_tile(p, i)][r] = v;
        }
  }
  chol_apply_tiles<DPAD, APPLY>(ca, ap, c, G);
}

// Period boundary of the single-level AdaptiveMetropolis pipeline in ONE launch (round 4): the moment recursion over the block's
// states, then C <- Sigma and the next block's increments from the tiles the recursion has just left in registers.  As two launches
// (k_adapt, k_chol_apply_blk) Sigma went out to HBM (80 MiB at 4096 chains) and straight back in through a transposing gather (the
// factorisation wants the upper tiles, k_adapt keeps the lower ones), with a kernel boundary in between.  Here the upper tiles are
// formed by a 16 x 16 register transpose of their mirror images (ds_bpermute; Sigma is bitwise symmetric), and Sigma is stored
// once, for the next period's recursion.  Same arithmetic, same results bit for bit (tests/test_gpu_switches.py).
template <int DPAD, bool CIRC = false>
__global__ void __launch_bounds__(64, 2) k_adapt_chol_apply(const AdaptArgs a, const CholArgs ca, const ApplyArgs ap) {
  static_assert(DPAD == 64, "the blocked swap is written for four 16-column panels");
  int lane = threadIdx.x;
  const int64_t c = blockIdx.x;
  double G[10][4];
  if (c < a.N) {
    double Sg[am_tiles<DPAD>()][4];
    if constexpr (CIRC) {
      adapt_am_chain_c64<true>(a, c, lane, Sg);
      // (the lane index of everything behind the recursion: one the compiler cannot identify with threadIdx.x, see adapt_am_chain_c64)
      lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      asm volatile("" : "+v"(lane));
      __builtin_assume(lane >= 0 && lane < 64);
    } else {
      adapt_am_chain<DPAD>(a, c, lane, Sg);
    }
    const int lc = lane & 15, hi = lane >> 4;
    adapt_scaling(a, c, lane);
    const int d = ca.d;
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int i = p; i < 4; ++i) {
        double t[4];
        if (p == i) {
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = Sg[p * (p + 1) / 2 + p][r];
        } else {
          // upper tile (p, i) = transpose of the lower tile (i, p): element [hi + 4 r][lc] of the result is element [lc][hi + 4 r]
          // of the source, which sits in lane ((lc & 3), hi + 4 r), register lc >> 2
          const int idx = i * (i + 1) / 2 + p;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int src = (lc & 3) * 16 + hi + 4 * r;
            const double t0 = __shfl(Sg[idx][0], src), t1 = __shfl(Sg[idx][1], src), t2 = __shfl(Sg[idx][2], src), t3 = __shfl(Sg[idx][3], src);
            const int sr = lc >> 2;
            t[r] = sr == 0 ? t0 : (sr == 1 ? t1 : (sr == 2 ? t2 : t3));
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = 16 * p + hi + 4 * r, col = 16 * i + lc;
          G[up_tile(p, i)][r] = (row >= d || col >= d) ? ((row == col) ? 1.0 : 0.0) : t[r];  // padded rows / columns: identity
        }
      }
  }
  chol_apply_tiles<DPAD, true>(ca, ap, c, G, lane);
}

}  // namespace tda
