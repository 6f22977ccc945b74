// DREAM(Z): draw / step / adapt kernels and the archive column-sum reduction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "tda_kernels_mh.h"

namespace tda {

// ------------------------------------------------------------------------------------------------
// DREAM(Z)  (tinyDA/proposal.py:608-852) for single-level chains.
//   k_dreamz_draw   wave per chain: everything make_proposal draws that does not depend on the chain state
//                   (archive row indices, crossover index, subspace mask, (1+e) gamma, eps) for a block of steps
//   k_dreamz_steps  16-chain tile: theta' = theta + mask ((1+e) gamma (sum Z_r1 - sum Z_r2) + eps) with the rows
//                   gathered from the chain's archive in HBM, evaluation (linear model on MFMA, or the
//                   Rosenbrock chain on VALU), accept, record, archive append (proposal.py:794)
//   k_dreamz_adapt  wave per chain: archive column sums catch-up, global scaling, pCR update (proposal.py:797-809)
// RNG contract of DREAM(Z) (round 3; counter = (block, step field, global chain id, stream tag)):
//   stream 4, step field = step, per chain and step:
//     block i < delta : archive rows r1 = (x0*M)>>32, r2 = (x1*(M-1))>>32, r2 += r2>=r1
//     block delta     : mCR by inverse cdf of u53(x0,x1) over pCR, forced index = (x2*d)>>32
//   per parameter j (block = delta + 1 + j), in double precision as the reference draws them (proposal.py:846-847):
//     stream 5, step field = step      : crossover uniform u53(x0, x1), e-uniform u53(x2, x3)
//     stream 7, step field = step >> 1 : eps normals, the Box-Muller map of the proposal normals (tda_philox.h normal_pair); z0 for
//                the even step, z1 for the odd one (the normals actually used are exported to the oracle like every normal of the engine)
//   1.5 generator calls per parameter and step.  Round 3 cut this to 0.75 with 32-bit uniforms and single-precision normals and
//   gained < 5 % (the kernel is bound by the per-chain post-processing, not by the generator): not worth narrower variates than
//   the reference's.
// ------------------------------------------------------------------------------------------------
enum : uint32_t { STREAM_DREAM = 4, STREAM_DREAM_MASK = 5, STREAM_DREAM_E = 6, STREAM_DREAM_EPS = 7 };

// out of line: the polynomial constants of log / sincospi inlined into the unrolled group loop push k_dreamz_draw past its 128 registers
__device__ __attribute__((noinline)) double2 dz_normal_pair_call(uint64_t seed, uint32_t chain, uint32_t step2, uint32_t block) {
  double z0, z1;
  normal_pair(seed, chain, step2, STREAM_DREAM_EPS, block, z0, z1);
  return double2{z0, z1};
}
__device__ __attribute__((noinline)) double dz_log_call(double u) { return log(u); }  // (fused block: the accept uniforms' logarithms, out of line for the same reason)
constexpr int MAX_NCR = 8;
constexpr int MAX_DELTA = 4;
constexpr int MAX_PEERS = 16;

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
  // shared archive (frozen inside the block): the jumps do not depend on the chain states, so this kernel -- parallel over
  // chains AND free of the step kernel's accept/reject chain -- gathers the archive rows and writes the finished jump
  // mask * ((1 + e) * gamma * (sum Z[r1] - sum Z[r2]) + eps) into `coef`; k_dreamz_steps then only adds it (jump_ready)
  const double* arch_shared;  // [cap][DPAD] or null
  // distributed shared archive (tda_engine_set_archive_peers): the archive is NOT replicated -- rank o keeps the rows of its own
  // chains, seg[o] = [M0 shared initial rows][step][n_local chains][DPAD], reachable from every rank (peer-mapped).  Global row
  // r >= M0 is (step s, global chain g) = divmod(r - M0, n_total), owned by rank g / n_local at local row M0 + s n_local + g % n_local.
  int dist_ranks;              // 0 = off
  int dist_me;
  long long dist_M0, dist_nloc, dist_ntot;
  const double* const* seg;    // [dist_ranks] device table of the segments' addresses (a table in memory: indexing an array inside
                               // the kernel arguments per lane would copy the whole argument block to scratch)
};

// The step half of a FUSED block (k_dreamz_draw<32, false, true>, round 5): shared archive, the built-in Rosenbrock model, diagonal
// prior, engine-generated variates, 32 parameters -- BASELINE configs[3].  The draws of a step do not depend on the chain's state and
// the step itself is ~100 instructions: done in the kernel that draws, the finished jumps never go to memory ([S][NP][32] doubles
// written by one kernel and read by the next: a third of the configuration's traffic), and the step's memory-bound tail overlaps the
// next step's generator arithmetic.  Same arithmetic in the same order as k_dreamz_draw -> k_dreamz_steps_wave (a parameter per
// lane here, two per lane there: the pair is added first, then the same rotation tree), so the two paths give the same bits
// (TINYDA_DZ_FUSED=0 is the A/B switch, tests/test_gpu_switches.py).
struct DreamFuseArgs {
  const double* pr_mean;
  const double* pr_pinv;
  const double* pr_lo;  // or null
  const double* pr_hi;
  double logconst, ros_a, ros_b, ros_data, var;
  double* theta;        // [NP][32]
  double* theta_prev;
  double* lp;
  double* ll;
  int32_t* acc_count;
  double* rec_params;   // [S][N][d] or null
  double* rec_stats;
  uint8_t* rec_acc;
  double* blk_states;   // [S][NP][32]: the block's states (the archive's next rows)
};
// sum over the 32 lanes of a chain (two DPP rows) of one value per parameter, in the order k_dreamz_steps_wave adds them: the two
// parameters a lane holds there first, then its 16-lane rotations 8, 4, 2, 1 = pairs 16 lanes apart, then rotations 8, 4, 2 here
__device__ __forceinline__ double dz_sum_chain32(double v) {
  v += dpp_move<0xB1>(v);  // quad_perm [1, 0, 3, 2]: the neighbouring parameter
  {
    const unsigned lo = __double2loint(v), hi = __double2hiint(v);
    const uint2_t a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const uint2_t b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = __hiloint2double(b.x, a.x) + __hiloint2double(b.y, a.y);  // lane l + lane l ^ 16
  }
  v += dpp_move<0x128>(v);  // row_ror:8
  v += dpp_move<0x124>(v);  // row_ror:4
  v += dpp_move<0x122>(v);  // row_ror:2
  return v;
}

// 64 / DPAD chains share a wave (lane = chain-in-wave * DPAD + parameter), so small dimensions do not idle lanes
template <int DPAD>
constexpr int dz_chains_per_wave() {
  return 64 / DPAD;
}

// DIST: the shared archive is distributed over the ranks (a template parameter: the plain kernel stays as it was)
#ifndef DZ_FUSED_WAVES
#define DZ_FUSED_WAVES 3  // (the fused block: 168 registers, 6 spilled at three waves per SIMD; 176 / none at two; 128 / 61 spilled at four)
#endif
template <int DPAD, bool DIST = false, bool FUSED = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FUSED ? DZ_FUSED_WAVES : 4, FUSED ? DZ_FUSED_WAVES : 4))) k_dreamz_draw(const DreamDrawArgs a, const DreamFuseArgs f) {
  static_assert(!FUSED || (DPAD == 32 && !DIST), "the fused block is built for 32 parameters, one archive in this process");
  constexpr int CPW = dz_chains_per_wave<DPAD>();
  const int seg = threadIdx.x / DPAD;
  const int lane = threadIdx.x % DPAD;  // parameter index within the chain
  const int64_t c = (int64_t)blockIdx.x * CPW + seg;  // NP is a multiple of 16 >= CPW: every c < NP
  const bool real_chain = c < a.N;
  const uint32_t gc = (uint32_t)(a.chain_offset + c);
  const bool lj = lane < a.d;
  const double scaling = a.scaling[c];
  // cumulative crossover probabilities of the wave's chains: in LDS, not in 16 registers per lane (the kernel is held to 128)
  __shared__ double s_cdf[CPW][MAX_NCR];
  __shared__ double s_CR[MAX_NCR];  // crossover probability of index k: (k + 1) / nCR (a table lookup per step instead of a division)
  __shared__ int s_rows[CPW][DPAD][2 * MAX_DELTA];  // archive row pairs of the chunk's steps (written by the step's owner lane)
  __shared__ int s_mf[CPW][DPAD];                    // crossover index | forced index << 8
  __shared__ double s_uacc[FUSED ? CPW : 1][FUSED ? DPAD : 1];  // fused block: the accept uniform of the owner lane's step
  __shared__ double s_lacc[FUSED ? CPW : 1][FUSED ? DPAD : 1];  // ... and its logarithm (taken once, by the owner lane: sixteen steps side by side)
  if (lane == 0) {
    double run = 0.0;
    for (int k = 0; k < MAX_NCR; ++k) {
      run += k < a.nCR ? a.pCR[c * MAX_NCR + k] : 0.0;
      s_cdf[seg][k] = run;
    }
  }
  if (threadIdx.x < MAX_NCR) s_CR[threadIdx.x] = (double)(threadIdx.x + 1) / (double)a.nCR;
  __syncthreads();
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  // distributed archive: the ranks' segment addresses in LDS (a per-lane pick from a table in memory would put a second
  // dependent trip to memory in front of every gathered row)
  __shared__ const double* s_seg[MAX_PEERS];
  if constexpr (DIST) {
    if (threadIdx.x < (unsigned)a.dist_ranks) s_seg[threadIdx.x] = a.seg[threadIdx.x];
    __syncthreads();
  }
  // jump scale for every possible subspace size (proposal.py:842-844), lane k - 1 of the chain holds the one for k
  // dimensions: a shuffle per step instead of a square root and a division
  const double gam_tab = scaling * 2.38 / sqrt((double)(2 * a.delta * (lane + 1)));
  const bool philox_normals = a.sub_rep == nullptr;  // wave-uniform
  // fused block: the chain's state, a parameter per lane
  // (what only one step or a rare prior needs stays in memory: the state before the block's last step is stored at that step, support
  // bounds are read where they are tested -- as registers they were the six that did not fit three waves per SIMD)
  double th = 0.0, lpc = 0.0, llc = 0.0, pm = 0.0, pinv = 0.0;
  int nacc = 0;
  if constexpr (FUSED) {
    th = f.theta[c * DPAD + lane];
    lpc = f.lp[c];
    llc = f.ll[c];
    pm = f.pr_mean[lane];
    pinv = f.pr_pinv[lane];
  }
  // per-parameter variates: one block each for the crossover uniforms, the e-uniforms and the eps normals of FOUR steps
  double2 q_z{0.0, 0.0};  // eps normals of the step pair in progress
  constexpr int GS = DPAD < 4 ? DPAD : 4;  // steps per group: their archive rows are requested together, one group ahead
  // The per-chain scalars of a step -- delta row pairs, the crossover draw, the accept uniform: delta + 2 <= 6 Philox blocks and
  // their post-processing (53-bit uniforms, the inverse cdf, two 64-bit products per row pair) -- are worked out ONCE per chain
  // and step: lane l of the chain owns step c0 + l of a chunk of DPAD steps, and the per-step loop fetches what it needs with
  // a shuffle.  (Rounds 1-2 had every lane of the chain repeat that post-processing in every step: 250 vector instructions per
  // lane and step, of which the generator was the smaller part -- profiles/r03_c4_pmc_sq.json.)
  // With a shared archive this kernel also gathers the rows of the jump; the rows of group g + 1 are requested before the
  // per-parameter work of group g, so a gather has a whole group of arithmetic to arrive under (round 2 requested them inside
  // the step that used them: a full memory latency per step).
  struct Group {
    double zd[GS];  // sum Z[r1] - sum Z[r2] of this lane's parameter (shared archive)
    int mf[GS];     // crossover index | forced index << 8
  };
  auto row_of = [&](int r) -> const double* {
    if constexpr (!DIST) return a.arch_shared + (size_t)r * DPAD;
    if (r < a.dist_M0) return a.arch_shared + (size_t)r * DPAD;  // (this rank's own copy of the shared initial rows)
    // 32-bit arithmetic: row indices are ints (the row-pair draw), and a 64-bit division costs ~100 instructions per lane
    const uint32_t nt = (uint32_t)a.dist_ntot, nlc = (uint32_t)a.dist_nloc;
    const uint32_t q = (uint32_t)r - (uint32_t)a.dist_M0, sg = q / nt, g = q - sg * nt;
    const uint32_t o = g / nlc, l = g - o * nlc;
    return s_seg[o] + ((size_t)a.dist_M0 + (size_t)sg * nlc + l) * DPAD;
  };
  for (int c0 = 0; c0 < a.S; c0 += DPAD) {
    // ---- chain-level draws of steps c0 .. c0 + DPAD - 1: this lane's step is c0 + lane ----
    __syncthreads();  // (one wave per workgroup: orders this chunk's table writes behind the previous chunk's reads)
    {
      const int s = c0 + lane;
      if (s < a.S) {
        const uint32_t step = (uint32_t)(a.step0 + s);
        const int64_t M = a.M_base + (a.grow ? s : 0);
        const size_t row = (size_t)s * a.N + c;  // replay / export row (real chains only)
        // archive row pairs (proposal.py:823-826)
        for (int i = 0; i < a.delta; ++i) {
          int r1, r2;
          if (a.r_rep && real_chain) {
            r1 = a.r_rep[(row * a.delta + i) * 2 + 0];
            r2 = a.r_rep[(row * a.delta + i) * 2 + 1];
          } else {
            const u32x4 x = philox4x32_10(u32x4{(uint32_t)i, step, gc, STREAM_DREAM}, k0, k1);
            r1 = (int)(((uint64_t)x.x * (uint64_t)M) >> 32);
            r2 = (int)(((uint64_t)x.y * (uint64_t)(M - 1)) >> 32);
            r2 += r2 >= r1 ? 1 : 0;
          }
          if constexpr (!FUSED) {  // (the fused block gathers the rows itself and nobody reads the indices)
            a.ridx[((size_t)s * a.NP + c) * (2 * MAX_DELTA) + 2 * i + 0] = r1;
            a.ridx[((size_t)s * a.NP + c) * (2 * MAX_DELTA) + 2 * i + 1] = r2;
          }
          s_rows[seg][lane][2 * i + 0] = r1;
          s_rows[seg][lane][2 * i + 1] = r2;
        }
        // crossover index and the index forced when the subspace is empty (proposal.py:829-839)
        int mcr, forced;
        if (a.mcr_rep && real_chain) {
          mcr = a.mcr_rep[row];
          forced = a.forced_rep[row];
        } else {
          const u32x4 x = philox4x32_10(u32x4{(uint32_t)a.delta, step, gc, STREAM_DREAM}, k0, k1);
          const double uu = u53(x.x, x.y);
          mcr = a.nCR - 1;
          for (int k = a.nCR - 1; k >= 0; --k)
            if (s_cdf[seg][k] > uu) mcr = k;
          forced = (int)(((uint64_t)x.z * (uint64_t)a.d) >> 32);
        }
        s_mf[seg][lane] = mcr | (forced << 8);
        if (s == a.S - 1) a.mcr_last[c] = mcr;  // crossover index of the block's last step (proposal.py:801)
        // accept_uniform(seed, chain, step, level 0): u53 of the first two words of that block
        double u = 0.5;
        if (real_chain) {
          if (a.u_rep) {
            u = a.u_rep[row];
          } else {
            const u32x4 x = philox4x32_10(u32x4{0u, step, gc, STREAM_ACCEPT}, k0, k1);
            u = u53(x.x, x.y);
          }
          if (a.u_export) a.u_export[row] = u;
        }
        if constexpr (!FUSED) a.u[(size_t)s * a.NP + c] = u;
        if constexpr (FUSED) {
          s_uacc[seg][lane] = u;
          s_lacc[seg][lane] = dz_log_call(u);
        }
      }
    }
    __syncthreads();
    const int c1 = a.S - c0 < DPAD ? a.S : c0 + DPAD;  // end of this chunk
    // what the per-step loop needs of steps s0 .. s0 + GS - 1 (from their owner lanes), and the requests for their archive rows
    auto draw_group = [&](int s0, Group& g) {
      int sl[GS];  // the steps' slots in the chunk's tables
#pragma unroll
      for (int q = 0; q < GS; ++q) {
        sl[q] = (s0 + q - c0) & (DPAD - 1);  // (past the chunk's end: some slot, never used)
        g.mf[q] = s_mf[seg][sl[q]];
        g.zd[q] = 0.0;
      }
      if (a.arch_shared) {  // the row requests of the group back to back (nothing between them that they could alias with)
        double zs1[GS], zs2[GS];  // (the order of the sums the step kernel's own gather uses: ascending pair index, r1 and r2 apart)
#pragma unroll
        for (int q = 0; q < GS; ++q) zs1[q] = zs2[q] = 0.0;
        for (int i = 0; i < a.delta; ++i) {
          double z1[GS], z2[GS];
#pragma unroll
          for (int q = 0; q < GS; ++q) {
            const int vq = s0 + q < c1 ? sl[q] : 0;  // (a step past the chunk's end gathers a valid row: harmless, never used)
            z1[q] = row_of(s_rows[seg][vq][2 * i + 0])[lane];
            z2[q] = row_of(s_rows[seg][vq][2 * i + 1])[lane];
          }
#pragma unroll
          for (int q = 0; q < GS; ++q) {
            zs1[q] += z1[q];
            zs2[q] += z2[q];
          }
        }
#pragma unroll
        for (int q = 0; q < GS; ++q) g.zd[q] = zs1[q] - zs2[q];
      }
    };
    Group cur, nxt;
    draw_group(c0, cur);
    for (int s0 = c0; s0 < c1; s0 += GS) {
      if (s0 + GS < c1) draw_group(s0 + GS, nxt);  // its rows arrive under this group's per-parameter arithmetic
#pragma unroll
      for (int q = 0; q < GS; ++q) {
        const int s = s0 + q;
        if (s < c1) {  // (no `break`: the loop must unroll completely, or the group arrays are indexed dynamically and live in scratch)
          if constexpr (FUSED) __builtin_amdgcn_sched_barrier(0);  // (one step's generator arithmetic at a time: hoisted across the four steps of a group it is 60 registers more)
          const uint32_t step = (uint32_t)(a.step0 + s);
          const size_t row = (size_t)s * a.N + c;
          const int mcr = cur.mf[q] & 255, forced = cur.mf[q] >> 8;
          const double CR = s_CR[mcr];  // (mcr + 1) / nCR
          // ---- per-parameter draws ----
          u32x4 q_u{0u, 0u, 0u, 0u};
          if (philox_normals) {
            const uint32_t blk = (uint32_t)(a.delta + 1 + lane);
            q_u = philox4x32_10(u32x4{blk, step, gc, STREAM_DREAM_MASK}, k0, k1);
            if (s == 0 || (step & 1u) == 0u) q_z = dz_normal_pair_call(a.seed, gc, step >> 1, blk);
          }
          double su = 2.0, eu = 0.5, en = 0.0;
          if (lj) {
            if (!philox_normals) {
              if (real_chain) {
                su = a.sub_rep[row * a.d + lane];
                eu = a.e_rep[row * a.d + lane];
                en = a.eps_rep[row * a.d + lane];
              }
            } else {
              su = u53(q_u.x, q_u.y);
              eu = u53(q_u.z, q_u.w);
              en = (step & 1u) ? q_z.y : q_z.x;
            }
            if (a.eps_export && real_chain) a.eps_export[row * a.d + lane] = en;
          }
          bool ind = lj && (su < CR);
          unsigned long long bal = __ballot(ind);
          if (CPW > 1) bal = (bal >> (seg * DPAD)) & ((1ull << (DPAD & 63)) - 1ull);  // this chain's lanes
          int dsub = __popcll(bal);
          if (dsub == 0) {  // proposal.py:838-839
            ind = lane == forced;
            dsub = 1;
          }
          const double gam = __shfl(gam_tab, seg * DPAD + dsub - 1);  // scaling * 2.38 / sqrt(2 delta d'), proposal.py:842-844
          const double e = -a.b + (a.b - (-a.b)) * eu;
          const double eps = 0.0 + a.b_star * en;
          if (lane < DPAD) {
            const double cf = ind ? (1.0 + e) * gam : 0.0, em = ind ? eps : 0.0;
            const size_t o = ((size_t)s * a.NP + c) * DPAD + lane;
            if constexpr (FUSED) {
              // ---- the step (k_dreamz_steps_wave, jump_ready): proposal, prior, Rosenbrock chain, decision, records ----
              const double prp = th + (cf * cur.zd[q] + em);  // proposal.py:850-852
              const double dv = prp - pm;
              double p = dv * dv * pinv;
              if (f.pr_lo && (prp < f.pr_lo[lane] || prp > f.pr_hi[lane])) p = INFINITY;  // uniform components: zero density outside their support
              const double maha = dz_sum_chain32(p);
              const double xnb = __shfl_down(prp, 1, DPAD);
              const double t0 = f.ros_a - prp, t1 = xnb - prp * prp;
              const double fs = dz_sum_chain32(lane + 1 < a.d ? t0 * t0 + f.ros_b * (t1 * t1) : 0.0);
              const double rr = fs - f.ros_data;
              const double ll_n = -0.5 * (rr * rr) / f.var;
              const double lp_n = -0.5 * (f.logconst + maha);
              const double post_n = lp_n + ll_n;
              // u < exp(delta), decided on the logarithms unless they are within 1e-9 of each other (then exactly as the step kernel
              // does): the exponential is ~40 instructions that both chains of the wave would execute at every step
              const double delta = post_n - (lpc + llc);
              const int slot = (s - c0) & (DPAD - 1);  // the step's slot in the chunk's tables (read here: as registers of the look-ahead group they were 32 more)
              const double lu = s_lacc[seg][slot];
              bool acc;
              if (fabs(lu - delta) > 1e-9 || delta != delta) {
                acc = (post_n == post_n) && (lu < delta);
              } else {
                double alpha = exp(delta);
                if (post_n != post_n) alpha = 0.0;
                acc = s_uacc[seg][slot] < alpha;
              }
              if (acc) {
                lpc = lp_n;
                llc = ll_n;
              }
              nacc += acc ? 1 : 0;
              if (lane == 0) {
                if (f.rec_stats) {
                  f.rec_stats[row * 3 + 0] = lpc;
                  f.rec_stats[row * 3 + 1] = llc;
                  f.rec_stats[row * 3 + 2] = lpc + llc;
                }
                if (f.rec_acc) f.rec_acc[row] = acc ? 1 : 0;
              }
              if (s == a.S - 1) f.theta_prev[c * DPAD + lane] = th;  // the state before the block's last step (jumping distance, proposal.py:800)
              th = acc ? prp : th;
              if (f.rec_params && lj) f.rec_params[row * a.d + lane] = th;
              f.blk_states[o] = th;
            } else if (a.arch_shared) {
              a.coef[o] = cf * cur.zd[q] + em;  // the jump itself (proposal.py:850-852)
            } else {
              a.coef[o] = cf;
              a.epsm[o] = em;
            }
          }
        }
      }
      cur = nxt;
    }
  }
  if constexpr (FUSED) {
    f.theta[c * DPAD + lane] = th;
    if (lane == 0) {
      f.lp[c] = lpc;
      f.ll[c] = llc;
      f.acc_count[c] += nacc;
    }
  }
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
  int jump_ready;      // 1: `coef` holds the finished jumps (k_dreamz_draw gathered the shared archive), no gathers here
};

// DENSE: the instance for a dense observation covariance (a template parameter: as a run-time branch it cost the plain instances
// 12 / 30 more spilled registers)
// (waves per SIMD: at 64 parameters -- and at 32 with the dense quadratic form -- the tile's state needs up to ~400 registers per
// wave; at two waves per SIMD those instances spilled 154 / 163 / 6 registers with reloads inside the step loop)
#ifndef DZ_TILE_WAVES_PER_EU
#define DZ_TILE_WAVES_PER_EU ((DPAD >= 64 || (DENSE && DPAD >= 32)) ? 1 : 2)
#endif
template <int DPAD, bool DENSE = false>
__global__ void __launch_bounds__(256, DZ_TILE_WAVES_PER_EU) k_dreamz_steps(const DreamStepArgs a) {
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
  // dense observation covariance (round 4): the tile's residuals [16][m_pad + 2] for the Sigma^-1 quadratic form on the matrix cores
  double* s_R = s_py + (prior_dense ? a.pr.ncb * 16 : 0);

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
  double jnext[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) jnext[e] = (a.jump_ready && active) ? a.coef[(size_t)gct * DPAD + q_ * EPT + e] : 0.0;
  double* arch_c = a.shared ? a.arch : a.arch + (size_t)gct * a.cap * DPAD;
  const FragSrc fbase = frag_src(a.lv.Apk, lane);
  __syncthreads();

  for (int s = 0; s < a.S; ++s) {
    double2 f0[KS / 2], f1[KS / 2];
    if (linear) {
      frag_load<DPAD>(fbase, wave, a.lv.ncb, f0);
      frag_load<DPAD>(fbase, wave + 4, a.lv.ncb, f1);
    }
    // ---- proposal (proposal.py:850-852) ----
    if (a.jump_ready) {
      if (active) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          prp[e] = cur[e] + jnext[e];
          s_prop[c * LDP + q_ * EPT + e] = prp[e];
          if (s + 1 < a.S) jnext[e] = a.coef[((size_t)(s + 1) * a.NP + gct) * DPAD + q_ * EPT + e];  // flies under this step
        }
      }
    } else if (active) {
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
      if (a.pr.lo) {  // uniform components: zero density outside their support
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
          if (th[kk] < a.pr.lo[4 * kk + hi] || th[kk] > a.pr.hi[4 * kk + hi]) p = INFINITY;
      }
      p = sum_rows(p);
      maha = p;
    } else {
      const FragSrc pbase = frag_src(a.pr.Wpk, lane);
      double2 p0[KS / 2], p1[KS / 2];
      frag_load<DPAD>(pbase, wave, a.pr.ncb, p0);
      frag_load<DPAD>(pbase, wave + 4, a.pr.ncb, p1);
      double p = level_sse_partial<DPAD, 0>(a.pr.Wpk, a.pr.ncb, s_py, nullptr, th, wave, lane, p0, p1);
      p = sum_rows(p);
      if (lane < 16) s_redp[wave * 16 + lane] = p;
    }
    // ---- likelihood ----
    double ll_n;
    if constexpr (DENSE) {  // DefaultGaussianLogLike (distributions.py:246-301) under DREAM(Z): as k_mh_steps / k_ml_steps evaluate it
      const int RSd = a.lv.m_pad + 2;
      (void)level_sse_partial<DPAD, 2>(a.lv.Apk, a.lv.ncb, s_y, s_R + lc * RSd, th, wave, lane, f0, f1);
      __syncthreads();
      double qs = sum_rows(dense_quadform<4>(a.lv.Ppk, a.lv.ncb, a.lv.m_pad, s_R, RSd, wave, lane));
      if (lane < 16) s_red[wave * 16 + lane] = qs;
      __syncthreads();
      ll_n = -0.5 * (((s_red[lc] + s_red[16 + lc]) + s_red[32 + lc]) + s_red[48 + lc]);
    } else if (linear) {
      double sse = diag ? level_sse_partial<DPAD, 1>(a.lv.Apk, a.lv.ncb, s_y, s_w, th, wave, lane, f0, f1)
                        : level_sse_partial<DPAD, 0>(a.lv.Apk, a.lv.ncb, s_y, nullptr, th, wave, lane, f0, f1);
      sse = sum_rows(sse);
      if (lane < 16) s_red[wave * 16 + lane] = sse;
      __syncthreads();
      const double tot = ((s_red[lc] + s_red[16 + lc]) + s_red[32 + lc]) + s_red[48 + lc];
      ll_n = diag ? -0.5 * tot : -0.5 * tot / a.lv.var;
    } else {
      // Rosenbrock chain: f = sum_i (a - x_i)^2 + b (x_{i+1} - x_i^2)^2 ; loglike = -0.5 (f - data)^2 / var
      // the four lanes (lc, hi) of a chain take every fourth term; sum_rows adds the partial sums
      double f = 0.0;
      for (int i = hi; i + 1 < a.d; i += 4) {
        const double x0 = s_prop[lc * LDP + i], x1 = s_prop[lc * LDP + i + 1];
        const double t0 = a.ros_a - x0, t1 = x1 - x0 * x0;
        f += t0 * t0 + a.ros_b * (t1 * t1);
      }
      f = sum_rows(f);
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

// sum over the LPC lanes of a chain, every lane receives the same bits (rotations / xor butterfly: commutative pairs)
template <int LPC>
__device__ __forceinline__ double sum_chain_lanes(double v) {
  if constexpr (LPC == 16) {
    v += dpp_move<0x128>(v);  // row_ror:8
    v += dpp_move<0x124>(v);  // row_ror:4
    v += dpp_move<0x122>(v);  // row_ror:2
    v += dpp_move<0x121>(v);  // row_ror:1
  } else {
#pragma unroll
    for (int off = LPC / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  }
  return v;
}

// The Rosenbrock example model under a diagonal prior needs neither the matrix cores nor a 16-chain tile: k_dreamz_steps
// spends its step in two workgroup barriers and an LDS round trip while its four waves repeat the same evaluation
// (2.4 us per step of 8192 chains, 2 waves per SIMD).  Here a chain is 16 lanes of a wave (8 at DPAD = 8) holding DPAD / 16
// parameters each; the prior and the model reduce over those lanes with DPP rotations, the next parameter of the Rosenbrock
// coupling comes from the neighbouring lane, and the only memory traffic of a step is its algorithmic traffic (jump in, record
// / archive row out), prefetched one step ahead.  Same DreamStepArgs, same records; log-densities equal k_dreamz_steps' to
// rounding (a different summation order).
template <int DPAD>
__global__ void __launch_bounds__(64) k_dreamz_steps_wave(const DreamStepArgs a) {
  constexpr int LPC = DPAD >= 16 ? 16 : DPAD;
  constexpr int EPT = DPAD / LPC;
  constexpr int CPW = 64 / LPC;
  const int lane = threadIdx.x;
  const int q = lane % LPC;
  const int64_t c = (int64_t)blockIdx.x * CPW + lane / LPC;  // NP is a multiple of 16: every c < NP
  const bool real = c < a.N;
  const int j0 = q * EPT;
  double pm[EPT], pinv[EPT], blo[EPT], bhi[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    pm[e] = a.pr.mean[j0 + e];
    pinv[e] = a.pr.pinv[j0 + e];
    blo[e] = a.pr.lo ? a.pr.lo[j0 + e] : -INFINITY;
    bhi[e] = a.pr.lo ? a.pr.hi[j0 + e] : INFINITY;
  }
  double cur[EPT], prp[EPT], prev[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    cur[e] = a.theta[c * DPAD + j0 + e];
    prev[e] = cur[e];
  }
  double lp = a.lp[c], ll = a.ll[c];
  int nacc = 0;
  double* arch_c = a.shared ? a.arch : a.arch + (size_t)c * a.cap * DPAD;
  // Finished jumps (k_dreamz_draw): the jumps and uniforms of DZ_SB steps come into LDS in ONE round of loads, and the step loop
  // itself issues no load.  With a load per step, its wait was a wait for the record stores of the step before as well -- loads
  // and stores count on one counter (vmcnt) on gfx9 and the stores sit in branches the compiler cannot count, so it waits for
  // zero: 1.7 us per step of 8192 chains, a store acknowledgement each, for 150 instructions of arithmetic.
  constexpr int DZ_SB = 16;
  __shared__ double s_jump[DZ_SB * CPW * DPAD];
  __shared__ double s_unif[DZ_SB * CPW];
  const int cw = lane / LPC;
  double unext = a.jump_ready ? 0.0 : a.u[c];
  // shared (frozen) archive gathered here: the rows of step s + 1 and the row indices of step s + 2 are requested while step s
  // computes -- two dependent trips to HBM per step otherwise (5 us per step of 8192 chains instead of 1.8)
  const bool ahead = !a.jump_ready && a.shared;
  int rn[2 * MAX_DELTA];
  double g1[MAX_DELTA][EPT], g2[MAX_DELTA][EPT], cfn[EPT], emn[EPT];
  auto request_rows = [&](int s1) {  // rows, coefficient and noise of step s1 from the indices in rn
#pragma unroll
    for (int i = 0; i < MAX_DELTA; ++i)
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        g1[i][e] = i < a.delta ? arch_c[(size_t)rn[2 * i] * DPAD + j0 + e] : 0.0;
        g2[i][e] = i < a.delta ? arch_c[(size_t)rn[2 * i + 1] * DPAD + j0 + e] : 0.0;
      }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const size_t o = ((size_t)s1 * a.NP + c) * DPAD + j0 + e;
      cfn[e] = a.coef[o];
      emn[e] = a.epsm[o];
    }
  };
  auto request_indices = [&](int s2) {
#pragma unroll
    for (int i = 0; i < 2 * MAX_DELTA; ++i) rn[i] = (i >> 1) < a.delta ? a.ridx[((size_t)s2 * a.NP + c) * (2 * MAX_DELTA) + i] : 0;
  };
  if (ahead) {
    request_indices(0);
    request_rows(0);
    if (a.S > 1) request_indices(1);
  }
  for (int s = 0; s < a.S; ++s) {
    double u = unext;
    if (a.jump_ready) {
      const int k = s % DZ_SB;
      if (k == 0) {  // (wave-uniform) the next DZ_SB steps: every lane its own elements, lane 0 of a chain its uniforms
        const int n = a.S - s < DZ_SB ? a.S - s : DZ_SB;
        double jl[DZ_SB][EPT], ul[DZ_SB];
#pragma unroll
        for (int i = 0; i < DZ_SB; ++i) {
#pragma unroll
          for (int e = 0; e < EPT; ++e) jl[i][e] = i < n ? a.coef[((size_t)(s + i) * a.NP + c) * DPAD + j0 + e] : 0.0;
          ul[i] = (i < n && q == 0) ? a.u[(size_t)(s + i) * a.NP + c] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < DZ_SB; ++i) {
#pragma unroll
          for (int e = 0; e < EPT; ++e) s_jump[(i * CPW + cw) * DPAD + j0 + e] = jl[i][e];
          if (q == 0) s_unif[i * CPW + cw] = ul[i];
        }
      }
      u = s_unif[k * CPW + cw];
#pragma unroll
      for (int e = 0; e < EPT; ++e) prp[e] = cur[e] + s_jump[(k * CPW + cw) * DPAD + j0 + e];
    } else if (s + 1 < a.S) {
      unext = a.u[(size_t)(s + 1) * a.NP + c];
    }
    // ---- proposal (proposal.py:850-852) ----
    if (a.jump_ready) {
    } else if (ahead) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        double z1 = 0.0, z2 = 0.0;  // rows beyond delta are zeros: the sums are those of the loop over delta below
#pragma unroll
        for (int i = 0; i < MAX_DELTA; ++i) {
          z1 += g1[i][e];
          z2 += g2[i][e];
        }
        prp[e] = cur[e] + (cfn[e] * (z1 - z2) + emn[e]);
      }
      if (s + 1 < a.S) {
        request_rows(s + 1);
        if (s + 2 < a.S) request_indices(s + 2);
      }
    } else {
      double z1[EPT], z2[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) z1[e] = z2[e] = 0.0;
      for (int i = 0; i < a.delta; ++i) {
        const int r1 = a.ridx[((size_t)s * a.NP + c) * (2 * MAX_DELTA) + 2 * i + 0];
        const int r2 = a.ridx[((size_t)s * a.NP + c) * (2 * MAX_DELTA) + 2 * i + 1];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          z1[e] += arch_c[(size_t)r1 * DPAD + j0 + e];
          z2[e] += arch_c[(size_t)r2 * DPAD + j0 + e];
        }
      }
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const size_t o = ((size_t)s * a.NP + c) * DPAD + j0 + e;
        prp[e] = cur[e] + (a.coef[o] * (z1[e] - z2[e]) + a.epsm[o]);
      }
    }
    // ---- prior ----
    double p = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const double dv = prp[e] - pm[e];
      p += dv * dv * pinv[e];
      if (prp[e] < blo[e] || prp[e] > bhi[e]) p = INFINITY;  // uniform components: zero density outside their support
    }
    const double maha = sum_chain_lanes<LPC>(p);
    // ---- Rosenbrock chain: f = sum_i (a - x_i)^2 + b (x_{i+1} - x_i^2)^2 ; loglike = -0.5 (f - data)^2 / var ----
    const double xnb = __shfl_down(prp[0], 1, LPC);  // first parameter of the next lane (unused by the chain's last lane)
    double f = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const double x0 = prp[e], x1 = e + 1 < EPT ? prp[e + 1 < EPT ? e + 1 : e] : xnb;
      const double t0 = a.ros_a - x0, t1 = x1 - x0 * x0;
      if (j0 + e + 1 < a.d) f += t0 * t0 + a.ros_b * (t1 * t1);
    }
    f = sum_chain_lanes<LPC>(f);
    const double r = f - a.ros_data;
    const double ll_n = -0.5 * (r * r) / a.lv.var;
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
    if (q == 0 && real) {
      const size_t rr = (size_t)s * a.N + c;
      if (a.rec_stats) {
        a.rec_stats[rr * 3 + 0] = lp;
        a.rec_stats[rr * 3 + 1] = ll;
        a.rec_stats[rr * 3 + 2] = lp + ll;
      }
      if (a.rec_acc) a.rec_acc[rr] = acc ? 1 : 0;
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      if (s == a.S - 1) prev[e] = cur[e];
      cur[e] = acc ? prp[e] : cur[e];
      const int j = j0 + e;
      if (a.rec_params && real && j < a.d) a.rec_params[((size_t)s * a.N + c) * a.d + j] = cur[e];
      // archive append (proposal.py:794): per-chain archives see it at once, the shared one after the block
      if (!a.shared) arch_c[(size_t)(a.M_base + s) * DPAD + j] = cur[e];
      if (a.blk_states) a.blk_states[((size_t)s * a.NP + c) * DPAD + j] = cur[e];
    }
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    a.theta[c * DPAD + j0 + e] = cur[e];
    a.theta_prev[c * DPAD + j0 + e] = prev[e];
  }
  if (q == 0) {
    a.lp[c] = lp;
    a.ll[c] = ll;
    a.acc_count[c] += nacc;
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
  // below a Delayed Acceptance / MLDA hierarchy the accept-flag window of the base proposal also holds the alignment entries
  // the upper levels append (chain.py:363,389,397; proposal.py:1486): a ring of flags instead of a counter (as k_adapt)
  const uint8_t* ring;     // [ring_P][NP] or null
  int ring_P;
  int64_t ring_hi;         // absolute list position just after the boundary base step's own flag
};

// column sums / sums of squares of rows [row0 + 256 b, row0 + 256 (b+1)) of a row-major [.][DPAD] matrix.  The 64 / DPAD lane
// groups of the wave take every (64 / DPAD)-th row, sixteen loads in flight per lane (the kernel is bound by the latency of its
// loads); the groups are added in ascending order, so the result depends on the chunk only.
constexpr int COLSUM_CHUNK = 256;
template <int DPAD>
__global__ void __launch_bounds__(64) k_colsum_partial(const double* __restrict__ m, int64_t row0, int64_t nrows,
                                                       double* __restrict__ partial) {
  const int lane = threadIdx.x;
  const int64_t b = blockIdx.x;
  const int64_t lo = b * COLSUM_CHUNK, hi = lo + COLSUM_CHUNK < nrows ? lo + COLSUM_CHUNK : nrows;
  if constexpr (DPAD % 2 == 0 && DPAD >= 2) {
    // two columns per lane (16-byte loads): 2 * 64 / DPAD rows per instruction.  Round 5 (C4: 210 MB per adaptation boundary): the
    // 8-byte form moved 3.8 TB/s.  The sums of a chunk are formed row group by row group, then over the groups in ascending order:
    // a fixed order for a given archive, whoever computes it.
    constexpr int LPR = DPAD / 2, G = 64 / LPR;  // lanes per row, rows per instruction
    const int cp = lane % LPR, g = lane / LPR;
    double zs0 = 0.0, zs1 = 0.0, zq0 = 0.0, zq1 = 0.0;
    if (g < G) {
#pragma unroll 32
      for (int64_t r = lo + g; r < hi; r += G) {
        const double2 z = *reinterpret_cast<const double2*>(m + (size_t)(row0 + r) * DPAD + 2 * cp);
        zs0 += z.x;
        zs1 += z.y;
        zq0 += z.x * z.x;
        zq1 += z.y * z.y;
      }
    }
    double t0 = 0.0, t1 = 0.0, q0 = 0.0, q1 = 0.0;
#pragma unroll
    for (int k = 0; k < G; ++k) {
      t0 += __shfl(zs0, cp + LPR * k);
      t1 += __shfl(zs1, cp + LPR * k);
      q0 += __shfl(zq0, cp + LPR * k);
      q1 += __shfl(zq1, cp + LPR * k);
    }
    if (lane < LPR) {
      partial[((size_t)b * 2 + 0) * DPAD + 2 * lane] = t0;
      partial[((size_t)b * 2 + 0) * DPAD + 2 * lane + 1] = t1;
      partial[((size_t)b * 2 + 1) * DPAD + 2 * lane] = q0;
      partial[((size_t)b * 2 + 1) * DPAD + 2 * lane + 1] = q1;
    }
  } else {
    constexpr int G = 64 / DPAD;
    const int col = lane % DPAD, g = lane / DPAD;
    double zs = 0.0, zq = 0.0;
#pragma unroll 64
    for (int64_t r = lo + g; r < hi; r += G) {
      const double z = m[(size_t)(row0 + r) * DPAD + col];
      zs += z;
      zq += z * z;
    }
    double ts = 0.0, tq = 0.0;
#pragma unroll
    for (int k = 0; k < G; ++k) {
      ts += __shfl(zs, col + DPAD * k);
      tq += __shfl(zq, col + DPAD * k);
    }
    if (lane < DPAD) {
      partial[((size_t)b * 2 + 0) * DPAD + lane] = ts;
      partial[((size_t)b * 2 + 1) * DPAD + lane] = tq;
    }
  }
}
// chunk sums -> sums of COLSUM_FOLD consecutive chunks, in order (round 5: one workgroup adding 3 200 chunks of a C4 boundary was bound by ITS
// compute unit's load path: 3.3 MB through one L1 = 24 us; folded over ~25 workgroups first it is two short launches)
constexpr int COLSUM_FOLD = 128;
template <int DPAD>
__global__ void __launch_bounds__(64) k_colsum_fold(const double* __restrict__ partial, int64_t npart, double* __restrict__ folded) {
  constexpr int G = 64 / DPAD;
  const int lane = threadIdx.x;
  const int col = lane % DPAD, g = lane / DPAD;
  const int64_t lo = (int64_t)blockIdx.x * COLSUM_FOLD, hi = lo + COLSUM_FOLD < npart ? lo + COLSUM_FOLD : npart;
  double zs = 0.0, zq = 0.0;
#pragma unroll 32
  for (int64_t b = lo + g; b < hi; b += G) {
    zs += partial[((size_t)b * 2 + 0) * DPAD + col];
    zq += partial[((size_t)b * 2 + 1) * DPAD + col];
  }
  double ts = 0.0, tq = 0.0;
#pragma unroll
  for (int k = 0; k < G; ++k) {
    ts += __shfl(zs, col + DPAD * k);
    tq += __shfl(zq, col + DPAD * k);
  }
  if (lane < DPAD) {
    folded[((size_t)blockIdx.x * 2 + 0) * DPAD + lane] = ts;
    folded[((size_t)blockIdx.x * 2 + 1) * DPAD + lane] = tq;
  }
}
constexpr int COLSUM_FINAL_WAVES = 16;
template <int DPAD>
__global__ void __launch_bounds__(64 * COLSUM_FINAL_WAVES) k_colsum_final(const double* __restrict__ partial, int64_t npart,
                                                                          double* __restrict__ zsum, double* __restrict__ zsq) {
  constexpr int G = 64 / DPAD;
  __shared__ double s_part[COLSUM_FINAL_WAVES * G][2][DPAD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = lane % DPAD, g = lane / DPAD;
  {
    double zs = 0.0, zq = 0.0;
#pragma unroll 32
    for (int64_t b = w * G + g; b < npart; b += COLSUM_FINAL_WAVES * G) {
      zs += partial[((size_t)b * 2 + 0) * DPAD + col];
      zq += partial[((size_t)b * 2 + 1) * DPAD + col];
    }
    s_part[w * G + g][0][col] = zs;
    s_part[w * G + g][1][col] = zq;
  }
  __syncthreads();
  if (w == 0 && lane < DPAD) {
    double zs = zsum[lane], zq = zsq[lane];
    for (int k = 0; k < COLSUM_FINAL_WAVES * G; ++k) {
      zs += s_part[k][0][lane];
      zq += s_part[k][1][lane];
    }
    zsum[lane] = zs;
    zsq[lane] = zq;
  }
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
      int hits = 0;
      if (a.ring) {
        for (int i = 1; i <= a.period; ++i) hits += a.ring[(size_t)((a.ring_hi - i) % a.ring_P) * a.NP + c];
      } else {
        hits = a.acc_count[c];
      }
      const double rate = (double)hits / (double)a.period;
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

}  // namespace tda
