// EXPERIMENT, not built: producer / consumer waves for the Delayed-Acceptance tile (round 3).  Parity-green, zero scratch,
// 180 registers -- and SLOWER than k_da_steps: see tools/experimental/README.md.
#pragma once
#include "tda_kernels_ml.h"
namespace tda {
// ------------------------------------------------------------------------------------------------
// The same hierarchy as k_da_steps (same MLArgs, records, RNG contract, level actions) with the eight waves of the tile in
// two ROLES.  A coarse step needs the model outputs of theta' = keep theta + s inc, and G = A (s inc) does not depend on any
// decision, so
//   PRODUCERS (waves 4-7, one per SIMD; the coarse operator in registers, 2 RB 16-row blocks each) run one step AHEAD:
//             G[s + 1] = A (s inc_{s+1}) for the 16 chains of the tile on the matrix cores, written to LDS (two rows in turn);
//   CONSUMERS (waves 0-3, one per SIMD; 16 lanes per chain: 8 RB model outputs and DPAD / 16 parameters per lane) take
//             step s from G[s]:  F' = keep F + G,  sse = sum w (F' - y)^2,  prior of theta',  decision, records -- both
//             reductions are four DPP rotations inside the 16 lanes, nothing leaves the wave;
// one barrier per step, which neither role normally waits at for long.  The matrix pipe works on step s + 1 while step s is
// reduced and decided (fp64 vector and matrix instructions share the lanes, but two thirds of the consumer's instructions
// are moves, selects, LDS and store traffic), and the roles are separate code paths, so the registers of one do not count
// against the other (168 / 190 instead of 255 with spills).  k_da_steps: 4 000 (m0 = 128) .. 11 000 (m0 = 256) cycles per
// coarse step against 2 048 / 4 096 of matrix work; here a step costs its matrix work at ~85 % of the pipe.
// Model outputs of the states the upper levels hold are kept beside the coarse chain's (Fs): a rejected level step puts its
// outputs back with its state, so nothing is re-derived after a level action; the outputs are re-derived from theta
// (ANCHORED, one more producer column) at every launch, so rounding cannot accumulate beyond one block.
// Level actions: all eight waves stream the level's fragments (as k_da_steps), only the consumers decide.
// LDS slot p of a chain's output row: block (p >> 4), then hi = (p >> 2) & 3, r = p & 3 <-> output (p & ~15) + hi + 4 r, so
// that an MFMA lane stores its four accumulator rows as one 32-byte piece and a consumer lane reads 32 contiguous bytes.
// ------------------------------------------------------------------------------------------------
template <int DPAD>
__host__ __device__ constexpr int da_pc_lds_doubles(int stage_total, int RB) {
  return 16 * (DPAD + 2) + 2 * 64 * (DPAD / 4 + 2) + 2 * 16 * (128 * RB + 2) + 8 * 16 + 2 * 2 * 16 + 4 * DPAD + 2 * 128 * RB + 16 * 256 + stage_total;
}

template <int CTRL>
__device__ __forceinline__ double dpp_rot(double v) {  // (bound_ctrl: every lane of a row rotation is valid, no old value to keep)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum_row16(double s) {  // sum over the 16 lanes of a DPP row, in every lane
  s += dpp_rot<0x128>(s);  // row_ror:8
  s += dpp_rot<0x124>(s);  // row_ror:4
  s += dpp_rot<0x122>(s);  // row_ror:2
  s += dpp_rot<0x121>(s);  // row_ror:1
  return s;
}

// The residual sums of one level for the 16 states staged in s_prop, this wave's share into s_red -- OUT OF LINE for the
// consumer waves: the fragment pipeline (two blocks + the states = 96 registers) would otherwise count against the registers
// their step loop keeps alive; as a call, what does not fit is saved around the call, once per level action.
template <int DPAD>
__device__ __attribute__((noinline)) void da_level_sums(const double* Apk, int ncb, int diag, const double* s_y, double* s_w,
                                                        const double* s_prop, double* s_red, int wave, int lane) {
  constexpr int KS = DPAD / 4, K2 = DPAD / 8, LDP = DPAD + 2;
  const int lc = lane & 15, hi = lane >> 4;
  double2 fa[K2];
  frag_load_buf<DPAD>(frag_src(Apk, lane), wave < ncb ? wave : ncb - 1, fa);
  double th[KS];
#pragma unroll
  for (int kk = 0; kk < KS; ++kk) th[kk] = s_prop[lc * LDP + 4 * kk + hi];
  double sq = diag ? level_sse_single<DPAD, 1, 8>(Apk, ncb, s_y, s_w, th, wave, lane, fa)
                   : level_sse_single<DPAD, 0, 8>(Apk, ncb, s_y, nullptr, th, wave, lane, fa);
  sq = sum_rows(sq);
  if (lane < 16) s_red[wave * 16 + lane] = sq;
}

#ifdef TDA_DA_TRACE
#define PC_STAMP(i) \
  if (blockIdx.x == 0 && lane == 0 && s < 128) g_da_trace[((size_t)s * 8 + wave) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define PC_STAMP(i)
#endif
template <int DPAD, int RB, bool PCN, int NZ0, int NLEV = 2>
__global__ void __launch_bounds__(512, 2) k_da_pc(const MLArgs a) {
  static_assert(NLEV == 2 || NLEV == 3, "two-level Delayed Acceptance or three-level MLDA");
  constexpr int NPAIR = NLEV * (NLEV - 1) / 2;
  constexpr int NW = 8, NT = 64 * NW;
  constexpr int KS = DPAD / 4, K2 = DPAD / 8, LDP = DPAD + 2, RSX = KS + 2;
  constexpr int NB = 2 * RB;          // operator blocks per producer wave: pw + 4 i
  constexpr int NS = 8 * RB;          // output slots per consumer lane: 64 i2 + 4 j + r
  constexpr int LDG = 128 * RB + 2;   // row of one chain's outputs of one step (+ 2: the 16 chains of an MFMA store hit 16 bank groups)
  constexpr int EPT = DPAD >= 16 ? DPAD / 16 : 1;  // parameters per consumer lane
  constexpr bool dg0 = NZ0 != 0;
  constexpr bool aemd0 = NZ0 == 2;
  constexpr bool is_pcn = PCN;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* s_prop = smem;                    // [16][LDP] states in natural order (level actions, anchors)
  double* s_inc = s_prop + 16 * LDP;        // [2][64][RSX] scaled increments of steps s, s + 1 in fragment order
  double* s_G = s_inc + 2 * 64 * RSX;       // [2][16][LDG] A (s inc) of steps s, s + 1 in slot order (prologue: set 1 takes the anchors)
  double* s_red = s_G + 2 * 16 * LDG;       // [NW][16] (level actions)
  double* s_lu = s_red + 8 * 16;            // [2][2][16] log-uniforms and uniforms of steps s, s + 1
  double* s_pm = s_lu + 2 * 2 * 16;
  double* s_pinv = s_pm + DPAD;
  double* s_lo = s_pinv + DPAD;             // support bounds of the prior (+-inf without)
  double* s_hi = s_lo + DPAD;
  double* s_ys = s_hi + DPAD;               // [128 RB] data and weights of the coarse level in slot order
  double* s_ws = s_ys + 128 * RB;
  double* s_Fs = s_ws + 128 * RB;           // [(NLEV - 1) NS / 2][256][2] model outputs of the upper levels' states, per consumer lane
  double* s_stage = s_Fs + 16 * 256;
  static_assert((NLEV - 1) * 8 * RB <= 16, "s_Fs holds 16 doubles per consumer lane");

  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t tile = blockIdx.x;
  const int lc = lane & 15, hi = lane >> 4;
  const int ncb0 = a.lv[0].ncb;
  const int L0 = a.sl[0];

#pragma unroll
  for (int k = 0; k < NLEV; ++k) {
    if (k > 0 && !a.cascade) break;  // (host-sequenced level actions: only the base level is evaluated, and staged)
    for (int i = tid; i < a.lv[k].m_pad; i += NT) {
      s_stage[a.lds_y[k] + i] = a.lv[k].ytil[i];
      if (a.lv[k].noise_kind == 1) s_stage[a.lds_w[k] + i] = a.lv[k].w[i];
    }
  }
  for (int i = tid; i < DPAD; i += NT) {
    s_pm[i] = a.pr.mean[i];
    s_pinv[i] = a.pr.pinv[i];
    s_lo[i] = a.pr.lo ? a.pr.lo[i] : -INFINITY;
    s_hi[i] = a.pr.lo ? a.pr.hi[i] : INFINITY;
  }
  for (int p = tid; p < 128 * RB; p += NT) {
    const int o = (p & ~15) + ((p >> 2) & 3) + 4 * (p & 3);
    const bool valid = (p >> 4) < ncb0;
    s_ys[p] = valid ? a.lv[0].ytil[o] : 0.0;
    s_ws[p] = (NZ0 == 1) ? (valid ? a.lv[0].w[o] : 0.0) : 1.0;
  }
  for (int i = tid; i < 2 * 16 * LDG; i += NT) s_G[i] = 0.0;  // slots of operator blocks that do not exist stay zero
  __syncthreads();

  auto chainmm = [&](const double2 (&f)[K2], const double2 (&b)[K2]) {
    double4_t g = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < K2; ++k) {
      g = mfma_f64(f[k].x, b[k].x, g);
      g = mfma_f64(f[k].y, b[k].y, g);
    }
    return g;
  };
  // the residual sums of level q at the states staged in s_prop: every wave's share into s_red (between two barriers of the caller)
  auto level_sums = [&](int q, double2 (&fa)[K2]) {
    const LevelDev& L = a.lv[q];
    double th[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) th[kk] = s_prop[lc * LDP + 4 * kk + hi];
    double sq = L.noise_kind == 1 ? level_sse_single<DPAD, 1, NW>(L.Apk, L.ncb, s_stage + a.lds_y[q], s_stage + a.lds_w[q], th, wave, lane, fa)
                                  : level_sse_single<DPAD, 0, NW>(L.Apk, L.ncb, s_stage + a.lds_y[q], nullptr, th, wave, lane, fa);
    sq = sum_rows(sq);
    if (lane < 16) s_red[wave * 16 + lane] = sq;
  };
  auto first_block = [&](int q, double2 (&fa)[K2]) {  // level q's first fragment block of this wave, on its way early
    frag_load_buf<DPAD>(frag_src(a.lv[q].Apk, lane), wave < a.lv[q].ncb ? wave : a.lv[q].ncb - 1, fa);
  };

#ifdef TDA_PC_CONSUMER_ONLY
  if (wave >= 4) return;
#endif
#ifdef TDA_PC_PRODUCER_ONLY
  if (wave < 4) return;
#endif
  if (wave >= 4) {
    // =====================================================================================================================
    // PRODUCER
    // =====================================================================================================================
    const int pw = wave - 4;
    bool has_pb[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) has_pb[i] = pw + 4 * i < ncb0;
    const FragSrc src0 = frag_src(a.lv[0].Apk, lane);
    double2 fA[NB][K2];
    auto load_coarse_operator = [&]() {
#pragma unroll
      for (int i = 0; i < NB; ++i) frag_load_buf<DPAD>(src0, has_pb[i] ? pw + 4 * i : (ncb0 - 1), fA[i]);
    };
    auto store_outputs = [&](int set, int i, const double4_t& g) {  // this lane's accumulator rows (chain lc) as one 32-byte piece
      double2* __restrict__ dst = reinterpret_cast<double2*>(s_G + (set * 16 + lc) * LDG + (pw + 4 * i) * 16 + hi * 4);
      dst[0] = double2{g[0], g[1]};
      dst[1] = double2{g[2], g[3]};
    };
    auto produce = [&](int s) {  // G[s] from the staged increments of step s
      const double2* __restrict__ row = reinterpret_cast<const double2*>(s_inc + (s & 1) * 64 * RSX + lane * RSX);
      double2 b[K2];
#pragma unroll
      for (int kk = 0; kk < K2; ++kk) b[kk] = row[kk];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const double4_t g = chainmm(fA[i], b);
        if (has_pb[i]) store_outputs(s & 1, i, g);
      }
    };
    auto anchor = [&]() {  // A theta of the states staged in s_prop -> row set 1 (no step has been produced yet)
      double2 b[K2];
#pragma unroll
      for (int kk = 0; kk < K2; ++kk) b[kk] = double2{s_prop[lc * LDP + 8 * kk + hi], s_prop[lc * LDP + 8 * kk + 4 + hi]};
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const double4_t g = chainmm(fA[i], b);
        if (has_pb[i]) store_outputs(1, i, g);
      }
    };
    load_coarse_operator();
    // ---- prologue: one anchor per level state, then step 0 ----
    for (int q = NLEV - 1; q >= 0; --q) {
      if (q > 0 && !a.cascade) continue;
      __syncthreads();  // states of level q staged
      anchor();
      __syncthreads();  // ... and taken
    }
    __syncthreads();  // increments of steps 0, 1 staged
    if (a.S > 0) produce(0);
    __syncthreads();
    int cnt0 = a.cnt[0], cntU[NLEV - 1];
#pragma unroll
    for (int q = 1; q < NLEV; ++q) cntU[q - 1] = a.cnt[q];
    for (int s = 0; s < a.S; ++s) {
      PC_STAMP(0);
      if (s + 1 < a.S) produce(s + 1);
      PC_STAMP(1);
      const bool act = a.cascade && cnt0 + 1 == L0;
      if (act) first_block(1, fA[0]);  // (the coarse operator is fetched again behind the action)
      PC_STAMP(4);
      __syncthreads();
      PC_STAMP(5);
      cnt0 += 1;
      if (!act) continue;
      bool more = true;
#pragma unroll
      for (int k = 0; k < NLEV - 1; ++k) {
        if (!more) break;
        const int q = k + 1;
        const bool more_after = q < NLEV - 1 && cntU[q - 1] + 1 == a.sl[q];
        __syncthreads();  // states staged
        level_sums(q, fA[0]);
        if (more_after) first_block(q + 1, fA[0]);
        else load_coarse_operator();
        __syncthreads();  // sums in s_red
        if (k == 0) cnt0 = 0;
        else cntU[k - 1] = 0;
        cntU[q - 1] += 1;
        more = more_after;
      }
    }
    return;
  }

  // =======================================================================================================================
  // CONSUMER: wave w owns chains 4 w .. 4 w + 3, 16 lanes per chain
  // =======================================================================================================================
  const int cc = lane >> 4, j = lane & 15;
  const int ch = 4 * wave + cc;               // chain inside the tile
  const int64_t gch = tile * 16 + ch;
  const bool actv = j * EPT < DPAD;           // (fewer than 16 parameters: the upper lanes hold none)
  const bool live = gch < a.N;
  const uint32_t gchain = (uint32_t)(a.chain_offset + gch);
  const bool prior_std = a.pr.kind == PRIOR_STANDARD;
  const bool has_logu = a.logu0 != nullptr;
  double cur0[EPT], curU[NLEV - 1][EPT], prp[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    cur0[e] = actv ? a.theta[gch * DPAD + j * EPT + e] : 0.0;
#pragma unroll
    for (int q = 1; q < NLEV; ++q) curU[q - 1][e] = actv ? a.theta[((size_t)q * a.NP + gch) * DPAD + j * EPT + e] : 0.0;
  }
  double lp0 = a.lp[gch], ll0 = a.ll[gch], lpU[NLEV - 1], llU[NLEV - 1];
  int any0 = a.anyacc[gch], anyU[NLEV - 1], cntU[NLEV - 1], nrecU[NLEV - 1];
  int64_t stepU[NLEV - 1];
#pragma unroll
  for (int q = 1; q < NLEV; ++q) {
    lpU[q - 1] = a.lp[(size_t)q * a.NP + gch];
    llU[q - 1] = a.ll[(size_t)q * a.NP + gch];
    anyU[q - 1] = a.anyacc[(size_t)q * a.NP + gch];  // (the finest level's is never read: carried unchanged)
    cntU[q - 1] = a.cnt[q];
    stepU[q - 1] = a.done[q];
    nrecU[q - 1] = 0;
  }
  double Slp[NPAIR], Sll[NPAIR];  // level j at the start of level q's current step, pair_index(j, q)
#pragma unroll
  for (int p = 0; p < NPAIR; ++p) {
    Slp[p] = a.Sst[((size_t)p * 2 + 0) * a.NP + gch];
    Sll[p] = a.Sst[((size_t)p * 2 + 1) * a.NP + gch];
  }
  const double scal = a.scaling[gch];
  const double keep = is_pcn ? sqrt(1.0 - scal * scal) : 1.0;
  int cnt0 = a.cnt[0];
  int64_t step0 = a.done[0];
  int nrec0 = 0;
  int ringidx = (int)(a.ring_pos % a.ring_P);  // (the 64-bit remainder costs ~150 scalar instructions: once, not per step)
  const double llscale = dg0 ? -0.5 : -0.5 / a.lv[0].var;
  const double logconst = a.pr.logconst;

  // where this lane's elements of an increment go in a fragment-ordered tile: row = fragment lane (dim & 3) * 16 + chain
  int st_dst[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int dm = j * EPT + e;
    st_dst[e] = ((dm & 3) * 16 + ch) * RSX + (dm >> 2);
  }
  // the diagonal error model: this chain's corrected data y - b and inverse variances (registers; one operator block per wave)
  double yc[aemd0 ? NS : 1], wc[aemd0 ? NS : 1];
  if constexpr (aemd0) {
    const int m0 = a.aem_ld;  // real output count = row stride of the [N][m] bias / inverse-variance arrays
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      const int blk = 4 * (t >> 2) + (j >> 2);
      const int o = blk * 16 + (j & 3) + 4 * (t & 3);
      const bool in = blk < ncb0 && live && o < m0;
      yc[t] = s_ys[64 * (t >> 2) + 4 * j + (t & 3)] - (in ? a.aem_bias[(size_t)gch * m0 + o] : 0.0);
      wc[t] = in ? a.aem_P[(size_t)gch * m0 + o] : 0.0;
    }
  } else {
    yc[0] = wc[0] = 0.0;
  }
  auto stage_state = [&](const double (&st)[EPT]) {
    if (actv) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) s_prop[ch * LDP + j * EPT + e] = st[e];
    }
  };
  auto read_outputs = [&](int set, double (&F)[NS]) {
#pragma unroll
    for (int i2 = 0; i2 < NS / 4; ++i2) {
      const double2* __restrict__ row = reinterpret_cast<const double2*>(s_G + (set * 16 + ch) * LDG + 64 * i2 + 4 * j);
      const double2 v0 = row[0], v1 = row[1];
      F[4 * i2 + 0] = v0.x;
      F[4 * i2 + 1] = v0.y;
      F[4 * i2 + 2] = v1.x;
      F[4 * i2 + 3] = v1.y;
    }
  };
  // increments of step s as loaded, and in lanes 0 / 1 of the chain its log-uniform / uniform
  double nx[EPT], lun = 0.0;
  const double* __restrict__ const inc_lane = a.inc + gch * DPAD + j * EPT;      // + s NP DPAD
  const double* __restrict__ const u_lane = ((j == 0 && has_logu) ? a.logu0 : a.u0) + gch;  // + s NP (lanes 0 / 1 of the chain)
  auto step_load = [&](int s) {
    const double* __restrict__ src = inc_lane + (size_t)s * a.NP * DPAD;
#pragma unroll
    for (int e = 0; e < EPT; ++e) nx[e] = actv ? src[e] : 0.0;
    if (j < 2) lun = u_lane[(size_t)s * a.NP];
  };
  auto step_stage = [&](int s) {
    if (actv) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) s_inc[(s & 1) * 64 * RSX + st_dst[e]] = scal * nx[e];
    }
    if (j < 2) s_lu[((s & 1) * 2 + j) * 16 + ch] = lun;
  };

  // ---- prologue: model outputs of every level's state (anchors), increments of steps 0 and 1 ----
  // (the outputs at the states of the levels above wait in LDS, 16-byte pieces per lane: they move only when a level acts)
  double Fc[NS];
  double2* __restrict__ const fs_lane = reinterpret_cast<double2*>(s_Fs) + tid;  // piece (q - 1) NS / 2 + t2 at stride 256
  auto fs_put = [&](int q, const double (&F)[NS]) {
#pragma unroll
    for (int t2 = 0; t2 < NS / 2; ++t2) fs_lane[((q - 1) * (NS / 2) + t2) * 256] = double2{F[2 * t2], F[2 * t2 + 1]};
  };
  auto fs_get = [&](int q, double (&F)[NS]) {
#pragma unroll
    for (int t2 = 0; t2 < NS / 2; ++t2) {
      const double2 v = fs_lane[((q - 1) * (NS / 2) + t2) * 256];
      F[2 * t2] = v.x;
      F[2 * t2 + 1] = v.y;
    }
  };
  for (int q = NLEV - 1; q >= 0; --q) {  // (level 0 last: its outputs stay in registers)
    if (q > 0 && !a.cascade) continue;
    if (q == 0) stage_state(cur0);
    else stage_state(curU[q - 1]);
    __syncthreads();
    __syncthreads();
    read_outputs(1, Fc);
    if (q > 0) fs_put(q, Fc);
  }
  if (a.S > 0) {
    step_load(0);
    step_stage(0);
  }
  if (a.S > 1) {
    step_load(1);
    step_stage(1);
  }
  if (a.S > 2) step_load(2);
  __syncthreads();
  __syncthreads();  // G[0] there

  for (int s = 0; s < a.S; ++s) {
    // ================= one Metropolis-Hastings step of the coarse level (chain.py:404-444) =================
    const int set = s & 1;
    PC_STAMP(0);
    double sx[EPT];
    const double lu = s_lu[(set * 2 + 0) * 16 + ch], u_ex = s_lu[(set * 2 + 1) * 16 + ch];
#pragma unroll
    for (int e = 0; e < EPT; ++e) sx[e] = actv ? s_inc[set * 64 * RSX + st_dst[e]] : 0.0;  // (this lane's own staged values)
    PC_STAMP(1);
    double pp = 0.0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) prp[e] = is_pcn ? keep * cur0[e] + sx[e] : cur0[e] + sx[e];  // (lanes without parameters: zeros)
    if (prior_std) {  // (one wave-uniform branch around the four elements, not one per element)
#pragma unroll
      for (int e = 0; e < EPT; ++e) pp += prp[e] * prp[e];
    } else if (actv) {
      bool outside = false;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const double dv = prp[e] - s_pm[j * EPT + e];
        pp += dv * dv * s_pinv[j * EPT + e];
        outside = outside || prp[e] < s_lo[j * EPT + e] || prp[e] > s_hi[j * EPT + e];
      }
      if (outside) pp = INFINITY;
    }
    // F' = keep F + G, residuals (all reads first: four to eight independent LDS round trips, not a chain of them)
    double g[NS], yv[aemd0 ? 1 : NS], wv[(dg0 && !aemd0) ? NS : 1], Fp[NS], sse = 0.0;
    read_outputs(set, g);
    if constexpr (!aemd0) {
#pragma unroll
      for (int i2 = 0; i2 < NS / 4; ++i2) {
        const double2* __restrict__ yr = reinterpret_cast<const double2*>(s_ys + 64 * i2 + 4 * j);
        const double2 y0 = yr[0], y1 = yr[1];
        yv[4 * i2 + 0] = y0.x, yv[4 * i2 + 1] = y0.y, yv[4 * i2 + 2] = y1.x, yv[4 * i2 + 3] = y1.y;
        if constexpr (dg0) {
          const double2* __restrict__ wr = reinterpret_cast<const double2*>(s_ws + 64 * i2 + 4 * j);
          const double2 w0 = wr[0], w1 = wr[1];
          wv[4 * i2 + 0] = w0.x, wv[4 * i2 + 1] = w0.y, wv[4 * i2 + 2] = w1.x, wv[4 * i2 + 3] = w1.y;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < NS; ++t) {
      Fp[t] = is_pcn ? keep * Fc[t] + g[t] : Fc[t] + g[t];
      const double res = Fp[t] - (aemd0 ? yc[aemd0 ? t : 0] : yv[aemd0 ? 0 : t]);
      const double sq = res * res;
      sse += dg0 ? sq * (aemd0 ? wc[aemd0 ? t : 0] : wv[(dg0 && !aemd0) ? t : 0]) : sq;
    }
    PC_STAMP(2);
    sse = sum_row16(sse);
    pp = sum_row16(pp);
    const double ll_n = llscale * sse;
    const double lp_n = -0.5 * (logconst + pp);
    const double post_n = lp_n + ll_n;
    const double delta = is_pcn ? ll_n - ll0 : post_n - (lp0 + ll0);
    bool acc0;
    PC_STAMP(3);
    if (has_logu && (fabs(lu - delta) > 1e-9 || delta != delta)) {
      acc0 = (post_n == post_n) && (lu < delta);
    } else {
      acc0 = accept_exact(has_logu ? u_ex : lu, delta, post_n);  // (without log-uniforms slot 0 holds the uniform itself)
    }
    if (acc0) {
      lp0 = lp_n;
      ll0 = ll_n;
#pragma unroll
      for (int t = 0; t < NS; ++t) Fc[t] = Fp[t];
#pragma unroll
      for (int e = 0; e < EPT; ++e) cur0[e] = prp[e];
      any0 = 1;
    }
    // step s + 2 takes the slots of step s (this wave alone reads and writes them), step s + 3 starts its way.  HERE: the wait
    // for the loads is a wait for everything this wave has in flight (one counter for loads and stores on gfx9, and the
    // stores sit in branches the compiler cannot count) -- behind the decision the record stores of step s - 1 are long done
    // and those of step s not yet issued.
    if (s + 2 < a.S) {
#pragma unroll
      for (int e = 0; e < EPT; ++e) asm volatile("" : "+v"(nx[e]));
      asm volatile("" : "+v"(lun));
      step_stage(s + 2);
      if (s + 3 < a.S) step_load(s + 3);
    }
    // record of the step (state after the decision, its densities, the flag, the accept-flag window)
    if (live) {
      if (a.rec_params[0] && actv) {
        double* __restrict__ rp = a.rec_params[0] + ((size_t)nrec0 * a.N + gch) * a.d + j * EPT;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
          if (j * EPT + e < a.d) rp[e] = cur0[e];
      }
      if (j == 0) {
        const size_t r = (size_t)nrec0 * a.N + gch;
        if (a.rec_stats[0]) {
          a.rec_stats[0][r * 3 + 0] = lp0;
          a.rec_stats[0][r * 3 + 1] = ll0;
          a.rec_stats[0][r * 3 + 2] = lp0 + ll0;
        }
        if (a.rec_acc[0]) a.rec_acc[0][r] = acc0 ? 1 : 0;
      }
    }
    if (j == 0) {
      a.ring[(size_t)ringidx * a.NP + gch] = acc0 ? 1 : 0;
      if (a.sid && acc0) a.sid[gch] = step0 + 1;
    }
    ringidx = ringidx + 1 == a.ring_P ? 0 : ringidx + 1;
    nrec0 += 1;
    step0 += 1;
    const bool act = a.cascade && cnt0 + 1 == L0;
    if (act) stage_state(cur0);
    PC_STAMP(4);
    __syncthreads();
    PC_STAMP(5);
    cnt0 += 1;
    if (!act) continue;

    // ================= upper levels whose subchain just completed (chain.py:354-402; MLDA: proposal.py:1441-1530) =========
    // Level q = k + 1 is evaluated directly at y = the state of the levels below it (after an action of level q - 1 these all
    // coincide with level 0's, so y is always cur0).  One prior for all levels: log-prior(y) = lp0.
    auto LP = [&](int jl) -> double& { return jl == 0 ? lp0 : lpU[jl - 1]; };
    auto LL = [&](int jl) -> double& { return jl == 0 ? ll0 : llU[jl - 1]; };
    bool more = true;
#pragma unroll
    for (int k = 0; k < NLEV - 1; ++k) {
      if (!more) break;
      const int q = k + 1;
      const LevelDev& L = a.lv[q];
      double uq;  // (independent of the residuals: taken before them)
      if (a.u_rep[q]) uq = a.u_rep[q][(size_t)(stepU[q - 1] - a.done[q]) * a.N + (live ? gch : 0)];
      else uq = accept_uniform(a.seed, gchain, (uint32_t)stepU[q - 1], (uint32_t)q);
      const bool more_after = q < NLEV - 1 && cntU[q - 1] + 1 == a.sl[q];  // the level above acts right after this one
      __syncthreads();  // states staged (before the loop, or behind the previous action)
      da_level_sums<DPAD>(L.Apk, L.ncb, L.noise_kind == 1, s_stage + a.lds_y[q], s_stage + a.lds_w[q], s_prop, s_red, wave, lane);
      __syncthreads();
      double t1 = s_red[ch];
#pragma unroll
      for (int w = 1; w < NW; ++w) t1 += s_red[w * 16 + ch];
      const double llq = L.noise_kind == 1 ? -0.5 * t1 : -0.5 * t1 / L.var;
      const double y_lp = LP(k), y_ll = LL(k);
      const double lpq = y_lp;
      const int pkq = pair_index(k, q);
      const double alq = exp(((lpq + llq) - (lpU[q - 1] + llU[q - 1])) + (Slp[pkq] + Sll[pkq]) - (y_lp + y_ll));  // chain.py:475-483
      const int any_below = k == 0 ? any0 : anyU[k - 1];
      const bool accq = (any_below != 0) && (uq < alq);  // skip rule: nothing accepted below -> a recorded rejection (:357-364)
      if (accq) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) curU[q - 1][e] = cur0[e];
        fs_put(q, Fc);
        lpU[q - 1] = lpq;
        llU[q - 1] = llq;
      } else {  // every level below q restarts from theta_q, with its model outputs and densities there (:360-362, 394-396)
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          cur0[e] = curU[q - 1][e];
#pragma unroll
          for (int jl = 1; jl < q; ++jl) curU[jl - 1][e] = curU[q - 1][e];
        }
        fs_get(q, Fc);
#pragma unroll
        for (int jl = 1; jl < q; ++jl) fs_put(jl, Fc);
#pragma unroll
        for (int jl = 0; jl < q; ++jl) {
          LP(jl) = Slp[pair_index(jl, q)];
          LL(jl) = Sll[pair_index(jl, q)];
        }
      }
#pragma unroll
      for (int jl = 0; jl < q; ++jl)
#pragma unroll
        for (int q2 = jl + 1; q2 <= q; ++q2) {
          Slp[pair_index(jl, q2)] = LP(jl);
          Sll[pair_index(jl, q2)] = LL(jl);
        }
      if (k == 0) any0 = 0;
      else anyU[k - 1] = 0;
      if (q < NLEV - 1) anyU[q - 1] |= accq ? 1 : 0;
      if (live) {
        if (a.rec_params[q] && actv) {
          double* __restrict__ rp = a.rec_params[q] + ((size_t)nrecU[q - 1] * a.N + gch) * a.d + j * EPT;
#pragma unroll
          for (int e = 0; e < EPT; ++e)
            if (j * EPT + e < a.d) rp[e] = curU[q - 1][e];
        }
        if (j == 0) {
          const size_t r = (size_t)nrecU[q - 1] * a.N + gch;
          if (a.rec_stats[q]) {
            a.rec_stats[q][r * 3 + 0] = lpU[q - 1];
            a.rec_stats[q][r * 3 + 1] = llU[q - 1];
            a.rec_stats[q][r * 3 + 2] = lpU[q - 1] + llU[q - 1];
          }
          if (a.rec_acc[q]) a.rec_acc[q][r] = accq ? 1 : 0;
        }
      }
      if (j == 0) a.ring[(size_t)ringidx * a.NP + gch] = accq ? 1 : 0;  // alignment entry of the coarse accept list (:363,389,397)
      ringidx = ringidx + 1 == a.ring_P ? 0 : ringidx + 1;
      nrecU[q - 1] += 1;
      stepU[q - 1] += 1;
      if (k == 0) cnt0 = 0;
      else cntU[k - 1] = 0;
      cntU[q - 1] += 1;
      more = more_after;
      if (more) stage_state(cur0);  // the level above is evaluated at the state this action left
    }
  }

#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    if (actv) {
      a.theta[gch * DPAD + j * EPT + e] = cur0[e];
#pragma unroll
      for (int q = 1; q < NLEV; ++q) a.theta[((size_t)q * a.NP + gch) * DPAD + j * EPT + e] = curU[q - 1][e];
    }
  }
  if (j == 0) {
    a.lp[gch] = lp0;
    a.ll[gch] = ll0;
    a.anyacc[gch] = any0;
#pragma unroll
    for (int q = 1; q < NLEV; ++q) {
      a.lp[(size_t)q * a.NP + gch] = lpU[q - 1];
      a.ll[(size_t)q * a.NP + gch] = llU[q - 1];
      a.anyacc[(size_t)q * a.NP + gch] = anyU[q - 1];
    }
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      a.Sst[((size_t)p * 2 + 0) * a.NP + gch] = Slp[p];
      a.Sst[((size_t)p * 2 + 1) * a.NP + gch] = Sll[p];
    }
  }
}

}  // namespace tda
