// libtinyda_hip.so : C-ABI (include/tinyda_amd.h) over the gfx950 kernels in tda_kernels.h.
// Host side = problem lowering (fragment packing, Cholesky of shared covariances), block scheduling and
// record plumbing.  No CPU compute fallback exists: every entry point that advances chains launches HIP
// kernels or fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tda_kernels.h"
#include "tinyda_amd.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return fail(TDA_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  hipError_t e = hipPointerGetAttributes(&at, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged;
}

bool is_pinned_host_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t at;
  hipError_t er = hipPointerGetAttributes(&at, p);
  if (er != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

int dpad_for(int d) { return d <= 8 ? 8 : d <= 16 ? 16 : d <= 32 ? 32 : d <= 64 ? 64 : 128; }

// lower Cholesky of a d x d row-major SPD matrix; returns false if not positive definite
bool cholesky_host(const double* C, int d, std::vector<double>& L) {
  L.assign((size_t)d * d, 0.0);
  for (int k = 0; k < d; ++k) {
    for (int i = k; i < d; ++i) {
      double s = C[(size_t)i * d + k];
      for (int p = 0; p < k; ++p) s = std::fma(-L[(size_t)i * d + p], L[(size_t)k * d + p], s);
      if (i == k) {
        if (!(s > 0.0)) return false;
        L[(size_t)k * d + k] = std::sqrt(s);
      } else {
        L[(size_t)i * d + k] = s / L[(size_t)k * d + k];
      }
    }
  }
  return true;
}

// inverse of a lower-triangular matrix
void tri_inverse_host(const std::vector<double>& L, int d, std::vector<double>& W) {
  W.assign((size_t)d * d, 0.0);
  for (int j = 0; j < d; ++j) {
    W[(size_t)j * d + j] = 1.0 / L[(size_t)j * d + j];
    for (int i = j + 1; i < d; ++i) {
      double s = 0.0;
      for (int p = j; p < i; ++p) s += L[(size_t)i * d + p] * W[(size_t)p * d + j];
      W[(size_t)i * d + j] = -s / L[(size_t)i * d + i];
    }
  }
}

// The error model's FACTOR FORM (tda_kernels_aemr.h) of an m x m lower Cholesky factor L, padded with identity to MP rows: the
// off-diagonal 16 x 16 tiles as U = L^T tiles, the diagonal tiles replaced by their inverses.  out: aemr_v_doubles(MP) doubles.
void factor_form_pack_host(const std::vector<double>& L, int m, int MP, std::vector<double>& out) {
  out.assign(tda::aemr_v_doubles(MP), 0.0);
  auto Lp = [&](int i, int j) { return (i < m && j < m) ? L[(size_t)i * m + j] : (i == j ? 1.0 : 0.0); };
  for (int q = 0; q < MP / 16; ++q) {
    std::vector<double> D(256), Di;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) D[i * 16 + j] = j <= i ? Lp(16 * q + i, 16 * q + j) : 0.0;
    tri_inverse_host(D, 16, Di);
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j <= i; ++j) out[tda::aemr_v_offset(16 * q + i, 16 * q + j)] = Di[i * 16 + j];
  }
  for (int i = 0; i < MP; ++i)
    for (int j = 0; j < (i & ~15); ++j) out[tda::aemr_w_offset_offdiag(i, j)] = Lp(i, j);
}
// ... and back: L (m x m, row-major) from one chain's factor form
void factor_form_unpack_host(const double* w, int m, std::vector<double>& L) {
  L.assign((size_t)m * m, 0.0);
  for (int q = 0; q * 16 < m; ++q) {
    std::vector<double> Di(256, 0.0), D;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j <= i; ++j) Di[i * 16 + j] = w[tda::aemr_v_offset(16 * q + i, 16 * q + j)];
    tri_inverse_host(Di, 16, D);
    for (int i = 0; i < 16 && 16 * q + i < m; ++i)
      for (int j = 0; j <= i; ++j) L[(size_t)(16 * q + i) * m + 16 * q + j] = D[i * 16 + j];
  }
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < (i & ~15); ++j) L[(size_t)i * m + j] = w[tda::aemr_w_offset_offdiag(i, j)];
}

// pack an (rows x cols) row-major matrix into MFMA A-operand fragments, see LevelDev::Apk; kpad = padded K
void pack_fragments(const double* A, int rows, int cols, int dpad, std::vector<double>& out, int& ncb) {
  ncb = (rows + 15) / 16;
  const int K2 = dpad / 8;
  out.assign((size_t)ncb * K2 * 64 * 2, 0.0);
  for (int cb = 0; cb < ncb; ++cb)
    for (int k2 = 0; k2 < K2; ++k2)
      for (int l = 0; l < 64; ++l)
        for (int e = 0; e < 2; ++e) {
          const int r = cb * 16 + (l & 15), cidx = 4 * (2 * k2 + e) + (l >> 4);
          if (r < rows && cidx < cols)
            out[(((size_t)cb * K2 + k2) * 64 + l) * 2 + e] = A[(size_t)r * cols + cidx];
        }
}

// Device-memory pool: an engine holds ~1.2 GB of block buffers at BASELINE config 2, and tda.sample() creates and destroys
// one engine per call.  hipMalloc / hipFree of that set cost 10-20 ms per call (hipFree synchronises the device), which is
// as long as 1000 iterations of 4096 chains take.  Released buffers are therefore kept per (device, byte count) and handed
// out again -- zeroed, as fresh allocations are -- up to TINYDA_POOL_GB GiB PER DEVICE (default 8, at most an eighth of the device's memory); tda_release_cached_memory() returns
// them to the driver, as does an allocation failure before it is reported.
struct DevPool {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void*> idle;
  size_t held = 0;                       // all devices (what trim() reports)
  std::map<int, size_t> held_dev, cap_dev;  // per device: bytes parked, cap
  // The cap is per DEVICE (one process-wide figure read on whichever device was current at first use could come from
  // the wrong GPU, or from a moment when another engine's peak buffers were live): TINYDA_POOL_GB GiB if set, otherwise 8 GiB but
  // never more than an eighth of that device's TOTAL memory -- a figure that does not depend on what happens to be allocated.
  size_t capacity(int dev) {
    auto it = cap_dev.find(dev);
    if (it != cap_dev.end()) return it->second;
    const char* v = getenv("TINYDA_POOL_GB");
    size_t cap = (size_t)((v ? atof(v) : 8.0) * 1073741824.0);
    if (!v) {
      int cur = 0;
      (void)hipGetDevice(&cur);
      size_t free_b = 0, total_b = 0;
      if (hipSetDevice(dev) == hipSuccess && hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b / 8 < cap) cap = total_b / 8;
      (void)hipSetDevice(cur);
    }
    cap_dev[dev] = cap;
    return cap;
  }
  void* take(int dev, size_t bytes) {
    std::lock_guard<std::mutex> g(mu);
    auto it = idle.find({dev, bytes});
    if (it == idle.end()) return nullptr;
    void* p = it->second;
    idle.erase(it);
    held -= bytes;
    held_dev[dev] -= bytes;
    return p;
  }
  bool give(int dev, size_t bytes, void* p) {
    std::lock_guard<std::mutex> g(mu);
    if (bytes < (64u << 10) || held_dev[dev] + bytes > capacity(dev)) return false;  // small buffers: the runtime's own sub-allocator is fast
    idle.insert({{dev, bytes}, p});
    held += bytes;
    held_dev[dev] += bytes;
    return true;
  }
  size_t trim() {
    std::lock_guard<std::mutex> g(mu);
    const size_t was = held;
    int cur = 0;
    (void)hipGetDevice(&cur);
    for (auto& kv : idle) {
      (void)hipSetDevice(kv.first.first);
      (void)hipFree(kv.second);
    }
    (void)hipSetDevice(cur);
    idle.clear();
    held = 0;
    held_dev.clear();
    return was;
  }
};
DevPool g_pool;
thread_local bool g_pool_accepting = false;

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  int dev = 0;
  int alloc(size_t count) {
    release();
    n = count;
    if (count == 0) return TDA_OK;
    const size_t bytes = count * sizeof(T);
    (void)hipGetDevice(&dev);
    if (void* q = g_pool.take(dev, bytes)) {
      p = (T*)q;
      hipError_t e = hipMemset(p, 0, bytes);  // what a fresh allocation holds
      if (e != hipSuccess) return fail(TDA_ERR_HIP, "hipMemset(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
      return TDA_OK;
    }
    hipError_t e = hipMalloc((void**)&p, bytes);
    if (e != hipSuccess && g_pool.trim() > 0) {
      (void)hipGetLastError();
      e = hipMalloc((void**)&p, bytes);
    }
    if (e != hipSuccess) {
      p = nullptr;
      return fail(TDA_ERR_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    return TDA_OK;
  }
  int upload(const std::vector<T>& h) {
    int rc = alloc(h.size());
    if (rc) return rc;
    if (!h.empty()) HIP_TRY(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return TDA_OK;
  }
  void release() {
    // only buffers of an engine being destroyed go back to the pool: its streams have been synchronised by then, whereas a
    // buffer replaced in the middle of a run may still be read by queued kernels (hipFree waits for them, the pool would not)
    if (p && !(g_pool_accepting && g_pool.give(dev, n * sizeof(T), p))) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  ~DevBuf() { release(); }
};

}  // namespace
#include "tda_diag.inc"
#include "tda_usermodel.inc"
namespace {

struct Level {
  bool set = false;
  int m = 0, m_pad = 0, ncb = 0, noise_kind = 0;
  double var = 1.0;
  int model = 0;  // tda::MODEL_LINEAR / MODEL_ROSENBROCK / MODEL_USER (hiprtc-compiled source, tda_usermodel.inc) /
                  // MODEL_CALLBACK (batched host callback, tda_kernels_ext.h)
  tda_forward_batch_fn cb_fn = nullptr;
  void* cb_user = nullptr;
  double* cb_theta_h = nullptr;  // page-locked [N][d] / [N][m]
  double* cb_F_h = nullptr;
  DevBuf<double> cb_prop, cb_F;
  void release_callback_buffers() {
    if (cb_theta_h) (void)hipHostFree(cb_theta_h);
    if (cb_F_h) (void)hipHostFree(cb_F_h);
    cb_theta_h = cb_F_h = nullptr;
  }
  hipModule_t umod = nullptr;
  hipFunction_t ufn = nullptr, ufn_eval = nullptr, ufn_level = nullptr;
  DevBuf<double> udata, uw;
  double ros_a = 1.0, ros_b = 10.0, ros_data = 0.0;
  DevBuf<double> Apk, ytil, w, Ppk;
  // adaptive error model: plain row-major copies for the wave-per-chain kernel
  std::vector<double> A_h, ytil_h, data_h, cov_h;
  std::vector<double> w_h, Pinv_h;  // diagonal weights 1 / sigma_i^2, dense Sigma_e^-1 [m][m] (MALA's gradient operator)
  DevBuf<double> A_rm, ytil64, data64, cov64;  // error-model copies, row stride em_ld (cov64: Sigma_e as upper tiles, tda_kernels_aemr.h)
  DevBuf<double> Pd;                           // callback / source-defined level with dense noise: Sigma^-1 [m][m]
  DevBuf<double> A_dev, b_dev;                 // hierarchies: row-major [m][d] and offset [m] for k_ext_linear_eval (host-sequenced mode)
  int em_ld = 0;                               // 64 (m <= 64), 128 (m <= 128) or 256 (m <= 256); 0: level too large for an error model
};

struct TimedLaunch {
  hipEvent_t a, b;
  int kind;
};

}  // namespace

struct tda_engine {
  tda_config cfg{};
  int d = 0, DP = 0;
  int64_t N = 0, NP = 0;
  int SMAX = 128;
  hipStream_t stream = nullptr;
  bool own_stream = false;

  // prior
  bool prior_set = false;
  int prior_kind = tda::PRIOR_DIAG;
  bool prior_is_standard = false;  // N(0, I): the single-level tile kernel skips the constant loads
  bool prior_bounded = false;      // JointPrior with uniform components: support bounds in prior_lo / prior_hi
  double prior_logconst = 0.0;
  std::vector<double> prior_mean_h, prior_cov_h, prior_L_h;
  DevBuf<double> prior_lo, prior_hi;
  DevBuf<double> prior_mean, prior_pinv, prior_Wpk, prior_wmu;
  int prior_ncb = 0;

  std::vector<Level> levels;

  // proposal
  bool prop_set = false;
  tda_proposal_params pp{};
  std::vector<double> prop_C_h, q_mean_h;  // q_mean_h: independence sampler
  std::vector<double> ow_state_h, ow_noise_h;  // OperatorWeightedCrankNicolson operators [d][d]
  DevBuf<double> ow_SopT;                      // state operator, transposed and padded: [DP][DP], SopT[j][i] = S[i][j]
  // OperatorWeightedCrankNicolson with per-chain operators (tda_engine_set_proposal_spectrum): B = V diag(lambda) V^T
  bool ow_spectral = false;
  double ow_scaling0 = 1.0;
  std::vector<double> ow_V_h, ow_lam_h;        // V [d][d] row-major (columns = eigenvectors), lambda [d]
  DevBuf<double> ow_VVt, ow_lam;               // [2][DP][DP]: V, then V^T; [DP]
  DevBuf<double> mala_H, mala_c, mala_grad;    // MALA: H [DP][DP] (symmetric), c [DP], grad log post of the current states [NP][DP]
  DevBuf<double> q_mean_d, lq, qzblk, qzblk2[2];
  double am_sd = 1.0;
  bool L_shared = true;
  bool L_identity = false;  // the shared proposal factor is the identity: increments are the normals themselves (k_rng_direct)
  DevBuf<double> Lk, am_mu, am_sigma, scaling;
  DevBuf<int32_t> acc_count, flags;

  // chain state
  bool inited = false;
  DevBuf<double> theta, lp, ll;
  DevBuf<double> theta_s, lp_s, ll_s;  // scratch state for tda_engine_evaluate
  int64_t t = 0;                        // proposal.t : adapt() calls so far
  int64_t k_adapt = 0;                  // diminishing-adaptation counter

  // block buffers
  DevBuf<double> inc, ublk, lublk, rec_params, rec_stats;
  // records into PINNED host memory: device block buffers double-buffered, copies on their own stream under the next block
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_rec[2] = {nullptr, nullptr}, ev_cp[2] = {nullptr, nullptr};
  DevBuf<double> rec_params2, rec_stats2;
  DevBuf<uint8_t> rec_acc2;
  // split proposal path (Philox mode): normals of block b+1 are drawn on a second stream under block b's steps
  hipStream_t rng_stream = nullptr;
  hipEvent_t ev_rng[2] = {nullptr, nullptr}, ev_apply[2] = {nullptr, nullptr}, ev_steps[2] = {nullptr, nullptr};
  DevBuf<double> zfrag[2], ublk2[2], lublk2[2];
  // hierarchies under AdaptiveMetropolis whose blocks are shorter than the adaptation period (the error model cuts them at every
  // base subchain): the states of the current period not yet folded into the moments (run_multilevel)
  DevBuf<double> am_stage;
  int64_t am_pending = 0, am_pending_t0 = 0;
  const double* am_pending_rows = nullptr;
  DevBuf<uint8_t> rec_acc;

  // multi-level state (n_levels > 1)
  static_assert(tda::MAXLEV == 6, "the initialisers of sl / aem_bt below list MAXLEV ones");
  int nlev = 1;
  int sl[tda::MAXLEV] = {1, 1, 1, 1, 1, 1};
  bool sub_set = false;
  int randomize = 0;
  int cnt[tda::MAXLEV] = {0, 0, 0, 0};
  int64_t done[tda::MAXLEV] = {0, 0, 0, 0};
  int64_t ring_pos = 0;
  int ring_P = 1;
  DevBuf<double> ml_theta, ml_lp, ml_ll, ml_S, ml_ysnap;
  DevBuf<int32_t> ml_anyacc, ml_pick;
  DevBuf<uint8_t> ml_ring;
  DevBuf<double> ml_rec_params[tda::MAXLEV], ml_rec_stats[tda::MAXLEV];
  DevBuf<uint8_t> ml_rec_acc[tda::MAXLEV];
  DevBuf<double> ml_rec_params2[tda::MAXLEV], ml_rec_stats2[tda::MAXLEV];  // second set: pinned-host async copies
  DevBuf<uint8_t> ml_rec_acc2[tda::MAXLEV];
  // adaptive error model
  int aem = 0;
  int aem_m = 0, aem_ld = 64;
  // diagonal error model (TDA_AEM_STATE_INDEPENDENT_DIAGONAL, tda_kernels_aemd.h): everything [N][m]
  DevBuf<double> aemd_F[tda::MAXLEV], aemd_Fst, aemd_bias[tda::MAXLEV], aemd_w[tda::MAXLEV], aemd_sig2[tda::MAXLEV], aemd_wfin;
  DevBuf<double> aemd_mu[tda::MAXLEV], aemd_var[tda::MAXLEV], aemd_md[tda::MAXLEV], aemd_data[tda::MAXLEV];
  DevBuf<double> theta_last;  // DREAMZ below a hierarchy: level-0 state right after a block's last base step (jump distance of the pCR update)
  bool ext_hier = false;  // hierarchy with callback / source-defined levels: sequenced by the host (run_multilevel)
  DevBuf<double> ext_Fcur[tda::MAXLEV], ext_Fst;  // error model there: outputs of the current links [NP][MP], of level j at theta_q [npairs][NP][MP]
  DevBuf<double> aem_bias[tda::MAXLEV], aem_covinv[tda::MAXLEV], aem_bmu[tda::MAXLEV], aem_bsig[tda::MAXLEV], aem_mdiff[tda::MAXLEV];
  DevBuf<double> aem_rvec;  // [NP][aem_ld] bias-corrected residual the action kernels leave for k_aem_refresh's update_link
  DevBuf<double> aem_upd;   // [NP][3][aem_ld] vectors of the tracker covariance update they leave for it
  DevBuf<double> aem_F;     // [4][NP][aem_ld] model outputs of levels q, q - 1 at the states of levels q - 1, q (k_linear_outputs_multi -> k_aem_action)
  int64_t aem_bt[tda::MAXLEV] = {1, 1, 1, 1, 1, 1};
  DevBuf<int64_t> ml_sid;
  DevBuf<double> prior_W_rm;
  DevBuf<double> u_rep_lv[tda::MAXLEV], ridx_rep;
  int64_t u_rep_lv_n[tda::MAXLEV] = {0, 0, 0, 0}, ridx_rep_n = 0;
  int64_t u_rep_lv_pos[tda::MAXLEV] = {0, 0, 0, 0}, ridx_rep_pos = 0;

  // single-level chains with 65 .. 128 parameters (tda_kernels_wide.h): the proposal factor lives in e->Lk as two buffers of
  // factor-form tiles; the diagonal tiles, the current-buffer selector, the padding "Sigma_e" of the swap
  bool wide = false;
  DevBuf<double> wide_ud, wide_cov;
  DevBuf<int32_t> wide_sel;

  // replay / export
  DevBuf<double> z_rep, u_rep;
  int64_t rep_steps = 0, rep_pos = 0;
  double *z_exp = nullptr, *u_exp = nullptr;  // caller pointers
  bool exp_dev = false;
  DevBuf<double> z_exp_d, u_exp_d;
  int64_t exp_steps = 0, exp_pos = 0;

  // DREAM(Z)
  bool is_dreamz = false;
  tda_dreamz_params dz{};
  bool arch_set = false, arch_given = false, auto_append = true;  // arch_given: the caller supplied the initial archive
  std::vector<double> Z0_h;
  int64_t arch_rows = 0;    // rows currently in the archive(s)
  int64_t sums_rows = 0;    // shared archive: rows already in zsum / zsq (the column sums are caught up at adaptation boundaries only)
  // distributed shared archive (tda_engine_set_archive_peers): this rank's segment is e->arch, the others are peer-mapped
  int dist_ranks = 0, dist_me = 0;
  const double* dist_seg[tda::MAX_PEERS] = {};
  DevBuf<const double*> dist_seg_dev;        // the same table in device memory (k_dreamz_draw<., true>)
  void* dist_opened[tda::MAX_PEERS] = {};   // pointers obtained from hipIpcOpenMemHandle (closed in destroy)
  int64_t dist_steps = 0;       // global steps whose rows are visible to the proposals (published)
  int64_t dist_pending = 0;     // steps of the unpublished blocks (at most two: a block may run while its predecessor's collective
                                // is still in flight -- the rows of block b then become visible from block b + 2)
  int64_t dist_unpub[2] = {0, 0};  // their step counts, oldest first
  int dist_n_unpub = 0;
  int64_t dist_adapt_rows = 0;  // archive size the block that left an adaptation pending proposed from
  int64_t dist_sum_steps = 0;   // steps whose local rows are already in the column sums
  bool dist_m0_summed = false;
  bool dist_adapt_pending = false;
  double dist_adapt_gamma = 1.0;
  int64_t arch_cap = 0;
  int64_t pending_steps = 0;  // shared mode: steps whose states are in blk_hist but not yet appended
  DevBuf<double> arch, zsum, zsq, dz_pCR, dz_LCR, dz_Delta, dz_coef, dz_epsm, theta_prev, blk_states, blk_hist;
  DevBuf<int32_t> dz_ridx, dz_mcr_last;
  DevBuf<double> dz_coef2, dz_epsm2, dz_u2;  // second set of draw outputs (shared DREAM: block b + 1 is drawn under block b's steps)
  DevBuf<int32_t> dz_ridx2;
  hipEvent_t ev_dz_adapt = nullptr;
  DevBuf<double> dz_partial;
  DevBuf<int32_t> rp_r, rp_mcr, rp_forced;
  DevBuf<double> rp_sub, rp_e, rp_eps, rp_u;
  int64_t rp_steps = 0, rp_pos = 0;

  // profiling
  bool profiling = false;
  std::vector<TimedLaunch> timed;

  // records: only every `thin`-th iteration (those with (t + 1) % thin == 0, t = iterations before it) reaches the caller
  int thin = 1;
  // progress (tda_engine_set_progress): at the end of every block a one-workgroup kernel writes the iterations completed so far
  // and the block's mean accept flag into page-locked host memory; tda_engine_get_progress reads it -- polled, never a sync
  double* prog_h = nullptr;  // [0] iterations completed, [1] mean accept flag of the last block (-1: not recorded)
  int64_t prog_queued = 0;   // iterations queued by run() so far (host side)
};

namespace {

using namespace tda;

inline int am_tiles_rt(int dp) {
  const int t = dp >= 16 ? dp / 16 : 1;
  return t * (t + 1) / 2;
}

int g_steps_waves = 0;  // 0 = decide per launch; TINYDA_STEPS_WAVES=4|8 pins it (A/B measurements)

template <int DPAD>
void launch_steps(const StepArgs& a, int64_t tiles, size_t lds, hipStream_t st) {
  if (g_steps_waves == 0) {
    const char* ev = getenv("TINYDA_STEPS_WAVES");
    g_steps_waves = (ev && atoi(ev) == 4) ? 4 : 8;
  }
  // dense noise keeps a 128 KiB residual tile and long MFMA chains per wave: 4 waves (512 registers) there
  const bool eight = g_steps_waves == 8 && a.lv.noise_kind != TDA_NOISE_DENSE;
  const bool ind = a.prop_kind == TDA_PROP_INDEPENDENCE;
  const bool os = a.prop_kind == TDA_PROP_OWCN && a.mode == MODE_STEP && a.cvec != nullptr;  // per-chain operators from the spectrum of B
  const bool ow = a.prop_kind == TDA_PROP_OWCN && a.mode == MODE_STEP && !os;  // + current-state tile and the state operator in LDS
  const bool ma = a.prop_kind == TDA_PROP_MALA && a.mode == MODE_STEP;  // + the gradient operator and 32 transition-density slots
  if (ow || ma) lds += ((size_t)16 * (DPAD + 2) + (size_t)DPAD * DPAD) * sizeof(double);
  if (os) lds += ((size_t)16 * (DPAD + 2) + (size_t)2 * DPAD * DPAD) * sizeof(double);
  auto go = [&](auto kern, unsigned threads, size_t bytes) {
    if (bytes > 64 * 1024)  // beyond the default dynamic-LDS window (gfx950 has 160 KiB per CU)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(threads), bytes, st, a);
  };
  if (eight) {
    const size_t l8 = lds + 2 * 64 * sizeof(double);  // two more [4][16] reduction slabs
    if (ind) go(&k_mh_steps<DPAD, 8, true>, 512, l8);
    else if (ow) go(&k_mh_steps<DPAD, 8, false, 1>, 512, l8);
    else if (os) go(&k_mh_steps<DPAD, 8, false, 3>, 512, l8);
    else if (ma) go(&k_mh_steps<DPAD, 8, false, 2>, 512, l8);
    else go(&k_mh_steps<DPAD, 8, false>, 512, l8);
  } else {
    if (ind) go(&k_mh_steps<DPAD, 4, true>, 256, lds);
    else if (ow) go(&k_mh_steps<DPAD, 4, false, 1>, 256, lds);
    else if (os) go(&k_mh_steps<DPAD, 4, false, 3>, 256, lds);
    else if (ma) go(&k_mh_steps<DPAD, 4, false, 2>, 256, lds);
    else go(&k_mh_steps<DPAD, 4, false>, 256, lds);
  }
}
template <int DPAD>
void launch_propose(const ProposeArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_propose<DPAD>, dim3((unsigned)a.NP), dim3(64), 0, st, a);
}
template <int DPAD>
void launch_rng_direct(const RngArgs& a, double* inc, hipStream_t st) {
  hipLaunchKernelGGL(k_rng_direct<DPAD>, dim3((unsigned)a.NP, (unsigned)((a.S + 15) / 16)), dim3(64), 0, st, a, inc);
  hipLaunchKernelGGL(k_rng_uniforms, dim3((unsigned)(((int64_t)a.S * a.NP + 255) / 256)), dim3(256), 0, st, a);
}
template <int DPAD>
void launch_rng(const RngArgs& a, hipStream_t st) {
  // (the 5 us of uniforms first: the long kernel then ends the generator stream's work of a block, and nothing small queues behind it)
  hipLaunchKernelGGL(k_rng_uniforms, dim3((unsigned)(((int64_t)a.S * a.NP + 255) / 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(k_rng<DPAD>, dim3((unsigned)a.NP, (unsigned)((a.S + 15) / 16)), dim3(64), 0, st, a);
}
template <int DPAD>
void launch_apply(const ApplyArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_apply<DPAD>, dim3((unsigned)a.NP), dim3(64), 0, st, a);
}
static bool adapt_circ() {
  static const bool on = !(getenv("TINYDA_ADAPT_CIRC") && atoi(getenv("TINYDA_ADAPT_CIRC")) == 0);
  return on;
}
template <int DPAD>
void launch_adapt(const AdaptArgs& a, hipStream_t st) {
  if constexpr (DPAD == 64) {  // TINYDA_ADAPT_SPLIT=1 (A/B): the ten tiles of a chain dealt to two waves, three waves per SIMD (k_adapt_split)
    static const bool split = getenv("TINYDA_ADAPT_SPLIT") && atoi(getenv("TINYDA_ADAPT_SPLIT")) == 1;
    if (split && a.do_am && !a.block_moments) {
      hipLaunchKernelGGL(k_adapt_split<DPAD>, dim3((unsigned)(2 * a.N)), dim3(64), 0, st, a);
      return;
    }
  }
  if (a.do_am && a.block_moments) { hipLaunchKernelGGL(k_adapt_block<DPAD>, dim3((unsigned)a.N), dim3(64), 0, st, a); return; }
  if constexpr (DPAD == 64) {  // TINYDA_ADAPT_CIRC=0 (A/B): the diagonal blocks as full tiles (rounds 2-4)
    if (adapt_circ() && a.do_am) { hipLaunchKernelGGL((k_adapt<DPAD, true>), dim3((unsigned)a.N), dim3(64), 0, st, a); return; }
  }
  hipLaunchKernelGGL(k_adapt<DPAD>, dim3((unsigned)a.N), dim3(64), 0, st, a);
}
template <int DPAD>
void launch_chol(const CholArgs& a, hipStream_t st) {
  if constexpr (DPAD == 64) {  // the blocked factorisation (k_chol_apply_blk without its second half)
    static const bool blocked_ok = !(getenv("TINYDA_CHOL_BLOCKED") && atoi(getenv("TINYDA_CHOL_BLOCKED")) == 0);
    if (blocked_ok) {
      hipLaunchKernelGGL((k_chol_apply_blk<DPAD, false>), dim3((unsigned)a.N), dim3(64), 0, st, a, ApplyArgs{});
      return;
    }
  }
  hipLaunchKernelGGL(k_chol<DPAD>, dim3((unsigned)a.N), dim3(64), 0, st, a);
}
template <int DPAD>
void launch_chol_apply(const CholArgs& a, const ApplyArgs& ap, hipStream_t st) {
  if constexpr (DPAD == 64) {  // four 16-column panels: the blocked kernel (TINYDA_CHOL_BLOCKED=0: the row-per-lane one, for A/B measurements)
    static const bool blocked_ok = !(getenv("TINYDA_CHOL_BLOCKED") && atoi(getenv("TINYDA_CHOL_BLOCKED")) == 0);
    if (blocked_ok) {
      hipLaunchKernelGGL(k_chol_apply_blk<DPAD>, dim3((unsigned)ap.NP), dim3(64), 0, st, a, ap);
      return;
    }
  }
  hipLaunchKernelGGL(k_chol_apply<DPAD>, dim3((unsigned)ap.NP), dim3(64), 0, st, a, ap);
}

// the period boundary of the single-level AdaptiveMetropolis pipeline in one launch (k_adapt_chol_apply, DPAD = 64 only)
template <int DPAD>
bool launch_adapt_chol_apply(const AdaptArgs& aa, const CholArgs& ca, const ApplyArgs& ap, hipStream_t st) {
  if constexpr (DPAD == 64) {
    static const bool ok = !(getenv("TINYDA_FUSE_ADAPT_CHOL") && atoi(getenv("TINYDA_FUSE_ADAPT_CHOL")) == 0) &&  // A/B switches
                           !(getenv("TINYDA_CHOL_BLOCKED") && atoi(getenv("TINYDA_CHOL_BLOCKED")) == 0);
    if (ok && aa.do_am && !aa.block_moments) {
      if (adapt_circ()) hipLaunchKernelGGL((k_adapt_chol_apply<DPAD, true>), dim3((unsigned)ap.NP), dim3(64), 0, st, aa, ca, ap);
      else hipLaunchKernelGGL((k_adapt_chol_apply<DPAD, false>), dim3((unsigned)ap.NP), dim3(64), 0, st, aa, ca, ap);
      return true;
    }
  }
  return false;
}

// ---- 65 .. 128 parameters (tda_kernels_wide.h): the same launch points, other kernels ----
template <>
void launch_steps<128>(const StepArgs& a, int64_t tiles, size_t lds, hipStream_t st) {
  auto kern = &k_mh_steps<128, 4, false>;  // one 512-register wave per SIMD (set_proposal admits GaussianRandomWalk / CrankNicolson / AdaptiveMetropolis)
  if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, a);
}
static WideApplyArgs wide_apply_args(const ApplyArgs& a) {
  WideApplyArgs w{};
  w.NP = a.NP;
  w.S = a.S;
  w.fac = a.Lk;
  w.ud = a.ud;
  w.sel = a.sel;
  w.chain_stride = a.L_stride ? 1 : 0;
  w.NPf = a.NPf;
  w.zf = a.zf;
  w.inc = a.inc;
  return w;
}
template <>
void launch_apply<128>(const ApplyArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_wide_apply<WIDE_T>, dim3((unsigned)a.NP), dim3(64), 0, st, wide_apply_args(a));
}
// replay mode (and TINYDA_SPLIT_PROPOSE=0): uniforms, the normals as fragments (recorded ones converted, or k_rng<128>), the product
template <>
void launch_propose<128>(const ProposeArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_wide_uniforms, dim3((unsigned)(((int64_t)a.S * a.NP + 255) / 256)), dim3(256), 0, st, a);
  const dim3 g((unsigned)a.NP, (unsigned)((a.S + 15) / 16));
  if (a.z_replay) {
    WideReplayArgs w{};
    w.N = a.N;
    w.NP = a.NP;
    w.d = a.d;
    w.S = a.S;
    w.z = a.z_replay;
    w.zf = a.zf_tmp;
    hipLaunchKernelGGL(k_wide_replay_frags<WIDE_T>, g, dim3(64), 0, st, w);
  } else {
    RngArgs ra{};
    ra.N = a.N;
    ra.NP = a.NP;
    ra.chain_offset = a.chain_offset;
    ra.d = a.d;
    ra.S = a.S;
    ra.step0 = a.step0;
    ra.seed = a.seed;
    ra.zf = a.zf_tmp;
    ra.z_export = a.z_export;
    hipLaunchKernelGGL(k_rng<128>, g, dim3(64), 0, st, ra);
  }
  ApplyArgs ap{};
  ap.NP = a.NP;
  ap.S = a.S;
  ap.Lk = a.Lk;
  ap.L_stride = a.L_stride;
  ap.zf = a.zf_tmp;
  ap.inc = a.inc;
  ap.ud = a.ud;
  ap.sel = a.sel;
  ap.NPf = a.NPf;
  launch_apply<128>(ap, st);
}
template <>
void launch_adapt<128>(const AdaptArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_wide_adapt<WIDE_T>, dim3((unsigned)a.N), dim3(128), 0, st, a);  // two waves per chain
}
template <>
void launch_chol<128>(const CholArgs& a, hipStream_t st) {  // C <- Sigma: the error model's factorisation (AemRefreshArgs::wide)
  AemRefreshArgs ra{};
  ra.N = a.N;
  ra.NP = a.NP;
  ra.m = a.d;
  ra.MP = 128;
  ra.nsum = 1;
  ra.cov = a.pad_cov;
  ra.sig[0] = const_cast<double*>(a.am_sigma);
  ra.V = a.Lk;
  ra.b_t = 1;
  ra.wide = 1;
  ra.Ud = a.ud;
  ra.sel = a.sel;
  ra.flags = a.flags;
  hipLaunchKernelGGL((k_aem_refresh<WIDE_T, 1>), dim3((unsigned)a.N), dim3(64), 0, st, ra);
}
template <>
void launch_chol_apply<128>(const CholArgs& a, const ApplyArgs& ap, hipStream_t st) {
  launch_chol<128>(a, st);
  launch_apply<128>(ap, st);
}

// k_aem_refresh<T, NSUM> for the engine's row stride (64 / 128 -> 4 / 8 tile rows; 256: k_aem_refresh_big<16, NSUM>) and the number of trackers summed
static int launch_aem_refresh(const tda::AemRefreshArgs& ra, hipStream_t st) {
  using namespace tda;
  const dim3 g((unsigned)ra.N), b(64);
#define TDA_AEMR(TT, NS) hipLaunchKernelGGL((k_aem_refresh<TT, NS>), g, b, 0, st, ra)
  if (ra.MP == 256) {  // 129 .. 256 outputs: the factorisation with the chain's factor buffer as its workspace
    if (ra.nsum == 1) hipLaunchKernelGGL((k_aem_refresh_big<16, 1>), g, b, 0, st, ra);
    else if (ra.nsum == 2) hipLaunchKernelGGL((k_aem_refresh_big<16, 2>), g, b, 0, st, ra);
    else hipLaunchKernelGGL((k_aem_refresh_big<16, 3>), g, b, 0, st, ra);
  } else if (ra.MP == 64) {
    if (ra.nsum == 1) TDA_AEMR(4, 1); else if (ra.nsum == 2) TDA_AEMR(4, 2); else TDA_AEMR(4, 3);
  } else {
    if (ra.nsum == 1) TDA_AEMR(8, 1); else if (ra.nsum == 2) TDA_AEMR(8, 2); else TDA_AEMR(8, 3);
  }
#undef TDA_AEMR
  return TDA_OK;
}

// Two levels with a small coarse model (m0 <= 256), a fixed subchain length, iso / diag noise and a diagonal prior
// (BASELINE config 3) run on the pipelined 8-wave Delayed-Acceptance kernel; everything else (3-4 levels, error model,
// randomised subchains, dense prior, larger coarse models) on the generic one.
inline bool da_lean_eligible(const MLArgs& a) {
  // (the diagonal error model, aem_on == 2, runs its base subchains here with the level actions sequenced by the host)
  if ((a.nlev != 2 && a.nlev != 3) || a.aem_on == 1 || (a.aem_on == 2) == (a.cascade != 0) || a.randomize || a.pr.kind == PRIOR_DENSE) return false;
  // the coarse operator lives in registers: two 16-row blocks per wave (one with a third level's state beside it)
  if (a.lv[0].ncb > ((a.nlev == 2 && a.aem_on != 2) ? 16 : 8)) return false;
  if (a.aem_on == 2 && a.d <= 32) return false;  // (its instances exist for the 64-parameter padding only)
  for (int k = 0; k < a.nlev; ++k)
    if (a.lv[k].noise_kind != 0 && a.lv[k].noise_kind != 1) return false;
  static const bool off = getenv("TINYDA_DA_LEAN") && atoi(getenv("TINYDA_DA_LEAN")) == 0;  // A/B switch for measurements
  return !off;
}

// free_regs != nullptr: no launch; *free_regs = the vector registers per SIMD lane that one resident tile of the kernel this call
// would launch leaves free (512 - waves per SIMD x allocated registers) -- run_multilevel puts the next block's draws on a second
// stream when the generator's 64-register waves fit beside the tile
// lean224: the caller can put the next block's draws beside this kernel (run_multilevel): where the 224-register entry point of the
// two-level kernel exists it is launched instead of the plain one (TINYDA_DA_R224=0: never)
template <int DPAD>
int launch_ml(const MLArgs& a, int64_t tiles, size_t lds, hipStream_t st, int* free_regs = nullptr, bool lean224 = false) {
  static const bool r224_ok = !(getenv("TINYDA_DA_R224") && atoi(getenv("TINYDA_DA_R224")) == 0);
  auto regs_left = [&](const void* kern, int waves_per_simd) -> int {
    hipFuncAttributes fa{};
    HIP_TRY(hipFuncGetAttributes(&fa, kern));
    *free_regs = 512 - waves_per_simd * ((fa.numRegs + 7) / 8 * 8);
    return TDA_OK;
  };
  // (the lean kernel keeps 64 KB of model outputs per tile in LDS: with very long data vectors staged beside them it does not fit)
  const size_t lds8 = (size_t)da_lds_doubles<DPAD>(a.lds_total) * sizeof(double);
  if (da_lean_eligible(a) && lds8 <= 160 * 1024) {
    const bool pcn = a.prop_kind == TDA_PROP_PCN, dg0 = a.lv[0].noise_kind == 1, one = a.lv[0].ncb <= 8;
#define TDA_DA_LAUNCH(RBV, PCNV, NZV, NLV)                                                                                         \
  do {                                                                                                                             \
    if (free_regs) return regs_left(reinterpret_cast<const void*>(&k_da_steps<DPAD, RBV, PCNV, NZV, NLV>), 2);                     \
    if (lds8 > 64 * 1024)                                                                                                          \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_da_steps<DPAD, RBV, PCNV, NZV, NLV>),                           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));                                         \
    hipLaunchKernelGGL((k_da_steps<DPAD, RBV, PCNV, NZV, NLV>), dim3((unsigned)tiles), dim3(512), lds8, st, a);                    \
  } while (0)
    auto go224 = [&](auto kern) -> int {  // (the 224-register entry points)
      if (free_regs) return regs_left(reinterpret_cast<const void*>(kern), 2);
      if (lds8 > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));
      hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds8, st, a);
      return TDA_OK;
    };
    (void)go224;
    if (a.aem_on == 2) {  // (instantiated for 33..64 parameters and one operator block per wave: da_lean_eligible)
      if constexpr (DPAD == 64) {
        if (a.nlev == 3) { if (pcn) TDA_DA_LAUNCH(1, true, 2, 3); else TDA_DA_LAUNCH(1, false, 2, 3); }
        else { if (pcn) TDA_DA_LAUNCH(1, true, 2, 2); else TDA_DA_LAUNCH(1, false, 2, 2); }
      }
    } else if (a.nlev == 3) {  // (one operator block per wave: da_lean_eligible)
      if constexpr (DPAD == 64) {
        if (lean224 && r224_ok) {
          if (pcn) return dg0 ? go224(&k_da_steps_r224<DPAD, 1, true, 1, 3>) : go224(&k_da_steps_r224<DPAD, 1, true, 0, 3>);
          return dg0 ? go224(&k_da_steps_r224<DPAD, 1, false, 1, 3>) : go224(&k_da_steps_r224<DPAD, 1, false, 0, 3>);
        }
      }
      if (pcn) { if (dg0) TDA_DA_LAUNCH(1, true, 1, 3); else TDA_DA_LAUNCH(1, true, 0, 3); }
      else { if (dg0) TDA_DA_LAUNCH(1, false, 1, 3); else TDA_DA_LAUNCH(1, false, 0, 3); }
    } else if (one) {
      if (pcn) { if (dg0) TDA_DA_LAUNCH(1, true, 1, 2); else TDA_DA_LAUNCH(1, true, 0, 2); }
      else { if (dg0) TDA_DA_LAUNCH(1, false, 1, 2); else TDA_DA_LAUNCH(1, false, 0, 2); }
    } else {
      if constexpr (DPAD == 64) {  // (smaller paddings leave the room as they are; with diagonal noise the budget would spill 12)
        if (lean224 && r224_ok && !dg0) {
          return pcn ? go224(&k_da_steps_r224<DPAD, 2, true, 0, 2>) : go224(&k_da_steps_r224<DPAD, 2, false, 0, 2>);
        }
      }
      if (pcn) { if (dg0) TDA_DA_LAUNCH(2, true, 1, 2); else TDA_DA_LAUNCH(2, true, 0, 2); }
      else { if (dg0) TDA_DA_LAUNCH(2, false, 1, 2); else TDA_DA_LAUNCH(2, false, 0, 2); }
    }
#undef TDA_DA_LAUNCH
    return TDA_OK;
  }
  auto go = [&](auto kern) -> int {
    if (free_regs) return regs_left(reinterpret_cast<const void*>(kern), 1);
    if (lds > 64 * 1024)  // beyond the default dynamic-LDS window (the residual tile of an error model at several hundred outputs)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, a);
    return TDA_OK;
  };
  // host-sequenced level actions (the dense error model): the kernel advances the base level only and would carry the states
  // and densities of the levels above through the step loop untouched -- in the three- and four-level instances that is 33 / 91
  // spilled registers whose reloads drain the prefetch of the per-chain precision matrices; the one-level instance leaves them
  // in memory
  bool dense = false;  // a dense observation covariance on some level: the instances that carry the Sigma^-1 quadratic form
  for (int k = 0; k < a.nlev; ++k) dense = dense || a.lv[k].noise_kind == TDA_NOISE_DENSE;
  if (dense) {
    switch (a.nlev) {
      case 2: return go(&k_ml_steps<DPAD, 2, 4, true>);
      case 3: return go(&k_ml_steps<DPAD, 3, 4, true>);
      case 4: return go(&k_ml_steps<DPAD, 4, 4, true>);
      default: return fail(TDA_ERR_UNSUPPORTED, "a dense observation covariance in a hierarchy: at most %d levels", (int)AEM_MAXLEV);
    }
  }
  if (!a.cascade && !a.randomize) return go(&k_ml_steps<DPAD, 1>);
  switch (a.nlev) {
    case 2: return go(&k_ml_steps<DPAD, 2>);
    case 3: return go(&k_ml_steps<DPAD, 3>);
    case 4: return go(&k_ml_steps<DPAD, 4>);
    case 5: return go(&k_ml_steps<DPAD, 5>);  // (0.5: five and six levels -- 51 / 95 spilled registers at 64 parameters, none at 32)
    default: return go(&k_ml_steps<DPAD, 6>);
  }
}
// 65 .. 128 parameters: Delayed Acceptance and MLDA on the generic level kernel, one 512-register wave per SIMD, ONE observation
// block in flight per wave (k_ml_steps: paired blocks and the prior in registers are 256 registers at this width -- with them the
// two-level instance spilled 385; without, two levels spill nothing, three 2 registers, four 44).  tda_engine_init refuses error
// models and everything but linear levels above 64 parameters.
template <>
int launch_ml<128>(const MLArgs& a, int64_t tiles, size_t lds, hipStream_t st, int* free_regs, bool) {
  if (free_regs) {
    *free_regs = 0;  // (no room for the generator beside it)
    return TDA_OK;
  }
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, a);
    return TDA_OK;
  };
  if (!a.cascade && !a.randomize) return go(&k_ml_steps<128, 1>);  // (dense error model: the base level alone, level actions sequenced by the host)
  switch (a.nlev) {
    case 2: return go(&k_ml_steps<128, 2>);
    case 3: return go(&k_ml_steps<128, 3>);
    case 4: return go(&k_ml_steps<128, 4>);
    default: return fail(TDA_ERR_UNSUPPORTED, "more than 64 parameters: hierarchies of two to four levels");
  }
}

template <int DPAD>
void launch_dz_draw(const DreamDrawArgs& a, hipStream_t st, const DreamFuseArgs* fuse = nullptr) {
  const dim3 g((unsigned)(a.NP / dz_chains_per_wave<DPAD>()));
  if constexpr (DPAD == 32) {
    if (fuse) {  // draws AND steps of the block in one launch (run_dreamz decides: shared archive, Rosenbrock, diagonal prior, own variates)
      hipLaunchKernelGGL((k_dreamz_draw<32, false, true>), g, dim3(64), 0, st, a, *fuse);
      return;
    }
  }
  const DreamFuseArgs none{};
  if (a.dist_ranks) hipLaunchKernelGGL((k_dreamz_draw<DPAD, true>), g, dim3(64), 0, st, a, none);
  else hipLaunchKernelGGL((k_dreamz_draw<DPAD, false>), g, dim3(64), 0, st, a, none);
}
template <int DPAD>
void launch_dz_steps(const DreamStepArgs& a, size_t lds, hipStream_t st) {
  // the built-in non-linear model under a diagonal prior: chains as lane groups of a wave, no tile, no barriers
  // (TINYDA_DZ_WAVE=0: the 16-chain tile kernel, for A/B measurements)
  static const bool wave_ok = [] {
    const char* v = getenv("TINYDA_DZ_WAVE");
    return !(v && v[0] == '0');
  }();
  if (wave_ok && a.model == MODEL_ROSENBROCK && a.pr.kind != PRIOR_DENSE) {
    constexpr int CPW = 64 / (DPAD >= 16 ? 16 : DPAD);
    hipLaunchKernelGGL(k_dreamz_steps_wave<DPAD>, dim3((unsigned)(a.NP / CPW)), dim3(64), 0, st, a);
    return;
  }
  if (a.model == MODEL_LINEAR && a.lv.noise_kind == TDA_NOISE_DENSE) {
    if (lds > 64 * 1024)  // beyond the default dynamic-LDS window (the residual tile at several hundred outputs)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dreamz_steps<DPAD, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_dreamz_steps<DPAD, true>), dim3((unsigned)(a.NP / 16)), dim3(256), lds, st, a);
    return;
  }
  hipLaunchKernelGGL((k_dreamz_steps<DPAD, false>), dim3((unsigned)(a.NP / 16)), dim3(256), lds, st, a);
}
template <int DPAD>
void launch_colsum(const double* m, int64_t row0, int64_t nrows, double* partial, int64_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_colsum_partial<DPAD>, dim3((unsigned)nb), dim3(64), 0, st, m, row0, nrows, partial);
}
// `partial` holds nb chunk sums and has room for ceil(nb / COLSUM_FOLD) folded ones behind them (colsum_partial_doubles)
inline size_t colsum_partial_doubles(int64_t nb, int DP) { return (size_t)(nb + (nb + COLSUM_FOLD - 1) / COLSUM_FOLD) * 2 * DP; }
template <int DPAD>
void launch_colsum_final(double* partial, int64_t nb, double* zsum, double* zsq, hipStream_t st) {
  if (nb > 2 * COLSUM_FOLD) {  // many chunks: fold them over several workgroups first (a fixed order for a given count)
    const int64_t nf = (nb + COLSUM_FOLD - 1) / COLSUM_FOLD;
    double* folded = partial + (size_t)nb * 2 * DPAD;
    hipLaunchKernelGGL(k_colsum_fold<DPAD>, dim3((unsigned)nf), dim3(64), 0, st, partial, nb, folded);
    hipLaunchKernelGGL(k_colsum_final<DPAD>, dim3(1), dim3(64 * COLSUM_FINAL_WAVES), 0, st, folded, nf, zsum, zsq);
    return;
  }
  hipLaunchKernelGGL(k_colsum_final<DPAD>, dim3(1), dim3(64 * COLSUM_FINAL_WAVES), 0, st, partial, nb, zsum, zsq);
}
template <int DPAD>
void launch_dz_adapt(const DreamAdaptArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_dreamz_adapt<DPAD>, dim3((unsigned)a.N), dim3(64), 0, st, a);
}

#define DISPATCH_DPAD(dp, CALL)                  \
  switch (dp) {                                  \
    case 8: { constexpr int DPAD = 8; CALL; } break;   \
    case 16: { constexpr int DPAD = 16; CALL; } break; \
    case 32: { constexpr int DPAD = 32; CALL; } break; \
    default: { constexpr int DPAD = 64; CALL; } break; \
  }

// the single-level sites (run_single, evaluate): 65 .. 128 parameters too (launch_*<128>: tda_kernels_wide.h)
#define DISPATCH_DPAD_W(dp, CALL)                    \
  switch (dp) {                                      \
    case 8: { constexpr int DPAD = 8; CALL; } break;     \
    case 16: { constexpr int DPAD = 16; CALL; } break;   \
    case 32: { constexpr int DPAD = 32; CALL; } break;   \
    case 128: { constexpr int DPAD = 128; CALL; } break; \
    default: { constexpr int DPAD = 64; CALL; } break;   \
  }

struct ScopedTimer {
  tda_engine* e;
  TimedLaunch tl{};
  bool on;
  ScopedTimer(tda_engine* e_, int kind) : e(e_), on(e_->profiling) {
    if (on) {
      (void)hipEventCreate(&tl.a);
      (void)hipEventCreate(&tl.b);
      tl.kind = kind;
      (void)hipEventRecord(tl.a, e->stream);
    }
  }
  ~ScopedTimer() {
    if (on) {
      (void)hipEventRecord(tl.b, e->stream);
      e->timed.push_back(tl);
    }
  }
};

size_t steps_lds_bytes(const tda_engine* e, const Level& lv) {
  const bool diag = lv.noise_kind == TDA_NOISE_DIAG;
  const int prow = e->prior_kind == PRIOR_DENSE ? e->prior_ncb * 16 : 0;
  size_t nd = (size_t)16 * (e->DP + 2) + 128 + 2 * e->DP + lv.m_pad + (diag ? lv.m_pad : 0) + prow;
  if (lv.noise_kind == TDA_NOISE_DENSE) nd += (size_t)16 * (lv.m_pad + 2);
  return nd * sizeof(double);
}

void fill_level(const tda_engine* e, const Level& lv, StepArgs& a) {
  a.lv.Apk = lv.Apk.p;
  a.lv.ytil = lv.ytil.p;
  a.lv.w = lv.w.p;
  a.lv.Ppk = lv.Ppk.p;
  a.lv.ncb = lv.ncb;
  a.lv.m_pad = lv.m_pad;
  a.lv.noise_kind = lv.noise_kind;
  a.lv.var = lv.var;
  a.pr.mean = e->prior_mean.p;
  a.pr.pinv = e->prior_pinv.p;
  a.pr.Wpk = e->prior_Wpk.p;
  a.pr.wmu = e->prior_wmu.p;
  a.pr.ncb = e->prior_ncb;
  a.pr.kind = (e->prior_kind == PRIOR_DIAG && e->prior_is_standard) ? PRIOR_STANDARD : e->prior_kind;
  a.pr.logconst = e->prior_logconst;
  a.pr.lo = e->prior_bounded ? e->prior_lo.p : nullptr;
  a.pr.hi = e->prior_bounded ? e->prior_hi.p : nullptr;
  a.N = e->N;
  a.NP = e->NP;
  a.d = e->d;
}

constexpr int MODEL_USER = 2;

int fill_user_args(tda_engine* e, const Level& lv, UserStepArgs& ua) {
  if (e->prior_kind == PRIOR_DENSE) return fail(TDA_ERR_UNSUPPORTED, "source-defined forward models need a diagonal prior covariance");
  ua.N = e->N;
  ua.NP = e->NP;
  ua.d = e->d;
  ua.DP = e->DP;
  ua.m = lv.m;
  ua.data = lv.udata.p;
  ua.w = lv.noise_kind == TDA_NOISE_DIAG ? lv.uw.p : nullptr;
  ua.var = lv.var;
  ua.pr_mean = e->prior_mean.p;
  ua.pr_pinv = e->prior_pinv.p;
  ua.pr_lo = e->prior_bounded ? e->prior_lo.p : nullptr;
  ua.pr_hi = e->prior_bounded ? e->prior_hi.p : nullptr;
  ua.logconst = e->prior_logconst;
  return TDA_OK;
}

constexpr int MODEL_CALLBACK = 3;

void fill_ext_args(tda_engine* e, const Level& lv, ExtArgs& xa) {
  xa.N = e->N;
  xa.NP = e->NP;
  xa.d = e->d;
  xa.DP = e->DP;
  xa.m = lv.m;
  xa.prop = lv.cb_prop.p;
  xa.F = lv.cb_F.p;
  xa.data = lv.udata.p;
  xa.w = lv.noise_kind == TDA_NOISE_DIAG ? lv.uw.p : nullptr;
  xa.Pd = lv.noise_kind == TDA_NOISE_DENSE ? lv.Pd.p : nullptr;
  xa.var = lv.var;
  xa.pr_mean = e->prior_mean.p;
  xa.pr_pinv = e->prior_pinv.p;
  xa.pr_lo = e->prior_bounded ? e->prior_lo.p : nullptr;
  xa.pr_hi = e->prior_bounded ? e->prior_hi.p : nullptr;
  xa.logconst = e->prior_logconst;
}

// AdaptiveGaussianLogLike on a callback / source-defined level: host and device copies of Sigma_e and the data vector in
// the error-model layout (row stride 64 / 128), as tda_engine_set_level keeps them for linear levels
int ext_level_adaptive(tda_engine* e, Level& lv, int m, const double* data, const double* cov) {
  if (m > AEM_MP_MAX_EXT)
    return fail(TDA_ERR_UNSUPPORTED, "AdaptiveGaussianLogLike on a callback / source-defined level is limited to m <= %d observations", (int)AEM_MP_MAX_EXT);
  std::vector<double> Lc;
  if (!cholesky_host(cov, m, Lc)) return fail(TDA_ERR_NUMERIC, "noise covariance is not positive definite");
  const int MP = (m <= 64 && e->DP <= 64) ? 64 : (m <= 128 ? 128 : 256);  // (65 .. 128 parameters: a thread of the error-model kernels is a parameter too)
  lv.em_ld = MP;
  lv.cov_h.assign(cov, cov + (size_t)m * m);
  lv.ytil_h.assign(MP, 0.0);
  lv.data_h.assign(MP, 0.0);
  for (int i = 0; i < m; ++i) lv.ytil_h[i] = lv.data_h[i] = data[i];
  // Sigma_e as the UPPER 16 x 16 tiles k_aem_refresh factors (tda_kernels_aemr.h), identity in the padding rows / columns
  std::vector<double> c64(tda::aemr_v_doubles(MP), 0.0);
  for (int i = 0; i < MP; ++i)
    for (int j = i; j < MP; ++j) c64[tda::aemr_u_offset(MP, i, j)] = (i < m && j < m) ? cov[(size_t)i * m + j] : (i == j ? 1.0 : 0.0);
  int rc;
  if ((rc = lv.cov64.upload(c64))) return rc;
  if ((rc = lv.data64.upload(lv.data_h))) return rc;
  if ((rc = lv.ytil64.upload(lv.ytil_h))) return rc;
  lv.var = 1.0;
  return TDA_OK;
}

// DefaultGaussianLogLike (dense data covariance) on a callback / source-defined level: Sigma^-1 through the Cholesky factor
int ext_level_dense(tda_engine* /*e*/, Level& lv, int m, const double* cov) {
  if (m > 2048) return fail(TDA_ERR_UNSUPPORTED, "dense noise on callback / source-defined levels: m <= 2048");
  std::vector<double> Lc, W, P((size_t)m * m, 0.0);
  if (!cholesky_host(cov, m, Lc)) return fail(TDA_ERR_NUMERIC, "noise covariance is not positive definite");
  tri_inverse_host(Lc, m, W);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j <= i; ++j) {
      double sacc = 0.0;
      for (int k = i; k < m; ++k) sacc += W[(size_t)k * m + i] * W[(size_t)k * m + j];
      P[(size_t)i * m + j] = P[(size_t)j * m + i] = sacc;
    }
  lv.var = 1.0;
  return lv.Pd.upload(P);
}

// F[N][m] <- model(prop[N][d]) of a level whose model lives outside the engine's kernels: a batched host callback (through
// page-locked staging buffers, one synchronisation) or a source-defined model (tda_user_eval, stays on the stream)
int ext_model_outputs(tda_engine* e, const Level& lv) {
  if (lv.model == MODEL_USER) return launch_user_eval(lv.ufn_eval, e->N, e->d, lv.m, lv.cb_prop.p, lv.cb_F.p, e->stream);
  if (lv.model == MODEL_LINEAR) {
    if (lv.Apk.p) {  // on the matrix cores, every operator fragment serving a 16-chain tile
      DISPATCH_DPAD_W(e->DP, hipLaunchKernelGGL((k_linear_outputs<DPAD>), dim3((unsigned)(e->NP / 16)), dim3(256), 0, e->stream, (long long)e->N,
                                              e->d, lv.m, lv.Apk.p, lv.ncb, lv.b_dev.p, lv.cb_prop.p, e->d, lv.cb_F.p));
      return TDA_OK;
    }
    hipLaunchKernelGGL(k_ext_linear_eval, dim3((unsigned)((e->N + EXT_WAVES - 1) / EXT_WAVES)), dim3(64 * EXT_WAVES), 0, e->stream,
                       (long long)e->N, e->d, lv.m, lv.A_dev.p, lv.b_dev.p, lv.cb_prop.p, lv.cb_F.p);
    return TDA_OK;
  }
  HIP_TRY(hipMemcpyAsync(lv.cb_theta_h, lv.cb_prop.p, (size_t)e->N * e->d * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const int crc = lv.cb_fn(lv.cb_user, lv.cb_theta_h, lv.cb_F_h, e->N, e->d, lv.m);
  if (crc != 0) return fail(TDA_ERR_CALLBACK, "the forward-model callback returned %d", crc);
  HIP_TRY(hipMemcpyAsync(lv.cb_F.p, lv.cb_F_h, (size_t)e->N * lv.m * sizeof(double), hipMemcpyHostToDevice, e->stream));
  return TDA_OK;
}

// outputs of a linear level with a packed operator at the states theta[N][ld] (a level's slab of the state array), into out[N][m]
void linear_outputs_at(tda_engine* e, const Level& lv, const double* theta, int ld, double* out) {
  DISPATCH_DPAD_W(e->DP, hipLaunchKernelGGL((k_linear_outputs<DPAD>), dim3((unsigned)(e->NP / 16)), dim3(256), 0, e->stream, (long long)e->N,
                                          e->d, lv.m, lv.Apk.p, lv.ncb, lv.b_dev.p, theta, ld, out));
}

// one step of such a level: proposals -> model outputs -> accept (xa.s, xa.mode set)
int ext_step(tda_engine* e, const Level& lv, const ExtArgs& xa) {
  if (e->prior_kind == PRIOR_DENSE) return fail(TDA_ERR_UNSUPPORTED, "callback forward models need a diagonal prior covariance");
  const unsigned grid = (unsigned)((e->N + EXT_WAVES - 1) / EXT_WAVES);
  hipLaunchKernelGGL(k_ext_propose, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, xa);
  int mrc = ext_model_outputs(e, lv);
  if (mrc) return mrc;
  hipLaunchKernelGGL(k_ext_accept, dim3(grid), dim3(64 * EXT_WAVES), xa.Pd ? (size_t)EXT_WAVES * lv.m * sizeof(double) : 0, e->stream, xa);
  HIP_TRY(hipGetLastError());
  return TDA_OK;
}

int launch_eval(tda_engine* e, int level, double* theta, double* lp, double* ll) {
  if (e->levels[level].model == MODEL_CALLBACK || (e->levels[level].model == MODEL_USER && e->levels[level].noise_kind == TDA_NOISE_DENSE)) {
    ExtArgs xa{};
    fill_ext_args(e, e->levels[level], xa);
    xa.mode = 1;
    xa.theta = theta;
    xa.lp = lp;
    xa.ll = ll;
    xa.scaling = e->scaling.p;
    return ext_step(e, e->levels[level], xa);
  }
  if (e->levels[level].model == MODEL_USER) {
    UserStepArgs ua{};
    int rc = fill_user_args(e, e->levels[level], ua);
    if (rc) return rc;
    ua.S = 1;
    ua.mode = 1;
    ua.theta = theta;
    ua.lp = lp;
    ua.ll = ll;
    ua.scaling = e->scaling.p;
    return launch_user_steps(e->levels[level].ufn, ua, e->stream);
  }
  StepArgs a{};
  fill_level(e, e->levels[level], a);
  a.S = 1;
  a.mode = MODE_EVAL;
  a.prop_kind = TDA_PROP_GRW;
  a.theta = theta;
  a.lp = lp;
  a.ll = ll;
  a.scaling = e->scaling.p;
  const size_t lds = steps_lds_bytes(e, e->levels[level]);
  DISPATCH_DPAD_W(e->DP, launch_steps<DPAD>(a, e->NP / 16, lds, e->stream));
  HIP_TRY(hipGetLastError());
  return TDA_OK;
}

// copy [n][d] caller layout -> [NP][DP] padded device layout
int upload_states(tda_engine* e, const double* src, int64_t n, double* dst) {
  std::vector<double> h((size_t)e->NP * e->DP, 0.0);
  std::vector<double> tmp;
  const double* hs = src;
  if (is_device_ptr(src)) {
    tmp.resize((size_t)n * e->d);
    HIP_TRY(hipMemcpy(tmp.data(), src, tmp.size() * sizeof(double), hipMemcpyDeviceToHost));
    hs = tmp.data();
  }
  for (int64_t c = 0; c < n; ++c)
    for (int j = 0; j < e->d; ++j) h[(size_t)c * e->DP + j] = hs[(size_t)c * e->d + j];
  HIP_TRY(hipMemcpyAsync(dst, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return TDA_OK;
}

int copy_out(tda_engine* e, void* dst, const void* src_dev, size_t bytes) {
  if (!dst || bytes == 0) return TDA_OK;
  HIP_TRY(hipMemcpyAsync(dst, src_dev, bytes, is_device_ptr(dst) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                         e->stream));
  return TDA_OK;
}

}  // namespace

// Host-side unpacking of per-chain matrices is split over a few threads: f(c0, c1) handles chains [c0, c1).
template <class F>
static void host_chain_ranges(int64_t n, F f) {
  const int64_t nt = std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 16, n / 64}));
  if (nt == 1) return f(0, n);
  std::vector<std::thread> pool;
  for (int64_t t = 0; t < nt; ++t) pool.emplace_back(f, n * t / nt, n * (t + 1) / nt);
  for (auto& th : pool) th.join();
}

namespace {
// progress mark at the end of a block: mean of its accept flags and the iteration count, into device-visible host memory
__global__ void k_progress_mark(const uint8_t* __restrict__ acc, int64_t count, double iters_total, double* __restrict__ out_h) {
  __shared__ unsigned int part[256];
  unsigned int sum = 0;
  if (acc)
    for (int64_t i = threadIdx.x; i < count; i += 256) sum += acc[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out_h[1] = acc ? (double)part[0] / (double)count : -1.0;
    __threadfence_system();
    out_h[0] = iters_total;
  }
}

// rows src[(s0 + j * thin)][row_bytes] -> dst[j][row_bytes], j < nkeep (row_bytes a multiple of 8 or, for flags, any size)
__global__ void k_thin_rows(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, int64_t row_bytes, int64_t s0, int thin, int64_t nkeep) {
  const int64_t j = blockIdx.y;
  if (j >= nkeep) return;
  const uint8_t* sp = src + (size_t)(s0 + j * thin) * row_bytes;
  uint8_t* dp = dst + (size_t)j * row_bytes;
  if (((row_bytes | (int64_t)(uintptr_t)sp | (int64_t)(uintptr_t)dp) & 7) == 0) {
    const int64_t n8 = row_bytes >> 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x)
      reinterpret_cast<uint64_t*>(dp)[i] = reinterpret_cast<const uint64_t*>(sp)[i];
  } else {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < row_bytes; i += (int64_t)gridDim.x * blockDim.x) dp[i] = sp[i];
  }
}

// records of this run() that reach the caller under thinning: iterations g in [t, t + n) with (g + 1) % thin == 0
inline int64_t kept_records(int64_t t, int64_t n, int thin) { return thin <= 1 ? n : (t + n) / thin - t / thin; }

// kept rows of one block buffer -> the caller's buffer (device: one gather launch; host: one copy per kept row on `cs`)
int thin_out(tda_engine* e, void* dst, bool dst_dev, const void* blk, size_t row_bytes, int64_t s0, int64_t nkeep, hipStream_t cs) {
  if (!dst || nkeep <= 0) return TDA_OK;
  if (dst_dev) {
    const unsigned gx = (unsigned)std::min<int64_t>(64, std::max<int64_t>(1, (int64_t)(row_bytes / 8 + 255) / 256));
    hipLaunchKernelGGL(k_thin_rows, dim3(gx, (unsigned)nkeep), dim3(256), 0, cs, (uint8_t*)dst, (const uint8_t*)blk, (int64_t)row_bytes, s0, e->thin, nkeep);
    HIP_TRY(hipGetLastError());
    return TDA_OK;
  }
  for (int64_t j = 0; j < nkeep; ++j)
    HIP_TRY(hipMemcpyAsync((uint8_t*)dst + (size_t)j * row_bytes, (const uint8_t*)blk + (size_t)(s0 + j * e->thin) * row_bytes, row_bytes,
                           hipMemcpyDeviceToHost, cs));
  return TDA_OK;
}

// end of a block of S iterations: progress mark (when enabled)
int progress_mark(tda_engine* e, int64_t S, const uint8_t* blk_acc, int64_t n_flags) {
  e->prog_queued += S;
  if (!e->prog_h) return TDA_OK;
  hipLaunchKernelGGL(k_progress_mark, dim3(1), dim3(256), 0, e->stream, blk_acc, n_flags, (double)e->prog_queued, e->prog_h);
  HIP_TRY(hipGetLastError());
  return TDA_OK;
}
}  // namespace

extern "C" {

const char* tda_last_error(void) { return g_err.c_str(); }
const char* tda_version(void) { return "tinyda_amd 0.5 (gfx950)"; }

int tda_engine_create(const tda_config* cfg, tda_engine** out) {
  if (!cfg || !out) return fail(TDA_ERR_INVALID, "null argument");
  if (cfg->struct_size != sizeof(tda_config)) return fail(TDA_ERR_INVALID, "tda_config.struct_size mismatch");
  if (cfg->dim < 1 || cfg->dim > 128)
    return fail(TDA_ERR_UNSUPPORTED, "dim=%d outside the device engine's range 1..128", cfg->dim);
  if (cfg->n_chains < 1) return fail(TDA_ERR_INVALID, "n_chains must be >= 1");
  if (cfg->n_levels < 1 || cfg->n_levels > MAXLEV)
    return fail(TDA_ERR_UNSUPPORTED, "n_levels=%d outside 1..%d", cfg->n_levels, (int)MAXLEV);
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(TDA_ERR_HIP, "no HIP device visible: the MH engine has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(TDA_ERR_INVALID, "device %d of %d", cfg->device, ndev);
  HIP_TRY(hipSetDevice(cfg->device));
  tda_engine* e = new tda_engine();
  e->cfg = *cfg;
  e->d = cfg->dim;
  e->DP = dpad_for(cfg->dim);
  e->wide = cfg->dim > 64;
  e->N = cfg->n_chains;
  e->NP = (cfg->n_chains + 15) / 16 * 16;
  e->SMAX = cfg->block_steps > 0 ? cfg->block_steps : 128;
  e->levels.resize(cfg->n_levels);
  e->nlev = cfg->n_levels;
  if (cfg->stream) {
    e->stream = (hipStream_t)cfg->stream;
  } else {
    hipError_t er = hipStreamCreate(&e->stream);
    if (er != hipSuccess) {
      delete e;
      return fail(TDA_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(er));
    }
    e->own_stream = true;
  }
  *out = e;
  return TDA_OK;
}

void tda_engine_destroy(tda_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  (void)hipStreamSynchronize(e->stream);
  for (auto& t : e->timed) {
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  for (auto& lv : e->levels) {
    if (lv.umod) (void)hipModuleUnload(lv.umod);
    lv.release_callback_buffers();
  }
  if (e->copy_stream) {
    (void)hipStreamSynchronize(e->copy_stream);
    (void)hipStreamDestroy(e->copy_stream);
    for (int i = 0; i < 2; ++i) {
      (void)hipEventDestroy(e->ev_rec[i]);
      (void)hipEventDestroy(e->ev_cp[i]);
    }
  }
  for (int r = 0; r < tda::MAX_PEERS; ++r)
    if (e->dist_opened[r]) (void)hipIpcCloseMemHandle(e->dist_opened[r]);
  if (e->ev_dz_adapt) (void)hipEventDestroy(e->ev_dz_adapt);
  if (e->rng_stream) {
    (void)hipStreamSynchronize(e->rng_stream);
    (void)hipStreamDestroy(e->rng_stream);
    for (int i = 0; i < 2; ++i) {
      (void)hipEventDestroy(e->ev_rng[i]);
      (void)hipEventDestroy(e->ev_apply[i]);
      (void)hipEventDestroy(e->ev_steps[i]);
    }
  }
  if (e->own_stream) (void)hipStreamDestroy(e->stream);
  if (e->prog_h) (void)hipHostFree(e->prog_h);
  if (e->dist_ranks) e->arch.release();  // exported to the peers as an IPC handle: not a candidate for reuse
  g_pool_accepting = true;  // every stream of the engine is idle: its buffers may serve the next engine
  delete e;
  g_pool_accepting = false;
}

int64_t tda_release_cached_memory(void) { return (int64_t)g_pool.trim(); }

}  // extern "C"

// ---- the rest of the ABI, by family (textual includes: one translation unit) ----
#include "tda_host_setup.inc"
#include "tda_host_dreamz_archive.inc"
#include "tda_host_state.inc"
#include "tda_host_hierarchy_setup.inc"
#include "tda_host_init.inc"
#include "tda_host_single.inc"
#include "tda_host_multilevel.inc"
#include "tda_host_dreamz_run.inc"
#include "tda_host_query.inc"
#include "tda_host_inspect.inc"
