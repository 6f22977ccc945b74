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

int dpad_for(int d) { return d <= 8 ? 8 : d <= 16 ? 16 : d <= 32 ? 32 : 64; }

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
// out again -- zeroed, as fresh allocations are -- up to TINYDA_POOL_GB (default 8) GiB; tda_release_cached_memory() returns
// them to the driver, as does an allocation failure before it is reported.
struct DevPool {
  std::mutex mu;
  std::multimap<std::pair<int, size_t>, void*> idle;
  size_t held = 0, cap = 0;
  bool cap_read = false;
  size_t capacity() {
    if (!cap_read) {
      const char* v = getenv("TINYDA_POOL_GB");
      cap = (size_t)((v ? atof(v) : 8.0) * 1073741824.0);
      cap_read = true;
    }
    return cap;
  }
  void* take(int dev, size_t bytes) {
    std::lock_guard<std::mutex> g(mu);
    auto it = idle.find({dev, bytes});
    if (it == idle.end()) return nullptr;
    void* p = it->second;
    idle.erase(it);
    held -= bytes;
    return p;
  }
  bool give(int dev, size_t bytes, void* p) {
    std::lock_guard<std::mutex> g(mu);
    if (bytes < (64u << 10) || held + bytes > capacity()) return false;  // small buffers: the runtime's own sub-allocator is fast
    idle.insert({{dev, bytes}, p});
    held += bytes;
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
  DevBuf<double> A_rm, ytil64, data64, cov64;  // error-model copies, row stride em_ld
  DevBuf<double> Pd;                           // callback / source-defined level with dense noise: Sigma^-1 [m][m]
  DevBuf<double> A_dev, b_dev;                 // hierarchies: row-major [m][d] and offset [m] for k_ext_linear_eval (host-sequenced mode)
  int em_ld = 0;                               // 64 (m <= 64) or 128 (m <= 128); 0: level too large for an error model
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
  DevBuf<uint8_t> rec_acc;

  // multi-level state (n_levels > 1)
  int nlev = 1;
  int sl[tda::MAXLEV] = {1, 1, 1, 1};
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
  int64_t aem_bt[tda::MAXLEV] = {1, 1, 1, 1};
  DevBuf<int64_t> ml_sid;
  DevBuf<double> prior_W_rm;
  DevBuf<double> u_rep_lv[tda::MAXLEV], ridx_rep;
  int64_t u_rep_lv_n[tda::MAXLEV] = {0, 0, 0, 0}, ridx_rep_n = 0;
  int64_t u_rep_lv_pos[tda::MAXLEV] = {0, 0, 0, 0}, ridx_rep_pos = 0;

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
  const bool ow = a.prop_kind == TDA_PROP_OWCN && a.mode == MODE_STEP;  // + current-state tile and the state operator in LDS
  const bool ma = a.prop_kind == TDA_PROP_MALA && a.mode == MODE_STEP;  // + the gradient operator and 32 transition-density slots
  if (ow || ma) lds += ((size_t)16 * (DPAD + 2) + (size_t)DPAD * DPAD) * sizeof(double);
  auto go = [&](auto kern, unsigned threads, size_t bytes) {
    if (bytes > 64 * 1024)  // beyond the default dynamic-LDS window (gfx950 has 160 KiB per CU)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(threads), bytes, st, a);
  };
  if (eight) {
    const size_t l8 = lds + 2 * 64 * sizeof(double);  // two more [4][16] reduction slabs
    if (ind) go(&k_mh_steps<DPAD, 8, true>, 512, l8);
    else if (ow) go(&k_mh_steps<DPAD, 8, false, 1>, 512, l8);
    else if (ma) go(&k_mh_steps<DPAD, 8, false, 2>, 512, l8);
    else go(&k_mh_steps<DPAD, 8, false>, 512, l8);
  } else {
    if (ind) go(&k_mh_steps<DPAD, 4, true>, 256, lds);
    else if (ow) go(&k_mh_steps<DPAD, 4, false, 1>, 256, lds);
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
  hipLaunchKernelGGL(k_rng<DPAD>, dim3((unsigned)a.NP, (unsigned)((a.S + 15) / 16)), dim3(64), 0, st, a);
  hipLaunchKernelGGL(k_rng_uniforms, dim3((unsigned)(((int64_t)a.S * a.NP + 255) / 256)), dim3(256), 0, st, a);
}
template <int DPAD>
void launch_apply(const ApplyArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_apply<DPAD>, dim3((unsigned)a.NP), dim3(64), 0, st, a);
}
template <int DPAD>
void launch_adapt(const AdaptArgs& a, hipStream_t st) {
  if (a.do_am && a.block_moments) hipLaunchKernelGGL(k_adapt_block<DPAD>, dim3((unsigned)a.N), dim3(64), 0, st, a);
  else hipLaunchKernelGGL(k_adapt<DPAD>, dim3((unsigned)a.N), dim3(64), 0, st, a);
}
template <int DPAD>
void launch_chol(const CholArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(k_chol<DPAD>, dim3((unsigned)a.N), dim3(64), 0, st, a);
}
template <int DPAD>
void launch_chol_apply(const CholArgs& a, const ApplyArgs& ap, hipStream_t st) {
  hipLaunchKernelGGL(k_chol_apply<DPAD>, dim3((unsigned)ap.NP), dim3(64), 0, st, a, ap);
}

// dynamic LDS of k_aem_inverse: the 16 x 16 blocks on or below the diagonal, row stride 17
constexpr size_t aem_inverse_lds_bytes(int nb) { return (size_t)(nb * (nb + 1) / 2) * AEM_BS * sizeof(double); }

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

template <int DPAD>
int launch_ml(const MLArgs& a, int64_t tiles, size_t lds, hipStream_t st) {
  if (da_lean_eligible(a)) {
    const size_t lds8 = (size_t)da_lds_doubles<DPAD>(a.lds_total) * sizeof(double);
    const bool pcn = a.prop_kind == TDA_PROP_PCN, dg0 = a.lv[0].noise_kind == 1, one = a.lv[0].ncb <= 8;
#define TDA_DA_LAUNCH(RBV, PCNV, NZV, NLV)                                                                                         \
  do {                                                                                                                             \
    if (lds8 > 64 * 1024)                                                                                                          \
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_da_steps<DPAD, RBV, PCNV, NZV, NLV>),                           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds8));                                         \
    hipLaunchKernelGGL((k_da_steps<DPAD, RBV, PCNV, NZV, NLV>), dim3((unsigned)tiles), dim3(512), lds8, st, a);                    \
  } while (0)
    if (a.aem_on == 2) {  // (instantiated for 33..64 parameters and one operator block per wave: da_lean_eligible)
      if constexpr (DPAD == 64) {
        if (a.nlev == 3) { if (pcn) TDA_DA_LAUNCH(1, true, 2, 3); else TDA_DA_LAUNCH(1, false, 2, 3); }
        else { if (pcn) TDA_DA_LAUNCH(1, true, 2, 2); else TDA_DA_LAUNCH(1, false, 2, 2); }
      }
    } else if (a.nlev == 3) {  // (one operator block per wave: da_lean_eligible)
      if (pcn) { if (dg0) TDA_DA_LAUNCH(1, true, 1, 3); else TDA_DA_LAUNCH(1, true, 0, 3); }
      else { if (dg0) TDA_DA_LAUNCH(1, false, 1, 3); else TDA_DA_LAUNCH(1, false, 0, 3); }
    } else if (one) {
      if (pcn) { if (dg0) TDA_DA_LAUNCH(1, true, 1, 2); else TDA_DA_LAUNCH(1, true, 0, 2); }
      else { if (dg0) TDA_DA_LAUNCH(1, false, 1, 2); else TDA_DA_LAUNCH(1, false, 0, 2); }
    } else {
      if (pcn) { if (dg0) TDA_DA_LAUNCH(2, true, 1, 2); else TDA_DA_LAUNCH(2, true, 0, 2); }
      else { if (dg0) TDA_DA_LAUNCH(2, false, 1, 2); else TDA_DA_LAUNCH(2, false, 0, 2); }
    }
#undef TDA_DA_LAUNCH
    return TDA_OK;
  }
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024)  // beyond the default dynamic-LDS window (the residual tile of an error model at several hundred outputs)
      HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), lds, st, a);
    return TDA_OK;
  };
  switch (a.nlev) {
    case 2: return go(&k_ml_steps<DPAD, 2>);
    case 3: return go(&k_ml_steps<DPAD, 3>);
    default: return go(&k_ml_steps<DPAD, 4>);
  }
}

template <int DPAD>
void launch_dz_draw(const DreamDrawArgs& a, hipStream_t st) {
  if (a.dist_ranks) hipLaunchKernelGGL((k_dreamz_draw<DPAD, true>), dim3((unsigned)(a.NP / dz_chains_per_wave<DPAD>())), dim3(64), 0, st, a);
  else hipLaunchKernelGGL((k_dreamz_draw<DPAD, false>), dim3((unsigned)(a.NP / dz_chains_per_wave<DPAD>())), dim3(64), 0, st, a);
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
  hipLaunchKernelGGL(k_dreamz_steps<DPAD>, dim3((unsigned)(a.NP / 16)), dim3(256), lds, st, a);
}
template <int DPAD>
void launch_colsum(const double* m, int64_t row0, int64_t nrows, double* partial, int64_t nb, hipStream_t st) {
  hipLaunchKernelGGL(k_colsum_partial<DPAD>, dim3((unsigned)nb), dim3(64), 0, st, m, row0, nrows, partial);
}
template <int DPAD>
void launch_colsum_final(const double* partial, int64_t nb, double* zsum, double* zsq, hipStream_t st) {
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
int ext_level_adaptive(tda_engine* /*e*/, Level& lv, int m, const double* data, const double* cov) {
  if (m > AEM_MP_MAX) return fail(TDA_ERR_UNSUPPORTED, "AdaptiveGaussianLogLike on the device is limited to m <= %d observations", (int)AEM_MP_MAX);
  std::vector<double> Lc;
  if (!cholesky_host(cov, m, Lc)) return fail(TDA_ERR_NUMERIC, "noise covariance is not positive definite");
  const int MP = m <= 64 ? 64 : 128;
  lv.em_ld = MP;
  lv.cov_h.assign(cov, cov + (size_t)m * m);
  lv.ytil_h.assign(MP, 0.0);
  lv.data_h.assign(MP, 0.0);
  for (int i = 0; i < m; ++i) lv.ytil_h[i] = lv.data_h[i] = data[i];
  std::vector<double> c64((size_t)MP * MP, 0.0);
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) c64[(size_t)i * MP + j] = cov[(size_t)i * m + j];
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
      DISPATCH_DPAD(e->DP, hipLaunchKernelGGL((k_linear_outputs<DPAD>), dim3((unsigned)(e->NP / 16)), dim3(256), 0, e->stream, (long long)e->N,
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
  DISPATCH_DPAD(e->DP, hipLaunchKernelGGL((k_linear_outputs<DPAD>), dim3((unsigned)(e->NP / 16)), dim3(256), 0, e->stream, (long long)e->N,
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
  DISPATCH_DPAD(e->DP, launch_steps<DPAD>(a, e->NP / 16, lds, e->stream));
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
const char* tda_version(void) { return "tinyda_amd 0.3 (gfx950)"; }

int tda_engine_create(const tda_config* cfg, tda_engine** out) {
  if (!cfg || !out) return fail(TDA_ERR_INVALID, "null argument");
  if (cfg->struct_size != sizeof(tda_config)) return fail(TDA_ERR_INVALID, "tda_config.struct_size mismatch");
  if (cfg->dim < 1 || cfg->dim > 64)
    return fail(TDA_ERR_UNSUPPORTED, "dim=%d outside the device engine's range 1..64", cfg->dim);
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

int tda_engine_set_prior(tda_engine* e, const double* mean, const double* cov) {
  if (!e || !mean || !cov) return fail(TDA_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int d = e->d, DP = e->DP;
  e->prior_mean_h.assign(mean, mean + d);
  e->prior_cov_h.assign(cov, cov + (size_t)d * d);
  if (!cholesky_host(cov, d, e->prior_L_h))
    return fail(TDA_ERR_NUMERIC, "prior covariance is not positive definite");
  double logdet = 0.0;
  for (int j = 0; j < d; ++j) logdet += 2.0 * std::log(e->prior_L_h[(size_t)j * d + j]);
  bool diag = true;
  for (int i = 0; i < d && diag; ++i)
    for (int j = 0; j < d; ++j)
      if (i != j && cov[(size_t)i * d + j] != 0.0) {
        diag = false;
        break;
      }
  std::vector<double> mh(DP, 0.0), ph(DP, 0.0);
  for (int j = 0; j < d; ++j) mh[j] = mean[j];
  int rc;
  if ((rc = e->prior_mean.upload(mh))) return rc;
  if (diag) {
    e->prior_kind = PRIOR_DIAG;
    logdet = 0.0;
    bool standard = true;
    for (int j = 0; j < d; ++j) {
      ph[j] = 1.0 / cov[(size_t)j * d + j];
      logdet += std::log(cov[(size_t)j * d + j]);
      standard = standard && cov[(size_t)j * d + j] == 1.0 && mean[j] == 0.0;
    }
    e->prior_is_standard = standard;
    e->prior_ncb = 0;
  } else {
    e->prior_kind = PRIOR_DENSE;
    std::vector<double> W, Wpk;
    tri_inverse_host(e->prior_L_h, d, W);
    pack_fragments(W.data(), d, d, DP, Wpk, e->prior_ncb);
    std::vector<double> wmu((size_t)e->prior_ncb * 16, 0.0);
    for (int i = 0; i < d; ++i) {
      double s = 0.0;
      for (int j = 0; j < d; ++j) s += W[(size_t)i * d + j] * mean[j];
      wmu[i] = s;
    }
    if ((rc = e->prior_Wpk.upload(Wpk))) return rc;
    if ((rc = e->prior_wmu.upload(wmu))) return rc;
  }
  if ((rc = e->prior_pinv.upload(ph))) return rc;
  e->prior_logconst = d * std::log(2.0 * M_PI) + logdet;
  e->prior_bounded = false;
  e->prior_set = true;
  return TDA_OK;
}

int tda_engine_set_prior_joint(tda_engine* e, const int32_t* kind, const double* loc, const double* scale) {
  if (!e || !kind || !loc || !scale) return fail(TDA_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int d = e->d, DP = e->DP;
  std::vector<double> mh(DP, 0.0), ph(DP, 0.0), lo(DP, -INFINITY), hi(DP, INFINITY);
  e->prior_mean_h.assign(d, 0.0);
  e->prior_cov_h.assign((size_t)d * d, 0.0);
  e->prior_L_h.assign((size_t)d * d, 0.0);
  double logconst = 0.0;
  bool bounded = false;
  for (int j = 0; j < d; ++j) {
    if (!(scale[j] > 0.0)) return fail(TDA_ERR_NUMERIC, "prior component %d: scale must be positive", j);
    if (kind[j] == 0) {  // scipy.stats.norm(loc, scale)
      mh[j] = loc[j];
      ph[j] = 1.0 / (scale[j] * scale[j]);
      logconst += std::log(2.0 * M_PI) + 2.0 * std::log(scale[j]);
      e->prior_mean_h[j] = loc[j];
      e->prior_cov_h[(size_t)j * d + j] = scale[j] * scale[j];
      e->prior_L_h[(size_t)j * d + j] = scale[j];
    } else if (kind[j] == 1) {  // scipy.stats.uniform(loc, scale): density 1/scale on [loc, loc + scale]
      lo[j] = loc[j];
      hi[j] = loc[j] + scale[j];
      logconst += 2.0 * std::log(scale[j]);
      bounded = true;
      e->prior_mean_h[j] = loc[j] + 0.5 * scale[j];
      e->prior_cov_h[(size_t)j * d + j] = scale[j] * scale[j] / 12.0;
      e->prior_L_h[(size_t)j * d + j] = scale[j] / std::sqrt(12.0);
    } else {
      return fail(TDA_ERR_UNSUPPORTED, "prior component %d: kind %d (0 = normal, 1 = uniform)", j, (int)kind[j]);
    }
  }
  int rc;
  if ((rc = e->prior_mean.upload(mh)) || (rc = e->prior_pinv.upload(ph)) || (rc = e->prior_lo.upload(lo)) || (rc = e->prior_hi.upload(hi)))
    return rc;
  e->prior_kind = PRIOR_DIAG;
  e->prior_is_standard = false;
  e->prior_ncb = 0;
  e->prior_logconst = logconst;
  e->prior_bounded = bounded;
  e->prior_set = true;
  return TDA_OK;
}

int tda_engine_set_level(tda_engine* e, int level, int m, const double* A, const double* b, const double* data,
                         int noise_kind, const double* noise) {
  if (!e || !A || !data || !noise) return fail(TDA_ERR_INVALID, "null argument");
  if (level < 0 || level >= (int)e->levels.size()) return fail(TDA_ERR_INVALID, "level %d out of range", level);
  if (m < 1) return fail(TDA_ERR_INVALID, "m must be >= 1");
  HIP_TRY(hipSetDevice(e->cfg.device));
  Level& lv = e->levels[level];
  if (noise_kind < TDA_NOISE_ISO || noise_kind > TDA_NOISE_ADAPTIVE) return fail(TDA_ERR_INVALID, "noise_kind %d", noise_kind);
  if (noise_kind == TDA_NOISE_ADAPTIVE && m > AEM_MP_MAX)
    return fail(TDA_ERR_UNSUPPORTED, "AdaptiveGaussianLogLike on the device is limited to m <= %d observations (per-chain m x m state)", (int)AEM_MP_MAX);
  const int AEM_MP = m <= 64 ? 64 : 128;  // row stride of this level's error-model copies
  if (noise_kind == TDA_NOISE_DENSE && (e->nlev != 1))
    return fail(TDA_ERR_UNSUPPORTED, "dense noise covariance is lowered for single-level chains only so far");
  std::vector<double> Apk;
  pack_fragments(A, m, e->d, e->DP, Apk, lv.ncb);
  lv.m = m;
  lv.m_pad = lv.ncb * 16;
  lv.noise_kind = noise_kind;
  std::vector<double> yt(lv.m_pad, 0.0), w;
  for (int i = 0; i < m; ++i) yt[i] = data[i] - (b ? b[i] : 0.0);
  // row-major copies (error-model kernel)
  lv.A_h.assign(A, A + (size_t)m * e->d);
  lv.ytil_h.assign(AEM_MP > m ? AEM_MP : m, 0.0);
  lv.data_h.assign(AEM_MP > m ? AEM_MP : m, 0.0);
  for (int i = 0; i < m; ++i) {
    lv.ytil_h[i] = data[i] - (b ? b[i] : 0.0);
    lv.data_h[i] = data[i];
  }
  lv.cov_h.clear();
  std::vector<double> Ppk;
  if (noise_kind == TDA_NOISE_ADAPTIVE) {
    std::vector<double> Lc;
    if (!cholesky_host(noise, m, Lc)) return fail(TDA_ERR_NUMERIC, "noise covariance is not positive definite");
    lv.cov_h.assign(noise, noise + (size_t)m * m);
    lv.var = 1.0;
  } else if (noise_kind == TDA_NOISE_DENSE) {
    // Sigma^-1 (distributions.py:280) through the Cholesky factor: P = L^-T L^-1, symmetric by construction
    std::vector<double> Lc, W;
    if (!cholesky_host(noise, m, Lc)) return fail(TDA_ERR_NUMERIC, "noise covariance is not positive definite");
    tri_inverse_host(Lc, m, W);
    std::vector<double> P((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = 0.0;
        for (int k = i; k < m; ++k) s += W[(size_t)k * m + i] * W[(size_t)k * m + j];
        P[(size_t)i * m + j] = P[(size_t)j * m + i] = s;
      }
    int ncb2 = 0;
    pack_fragments(P.data(), m, m, lv.m_pad, Ppk, ncb2);
    lv.Pinv_h = P;
    lv.var = 1.0;
  } else if (noise_kind == TDA_NOISE_ISO) {
    if (!(noise[0] > 0.0)) return fail(TDA_ERR_NUMERIC, "noise variance must be positive");
    lv.var = noise[0];
  } else {
    w.assign(lv.m_pad, 0.0);
    for (int i = 0; i < m; ++i) {
      if (!(noise[i] > 0.0)) return fail(TDA_ERR_NUMERIC, "noise variance must be positive");
      w[i] = 1.0 / noise[i];
    }
    lv.w_h.assign(w.begin(), w.begin() + m);
    lv.var = 1.0;
  }
  const size_t lds = ((size_t)16 * (e->DP + 2) + 256 + 2 * e->DP + (size_t)lv.m_pad * 2 + 64 +
                      (noise_kind == TDA_NOISE_DENSE ? (size_t)16 * (lv.m_pad + 2) : 0)) * sizeof(double);
  if (lds > 158 * 1024) return fail(TDA_ERR_UNSUPPORTED, "m=%d observations exceed the LDS staging budget", m);
  int rc;
  lv.em_ld = m <= AEM_MP_MAX ? AEM_MP : 0;
  if (m <= AEM_MP_MAX) {
    std::vector<double> y64(lv.ytil_h.begin(), lv.ytil_h.begin() + AEM_MP), c64;
    {  // column-major [d][AEM_MP] for k_aem_action: lane = observation reads consecutive addresses
      std::vector<double> acm((size_t)e->d * AEM_MP, 0.0);
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < e->d; ++j) acm[(size_t)j * AEM_MP + i] = lv.A_h[(size_t)i * e->d + j];
      if ((rc = lv.A_rm.upload(acm))) return rc;
    }
    if ((rc = lv.ytil64.upload(y64))) return rc;
    std::vector<double> d64(lv.data_h.begin(), lv.data_h.begin() + AEM_MP);
    if ((rc = lv.data64.upload(d64))) return rc;
    if (!lv.cov_h.empty()) {
      c64.assign((size_t)AEM_MP * AEM_MP, 0.0);
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) c64[(size_t)i * AEM_MP + j] = lv.cov_h[(size_t)i * m + j];
      if ((rc = lv.cov64.upload(c64))) return rc;
    }
  }
  if (e->nlev > 1 && noise_kind != TDA_NOISE_DENSE) {
    // a linear level may sit in a hierarchy with callback / source-defined levels, which is sequenced by the host: the
    // level kernels of that mode take model outputs, residual data and weights from these buffers
    std::vector<double> yd(lv.data_h.begin(), lv.data_h.begin() + m), bd(m, 0.0);
    for (int i = 0; i < m; ++i) bd[i] = b ? b[i] : 0.0;
    if ((rc = lv.A_dev.upload(lv.A_h))) return rc;
    if ((rc = lv.b_dev.upload(bd))) return rc;
    if ((rc = lv.udata.upload(yd))) return rc;
    if (noise_kind == TDA_NOISE_DIAG) {
      std::vector<double> wd(w.begin(), w.begin() + m);
      if ((rc = lv.uw.upload(wd))) return rc;
    }
    if ((rc = lv.cb_prop.alloc((size_t)e->N * e->d))) return rc;
    if ((rc = lv.cb_F.alloc((size_t)e->N * m))) return rc;
  }
  if ((rc = lv.Ppk.upload(Ppk))) return rc;
  if ((rc = lv.Apk.upload(Apk))) return rc;
  if ((rc = lv.ytil.upload(yt))) return rc;
  if ((rc = lv.w.upload(w))) return rc;
  lv.set = true;
  return TDA_OK;
}

int tda_engine_set_proposal(tda_engine* e, const tda_proposal_params* p) {
  if (!e || !p) return fail(TDA_ERR_INVALID, "null argument");
  if (p->struct_size != sizeof(tda_proposal_params)) return fail(TDA_ERR_INVALID, "tda_proposal_params.struct_size mismatch");
  const bool indep = p->kind == TDA_PROP_INDEPENDENCE, owcn = p->kind == TDA_PROP_OWCN, mala = p->kind == TDA_PROP_MALA;
  if ((p->kind < TDA_PROP_GRW || p->kind > TDA_PROP_AM) && !indep && !owcn && !mala) return fail(TDA_ERR_UNSUPPORTED, "proposal kind %d", p->kind);
  if (mala && e->nlev != 1) return fail(TDA_ERR_UNSUPPORTED, "MALA is lowered for single-level chains only");
  if (mala && !(p->scaling > 0.0)) return fail(TDA_ERR_INVALID, "MALA: scaling must be positive");
  if (owcn && e->nlev != 1) return fail(TDA_ERR_UNSUPPORTED, "the operator-weighted pCN proposal is lowered for single-level chains only");
  if (owcn && p->adaptive) return fail(TDA_ERR_UNSUPPORTED, "operator-weighted pCN: adaptive scaling (per-chain operators) is not lowered");
  if ((p->kind == TDA_PROP_GRW || p->kind == TDA_PROP_AM || indep) && !p->C) return fail(TDA_ERR_INVALID, "proposal covariance missing");
  if (indep && !p->q_mean) return fail(TDA_ERR_INVALID, "independence sampler: q_mean missing");
  if (indep && e->nlev != 1) return fail(TDA_ERR_UNSUPPORTED, "the independence sampler is lowered for single-level chains only");
  if (p->period < 1) return fail(TDA_ERR_INVALID, "period must be >= 1");
  e->pp = *p;
  e->pp.q_mean = nullptr;
  if (owcn) {  // the scaling is inside the operators (proposal.py:579-580)
    e->pp.scaling = 1.0;
    e->ow_state_h.clear();
    e->ow_noise_h.clear();
  }
  if (indep) {  // never adapts (proposal.py:107-111); scaling plays no role
    e->pp.adaptive = 0;
    e->pp.scaling = 1.0;
    e->q_mean_h.assign(p->q_mean, p->q_mean + e->d);
  }
  if (p->C) e->prop_C_h.assign(p->C, p->C + (size_t)e->d * e->d);
  e->pp.C = nullptr;
  e->am_sd = p->sd > 0.0 ? p->sd : std::min(1.0, 2.4 * 2.4 / e->d);
  e->prop_set = true;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_set_proposal_operators(tda_engine* e, const double* state_operator, const double* noise_operator) {
  if (!e || !state_operator || !noise_operator) return fail(TDA_ERR_INVALID, "null argument");
  if (!e->prop_set || e->pp.kind != TDA_PROP_OWCN) return fail(TDA_ERR_STATE, "set_proposal with kind TDA_PROP_OWCN must precede set_proposal_operators");
  const size_t dd = (size_t)e->d * e->d;
  for (size_t i = 0; i < dd; ++i)
    if (!std::isfinite(state_operator[i]) || !std::isfinite(noise_operator[i])) return fail(TDA_ERR_NUMERIC, "proposal operators must be finite");
  e->ow_state_h.assign(state_operator, state_operator + dd);
  e->ow_noise_h.assign(noise_operator, noise_operator + dd);
  e->inited = false;
  return TDA_OK;
}

int tda_engine_set_level_rosenbrock(tda_engine* e, int level, double a, double b, double data, double noise_var) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  if (level < 0 || level >= (int)e->levels.size()) return fail(TDA_ERR_INVALID, "level %d out of range", level);
  if (e->d < 2) return fail(TDA_ERR_INVALID, "the Rosenbrock chain needs dim >= 2");
  if (!(noise_var > 0.0)) return fail(TDA_ERR_NUMERIC, "noise variance must be positive");
  Level& lv = e->levels[level];
  lv.model = MODEL_ROSENBROCK;
  lv.ros_a = a;
  lv.ros_b = b;
  lv.ros_data = data;
  lv.m = 1;
  lv.m_pad = 0;
  lv.ncb = 0;
  lv.noise_kind = TDA_NOISE_ISO;
  lv.var = noise_var;
  lv.set = true;
  return TDA_OK;
}

int tda_engine_set_level_source(tda_engine* e, int level, const char* source, int32_t m, const double* data, int32_t noise_kind,
                                const double* noise) {
  if (!e || !source || !data || !noise) return fail(TDA_ERR_INVALID, "null argument");
  if (level < 0 || level >= (int)e->levels.size()) return fail(TDA_ERR_INVALID, "level %d out of range", level);
  if (m < 1) return fail(TDA_ERR_INVALID, "m must be >= 1");
  if (noise_kind != TDA_NOISE_ISO && noise_kind != TDA_NOISE_DIAG && noise_kind != TDA_NOISE_DENSE && !(noise_kind == TDA_NOISE_ADAPTIVE && e->nlev > 1))
    return fail(TDA_ERR_UNSUPPORTED, "noise kind %d (AdaptiveGaussianLogLike only below the finest level of a hierarchy)", noise_kind);
  HIP_TRY(hipSetDevice(e->cfg.device));
  Level& lv = e->levels[level];
  if (lv.umod) {
    (void)hipModuleUnload(lv.umod);
    lv.umod = nullptr;
    lv.ufn = nullptr;
    lv.ufn_eval = nullptr;
    lv.ufn_level = nullptr;
  }
  int rc = compile_user_model(source, &lv.umod, &lv.ufn, &lv.ufn_eval, &lv.ufn_level);
  if (rc) return rc;
  std::vector<double> y(data, data + m), w;
  if (noise_kind == TDA_NOISE_ADAPTIVE) {
    if ((rc = ext_level_adaptive(e, lv, m, data, noise))) return rc;
  } else if (noise_kind == TDA_NOISE_DENSE) {
    if ((rc = ext_level_dense(e, lv, m, noise))) return rc;
  } else if (noise_kind == TDA_NOISE_ISO) {
    if (!(noise[0] > 0.0)) return fail(TDA_ERR_NUMERIC, "noise variance must be positive");
    lv.var = noise[0];
  } else {
    w.resize(m);
    for (int i = 0; i < m; ++i) {
      if (!(noise[i] > 0.0)) return fail(TDA_ERR_NUMERIC, "noise variance must be positive");
      w[i] = 1.0 / noise[i];
    }
    lv.var = 1.0;
    if ((rc = lv.uw.upload(w))) return rc;
  }
  if ((rc = lv.udata.upload(y))) return rc;
  if (noise_kind != TDA_NOISE_ADAPTIVE) {  // host / padded device copies of the data (error-model initialisation and kernels)
    const int mp = m > 128 ? m : 128;
    lv.ytil_h.assign(mp, 0.0);
    lv.data_h.assign(mp, 0.0);
    for (int i = 0; i < m; ++i) lv.ytil_h[i] = lv.data_h[i] = data[i];
    if ((rc = lv.data64.upload(lv.data_h))) return rc;
  }
  // hierarchies and DREAM(Z): proposals / outputs of a step pass through device buffers (tda_user_eval)
  if ((rc = lv.cb_prop.alloc((size_t)e->N * e->d))) return rc;
  if ((rc = lv.cb_F.alloc((size_t)e->N * m))) return rc;
  lv.model = MODEL_USER;
  lv.m = m;
  lv.m_pad = 16;  // (no MFMA staging; keeps the shared LDS-size arithmetic of the run loop valid)
  lv.ncb = 1;
  lv.noise_kind = noise_kind;
  lv.set = true;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_set_level_callback(tda_engine* e, int level, tda_forward_batch_fn fn, void* user, int32_t m, const double* data,
                                  int32_t noise_kind, const double* noise) {
  if (!e || !fn || !data || !noise) return fail(TDA_ERR_INVALID, "null argument");
  if (level < 0 || level >= (int)e->levels.size()) return fail(TDA_ERR_INVALID, "level %d out of range", level);
  if (m < 1) return fail(TDA_ERR_INVALID, "m must be >= 1");
  if (noise_kind != TDA_NOISE_ISO && noise_kind != TDA_NOISE_DIAG && noise_kind != TDA_NOISE_DENSE && !(noise_kind == TDA_NOISE_ADAPTIVE && e->nlev > 1))
    return fail(TDA_ERR_UNSUPPORTED, "noise kind %d (AdaptiveGaussianLogLike only below the finest level of a hierarchy)", noise_kind);
  HIP_TRY(hipSetDevice(e->cfg.device));
  Level& lv = e->levels[level];
  std::vector<double> y(data, data + m), w;
  int rc;
  if (noise_kind == TDA_NOISE_ADAPTIVE) {
    if ((rc = ext_level_adaptive(e, lv, m, data, noise))) return rc;
  } else if (noise_kind == TDA_NOISE_DENSE) {
    if ((rc = ext_level_dense(e, lv, m, noise))) return rc;
  } else if (noise_kind == TDA_NOISE_ISO) {
    if (!(noise[0] > 0.0)) return fail(TDA_ERR_NUMERIC, "noise variance must be positive");
    lv.var = noise[0];
  } else {
    w.resize(m);
    for (int i = 0; i < m; ++i) {
      if (!(noise[i] > 0.0)) return fail(TDA_ERR_NUMERIC, "noise variance must be positive");
      w[i] = 1.0 / noise[i];
    }
    lv.var = 1.0;
    if ((rc = lv.uw.upload(w))) return rc;
  }
  if ((rc = lv.udata.upload(y))) return rc;
  if (noise_kind != TDA_NOISE_ADAPTIVE) {  // host / padded device copies of the data (error-model initialisation and kernels)
    const int mp = m > 128 ? m : 128;
    lv.ytil_h.assign(mp, 0.0);
    lv.data_h.assign(mp, 0.0);
    for (int i = 0; i < m; ++i) lv.ytil_h[i] = lv.data_h[i] = data[i];
    if ((rc = lv.data64.upload(lv.data_h))) return rc;
  }
  lv.release_callback_buffers();
  HIP_TRY(hipHostMalloc((void**)&lv.cb_theta_h, (size_t)e->N * e->d * sizeof(double), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&lv.cb_F_h, (size_t)e->N * m * sizeof(double), hipHostMallocDefault));
  if ((rc = lv.cb_prop.alloc((size_t)e->N * e->d))) return rc;
  if ((rc = lv.cb_F.alloc((size_t)e->N * m))) return rc;
  lv.cb_fn = fn;
  lv.cb_user = user;
  lv.model = MODEL_CALLBACK;
  lv.m = m;
  lv.m_pad = 16;  // (no MFMA staging; keeps the shared LDS-size arithmetic of the run loop valid)
  lv.ncb = 1;
  lv.noise_kind = noise_kind;
  lv.set = true;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_set_proposal_dreamz(tda_engine* e, const tda_dreamz_params* p) {
  if (!e || !p) return fail(TDA_ERR_INVALID, "null argument");
  if (p->struct_size != sizeof(tda_dreamz_params)) return fail(TDA_ERR_INVALID, "tda_dreamz_params.struct_size mismatch");
  if (e->nlev != 1 && p->shared)
    return fail(TDA_ERR_UNSUPPORTED, "below a Delayed Acceptance / MLDA hierarchy DREAMZ keeps one archive per chain (DREAM's shared archive is single-level)");
  if (p->M0 < 2 * 1 + 1) return fail(TDA_ERR_INVALID, "M0 too small");
  if (p->delta < 1 || p->delta > MAX_DELTA) return fail(TDA_ERR_UNSUPPORTED, "delta=%d outside 1..%d", p->delta, (int)MAX_DELTA);
  if (p->nCR < 1 || p->nCR > MAX_NCR) return fail(TDA_ERR_UNSUPPORTED, "nCR=%d outside 1..%d", p->nCR, (int)MAX_NCR);
  if (p->period < 1) return fail(TDA_ERR_INVALID, "period must be >= 1");
  if (p->capacity < p->M0) return fail(TDA_ERR_INVALID, "archive capacity smaller than M0");
  e->dz = *p;
  e->is_dreamz = true;
  e->pp = tda_proposal_params{};
  e->pp.struct_size = sizeof(tda_proposal_params);
  e->pp.kind = TDA_PROP_DREAMZ;
  e->pp.scaling = 1.0;  // proposal.py:715
  e->pp.adaptive = p->adaptive;
  e->pp.period = p->period;
  e->pp.gamma = p->gamma;
  e->prop_set = true;
  e->arch_set = false;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_set_archive(tda_engine* e, const double* Z0) {
  if (!e || !e->is_dreamz) return fail(TDA_ERR_STATE, "set_proposal_dreamz first");
  if (!e->prior_set) return fail(TDA_ERR_STATE, "set_prior first");
  const int d = e->d;
  const int64_t rows = (e->dz.shared ? 1 : e->N) * (int64_t)e->dz.M0;
  e->Z0_h.resize((size_t)rows * d);
  e->arch_given = Z0 != nullptr;
  if (Z0) {
    if (is_device_ptr(Z0))
      HIP_TRY(hipMemcpy(e->Z0_h.data(), Z0, e->Z0_h.size() * sizeof(double), hipMemcpyDeviceToHost));
    else
      std::copy(Z0, Z0 + e->Z0_h.size(), e->Z0_h.begin());
  } else {  // prior.rvs(M0) (proposal.py:788) from RNG stream 2; per-chain archives keyed by the global chain id
    std::vector<double> z(d + 1);
    for (int64_t r = 0; r < rows; ++r) {
      const uint32_t key_chain = e->dz.shared ? 0xFFFFFFFFu : (uint32_t)(e->cfg.chain_offset + r / e->dz.M0);
      const uint32_t row = (uint32_t)(e->dz.shared ? r : r % e->dz.M0);
      for (int b = 0; b < (d + 1) / 2; ++b) normal_pair(e->cfg.seed, key_chain, 1u + row, STREAM_INIT, (uint32_t)b, z[2 * b], z[2 * b + 1]);
      for (int i = 0; i < d; ++i) {
        double s = 0.0;
        for (int k = 0; k <= i; ++k) s = std::fma(e->prior_L_h[(size_t)i * d + k], z[k], s);
        e->Z0_h[(size_t)r * d + i] = e->prior_mean_h[i] + s;
      }
    }
  }
  e->arch_set = true;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_set_replay_dreamz(tda_engine* e, const int32_t* r, const int32_t* mcr, const double* sub_u,
                                 const int32_t* forced, const double* e_u, const double* eps_n, const double* u,
                                 int64_t n_steps) {
  if (!e || !e->is_dreamz) return fail(TDA_ERR_STATE, "set_proposal_dreamz first");
  HIP_TRY(hipSetDevice(e->cfg.device));
  e->rp_steps = e->rp_pos = 0;
  if (!r || n_steps <= 0) return TDA_OK;
  if (!mcr || !sub_u || !forced || !e_u || !eps_n || !u) return fail(TDA_ERR_INVALID, "all DREAMZ replay arrays are required");
  const size_t N = e->N, d = e->d, T = n_steps;
  int rc;
  std::vector<int32_t> vi;
  std::vector<double> vd;
#define UP_I(buf, src, count)                     \
  vi.assign(src, src + (count));                  \
  if ((rc = e->buf.upload(vi))) return rc;
#define UP_D(buf, src, count)                     \
  vd.assign(src, src + (count));                  \
  if ((rc = e->buf.upload(vd))) return rc;
  UP_I(rp_r, r, T * N * e->dz.delta * 2)
  UP_I(rp_mcr, mcr, T * N)
  UP_I(rp_forced, forced, T * N)
  UP_D(rp_sub, sub_u, T * N * d)
  UP_D(rp_e, e_u, T * N * d)
  UP_D(rp_eps, eps_n, T * N * d)
  UP_D(rp_u, u, T * N)
#undef UP_I
#undef UP_D
  e->rp_steps = n_steps;
  return TDA_OK;
}

int tda_engine_get_dreamz_state(tda_engine* e, double* pCR, int64_t* archive_rows) {
  if (!e || !e->inited || !e->is_dreamz) return fail(TDA_ERR_STATE, "no DREAMZ engine initialised");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (pCR) {
    std::vector<double> hbuf((size_t)e->NP * MAX_NCR);
    HIP_TRY(hipMemcpy(hbuf.data(), e->dz_pCR.p, hbuf.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < e->N; ++c)
      for (int k = 0; k < e->dz.nCR; ++k) pCR[c * e->dz.nCR + k] = hbuf[(size_t)c * MAX_NCR + k];
  }
  if (archive_rows) *archive_rows = e->arch_rows;
  return TDA_OK;
}

int tda_engine_set_archive_auto_append(tda_engine* e, int on) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  e->auto_append = on != 0;
  return TDA_OK;
}

static int dreamz_sums_catchup(tda_engine* e, int64_t row0, int64_t nrows, bool boundary, bool scale, double gamma_pow,
                               const double* theta_now = nullptr, const uint8_t* ring = nullptr, int ring_P = 0, int64_t ring_hi = 0);

int tda_engine_archive_take(tda_engine* e, double* rows, int64_t* n_steps) {
  if (!e || !e->inited || !e->is_dreamz || !e->dz.shared) return fail(TDA_ERR_STATE, "no shared-archive engine initialised");
  if (e->dist_ranks) return fail(TDA_ERR_STATE, "the archive is distributed: its rows stay where they are written (tda_engine_archive_publish)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (n_steps) *n_steps = e->pending_steps;
  if (rows && e->pending_steps) {  // [steps][N][d] without padding
    const hipMemcpyKind kind = is_device_ptr(rows) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if (e->N == e->NP) {  // no padded chains: the pending rows are one [steps * N][DP] matrix
      HIP_TRY(hipMemcpy2DAsync(rows, e->d * sizeof(double), e->blk_hist.p, e->DP * sizeof(double), e->d * sizeof(double),
                               e->pending_steps * e->N, kind, e->stream));
    } else {
      for (int64_t s = 0; s < e->pending_steps; ++s)
        HIP_TRY(hipMemcpy2DAsync(rows + (size_t)s * e->N * e->d, e->d * sizeof(double), e->blk_hist.p + (size_t)s * e->NP * e->DP,
                                 e->DP * sizeof(double), e->d * sizeof(double), e->N, kind, e->stream));
    }
    if (kind == hipMemcpyDeviceToHost) HIP_TRY(hipStreamSynchronize(e->stream));  // device buffers: stream ordered, no host wait
  }
  if (rows) e->pending_steps = 0;  // a NULL buffer only asks how many steps are pending
  return TDA_OK;
}

int tda_engine_archive_append(tda_engine* e, const double* rows, int64_t n_rows) {
  if (!e || !e->inited || !e->is_dreamz || !e->dz.shared) return fail(TDA_ERR_STATE, "no shared-archive engine initialised");
  if (e->dist_ranks) return fail(TDA_ERR_STATE, "the archive is distributed: its rows stay where they are written (tda_engine_archive_publish)");
  if (n_rows <= 0) return TDA_OK;
  if (e->arch_rows + n_rows > e->arch_cap) return fail(TDA_ERR_INVALID, "shared archive capacity (%lld rows) exceeded", (long long)e->arch_cap);
  HIP_TRY(hipSetDevice(e->cfg.device));
  const hipMemcpyKind kind = is_device_ptr(rows) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIP_TRY(hipMemcpy2DAsync(e->arch.p + (size_t)e->arch_rows * e->DP, e->DP * sizeof(double), rows, e->d * sizeof(double),
                           e->d * sizeof(double), n_rows, kind, e->stream));
  int rc = dreamz_sums_catchup(e, e->arch_rows, n_rows, false, false, 1.0);
  if (rc) return rc;
  e->arch_rows += n_rows;
  if (kind == hipMemcpyHostToDevice) HIP_TRY(hipStreamSynchronize(e->stream));
  return TDA_OK;
}

// ---- distributed shared archive: every rank keeps the rows of its own chains, proposals read peers' rows in place ----
int tda_engine_archive_ipc_handle(tda_engine* e, void* handle) {
  if (!e || !handle) return fail(TDA_ERR_INVALID, "null argument");
  if (!e->inited || !e->is_dreamz || !e->dz.shared) return fail(TDA_ERR_STATE, "no shared-archive engine initialised");
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the header promises 64-byte handles");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipIpcMemHandle_t h;
  HIP_TRY(hipIpcGetMemHandle(&h, e->arch.p));
  memcpy(handle, &h, sizeof h);
  return TDA_OK;
}

int tda_engine_archive_pointer(tda_engine* e, void** pointer) {
  if (!e || !pointer) return fail(TDA_ERR_INVALID, "null argument");
  if (!e->inited || !e->is_dreamz || !e->dz.shared) return fail(TDA_ERR_STATE, "no shared-archive engine initialised");
  *pointer = e->arch.p;
  return TDA_OK;
}

int tda_engine_set_archive_peers(tda_engine* e, int n_ranks, int my_rank, const void* handles, const double* const* pointers) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  if (!e->inited || !e->is_dreamz || !e->dz.shared) return fail(TDA_ERR_STATE, "no shared-archive engine initialised");
  if (n_ranks < 1 || n_ranks > MAX_PEERS || my_rank < 0 || my_rank >= n_ranks) return fail(TDA_ERR_INVALID, "1 <= n_ranks <= %d, 0 <= my_rank < n_ranks", MAX_PEERS);
  if (n_ranks > 1 && !handles && !pointers) return fail(TDA_ERR_INVALID, "peer segments need IPC handles or device pointers");
  if (e->N != e->NP) return fail(TDA_ERR_UNSUPPORTED, "the distributed archive needs a chain count that is a multiple of 16");
  if (e->arch_rows != e->dz.M0 || e->pending_steps || e->dist_ranks) return fail(TDA_ERR_STATE, "set the peers once, right after init");
  const Level& lv = e->levels[0];
  if (lv.model == MODEL_CALLBACK || lv.model == MODEL_USER) return fail(TDA_ERR_UNSUPPORTED, "the distributed archive serves the engine's own models");
  if (e->nlev > 1) return fail(TDA_ERR_UNSUPPORTED, "the distributed archive serves single-level DREAM runs (below a hierarchy the archive is replicated)");
  HIP_TRY(hipSetDevice(e->cfg.device));
  for (int r = 0; r < n_ranks; ++r) {
    if (r == my_rank) {
      e->dist_seg[r] = e->arch.p;
    } else if (pointers) {
      if (!pointers[r]) return fail(TDA_ERR_INVALID, "null segment pointer of rank %d", r);
      e->dist_seg[r] = pointers[r];
    } else {
      hipIpcMemHandle_t h;
      memcpy(&h, (const char*)handles + (size_t)r * sizeof h, sizeof h);
      void* p = nullptr;
      HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
      e->dist_opened[r] = p;
      e->dist_seg[r] = (const double*)p;
    }
  }
  {
    int rc = e->dist_seg_dev.alloc((size_t)n_ranks);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(e->dist_seg_dev.p, e->dist_seg, (size_t)n_ranks * sizeof(const double*), hipMemcpyHostToDevice));
  }
  e->dist_ranks = n_ranks;
  e->dist_me = my_rank;
  e->dist_steps = e->dist_pending = e->dist_sum_steps = 0;
  e->dist_n_unpub = 0;
  e->dist_m0_summed = false;
  e->auto_append = false;
  return TDA_OK;
}

// column sums and sums of squares [2][d] (host) of this rank's rows that are visible but not yet in the archive sums: the ranks
// exchange these (2 d doubles) instead of the rows themselves
int tda_engine_archive_local_sums(tda_engine* e, double* sums) {
  if (!e || !sums) return fail(TDA_ERR_INVALID, "null argument");
  if (!e->dist_ranks) return fail(TDA_ERR_STATE, "no distributed archive");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int d = e->d, DP = e->DP;
  std::fill(sums, sums + 2 * d, 0.0);
  const int64_t from = e->dist_sum_steps, upto = e->dist_steps;  // rows before the block that waits for publication
  if (upto <= from) return TDA_OK;
  const int64_t nrows = (upto - from) * e->N, row0 = e->dz.M0 + from * e->N;
  const int64_t nb = (nrows + COLSUM_CHUNK - 1) / COLSUM_CHUNK;
  int rc;
  if (e->dz_partial.n < (size_t)nb * 2 * DP) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    if ((rc = e->dz_partial.alloc((size_t)nb * 2 * DP))) return rc;
  }
  DevBuf<double> tmp;
  if ((rc = tmp.alloc((size_t)2 * DP))) return rc;
  HIP_TRY(hipMemsetAsync(tmp.p, 0, 2 * DP * sizeof(double), e->stream));
  DISPATCH_DPAD(DP, launch_colsum<DPAD>(e->arch.p, row0, nrows, e->dz_partial.p, nb, e->stream));
  DISPATCH_DPAD(DP, launch_colsum_final<DPAD>(e->dz_partial.p, nb, tmp.p, tmp.p + DP, e->stream));
  std::vector<double> h((size_t)2 * DP);
  HIP_TRY(hipMemcpyAsync(h.data(), tmp.p, h.size() * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (int j = 0; j < d; ++j) {
    sums[j] = h[j];
    sums[d + j] = h[DP + j];
  }
  return TDA_OK;
}

// every rank has finished the block: its rows become visible to the proposals; sums_total ([2][d], host; may be NULL when the
// proposal does not adapt) = the ranks' local sums added up -- the same on every rank --, then the adaptation the block left pending
int tda_engine_archive_publish(tda_engine* e, const double* sums_total) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  if (!e->dist_ranks) return fail(TDA_ERR_STATE, "no distributed archive");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int d = e->d, DP = e->DP;
  if (e->dist_n_unpub == 0) return fail(TDA_ERR_STATE, "distributed archive: nothing to publish");
  const int64_t rows_before = e->dz.M0 + e->dist_steps * e->N * e->dist_ranks;  // the published archive (what the latest block proposed from)
  if (sums_total) {
    std::vector<double> zs(DP), zq(DP);
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(zs.data(), e->zsum.p, DP * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(zq.data(), e->zsq.p, DP * sizeof(double), hipMemcpyDeviceToHost));
    if (!e->dist_m0_summed) {  // the M0 shared initial rows: every rank holds them, nobody exchanges them
      std::vector<double> z0((size_t)e->dz.M0 * DP);
      HIP_TRY(hipMemcpy(z0.data(), e->arch.p, z0.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (int64_t r = 0; r < e->dz.M0; ++r)
        for (int j = 0; j < d; ++j) {
          zs[j] += z0[(size_t)r * DP + j];
          zq[j] += z0[(size_t)r * DP + j] * z0[(size_t)r * DP + j];
        }
      e->dist_m0_summed = true;
    }
    for (int j = 0; j < d; ++j) {
      zs[j] += sums_total[j];
      zq[j] += sums_total[d + j];
    }
    HIP_TRY(hipMemcpy(e->zsum.p, zs.data(), DP * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->zsq.p, zq.data(), DP * sizeof(double), hipMemcpyHostToDevice));
    e->dist_sum_steps = e->dist_steps;
    e->sums_rows = rows_before;
  }
  if (e->dist_adapt_pending) {
    if (e->dz.adaptive && !sums_total) return fail(TDA_ERR_INVALID, "an adaptation is pending: publish needs the archive sums");
    if (e->dist_adapt_rows != rows_before) return fail(TDA_ERR_STATE, "distributed archive: publish the adaptation before anything else");
    int rc = dreamz_sums_catchup(e, rows_before, 0, true, e->dz.adaptive != 0, e->dist_adapt_gamma);
    if (rc) return rc;
    e->k_adapt += 1;
    e->dist_adapt_pending = false;
  }
  const int64_t k = e->dist_unpub[0];  // the oldest unpublished block
  e->dist_unpub[0] = e->dist_unpub[1];
  e->dist_unpub[1] = 0;
  e->dist_n_unpub -= 1;
  e->dist_pending -= k;
  e->dist_steps += k;
  e->arch_rows = e->dz.M0 + e->dist_steps * e->N * e->dist_ranks;
  return TDA_OK;
}

int tda_engine_reduce_moments(tda_engine* e, const double* rows, int64_t n_rows, double* out) {
  if (!e || !rows || !out) return fail(TDA_ERR_INVALID, "null argument");
  if (!is_device_ptr(rows)) return fail(TDA_ERR_INVALID, "reduce_moments needs a device record buffer");
  if (n_rows < 1) return fail(TDA_ERR_INVALID, "n_rows must be >= 1");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int d = e->d, DP = e->DP;
  const int64_t nb = (n_rows + MOM_CHUNK - 1) / MOM_CHUNK;
  DevBuf<double> partial, outd;
  int rc;
  if ((rc = partial.alloc((size_t)nb * (DP + (size_t)DP * DP)))) return rc;
  const bool odev = is_device_ptr(out);
  const size_t nout = 1 + d + (size_t)d * d;
  if (!odev && (rc = outd.alloc(nout))) return rc;
  double* o = odev ? out : outd.p;
  DISPATCH_DPAD(DP, hipLaunchKernelGGL(k_moments_partial<DPAD>, dim3((unsigned)nb), dim3(64), 0, e->stream, rows, n_rows, d, partial.p));
  DISPATCH_DPAD(DP, hipLaunchKernelGGL(k_moments_final<DPAD>, dim3((unsigned)(d + 1)), dim3(64), 0, e->stream, partial.p, nb, n_rows, d, o));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (!odev) HIP_TRY(hipMemcpy(out, outd.p, nout * sizeof(double), hipMemcpyDeviceToHost));
  return TDA_OK;
}

int tda_engine_set_proposal_covariance(tda_engine* e, const double* C) {
  if (!e || !C) return fail(TDA_ERR_INVALID, "null argument");
  if (!e->inited || e->pp.kind != TDA_PROP_GRW || !e->L_shared)
    return fail(TDA_ERR_STATE, "set_proposal_covariance applies to an initialised GaussianRandomWalk engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int d = e->d, DP = e->DP;
  std::vector<double> Ch(C, C + (size_t)d * d), L;
  if (!cholesky_host(Ch.data(), d, L)) return fail(TDA_ERR_NUMERIC, "proposal covariance is not positive definite");
  std::vector<double> Lk((size_t)DP * DP, 0.0);
  for (int j = 0; j < d; ++j)
    for (int k = 0; k <= j; ++k) Lk[(size_t)k * DP + j] = L[(size_t)j * d + k];
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(e->Lk.p, Lk.data(), Lk.size() * sizeof(double), hipMemcpyHostToDevice));
  e->prop_C_h = Ch;
  e->L_identity = false;
  return TDA_OK;
}

namespace {
struct StateItem {
  void* p;
  size_t bytes;
  bool device;
};

// every piece of mutable state of an initialised engine, in a fixed order for its configuration
void enumerate_state(tda_engine* e, std::vector<StateItem>& v) {
  auto dev = [&](auto& buf) {
    if (buf.p && buf.n) v.push_back({(void*)buf.p, buf.n * sizeof(*buf.p), true});
  };
  auto host = [&](void* p, size_t b) { v.push_back({p, b, false}); };
  host(&e->t, sizeof e->t);
  host(&e->k_adapt, sizeof e->k_adapt);
  dev(e->theta);
  dev(e->lp);
  dev(e->ll);
  dev(e->scaling);
  dev(e->acc_count);
  dev(e->flags);
  dev(e->Lk);
  dev(e->am_mu);
  dev(e->am_sigma);
  dev(e->lq);
  dev(e->mala_grad);
  if (e->nlev > 1) {
    host(e->cnt, sizeof e->cnt);
    host(e->done, sizeof e->done);
    host(&e->ring_pos, sizeof e->ring_pos);
    host(e->aem_bt, sizeof e->aem_bt);
    dev(e->ml_theta);
    dev(e->ml_lp);
    dev(e->ml_ll);
    dev(e->ml_S);
    dev(e->ml_anyacc);
    dev(e->ml_ysnap);
    dev(e->ml_pick);
    dev(e->ml_ring);
    dev(e->ml_sid);
    for (int k = 0; k < MAXLEV; ++k) {
      dev(e->aem_bias[k]);
      dev(e->aem_covinv[k]);
      dev(e->aem_bmu[k]);
      dev(e->aem_bsig[k]);
      dev(e->aem_mdiff[k]);
      dev(e->ext_Fcur[k]);
      dev(e->aemd_F[k]);
      dev(e->aemd_bias[k]);
      dev(e->aemd_w[k]);
      dev(e->aemd_mu[k]);
      dev(e->aemd_var[k]);
      dev(e->aemd_md[k]);
    }
    dev(e->ext_Fst);
    dev(e->aemd_Fst);
  }
  if (e->is_dreamz) {
    host(&e->arch_rows, sizeof e->arch_rows);
    host(&e->sums_rows, sizeof e->sums_rows);
    host(&e->pending_steps, sizeof e->pending_steps);
    if (e->dist_ranks) {  // distributed archive: the publish protocol's position (the peer mappings belong to the process, not the state)
      host(&e->dist_steps, sizeof e->dist_steps);
      host(&e->dist_pending, sizeof e->dist_pending);
      host(e->dist_unpub, sizeof e->dist_unpub);
      host(&e->dist_n_unpub, sizeof e->dist_n_unpub);
      host(&e->dist_adapt_rows, sizeof e->dist_adapt_rows);
      host(&e->dist_sum_steps, sizeof e->dist_sum_steps);
      host(&e->dist_m0_summed, sizeof e->dist_m0_summed);
      host(&e->dist_adapt_pending, sizeof e->dist_adapt_pending);
      host(&e->dist_adapt_gamma, sizeof e->dist_adapt_gamma);
    }
    dev(e->arch);
    dev(e->zsum);
    dev(e->zsq);
    dev(e->dz_pCR);
    dev(e->dz_LCR);
    dev(e->dz_Delta);
    dev(e->dz_mcr_last);
    dev(e->theta_prev);
    dev(e->blk_hist);
  }
}
}  // namespace

int64_t tda_engine_state_size(tda_engine* e) {
  if (!e || !e->inited) return fail(TDA_ERR_STATE, "engine not initialised");
  std::vector<StateItem> v;
  enumerate_state(e, v);
  int64_t n = 16;  // header: magic + item count
  for (auto& it : v) n += 8 + (int64_t)it.bytes;
  return n;
}

int tda_engine_get_state(tda_engine* e, void* blob, int64_t bytes) {
  if (!e || !e->inited || !blob) return fail(TDA_ERR_STATE, "engine not initialised");
  if (bytes != tda_engine_state_size(e)) return fail(TDA_ERR_INVALID, "state blob must be exactly tda_engine_state_size() bytes");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::vector<StateItem> v;
  enumerate_state(e, v);
  char* o = (char*)blob;
  const uint64_t magic = 0x3141445454414453ull, cnt = v.size();
  memcpy(o, &magic, 8);
  memcpy(o + 8, &cnt, 8);
  o += 16;
  for (auto& it : v) {
    const uint64_t b = it.bytes;
    memcpy(o, &b, 8);
    o += 8;
    if (it.device) HIP_TRY(hipMemcpy(o, it.p, it.bytes, hipMemcpyDeviceToHost));
    else memcpy(o, it.p, it.bytes);
    o += it.bytes;
  }
  return TDA_OK;
}

int tda_engine_set_state(tda_engine* e, const void* blob, int64_t bytes) {
  if (!e || !e->inited || !blob) return fail(TDA_ERR_STATE, "engine not initialised (configure and init() it like the saved one first)");
  if (bytes != tda_engine_state_size(e)) return fail(TDA_ERR_INVALID, "state blob does not match this engine's configuration");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::vector<StateItem> v;
  enumerate_state(e, v);
  const char* o = (const char*)blob;
  uint64_t magic, cnt;
  memcpy(&magic, o, 8);
  memcpy(&cnt, o + 8, 8);
  if (magic != 0x3141445454414453ull || cnt != v.size()) return fail(TDA_ERR_INVALID, "not a state blob of this engine configuration");
  o += 16;
  for (auto& it : v) {
    uint64_t b;
    memcpy(&b, o, 8);
    if (b != it.bytes) return fail(TDA_ERR_INVALID, "state blob layout mismatch");
    o += 8;
    if (it.device) HIP_TRY(hipMemcpy(it.p, o, it.bytes, hipMemcpyHostToDevice));
    else memcpy(it.p, o, it.bytes);
    o += it.bytes;
  }
  if (e->L_identity) {  // derived from the factor at init(): the restored one may come from tda_engine_set_proposal_covariance
    const int d = e->d, DP = e->DP;
    std::vector<double> Lh((size_t)DP * DP);
    HIP_TRY(hipMemcpy(Lh.data(), e->Lk.p, Lh.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int k = 0; k < d && e->L_identity; ++k)
      for (int j = 0; j < d; ++j)
        if (Lh[(size_t)k * DP + j] != (j == k ? 1.0 : 0.0)) {
          e->L_identity = false;
          break;
        }
  }
  return TDA_OK;
}

int tda_engine_set_error_model(tda_engine* e, int kind) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  if (kind < TDA_AEM_NONE || kind > TDA_AEM_STATE_INDEPENDENT_DIAGONAL) return fail(TDA_ERR_INVALID, "Adaptive error model can only be state-dependent, state-independent or None.");
  if (kind != TDA_AEM_NONE && e->nlev < 2) return fail(TDA_ERR_STATE, "the error model needs at least two levels");
  if (kind == TDA_AEM_STATE_DEPENDENT && e->nlev != 2) return fail(TDA_ERR_UNSUPPORTED, "state-dependent error model is two-level (DA) only");
  e->aem = kind;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_get_error_model(tda_engine* e, int level, double* bias, double* cov_inverse) {
  if (!e || !e->inited || !e->aem) return fail(TDA_ERR_STATE, "no error model active");
  if (level < 0 || level >= e->nlev - 1) return fail(TDA_ERR_INVALID, "level %d has no adaptive likelihood", level);
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->aem == TDA_AEM_STATE_INDEPENDENT_DIAGONAL) {
    const int m = e->aem_m;
    if (bias) HIP_TRY(hipMemcpy(bias, e->aemd_bias[level].p, (size_t)e->N * m * sizeof(double), hipMemcpyDeviceToHost));
    if (cov_inverse) {
      std::vector<double> hw((size_t)e->N * m);
      HIP_TRY(hipMemcpy(hw.data(), e->aemd_w[level].p, hw.size() * sizeof(double), hipMemcpyDeviceToHost));
      std::fill(cov_inverse, cov_inverse + (size_t)e->N * m * m, 0.0);
      for (int64_t c = 0; c < e->N; ++c)
        for (int i = 0; i < m; ++i) cov_inverse[((size_t)c * m + i) * m + i] = hw[(size_t)c * m + i];
    }
    return TDA_OK;
  }
  const int m = e->aem_m, AEM_MP = e->aem_ld;
  if (bias) {
    std::vector<double> hb((size_t)e->NP * AEM_MP);
    HIP_TRY(hipMemcpy(hb.data(), e->aem_bias[level].p, hb.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < e->N; ++c)
      for (int i = 0; i < m; ++i) bias[c * m + i] = hb[(size_t)c * AEM_MP + i];
  }
  if (cov_inverse) {
    std::vector<double> hp((size_t)AEM_MP * AEM_MP);
    for (int64_t c = 0; c < e->N; ++c) {
      HIP_TRY(hipMemcpy(hp.data(), e->aem_covinv[level].p + (size_t)c * AEM_MP * AEM_MP, hp.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) cov_inverse[((size_t)c * m + i) * m + j] = hp[(size_t)i * AEM_MP + j];
    }
  }
  return TDA_OK;
}

int tda_engine_set_subchains(tda_engine* e, const int32_t* lengths, int randomize) {
  if (!e || !lengths) return fail(TDA_ERR_INVALID, "null argument");
  if (e->nlev < 2) return fail(TDA_ERR_STATE, "subchain lengths only apply to n_levels >= 2");
  for (int k = 0; k < e->nlev - 1; ++k) {
    if (lengths[k] < 1) return fail(TDA_ERR_INVALID, "subchain length must be >= 1");
    e->sl[k] = lengths[k];
  }
  if (randomize) {
    if (e->nlev != 2) return fail(TDA_ERR_UNSUPPORTED, "randomize_subchain_length is a two-level (DA) option");
    if (e->sl[0] == 1) return fail(TDA_ERR_INVALID, "Randomize subchain length requires a subchain_length > 1.");
  }
  e->randomize = randomize ? 1 : 0;
  e->sub_set = true;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_set_replay_level(tda_engine* e, int level, const double* u, int64_t n_steps) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (level == -1) {
    e->ridx_rep.release();
    e->ridx_rep_n = e->ridx_rep_pos = 0;
    if (!u || n_steps <= 0) return TDA_OK;
    int rc = e->ridx_rep.alloc((size_t)n_steps * e->N);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(e->ridx_rep.p, u, (size_t)n_steps * e->N * sizeof(double), is_device_ptr(u) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    e->ridx_rep_n = n_steps;
    return TDA_OK;
  }
  if (level < 1 || level >= e->nlev) return fail(TDA_ERR_INVALID, "level %d out of range for replay_level", level);
  e->u_rep_lv[level].release();
  e->u_rep_lv_n[level] = e->u_rep_lv_pos[level] = 0;
  if (!u || n_steps <= 0) return TDA_OK;
  int rc = e->u_rep_lv[level].alloc((size_t)n_steps * e->N);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(e->u_rep_lv[level].p, u, (size_t)n_steps * e->N * sizeof(double), is_device_ptr(u) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  e->u_rep_lv_n[level] = n_steps;
  return TDA_OK;
}

static int read_level_state(tda_engine* e, const double* th_dev, const double* lp_dev, const double* ll_dev, double* theta,
                            double* stats) {
  const int64_t N = e->N, NP = e->NP;
  if (theta) {
    std::vector<double> hh((size_t)NP * e->DP), o((size_t)N * e->d);
    HIP_TRY(hipMemcpy(hh.data(), th_dev, hh.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < N; ++c)
      for (int j = 0; j < e->d; ++j) o[(size_t)c * e->d + j] = hh[(size_t)c * e->DP + j];
    HIP_TRY(hipMemcpy(theta, o.data(), o.size() * sizeof(double), is_device_ptr(theta) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
  }
  if (stats) {
    std::vector<double> a(NP), b(NP), o((size_t)N * 3);
    HIP_TRY(hipMemcpy(a.data(), lp_dev, NP * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b.data(), ll_dev, NP * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < N; ++c) {
      o[c * 3] = a[c];
      o[c * 3 + 1] = b[c];
      o[c * 3 + 2] = a[c] + b[c];
    }
    HIP_TRY(hipMemcpy(stats, o.data(), o.size() * sizeof(double), is_device_ptr(stats) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
  }
  return TDA_OK;
}

int tda_engine_get_level_state(tda_engine* e, int level, double* theta, double* stats) {
  if (!e || !e->inited) return fail(TDA_ERR_STATE, "engine not initialised");
  if (level < 0 || level >= e->nlev) return fail(TDA_ERR_INVALID, "level %d out of range", level);
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->nlev == 1) return read_level_state(e, e->theta.p, e->lp.p, e->ll.p, theta, stats);
  return read_level_state(e, e->ml_theta.p + (size_t)level * e->NP * e->DP, e->ml_lp.p + (size_t)level * e->NP,
                          e->ml_ll.p + (size_t)level * e->NP, theta, stats);
}


// ---- diagonal error model (tda_kernels_aemd.h): argument block shared by its three kernels ----
static void fill_aemd_args(tda_engine* e, AemdArgs& a) {
  const int nl = e->nlev;
  a.N = e->N;
  a.NP = e->NP;
  a.chain_offset = e->cfg.chain_offset;
  a.d = e->d;
  a.DP = e->DP;
  a.m = e->aem_m;
  a.nlev = nl;
  a.is_da = nl == 2;
  a.prop_kind = e->pp.kind;
  a.seed = e->cfg.seed;
  for (int k = 0; k < nl; ++k) {
    a.data[k] = e->aemd_data[k].p;
    a.sig2[k] = e->aemd_sig2[k].p;
    a.Fcur[k] = e->aemd_F[k].p;
    a.bias[k] = e->aemd_bias[k].p;
    a.w[k] = e->aemd_w[k].p;
    a.mu[k] = e->aemd_mu[k].p;
    a.var[k] = e->aemd_var[k].p;
    a.md[k] = e->aemd_md[k].p;
  }
  a.wfin = e->aemd_wfin.p;
  a.var_finest = e->levels[nl - 1].var;
  a.Fst = e->aemd_Fst.p;
  a.theta = e->ml_theta.p;
  a.lp = e->ml_lp.p;
  a.ll = e->ml_ll.p;
  a.Sst = e->ml_S.p;
  a.anyacc = e->ml_anyacc.p;
  a.sid = reinterpret_cast<long long*>(e->ml_sid.p);
  a.pr_mean = e->prior_mean.p;
  a.pr_pinv = e->prior_pinv.p;
  a.pr_lo = e->prior_bounded ? e->prior_lo.p : nullptr;
  a.pr_hi = e->prior_bounded ? e->prior_hi.p : nullptr;
  a.logconst = e->prior_logconst;
}

// set-up at theta0: model outputs of every level through the path the steps use, then k_aemd_init
static int init_error_model_diagonal(tda_engine* e) {
  const int nl = e->nlev, d = e->d;
  const int64_t N = e->N;
  if (e->randomize) return fail(TDA_ERR_UNSUPPORTED, "randomize_subchain_length together with an error model is not lowered");
  if (e->prior_kind == PRIOR_DENSE) return fail(TDA_ERR_UNSUPPORTED, "the diagonal error model needs a diagonal prior covariance");
  if (e->is_dreamz) return fail(TDA_ERR_UNSUPPORTED, "DREAMZ below a hierarchy with an adaptive error model is not lowered");
  const int m = e->levels[0].m;
  for (int k = 0; k < nl; ++k) {
    const Level& lv = e->levels[k];
    if (lv.m != m) return fail(TDA_ERR_INVALID, "error model: all levels must share the output dimension");
    if (lv.noise_kind != TDA_NOISE_ISO && lv.noise_kind != TDA_NOISE_DIAG)
      return fail(TDA_ERR_INVALID, "diagonal error model: level %d needs isotropic or diagonal noise (its variances are Sigma_e)", k);
    if (lv.model == MODEL_LINEAR && !lv.A_dev.p) return fail(TDA_ERR_STATE, "diagonal error model: linear level %d has no row-major operator", k);
  }
  e->aem_m = m;
  int rc;
  const size_t nm = (size_t)N * m;
  const unsigned grid = (unsigned)((N + EXT_WAVES - 1) / EXT_WAVES);
  for (int k = 0; k < nl; ++k) {
    const Level& lv = e->levels[k];
    std::vector<double> data(m), s2(m, lv.var), winv(m, 0.0);
    HIP_TRY(hipMemcpy(data.data(), lv.udata.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
    if (lv.noise_kind == TDA_NOISE_DIAG) {
      HIP_TRY(hipMemcpy(winv.data(), lv.uw.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
      for (int o = 0; o < m; ++o) s2[o] = 1.0 / winv[o];
    }
    if ((rc = e->aemd_data[k].upload(data)) || (rc = e->aemd_sig2[k].upload(s2))) return rc;
    if (k == nl - 1) {
      if (lv.noise_kind == TDA_NOISE_DIAG) {
        if ((rc = e->aemd_wfin.upload(winv))) return rc;
      } else {
        e->aemd_wfin.release();
      }
    }
    if ((rc = e->aemd_F[k].alloc(nm)) || (rc = e->aemd_bias[k].alloc(nm)) || (rc = e->aemd_w[k].alloc(nm)) ||
        (rc = e->aemd_mu[k].alloc(nm)) || (rc = e->aemd_var[k].alloc(nm)) || (rc = e->aemd_md[k].alloc(nm)))
      return rc;
    // F_k(theta0): "proposals" = the initial states, outputs through callback / source / linear evaluation
    ExtArgs ya{};
    fill_ext_args(e, lv, ya);
    ya.mode = 1;
    ya.theta = e->theta.p;
    ya.scaling = e->scaling.p;
    hipLaunchKernelGGL(k_ext_propose, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, ya);
    if ((rc = ext_model_outputs(e, lv))) return rc;
    HIP_TRY(hipMemcpyAsync(e->aemd_F[k].p, lv.cb_F.p, nm * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    e->aem_bt[k] = 1;
  }
  if ((rc = e->aemd_Fst.alloc((size_t)(nl * (nl - 1) / 2) * nm))) return rc;
  AemdArgs a{};
  fill_aemd_args(e, a);
  hipLaunchKernelGGL(k_aemd_init, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, a);
  HIP_TRY(hipGetLastError());
  (void)d;
  return TDA_OK;
}

int tda_engine_init(tda_engine* e, const double* theta0) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  if (!e->prior_set || !e->prop_set) return fail(TDA_ERR_STATE, "set_prior and set_proposal must precede init");
  for (auto& lv : e->levels)
    if (!lv.set) return fail(TDA_ERR_STATE, "set_level missing");
  if (e->prop_set && !e->is_dreamz && e->pp.kind == TDA_PROP_OWCN) {
    if (e->ow_state_h.empty()) return fail(TDA_ERR_STATE, "operator-weighted pCN: set_proposal_operators missing");
    if (e->levels[0].model != MODEL_LINEAR && e->levels[0].model != MODEL_CALLBACK && e->levels[0].model != MODEL_USER)
      return fail(TDA_ERR_UNSUPPORTED, "operator-weighted pCN is lowered for linear, callback and source-defined forward models");
    if (e->prior_bounded) return fail(TDA_ERR_UNSUPPORTED, "operator-weighted pCN needs a Gaussian prior");
  }
  if (e->prop_set && !e->is_dreamz && e->pp.kind == TDA_PROP_MALA) {
    if (e->levels[0].model != MODEL_LINEAR) return fail(TDA_ERR_UNSUPPORTED, "MALA is lowered for linear forward models only (exact gradient)");
    if (e->prior_bounded) return fail(TDA_ERR_UNSUPPORTED, "MALA needs a Gaussian prior");
    if (e->levels[0].noise_kind == TDA_NOISE_ADAPTIVE) return fail(TDA_ERR_UNSUPPORTED, "MALA: adaptive likelihoods are not lowered");
  }
  {
    int n_cb = 0;
    for (auto& lv : e->levels) n_cb += (lv.model == MODEL_CALLBACK || lv.model == MODEL_USER) ? 1 : 0;
    // DREAMZ below a hierarchy (the reference's MLDA notebook): its jump needs the chain's growing archive at every base step,
    // so such a hierarchy is sequenced by the host like one with callback / source-defined levels, linear levels included
    const bool dz_hier = e->is_dreamz && e->nlev > 1;
    const bool aemd = e->aem == TDA_AEM_STATE_INDEPENDENT_DIAGONAL;  // the diagonal error model works on model-output arrays
    e->ext_hier = (n_cb || dz_hier || aemd) && e->nlev > 1;
    if (e->ext_hier) {  // Delayed Acceptance / MLDA with callback / source-defined models (host-sequenced level actions)
      for (auto& lv : e->levels)  // linear levels may be mixed in (k_ext_linear_eval); anything else may not
        if (lv.model != MODEL_CALLBACK && lv.model != MODEL_USER && (lv.model != MODEL_LINEAR || !lv.A_dev.p))
          return fail(TDA_ERR_UNSUPPORTED, "hierarchies with callback / source-defined levels take linear levels with isotropic or diagonal noise beside them");
      if (e->pp.kind != TDA_PROP_GRW && e->pp.kind != TDA_PROP_PCN && e->pp.kind != TDA_PROP_AM && !dz_hier)
        return fail(TDA_ERR_UNSUPPORTED, "callback hierarchies take GaussianRandomWalk / CrankNicolson / AdaptiveMetropolis / DREAMZ proposals");
      if (dz_hier && e->aem) return fail(TDA_ERR_UNSUPPORTED, "DREAMZ below a hierarchy with an adaptive error model is not lowered");
      if (dz_hier && e->randomize) return fail(TDA_ERR_UNSUPPORTED, "DREAMZ with randomize_subchain_length is not lowered");
      if (e->prior_kind == PRIOR_DENSE) return fail(TDA_ERR_UNSUPPORTED, "callback forward models need a diagonal prior covariance");
    }
  }
  if (e->prior_bounded) {  // JointPrior with uniform components
    // single-level GRW / AM (fused), or a host-sequenced hierarchy (callback / source-defined levels): there the base-level
    // kernels test the support bounds and the upper levels carry the log-prior of the states they promote
    if (e->is_dreamz && e->levels[0].model == MODEL_ROSENBROCK)
      return fail(TDA_ERR_UNSUPPORTED, "priors with uniform components are not lowered for the Rosenbrock example model");
    if (e->is_dreamz && !(e->arch_set && e->arch_given)) return fail(TDA_ERR_INVALID, "priors with uniform components: DREAM(Z) needs an explicit initial archive");
    if (e->pp.kind == TDA_PROP_PCN) return fail(TDA_ERR_UNSUPPORTED, "pCN needs a Gaussian prior");
    if (e->levels[0].noise_kind == TDA_NOISE_DENSE) return fail(TDA_ERR_UNSUPPORTED, "priors with uniform components: iso / diag noise only");
    if (!theta0) return fail(TDA_ERR_INVALID, "priors with uniform components need explicit initial parameters");
  }
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int d = e->d, DP = e->DP;
  const int64_t N = e->N, NP = e->NP;
  int rc;
  if ((rc = e->theta.alloc((size_t)NP * DP))) return rc;
  if ((rc = e->lp.alloc(NP))) return rc;
  if ((rc = e->ll.alloc(NP))) return rc;
  if ((rc = e->acc_count.alloc(NP))) return rc;
  if ((rc = e->flags.alloc(NP))) return rc;
  HIP_TRY(hipMemsetAsync(e->acc_count.p, 0, NP * sizeof(int32_t), e->stream));
  HIP_TRY(hipMemsetAsync(e->flags.p, 0, NP * sizeof(int32_t), e->stream));
  HIP_TRY(hipMemsetAsync(e->lp.p, 0, NP * sizeof(double), e->stream));
  HIP_TRY(hipMemsetAsync(e->ll.p, 0, NP * sizeof(double), e->stream));

  // initial parameters
  std::vector<double> th0;
  if (!theta0) {  // theta0 ~ prior (sampler.py:209) from RNG stream 2
    th0.resize((size_t)N * d);
    // (a few host threads: 4096 chains x 32 Philox / Box-Muller pairs were 17 of the 20 ms an init took)
    host_chain_ranges(N, [&](int64_t c0, int64_t c1) {
      std::vector<double> z(d + 1);
      for (int64_t c = c0; c < c1; ++c) {
        for (int b = 0; b < (d + 1) / 2; ++b)
          normal_pair(e->cfg.seed, (uint32_t)(e->cfg.chain_offset + c), 0u, STREAM_INIT, (uint32_t)b, z[2 * b], z[2 * b + 1]);
        for (int i = 0; i < d; ++i) {
          double s = 0.0;
          for (int k = 0; k <= i; ++k) s = std::fma(e->prior_L_h[(size_t)i * d + k], z[k], s);
          th0[(size_t)c * d + i] = e->prior_mean_h[i] + s;
        }
      }
    });
    theta0 = th0.data();
  }
  if ((rc = upload_states(e, theta0, N, e->theta.p))) return rc;
  if (e->pp.kind == TDA_PROP_INDEPENDENCE) {
    // log q(theta0) up to q's constant: -|L^-1 (theta0 - q_mean)|^2 / 2 (the kernels carry -|z|^2 / 2 for proposals)
    std::vector<double> Lq, th_h;
    if (!cholesky_host(e->prop_C_h.data(), d, Lq)) return fail(TDA_ERR_NUMERIC, "proposal covariance is not positive definite");
    const double* hs = theta0;
    if (is_device_ptr(theta0)) {
      th_h.resize((size_t)N * d);
      HIP_TRY(hipMemcpy(th_h.data(), theta0, th_h.size() * sizeof(double), hipMemcpyDeviceToHost));
      hs = th_h.data();
    }
    std::vector<double> lq0(NP, 0.0), w(d), qm(DP, 0.0);
    for (int64_t c = 0; c < N; ++c) {
      double s2 = 0.0;
      for (int i = 0; i < d; ++i) {
        double s = hs[(size_t)c * d + i] - e->q_mean_h[i];
        for (int k = 0; k < i; ++k) s -= Lq[(size_t)i * d + k] * w[k];
        w[i] = s / Lq[(size_t)i * d + i];
        s2 += w[i] * w[i];
      }
      lq0[c] = -0.5 * s2;
    }
    for (int j = 0; j < d; ++j) qm[j] = e->q_mean_h[j];
    if ((rc = e->lq.upload(lq0)) || (rc = e->q_mean_d.upload(qm))) return rc;
    if ((rc = e->qzblk.alloc((size_t)e->SMAX * NP)) || (rc = e->qzblk2[0].alloc((size_t)e->SMAX * NP)) ||
        (rc = e->qzblk2[1].alloc((size_t)e->SMAX * NP)))
      return rc;
  }

  // proposal state (chain.py:74-76)
  std::vector<double> sc(NP, (e->pp.kind == TDA_PROP_AM || e->pp.kind == TDA_PROP_INDEPENDENCE) ? 1.0 : e->pp.scaling);
  if ((rc = e->scaling.upload(sc))) return rc;
  if (e->is_dreamz) {
    if (!e->arch_set && (rc = tda_engine_set_archive(e, nullptr))) return rc;
    const bool sh = e->dz.shared != 0;
    const int64_t M0 = e->dz.M0;
    e->arch_cap = e->dz.capacity;
    const size_t narch = (size_t)(sh ? 1 : NP) * e->arch_cap * DP;
    if ((rc = e->arch.alloc(narch))) return rc;
    HIP_TRY(hipMemsetAsync(e->arch.p, 0, narch * sizeof(double), e->stream));
    {  // Z0 -> padded device layout
      const int64_t nA = sh ? 1 : N;
      std::vector<double> pad((size_t)M0 * DP, 0.0);
      for (int64_t c = 0; c < nA; ++c) {
        for (int64_t r = 0; r < M0; ++r)
          for (int j = 0; j < d; ++j) pad[(size_t)r * DP + j] = e->Z0_h[((size_t)c * M0 + r) * d + j];
        HIP_TRY(hipMemcpy(e->arch.p + (size_t)c * e->arch_cap * DP, pad.data(), pad.size() * sizeof(double), hipMemcpyHostToDevice));
      }
    }
    e->arch_rows = M0;
    e->sums_rows = 0;
    const size_t nsum = (size_t)(sh ? 1 : NP) * DP;
    std::vector<double> zero(nsum, 0.0), pcr((size_t)NP * MAX_NCR, 0.0), ones((size_t)NP * MAX_NCR, 1.0), zer2((size_t)NP * MAX_NCR, 0.0);
    for (int64_t c = 0; c < NP; ++c)
      for (int k = 0; k < e->dz.nCR; ++k) pcr[(size_t)c * MAX_NCR + k] = 1.0 / e->dz.nCR;  // proposal.py:731
    if ((rc = e->zsum.upload(zero))) return rc;
    if ((rc = e->zsq.upload(zero))) return rc;
    if ((rc = e->dz_pCR.upload(pcr))) return rc;
    if ((rc = e->dz_Delta.upload(ones))) return rc;  // proposal.py:755
    if ((rc = e->dz_LCR.upload(zer2))) return rc;    // proposal.py:754
    if ((rc = e->dz_mcr_last.alloc(NP))) return rc;
    HIP_TRY(hipMemsetAsync(e->dz_mcr_last.p, 0, NP * sizeof(int32_t), e->stream));
    if ((rc = e->dz_coef.alloc((size_t)e->SMAX * NP * DP))) return rc;
    if ((rc = e->dz_epsm.alloc((size_t)e->SMAX * NP * DP))) return rc;
    if ((rc = e->dz_ridx.alloc((size_t)e->SMAX * NP * 2 * MAX_DELTA))) return rc;
    if ((rc = e->theta_prev.alloc((size_t)NP * DP))) return rc;
    HIP_TRY(hipMemcpyAsync(e->theta_prev.p, e->theta.p, (size_t)NP * DP * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    if (sh) {
      if ((rc = e->blk_states.alloc((size_t)e->SMAX * NP * DP))) return rc;
      if ((rc = e->blk_hist.alloc((size_t)e->SMAX * NP * DP))) return rc;
    }
    if ((rc = e->ublk.alloc((size_t)e->SMAX * NP))) return rc;
    if ((rc = e->rec_params.alloc((size_t)e->SMAX * N * d))) return rc;
    if ((rc = e->rec_stats.alloc((size_t)e->SMAX * N * 3))) return rc;
    if ((rc = e->rec_acc.alloc((size_t)e->SMAX * N))) return rc;
    e->t = 0;
    e->k_adapt = 0;
    e->rp_pos = 0;
    e->exp_pos = 0;
    e->pending_steps = 0;
    e->inited = true;  // dreamz_sums_catchup checks nothing else
    if ((rc = dreamz_sums_catchup(e, 0, M0, false, false, 1.0))) return rc;
    if ((rc = e->theta_last.alloc((size_t)NP * DP))) return rc;
    // initial link (chain.py:70); below a hierarchy the level set-up further down evaluates every level at theta0
    if (e->nlev > 1) {
      if (e->levels[0].model == MODEL_ROSENBROCK) return fail(TDA_ERR_UNSUPPORTED, "the Rosenbrock example model is single-level");
    } else if (e->levels[0].model == MODEL_LINEAR || e->levels[0].model == MODEL_CALLBACK || e->levels[0].model == MODEL_USER) {
      if ((rc = launch_eval(e, 0, e->theta.p, e->lp.p, e->ll.p))) return rc;
    } else {
      // Rosenbrock level: evaluate theta0 with a zero-jump DREAMZ step (coef = eps = 0, u = 0 -> accepted)
      HIP_TRY(hipMemsetAsync(e->dz_coef.p, 0, (size_t)NP * DP * sizeof(double), e->stream));
      HIP_TRY(hipMemsetAsync(e->dz_epsm.p, 0, (size_t)NP * DP * sizeof(double), e->stream));
      HIP_TRY(hipMemsetAsync(e->dz_ridx.p, 0, (size_t)NP * 2 * MAX_DELTA * sizeof(int32_t), e->stream));
      std::vector<double> uz(NP, -1.0), big(NP, -1e300);
      HIP_TRY(hipMemcpy(e->ublk.p, uz.data(), NP * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(e->lp.p, big.data(), NP * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(hipMemsetAsync(e->ll.p, 0, NP * sizeof(double), e->stream));
      DreamStepArgs sa{};
      extern void fill_dreamz_step_args(tda_engine*, DreamStepArgs&);
      fill_dreamz_step_args(e, sa);
      sa.S = 1;
      sa.shared = 1;  // no archive append for this evaluation
      sa.blk_states = nullptr;
      sa.rec_params = nullptr;
      sa.rec_stats = nullptr;
      sa.rec_acc = nullptr;
      const size_t lds = ((size_t)16 * (DP + 2) + 128 + 64) * sizeof(double);
      DISPATCH_DPAD(DP, launch_dz_steps<DPAD>(sa, lds, e->stream));
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemsetAsync(e->acc_count.p, 0, NP * sizeof(int32_t), e->stream));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->nlev == 1) return TDA_OK;
  }
  const bool owcn = !e->is_dreamz && e->pp.kind == TDA_PROP_OWCN, mala = !e->is_dreamz && e->pp.kind == TDA_PROP_MALA;
  if (!e->is_dreamz) {  // ---- Gaussian proposals: factor of the proposal covariance (DREAMZ has its archive instead) ----
  std::vector<double> L;
  const double* Cuse = (e->pp.kind == TDA_PROP_PCN || owcn) ? e->prior_cov_h.data() : e->prop_C_h.data();  // proposal.py:336-341
  if (!mala && !cholesky_host(Cuse, d, L)) return fail(TDA_ERR_NUMERIC, "proposal covariance is not positive definite");
  std::vector<double> Lk((size_t)DP * DP, 0.0);
  if (mala) {
    // np.random.normal(size = d) (proposal.py:955): unit factor.  Gradient of the log-posterior of the linear-Gaussian
    // target (proposal.py:986-998; utils.py:273-287) in closed form: grad = c - H theta with
    //   H = Sigma_prior^-1 + A^T Sigma_e^-1 A,   c = Sigma_prior^-1 mu + A^T Sigma_e^-1 (data - b)
    for (int j = 0; j < d; ++j) Lk[(size_t)j * DP + j] = 1.0;
    const Level& lv0 = e->levels[0];
    const int m = lv0.m;
    std::vector<double> Wp, H((size_t)DP * DP, 0.0), cv(DP, 0.0);
    tri_inverse_host(e->prior_L_h, d, Wp);
    for (int i = 0; i < d; ++i)
      for (int j = 0; j <= i; ++j) {
        double sacc = 0.0;
        for (int k = i; k < d; ++k) sacc += Wp[(size_t)k * d + i] * Wp[(size_t)k * d + j];
        H[(size_t)i * DP + j] = H[(size_t)j * DP + i] = sacc;
      }
    for (int i = 0; i < d; ++i) {
      double sacc = 0.0;
      for (int j = 0; j < d; ++j) sacc += H[(size_t)i * DP + j] * e->prior_mean_h[j];
      cv[i] = sacc;
    }
    // WA = Sigma_e^-1 A  [m][d]
    std::vector<double> WA((size_t)m * d, 0.0);
    if (lv0.noise_kind == TDA_NOISE_DENSE) {
      for (int o = 0; o < m; ++o)
        for (int q = 0; q < m; ++q) {
          const double pv = lv0.Pinv_h[(size_t)o * m + q];
          for (int j = 0; j < d; ++j) WA[(size_t)o * d + j] += pv * lv0.A_h[(size_t)q * d + j];
        }
    } else {
      for (int o = 0; o < m; ++o) {
        const double wv = lv0.noise_kind == TDA_NOISE_DIAG ? lv0.w_h[o] : 1.0 / lv0.var;
        for (int j = 0; j < d; ++j) WA[(size_t)o * d + j] = wv * lv0.A_h[(size_t)o * d + j];
      }
    }
    for (int o = 0; o < m; ++o)
      for (int i = 0; i < d; ++i) {
        const double ai = lv0.A_h[(size_t)o * d + i];
        for (int j = 0; j < d; ++j) H[(size_t)i * DP + j] += ai * WA[(size_t)o * d + j];
      }
    for (int o = 0; o < m; ++o)
      for (int j = 0; j < d; ++j) cv[j] += WA[(size_t)o * d + j] * lv0.ytil_h[o];
    if ((rc = e->mala_H.upload(H))) return rc;
    if ((rc = e->mala_c.upload(cv))) return rc;
    if ((rc = e->mala_grad.alloc((size_t)NP * DP))) return rc;
  } else if (owcn) {
    // increments = noise_operator N(0, C_prior) = (noise_operator chol(C_prior)) z (proposal.py:596-598): a full factor
    for (int j = 0; j < d; ++j)
      for (int k = 0; k < d; ++k) {
        double g = 0.0;
        for (int i = k; i < d; ++i) g = std::fma(e->ow_noise_h[(size_t)j * d + i], L[(size_t)i * d + k], g);
        Lk[(size_t)k * DP + j] = g;
      }
    std::vector<double> St((size_t)DP * DP, 0.0);
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < d; ++j) St[(size_t)j * DP + i] = e->ow_state_h[(size_t)i * d + j];
    if ((rc = e->ow_SopT.upload(St))) return rc;
  } else {
    for (int j = 0; j < d; ++j)
      for (int k = 0; k <= j; ++k) Lk[(size_t)k * DP + j] = L[(size_t)j * d + k];
  }
  if (e->pp.kind == TDA_PROP_AM) {
    e->L_shared = false;
    if ((rc = e->Lk.alloc((size_t)NP * DP * DP))) return rc;
    // one factor over PCIe, replicated to every chain by doubling device copies (the 128 MiB of 4096 copies took 10 ms from
    // pageable host memory)
    HIP_TRY(hipMemcpyAsync(e->Lk.p, Lk.data(), (size_t)DP * DP * sizeof(double), hipMemcpyHostToDevice, e->stream));
    for (int64_t have = 1; have < NP; have *= 2)
      HIP_TRY(hipMemcpyAsync(e->Lk.p + (size_t)have * DP * DP, e->Lk.p, (size_t)std::min<int64_t>(have, NP - have) * DP * DP * sizeof(double),
                             hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));  // (Lk is a local: the first copy must have left the host before it goes)
    // RecursiveSampleMoments(mu0 = theta0, sigma0 = 0) (proposal.py:495-500)
    if ((rc = e->am_mu.alloc((size_t)NP * DP))) return rc;
    const size_t nsig = (size_t)NP * am_tiles_rt(DP) * 256;  // lower 16x16 tiles in MFMA C/D layout (k_adapt)
    if ((rc = e->am_sigma.alloc(nsig))) return rc;
    HIP_TRY(hipMemcpyAsync(e->am_mu.p, e->theta.p, (size_t)NP * DP * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipMemsetAsync(e->am_sigma.p, 0, nsig * sizeof(double), e->stream));
  } else {
    e->L_shared = true;
    if ((rc = e->Lk.upload(Lk))) return rc;
  }
  e->L_identity = false;
  if (e->L_shared && (e->pp.kind == TDA_PROP_GRW || e->pp.kind == TDA_PROP_PCN)) {
    bool ident = true;
    for (int k = 0; k < d && ident; ++k)
      for (int j = 0; j < d; ++j)
        if (Lk[(size_t)k * DP + j] != (j == k ? 1.0 : 0.0)) {
          ident = false;
          break;
        }
    e->L_identity = ident;
  }
  }

  // block buffers
  if (!e->is_dreamz && (rc = e->inc.alloc((size_t)e->SMAX * NP * DP))) return rc;
  if ((rc = e->ublk.alloc((size_t)e->SMAX * NP))) return rc;
  if ((rc = e->lublk.alloc((size_t)e->SMAX * NP))) return rc;
  if ((rc = e->rec_params.alloc((size_t)e->SMAX * N * d))) return rc;
  if ((rc = e->rec_stats.alloc((size_t)e->SMAX * N * 3))) return rc;
  if ((rc = e->rec_acc.alloc((size_t)e->SMAX * N))) return rc;

  e->t = 0;
  e->k_adapt = 0;
  e->rep_pos = 0;
  e->exp_pos = 0;
  // initial links (chain.py:70)
  if ((rc = launch_eval(e, 0, e->theta.p, e->lp.p, e->ll.p))) return rc;
  if (mala) {
    const int64_t nt = NP * DP;
    hipLaunchKernelGGL(k_mala_grad0, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, e->stream, NP, DP, e->mala_H.p, e->mala_c.p,
                       e->theta.p, e->mala_grad.p);
    HIP_TRY(hipGetLastError());
  }
  if (e->nlev > 1) {
    // every level starts from theta0 (chain.py:253-261; proposal.py:1379); S[j][q] = level j's densities there
    if (!e->sub_set) return fail(TDA_ERR_STATE, "set_subchains must precede init for n_levels > 1");
    const int nl = e->nlev, npair = nl * (nl - 1) / 2;
    if ((rc = e->ml_theta.alloc((size_t)nl * NP * DP))) return rc;
    if ((rc = e->ml_lp.alloc((size_t)nl * NP))) return rc;
    if ((rc = e->ml_ll.alloc((size_t)nl * NP))) return rc;
    if ((rc = e->ml_S.alloc((size_t)npair * 2 * NP))) return rc;
    if ((rc = e->ml_anyacc.alloc((size_t)nl * NP))) return rc;
    if ((rc = e->ml_ysnap.alloc((size_t)NP * (DP + 2)))) return rc;
    if ((rc = e->ml_pick.alloc(NP))) return rc;
    e->ring_P = e->pp.adaptive ? e->pp.period + MAXLEV : 1;
    if ((rc = e->ml_ring.alloc((size_t)e->ring_P * NP))) return rc;
    HIP_TRY(hipMemsetAsync(e->ml_anyacc.p, 0, (size_t)nl * NP * sizeof(int32_t), e->stream));
    HIP_TRY(hipMemsetAsync(e->ml_ysnap.p, 0, (size_t)NP * (DP + 2) * sizeof(double), e->stream));
    HIP_TRY(hipMemsetAsync(e->ml_pick.p, 0, NP * sizeof(int32_t), e->stream));
    HIP_TRY(hipMemsetAsync(e->ml_ring.p, 0, (size_t)e->ring_P * NP, e->stream));
    for (int k = 0; k < nl; ++k) {
      double* thk = e->ml_theta.p + (size_t)k * NP * DP;
      HIP_TRY(hipMemcpyAsync(thk, e->theta.p, (size_t)NP * DP * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      if ((rc = launch_eval(e, k, thk, e->ml_lp.p + (size_t)k * NP, e->ml_ll.p + (size_t)k * NP))) return rc;
    }
    for (int q = 1; q < nl; ++q)
      for (int j = 0; j < q; ++j) {
        const int p = q * (q - 1) / 2 + j;
        HIP_TRY(hipMemcpyAsync(e->ml_S.p + ((size_t)p * 2 + 0) * NP, e->ml_lp.p + (size_t)j * NP, NP * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        HIP_TRY(hipMemcpyAsync(e->ml_S.p + ((size_t)p * 2 + 1) * NP, e->ml_ll.p + (size_t)j * NP, NP * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      }
    for (int k = 0; k < MAXLEV; ++k) {
      e->cnt[k] = 0;
      e->done[k] = 0;
      e->u_rep_lv_pos[k] = 0;
    }
    e->ridx_rep_pos = 0;
    e->ring_pos = 0;
    for (int k = 0; k < nl; ++k) {
      if ((rc = e->ml_rec_params[k].alloc((size_t)e->SMAX * N * d))) return rc;
      if ((rc = e->ml_rec_stats[k].alloc((size_t)e->SMAX * N * 3))) return rc;
      if ((rc = e->ml_rec_acc[k].alloc((size_t)e->SMAX * N))) return rc;
    }
    if ((rc = e->ml_sid.alloc((size_t)nl * NP))) return rc;
    HIP_TRY(hipMemsetAsync(e->ml_sid.p, 0, (size_t)nl * NP * sizeof(int64_t), e->stream));
    if (e->aem == TDA_AEM_STATE_INDEPENDENT_DIAGONAL) {
      if ((rc = init_error_model_diagonal(e))) return rc;
    } else if (e->aem) {
      // ---- adaptive error model set-up (chain.py:268-305; :643-678; proposal.py:1407-1467) ----
      if (e->randomize) return fail(TDA_ERR_UNSUPPORTED, "randomize_subchain_length together with an error model is not lowered");
      if (e->prior_kind == PRIOR_DENSE && e->aem == TDA_AEM_STATE_DEPENDENT && e->pp.kind != TDA_PROP_PCN) {}
      const int m = e->levels[0].m;
      for (int k = 0; k < nl; ++k) {
        if (e->levels[k].m != m) return fail(TDA_ERR_INVALID, "error model: all levels must share the output dimension");
        const bool want_adaptive = k < nl - 1;
        if (want_adaptive && e->levels[k].noise_kind != TDA_NOISE_ADAPTIVE)
          return fail(TDA_ERR_INVALID, "error model: level %d needs an AdaptiveGaussianLogLike (TDA_NOISE_ADAPTIVE)", k);
        if (!want_adaptive && e->levels[k].noise_kind != TDA_NOISE_ISO)
          return fail(TDA_ERR_UNSUPPORTED, "error model: the finest level must have an isotropic likelihood on the device");
      }
      e->aem_m = m;
      const int AEM_MP = e->aem_ld = e->levels[0].em_ld;
      HIP_TRY(hipStreamSynchronize(e->stream));
      // host copies of theta0 / log-priors, model outputs of every level
      std::vector<double> th((size_t)NP * DP), lph(NP);
      HIP_TRY(hipMemcpy(th.data(), e->theta.p, th.size() * sizeof(double), hipMemcpyDeviceToHost));
      std::vector<std::vector<double>> F(nl, std::vector<double>((size_t)N * m));
      if (e->ext_hier) {
        // callback / source-defined (and linear) levels of a host-sequenced hierarchy: the model outputs at theta0 come
        // from the same evaluation path the steps use
        for (int k = 0; k < nl; ++k) {
          const Level& lvk = e->levels[k];
          ExtArgs ya{};
          fill_ext_args(e, lvk, ya);
          ya.mode = 1;
          ya.theta = e->theta.p;
          ya.scaling = e->scaling.p;
          hipLaunchKernelGGL(k_ext_propose, dim3((unsigned)((N + EXT_WAVES - 1) / EXT_WAVES)), dim3(64 * EXT_WAVES), 0, e->stream, ya);
          if ((rc = ext_model_outputs(e, lvk))) return rc;
          HIP_TRY(hipStreamSynchronize(e->stream));
          HIP_TRY(hipMemcpy(F[k].data(), lvk.cb_F.p, F[k].size() * sizeof(double), hipMemcpyDeviceToHost));
          for (int64_t c = 0; c < N; ++c)
            for (int o = 0; o < m; ++o) F[k][(size_t)c * m + o] -= lvk.data_h[o];  // residual form; Fout() adds the data back
        }
      } else
      for (int k = 0; k < nl; ++k)
        for (int64_t c = 0; c < N; ++c)
          for (int o = 0; o < m; ++o) {
            double f = 0.0;
            for (int j = 0; j < d; ++j) f = std::fma(e->levels[k].A_h[(size_t)o * d + j], th[(size_t)c * DP + j], f);
            F[k][(size_t)c * m + o] = f - e->levels[k].ytil_h[o];  // F - y + (y - ytil) handled below: residual form
          }
      // residual r_k = F_k - ytil_k ; model output F_k = r_k + ytil_k + b_k - b_k ... differences of outputs:
      // F_q - F_{q-1} = (r_q + y_q) - (r_{q-1} + y_{q-1}) with y = data (ytil = data - b, F = A theta + b)
      auto Fout = [&](int k, int64_t c, int o) { return F[k][(size_t)c * m + o] + e->levels[k].data_h[o]; };
      const size_t MM = (size_t)AEM_MP * AEM_MP;
      std::vector<std::vector<double>> mdiff(nl, std::vector<double>((size_t)NP * AEM_MP, 0.0));
      for (int q = 1; q < nl; ++q)
        for (int64_t c = 0; c < N; ++c)
          for (int o = 0; o < m; ++o) mdiff[q][(size_t)c * AEM_MP + o] = Fout(q, c, o) - Fout(q - 1, c, o);
      for (int q = 1; q < nl; ++q) {
        if ((rc = e->aem_mdiff[q].upload(mdiff[q]))) return rc;
        if ((rc = e->aem_bmu[q].upload(mdiff[q]))) return rc;  // RecursiveSampleMoments(mu0 = model_diff, sigma0 = 0)
        if ((rc = e->aem_bsig[q].alloc((size_t)NP * MM))) return rc;
        HIP_TRY(hipMemset(e->aem_bsig[q].p, 0, (size_t)NP * MM * sizeof(double)));
        e->aem_bt[q] = 1;
      }
      std::vector<double> llh(NP, 0.0);
      for (int k = 0; k < nl - 1; ++k) {
        // Sigma_e^-1 (distributions.py:280), identical for all chains until a bias covariance exceeds 1e-9
        std::vector<double> Lc, W, P(MM, 0.0);
        cholesky_host(e->levels[k].cov_h.data(), m, Lc);
        tri_inverse_host(Lc, m, W);
        for (int i = 0; i < m; ++i)
          for (int j = 0; j <= i; ++j) {
            double s = 0.0;
            for (int r = i; r < m; ++r) s += W[(size_t)r * m + i] * W[(size_t)r * m + j];
            P[(size_t)i * AEM_MP + j] = P[(size_t)j * AEM_MP + i] = s;
          }
        if ((rc = e->aem_covinv[k].alloc((size_t)NP * MM))) return rc;
        HIP_TRY(hipMemcpy(e->aem_covinv[k].p, P.data(), MM * sizeof(double), hipMemcpyHostToDevice));
        for (int64_t have = 1; have < NP; have *= 2)  // replicate to every chain by doubling device copies
          HIP_TRY(hipMemcpy(e->aem_covinv[k].p + (size_t)have * MM, e->aem_covinv[k].p,
                            (size_t)std::min<int64_t>(have, NP - have) * MM * sizeof(double), hipMemcpyDeviceToDevice));
        // total bias: state-dependent = the difference itself, otherwise the sum of the means of all trackers above
        std::vector<double> bt((size_t)NP * AEM_MP, 0.0);
        for (int64_t c = 0; c < N; ++c)
          for (int o = 0; o < m; ++o) {
            double s = 0.0;
            if (e->aem == TDA_AEM_STATE_DEPENDENT) s = mdiff[k + 1][(size_t)c * AEM_MP + o];
            else
              for (int p = k + 1; p < nl; ++p) s += mdiff[p][(size_t)c * AEM_MP + o];
            bt[(size_t)c * AEM_MP + o] = s;
          }
        if ((rc = e->aem_bias[k].upload(bt))) return rc;
        // update_link of the initial link of level k
        for (int64_t c = 0; c < N; ++c) {
          double acc = 0.0;
          for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int j = 0; j < m; ++j) s += P[(size_t)i * AEM_MP + j] * (F[k][(size_t)c * m + j] + bt[(size_t)c * AEM_MP + j]);
            acc += (F[k][(size_t)c * m + i] + bt[(size_t)c * AEM_MP + i]) * s;
          }
          llh[c] = -0.5 * acc;
        }
        HIP_TRY(hipMemcpy(e->ml_ll.p + (size_t)k * NP, llh.data(), NP * sizeof(double), hipMemcpyHostToDevice));
      }
      // S[j][q] = densities of level j at theta0 under the initial error model
      for (int qq = 1; qq < nl; ++qq)
        for (int j = 0; j < qq; ++j) {
          const int p = qq * (qq - 1) / 2 + j;
          HIP_TRY(hipMemcpy(e->ml_S.p + ((size_t)p * 2 + 0) * NP, e->ml_lp.p + (size_t)j * NP, NP * sizeof(double), hipMemcpyDeviceToDevice));
          HIP_TRY(hipMemcpy(e->ml_S.p + ((size_t)p * 2 + 1) * NP, e->ml_ll.p + (size_t)j * NP, NP * sizeof(double), hipMemcpyDeviceToDevice));
        }
      if (e->ext_hier) {  // model outputs of the current links and of level j at theta_q: all at theta0
        const int npair = nl * (nl - 1) / 2;
        std::vector<std::vector<double>> Fo(nl, std::vector<double>((size_t)NP * AEM_MP, 0.0));
        for (int k = 0; k < nl; ++k) {
          for (int64_t c = 0; c < N; ++c)
            for (int o = 0; o < m; ++o) Fo[k][(size_t)c * AEM_MP + o] = Fout(k, c, o);
          if ((rc = e->ext_Fcur[k].upload(Fo[k]))) return rc;
        }
        if ((rc = e->ext_Fst.alloc((size_t)npair * NP * AEM_MP))) return rc;
        for (int qq = 1; qq < nl; ++qq)
          for (int j = 0; j < qq; ++j)
            HIP_TRY(hipMemcpy(e->ext_Fst.p + (size_t)(qq * (qq - 1) / 2 + j) * NP * AEM_MP, Fo[j].data(), Fo[j].size() * sizeof(double), hipMemcpyHostToDevice));
      }
      // whitening matrix of the prior, row-major (pCN transition densities of the state-dependent acceptance)
      std::vector<double> Wp;
      tri_inverse_host(e->prior_L_h, d, Wp);
      if ((rc = e->prior_W_rm.upload(Wp))) return rc;
    }
  }
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->inited = true;
  return TDA_OK;
}

int tda_engine_get_current(tda_engine* e, double* theta, double* stats) {
  if (!e || !e->inited) return fail(TDA_ERR_STATE, "engine not initialised");
  if (e->nlev > 1) return tda_engine_get_level_state(e, e->nlev - 1, theta, stats);
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const int64_t N = e->N, NP = e->NP;
  if (theta) {
    std::vector<double> h((size_t)NP * e->DP), o((size_t)N * e->d);
    HIP_TRY(hipMemcpy(h.data(), e->theta.p, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < N; ++c)
      for (int j = 0; j < e->d; ++j) o[(size_t)c * e->d + j] = h[(size_t)c * e->DP + j];
    HIP_TRY(hipMemcpy(theta, o.data(), o.size() * sizeof(double), is_device_ptr(theta) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
  }
  if (stats) {
    std::vector<double> a(NP), b(NP), o((size_t)N * 3);
    HIP_TRY(hipMemcpy(a.data(), e->lp.p, NP * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(b.data(), e->ll.p, NP * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t c = 0; c < N; ++c) {
      o[c * 3] = a[c];
      o[c * 3 + 1] = b[c];
      o[c * 3 + 2] = a[c] + b[c];
    }
    HIP_TRY(hipMemcpy(stats, o.data(), o.size() * sizeof(double), is_device_ptr(stats) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
  }
  return TDA_OK;
}

int tda_engine_set_replay(tda_engine* e, const double* z, const double* u, int64_t n_steps) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  e->rep_steps = 0;
  e->rep_pos = 0;
  e->z_rep.release();
  e->u_rep.release();
  if (!z || !u || n_steps <= 0) return TDA_OK;
  int rc;
  const size_t nz = (size_t)n_steps * e->N * e->d, nu = (size_t)n_steps * e->N;
  if ((rc = e->z_rep.alloc(nz))) return rc;
  if ((rc = e->u_rep.alloc(nu))) return rc;
  HIP_TRY(hipMemcpy(e->z_rep.p, z, nz * sizeof(double), is_device_ptr(z) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(e->u_rep.p, u, nu * sizeof(double), is_device_ptr(u) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  e->rep_steps = n_steps;
  return TDA_OK;
}

int tda_engine_set_export(tda_engine* e, double* z, double* u, int64_t n_steps) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  e->z_exp = e->u_exp = nullptr;
  e->exp_steps = 0;
  e->exp_pos = 0;
  e->z_exp_d.release();
  e->u_exp_d.release();
  if (!z || !u || n_steps <= 0) return TDA_OK;
  if (is_device_ptr(z) != is_device_ptr(u)) return fail(TDA_ERR_INVALID, "export buffers must both be host or both be device");
  e->exp_dev = is_device_ptr(z);
  if (!e->exp_dev) {
    int rc;
    if ((rc = e->z_exp_d.alloc((size_t)n_steps * e->N * e->d))) return rc;
    if ((rc = e->u_exp_d.alloc((size_t)n_steps * e->N))) return rc;
  }
  e->z_exp = z;
  e->u_exp = u;
  e->exp_steps = n_steps;
  return TDA_OK;
}


// A caller's record buffers must hold `need` records of this level (tda_outputs.rows states the capacity); for device
// pointers the extent of the underlying allocation is checked as well, so a short buffer is an error code, never a fault.
static int check_out_capacity(const tda_outputs* o, int level, int64_t need, int64_t N, int d) {
  if (!o || (!o->params && !o->stats && !o->accepted)) return TDA_OK;
  if ((int64_t)o->rows < need)
    return fail(TDA_ERR_INVALID, "tda_outputs[%d].rows = %u but this run() produces %lld records for that level", level, o->rows,
                (long long)need);
  const void* ptr[3] = {o->params, o->stats, o->accepted};
  const size_t bytes[3] = {(size_t)need * N * d * sizeof(double), (size_t)need * N * 3 * sizeof(double), (size_t)need * N};
  static const char* const nm[3] = {"params", "stats", "accepted"};
  for (int i = 0; i < 3; ++i) {
    if (!ptr[i] || !is_device_ptr(ptr[i])) continue;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr[i]) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    const size_t room = size - (size_t)((const char*)ptr[i] - (const char*)base);
    if (room < bytes[i])
      return fail(TDA_ERR_INVALID, "tda_outputs[%d].%s: device allocation has %zu bytes after the pointer, %zu needed", level, nm[i],
                  room, bytes[i]);
  }
  return TDA_OK;
}

static int run_multilevel(tda_engine* e, int64_t n_fine, const tda_outputs* outs);
static int run_dreamz(tda_engine* e, int64_t n_iter, const tda_outputs* out);

int tda_engine_run(tda_engine* e, int64_t n_iter, const tda_outputs* out) {
  if (!e || !e->inited) return fail(TDA_ERR_STATE, "engine not initialised");
  if (n_iter < 0) return fail(TDA_ERR_INVALID, "n_iterations < 0");
  if (e->nlev > 1) return run_multilevel(e, n_iter, out);
  if (e->is_dreamz) return run_dreamz(e, n_iter, out);
  if (out && out->struct_size != sizeof(tda_outputs)) return fail(TDA_ERR_INVALID, "tda_outputs.struct_size mismatch");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (e->rep_steps && e->rep_pos + n_iter > e->rep_steps)
    return fail(TDA_ERR_INVALID, "replay buffer holds %lld steps, %lld requested", (long long)(e->rep_steps - e->rep_pos), (long long)n_iter);
  if (e->exp_steps && e->exp_pos + n_iter > e->exp_steps)
    return fail(TDA_ERR_INVALID, "export buffer too small");
  const int thin = e->thin;
  if (int crc = check_out_capacity(out, 0, kept_records(e->t, n_iter, thin), e->N, e->d)) return crc;

  const int d = e->d;
  const int64_t N = e->N, NP = e->NP;
  int64_t out_row = 0;  // thinning: next row of the caller's buffers
  double* o_params = out ? out->params : nullptr;
  double* o_stats = out ? out->stats : nullptr;
  uint8_t* o_acc = out ? out->accepted : nullptr;
  const bool p_dev = is_device_ptr(o_params), s_dev = is_device_ptr(o_stats), a_dev = is_device_ptr(o_acc);
  const bool is_am = e->pp.kind == TDA_PROP_AM;
  const bool adaptive = e->pp.adaptive != 0;
  const bool periodic = is_am || adaptive;
  const int period = e->pp.period;
  const Level& lv = e->levels[0];
  const size_t lds = steps_lds_bytes(e, lv);
  bool host_copies = false;
  // Host outputs in pinned memory (hipHostMalloc / torch pin_memory): the block's records are copied by a second
  // stream while the next block computes (two sets of device block buffers).  Pageable memory: one synchronous copy
  // per block, as before.
  const bool any_host = (o_params && !p_dev) || (o_stats && !s_dev) || (o_acc && !a_dev);
  const bool async_host = any_host && (!o_params || p_dev || is_pinned_host_ptr(o_params)) &&
                          (!o_stats || s_dev || is_pinned_host_ptr(o_stats)) && (!o_acc || a_dev || is_pinned_host_ptr(o_acc));
  if (async_host && !e->copy_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(hipEventCreateWithFlags(&e->ev_rec[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->ev_cp[i], hipEventDisableTiming));
    }
    int rc;
    if ((rc = e->rec_params2.alloc((size_t)e->SMAX * N * d)) || (rc = e->rec_stats2.alloc((size_t)e->SMAX * N * 3)) ||
        (rc = e->rec_acc2.alloc((size_t)e->SMAX * N)))
      return rc;
  }

  if (e->profiling) {
    for (auto& t : e->timed) {
      (void)hipEventDestroy(t.a);
      (void)hipEventDestroy(t.b);
    }
    e->timed.clear();
  }

  // Philox mode: the normals and uniforms of a block do not depend on the chains, so block b+1's are drawn on a second
  // stream while block b's k_mh_steps runs (k_rng fits into the registers the step kernel leaves free); only
  // INC = Z L^T (k_apply) stays on the critical path behind the Cholesky swap.  Replay mode keeps the fused k_propose.
  static const bool split_ok = !(getenv("TINYDA_SPLIT_PROPOSE") && atoi(getenv("TINYDA_SPLIT_PROPOSE")) == 0);
  const bool split = split_ok && !e->rep_steps;
  if (split && !e->rng_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&e->rng_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(hipEventCreateWithFlags(&e->ev_rng[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->ev_apply[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->ev_steps[i], hipEventDisableTiming));
      int rc;
      if ((rc = e->zfrag[i].alloc((size_t)((e->SMAX + 15) / 16 * 16) * NP * e->DP))) return rc;
      if ((rc = e->ublk2[i].alloc((size_t)e->SMAX * NP))) return rc;
      if ((rc = e->lublk2[i].alloc((size_t)e->SMAX * NP))) return rc;
    }
  }
  auto block_len = [&](int64_t t_now, int64_t left) {
    int64_t S = std::min<int64_t>(left, e->SMAX);
    if (periodic) S = std::min<int64_t>(S, period - (t_now % period));
    return S;
  };
  // draw block `blk` (steps t0 .. t0 + S - 1, the exp_off-th exported step) into buffer blk & 1 on the rng stream
  auto enqueue_rng = [&](int64_t blk, int64_t t0, int64_t S, int64_t exp_off) -> int {
    const int b = (int)(blk & 1);
    RngArgs ra{};
    ra.N = N;
    ra.NP = NP;
    ra.chain_offset = e->cfg.chain_offset;
    ra.d = d;
    ra.S = (int)S;
    ra.step0 = t0;
    ra.seed = e->cfg.seed;
    ra.zf = e->zfrag[b].p;
    ra.u = e->ublk2[b].p;
    ra.logu = e->lublk2[b].p;
    ra.qz = e->pp.kind == TDA_PROP_INDEPENDENCE ? e->qzblk2[b].p : nullptr;
    if (e->exp_steps) {
      ra.z_export = (e->exp_dev ? e->z_exp : e->z_exp_d.p) + (size_t)exp_off * N * d;
      ra.u_export = (e->exp_dev ? e->u_exp : e->u_exp_d.p) + (size_t)exp_off * N;
    }
    if (blk >= 2) {  // the buffers were last read by block blk - 2: its k_apply (fragments) and k_mh_steps (uniforms)
      HIP_TRY(hipStreamWaitEvent(e->rng_stream, e->ev_apply[b], 0));
      HIP_TRY(hipStreamWaitEvent(e->rng_stream, e->ev_steps[b], 0));
    }
    DISPATCH_DPAD(e->DP, launch_rng<DPAD>(ra, e->rng_stream));
    HIP_TRY(hipEventRecord(e->ev_rng[b], e->rng_stream));
    return TDA_OK;
  };

  int64_t done = 0, blk = 0;
  bool inc_ready = false;  // the block about to start already has its increments (k_chol_apply of the block before it)
  if (split && n_iter > 0) {
    // everything queued on the main stream so far (init, earlier run() calls) precedes the first draw
    HIP_TRY(hipEventRecord(e->ev_steps[0], e->stream));
    HIP_TRY(hipStreamWaitEvent(e->rng_stream, e->ev_steps[0], 0));
    int rc = enqueue_rng(0, e->t, block_len(e->t, n_iter), e->exp_pos);
    if (rc) return rc;
  }
  while (done < n_iter) {
    const int64_t S = block_len(e->t, n_iter - done);
    const double* u_blk = e->ublk.p;
    const double* lu_blk = e->lublk.p;
    const double* qz_blk = e->qzblk.p;

    // ---- proposal increments + uniforms ----
    if (split) {
      const int b = (int)(blk & 1);
      if (!inc_ready) {  // (after a covariance swap the previous block's k_chol_apply has produced this block's increments already)
        HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_rng[b], 0));
        ApplyArgs ap{};
        ap.NP = NP;
        ap.S = (int)S;
        ap.Lk = e->Lk.p;
        ap.L_stride = e->L_shared ? 0 : (int64_t)e->DP * e->DP;
        ap.zf = e->zfrag[b].p;
        ap.inc = e->inc.p;
        ScopedTimer tm(e, 0);
        DISPATCH_DPAD(e->DP, launch_apply<DPAD>(ap, e->stream));
      }
      inc_ready = false;
      HIP_TRY(hipEventRecord(e->ev_apply[b], e->stream));
      u_blk = e->ublk2[b].p;
      lu_blk = e->lublk2[b].p;
      qz_blk = e->qzblk2[b].p;
      if (done + S < n_iter) {
        // next block's draws go under THIS block's steps: released by this block's k_apply (the host runs ahead of the
        // GPU; without the wait the draws would start as soon as block blk - 1's steps end, i.e. next to its k_adapt,
        // which is itself bound by the VALU)
        HIP_TRY(hipStreamWaitEvent(e->rng_stream, e->ev_apply[b], 0));
        int rc = enqueue_rng(blk + 1, e->t + S, block_len(e->t + S, n_iter - done - S), e->exp_pos + S);
        if (rc) return rc;
      }
    } else {
    ProposeArgs pa{};
    pa.N = N;
    pa.NP = NP;
    pa.chain_offset = e->cfg.chain_offset;
    pa.d = d;
    pa.S = (int)S;
    pa.step0 = e->t;
    pa.seed = e->cfg.seed;
    pa.Lk = e->Lk.p;
    pa.L_stride = e->L_shared ? 0 : (int64_t)e->DP * e->DP;
    pa.inc = e->inc.p;
    pa.u = e->ublk.p;
    pa.logu = e->lublk.p;
    pa.qz = e->pp.kind == TDA_PROP_INDEPENDENCE ? e->qzblk.p : nullptr;
    if (e->rep_steps) {
      pa.z_replay = e->z_rep.p + (size_t)e->rep_pos * N * d;
      pa.u_replay = e->u_rep.p + (size_t)e->rep_pos * N;
    }
    if (e->exp_steps) {
      pa.z_export = (e->exp_dev ? e->z_exp : e->z_exp_d.p) + (size_t)e->exp_pos * N * d;
      pa.u_export = (e->exp_dev ? e->u_exp : e->u_exp_d.p) + (size_t)e->exp_pos * N;
    }
    {
      ScopedTimer tm(e, 0);
      DISPATCH_DPAD(e->DP, launch_propose<DPAD>(pa, e->stream));
    }
    }

    // ---- fused MH steps ----
    StepArgs sa{};
    fill_level(e, lv, sa);
    sa.S = (int)S;
    sa.mode = MODE_STEP;
    sa.prop_kind = e->pp.kind;
    sa.theta = e->theta.p;
    sa.lp = e->lp.p;
    sa.ll = e->ll.p;
    sa.scaling = e->scaling.p;
    sa.acc_count = e->acc_count.p;
    sa.inc = e->inc.p;
    sa.u = u_blk;
    sa.logu = lu_blk;
    sa.SopT = e->pp.kind == TDA_PROP_OWCN ? e->ow_SopT.p : (e->pp.kind == TDA_PROP_MALA ? e->mala_H.p : nullptr);
    sa.cvec = e->pp.kind == TDA_PROP_MALA ? e->mala_c.p : nullptr;
    sa.grad = e->pp.kind == TDA_PROP_MALA ? e->mala_grad.p : nullptr;
    if (e->pp.kind == TDA_PROP_INDEPENDENCE) {
      sa.q_mean = e->q_mean_d.p;
      sa.qz = qz_blk;
      sa.lq = e->lq.p;
    }
    // records go straight into caller memory when it is device memory; AM needs the states either way
    double* const blk_params = (async_host && (blk & 1)) ? e->rec_params2.p : e->rec_params.p;
    double* const blk_stats = (async_host && (blk & 1)) ? e->rec_stats2.p : e->rec_stats.p;
    uint8_t* const blk_acc = (async_host && (blk & 1)) ? e->rec_acc2.p : e->rec_acc.p;
    if (async_host && blk >= 2) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_cp[blk & 1], 0));  // buffer set free again
    sa.rec_params = (p_dev && thin == 1) ? o_params + (size_t)done * N * d : ((o_params || is_am) ? blk_params : nullptr);
    sa.rec_stats = (s_dev && thin == 1) ? o_stats + (size_t)done * N * 3 : (o_stats ? blk_stats : nullptr);
    sa.rec_acc = (a_dev && thin == 1) ? o_acc + (size_t)done * N : ((o_acc || (e->prog_h && thin > 1)) ? blk_acc : nullptr);
    const bool user_stepwise = lv.model == MODEL_USER && (lv.noise_kind == TDA_NOISE_DENSE || e->pp.kind == TDA_PROP_INDEPENDENCE || e->pp.kind == TDA_PROP_OWCN);
    if (lv.model == MODEL_CALLBACK || user_stepwise) {  // (the fused source-model kernel knows GRW / pCN steps and iso / diag noise)
      ExtArgs xa{};
      fill_ext_args(e, lv, xa);
      xa.mode = 0;
      xa.prop_kind = e->pp.kind;
      xa.theta = sa.theta;
      xa.lp = sa.lp;
      xa.ll = sa.ll;
      xa.scaling = sa.scaling;
      xa.acc_count = sa.acc_count;
      xa.inc = sa.inc;
      xa.u = sa.u;
      xa.rec_params = sa.rec_params;
      xa.rec_stats = sa.rec_stats;
      xa.rec_acc = sa.rec_acc;
      xa.q_mean = sa.q_mean;
      xa.qz = sa.qz;
      xa.lq = sa.lq;
      xa.SopT = sa.SopT;
      for (int s = 0; s < (int)S; ++s) {
        xa.s = s;
        int xrc = ext_step(e, lv, xa);
        if (xrc) return xrc;
      }
    } else if (lv.model == MODEL_USER) {
      UserStepArgs ua{};
      int urc = fill_user_args(e, lv, ua);
      if (urc) return urc;
      ua.S = (int)S;
      ua.mode = 0;
      ua.prop_kind = e->pp.kind;
      ua.theta = sa.theta;
      ua.lp = sa.lp;
      ua.ll = sa.ll;
      ua.scaling = sa.scaling;
      ua.acc_count = sa.acc_count;
      ua.inc = sa.inc;
      ua.u = sa.u;
      ua.rec_params = sa.rec_params;
      ua.rec_stats = sa.rec_stats;
      ua.rec_acc = sa.rec_acc;
      ScopedTimer tm(e, 1);
      if ((urc = launch_user_steps(lv.ufn, ua, e->stream))) return urc;
    } else {
      ScopedTimer tm(e, 1);
      DISPATCH_DPAD(e->DP, launch_steps<DPAD>(sa, NP / 16, lds, e->stream));
    }
    if (split) HIP_TRY(hipEventRecord(e->ev_steps[blk & 1], e->stream));

    // ---- adaptation (proposal.py:228-245, :502-512) ----
    const bool boundary = periodic && ((e->t + S) % period == 0);
    if (is_am || (boundary && adaptive)) {
      AdaptArgs aa{};
      aa.N = N;
      aa.NP = NP;
      aa.d = d;
      aa.S = (int)S;
      aa.t_base = e->t;
      aa.do_am = is_am;
      aa.boundary = boundary;
      aa.do_scale = adaptive;
      const bool do_swap = is_am && boundary && (e->t + S >= e->pp.t0);
      aa.do_swap = do_swap;
      aa.block_moments = e->pp.block_moments != 0;
      aa.period = period;
      aa.gamma_pow = std::pow(e->pp.gamma, -(double)e->k_adapt);
      aa.sd = e->am_sd;
      aa.alpha_star = e->pp.kind == TDA_PROP_MALA ? 0.57 : 0.24;  // proposal.py:899 / :169
      aa.eps = e->pp.epsilon;
      aa.rec_params = sa.rec_params;
      aa.am_mu = e->am_mu.p;
      aa.am_sigma = e->am_sigma.p;
      aa.scaling = e->scaling.p;
      aa.acc_count = e->acc_count.p;
      aa.flags = e->flags.p;
      ScopedTimer tm(e, 2);
      DISPATCH_DPAD(e->DP, launch_adapt<DPAD>(aa, e->stream));
      if (do_swap) {  // C <- Sigma (proposal.py:509-510)
        CholArgs ca{};
        ca.N = N;
        ca.d = d;
        ca.am_sigma = e->am_sigma.p;
        ca.Lk = e->Lk.p;
        ca.flags = e->flags.p;
        static const bool fuse_ok = !(getenv("TINYDA_FUSE_CHOL_APPLY") && atoi(getenv("TINYDA_FUSE_CHOL_APPLY")) == 0);  // A/B switch
        if (split && fuse_ok && done + S < n_iter) {
          // ... and the next block's increments from the new factor in the same launch (k_chol_apply); that block's normals were
          // drawn under this block's steps
          const int nb = (int)((blk + 1) & 1);
          HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_rng[nb], 0));
          ApplyArgs ap{};
          ap.NP = NP;
          ap.S = (int)block_len(e->t + S, n_iter - done - S);
          ap.Lk = e->Lk.p;
          ap.L_stride = (int64_t)e->DP * e->DP;
          ap.zf = e->zfrag[nb].p;
          ap.inc = e->inc.p;
          DISPATCH_DPAD(e->DP, launch_chol_apply<DPAD>(ca, ap, e->stream));
          inc_ready = true;
        } else {
          DISPATCH_DPAD(e->DP, launch_chol<DPAD>(ca, e->stream));
        }
      }
    } else if (boundary) {
      HIP_TRY(hipMemsetAsync(e->acc_count.p, 0, NP * sizeof(int32_t), e->stream));
    }
    HIP_TRY(hipGetLastError());
    if (boundary && adaptive) e->k_adapt += 1;

    // ---- progress mark; thinned / host-side records ----
    int rc;
    if ((rc = progress_mark(e, S, sa.rec_acc, S * N))) return rc;
    if (thin > 1) {
      // kept iterations of this block: s with (t + s + 1) % thin == 0
      const int64_t s0 = (thin - 1 - (e->t % thin)) % thin, nkeep = s0 < S ? (S - s0 + thin - 1) / thin : 0;
      hipStream_t cs = e->stream;
      if (async_host) {
        HIP_TRY(hipEventRecord(e->ev_rec[blk & 1], e->stream));
        HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->ev_rec[blk & 1], 0));
        cs = e->copy_stream;
      }
      // (device destinations are gathered on the main stream, host ones copied row by row on the copy stream when page-locked)
      if ((rc = thin_out(e, o_params ? o_params + (size_t)out_row * N * d : nullptr, p_dev, blk_params, (size_t)N * d * sizeof(double), s0, nkeep, p_dev ? e->stream : cs)) ||
          (rc = thin_out(e, o_stats ? o_stats + (size_t)out_row * N * 3 : nullptr, s_dev, blk_stats, (size_t)N * 3 * sizeof(double), s0, nkeep, s_dev ? e->stream : cs)) ||
          (rc = thin_out(e, o_acc ? o_acc + (size_t)out_row * N : nullptr, a_dev, blk_acc, (size_t)N, s0, nkeep, a_dev ? e->stream : cs)))
        return rc;
      if (async_host) HIP_TRY(hipEventRecord(e->ev_cp[blk & 1], e->copy_stream));
      else if (any_host) HIP_TRY(hipStreamSynchronize(e->stream));  // block buffers are reused next iteration
      out_row += nkeep;
    } else if (async_host) {
      const int i = (int)(blk & 1);
      HIP_TRY(hipEventRecord(e->ev_rec[i], e->stream));
      HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->ev_rec[i], 0));
      if (o_params && !p_dev)
        HIP_TRY(hipMemcpyAsync(o_params + (size_t)done * N * d, blk_params, (size_t)S * N * d * sizeof(double), hipMemcpyDeviceToHost, e->copy_stream));
      if (o_stats && !s_dev)
        HIP_TRY(hipMemcpyAsync(o_stats + (size_t)done * N * 3, blk_stats, (size_t)S * N * 3 * sizeof(double), hipMemcpyDeviceToHost, e->copy_stream));
      if (o_acc && !a_dev)
        HIP_TRY(hipMemcpyAsync(o_acc + (size_t)done * N, blk_acc, (size_t)S * N, hipMemcpyDeviceToHost, e->copy_stream));
      HIP_TRY(hipEventRecord(e->ev_cp[i], e->copy_stream));
    } else {
      if (o_params && !p_dev) {
        if ((rc = copy_out(e, o_params + (size_t)done * N * d, e->rec_params.p, (size_t)S * N * d * sizeof(double)))) return rc;
        host_copies = true;
      }
      if (o_stats && !s_dev) {
        if ((rc = copy_out(e, o_stats + (size_t)done * N * 3, e->rec_stats.p, (size_t)S * N * 3 * sizeof(double)))) return rc;
        host_copies = true;
      }
      if (o_acc && !a_dev) {
        if ((rc = copy_out(e, o_acc + (size_t)done * N, e->rec_acc.p, (size_t)S * N))) return rc;
        host_copies = true;
      }
      if (host_copies) HIP_TRY(hipStreamSynchronize(e->stream));  // block buffers are reused next iteration
    }

    e->t += S;
    done += S;
    blk += 1;
    if (e->rep_steps) e->rep_pos += S;
    if (e->exp_steps) e->exp_pos += S;
  }

  if (async_host) {  // host outputs: run() returns with the records in place
    HIP_TRY(hipStreamSynchronize(e->copy_stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  if (e->exp_steps && !e->exp_dev) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->z_exp, e->z_exp_d.p, (size_t)e->exp_pos * N * d * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(e->u_exp, e->u_exp_d.p, (size_t)e->exp_pos * N * sizeof(double), hipMemcpyDeviceToHost));
  }
  return TDA_OK;
}

// One block of S base steps of a hierarchy whose levels live outside the fused level kernel (run_multilevel): batched host
// callbacks, source-defined models, linear levels beside them.  `ma` carries the block's buffers (increments, uniforms,
// record rows per level, replay pointers) exactly as the fused kernel would get them.
static int run_ext_hierarchy_block(tda_engine* e, const MLArgs& ma, int64_t S, bool adaptive) {
  const int nl = e->nlev, d = e->d, DP = e->DP;
  const int64_t N = e->N, NP = e->NP;
  // every base step is propose -> model(level 0) -> accept; when the subchain of level k completes, level k + 1's model is
  // evaluated at the states of level k (one evaluation for all chains) and k_ext_level_action decides, aligns and records
  // (the cascade of k_ml_steps, one level at a time)
  const unsigned grid = (unsigned)((N + EXT_WAVES - 1) / EXT_WAVES);
  int cc[MAXLEV];
  int64_t row[MAXLEV] = {0, 0, 0, 0};
  int64_t rp = e->ring_pos;  // position in the base proposal's accepted list: one entry per base step and per level action
  for (int k = 0; k < MAXLEV; ++k) cc[k] = e->cnt[k];
  // diagonal error model over a hierarchy of linear levels: base subchains in the fused level kernel (TINYDA_AEMD_FUSED=0: one
  // propose / outputs / accept triple per base step, for A/B measurements); the residual tile of level 0 must fit into LDS
  static const bool aemd_fused_ok = !(getenv("TINYDA_AEMD_FUSED") && atoi(getenv("TINYDA_AEMD_FUSED")) == 0);
  bool all_linear = true;
  for (int k = 0; k < nl; ++k) all_linear = all_linear && e->levels[k].model == MODEL_LINEAR && e->levels[k].Apk.p != nullptr;
  // (LDS of k_ml_steps with only the base level staged: proposal tile, reduction slabs, level 0's data [and weights], prior rows,
  // the residual tile; lds_y[0] = 0, so the base level's slab is the first of the staging region)
  const int stage0 = e->levels[0].m_pad * (e->levels[0].noise_kind == TDA_NOISE_DIAG ? 2 : 1);
  const size_t aemd_lds = ((size_t)16 * (DP + 2) + 128 + stage0 + (e->prior_kind == PRIOR_DENSE ? e->prior_ncb * 16 : 0) +
                           16 * (e->levels[0].m_pad + 2) + 16) * sizeof(double);
  const bool aemd_fused = aemd_fused_ok && e->aem == TDA_AEM_STATE_INDEPENDENT_DIAGONAL && all_linear && !e->randomize && !e->is_dreamz &&
                          aemd_lds <= 160 * 1024;
  bool fused_base_ran = false;
  for (int64_t s = 0; s < S;) {
    if (e->randomize && cc[0] == 0) {  // Delayed Acceptance: draw the promoted index of the subchain that starts now
      hipLaunchKernelGGL(k_ext_pick, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, e->stream, (long long)N, e->sl[0],
                         (unsigned long long)e->cfg.seed, (long long)e->cfg.chain_offset, (long long)(e->done[1] + row[1]),
                         ma.ridx_rep ? ma.ridx_rep + (size_t)row[1] * N : nullptr, e->ml_pick.p);
    }
    if (e->levels[0].model == MODEL_USER && !e->aem && !e->randomize && !e->is_dreamz && e->levels[0].noise_kind != TDA_NOISE_DENSE) {
      // source-defined base level: the rest of the running subchain (inside this block) is ONE launch of the fused
      // step kernel compiled with the model
      const int64_t n = std::min<int64_t>(S - s, (int64_t)e->sl[0] - cc[0]);
      UserStepArgs ua{};
      int urc = fill_user_args(e, e->levels[0], ua);
      if (urc) return urc;
      ua.S = (int)n;
      ua.mode = 0;
      ua.prop_kind = e->pp.kind;
      ua.theta = e->ml_theta.p;
      ua.lp = e->ml_lp.p;
      ua.ll = e->ml_ll.p;
      ua.scaling = e->scaling.p;
      ua.acc_count = nullptr;
      ua.inc = e->inc.p + (size_t)s * NP * DP;
      ua.u = e->ublk.p + (size_t)s * NP;
      ua.rec_params = ma.rec_params[0] ? ma.rec_params[0] + (size_t)s * N * d : nullptr;
      ua.rec_stats = ma.rec_stats[0] ? ma.rec_stats[0] + (size_t)s * N * 3 : nullptr;
      ua.rec_acc = ma.rec_acc[0] ? ma.rec_acc[0] + (size_t)s * N : nullptr;
      ua.anyacc = e->ml_anyacc.p;
      ua.ring = adaptive ? e->ml_ring.p : nullptr;
      ua.ring_P = e->ring_P;
      ua.ring_pos = rp;
      if ((urc = launch_user_steps(e->levels[0].ufn, ua, e->stream))) return urc;
      rp += n;
      s += n;
      cc[0] += (int)n;
    } else if (aemd_fused) {
      // diagonal error model over linear levels: the rest of the running base subchain (inside this block) is ONE launch of the
      // fused level kernel with the corrected likelihood (k_ml_steps, aem_on = 2: bias and inverse variances of chain c from
      // the arrays the level actions maintain) instead of propose / outputs / accept launches per base step
      const int64_t n = std::min<int64_t>(S - s, (int64_t)e->sl[0] - cc[0]);
      MLArgs mb = ma;
      mb.S = (int)n;
      mb.cascade = 0;
      mb.randomize = 0;
      mb.aem_on = 2;
      mb.aem_mp = e->levels[0].m_pad;
      mb.aem_ld = e->levels[0].m;
      mb.aem_bias = e->aemd_bias[0].p;
      mb.aem_P = e->aemd_w[0].p;
      mb.lds_total = stage0;
      mb.inc = e->inc.p + (size_t)s * NP * DP;
      mb.u0 = e->ublk.p + (size_t)s * NP;
      mb.logu0 = nullptr;
      mb.cnt[0] = cc[0];
      mb.done[0] = e->done[0] + s;
      mb.ring_pos = rp;
      mb.rec_params[0] = ma.rec_params[0] ? ma.rec_params[0] + (size_t)s * N * d : nullptr;
      mb.rec_stats[0] = ma.rec_stats[0] ? ma.rec_stats[0] + (size_t)s * N * 3 : nullptr;
      mb.rec_acc[0] = ma.rec_acc[0] ? ma.rec_acc[0] + (size_t)s * N : nullptr;
      int lrc = TDA_OK;
      DISPATCH_DPAD(DP, lrc = launch_ml<DPAD>(mb, NP / 16, aemd_lds, e->stream));
      if (lrc) return lrc;
      HIP_TRY(hipGetLastError());
      fused_base_ran = true;
      rp += n;
      s += n;
      cc[0] += (int)n;
    } else if (e->aem == TDA_AEM_STATE_INDEPENDENT_DIAGONAL) {
      // base step under the corrected diagonal likelihood (per-chain bias and inverse variances, tda_kernels_aemd.h)
      const Level& l0 = e->levels[0];
      ExtArgs pa2{};
      fill_ext_args(e, l0, pa2);
      pa2.mode = 0;
      pa2.prop_kind = e->pp.kind;
      pa2.theta = e->ml_theta.p;
      pa2.scaling = e->scaling.p;
      pa2.inc = e->inc.p;
      pa2.s = (int)s;
      hipLaunchKernelGGL(k_ext_propose, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, pa2);
      int mrc0 = ext_model_outputs(e, l0);
      if (mrc0) return mrc0;
      AemdArgs da{};
      fill_aemd_args(e, da);
      da.s = (int)s;
      da.F = l0.cb_F.p;
      da.prop = l0.cb_prop.p;
      da.u0 = e->ublk.p;
      da.sid_value = e->done[0] + s + 1;
      da.ring = adaptive ? e->ml_ring.p : nullptr;
      da.ring_P = e->ring_P;
      da.ring_pos = rp++;
      da.rec_params = ma.rec_params[0];
      da.rec_stats = ma.rec_stats[0];
      da.rec_acc = ma.rec_acc[0];
      hipLaunchKernelGGL(k_aemd_accept, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, da);
      HIP_TRY(hipGetLastError());
      cc[0] += 1;
      s += 1;
    } else if (e->aem) {
      // base step under the bias-corrected likelihood of the error model (per-chain bias and inverse)
      const Level& l0 = e->levels[0];
      ExtArgs pa2{};
      fill_ext_args(e, l0, pa2);
      pa2.mode = 0;
      pa2.prop_kind = e->pp.kind;
      pa2.theta = e->ml_theta.p;
      pa2.scaling = e->scaling.p;
      pa2.inc = e->inc.p;
      pa2.s = (int)s;
      hipLaunchKernelGGL(k_ext_propose, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, pa2);
      int mrc0 = ext_model_outputs(e, l0);
      if (mrc0) return mrc0;
      ExtAemAcceptArgs ea{};
      ea.N = N;
      ea.NP = NP;
      ea.d = d;
      ea.DP = DP;
      ea.m = e->aem_m;
      ea.MP = e->aem_ld;
      ea.s = (int)s;
      ea.prop_kind = e->pp.kind;
      ea.theta = e->ml_theta.p;
      ea.lp = e->ml_lp.p;
      ea.ll = e->ml_ll.p;
      ea.u = e->ublk.p;
      ea.prop = l0.cb_prop.p;
      ea.F = l0.cb_F.p;
      ea.data = l0.data64.p;
      ea.bias = e->aem_bias[0].p;
      ea.P = e->aem_covinv[0].p;
      ea.Fcur = e->ext_Fcur[0].p;
      ea.pr_mean = e->prior_mean.p;
      ea.pr_pinv = e->prior_pinv.p;
      ea.pr_lo = e->prior_bounded ? e->prior_lo.p : nullptr;
      ea.pr_hi = e->prior_bounded ? e->prior_hi.p : nullptr;
      ea.logconst = e->prior_logconst;
      ea.anyacc = e->ml_anyacc.p;
      ea.sid = e->ml_sid.p;
      ea.sid_value = e->done[0] + s + 1;
      ea.ring = adaptive ? e->ml_ring.p : nullptr;
      ea.ring_P = e->ring_P;
      ea.ring_pos = rp++;
      ea.rec_params = ma.rec_params[0];
      ea.rec_stats = ma.rec_stats[0];
      ea.rec_acc = ma.rec_acc[0];
      if (e->aem_ld == 64) hipLaunchKernelGGL(k_ext_aem_accept<64>, dim3((unsigned)N), dim3(64), 0, e->stream, ea);
      else hipLaunchKernelGGL(k_ext_aem_accept<128>, dim3((unsigned)N), dim3(128), 0, e->stream, ea);
      HIP_TRY(hipGetLastError());
      cc[0] += 1;
      s += 1;
    } else {
    ExtArgs xa{};
    fill_ext_args(e, e->levels[0], xa);
    xa.mode = 0;
    xa.prop_kind = e->pp.kind;
    xa.theta = e->ml_theta.p;
    xa.lp = e->ml_lp.p;
    xa.ll = e->ml_ll.p;
    xa.scaling = e->scaling.p;
    xa.acc_count = nullptr;
    xa.anyacc = e->ml_anyacc.p;
    xa.inc = e->inc.p;
    xa.u = e->ublk.p;
    xa.s = (int)s;
    xa.rec_params = ma.rec_params[0];
    xa.rec_stats = ma.rec_stats[0];
    xa.rec_acc = ma.rec_acc[0];
    xa.ring = adaptive ? e->ml_ring.p : nullptr;
    xa.ring_P = e->ring_P;
    xa.ring_pos = rp++;
    if (e->randomize) {
      xa.pick = e->ml_pick.p;
      xa.cnt = cc[0];
      xa.ysnap = e->ml_ysnap.p;
    }
    if (e->is_dreamz) {
      // DREAMZ base step (proposal.py:811-852 / :790-809): the block's draws are in dz_coef / dz_epsm / dz_ridx (k_dreamz_draw);
      // apply the jump from the chain's archive, evaluate, decide, append the state the chain is left in to its archive
      if (s == S - 1)  // jumping distance of the crossover adaptation: the state before the block's last base step
        HIP_TRY(hipMemcpyAsync(e->theta_prev.p, e->ml_theta.p, (size_t)NP * DP * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      DzExtArgs za{};
      za.N = N;
      za.NP = NP;
      za.d = d;
      za.DP = DP;
      za.delta = e->dz.delta;
      za.s = (int)s;
      za.shared = 0;
      za.jump_ready = 0;
      za.M_base = e->arch_rows;
      za.cap = e->arch_cap;
      za.arch = e->arch.p;
      za.theta = e->ml_theta.p;
      za.coef = e->dz_coef.p;
      za.epsm = e->dz_epsm.p;
      za.ridx = e->dz_ridx.p;
      za.prop = e->levels[0].cb_prop.p;
      za.blk_states = nullptr;
      hipLaunchKernelGGL(k_dz_ext_propose, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, za);
      int mrc = ext_model_outputs(e, e->levels[0]);
      if (mrc) return mrc;
      xa.prop_kind = TDA_PROP_GRW;  // acceptance on the posterior ratio (proposal.py:253-258)
      hipLaunchKernelGGL(k_ext_accept, dim3(grid), dim3(64 * EXT_WAVES), xa.Pd ? (size_t)EXT_WAVES * e->levels[0].m * sizeof(double) : 0,
                         e->stream, xa);
      hipLaunchKernelGGL(k_dz_ext_append, dim3((unsigned)((NP + EXT_WAVES - 1) / EXT_WAVES)), dim3(64 * EXT_WAVES), 0, e->stream, za);
      if (s == S - 1)  // ... and the state right after it, before an upper level realigns the chain
        HIP_TRY(hipMemcpyAsync(e->theta_last.p, e->ml_theta.p, (size_t)NP * DP * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
      HIP_TRY(hipGetLastError());
    } else {
    int xrc = ext_step(e, e->levels[0], xa);
    if (xrc) return xrc;
    }
    cc[0] += 1;
    s += 1;
    }
    if (fused_base_ran && cc[0] == e->sl[0]) {
      // the level actions keep the model output of every level's current link; the fused base kernel does not: F_0(theta_0)
      // for all chains by the same product that evaluates linear levels anywhere else on this path
      linear_outputs_at(e, e->levels[0], e->ml_theta.p, DP, e->aemd_F[0].p);
      fused_base_ran = false;
    }
    for (int k = 0; k < nl - 1 && cc[k] == e->sl[k]; ++k) {
      const int q = k + 1;
      const Level& lq = e->levels[q];
      if (lq.model == MODEL_USER && !e->aem && lq.noise_kind != TDA_NOISE_DENSE) {
        // source-defined level without error model: evaluation, decision, alignment and records in one launch of the
        // kernel compiled with the model (+ the step's uniforms)
        double* ul = e->lublk.p;  // scratch of the single-level path, at least SMAX * NP doubles
        hipLaunchKernelGGL(k_ext_level_uniforms, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, e->stream, (long long)N,
                           (unsigned long long)e->cfg.seed, (long long)e->cfg.chain_offset, (long long)(e->done[q] + row[q]), q,
                           ma.u_rep[q] ? ma.u_rep[q] + (size_t)row[q] * N : nullptr, ul);
        UserLevelArgs ua{};
        ua.N = N;
        ua.NP = NP;
        ua.d = d;
        ua.DP = DP;
        ua.m = lq.m;
        ua.nlev = nl;
        ua.q = q;
        ua.data = lq.udata.p;
        ua.w = lq.noise_kind == TDA_NOISE_DIAG ? lq.uw.p : nullptr;
        ua.var = lq.var;
        ua.theta = e->ml_theta.p;
        ua.lp = e->ml_lp.p;
        ua.ll = e->ml_ll.p;
        ua.Sst = e->ml_S.p;
        ua.anyacc = e->ml_anyacc.p;
        ua.u = ul;
        ua.rec_params = ma.rec_params[q] ? ma.rec_params[q] + (size_t)row[q] * N * d : nullptr;
        ua.rec_stats = ma.rec_stats[q] ? ma.rec_stats[q] + (size_t)row[q] * N * 3 : nullptr;
        ua.rec_acc = ma.rec_acc[q] ? ma.rec_acc[q] + (size_t)row[q] * N : nullptr;
        ua.ring = adaptive ? e->ml_ring.p : nullptr;
        ua.ring_P = e->ring_P;
        ua.ring_pos = rp++;
        ua.ysnap = (e->randomize && k == 0) ? e->ml_ysnap.p : nullptr;
        int lrc = launch_user_level(lq.ufn_level, ua, e->stream);
        if (lrc) return lrc;
        cc[k] = 0;
        cc[q] += 1;
        row[q] += 1;
        continue;
      }
      const bool snap = e->randomize && k == 0;
      if (lq.model == MODEL_LINEAR && lq.Apk.p) {
        // linear level: its outputs at the states of level k straight from the state array (no copy of the states first)
        linear_outputs_at(e, lq, snap ? e->ml_ysnap.p : e->ml_theta.p + (size_t)k * NP * DP, snap ? DP + 2 : DP, lq.cb_F.p);
      } else {
        ExtArgs ya{};
        fill_ext_args(e, lq, ya);
        ya.mode = 1;  // "proposals" = the current states of level k (the promoted states of a randomised subchain)
        ya.theta = snap ? e->ml_ysnap.p : e->ml_theta.p + (size_t)k * NP * DP;
        ya.theta_ld = snap ? DP + 2 : 0;
        hipLaunchKernelGGL(k_ext_propose, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, ya);
        const int mrc = ext_model_outputs(e, lq);
        if (mrc) return mrc;
      }
      if (e->aem == TDA_AEM_STATE_INDEPENDENT_DIAGONAL) {  // decision, alignment, tracker, bias, update_link: one launch
        AemdArgs da{};
        fill_aemd_args(e, da);
        da.q = q;
        da.step = e->done[q] + row[q];
        da.F = lq.cb_F.p;
        da.b_t = e->aem_bt[q];
        da.u_rep = ma.u_rep[q] ? ma.u_rep[q] + (size_t)row[q] * N : nullptr;
        da.ring = adaptive ? e->ml_ring.p : nullptr;
        da.ring_P = e->ring_P;
        da.ring_pos = rp++;
        da.rec_params = ma.rec_params[q] ? ma.rec_params[q] + (size_t)row[q] * N * d : nullptr;
        da.rec_stats = ma.rec_stats[q] ? ma.rec_stats[q] + (size_t)row[q] * N * 3 : nullptr;
        da.rec_acc = ma.rec_acc[q] ? ma.rec_acc[q] + (size_t)row[q] * N : nullptr;
        hipLaunchKernelGGL(k_aemd_action, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, da);
        HIP_TRY(hipGetLastError());
        e->aem_bt[q] += 1;
        cc[k] = 0;
        cc[q] += 1;
        row[q] += 1;
        continue;
      }
      if (e->aem) {
        ExtAemArgs ga{};
        ga.N = N;
        ga.NP = NP;
        ga.chain_offset = e->cfg.chain_offset;
        ga.d = d;
        ga.DP = DP;
        ga.m = e->aem_m;
        ga.MP = e->aem_ld;
        ga.nlev = nl;
        ga.q = q;
        ga.is_da = nl == 2;
        ga.dependent = e->aem == TDA_AEM_STATE_DEPENDENT;
        ga.prop_kind = e->pp.kind;
        ga.pr_W = e->prior_W_rm.p;
        ga.pr_logdet = e->prior_logconst - d * std::log(2.0 * M_PI);
        ga.scaling = e->scaling.p;
        ga.seed = e->cfg.seed;
        ga.step = e->done[q] + row[q];
        ga.Fnew = lq.cb_F.p;
        for (int k2 = 0; k2 < nl; ++k2) {
          ga.data[k2] = e->levels[k2].data64.p;
          ga.Fcur[k2] = e->ext_Fcur[k2].p;
          ga.bias_tot[k2] = e->aem_bias[k2].p;
          ga.cov_inv[k2] = e->aem_covinv[k2].p;
          ga.b_mu[k2] = e->aem_bmu[k2].p;
          ga.b_sig[k2] = e->aem_bsig[k2].p;
          ga.mdiff[k2] = e->aem_mdiff[k2].p;
        }
        ga.var_finest = e->levels[nl - 1].var;
        ga.Fst = e->ext_Fst.p;
        ga.theta = e->ml_theta.p;
        ga.lp = e->ml_lp.p;
        ga.ll = e->ml_ll.p;
        ga.Sst = e->ml_S.p;
        ga.anyacc = e->ml_anyacc.p;
        ga.sid = e->ml_sid.p;
        ga.b_t = e->aem_bt[q];
        ga.u_rep = ma.u_rep[q] ? ma.u_rep[q] + (size_t)row[q] * N : nullptr;
        ga.ring = adaptive ? e->ml_ring.p : nullptr;
        ga.ring_P = e->ring_P;
        ga.ring_pos = rp++;
        ga.rec_params = ma.rec_params[q] ? ma.rec_params[q] + (size_t)row[q] * N * d : nullptr;
        ga.rec_stats = ma.rec_stats[q] ? ma.rec_stats[q] + (size_t)row[q] * N * 3 : nullptr;
        ga.rec_acc = ma.rec_acc[q] ? ma.rec_acc[q] + (size_t)row[q] * N : nullptr;
        AemInvArgs iv{};
        iv.N = N;
        iv.m = e->aem_m;
        iv.MP = e->aem_ld;
        iv.nb = (e->aem_m + 15) / 16;
        iv.cov = e->levels[k].cov64.p;
        if (ga.dependent) {
          iv.nsum = 1;
          iv.sig[0] = e->aem_bsig[q].p;
        } else {
          iv.nsum = nl - q;
          for (int p2 = q; p2 < nl; ++p2) iv.sig[p2 - q] = e->aem_bsig[p2].p;
        }
        iv.P = e->aem_covinv[k].p;
        const size_t inv_lds = aem_inverse_lds_bytes(iv.nb);
        if (inv_lds > 64 * 1024)
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aem_inverse<0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)inv_lds));
        for (int phase = 0; phase < 2; ++phase) {
          ga.phase = phase;
          if (e->aem_ld == 64) hipLaunchKernelGGL(k_ext_aem_action<64>, dim3((unsigned)N), dim3(64), 0, e->stream, ga);
          else hipLaunchKernelGGL(k_ext_aem_action<128>, dim3((unsigned)N), dim3(128), 0, e->stream, ga);
          if (phase == 0) {
            if (iv.nb > 4) hipLaunchKernelGGL((k_aem_inverse<0, 8>), dim3((unsigned)N), dim3(512), inv_lds, e->stream, iv);
            else hipLaunchKernelGGL((k_aem_inverse<0, 4>), dim3((unsigned)N), dim3(256), inv_lds, e->stream, iv);
          }
        }
        HIP_TRY(hipGetLastError());
        e->aem_bt[q] += 1;
        cc[k] = 0;
        cc[q] += 1;
        row[q] += 1;
        continue;
      }
      ExtLevelArgs la{};
      la.N = N;
      la.NP = NP;
      la.chain_offset = e->cfg.chain_offset;
      la.d = d;
      la.DP = DP;
      la.m = lq.m;
      la.nlev = nl;
      la.q = q;
      la.seed = e->cfg.seed;
      la.step = e->done[q] + row[q];
      la.F = lq.cb_F.p;
      la.data = lq.udata.p;
      la.w = lq.noise_kind == TDA_NOISE_DIAG ? lq.uw.p : nullptr;
      la.Pd = lq.noise_kind == TDA_NOISE_DENSE ? lq.Pd.p : nullptr;
      la.var = lq.var;
      la.theta = e->ml_theta.p;
      la.lp = e->ml_lp.p;
      la.ll = e->ml_ll.p;
      la.Sst = e->ml_S.p;
      la.anyacc = e->ml_anyacc.p;
      la.u_rep = ma.u_rep[q] ? ma.u_rep[q] + (size_t)row[q] * N : nullptr;
      la.rec_params = ma.rec_params[q] ? ma.rec_params[q] + (size_t)row[q] * N * d : nullptr;
      la.rec_stats = ma.rec_stats[q] ? ma.rec_stats[q] + (size_t)row[q] * N * 3 : nullptr;
      la.rec_acc = ma.rec_acc[q] ? ma.rec_acc[q] + (size_t)row[q] * N : nullptr;
      la.ring = adaptive ? e->ml_ring.p : nullptr;
      la.ring_P = e->ring_P;
      la.ring_pos = rp++;
      la.ysnap = (e->randomize && k == 0) ? e->ml_ysnap.p : nullptr;
      hipLaunchKernelGGL(k_ext_level_action, dim3(grid), dim3(64 * EXT_WAVES), la.Pd ? (size_t)EXT_WAVES * lq.m * sizeof(double) : 0, e->stream, la);
      HIP_TRY(hipGetLastError());
      cc[k] = 0;
      cc[q] += 1;
      row[q] += 1;
    }
  }
  return TDA_OK;
}

static int run_multilevel(tda_engine* e, int64_t n_fine, const tda_outputs* outs) {
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int nl = e->nlev, d = e->d, DP = e->DP;
  const int64_t N = e->N, NP = e->NP;
  if (outs)
    for (int k = 0; k < nl; ++k)
      if (outs[k].struct_size != sizeof(tda_outputs)) return fail(TDA_ERR_INVALID, "tda_outputs[%d].struct_size mismatch", k);
  int64_t mult[MAXLEV];  // local steps of level k per finest step
  mult[nl - 1] = 1;
  for (int k = nl - 2; k >= 0; --k) mult[k] = mult[k + 1] * e->sl[k];
  const int64_t total_base = n_fine * mult[0];
  if (outs)
    for (int k = 0; k < nl; ++k)
      if (int crc = check_out_capacity(outs + k, k, n_fine * mult[k], N, d)) return crc;
  if (e->rep_steps && e->rep_pos + total_base > e->rep_steps) return fail(TDA_ERR_INVALID, "replay buffer too short");
  if (e->exp_steps && e->exp_pos + total_base > e->exp_steps) return fail(TDA_ERR_INVALID, "export buffer too small");
  if (e->is_dreamz) {
    if (e->rp_steps && e->rp_pos + total_base > e->rp_steps) return fail(TDA_ERR_INVALID, "DREAMZ replay buffer too short");
    if (e->arch_rows + total_base > e->arch_cap) return fail(TDA_ERR_INVALID, "archive capacity (%lld rows) exceeded", (long long)e->arch_cap);
  }
  for (int k = 1; k < nl; ++k)
    if (e->u_rep_lv_n[k] && e->u_rep_lv_pos[k] + n_fine * mult[k] > e->u_rep_lv_n[k])
      return fail(TDA_ERR_INVALID, "replay buffer of level %d too short", k);
  if (e->rep_steps || (e->is_dreamz && e->rp_steps))
    for (int k = 1; k < nl; ++k)
      if (!e->u_rep_lv_n[k]) return fail(TDA_ERR_STATE, "replay mode needs uniforms for every level (set_replay_level)");

  const bool is_am = e->pp.kind == TDA_PROP_AM, adaptive = e->pp.adaptive != 0, periodic = is_am || adaptive;
  const int period = e->pp.period;
  // LDS staging layout of the per-level data vectors
  int lds_y[MAXLEV] = {0, 0, 0, 0}, lds_w[MAXLEV] = {0, 0, 0, 0}, off = 0;
  for (int k = 0; k < nl; ++k) {
    lds_y[k] = off;
    off += e->levels[k].m_pad;
    if (e->levels[k].noise_kind == TDA_NOISE_DIAG) {
      lds_w[k] = off;
      off += e->levels[k].m_pad;
    }
  }
  const int prow = e->prior_kind == PRIOR_DENSE ? e->prior_ncb * 16 : 0;
  const size_t lds = ((size_t)16 * (DP + 2) + 128 + off + prow + (e->aem ? 16 * (e->levels[0].m_pad + 2) + 16 : 0)) * sizeof(double);
  if (!e->ext_hier && lds > 160 * 1024)  // (host-sequenced hierarchies stage nothing: their level kernels stream the model outputs)
    return fail(TDA_ERR_UNSUPPORTED, "levels need %zu bytes of LDS staging", lds);

  if (e->profiling) {
    for (auto& t : e->timed) {
      (void)hipEventDestroy(t.a);
      (void)hipEventDestroy(t.b);
    }
    e->timed.clear();
  }
  int64_t rows_out[MAXLEV] = {0, 0, 0, 0};  // rows already written to the caller's buffers, per level
  // host outputs in pinned memory: copies on a second stream under the next block (see tda_engine_run)
  bool any_host = false, all_pinned = true;
  if (outs)
    for (int k = 0; k < nl; ++k) {
      const void* hp[3] = {outs[k].params, outs[k].stats, outs[k].accepted};
      for (const void* q : hp)
        if (q && !is_device_ptr(q)) {
          any_host = true;
          all_pinned = all_pinned && is_pinned_host_ptr(q);
        }
    }
  const bool async_host = any_host && all_pinned;
  if (async_host) {
    if (!e->copy_stream) {
      HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
      for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&e->ev_rec[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_cp[i], hipEventDisableTiming));
      }
    }
    for (int k = 0; k < nl; ++k)
      if (!e->ml_rec_params2[k].p) {
        int rc;
        if ((rc = e->ml_rec_params2[k].alloc((size_t)e->SMAX * N * d)) || (rc = e->ml_rec_stats2[k].alloc((size_t)e->SMAX * N * 3)) ||
            (rc = e->ml_rec_acc2[k].alloc((size_t)e->SMAX * N)))
          return rc;
      }
  }
  // (The draws of a block could run on a second stream under the previous block's steps as in tda_engine_run, but the level
  // kernels use the whole register file of a CU -- 2 x 256 or 1 x 512 per SIMD -- so k_rng would not be co-resident: measured
  // on C3, the split k_rng + k_apply pipeline made the run 8 % SLOWER than the fused k_propose, which therefore stays.)
  int64_t blk_ix = 0;
  int64_t done_base = 0;
  while (done_base < total_base) {
    int64_t S = std::min<int64_t>(total_base - done_base, e->SMAX);
    if (periodic) S = std::min<int64_t>(S, period - (e->t % period));
    if (e->aem && !e->ext_hier) S = std::min<int64_t>(S, e->sl[0] - e->cnt[0]);  // level actions sequenced after the block: stop when the base subchain completes (host-sequenced hierarchies run their actions inside the block)
    // how many local steps each level completes inside this block (uniform schedule)
    int c2[MAXLEV];
    int64_t nblk[MAXLEV] = {S, 0, 0, 0};
    for (int k = 0; k < MAXLEV; ++k) c2[k] = e->cnt[k];
    int64_t tail_appends = 0;  // alignment entries appended by the block's last base step
    for (int64_t s = 0; s < S; ++s) {
      c2[0] += 1;
      for (int k = 0; k < nl - 1 && c2[k] == e->sl[k]; ++k) {
        c2[k] = 0;
        nblk[k + 1] += 1;
        c2[k + 1] += 1;
        if (s == S - 1) tail_appends += 1;
      }
    }

    ProposeArgs pa{};
    pa.N = N;
    pa.NP = NP;
    pa.chain_offset = e->cfg.chain_offset;
    pa.d = d;
    pa.S = (int)S;
    pa.step0 = e->t;
    pa.seed = e->cfg.seed;
    pa.Lk = e->Lk.p;
    pa.L_stride = e->L_shared ? 0 : (int64_t)DP * DP;
    pa.inc = e->inc.p;
    pa.u = e->ublk.p;
    pa.logu = e->ext_hier ? nullptr : e->lublk.p;  // (the host-sequenced path uses that buffer as scratch)
    if (e->rep_steps) {
      pa.z_replay = e->z_rep.p + (size_t)e->rep_pos * N * d;
      pa.u_replay = e->u_rep.p + (size_t)e->rep_pos * N;
    }
    if (e->exp_steps) {
      pa.z_export = (e->exp_dev ? e->z_exp : e->z_exp_d.p) + (size_t)e->exp_pos * N * d;
      pa.u_export = (e->exp_dev ? e->u_exp : e->u_exp_d.p) + (size_t)e->exp_pos * N;
    }
    if (e->is_dreamz) {  // everything DREAMZ.make_proposal draws for the block's base steps (the archive grows by a row per step)
      DreamDrawArgs da{};
      da.N = N;
      da.NP = NP;
      da.chain_offset = e->cfg.chain_offset;
      da.d = d;
      da.S = (int)S;
      da.delta = e->dz.delta;
      da.nCR = e->dz.nCR;
      da.step0 = e->t;
      da.M_base = e->arch_rows;
      da.grow = 1;
      da.seed = e->cfg.seed;
      da.b = e->dz.b;
      da.b_star = e->dz.b_star;
      da.scaling = e->scaling.p;
      da.pCR = e->dz_pCR.p;
      da.coef = e->dz_coef.p;
      da.epsm = e->dz_epsm.p;
      da.ridx = e->dz_ridx.p;
      da.u = e->ublk.p;
      da.mcr_last = e->dz_mcr_last.p;
      if (e->rp_steps) {
        const size_t o = (size_t)e->rp_pos * N;
        da.r_rep = e->rp_r.p + o * e->dz.delta * 2;
        da.mcr_rep = e->rp_mcr.p + o;
        da.forced_rep = e->rp_forced.p + o;
        da.sub_rep = e->rp_sub.p + o * d;
        da.e_rep = e->rp_e.p + o * d;
        da.eps_rep = e->rp_eps.p + o * d;
        da.u_rep = e->rp_u.p + o;
      }
      if (e->exp_steps) {
        da.eps_export = (e->exp_dev ? e->z_exp : e->z_exp_d.p) + (size_t)e->exp_pos * N * d;
        da.u_export = (e->exp_dev ? e->u_exp : e->u_exp_d.p) + (size_t)e->exp_pos * N;
      }
      da.arch_shared = nullptr;
      ScopedTimer tm(e, 0);
      DISPATCH_DPAD(DP, launch_dz_draw<DPAD>(da, e->stream));
    } else if (e->L_identity && !e->rep_steps && pa.logu) {
      // identity proposal factor (CrankNicolson under a standard-normal prior): the increments are the normals, drawn at
      // eight waves per SIMD straight into the increment block (C3: k_propose 3.6 ms -> this, per 2000 base steps)
      RngArgs ra{};
      ra.N = N;
      ra.NP = NP;
      ra.chain_offset = e->cfg.chain_offset;
      ra.d = d;
      ra.S = (int)S;
      ra.step0 = e->t;
      ra.seed = e->cfg.seed;
      ra.u = pa.u;
      ra.logu = pa.logu;
      ra.z_export = pa.z_export;
      ra.u_export = pa.u_export;
      ScopedTimer tm(e, 0);
      DISPATCH_DPAD(DP, launch_rng_direct<DPAD>(ra, e->inc.p, e->stream));
    } else {
      ScopedTimer tm(e, 0);
      DISPATCH_DPAD(DP, launch_propose<DPAD>(pa, e->stream));
    }

    MLArgs ma{};
    ma.logu0 = pa.logu;
    for (int k = 0; k < nl; ++k) {
      const Level& lv = e->levels[k];
      ma.lv[k].Apk = lv.Apk.p;
      ma.lv[k].ytil = lv.ytil.p;
      ma.lv[k].w = lv.w.p;
      ma.lv[k].ncb = lv.ncb;
      ma.lv[k].m_pad = lv.m_pad;
      ma.lv[k].noise_kind = lv.noise_kind;
      ma.lv[k].var = lv.var;
      ma.lds_y[k] = lds_y[k];
      ma.lds_w[k] = lds_w[k];
      ma.sl[k] = e->sl[k];
      ma.cnt[k] = e->cnt[k];
      ma.done[k] = e->done[k];
    }
    ma.lds_total = off;
    ma.pr.mean = e->prior_mean.p;
    ma.pr.pinv = e->prior_pinv.p;
    ma.pr.Wpk = e->prior_Wpk.p;
    ma.pr.wmu = e->prior_wmu.p;
    ma.pr.ncb = e->prior_ncb;
    ma.pr.kind = e->prior_kind;
    ma.pr.lo = e->prior_bounded ? e->prior_lo.p : nullptr;
    ma.pr.hi = e->prior_bounded ? e->prior_hi.p : nullptr;
    ma.pr.logconst = e->prior_logconst;
    ma.N = N;
    ma.NP = NP;
    ma.d = d;
    ma.S = (int)S;
    ma.prop_kind = e->pp.kind;
    ma.nlev = nl;
    ma.randomize = e->randomize;
    ma.seed = e->cfg.seed;
    ma.chain_offset = e->cfg.chain_offset;
    ma.theta = e->ml_theta.p;
    ma.lp = e->ml_lp.p;
    ma.ll = e->ml_ll.p;
    ma.Sst = e->ml_S.p;
    ma.anyacc = e->ml_anyacc.p;
    ma.ysnap = e->ml_ysnap.p;
    ma.pick = e->ml_pick.p;
    ma.scaling = e->scaling.p;
    ma.ring = e->ml_ring.p;
    ma.ring_P = e->ring_P;
    ma.ring_pos = e->ring_pos;
    ma.inc = e->inc.p;
    ma.u0 = e->ublk.p;
    for (int k = 1; k < nl; ++k)
      ma.u_rep[k] = e->u_rep_lv_n[k] ? e->u_rep_lv[k].p + (size_t)e->u_rep_lv_pos[k] * N : nullptr;
    ma.ridx_rep = e->ridx_rep_n ? e->ridx_rep.p + (size_t)e->ridx_rep_pos * N : nullptr;
    ma.cascade = e->aem ? 0 : 1;
    ma.aem_on = e->aem ? 1 : 0;
    ma.aem_mp = e->aem ? e->levels[0].m_pad : 0;
    ma.aem_ld = e->aem_ld;
    ma.aem_bias = e->aem ? e->aem_bias[0].p : nullptr;
    ma.aem_P = e->aem ? e->aem_covinv[0].p : nullptr;
    ma.sid = e->ml_sid.p;
    bool dev_p[MAXLEV], dev_s[MAXLEV], dev_a[MAXLEV];
    for (int k = 0; k < nl; ++k) {
      double* op = outs ? outs[k].params : nullptr;
      double* os = outs ? outs[k].stats : nullptr;
      uint8_t* oa = outs ? outs[k].accepted : nullptr;
      dev_p[k] = is_device_ptr(op);
      dev_s[k] = is_device_ptr(os);
      dev_a[k] = is_device_ptr(oa);
      const bool need_p = op || (k == 0 && is_am);
      const bool second = async_host && (blk_ix & 1);
      ma.rec_params[k] = dev_p[k] ? op + (size_t)rows_out[k] * N * d : (need_p ? (second ? e->ml_rec_params2[k].p : e->ml_rec_params[k].p) : nullptr);
      ma.rec_stats[k] = dev_s[k] ? os + (size_t)rows_out[k] * N * 3 : (os ? (second ? e->ml_rec_stats2[k].p : e->ml_rec_stats[k].p) : nullptr);
      ma.rec_acc[k] = dev_a[k] ? oa + (size_t)rows_out[k] * N : (oa ? (second ? e->ml_rec_acc2[k].p : e->ml_rec_acc[k].p) : nullptr);
    }
    if (async_host && blk_ix >= 2) HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_cp[blk_ix & 1], 0));  // buffer set free again
    if (e->ext_hier) {
      const int xrc = run_ext_hierarchy_block(e, ma, S, adaptive);
      if (xrc) return xrc;
    } else {
      ScopedTimer tm(e, 1);
      int lrc = TDA_OK;
      DISPATCH_DPAD(DP, lrc = launch_ml<DPAD>(ma, NP / 16, lds, e->stream));
      if (lrc) return lrc;
    }

    const bool boundary = periodic && ((e->t + S) % period == 0);
    if (e->is_dreamz) {
      // archive column sums of the S rows the block appended; at a period boundary the global scaling (accept-flag window
      // incl. alignment entries) and the crossover probabilities (proposal.py:797-809)
      int64_t app = 0;
      for (int k = 0; k < nl; ++k) app += nblk[k];
      ScopedTimer tm(e, 2);
      const int drc = dreamz_sums_catchup(e, e->arch_rows, S, boundary, adaptive, std::pow(e->dz.gamma, -(double)e->k_adapt), e->theta_last.p,
                                          adaptive ? e->ml_ring.p : nullptr, e->ring_P, e->ring_pos + app - tail_appends);
      if (drc) return drc;
      e->arch_rows += S;
      if (e->rp_steps) e->rp_pos += S;
    } else if (is_am || (boundary && adaptive)) {
      AdaptArgs aa{};
      aa.N = N;
      aa.NP = NP;
      aa.d = d;
      aa.S = (int)S;
      aa.t_base = e->t;
      aa.do_am = is_am;
      aa.boundary = boundary;
      aa.do_scale = adaptive;
      const bool do_swap = is_am && boundary && (e->t + S >= e->pp.t0);
      aa.do_swap = do_swap;
      aa.block_moments = e->pp.block_moments != 0;
      aa.period = period;
      aa.gamma_pow = std::pow(e->pp.gamma, -(double)e->k_adapt);
      aa.sd = e->am_sd;
      aa.alpha_star = e->pp.kind == TDA_PROP_MALA ? 0.57 : 0.24;  // proposal.py:899 / :169
      aa.eps = e->pp.epsilon;
      aa.rec_params = ma.rec_params[0];
      aa.am_mu = e->am_mu.p;
      aa.am_sigma = e->am_sigma.p;
      aa.scaling = e->scaling.p;
      aa.acc_count = e->acc_count.p;
      aa.flags = e->flags.p;
      aa.ring = adaptive ? e->ml_ring.p : nullptr;
      aa.ring_P = e->ring_P;
      {
        int64_t app = 0;
        for (int k = 0; k < nl; ++k) app += nblk[k];
        aa.ring_hi = e->ring_pos + app - tail_appends;
      }
      ScopedTimer tm(e, 2);
      DISPATCH_DPAD(DP, launch_adapt<DPAD>(aa, e->stream));
      if (do_swap) {
        CholArgs ca{};
        ca.N = N;
        ca.d = d;
        ca.am_sigma = e->am_sigma.p;
        ca.Lk = e->Lk.p;
        ca.flags = e->flags.p;
        DISPATCH_DPAD(DP, launch_chol<DPAD>(ca, e->stream));
      }
    }
    HIP_TRY(hipGetLastError());
    if (boundary && adaptive) e->k_adapt += 1;

    if (e->aem && !e->ext_hier) {
      // host-sequenced upper levels: decision of level q, then the error-model update of level q-1, in ascending q
      int64_t extra = 0;
      for (int qq = 1; qq < nl; ++qq) {
        if (nblk[qq] == 0) continue;
        AemArgs ag{};
        ag.N = N;
        ag.NP = NP;
        ag.chain_offset = e->cfg.chain_offset;
        ag.d = d;
        ag.DP = DP;
        ag.m = e->aem_m;
        ag.MP = e->aem_ld;
        ag.nlev = nl;
        ag.q = qq;
        ag.is_da = nl == 2;
        ag.dependent = e->aem == TDA_AEM_STATE_DEPENDENT;
        ag.prop_kind = e->pp.kind;
        ag.seed = e->cfg.seed;
        ag.step = e->done[qq];
        for (int k = 0; k < nl; ++k) {
          ag.A[k] = e->levels[k].A_rm.p;
          ag.ytil[k] = e->levels[k].ytil64.p;
          ag.data[k] = e->levels[k].data64.p;
          ag.cov[k] = e->levels[k].cov64.p;
          ag.bias_tot[k] = e->aem_bias[k].p;
          ag.cov_inv[k] = e->aem_covinv[k].p;
          ag.b_mu[k] = e->aem_bmu[k].p;
          ag.b_sig[k] = e->aem_bsig[k].p;
          ag.mdiff[k] = e->aem_mdiff[k].p;
        }
        ag.var_finest = e->levels[nl - 1].var;
        ag.pr_mean = e->prior_mean.p;
        ag.pr_W = e->prior_W_rm.p;
        ag.pr_logdet = e->prior_logconst - d * std::log(2.0 * M_PI);
        ag.theta = e->ml_theta.p;
        ag.lp = e->ml_lp.p;
        ag.ll = e->ml_ll.p;
        ag.Sst = e->ml_S.p;
        ag.anyacc = e->ml_anyacc.p;
        ag.sid = e->ml_sid.p;
        ag.b_t = e->aem_bt[qq];
        ag.scaling = e->scaling.p;
        ag.u_rep = e->u_rep_lv_n[qq] ? e->u_rep_lv[qq].p + (size_t)e->u_rep_lv_pos[qq] * N : nullptr;
        ag.ring = e->ml_ring.p;
        ag.ring_P = e->ring_P;
        ag.ring_pos = e->ring_pos + S + extra;
        ag.rec_params = ma.rec_params[qq];
        ag.rec_stats = ma.rec_stats[qq];
        ag.rec_acc = ma.rec_acc[qq];
        ScopedTimer tm(e, 2);
        AemInvArgs iv{};
        iv.N = N;
        iv.m = e->aem_m;
        iv.MP = e->aem_ld;
        iv.nb = (e->aem_m + 15) / 16;
        iv.cov = e->levels[qq - 1].cov64.p;
        if (ag.dependent) {
          iv.nsum = 1;
          iv.sig[0] = e->aem_bsig[qq].p;
        } else {
          iv.nsum = nl - qq;
          for (int p = qq; p < nl; ++p) iv.sig[p - qq] = e->aem_bsig[p].p;
        }
        iv.P = e->aem_covinv[qq - 1].p;
        const size_t inv_lds = aem_inverse_lds_bytes(iv.nb);
        const bool inv8 = iv.nb > 4;  // eight waves per chain pay from five block rows on (tools/aem_inverse_probe.hip)
        if (inv_lds > 64 * 1024)  // beyond the default dynamic-LDS window
          HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aem_inverse<0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)inv_lds));
        for (int phase = 0; phase < 2; ++phase) {
          ag.phase = phase;
          if (e->aem_ld == 64) hipLaunchKernelGGL(k_aem_action<64>, dim3((unsigned)N), dim3(64), 0, e->stream, ag);
          else hipLaunchKernelGGL(k_aem_action<128>, dim3((unsigned)N), dim3(128), 0, e->stream, ag);
          if (phase == 0) {
            if (inv8) hipLaunchKernelGGL((k_aem_inverse<0, 8>), dim3((unsigned)N), dim3(512), inv_lds, e->stream, iv);
            else hipLaunchKernelGGL((k_aem_inverse<0, 4>), dim3((unsigned)N), dim3(256), inv_lds, e->stream, iv);
          }
        }
        e->aem_bt[qq] += 1;
        extra += 1;
      }
      HIP_TRY(hipGetLastError());
    }

    bool host_copies = false;
    int rc;
    if (async_host) {
      const int i = (int)(blk_ix & 1);
      HIP_TRY(hipEventRecord(e->ev_rec[i], e->stream));
      HIP_TRY(hipStreamWaitEvent(e->copy_stream, e->ev_rec[i], 0));
      for (int k = 0; k < nl; ++k) {
        if (outs[k].params && !dev_p[k])
          HIP_TRY(hipMemcpyAsync(outs[k].params + (size_t)rows_out[k] * N * d, ma.rec_params[k], (size_t)nblk[k] * N * d * sizeof(double), hipMemcpyDeviceToHost, e->copy_stream));
        if (outs[k].stats && !dev_s[k])
          HIP_TRY(hipMemcpyAsync(outs[k].stats + (size_t)rows_out[k] * N * 3, ma.rec_stats[k], (size_t)nblk[k] * N * 3 * sizeof(double), hipMemcpyDeviceToHost, e->copy_stream));
        if (outs[k].accepted && !dev_a[k])
          HIP_TRY(hipMemcpyAsync(outs[k].accepted + (size_t)rows_out[k] * N, ma.rec_acc[k], (size_t)nblk[k] * N, hipMemcpyDeviceToHost, e->copy_stream));
      }
      HIP_TRY(hipEventRecord(e->ev_cp[i], e->copy_stream));
    } else {
      for (int k = 0; k < nl; ++k) {
        if (!outs) break;
        if (outs[k].params && !dev_p[k]) {
          if ((rc = copy_out(e, outs[k].params + (size_t)rows_out[k] * N * d, e->ml_rec_params[k].p, (size_t)nblk[k] * N * d * sizeof(double)))) return rc;
          host_copies = true;
        }
        if (outs[k].stats && !dev_s[k]) {
          if ((rc = copy_out(e, outs[k].stats + (size_t)rows_out[k] * N * 3, e->ml_rec_stats[k].p, (size_t)nblk[k] * N * 3 * sizeof(double)))) return rc;
          host_copies = true;
        }
        if (outs[k].accepted && !dev_a[k]) {
          if ((rc = copy_out(e, outs[k].accepted + (size_t)rows_out[k] * N, e->ml_rec_acc[k].p, (size_t)nblk[k] * N))) return rc;
          host_copies = true;
        }
      }
      if (host_copies) HIP_TRY(hipStreamSynchronize(e->stream));
    }
    blk_ix += 1;
    if (int prc = progress_mark(e, nblk[nl - 1], nullptr, 0)) return prc;  // progress counts finest-level iterations (chain.py:343-351)

    int64_t appended = 0;
    for (int k = 0; k < nl; ++k) {
      e->cnt[k] = c2[k];
      e->done[k] += nblk[k];
      rows_out[k] += nblk[k];
      appended += nblk[k];
      if (k >= 1 && e->u_rep_lv_n[k]) e->u_rep_lv_pos[k] += nblk[k];
    }
    if (e->ridx_rep_n) e->ridx_rep_pos += nblk[1];
    e->ring_pos += appended;
    e->t += S;
    done_base += S;
    if (e->rep_steps) e->rep_pos += S;
    if (e->exp_steps) e->exp_pos += S;
  }
  if (async_host) {  // host outputs: run() returns with the records in place
    HIP_TRY(hipStreamSynchronize(e->copy_stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
  }
  if (e->exp_steps && !e->exp_dev) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->z_exp, e->z_exp_d.p, (size_t)e->exp_pos * N * d * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(e->u_exp, e->u_exp_d.p, (size_t)e->exp_pos * N * sizeof(double), hipMemcpyDeviceToHost));
  }
  return TDA_OK;
}

void fill_dreamz_step_args(tda_engine* e, DreamStepArgs& sa) {
  const Level& lv = e->levels[0];
  sa.lv.Apk = lv.Apk.p;
  sa.lv.ytil = lv.ytil.p;
  sa.lv.w = lv.w.p;
  sa.lv.ncb = lv.ncb;
  sa.lv.m_pad = lv.m_pad;
  sa.lv.noise_kind = lv.noise_kind;
  sa.lv.var = lv.var;
  sa.pr.mean = e->prior_mean.p;
  sa.pr.pinv = e->prior_pinv.p;
  sa.pr.Wpk = e->prior_Wpk.p;
  sa.pr.wmu = e->prior_wmu.p;
  sa.pr.ncb = e->prior_ncb;
  sa.pr.kind = e->prior_kind;
  sa.pr.lo = e->prior_bounded ? e->prior_lo.p : nullptr;
  sa.pr.hi = e->prior_bounded ? e->prior_hi.p : nullptr;
  sa.pr.logconst = e->prior_logconst;
  sa.model = lv.model;
  sa.ros_a = lv.ros_a;
  sa.ros_b = lv.ros_b;
  sa.ros_data = lv.ros_data;
  sa.N = e->N;
  sa.NP = e->NP;
  sa.d = e->d;
  sa.delta = e->dz.delta;
  sa.M_base = e->arch_rows;
  sa.shared = e->dz.shared;
  sa.cap = e->arch_cap;
  sa.arch = e->arch.p;
  sa.theta = e->theta.p;
  sa.theta_prev = e->theta_prev.p;
  sa.lp = e->lp.p;
  sa.ll = e->ll.p;
  sa.acc_count = e->acc_count.p;
  sa.coef = e->dz_coef.p;
  sa.epsm = e->dz_epsm.p;
  sa.ridx = e->dz_ridx.p;
  sa.u = e->ublk.p;
}

static int dreamz_sums_catchup(tda_engine* e, int64_t row0, int64_t nrows, bool boundary, bool scale, double gamma_pow,
                               const double* theta_now, const uint8_t* ring, int ring_P, int64_t ring_hi) {
  DreamAdaptArgs aa{};
  aa.N = e->N;
  aa.NP = e->NP;
  aa.d = e->d;
  aa.nCR = e->dz.nCR;
  aa.period = e->dz.period;
  aa.gamma_pow = gamma_pow;
  aa.shared = e->dz.shared;
  aa.cap = e->arch_cap;
  aa.arch = e->arch.p;
  aa.zsum = e->zsum.p;
  aa.zsq = e->zsq.p;
  aa.theta = theta_now ? theta_now : e->theta.p;
  aa.theta_prev = e->theta_prev.p;
  aa.ring = ring;
  aa.ring_P = ring_P;
  aa.ring_hi = ring_hi;
  aa.mcr_last = e->dz_mcr_last.p;
  aa.pCR = e->dz_pCR.p;
  aa.LCR = e->dz_LCR.p;
  aa.DeltaCR = e->dz_Delta.p;
  aa.scaling = e->scaling.p;
  aa.acc_count = e->acc_count.p;
  aa.row0 = row0;
  aa.nrows = nrows;
  aa.M_total = row0 + nrows;
  if (e->dz.shared) {
    // The column sums only feed the crossover adaptation (np.var(Z, axis=0), proposal.py:800), so appended rows are summed
    // when a boundary asks for them -- [sums_rows, row0 + nrows) in one pass -- and not after every exchange interval (two
    // launches per block, 18 % of the kernel time of C4 at an interval of 16 steps).  Chunked column sums in parallel, then
    // one workgroup accumulates the chunks in a fixed order: deterministic for a given sequence of appends and boundaries.
    if (!boundary) return TDA_OK;
    const int64_t upto = row0 + nrows, from = e->sums_rows;
    if (upto > from) {
      const int64_t nb = (upto - from + COLSUM_CHUNK - 1) / COLSUM_CHUNK;
      if (e->dz_partial.n < (size_t)nb * 2 * e->DP) {
        HIP_TRY(hipStreamSynchronize(e->stream));
        int rc = e->dz_partial.alloc((size_t)nb * 2 * e->DP);
        if (rc) return rc;
      }
      DISPATCH_DPAD(e->DP, launch_colsum<DPAD>(e->arch.p, from, upto - from, e->dz_partial.p, nb, e->stream));
      DISPATCH_DPAD(e->DP, launch_colsum_final<DPAD>(e->dz_partial.p, nb, e->zsum.p, e->zsq.p, e->stream));
      e->sums_rows = upto;
    }
    aa.nrows = 0;  // ... then every chain adapts against the finished sums
    aa.M_total = upto;
  }
  aa.boundary = boundary;
  aa.do_scale = scale;
  DISPATCH_DPAD(e->DP, launch_dz_adapt<DPAD>(aa, e->stream));
  HIP_TRY(hipGetLastError());
  return TDA_OK;
}

static int run_dreamz(tda_engine* e, int64_t n_iter, const tda_outputs* out) {
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (out && out->struct_size != sizeof(tda_outputs)) return fail(TDA_ERR_INVALID, "tda_outputs.struct_size mismatch");
  const int d = e->d, DP = e->DP;
  const int64_t N = e->N, NP = e->NP;
  const bool sh = e->dz.shared != 0, adaptive = e->dz.adaptive != 0;
  const int period = e->dz.period;
  if (e->rp_steps && e->rp_pos + n_iter > e->rp_steps) return fail(TDA_ERR_INVALID, "DREAMZ replay buffer too short");
  if (int crc = check_out_capacity(out, 0, n_iter, N, d)) return crc;
  if (e->exp_steps && e->exp_pos + n_iter > e->exp_steps) return fail(TDA_ERR_INVALID, "export buffer too small");
  if (!sh && e->arch_rows + n_iter > e->arch_cap) return fail(TDA_ERR_INVALID, "archive capacity (%lld rows) exceeded", (long long)e->arch_cap);
  if (sh && e->auto_append && e->arch_rows + n_iter * N > e->arch_cap)
    return fail(TDA_ERR_INVALID, "shared archive capacity (%lld rows) exceeded", (long long)e->arch_cap);
  double* o_params = out ? out->params : nullptr;
  double* o_stats = out ? out->stats : nullptr;
  uint8_t* o_acc = out ? out->accepted : nullptr;
  const bool p_dev = is_device_ptr(o_params), s_dev = is_device_ptr(o_stats), a_dev = is_device_ptr(o_acc);
  const Level& lv = e->levels[0];
  const bool linear = lv.model == MODEL_LINEAR;
  const bool diag = lv.noise_kind == TDA_NOISE_DIAG;
  const int prow = e->prior_kind == PRIOR_DENSE ? e->prior_ncb * 16 : 0;
  const size_t lds = ((size_t)16 * (DP + 2) + 128 + (linear ? lv.m_pad * (diag ? 2 : 1) : 0) + prow) * sizeof(double);
  if (e->profiling) {
    for (auto& t : e->timed) {
      (void)hipEventDestroy(t.a);
      (void)hipEventDestroy(t.b);
    }
    e->timed.clear();
  }
  const bool ext_model0 = lv.model == MODEL_CALLBACK || lv.model == MODEL_USER;
  // TINYDA_DZ_PIPELINE=1 (off by default -- measured SLOWER): one process, shared archive appended in place: everything
  // k_dreamz_draw produces except the archive gather depends on the step counter only (and on pCR / scaling, which change at
  // adaptation boundaries), so block b + 1 can be drawn on a second stream while block b steps, the step kernel gathering the
  // archive rows itself (the same sums in the same order: results do not depend on this switch, the sharding-invariance tests
  // pass either way).  On C4 (8192 chains, d = 32, interval 16) the two kernels do run concurrently but each then takes the sum
  // of their stand-alone times (draw 52 us and steps 28 us alone, 80-97 us each together: the draw saturates the VALU issue of
  // every SIMD and the steps' dependent chains wait behind it, s_setprio made no difference) and every block pays ~13 us for
  // the cross-stream events: 1.00e9 evals/s against 1.19e9 for the plain sequence.
  static const bool pipe_ok = getenv("TINYDA_DZ_PIPELINE") && atoi(getenv("TINYDA_DZ_PIPELINE")) == 1;
  const bool pipe = pipe_ok && sh && e->auto_append && N == NP && e->pending_steps == 0 && !ext_model0 && !e->dist_ranks;
  if (pipe && !e->dz_coef2.p) {
    int rc;
    if ((rc = e->dz_coef2.alloc((size_t)e->SMAX * NP * DP)) || (rc = e->dz_epsm2.alloc((size_t)e->SMAX * NP * DP)) ||
        (rc = e->dz_ridx2.alloc((size_t)e->SMAX * NP * 2 * MAX_DELTA)) || (rc = e->dz_u2.alloc((size_t)e->SMAX * NP)))
      return rc;
  }
  if (pipe && !e->rng_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&e->rng_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
      HIP_TRY(hipEventCreateWithFlags(&e->ev_rng[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->ev_apply[i], hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&e->ev_steps[i], hipEventDisableTiming));
    }
  }
  if (pipe && !e->ev_dz_adapt) HIP_TRY(hipEventCreateWithFlags(&e->ev_dz_adapt, hipEventDisableTiming));
  // Shared archive, one process appending its own rows: the exchange points are the CALL-RELATIVE multiples of sync_every -- where
  // the multi-rank driver (distributed.run_shared_dream: one run() per interval, then the all-gather) has them.  A block that an
  // adaptation boundary cuts in two is not an exchange point: the rows of its first half become visible together with the second
  // half's (`since` = steps since the last exchange point), so one process and N ranks see the same archive at every step whatever
  // the period (the engine used to restart full intervals after a boundary).
  int64_t since = 0;
  const bool pending_at_entry = e->pending_steps != 0;  // (rows taken over from an earlier call sit in blk_hist, not in place)
  auto block_len = [&](int64_t t_now, int64_t left, int64_t since_now) {
    int64_t S = std::min<int64_t>(left, e->SMAX);
    if (adaptive) S = std::min<int64_t>(S, period - (t_now % period));
    if (sh) {
      const int64_t K = e->dz.sync_every > 0 ? e->dz.sync_every : e->SMAX;
      S = std::min<int64_t>(S, K - (e->auto_append && !e->dist_ranks ? since_now % K : 0));
      if (!e->dist_ranks && !e->auto_append && e->pending_steps + S > e->SMAX) S = e->SMAX - e->pending_steps;
    }
    return S;
  };
  if (e->dist_ranks) {  // distributed archive: one exchange interval per call, then tda_engine_archive_publish (after the ranks met)
    if (e->dist_n_unpub >= 2) return fail(TDA_ERR_STATE, "distributed archive: two blocks wait for tda_engine_archive_publish");
    if (e->dist_adapt_pending) return fail(TDA_ERR_STATE, "distributed archive: an adaptation waits for tda_engine_archive_publish");
    const int64_t K = e->dz.sync_every > 0 ? e->dz.sync_every : e->SMAX;
    if (n_iter > K || (adaptive && n_iter > period - (e->t % period)))
      return fail(TDA_ERR_INVALID, "distributed archive: a run() call covers at most one exchange interval (%lld steps) and does not cross an adaptation boundary", (long long)K);
    if (e->dz.M0 + (e->dist_steps + e->dist_pending + n_iter) * N > e->arch_cap) return fail(TDA_ERR_INVALID, "archive segment capacity (%lld rows) exceeded", (long long)e->arch_cap);
  }
  // everything DREAMZ.make_proposal draws for S steps from step t0 on, the archive holding M_base rows: into buffer set `set`
  auto enqueue_draw = [&](int set, int64_t t0, int64_t M_base, int64_t S, int64_t rp_pos, int64_t exp_pos, bool gather, hipStream_t st) {
    DreamDrawArgs da{};
    da.N = N;
    da.NP = NP;
    da.chain_offset = e->cfg.chain_offset;
    da.d = d;
    da.S = (int)S;
    da.delta = e->dz.delta;
    da.nCR = e->dz.nCR;
    da.step0 = t0;
    da.M_base = M_base;
    da.grow = sh ? 0 : 1;
    da.seed = e->cfg.seed;
    da.b = e->dz.b;
    da.b_star = e->dz.b_star;
    da.scaling = e->scaling.p;
    da.pCR = e->dz_pCR.p;
    da.coef = set ? e->dz_coef2.p : e->dz_coef.p;
    da.epsm = set ? e->dz_epsm2.p : e->dz_epsm.p;
    da.ridx = set ? e->dz_ridx2.p : e->dz_ridx.p;
    da.u = set ? e->dz_u2.p : e->ublk.p;
    da.mcr_last = e->dz_mcr_last.p;
    if (e->rp_steps) {
      const size_t o = (size_t)rp_pos * N;
      da.r_rep = e->rp_r.p + o * e->dz.delta * 2;
      da.mcr_rep = e->rp_mcr.p + o;
      da.forced_rep = e->rp_forced.p + o;
      da.sub_rep = e->rp_sub.p + o * d;
      da.e_rep = e->rp_e.p + o * d;
      da.eps_rep = e->rp_eps.p + o * d;
      da.u_rep = e->rp_u.p + o;
    }
    if (e->exp_steps) {
      da.eps_export = (e->exp_dev ? e->z_exp : e->z_exp_d.p) + (size_t)exp_pos * N * d;
      da.u_export = (e->exp_dev ? e->u_exp : e->u_exp_d.p) + (size_t)exp_pos * N;
    }
    da.arch_shared = gather ? e->arch.p : nullptr;
    if (e->dist_ranks) {
      da.dist_ranks = e->dist_ranks;
      da.dist_me = e->dist_me;
      da.dist_M0 = e->dz.M0;
      da.dist_nloc = N;
      da.dist_ntot = N * e->dist_ranks;
      da.seg = e->dist_seg_dev.p;
    }
    DISPATCH_DPAD(DP, launch_dz_draw<DPAD>(da, st));
  };
  int64_t done = 0, blk = 0;
  if (pipe && n_iter > 0) {
    // everything queued on the main stream so far (init, earlier run() calls, their adaptation) precedes the first draw
    HIP_TRY(hipEventRecord(e->ev_dz_adapt, e->stream));
    HIP_TRY(hipStreamWaitEvent(e->rng_stream, e->ev_dz_adapt, 0));
    enqueue_draw(0, e->t, e->arch_rows, block_len(e->t, n_iter, 0), e->rp_pos, e->exp_pos, false, e->rng_stream);
    HIP_TRY(hipEventRecord(e->ev_rng[0], e->rng_stream));
  }
  while (done < n_iter) {
    const int64_t S = block_len(e->t, n_iter - done, since);
    if (S <= 0) return fail(TDA_ERR_STATE, "shared archive: call archive_take / archive_append before running further");
    const int set = pipe ? (int)(blk & 1) : 0;
    if (pipe) {
      HIP_TRY(hipStreamWaitEvent(e->stream, e->ev_rng[set], 0));
    } else {
      ScopedTimer tm(e, 0);
      enqueue_draw(0, e->t, e->arch_rows, S, e->rp_pos, e->exp_pos, sh, e->stream);
    }
    DreamStepArgs sa{};
    fill_dreamz_step_args(e, sa);
    sa.S = (int)S;
    sa.jump_ready = (sh && !pipe) ? 1 : 0;
    if (set) {
      sa.coef = e->dz_coef2.p;
      sa.epsm = e->dz_epsm2.p;
      sa.ridx = e->dz_ridx2.p;
      sa.u = e->dz_u2.p;
    }
    const bool ext_model = ext_model0;
    sa.rec_params = p_dev ? o_params + (size_t)done * N * d : (o_params ? e->rec_params.p : nullptr);
    sa.rec_stats = s_dev ? o_stats + (size_t)done * N * 3 : (o_stats ? e->rec_stats.p : nullptr);
    sa.rec_acc = a_dev ? o_acc + (size_t)done * N : (o_acc ? e->rec_acc.p : nullptr);
    // single process, no padding chains: the block's states ARE the next archive rows in canonical order (step-major,
    // chain minor), the kernel appends them in place (the jumps of this block were gathered before it started)
    const bool dist = e->dist_ranks != 0;
    const bool direct = !dist && sh && e->auto_append && N == NP && !pending_at_entry;
    // distributed archive: the block's states are this rank's next rows of its own segment, written in place
    // (in place = behind the visible rows and behind the rows of an interval's earlier blocks that are not visible yet)
    sa.blk_states = sh ? (dist ? e->arch.p + (size_t)(e->dz.M0 + (e->dist_steps + e->dist_pending) * N) * DP
                               : (direct ? e->arch.p + (size_t)(e->arch_rows + e->pending_steps * N) * DP : e->blk_states.p))
                       : nullptr;
    if (ext_model) {
      // model outside the engine's kernels: per step apply the jump, evaluate (callback: one host call for all chains;
      // source-defined: tda_user_eval on the stream), accept, append
      if (e->prior_kind == PRIOR_DENSE) return fail(TDA_ERR_UNSUPPORTED, "callback / source-defined forward models need a diagonal prior covariance");
      const unsigned grid = (unsigned)((N + EXT_WAVES - 1) / EXT_WAVES), gridp = (unsigned)((NP + EXT_WAVES - 1) / EXT_WAVES);
      for (int64_t s = 0; s < S; ++s) {
        if (s == S - 1)  // jumping distance of the adaptation: the state before the block's last step (proposal.py:800)
          HIP_TRY(hipMemcpyAsync(e->theta_prev.p, e->theta.p, (size_t)NP * DP * sizeof(double), hipMemcpyDeviceToDevice, e->stream));
        DzExtArgs za{};
        za.N = N;
        za.NP = NP;
        za.d = d;
        za.DP = DP;
        za.delta = e->dz.delta;
        za.s = (int)s;
        za.shared = sh ? 1 : 0;
        za.jump_ready = sa.jump_ready;
        za.M_base = e->arch_rows;
        za.cap = e->arch_cap;
        za.arch = e->arch.p;
        za.theta = e->theta.p;
        za.coef = e->dz_coef.p;
        za.epsm = e->dz_epsm.p;
        za.ridx = e->dz_ridx.p;
        za.prop = lv.cb_prop.p;
        za.blk_states = sa.blk_states;
        hipLaunchKernelGGL(k_dz_ext_propose, dim3(grid), dim3(64 * EXT_WAVES), 0, e->stream, za);
        int mrc = ext_model_outputs(e, lv);
        if (mrc) return mrc;
        ExtArgs xa{};
        fill_ext_args(e, lv, xa);
        xa.mode = 0;
        xa.prop_kind = TDA_PROP_GRW;  // posterior ratio (proposal.py:253-258)
        xa.theta = e->theta.p;
        xa.lp = e->lp.p;
        xa.ll = e->ll.p;
        xa.scaling = e->scaling.p;
        xa.acc_count = e->acc_count.p;
        xa.u = e->ublk.p;
        xa.s = (int)s;
        xa.rec_params = sa.rec_params;
        xa.rec_stats = sa.rec_stats;
        xa.rec_acc = sa.rec_acc;
        hipLaunchKernelGGL(k_ext_accept, dim3(grid), dim3(64 * EXT_WAVES), xa.Pd ? (size_t)EXT_WAVES * lv.m * sizeof(double) : 0, e->stream, xa);
        hipLaunchKernelGGL(k_dz_ext_append, dim3(gridp), dim3(64 * EXT_WAVES), 0, e->stream, za);
      }
    } else {
      ScopedTimer tm(e, 1);
      DISPATCH_DPAD(DP, launch_dz_steps<DPAD>(sa, lds, e->stream));
    }
    HIP_TRY(hipGetLastError());
    const bool boundary = adaptive && ((e->t + S) % period == 0);
    int rc;
    if (!sh) {  // per-chain archives grew inside the kernel
      ScopedTimer tm(e, 2);
      if ((rc = dreamz_sums_catchup(e, e->arch_rows, S, boundary, adaptive, std::pow(e->dz.gamma, -(double)e->k_adapt)))) return rc;
      e->arch_rows += S;
    } else if (dist) {
      // rows are in place in this rank's segment; they become visible (and an adaptation at this boundary runs) in
      // tda_engine_archive_publish, once every rank has finished the block
      e->dist_unpub[e->dist_n_unpub++] = S;
      e->dist_pending += S;
      e->dist_adapt_pending = boundary;
      e->dist_adapt_gamma = std::pow(e->dz.gamma, -(double)e->k_adapt);
      e->dist_adapt_rows = e->arch_rows;  // what this block's proposals could see
    } else {
      // keep this block's states for the exchange, [pending + s][NP][DP]
      if (!direct)
        HIP_TRY(hipMemcpyAsync(e->blk_hist.p + (size_t)e->pending_steps * NP * DP, e->blk_states.p, (size_t)S * NP * DP * sizeof(double),
                               hipMemcpyDeviceToDevice, e->stream));
      e->pending_steps += S;
      // crossover / scaling adaptation sees the archive the finished block proposed from (rows of this block are
      // appended afterwards), so the result does not depend on who appends when
      {
        ScopedTimer tm(e, 2);
        if (boundary && (rc = dreamz_sums_catchup(e, e->arch_rows, 0, true, adaptive, std::pow(e->dz.gamma, -(double)e->k_adapt)))) return rc;
      }
      since += S;
      const int64_t Ksync = e->dz.sync_every > 0 ? e->dz.sync_every : e->SMAX;
      if (e->auto_append && (since % Ksync == 0 || done + S >= n_iter)) {  // single process: the local rows are all rows; canonical order = step-major, chain minor
        if (direct) {
          // already in place
        } else if (N == NP) {
          HIP_TRY(hipMemcpyAsync(e->arch.p + (size_t)e->arch_rows * DP, e->blk_hist.p, (size_t)e->pending_steps * N * DP * sizeof(double),
                                 hipMemcpyDeviceToDevice, e->stream));
        } else {
          for (int64_t s = 0; s < e->pending_steps; ++s)
            HIP_TRY(hipMemcpy2DAsync(e->arch.p + (size_t)(e->arch_rows + s * N) * DP, DP * sizeof(double),
                                     e->blk_hist.p + (size_t)s * NP * DP, DP * sizeof(double), DP * sizeof(double), N,
                                     hipMemcpyDeviceToDevice, e->stream));
        }
        const int64_t new_rows = e->pending_steps * N;
        e->pending_steps = 0;
        if ((rc = dreamz_sums_catchup(e, e->arch_rows, new_rows, false, false, 1.0))) return rc;
        e->arch_rows += new_rows;
      }
    }
    if (boundary && !dist) e->k_adapt += 1;
    if (pipe) {
      // block b's steps (and its adaptation, if any) are queued: its buffer set is free once they have run.  The draws of
      // block b + 1 start as soon as the set they overwrite (block b - 1's) is free -- i.e. under block b's steps -- unless
      // block b ended on an adaptation boundary, whose new pCR / scaling they must see.
      HIP_TRY(hipEventRecord(e->ev_steps[set], e->stream));
      if (done + S < n_iter) {
        const int nset = (int)((blk + 1) & 1);
        if (blk >= 1) HIP_TRY(hipStreamWaitEvent(e->rng_stream, e->ev_steps[nset], 0));
        if (boundary) HIP_TRY(hipStreamWaitEvent(e->rng_stream, e->ev_steps[set], 0));
        enqueue_draw(nset, e->t + S, e->arch_rows, block_len(e->t + S, n_iter - done - S, since), e->rp_pos + S, e->exp_pos + S, false, e->rng_stream);
        HIP_TRY(hipEventRecord(e->ev_rng[nset], e->rng_stream));
      }
    }
    bool host_copies = false;
    if (o_params && !p_dev) {
      if ((rc = copy_out(e, o_params + (size_t)done * N * d, e->rec_params.p, (size_t)S * N * d * sizeof(double)))) return rc;
      host_copies = true;
    }
    if (o_stats && !s_dev) {
      if ((rc = copy_out(e, o_stats + (size_t)done * N * 3, e->rec_stats.p, (size_t)S * N * 3 * sizeof(double)))) return rc;
      host_copies = true;
    }
    if (o_acc && !a_dev) {
      if ((rc = copy_out(e, o_acc + (size_t)done * N, e->rec_acc.p, (size_t)S * N))) return rc;
      host_copies = true;
    }
    if (host_copies) HIP_TRY(hipStreamSynchronize(e->stream));
    if ((rc = progress_mark(e, S, nullptr, 0))) return rc;
    e->t += S;
    done += S;
    blk += 1;
    if (e->rp_steps) e->rp_pos += S;
    if (e->exp_steps) e->exp_pos += S;
  }
  if (e->exp_steps && !e->exp_dev) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipMemcpy(e->z_exp, e->z_exp_d.p, (size_t)e->exp_pos * N * d * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(e->u_exp, e->u_exp_d.p, (size_t)e->exp_pos * N * sizeof(double), hipMemcpyDeviceToHost));
  }
  return TDA_OK;
}

int tda_engine_sync(tda_engine* e) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return TDA_OK;
}

int tda_engine_set_record_thinning(tda_engine* e, int32_t thin) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  if (thin < 1) return fail(TDA_ERR_INVALID, "thin must be >= 1");
  if (thin > 1 && (e->nlev > 1 || e->is_dreamz)) return fail(TDA_ERR_UNSUPPORTED, "record thinning is implemented for single-level GRW / pCN / AM / MALA runs");
  e->thin = thin;
  return TDA_OK;
}

int tda_engine_set_progress(tda_engine* e, int enable) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  if (enable && !e->prog_h) {
    HIP_TRY(hipHostMalloc((void**)&e->prog_h, 2 * sizeof(double), hipHostMallocDefault));
    e->prog_h[0] = 0.0;
    e->prog_h[1] = -1.0;
    e->prog_queued = 0;
  } else if (!enable && e->prog_h) {
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipHostFree(e->prog_h));
    e->prog_h = nullptr;
  }
  return TDA_OK;
}

int tda_engine_get_progress(tda_engine* e, int64_t* iterations_done, int64_t* iterations_queued, double* accept_rate) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  if (!e->prog_h) return fail(TDA_ERR_STATE, "progress reporting is off (tda_engine_set_progress)");
  const volatile double* h = e->prog_h;  // written by the GPU at block ends; no HIP call, no synchronisation
  const double done = h[0];
  if (iterations_done) *iterations_done = (int64_t)done;
  if (accept_rate) *accept_rate = h[1];
  if (iterations_queued) *iterations_queued = e->prog_queued;
  return TDA_OK;
}

}  // extern "C"

namespace {
// what tda_engine_get_proposal_state reads: the engine's own buffers, or the ones a snapshot took over from it
struct PropView {
  int device = 0, d = 0, DP = 0, kind = 0;
  int64_t N = 0, NP = 0, t = 0, k = 0;
  bool L_shared = true;
  const double *Lk = nullptr, *am_mu = nullptr, *am_sigma = nullptr, *scaling = nullptr;
};

int read_proposal_view(const PropView& v, double* scaling, double* C, double* am_mu, double* am_sigma, int64_t* counters) {
  const int d = v.d, DP = v.DP;
  const int64_t N = v.N, NP = v.NP;
  if (scaling) {
    std::vector<double> h(NP);
    HIP_TRY(hipMemcpy(h.data(), v.scaling, NP * sizeof(double), hipMemcpyDeviceToHost));
    std::copy(h.begin(), h.begin() + N, scaling);
  }
  if (C) {  // C = L L^T from the factor in use
    const int64_t nL = v.L_shared ? 1 : NP;
    std::vector<double> h((size_t)nL * DP * DP);
    HIP_TRY(hipMemcpy(h.data(), v.Lk, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    // C = L L^T as d rank-1 updates per chain (the innermost loop runs over contiguous memory and vectorises; every
    // element still accumulates its products in ascending k); a factor shared by all chains is multiplied out once
    auto unpack = [&](int64_t c0, int64_t c1) {
      std::vector<double> acc((size_t)d * d);
      for (int64_t c = c0; c < c1; ++c) {
        const double* Lc = h.data() + (size_t)c * DP * DP;
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int k = 0; k < d; ++k) {
          const double* Lk = Lc + (size_t)k * DP;  // column k of L: Lk[i] = L[i][k], zero for i < k
          for (int i = k; i < d; ++i) {
            const double lik = Lk[i];
            double* row = acc.data() + (size_t)i * d;
            for (int j = k; j <= i; ++j) row[j] += lik * Lk[j];
          }
        }
        double* Cc = C + (size_t)c * d * d;
        for (int i = 0; i < d; ++i)
          for (int j = 0; j <= i; ++j) Cc[(size_t)i * d + j] = Cc[(size_t)j * d + i] = acc[(size_t)i * d + j];
      }
    };
    if (v.L_shared) {
      unpack(0, 1);
      for (int64_t c = 1; c < N; ++c) std::copy(C, C + (size_t)d * d, C + (size_t)c * d * d);
    } else {
      host_chain_ranges(N, unpack);
    }
  }
  if (am_mu || am_sigma) {
    if (v.kind != TDA_PROP_AM) return fail(TDA_ERR_STATE, "proposal has no running moments");
    if (am_mu) {
      std::vector<double> h((size_t)NP * DP);
      HIP_TRY(hipMemcpy(h.data(), v.am_mu, h.size() * sizeof(double), hipMemcpyDeviceToHost));
      for (int64_t c = 0; c < N; ++c)
        for (int j = 0; j < d; ++j) am_mu[(size_t)c * d + j] = h[(size_t)c * DP + j];
    }
    if (am_sigma) {  // dense symmetric matrices from the lower-tile storage
      const size_t per = (size_t)am_tiles_rt(DP) * 256;
      std::vector<double> h((size_t)N * per);
      HIP_TRY(hipMemcpy(h.data(), v.am_sigma, h.size() * sizeof(double), hipMemcpyDeviceToHost));
      host_chain_ranges(N, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) {
          const double* f = h.data() + (size_t)c * per;
          for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j)
              am_sigma[((size_t)c * d + i) * d + j] = f[am_sigma_offset(std::max(i, j), std::min(i, j))];
        }
      });
    }
  }
  if (counters) {
    counters[0] = v.t;
    counters[1] = v.k;
  }
  return TDA_OK;
}

PropView view_of(const tda_engine* e) {
  PropView v;
  v.device = e->cfg.device;
  v.d = e->d;
  v.DP = e->DP;
  v.kind = e->is_dreamz ? TDA_PROP_DREAMZ : e->pp.kind;
  v.N = e->N;
  v.NP = e->NP;
  v.t = e->t;
  v.k = e->k_adapt;
  v.L_shared = e->L_shared;
  v.Lk = e->Lk.p;
  v.am_mu = e->am_mu.p;
  v.am_sigma = e->am_sigma.p;
  v.scaling = e->scaling.p;
  return v;
}
}  // namespace

// the proposal state of an engine that is about to be destroyed: the buffers themselves, taken over without a copy
struct tda_proposal_snapshot {
  PropView v;
  DevBuf<double> Lk, am_mu, am_sigma, scaling;
};

extern "C" {

int tda_engine_get_proposal_state(tda_engine* e, double* scaling, double* C, double* am_mu, double* am_sigma,
                                  int64_t* counters) {
  if (!e || !e->inited) return fail(TDA_ERR_STATE, "engine not initialised");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (C && !e->Lk.p) return fail(TDA_ERR_STATE, "this proposal has no covariance factor");
  return read_proposal_view(view_of(e), scaling, C, am_mu, am_sigma, counters);
}

int tda_engine_detach_proposal_state(tda_engine* e, tda_proposal_snapshot** out) {
  if (!e || !e->inited || !out) return fail(TDA_ERR_STATE, "engine not initialised");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (e->rng_stream) HIP_TRY(hipStreamSynchronize(e->rng_stream));
  tda_proposal_snapshot* s = new tda_proposal_snapshot();
  s->v = view_of(e);
  auto take = [](DevBuf<double>& dst, DevBuf<double>& src) {
    dst.p = src.p;
    dst.n = src.n;
    dst.dev = src.dev;
    src.p = nullptr;
    src.n = 0;
  };
  take(s->Lk, e->Lk);
  take(s->am_mu, e->am_mu);
  take(s->am_sigma, e->am_sigma);
  take(s->scaling, e->scaling);
  e->inited = false;  // the engine has given its proposal away: only tda_engine_destroy is left for it
  *out = s;
  return TDA_OK;
}

int tda_proposal_snapshot_read(tda_proposal_snapshot* s, double* scaling, double* C, double* am_mu, double* am_sigma, int64_t* counters) {
  if (!s) return fail(TDA_ERR_INVALID, "null snapshot");
  HIP_TRY(hipSetDevice(s->v.device));
  if (C && !s->v.Lk) return fail(TDA_ERR_STATE, "this proposal has no covariance factor");
  return read_proposal_view(s->v, scaling, C, am_mu, am_sigma, counters);
}

void tda_proposal_snapshot_destroy(tda_proposal_snapshot* s) {
  if (!s) return;
  (void)hipSetDevice(s->v.device);
  g_pool_accepting = true;  // nothing is queued on these buffers (detach synchronised the engine's streams)
  delete s;
  g_pool_accepting = false;
}

int tda_engine_get_flags(tda_engine* e, int32_t* flags) {
  if (!e || !e->inited || !flags) return fail(TDA_ERR_STATE, "engine not initialised");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::vector<int32_t> h(e->NP);
  HIP_TRY(hipMemcpy(h.data(), e->flags.p, e->NP * sizeof(int32_t), hipMemcpyDeviceToHost));
  std::copy(h.begin(), h.begin() + e->N, flags);
  return TDA_OK;
}

int tda_engine_evaluate(tda_engine* e, int level, const double* theta, int64_t n, double* stats) {
  if (!e || !theta || !stats) return fail(TDA_ERR_INVALID, "null argument");
  if (!e->prior_set) return fail(TDA_ERR_STATE, "set_prior missing");
  if (level < 0 || level >= (int)e->levels.size() || !e->levels[level].set) return fail(TDA_ERR_STATE, "level %d not set", level);
  if (n < 1 || n > e->N) return fail(TDA_ERR_INVALID, "n must be in 1..n_chains");
  HIP_TRY(hipSetDevice(e->cfg.device));
  int rc;
  const int64_t NP = e->NP;
  if (!e->theta_s.p) {
    if ((rc = e->theta_s.alloc((size_t)NP * e->DP))) return rc;
    if ((rc = e->lp_s.alloc(NP))) return rc;
    if ((rc = e->ll_s.alloc(NP))) return rc;
  }
  if (!e->scaling.p) {
    std::vector<double> sc(NP, 1.0);
    if ((rc = e->scaling.upload(sc))) return rc;
  }
  if ((rc = upload_states(e, theta, n, e->theta_s.p))) return rc;
  if ((rc = launch_eval(e, level, e->theta_s.p, e->lp_s.p, e->ll_s.p))) return rc;
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::vector<double> a(NP), b(NP), o((size_t)n * 3);
  HIP_TRY(hipMemcpy(a.data(), e->lp_s.p, NP * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(b.data(), e->ll_s.p, NP * sizeof(double), hipMemcpyDeviceToHost));
  for (int64_t c = 0; c < n; ++c) {
    o[c * 3] = a[c];
    o[c * 3 + 1] = b[c];
    o[c * 3 + 2] = a[c] + b[c];
  }
  HIP_TRY(hipMemcpy(stats, o.data(), o.size() * sizeof(double), is_device_ptr(stats) ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
  return TDA_OK;
}

int tda_engine_rng_probe(tda_engine* e, int64_t step, double* z, double* u) {
  if (!e || !z || !u) return fail(TDA_ERR_INVALID, "null argument");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int d = e->d, DP = e->DP;
  const int64_t N = e->N, NP = e->NP;
  DevBuf<double> Lid, inc, ub, zd, ud;
  std::vector<double> I((size_t)DP * DP, 0.0);
  for (int j = 0; j < DP; ++j) I[(size_t)j * DP + j] = 1.0;
  int rc;
  if ((rc = Lid.upload(I))) return rc;
  if ((rc = inc.alloc((size_t)NP * DP))) return rc;
  if ((rc = ub.alloc(NP))) return rc;
  if ((rc = zd.alloc((size_t)N * d))) return rc;
  if ((rc = ud.alloc(N))) return rc;
  ProposeArgs pa{};
  pa.N = N;
  pa.NP = NP;
  pa.chain_offset = e->cfg.chain_offset;
  pa.d = d;
  pa.S = 1;
  pa.step0 = step;
  pa.seed = e->cfg.seed;
  pa.Lk = Lid.p;
  pa.L_stride = 0;
  pa.inc = inc.p;
  pa.u = ub.p;
  pa.z_export = zd.p;
  pa.u_export = ud.p;
  DISPATCH_DPAD(DP, launch_propose<DPAD>(pa, e->stream));
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(e->stream));
  HIP_TRY(hipMemcpy(z, zd.p, (size_t)N * d * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(u, ud.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
  return TDA_OK;
}

namespace {
__global__ void k_philox_probe(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
  if (threadIdx.x == 0) {
    const tda::u32x4 r = tda::philox4x32_10(tda::u32x4{in[0], in[1], in[2], in[3]}, in[4], in[5]);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
    out[3] = r.w;
  }
}
}  // namespace

int tda_rng_philox(int device, const uint32_t* counter, const uint32_t* key, uint32_t* out) {
  if (!counter || !key || !out) return fail(TDA_ERR_INVALID, "null argument");
  if (device < 0) {  // the same header compiled for the host
    const tda::u32x4 r = tda::philox4x32_10(tda::u32x4{counter[0], counter[1], counter[2], counter[3]}, key[0], key[1]);
    out[0] = r.x;
    out[1] = r.y;
    out[2] = r.z;
    out[3] = r.w;
    return TDA_OK;
  }
  HIP_TRY(hipSetDevice(device));
  DevBuf<uint32_t> in, res;
  int rc;
  std::vector<uint32_t> h = {counter[0], counter[1], counter[2], counter[3], key[0], key[1]};
  if ((rc = in.upload(h))) return rc;
  if ((rc = res.alloc(4))) return rc;
  hipLaunchKernelGGL(k_philox_probe, dim3(1), dim3(64), 0, nullptr, in.p, res.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, res.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost));
  return TDA_OK;
}

#ifdef TDA_DA_TRACE
extern "C" int tda_debug_da_trace(long long* out) {  // debug builds only (tools/da_trace.py)
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(tda::g_da_trace), sizeof(long long) * 128 * 8 * 8));
  return TDA_OK;
}
#endif

int tda_engine_set_profiling(tda_engine* e, int enable) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  e->profiling = enable != 0;
  return TDA_OK;
}

int tda_engine_get_profile(tda_engine* e, tda_profile* p) {
  if (!e || !p) return fail(TDA_ERR_INVALID, "null argument");
  if (p->struct_size != sizeof(tda_profile)) return fail(TDA_ERR_INVALID, "tda_profile.struct_size mismatch");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  const uint32_t ss = p->struct_size;
  memset(p, 0, sizeof *p);
  p->struct_size = ss;
  for (auto& t : e->timed) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t.a, t.b));
    if (t.kind == 0) {
      p->ms_propose += ms;
      p->n_launch_propose++;
    } else if (t.kind == 1) {
      p->ms_steps += ms;
      p->n_launch_steps++;
    } else {
      p->ms_adapt += ms;
      p->n_launch_adapt++;
    }
  }
  if (!e->timed.empty()) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->timed.front().a, e->timed.back().b));
    p->ms_total = ms;
  }
  return TDA_OK;
}

}  // extern "C"
