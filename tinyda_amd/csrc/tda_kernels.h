// All HIP kernels of the engine (gfx950 only), one header per family:
//   tda_kernels_mh.h      MFMA fragment pipeline, k_mh_steps, k_rng / k_apply / k_propose, k_adapt / k_adapt_block, k_chol
//                         (single-level MH, the hot path)
//   tda_kernels_ml.h      k_ml_steps (DA / MLDA state machine), k_da_steps / k_da_steps_r224 (tda_kernels_da_body.inc),
//                         k_aem_action (dense error model: level decision + tracker vectors)
//   tda_kernels_aemr.h    k_aem_refresh (dense error model: tracker update + factorisation + update_link, one wave per
//                         chain), k_aem_base_steps (its base subchain with one pass over the factor)
//   tda_kernels_aemd.h    k_aemd_* (diagonal error model, any output count)
//   tda_kernels_wide.h    k_wide_adapt / k_wide_apply: single-level chains with 65 .. 128 parameters (with k_mh_steps<128, 4>, k_rng<128>
//                         and k_aem_refresh<8, 1> as the covariance swap)
//   tda_kernels_dreamz.h  k_dreamz_draw / steps / adapt, k_colsum_partial
//   tda_kernels_pooled.h  k_moments_partial / final
//   tda_kernels_ext.h     k_ext_propose / k_ext_accept (batched host-callback forward models)
#pragma once
#include "tda_kernels_mh.h"
#include "tda_kernels_ml.h"
#include "tda_kernels_dreamz.h"
#include "tda_kernels_pooled.h"
#include "tda_kernels_ext.h"
#include "tda_kernels_aemd.h"
#include "tda_kernels_wide.h"
