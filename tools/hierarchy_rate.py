"""Rate of a Delayed-Acceptance hierarchy whose levels are source-defined (hiprtc) non-linear models: every base step is
three small launches (propose, tda_user_eval, accept), every fine step three more -- no host round trip.  4096 chains,
d = 16, coarse 64 / fine 256 outputs, subchain length 5."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinyda_amd.engine import Engine

SRC = """
__device__ double tda_forward(const double* theta, int dim, int o) {
  double s = 0.0;
  for (int j = 0; j < dim; ++j) s += (0.05 + 0.01 * ((o * 7 + j * 3) %% 11) + %(shift).4f) * theta[j];
  return sin(s) + 0.3 * theta[o %% dim] * theta[(o + 1) %% dim];
}
"""
N, d, L, n_fine = 4096, 16, 5, 60
ms = (64, 256)
rng = np.random.default_rng(0)
e = Engine(N, d, seed=3, n_levels=2)
e.set_prior(np.zeros(d), np.eye(d))
for k, m in enumerate(ms):
    e.set_level_source(k, SRC % dict(shift=0.002 * (1 - k)), 0.1 * rng.standard_normal(m), 0, [0.05 ** 2])
e.set_proposal(1, None, scaling=0.03)
e.set_subchains([L], False)
e.init(0.1 * rng.standard_normal((N, d)))
rows = e.rows_per_level(n_fine)
outs = [(torch.empty((r, N, d), dtype=torch.float64, device="cuda"), torch.empty((r, N, 3), dtype=torch.float64, device="cuda"),
         torch.empty((r, N), dtype=torch.uint8, device="cuda")) for r in rows]
e.run_levels(5, outs)
torch.cuda.synchronize(); t0 = time.perf_counter()
e.run_levels(n_fine, outs)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps(dict(config="DA, source-defined models 64 / 256 outputs, d = 16, pCN, subchain 5", chains=N, fine_iterations=n_fine,
                      coarse_evals_per_s=N * rows[0] / dt, us_per_base_step=dt / rows[0] * 1e6,
                      acceptance=[float(o[2].float().mean().item()) for o in outs])))
e.close()
