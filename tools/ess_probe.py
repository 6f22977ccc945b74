"""How long does per-chain AdaptiveMetropolis need on C2a before a 20k-iteration window has R-hat < 1.05?  (bench.py set-up study)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import c2_problem, D, SIGMA
from tinyda_amd.engine import Engine
from tinyda_amd import summaries as sm

N = 4096
A, _, y = c2_problem()
dev = torch.device("cuda", 0)
L = 20000
buf = torch.empty((L, N, D), dtype=torch.float64, device=dev)
acc = torch.empty((L, N), dtype=torch.uint8, device=dev)

def mk(kind, C0, **kw):
    e = Engine(N, D, seed=2026, device=0)
    e.set_prior(np.zeros(D), np.eye(D)); e.set_level(0, A, y, 0, SIGMA ** 2); e.set_proposal(kind, C0, **kw)
    return e

pil = mk(0, 1e-4 * np.eye(D))
pil.init(None)
for chunk in range(4):
    pil.run(5000, buf[:5000], None, acc[:5000])
    dd = sm.ess_rhat_device(buf[2500:5000], device=0)
    print("pilot GRW after %d its: acc %.3f max rhat %.3f min ess %.0f" % ((chunk + 1) * 5000, acc[:5000].float().mean().item(), np.nanmax(dd["rhat"]), np.nanmin(dd["ess"])), flush=True)
th, _ = pil.current()
pil.close()
e = mk(2, 1e-4 * np.eye(D), t0=100, period=100)
e.init(th)
done = 0
for burn in [0, 20000, 40000, 60000, 80000]:
    if burn:
        left = 20000
        while left:
            e.run(10000, None, None, None); left -= 10000
        done += 20000
    t = time.perf_counter()
    e.run(L, buf, None, acc)
    dt = time.perf_counter() - t
    done += L
    dd = sm.ess_rhat_device(buf[L // 2:], device=0)
    print("AM window ending at %d its: acc %.3f max rhat %.4f min ess %.0f med ess %.0f  ess/s %.3g" % (done, acc[L // 2:].float().mean().item(), np.nanmax(dd["rhat"]), np.nanmin(dd["ess"]), np.nanmedian(dd["ess"]), np.nanmin(dd["ess"]) / dt), flush=True)
