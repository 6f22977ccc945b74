import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import test_gpu_sweep as ts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad, worst = [], {"wide_single": 0.0, "wide_multilevel": 0.0, "deep": 0.0}
for i in range(18, 18 + n):
    c, flips, rel, over = ts._run_single(i, wide=True)
    worst["wide_single"] = max(worst["wide_single"], rel if not ts._small_am(c) else 0.0)
    lim = 1e-9 if ts._small_am(c) else (ts.AM_LOOSE_RTOL if c["kind"].startswith("am") else ts.RTOL)
    if flips or rel > lim: bad.append(("wide_single", i, flips, rel, c))
for i in range(14, 14 + n):
    c, flips, rel = ts._run_multilevel(i, wide=True)
    worst["wide_multilevel"] = max(worst["wide_multilevel"], rel)
    if flips or rel > (ts.AM_LOOSE_RTOL if c["kind"].startswith("am") else ts.RTOL): bad.append(("wide_multilevel", i, flips, rel, c))
for i in range(10, 10 + n):
    c, flips, rel = ts._run_multilevel(i, deep=True)
    worst["deep"] = max(worst["deep"], rel)
    if flips or rel > (ts.AM_LOOSE_RTOL if c["kind"].startswith("am") else ts.RTOL): bad.append(("deep", i, flips, rel, c))
print(json.dumps(dict(n_each=n, configurations=3 * n, failures=[(b[0], b[1], b[2], b[3]) for b in bad], worst_rel=worst)))
