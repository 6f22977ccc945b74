"""Cycle stamps of the pipelined DA kernel (debug build: hipcc ... -DTDA_DA_TRACE -o /tmp/libtda_trace.so; TINYDA_LIB=/tmp/libtda_trace.so)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tinyda_amd import _lib
_lib.LIB_PATH = os.environ["TINYDA_LIB"]
from tinyda_amd.engine import Engine
from tools.bench_configs import levels

N, d = 4096, 64
m0 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
lv = levels((m0, 2048))
e = Engine(N, d, seed=9, n_levels=2)
e.set_prior(np.zeros(d), np.eye(d))
for k, (A, y) in enumerate(lv):
    e.set_level(k, A, y, 0, 0.01)
e.set_proposal(kind=1, scaling=0.02)
e.set_subchains([L])
e.init(None)
e.run_levels(1, None)
out = np.zeros(128 * 8 * 8, dtype=np.int64)
e.lib.tda_debug_da_trace.argtypes = [C.c_void_p]
e.lib.tda_debug_da_trace(out.ctypes.data_as(C.c_void_p))
t = out.reshape(128, 8, 8)
print("m0 = %d; median cycles between stamps, columns = waves 0..7" % m0)
per = np.median(t[11:100, :, 0] - t[10:99, :, 0], axis=0)
print("step period   ", per.astype(int))
for i in range(6):
    print("%d -> %d        " % (i, i + 1), np.median(t[10:100, :, i + 1] - t[10:100, :, i], axis=0).astype(int))
print("6 -> next 0   ", np.median(t[11:100, :, 0] - t[10:99, :, 6], axis=0).astype(int))
e.close()
