set -e
cd $GRAFT_REPO_ROOT
hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared -w -DTDA_DA_TRACE -Iinclude -Itinyda_amd/csrc -o /tmp/libtda_trace.so tinyda_amd/csrc/tda_engine.hip -L/opt/rocm/lib -lhipfft -lhiprtc
TINYDA_LIB=/tmp/libtda_trace.so python tools/da_trace.py 256 1000
TINYDA_LIB=/tmp/libtda_trace.so python tools/da_trace.py 128 1000
