set -e
B="/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -w -Itinyda_amd/csrc -Itools"
$B -DTDA_STEP_TRACE -o /tmp/smbt tools/steps_microbench.hip
/tmp/smbt 3 2 1024; /tmp/smbt 3 2 256
