// Does a v_mul_f64 / v_add_f64 whose two VGPR sources lie in the same register-bank pair issue slower on gfx950?  (debug tool)
// 64 independent instructions per trip with explicit registers: destination v[D:D+1], sources v[A:A+1], v[B:B+1].
//   kind 0: sources (A mod 4, B mod 4) = (0, 2)   -- different bank pairs, destination = pair of A
//   kind 1: sources (0, 0)                         -- same bank pair
//   kind 2: sources (0, 2), destination on the pair of neither?  (there are only two pairs: destination alternates)
//   kind 3: one VGPR source + one SGPR source
// Build on the GPU box: hipcc -O3 -std=c++17 --offload-arch=gfx950 -w -o /tmp/vbp tools/vgpr_bank_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int ITER = 20000;

#define R2(a) a a
#define R4(a) R2(a) R2(a)
#define R8(a) R4(a) R4(a)

// 8 instructions on distinct destinations; sources chosen per kind.  Registers v[32..127] are clobbered.
#define BLK_DIFF \
  "v_mul_f64 v[32:33], v[64:65], v[98:99]\n v_mul_f64 v[36:37], v[68:69], v[102:103]\n v_add_f64 v[40:41], v[72:73], v[106:107]\n v_mul_f64 v[44:45], v[76:77], v[110:111]\n" \
  "v_mul_f64 v[48:49], v[80:81], v[114:115]\n v_add_f64 v[52:53], v[84:85], v[118:119]\n v_mul_f64 v[56:57], v[88:89], v[122:123]\n v_add_f64 v[60:61], v[92:93], v[126:127]\n"
#define BLK_SAME \
  "v_mul_f64 v[32:33], v[64:65], v[96:97]\n v_mul_f64 v[36:37], v[68:69], v[100:101]\n v_add_f64 v[40:41], v[72:73], v[104:105]\n v_mul_f64 v[44:45], v[76:77], v[108:109]\n" \
  "v_mul_f64 v[48:49], v[80:81], v[112:113]\n v_add_f64 v[52:53], v[84:85], v[116:117]\n v_mul_f64 v[56:57], v[88:89], v[120:121]\n v_add_f64 v[60:61], v[92:93], v[124:125]\n"
#define BLK_DIFF_DST2 \
  "v_mul_f64 v[34:35], v[64:65], v[98:99]\n v_mul_f64 v[38:39], v[68:69], v[102:103]\n v_add_f64 v[42:43], v[72:73], v[106:107]\n v_mul_f64 v[46:47], v[76:77], v[110:111]\n" \
  "v_mul_f64 v[50:51], v[80:81], v[114:115]\n v_add_f64 v[54:55], v[84:85], v[118:119]\n v_mul_f64 v[58:59], v[88:89], v[122:123]\n v_add_f64 v[62:63], v[92:93], v[126:127]\n"
#define BLK_SGPR \
  "v_mul_f64 v[32:33], v[64:65], s[20:21]\n v_mul_f64 v[36:37], v[68:69], s[20:21]\n v_add_f64 v[40:41], v[72:73], s[20:21]\n v_mul_f64 v[44:45], v[76:77], s[20:21]\n" \
  "v_mul_f64 v[48:49], v[80:81], s[20:21]\n v_add_f64 v[52:53], v[84:85], s[20:21]\n v_mul_f64 v[56:57], v[88:89], s[20:21]\n v_add_f64 v[60:61], v[92:93], s[20:21]\n"
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","s20","s21"

template <int KIND>
__global__ void __launch_bounds__(64) k_probe(long long* cyc) {
  // sources: whatever the registers hold (denormals are flushed? fp64 denormals are not: initialise to 1.0)
  asm volatile("s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3ff00000\n" ::: "s20", "s21");
#pragma unroll
  for (int i = 0; i < 1; ++i)
    asm volatile(
        "v_mov_b32 v64, 0\n v_mov_b32 v65, 0x3ff00000\n v_mov_b32 v68, 0\n v_mov_b32 v69, 0x3ff00000\n v_mov_b32 v72, 0\n v_mov_b32 v73, 0x3ff00000\n v_mov_b32 v76, 0\n v_mov_b32 v77, 0x3ff00000\n"
        "v_mov_b32 v80, 0\n v_mov_b32 v81, 0x3ff00000\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0x3ff00000\n v_mov_b32 v88, 0\n v_mov_b32 v89, 0x3ff00000\n v_mov_b32 v92, 0\n v_mov_b32 v93, 0x3ff00000\n"
        ::: "v64","v65","v68","v69","v72","v73","v76","v77","v80","v81","v84","v85","v88","v89","v92","v93");
  asm volatile(
      "v_mov_b32 v96, 0\n v_mov_b32 v97, 0x3ff00000\n v_mov_b32 v98, 0\n v_mov_b32 v99, 0x3ff00000\n v_mov_b32 v100, 0\n v_mov_b32 v101, 0x3ff00000\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0x3ff00000\n"
      "v_mov_b32 v104, 0\n v_mov_b32 v105, 0x3ff00000\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0x3ff00000\n v_mov_b32 v108, 0\n v_mov_b32 v109, 0x3ff00000\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0x3ff00000\n"
      "v_mov_b32 v112, 0\n v_mov_b32 v113, 0x3ff00000\n v_mov_b32 v114, 0\n v_mov_b32 v115, 0x3ff00000\n v_mov_b32 v116, 0\n v_mov_b32 v117, 0x3ff00000\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0x3ff00000\n"
      "v_mov_b32 v120, 0\n v_mov_b32 v121, 0x3ff00000\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0x3ff00000\n v_mov_b32 v124, 0\n v_mov_b32 v125, 0x3ff00000\n v_mov_b32 v126, 0\n v_mov_b32 v127, 0x3ff00000\n"
      ::: "v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119",
          "v120","v121","v122","v123","v124","v125","v126","v127");
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) asm volatile(R8(BLK_DIFF) ::: CLOB);
    else if (KIND == 1) asm volatile(R8(BLK_SAME) ::: CLOB);
    else if (KIND == 2) asm volatile(R8(BLK_DIFF_DST2) ::: CLOB);
    else asm volatile(R8(BLK_SGPR) ::: CLOB);
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  long long* cyc;
  const int maxb = 256 * 4 * 4;
  CK(hipMalloc(&cyc, (size_t)maxb * 8));
  const char* names[] = {"two VGPR sources, different bank pairs (A, D even quad; B odd)", "two VGPR sources, same bank pair", "different pairs, destination on B's pair", "one VGPR + one SGPR source"};
  for (int kind = 0; kind < 4; ++kind)
    for (int wps : {1, 2, 4}) {
      const int nb = 256 * 4 * wps;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        switch (kind) {
          case 0: hipLaunchKernelGGL(k_probe<0>, dim3(nb), dim3(64), 0, 0, cyc); break;
          case 1: hipLaunchKernelGGL(k_probe<1>, dim3(nb), dim3(64), 0, 0, cyc); break;
          case 2: hipLaunchKernelGGL(k_probe<2>, dim3(nb), dim3(64), 0, 0, cyc); break;
          default: hipLaunchKernelGGL(k_probe<3>, dim3(nb), dim3(64), 0, 0, cyc); break;
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      // instructions per SIMD = wps * ITER * 64; time per instruction in ns
      const double ns = best * 1e6 / ((double)wps * ITER * 64);
      printf("%-70s waves/SIMD %d: %.3f ms, %.2f ns per instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", names[kind], wps, best, ns, ns * 2.4);
    }
  return 0;
}
