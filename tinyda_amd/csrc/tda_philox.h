// Philox4x32-10 counter RNG and the engine's variate definitions (contract: include/tinyda_amd.h,
// "RNG stream").  Host + device; integer part is bit-exact everywhere, the Box-Muller map uses the
// platform's log / sqrt / sincospi.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#else  // plain C++ (the CPU twin of the ABI, oracle/tda_cpu_abi.cpp): the same integer generator and variate maps on the host
#include <cmath>
#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif
#endif
#include <stdint.h>

namespace tda {

enum : uint32_t { STREAM_PROPOSAL = 0, STREAM_ACCEPT = 1, STREAM_INIT = 2 };

struct u32x4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

__host__ __device__ inline u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 product per lane (v_mad_u64_u32 on the device) instead of a mul_hi / mul_lo pair: the integer
    // multiplies run at quarter rate and are half of the generator's cost
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c.x, p1 = (uint64_t)M1 * (uint64_t)c.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    u32x4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += W0;
    k1 += W1;
  }
  return c;
}

__host__ __device__ inline double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

#if defined(__HIP_DEVICE_COMPILE__)
// ln(x) for x in [2^-53, 1] (what Box-Muller needs), the classic fdlibm evaluation (e_log.c): x = 2^k m with m in [sqrt(1/2),
// sqrt(2)), f = m - 1, s = f / (2 + f), ln m = f - (f^2 / 2 - s (f^2 / 2 + R(s^2))), error below 1 ulp.  The device library's log
// carries its intermediate results in double-double arithmetic (49 v_add_f64 and 27 v_fmac_f64 of the 257 vector instructions of a
// normal pair); this form needs a third of that, and the generator -- not hidden under anything in the multi-level pipelines,
// and 30 us longer than the step kernel it runs beside in the headline pipeline -- is bound by exactly this instruction count.
__device__ __forceinline__ double tda_log_unit(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int k = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  k = low ? k - 1 : k;
  const double f = m - 1.0, dk = (double)k;
  const double den = 2.0 + f;
  double r = __builtin_amdgcn_rcp(den);
  r = fma(fma(-den, r, 1.0), r, r);
  r = fma(fma(-den, r, 1.0), r, r);
  double s = f * r;
  s = fma(fma(-den, s, f), r, s);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1, hfsq = 0.5 * f * f;
  return dk * 6.93147180369123816490e-01 - ((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f);
}
// (sin, cos)(2 pi u) for u in [0, 1): quadrant q = rint(4u) and a remainder of at most pi / 4, both exact in fp64, then the fdlibm
// kernels (k_sin.c, k_cos.c) on the remainder; error below 2 ulp, the same class as the device library's sincospi.
__device__ __forceinline__ void tda_sincos_turn(double u, double& sn, double& cs) {
  const double a = 4.0 * u, qf = __builtin_rint(a);
  const double y = (a - qf) * 1.57079632679489661923;
  const int q = (int)qf;
  const double z = y * y;
  const double ps = fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                               -1.98412698298579493134e-04), 8.33333333332248946124e-03);
  const double sy = fma(z * y, fma(z, ps, -1.66666666666666324348e-01), y);
  const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                          2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double cy = 1.0 - fma(0.5, z, -(z * pc));
  const bool swap = q & 1;
  const double s0 = swap ? cy : sy, c0 = swap ? sy : cy;
  sn = (q & 2) ? -s0 : s0;
  cs = ((q + 1) & 2) ? -c0 : c0;
}
#endif

// two standard normals from one counter block
__host__ __device__ inline void normal_pair(uint64_t seed, uint32_t chain, uint32_t step, uint32_t stream,
                                            uint32_t block, double& z0, double& z1) {
  const u32x4 r = philox4x32_10(u32x4{block, step, chain, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u1 = u53(r.x, r.y), u2 = u53(r.z, r.w);
#if defined(__HIP_DEVICE_COMPILE__)
  const double rad = sqrt(-2.0 * tda_log_unit(1.0 - u1));
#else
  const double rad = sqrt(-2.0 * log(1.0 - u1));
#endif
  double s, c;
#if defined(__HIP_DEVICE_COMPILE__)
  tda_sincos_turn(u2, s, c);
#else
  const double ang = 6.283185307179586476925286766559 * u2;
  s = sin(ang);
  c = cos(ang);
#endif
  z0 = rad * c;
  z1 = rad * s;
}

__host__ __device__ inline double accept_uniform(uint64_t seed, uint32_t chain, uint32_t step, uint32_t level) {
  const u32x4 r = philox4x32_10(u32x4{level, step, chain, STREAM_ACCEPT}, (uint32_t)seed, (uint32_t)(seed >> 32));
  return u53(r.x, r.y);
}

}  // namespace tda
