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

// two standard normals from one counter block
__host__ __device__ inline void normal_pair(uint64_t seed, uint32_t chain, uint32_t step, uint32_t stream,
                                            uint32_t block, double& z0, double& z1) {
  const u32x4 r = philox4x32_10(u32x4{block, step, chain, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u1 = u53(r.x, r.y), u2 = u53(r.z, r.w);
  const double rad = sqrt(-2.0 * log(1.0 - u1));
  double s, c;
#if defined(__HIP_DEVICE_COMPILE__)
  sincospi(2.0 * u2, &s, &c);
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
