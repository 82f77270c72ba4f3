// Shared device helpers for the gfx950 kernels (wave = 64 lanes, 4 SIMD/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/decompdiff_hip.h"

#define DD_H 128
#define DD_NH 16
#define DD_WAVE 64

#define DD_CHECK_LAUNCH()                              \
  do {                                                 \
    hipError_t e__ = hipGetLastError();                \
    if (e__ != hipSuccess) return DD_ERR_HIP;          \
  } while (0)

namespace dd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// 64-lane butterfly all-reduce (deterministic order).
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Gaussian smearing centres (models/common.py:18) and coefficient -0.5 (common.py:23).
__device__ __forceinline__ float gauss_offset(int g) {
  // 0,1,1.25,...,3 (step .25), 3.5..6 (step .5), 7,8,9,10
  return g == 0 ? 0.f : (g <= 9 ? 1.f + 0.25f * (float)(g - 1) : (g <= 15 ? 3.f + 0.5f * (float)(g - 9) : (float)(g - 9)));
}
__device__ __forceinline__ float gauss_feat(float d, int g) {
  float t = d - gauss_offset(g);
  return expf(-0.5f * (t * t));
}

// LayerNorm(128)+ReLU on a row held as 2 channels per lane (c = 2*lane, 2*lane+1).
// Two-pass mean / biased variance, eps = 1e-5 (torch.nn.LayerNorm default).
__device__ __forceinline__ void ln_relu2(float& a, float& b, float g0, float g1, float be0, float be1) {
  float mean = wave_sum(a + b) * (1.0f / 128.0f);
  float da = a - mean, db = b - mean;
  float var = wave_sum(da * da + db * db) * (1.0f / 128.0f);
  float rstd = 1.0f / sqrtf(var + 1e-5f);
  a = fmaxf(da * rstd * g0 + be0, 0.f);
  b = fmaxf(db * rstd * g1 + be1, 0.f);
}

// Reduce 16 per-lane partials over the 64 lanes.  On return every lane holds the full sum
// of p[h] for h = head_of_lane(lane); 4 consecutive-bit-pattern lanes hold the same head.
__device__ __forceinline__ int head_of_lane(int lane) {
  return ((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
}
__device__ __forceinline__ float reduce16(const float (&p)[16], int lane) {
  float a[8], b[4], c[2];
  bool hi = (lane & 32) != 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float keep = hi ? p[i + 8] : p[i];
    float send = hi ? p[i] : p[i + 8];
    a[i] = keep + __shfl_xor(send, 32, 64);
  }
  hi = (lane & 16) != 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float keep = hi ? a[i + 4] : a[i];
    float send = hi ? a[i] : a[i + 4];
    b[i] = keep + __shfl_xor(send, 16, 64);
  }
  hi = (lane & 8) != 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float keep = hi ? b[i + 2] : b[i];
    float send = hi ? b[i] : b[i + 2];
    c[i] = keep + __shfl_xor(send, 8, 64);
  }
  hi = (lane & 4) != 0;
  float keep = hi ? c[1] : c[0];
  float send = hi ? c[0] : c[1];
  float d = keep + __shfl_xor(send, 4, 64);
  d += __shfl_xor(d, 2, 64);
  d += __shfl_xor(d, 1, 64);
  return d;
}

// Philox4x32-10 counter RNG (production noise path).
struct Philox {
  uint32_t k0, k1;
  __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  __device__ __forceinline__ void gen(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) const {
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
      uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
      uint32_t n0 = hi1 ^ c1 ^ a, n1 = lo1, n2 = hi0 ^ c3 ^ b, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};
__device__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

}  // namespace dd
