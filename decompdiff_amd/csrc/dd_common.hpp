// Shared device helpers for the gfx950 kernels (wave = 64 lanes, 4 SIMD/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/decompdiff_hip.h"
#include "../../include/decompdiff_hip_debug.h"

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

// 1/sqrt(x) of the LayerNorms and the angle normalisation.  Default: v_rsq_f32 (1 ulp).  -DDD_EXACT_MATH=1 builds the
// correctly rounded sqrt + division (and a per-member softmax division) instead:
// `python -m decompdiff_amd.build --exact` -> lib/libdecompdiff_hip_exact.so, selected with DD_HIP_LIB.
__device__ __forceinline__ float dd_rsqrt(float x) {
#if defined(DD_EXACT_MATH) && DD_EXACT_MATH
  return 1.0f / sqrtf(x);
#else
  return __builtin_amdgcn_rsqf(x);
#endif
}

// ---- consumer side of an in-flight hand-off (guide section 6, Guideline 16): ONE lane polls the counter with relaxed
// agent-scope loads, ONE agent-scope acquire drops this CU's stale L1 lines, the barrier releases the workgroup, plain
// loads follow.  The producers (dd_gemm.hip::k_gemm_tail) store write-through (sc1) and bump the counter after their
// stores have drained.  Spins are bounded (~0.1 s): a waiter that gives up records `code` in the error word and goes on.
__device__ __forceinline__ void dd_poll_flag(const int32_t* flags, int idx, int target, int err_idx, int code) {
  if (idx < 0) return;
  // (polls back off: 0.4, 0.8, 1.6 us, ... capped at 3.3 us -- hundreds of waiters polling one line at full rate saturate
  //  its memory channel, which every tile's loads cross)
  for (unsigned spins = 0; __hip_atomic_load(flags + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
    if (spins == 0) __builtin_amdgcn_s_sleep(16);
    else if (spins == 1) __builtin_amdgcn_s_sleep(32);
    else if (spins == 2) __builtin_amdgcn_s_sleep(64);
    else __builtin_amdgcn_s_sleep(127);
    if (spins > (1u << 15)) {
      __hip_atomic_store(const_cast<int32_t*>(flags) + err_idx, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}
__device__ __forceinline__ void dd_wait_flags(const int32_t* flags, int idx0, int n0, int idx1, int n1, int err_idx, int code) {
#if !(defined(DD_DEBUG_OPTIONS) && DD_DEBUG_OPTIONS)
  (void)flags; (void)idx0; (void)n0; (void)idx1; (void)n1; (void)err_idx; (void)code;
  return;                                                // the default library never polls: graph edges only
#endif
  if (flags == nullptr) return;                          // (kernel-uniform)
  if (threadIdx.x == 0) {
    dd_poll_flag(flags, idx0, n0, err_idx, code);
    dd_poll_flag(flags, idx1, n1, err_idx, code + 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// ---- cross-lane primitives on the VALU (DPP + v_permlane{16,32}_swap): no LDS round trips ------------
// DPP controls: quad_perm[1,0,3,2]=0xB1 (lane^1), quad_perm[2,3,0,1]=0x4E (lane^2), row_half_mirror=0x141
// (i -> 7-i within 8 lanes), row_mirror=0x140 (i -> 15-i within 16), row_ror:8=0x128 (lane^8).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), CTRL, 0xF, 0xF, true));
}
// v_permlane16_swap a, b : rows 1,3 of a <-> rows 0,2 of b.  a' + b' is then, on rows 0/2, own a + partner's a
// and, on rows 1/3, own b + partner's b (partner = lane ^ 16): one exchange step of a butterfly, select-free.
__device__ __forceinline__ float swap16_sum(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float swap32_sum(float a, float b) {   // same with lanes 32-63 of a <-> lanes 0-31 of b
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float swap16_max(float a) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return fmaxf(__uint_as_float(r0), __uint_as_float(r1));
}
__device__ __forceinline__ float swap32_max(float a) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return fmaxf(__uint_as_float(r0), __uint_as_float(r1));
}

// broadcast lane `idx` (compile-time constant) of v to the whole wave through an SGPR
__device__ __forceinline__ float lane_bcast(float v, int idx) {
  return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)__float_as_uint(v), idx));
}

// 64-lane all-reduce, fixed (deterministic) combination order.
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  v = swap16_sum(v, v);
  v = swap32_sum(v, v);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  v = swap16_max(v);
  v = swap32_max(v);
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
  float rstd = dd_rsqrt(var + 1e-5f);   // v_rsq_f32, 1 ulp
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
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = swap32_sum(p[i], p[i + 8]);     // lane bit 5 selects p[i] / p[i+8]
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = swap16_sum(a[i], a[i + 4]);     // lane bit 4
  bool hi = (lane & 8) != 0;                                          // lane bit 3: partner = lane ^ 8
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float keep = hi ? b[i + 2] : b[i];
    const float send = hi ? b[i] : b[i + 2];
    c[i] = keep + dpp_mov<0x128>(send);
  }
  // all-reduce over lane bits 0,1 first, so that row_half_mirror (i -> i^7) serves as the lane^4 exchange
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    c[i] += dpp_mov<0xB1>(c[i]);
    c[i] += dpp_mov<0x4E>(c[i]);
  }
  hi = (lane & 4) != 0;                                               // lane bit 2
  const float keep = hi ? c[1] : c[0];
  const float send = hi ? c[0] : c[1];
  return keep + dpp_mov<0x141>(send);
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
