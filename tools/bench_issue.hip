// Do fp32 MFMA and fp32 VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?  One workgroup of 8 waves per
// CU (waves w and w + 4 share SIMD w % 4); waves 0-3 run `mode_a`, waves 4-7 `mode_b` (0 idle, 1 MFMA f32 16x16x4 with 4
// independent accumulators, 2 v_fma_f32 on 8 independent chains, 3 v_pk_fma_f32, 4 integer multiply-add, 5 MFMA bf16 16x16x32).  If a MFMA wave and a VALU wave on the same
// SIMD take max(t_mfma, t_valu) the pipes are independent; if they take the sum, fp32 MFMA and fp32 VALU share the issue
// slot / datapath and a kernel that needs both is bound by the SUM of its MFMA and VALU cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/bench_issue.hip -o tools/_build/bench_issue && tools/_build/bench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void run_mfma(int iters, float* out) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
  }
  f32x4 s = a0 + a1 + a2 + a3;
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) *out = s[0];
}
__device__ __forceinline__ void run_valu(int iters, float* out) {
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3f + k;
  const float a = 0.999f, b = 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], a, b);      // 16 v_fma_f32 per iteration
  }
  float s = 0;
  for (int k = 0; k < 8; ++k) s += v[k];
  if (s == 123.456f) *out = s;
}
__device__ __forceinline__ void run_pk(int iters, float* out) {
  f32x2 v[8];
  for (int k = 0; k < 8; ++k) v[k] = f32x2{threadIdx.x * 1e-3f + k, threadIdx.x * 2e-3f + k};
  const f32x2 a = {0.999f, 0.998f}, b = {1e-3f, 2e-3f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = __builtin_elementwise_fma(v[k], a, b);   // 16 v_pk_fma_f32 per iteration
  }
  float s = 0;
  for (int k = 0; k < 8; ++k) s += v[k][0] + v[k][1];
  if (s == 123.456f) *out = s;
}
__device__ __forceinline__ void run_int(int iters, float* out) {
  unsigned v[8];
  for (int k = 0; k < 8; ++k) v[k] = threadIdx.x + k;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) { v[k] = v[k] * 3u + 7u; asm volatile("" : "+v"(v[k])); }   // 16 integer multiply-adds per iteration
  }
  unsigned s = 0;
  for (int k = 0; k < 8; ++k) s += v[k];
  if (s == 123456u) *out = (float)s;
}
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
__device__ __forceinline__ void run_mfma_bf16(int iters, float* out) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  bf16x8 x, y;
  for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(threadIdx.x * 1e-3f + k); y[k] = (__bf16)(1.0f + k); }
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, a3, 0, 0, 0);
  }
  f32x4 s = a0 + a1 + a2 + a3;
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) *out = s[0];
}
// the SAME wave, hand-placed (inline asm: the compiler does not interleave them): per iteration 4 independent MFMAs, each
// followed by 4 independent v_fma_f32 (WHAT = 3), or only the MFMAs (1), or only the 16 FMAs (2)
template <int WHAT>
__device__ __forceinline__ void run_mix(int iters, float* out) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  const float a = 0.999f, b = 1e-3f;
#define M(acc, p, q) "v_mfma_f32_16x16x4_f32 " acc ", " p ", " q ", " acc "\n\t"
#define FA "v_fma_f32 %4, %4, %14, %15\n\tv_fma_f32 %5, %5, %14, %15\n\tv_fma_f32 %6, %6, %14, %15\n\tv_fma_f32 %7, %7, %14, %15\n\t"
#define FB "v_fma_f32 %8, %8, %14, %15\n\tv_fma_f32 %9, %9, %14, %15\n\tv_fma_f32 %10, %10, %14, %15\n\tv_fma_f32 %11, %11, %14, %15\n\t"
  for (int i = 0; i < iters; ++i) {
    if (WHAT == 3)
      asm volatile(M("%0", "%12", "%13") FA M("%1", "%13", "%12") FB M("%2", "%12", "%12") FA M("%3", "%13", "%13") FB
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
                   : "v"(x), "v"(y), "v"(a), "v"(b));
    else if (WHAT == 1)
      asm volatile(M("%0", "%12", "%13") M("%1", "%13", "%12") M("%2", "%12", "%12") M("%3", "%13", "%13")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
                   : "v"(x), "v"(y), "v"(a), "v"(b));
    else
      asm volatile(FA FB FA FB
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7)
                   : "v"(x), "v"(y), "v"(a), "v"(b));
  }
#undef M
#undef FA
#undef FB
  f32x4 s = a0 + a1 + a2 + a3;
  const float t = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  if (s[0] + s[1] + s[2] + s[3] + t == 123.456f) *out = s[0];
}
__global__ __launch_bounds__(512) void k(int mode_a, int mode_b, int iters, float* out) {
  const int mode = (threadIdx.x >> 6) < 4 ? mode_a : mode_b;
  if (mode == 1) run_mfma(iters, out);
  else if (mode == 2) run_valu(iters, out);
  else if (mode == 3) run_pk(iters, out);
  else if (mode == 4) run_int(iters, out);
  else if (mode == 5) run_mfma_bf16(iters, out);
  else if (mode == 6) run_mix<3>(iters, out);
  else if (mode == 7) run_mix<1>(iters, out);
  else if (mode == 8) run_mix<2>(iters, out);
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* nm[9] = {"idle", "mfma", "fma ", "pkfma", "int ", "mfma_bf16", "asm: 4 x (mfma, 4 fma) in ONE wave", "asm: 4 mfma", "asm: 16 fma"};
  for (int ma = 0; ma < 9; ++ma)
    for (int mb = 0; mb < 9; ++mb) {
      if ((ma >= 6 || mb >= 6) && !(ma == 0 || ma == mb)) continue;   // asm modes: one wave per SIMD, or two of the same
      if (ma == 0 && mb == 0) continue;
      if (ma > mb) continue;                            // (symmetric: one order is enough)
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, ma, mb, 100, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, ma, mb, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // per iteration: 4 MFMAs (4 x 2048 flop x 64... per wave) or 16 FMAs per lane
      printf("waves 0-3 %s | waves 4-7 %s : %8.3f ms  (%.1f cycles per iteration at 2.4 GHz)\n", nm[ma], nm[mb], ms, ms * 1e-3 * 2.4e9 / iters);
    }
  return 0;
}
