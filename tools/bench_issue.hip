// Do fp32 MFMA and fp32 VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?  One workgroup of 8 waves per
// CU (waves w and w + 4 share SIMD w % 4); waves 0-3 run `mode_a`, waves 4-7 `mode_b` (0 idle, 1 MFMA f32 16x16x4 with 4
// independent accumulators, 2 v_fma_f32 on 8 independent chains, 3 v_pk_fma_f32).  If a MFMA wave and a VALU wave on the same
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
__global__ __launch_bounds__(512) void k(int mode_a, int mode_b, int iters, float* out) {
  const int mode = (threadIdx.x >> 6) < 4 ? mode_a : mode_b;
  if (mode == 1) run_mfma(iters, out);
  else if (mode == 2) run_valu(iters, out);
  else if (mode == 3) run_pk(iters, out);
}
int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* nm[4] = {"idle", "mfma", "fma ", "pkfma"};
  for (int ma = 0; ma < 4; ++ma)
    for (int mb = 0; mb < 4; ++mb) {
      if (ma == 0 && mb == 0) continue;
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
