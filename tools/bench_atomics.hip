// Micro-benchmark: throughput of agent-scope atomics from all workgroups to (a) one word, (b) words on distinct cache lines,
// (c) sc1 flag stores; plus relaxed polls.  Prices the hand-off words of the layer-tail queue (dd_gemm.hip::k_gemm_tail).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_same(int* c, int n) { if (threadIdx.x == 0) for (int i = 0; i < n; ++i) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_same_ret(int* c, int n, int* out) { int s = 0; if (threadIdx.x == 0) { for (int i = 0; i < n; ++i) s += __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); out[blockIdx.x] = s; } }
__global__ void k_lines(int* c, int n, int nl) { if (threadIdx.x == 0) for (int i = 0; i < n; ++i) __hip_atomic_fetch_add(c + 32 * ((blockIdx.x + i) % nl), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ void k_store(int* c, int n) { if (threadIdx.x == 0) for (int i = 0; i < n; ++i) __hip_atomic_store(c + blockIdx.x * n + i, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main() {
  int* c; hipMalloc(&c, 1 << 24); hipMemset(c, 0, 1 << 24);
  int* out; hipMalloc(&out, 1 << 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto fn, long ops) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.1f us  %7.1f ns/op\n", name, ms * 1e3, ms * 1e6 / ops);
  };
  const int G = 1024, N = 8;
  time("same word, 1024 WGs x 8, no return", [&] { hipLaunchKernelGGL(k_same, dim3(G), dim3(256), 0, 0, c, N); }, (long)G * N);
  time("same word, 1024 WGs x 8, value used", [&] { hipLaunchKernelGGL(k_same_ret, dim3(G), dim3(256), 0, 0, c, N, out); }, (long)G * N);
  time("same word, 256 WGs x 8, value used", [&] { hipLaunchKernelGGL(k_same_ret, dim3(256), dim3(256), 0, 0, c, N, out); }, 256L * N);
  time("8 lines, 1024 WGs x 8", [&] { hipLaunchKernelGGL(k_lines, dim3(G), dim3(256), 0, 0, c, N, 8); }, (long)G * N);
  time("64 lines, 1024 WGs x 8", [&] { hipLaunchKernelGGL(k_lines, dim3(G), dim3(256), 0, 0, c, N, 64); }, (long)G * N);
  time("1024 lines, 1024 WGs x 8", [&] { hipLaunchKernelGGL(k_lines, dim3(G), dim3(256), 0, 0, c, N, 1024); }, (long)G * N);
  time("sc1 flag stores, own words, 1024 WGs x 8", [&] { hipLaunchKernelGGL(k_store, dim3(G), dim3(256), 0, 0, c, N); }, (long)G * N);
  time("empty-ish (1 atomic per WG, 1024 lines)", [&] { hipLaunchKernelGGL(k_lines, dim3(G), dim3(256), 0, 0, c, 1, 1024); }, (long)G);
  return 0;
}
