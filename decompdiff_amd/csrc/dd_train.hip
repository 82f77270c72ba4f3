// Training-step kernels: LayerNorm(128) + ReLU of the reference's MLPs (models/common.py:85-105,
// Linear -> LayerNorm -> ReLU -> Linear) as ONE forward and ONE backward kernel.
//
// The eager autograd chain is layer_norm + relu forward (2 launches, the activation written twice) and threshold_backward +
// native_layer_norm_backward (input gradient, then a second pass over x and dy for the gamma / beta partial sums) + a reduce:
// 250 MB of HBM traffic per backward call on the 97 440 triplet rows of a C-small batch of 4, 182 calls per step.  Here a wave owns
// a row at a time (lane l holds channels 2l, 2l+1: one 512-byte load per tensor and row); the forward writes y and (mean, rstd);
// the backward recomputes the ReLU mask from x, forms dx and keeps the gamma / beta sums of its rows in registers; workgroups write
// their partial sums to a scratch buffer and a second kernel adds them in a fixed order (no atomics: bitwise reproducible).
#include "dd_common.hpp"
#include "dd_kernels.hpp"

namespace dd {
namespace {

constexpr int LN_WG = 1024;                              // partial-sum slabs of the backward (<= workgroups launched)

// sum over the 16 lanes of a DPP row (the lanes of one input row)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// 16 lanes per row (lane j of the group holds channels 8j .. 8j+7: two 16-byte loads, a row is one contiguous 512-byte access of
// the group), four rows in flight per wave; mean / variance / the two backward sums are 4-step reductions inside a DPP row.
__global__ __launch_bounds__(256) void k_ln_relu_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ stats,
                                                     long rows) {
  const int j = threadIdx.x & 15;
  const long grp = (long)blockIdx.x * 16 + (threadIdx.x >> 4), n_grp = (long)gridDim.x * 16;
  float g[8], b[8];
  {
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + 8 * j), g1 = *reinterpret_cast<const float4*>(gamma + 8 * j + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + 8 * j), b1 = *reinterpret_cast<const float4*>(beta + 8 * j + 4);
    g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
    b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
  }
  // (every group of a wave runs the same number of trips: the DPP reductions need all lanes of the row active, and a wave's
  //  groups differ only in whether their last row exists)
  for (long r0 = (long)blockIdx.x * 16 + (threadIdx.x >> 6) * 4; r0 < rows; r0 += n_grp) {
    const long r = r0 + ((threadIdx.x >> 4) & 3);
    const bool live = r < rows;
    float v[8];
    if (live) {
      const float4 a0 = *reinterpret_cast<const float4*>(x + r * 128 + 8 * j), a1 = *reinterpret_cast<const float4*>(x + r * 128 + 8 * j + 4);
      v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) s += v[c];
    const float mean = row16_sum(s) * (1.0f / 128.0f);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) { v[c] -= mean; q = fmaf(v[c], v[c], q); }
    const float rstd = 1.0f / sqrtf(row16_sum(q) * (1.0f / 128.0f) + 1e-5f);
    if (live) {
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = fmaxf(fmaf(v[c] * rstd, g[c], b[c]), 0.f);
      *reinterpret_cast<float4*>(y + r * 128 + 8 * j) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(y + r * 128 + 8 * j + 4) = make_float4(o[4], o[5], o[6], o[7]);
      if (j == 0) *reinterpret_cast<float2*>(stats + 2 * r) = make_float2(mean, rstd);
    }
  }
  (void)grp;
}

__global__ __launch_bounds__(256) void k_ln_relu_bwd(const float* __restrict__ x, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ dyp, float* __restrict__ dxp,
                                                     float* __restrict__ part /*[gridDim.x][2][128]*/, long rows) {
  __shared__ float red[16][256];
  const int j = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const long n_grp = (long)gridDim.x * 16;
  float g[8], b[8], sg[8], sb[8];
  {
    const float4 g0 = *reinterpret_cast<const float4*>(gamma + 8 * j), g1 = *reinterpret_cast<const float4*>(gamma + 8 * j + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(beta + 8 * j), b1 = *reinterpret_cast<const float4*>(beta + 8 * j + 4);
    g[0] = g0.x; g[1] = g0.y; g[2] = g0.z; g[3] = g0.w; g[4] = g1.x; g[5] = g1.y; g[6] = g1.z; g[7] = g1.w;
    b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) { sg[c] = 0.f; sb[c] = 0.f; }
  for (long r0 = (long)blockIdx.x * 16 + (threadIdx.x >> 6) * 4; r0 < rows; r0 += n_grp) {
    const long r = r0 + (grp & 3);
    const bool live = r < rows;
    float h[8], a[8];
    float mean = 0.f, rstd = 0.f;
    if (live) {
      const float4 a0 = *reinterpret_cast<const float4*>(x + r * 128 + 8 * j), a1 = *reinterpret_cast<const float4*>(x + r * 128 + 8 * j + 4);
      const float4 d0 = *reinterpret_cast<const float4*>(dyp + r * 128 + 8 * j), d1 = *reinterpret_cast<const float4*>(dyp + r * 128 + 8 * j + 4);
      const float2 ms = *reinterpret_cast<const float2*>(stats + 2 * r);
      mean = ms.x; rstd = ms.y;
      h[0] = a0.x; h[1] = a0.y; h[2] = a0.z; h[3] = a0.w; h[4] = a1.x; h[5] = a1.y; h[6] = a1.z; h[7] = a1.w;
      a[0] = d0.x; a[1] = d0.y; a[2] = d0.z; a[3] = d0.w; a[4] = d1.x; a[5] = d1.y; a[6] = d1.z; a[7] = d1.w;
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) { h[c] = 0.f; a[c] = 0.f; }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      h[c] = (h[c] - mean) * rstd;                                            // normalised input
      const float z = fmaf(h[c], g[c], b[c]) > 0.f ? a[c] : 0.f;              // gradient through the ReLU
      sg[c] = fmaf(z, h[c], sg[c]);
      sb[c] += z;
      a[c] = z * g[c];                                                        // gradient w.r.t. the normalised input
      s1 += a[c];
      s2 = fmaf(a[c], h[c], s2);
    }
    s1 = row16_sum(s1) * (1.0f / 128.0f);
    s2 = row16_sum(s2) * (1.0f / 128.0f);
    if (live) {
      float o[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) o[c] = rstd * (a[c] - s1 - h[c] * s2);
      *reinterpret_cast<float4*>(dxp + r * 128 + 8 * j) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(dxp + r * 128 + 8 * j + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
  // the 16 row groups of the workgroup hold partial sums for the same 128 channels: added in group order
#pragma unroll
  for (int c = 0; c < 8; ++c) { red[grp][8 * j + c] = sg[c]; red[grp][128 + 8 * j + c] = sb[c]; }
  __syncthreads();
  const int t = threadIdx.x;                             // 256 threads = [gamma 128 | beta 128]
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) s += red[k][t];
  part[(long)blockIdx.x * 256 + t] = s;
}

__global__ __launch_bounds__(256) void k_ln_relu_bwd_reduce(const float* __restrict__ part, int slabs, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta) {
  const int t = threadIdx.x;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;          // four chains, slabs dealt round-robin, added in a fixed order
  int sl = 0;
  for (; sl + 3 < slabs; sl += 4) {
    s0 += part[(long)sl * 256 + t]; s1 += part[(long)(sl + 1) * 256 + t];
    s2 += part[(long)(sl + 2) * 256 + t]; s3 += part[(long)(sl + 3) * 256 + t];
  }
  for (; sl < slabs; ++sl) s0 += part[(long)sl * 256 + t];
  const float s = (s0 + s1) + (s2 + s3);
  if (t < 128) dgamma[t] = s;
  else dbeta[t - 128] = s;
}

inline int ln_grid(long rows) {
  long g = (rows + 15) / 16;                             // sixteen rows per workgroup and trip
  if (g > LN_WG) g = LN_WG;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace
}  // namespace dd

extern "C" size_t dd_ln_relu_scratch_floats(long rows) { return rows > 0 ? (size_t)dd::ln_grid(rows) * 256 : 0; }

extern "C" int dd_ln_relu_forward(const float* x, const float* gamma, const float* beta, float* y, float* stats, long rows, void* stream) {
  // (the kernels read x / y / gamma / beta as float4 and stats as float2)
  auto mis = [](const void* p, size_t m) { return (reinterpret_cast<size_t>(p) & m) != 0; };
  if (!x || !gamma || !beta || !y || !stats || rows < 0 || mis(x, 15) || mis(y, 15) || mis(gamma, 15) || mis(beta, 15) || mis(stats, 7))
    return DD_ERR_BAD_ARG;
  if (rows == 0) return DD_OK;
  hipLaunchKernelGGL(dd::k_ln_relu_fwd, dim3(dd::ln_grid(rows)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, stats, rows);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

extern "C" int dd_ln_relu_backward(const float* x, const float* stats, const float* gamma, const float* beta, const float* dy, float* dx,
                                   float* scratch, float* dgamma, float* dbeta, long rows, void* stream) {
  auto mis = [](const void* p, size_t m) { return (reinterpret_cast<size_t>(p) & m) != 0; };
  if (!x || !stats || !gamma || !beta || !dy || !dx || !scratch || !dgamma || !dbeta || rows < 0 || mis(x, 15) || mis(dy, 15) ||
      mis(dx, 15) || mis(gamma, 15) || mis(beta, 15) || mis(stats, 7))
    return DD_ERR_BAD_ARG;
  if (rows == 0) return DD_OK;                           // (as the forward: nothing to do; dgamma / dbeta are left untouched)
  const int grid = dd::ln_grid(rows);
  hipLaunchKernelGGL(dd::k_ln_relu_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, stats, gamma, beta, dy, dx, scratch, rows);
  DD_CHECK_LAUNCH();
  hipLaunchKernelGGL(dd::k_ln_relu_bwd_reduce, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, grid, dgamma, dbeta);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
