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

__global__ __launch_bounds__(256) void k_ln_relu_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ stats,
                                                     long rows) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
  const float2 g = *reinterpret_cast<const float2*>(gamma + 2 * lane), b = *reinterpret_cast<const float2*>(beta + 2 * lane);
  for (long r = wave; r < rows; r += n_waves) {
    const float2 v = *reinterpret_cast<const float2*>(x + r * 128 + 2 * lane);
    const float mean = wave_sum(v.x + v.y) * (1.0f / 128.0f);
    const float dx = v.x - mean, dy = v.y - mean;
    const float var = wave_sum(fmaf(dx, dx, dy * dy)) * (1.0f / 128.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    float2 o;
    o.x = fmaxf(fmaf(dx * rstd, g.x, b.x), 0.f);
    o.y = fmaxf(fmaf(dy * rstd, g.y, b.y), 0.f);
    *reinterpret_cast<float2*>(y + r * 128 + 2 * lane) = o;
    if (lane == 0) *reinterpret_cast<float2*>(stats + 2 * r) = make_float2(mean, rstd);
  }
}

__global__ __launch_bounds__(256) void k_ln_relu_bwd(const float* __restrict__ x, const float* __restrict__ stats,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ dyp, float* __restrict__ dxp,
                                                     float* __restrict__ part /*[gridDim.x][2][128]*/, long rows) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long wave = (long)blockIdx.x * 4 + w, n_waves = (long)gridDim.x * 4;
  const float2 g = *reinterpret_cast<const float2*>(gamma + 2 * lane), b = *reinterpret_cast<const float2*>(beta + 2 * lane);
  float2 sg = make_float2(0.f, 0.f), sb = make_float2(0.f, 0.f);
  for (long r = wave; r < rows; r += n_waves) {
    const float2 v = *reinterpret_cast<const float2*>(x + r * 128 + 2 * lane);
    const float2 d = *reinterpret_cast<const float2*>(dyp + r * 128 + 2 * lane);
    const float2 ms = *reinterpret_cast<const float2*>(stats + 2 * r);
    const float hx = (v.x - ms.x) * ms.y, hy = (v.y - ms.x) * ms.y;             // normalised input
    const float zx = fmaf(hx, g.x, b.x) > 0.f ? d.x : 0.f, zy = fmaf(hy, g.y, b.y) > 0.f ? d.y : 0.f;   // through the ReLU
    sg.x = fmaf(zx, hx, sg.x); sg.y = fmaf(zy, hy, sg.y);
    sb.x += zx; sb.y += zy;
    const float ax = zx * g.x, ay = zy * g.y;                                    // gradient w.r.t. the normalised input
    const float s1 = wave_sum(ax + ay) * (1.0f / 128.0f);
    const float s2 = wave_sum(fmaf(ax, hx, ay * hy)) * (1.0f / 128.0f);
    float2 o;
    o.x = ms.y * (ax - s1 - hx * s2);
    o.y = ms.y * (ay - s1 - hy * s2);
    *reinterpret_cast<float2*>(dxp + r * 128 + 2 * lane) = o;
  }
  red[w][2 * lane] = sg.x; red[w][2 * lane + 1] = sg.y;
  red[w][128 + 2 * lane] = sb.x; red[w][128 + 2 * lane + 1] = sb.y;
  __syncthreads();
  const int t = threadIdx.x;                             // 256 threads = [gamma 128 | beta 128], the 4 waves added in order
  part[(long)blockIdx.x * 256 + t] = ((red[0][t] + red[1][t]) + red[2][t]) + red[3][t];
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
  long g = (rows + 3) / 4;                               // one wave per row if the rows are few
  if (g > LN_WG) g = LN_WG;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace
}  // namespace dd

extern "C" size_t dd_ln_relu_scratch_floats(long rows) { return rows > 0 ? (size_t)dd::ln_grid(rows) * 256 : 0; }

extern "C" int dd_ln_relu_forward(const float* x, const float* gamma, const float* beta, float* y, float* stats, long rows, void* stream) {
  if (!x || !gamma || !beta || !y || !stats || rows < 0 || (reinterpret_cast<size_t>(x) & 7) || (reinterpret_cast<size_t>(y) & 7))
    return DD_ERR_BAD_ARG;
  if (rows == 0) return DD_OK;
  hipLaunchKernelGGL(dd::k_ln_relu_fwd, dim3(dd::ln_grid(rows)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, stats, rows);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

extern "C" int dd_ln_relu_backward(const float* x, const float* stats, const float* gamma, const float* beta, const float* dy, float* dx,
                                   float* scratch, float* dgamma, float* dbeta, long rows, void* stream) {
  if (!x || !stats || !gamma || !beta || !dy || !dx || !scratch || !dgamma || !dbeta || rows <= 0 ||
      (reinterpret_cast<size_t>(x) & 7) || (reinterpret_cast<size_t>(dy) & 7) || (reinterpret_cast<size_t>(dx) & 7))
    return DD_ERR_BAD_ARG;
  const int grid = dd::ln_grid(rows);
  hipLaunchKernelGGL(dd::k_ln_relu_bwd, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, stats, gamma, beta, dy, dx, scratch, rows);
  DD_CHECK_LAUNCH();
  hipLaunchKernelGGL(dd::k_ln_relu_bwd_reduce, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, grid, dgamma, dbeta);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
