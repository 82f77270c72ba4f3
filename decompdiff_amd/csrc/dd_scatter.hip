// Op-level message passing: the torch_scatter call pairs of the reference as stand-alone kernels (SURVEY.md 8b,
// "op-level boundary").  The sampling loop itself runs the fused kernels of dd_attention2.hip, where q / k / v never
// touch HBM; these are for hosts that keep the reference's Python layers and only swap
//
//     alpha = scatter_softmax((q[dst] * k / sqrt(d)).sum(-1), dst, dim=0)
//     out   = scatter_sum(alpha.unsqueeze(-1) * v, dst, dim=0, dim_size=N)
//
// (models/encoders/uni_transformer_edge.py:63-68 node, :158-164 triplet, :205-211 coordinates), and they are the kernels the
// HBM roofline of SURVEY.md 8d(i) is quoted on: every byte of q, k, v, e_w and the segment pointers is read once, the
// output written once.
//
// Edges must be grouped by destination (knn_graph, the dst-major bond list and the SparseTensor triplets all are):
// seg_ptr[s] .. seg_ptr[s+1] are the edges of destination s.  One wavefront owns one destination; lane l holds
// channels 2l, 2l+1 (head l / 4), so a k or v row is one 512-byte wave load.  Softmax is the single-pass form
// (running maximum, rescaled running sum) in edge order; 4 edges are in flight per wave.
#include "dd_common.hpp"
#include "dd_kernels.hpp"

namespace dd {

namespace {

constexpr int UNROLL = 4;

// sum over the 4 lanes of a head (lanes 4h .. 4h+3)
__device__ __forceinline__ float head_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  return v;
}

template <bool POS>
__global__ __launch_bounds__(256) void k_attn_aggregate(const float* __restrict__ q, int q_per_edge, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ e_w,
                                                        const float* __restrict__ rel_x, const int32_t* __restrict__ seg_ptr,
                                                        int n_seg, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  // the segment and its edge range are wave-uniform: kept in SGPRs, so e_w / rel_x / seg_ptr become scalar loads
  const int seg = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (seg >= n_seg) return;
  const int e0 = __builtin_amdgcn_readfirstlane(seg_ptr[seg]), e1 = __builtin_amdgcn_readfirstlane(seg_ptr[seg + 1]);
  if (e1 <= e0) {                                        // scatter_sum leaves untouched rows at zero
    if (POS) { if (lane < 3) out[(long)seg * 3 + lane] = 0.f; }
    else *reinterpret_cast<float2*>(out + (long)seg * 128 + 2 * lane) = make_float2(0.f, 0.f);
    return;
  }
  const float2 qv = *reinterpret_cast<const float2*>(q + (long)(q_per_edge ? e0 : seg) * 128 + 2 * lane);
  const float scale = 0.35355339059327373f;              // 1 / sqrt(8)
  float mx = -INFINITY, den = 0.f;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;                    // !POS: a0, a1 = channels 2l, 2l+1; POS: xyz of head l/4
  const int head = lane >> 2;
  for (int e = e0; e < e1; e += UNROLL) {
    float2 kk[UNROLL], vv[UNROLL];
    float w[UNROLL], r[UNROLL][3];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long ee = e + u < e1 ? e + u : e1 - 1;       // clamped: the tail replays the last edge and is masked below
      kk[u] = *reinterpret_cast<const float2*>(k + ee * 128 + 2 * lane);
      if (POS) {
        vv[u].x = v[ee * 16 + head];
        r[u][0] = rel_x[ee * 3]; r[u][1] = rel_x[ee * 3 + 1]; r[u][2] = rel_x[ee * 3 + 2];
      } else {
        vv[u] = *reinterpret_cast<const float2*>(v + ee * 128 + 2 * lane);
      }
      w[u] = e_w ? e_w[ee] : 1.0f;
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if (e + u >= e1) break;
      const float s = head_sum(fmaf(qv.y, kk[u].y, qv.x * kk[u].x)) * scale;
      const float mn = fmaxf(mx, s);
      const float corr = __expf(mx - mn), p = __expf(s - mn);    // (mx = -inf on the first edge: corr = 0)
      den = fmaf(den, corr, p);
      if (POS) {
        const float pv = p * (vv[u].x * w[u]);
        a0 = fmaf(a0, corr, pv * r[u][0]); a1 = fmaf(a1, corr, pv * r[u][1]); a2 = fmaf(a2, corr, pv * r[u][2]);
      } else {
        a0 = fmaf(a0, corr, p * (vv[u].x * w[u])); a1 = fmaf(a1, corr, p * (vv[u].y * w[u]));
      }
      mx = mn;
    }
  }
  const float inv = 1.0f / den;
  if (POS) {
    // mean over the 16 heads: one lane per head carries the head's vector
    const bool lead = (lane & 3) == 0;
    const float x = wave_sum(lead ? a0 * inv : 0.f), y = wave_sum(lead ? a1 * inv : 0.f), z = wave_sum(lead ? a2 * inv : 0.f);
    if (lane == 0) { out[(long)seg * 3] = x * 0.0625f; out[(long)seg * 3 + 1] = y * 0.0625f; out[(long)seg * 3 + 2] = z * 0.0625f; }
  } else {
    *reinterpret_cast<float2*>(out + (long)seg * 128 + 2 * lane) = make_float2(a0 * inv, a1 * inv);
  }
}


// ---- stand-alone torch_scatter drop-ins: scatter_sum / scatter_mean / scatter_min / scatter_max and
// scatter_softmax over dim 0 of a [E, F] fp32 tensor whose rows are grouped by destination (CSR segments).  Call sites in
// the reference: scatter_softmax / scatter_sum in the three attention layers (uni_transformer_edge.py:64,68,160,164,205,209),
// scatter_mean in center_pos (decompdiff.py:25), scatter_min in the arm-scaffold drift (guidance_funcs.py:52).
// One wavefront per (segment, 64-feature chunk).  F <= 32: the wave also splits the segment's rows over 64 / Fp lane
// groups (Fp = F rounded up to a power of two), combined at the end in a fixed butterfly order -> deterministic.
enum { OP_SUM = 0, OP_MEAN = 1, OP_MIN = 2, OP_MAX = 3 };

__device__ __forceinline__ int seg_fp(int F) { int p = 1; while (p < F && p < 64) p <<= 1; return p; }

template <int OP>
__global__ __launch_bounds__(256) void k_segment_reduce(const float* __restrict__ src, const int32_t* __restrict__ seg_ptr, int n_seg,
                                                        int F, int n_chunk, float* __restrict__ out, int64_t* __restrict__ arg_out,
                                                        long E) {
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long)n_seg * n_chunk) return;
  const int seg = (int)(w / n_chunk), chunk = (int)(w % n_chunk);
  const int e0 = seg_ptr[seg], e1 = seg_ptr[seg + 1];
  const int Fp = seg_fp(F), G = 64 / Fp;
  const int f = chunk * 64 + (lane & (Fp - 1)), g = lane / Fp;
  const bool fv = f < F;
  float acc = OP == OP_MIN ? INFINITY : (OP == OP_MAX ? -INFINITY : 0.f);
  long arg = E;
  for (int e = e0 + g; e < e1; e += G) {
    const float v = fv ? src[(long)e * F + f] : 0.f;
    if (OP == OP_MIN) { if (v < acc) { acc = v; arg = e; } }
    else if (OP == OP_MAX) { if (v > acc) { acc = v; arg = e; } }
    else acc += v;
  }
  for (int off = Fp; off < 64; off <<= 1) {              // combine the lane groups (ties: the smaller row index wins)
    const float o = __shfl_xor(acc, off);
    const long oa = __shfl_xor(arg, off);
    if (OP == OP_MIN) { if (o < acc || (o == acc && oa < arg)) { acc = o; arg = oa; } }
    else if (OP == OP_MAX) { if (o > acc || (o == acc && oa < arg)) { acc = o; arg = oa; } }
    else acc += o;
  }
  if (!fv || g != 0) return;
  if (e1 <= e0) { acc = 0.f; arg = E; }                  // torch_scatter: untouched rows are 0, their arg = src.size(dim)
  if (OP == OP_MEAN && e1 > e0) acc /= (float)(e1 - e0);
  out[(long)seg * F + f] = acc;
  if ((OP == OP_MIN || OP == OP_MAX) && arg_out) arg_out[(long)seg * F + f] = arg;
}

__global__ __launch_bounds__(256) void k_segment_softmax(const float* __restrict__ src, const int32_t* __restrict__ seg_ptr, int n_seg,
                                                         int F, int n_chunk, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long)n_seg * n_chunk) return;
  const int seg = (int)(w / n_chunk), chunk = (int)(w % n_chunk);
  const int e0 = seg_ptr[seg], e1 = seg_ptr[seg + 1];
  const int Fp = seg_fp(F), G = 64 / Fp;
  const int f = chunk * 64 + (lane & (Fp - 1)), g = lane / Fp;
  const bool fv = f < F;
  float mx = -INFINITY;
  for (int e = e0 + g; e < e1; e += G) mx = fmaxf(mx, fv ? src[(long)e * F + f] : 0.f);
  for (int off = Fp; off < 64; off <<= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
  for (int e = e0 + g; e < e1; e += G) sum += fv ? expf(src[(long)e * F + f] - mx) : 0.f;
  for (int off = Fp; off < 64; off <<= 1) sum += __shfl_xor(sum, off);
  if (!fv) return;
  for (int e = e0 + g; e < e1; e += G) out[(long)e * F + f] = expf(src[(long)e * F + f] - mx) / sum;
}

}  // namespace

}  // namespace dd

extern "C" int dd_attn_aggregate_node(const float* q, int q_per_edge, const float* k, const float* v, const float* e_w,
                                      const int32_t* seg_ptr, int n_seg, float* out, void* stream) {
  if (!q || !k || !v || !seg_ptr || !out || n_seg < 0) return DD_ERR_BAD_ARG;
  if (n_seg == 0) return DD_OK;
  hipLaunchKernelGGL(dd::k_attn_aggregate<false>, dim3((n_seg + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, q_per_edge, k, v, e_w,
                     nullptr, seg_ptr, n_seg, out);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

extern "C" int dd_attn_aggregate_triplet(const float* q, const float* k, const float* v, const int32_t* seg_ptr, int n_seg,
                                         float* out, void* stream) {
  return dd_attn_aggregate_node(q, 1, k, v, nullptr, seg_ptr, n_seg, out, stream);
}

extern "C" int dd_attn_aggregate_pos(const float* q, const float* k, const float* v16, const float* e_w, const float* rel_x,
                                     const int32_t* seg_ptr, int n_seg, float* out, void* stream) {
  if (!q || !k || !v16 || !rel_x || !seg_ptr || !out || n_seg < 0) return DD_ERR_BAD_ARG;
  if (n_seg == 0) return DD_OK;
  hipLaunchKernelGGL(dd::k_attn_aggregate<true>, dim3((n_seg + 3) / 4), dim3(256), 0, (hipStream_t)stream, q, 0, k, v16, e_w, rel_x,
                     seg_ptr, n_seg, out);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

extern "C" int dd_segment_reduce(const float* src, const int32_t* seg_ptr, int n_seg, int F, int op, long E, float* out,
                                 int64_t* arg_out, void* stream) {
  if (!seg_ptr || !out || n_seg < 0 || F <= 0 || E < 0 || (E > 0 && !src) || op < 0 || op > 3) return DD_ERR_BAD_ARG;
  if (n_seg == 0) return DD_OK;
  const int n_chunk = (F + 63) / 64;
  const dim3 grid((unsigned)(((long)n_seg * n_chunk + 3) / 4)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (op) {
    case dd::OP_SUM: hipLaunchKernelGGL(dd::k_segment_reduce<dd::OP_SUM>, grid, block, 0, st, src, seg_ptr, n_seg, F, n_chunk, out, arg_out, E); break;
    case dd::OP_MEAN: hipLaunchKernelGGL(dd::k_segment_reduce<dd::OP_MEAN>, grid, block, 0, st, src, seg_ptr, n_seg, F, n_chunk, out, arg_out, E); break;
    case dd::OP_MIN: hipLaunchKernelGGL(dd::k_segment_reduce<dd::OP_MIN>, grid, block, 0, st, src, seg_ptr, n_seg, F, n_chunk, out, arg_out, E); break;
    default: hipLaunchKernelGGL(dd::k_segment_reduce<dd::OP_MAX>, grid, block, 0, st, src, seg_ptr, n_seg, F, n_chunk, out, arg_out, E); break;
  }
  DD_CHECK_LAUNCH();
  return DD_OK;
}

extern "C" int dd_segment_softmax(const float* src, const int32_t* seg_ptr, int n_seg, int F, float* out, void* stream) {
  if (!seg_ptr || n_seg < 0 || F <= 0) return DD_ERR_BAD_ARG;
  if (n_seg == 0) return DD_OK;
  if (!src || !out) return DD_ERR_BAD_ARG;
  const int n_chunk = (F + 63) / 64;
  hipLaunchKernelGGL(dd::k_segment_softmax, dim3((unsigned)(((long)n_seg * n_chunk + 3) / 4)), dim3(256), 0, (hipStream_t)stream, src,
                     seg_ptr, n_seg, F, n_chunk, out);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
