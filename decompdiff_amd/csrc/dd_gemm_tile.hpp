// The 64 x 64 projection-GEMM tile (K = 128 in two halves through a 33 KB LDS image) as a device function, shared by the GEMM
// launches (dd_gemm.hip) and by the coordinate launch, whose leading workgroups run the projections of the new h themselves
// (dd_attention2.hip::k_attn2_pos_g).  A tile is computed by 256 threads; in a wider workgroup every group of 256 threads
// (threadIdx.x >> 8) works on its own tile and LDS image, the barriers are the workgroup's.
#pragma once
#include "dd_kernels.hpp"

namespace dd {

#ifndef DD_GEMM_EARLY_BIAS
#define DD_GEMM_EARLY_BIAS 1  // the bias (one float4 per thread) requested ahead of the MFMAs: -0.4 % (B = 8), -1.0 % (B = 1), 0 (B = 16); R6-5
#endif
#ifndef DD_GEMM_EARLY_HALF
#define DD_GEMM_EARLY_HALF 0  // both K halves + the bias requested before the first barrier: measured 1.8 % SLOWER at B = 8 (120 instead of
#endif                        // 104 registers per thread, EXPERIMENTS.md R6-5)
constexpr int GT = 64;        // tile rows / cols
constexpr int GP = 130;       // LDS row pitch (floats)


// element offset of logical row r: rows are grouped in batches of rows_per_b (stride_b apart), ld apart inside a
// batch.  The plain case (one batch) needs no division.
__device__ __forceinline__ long row_offset(int r, int rows_per_b, long stride_b, int ld, bool plain) {
  return plain ? (long)r * ld : (long)(r / rows_per_b) * stride_b + (long)(r % rows_per_b) * ld;
}

// Epilogue: the four 32x32 accumulators go through an LDS tile (pitch 68) so that every thread writes whole
// 16-byte pieces of output rows (4 x global_store_dwordx4 instead of 16 scalar stores).  (Re-measured at the end of
// round 2: 16 dword stores straight from the accumulators -- 128 contiguous bytes per half wave and register, no LDS
// round trip, no barrier -- cost +2.3 % step time at B = 8 and +1.3 % at B = 16.)
constexpr int EP = 68;
// Write-through (sc1) 16-byte accesses through a buffer descriptor: what a tile of the persistent layer-tail queue uses for
// everything another workgroup of the SAME launch (or of a concurrently running one) produced or will consume -- an sc1
// store is visible device-wide once the wave's vmcnt has drained, an sc1 load bypasses this CU's L1 (guide section 6,
// Guideline 16, form R1): no release / acquire fences in the tile path.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_sc1(const float* base, long off) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, -1, 0x00020000);
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4), 0, 16);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st4_sc1(float* base, long off, const float4& v) {
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, -1, 0x00020000);
  const u32x4 u = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(u, rs, (int)(off * 4), 0, 16);
}
// (bias_pre: the thread's four bias values -- columns col0 + 4 (tid & 15) .. + 3, the same for its four output rows -- requested
//  by the caller ahead of the MFMAs instead of behind the LDS round trip here; !have_pre: loaded here)
template <bool SC1 = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, float* sm /*>= 64*EP floats, free*/, const f32x16& acc,
                                              int row0, int col0, const bool have_pre = false,
                                              const float4 bias_pre = float4{0.f, 0.f, 0.f, 0.f}) {
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, li = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;      // C/D map: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    sm[(wr * 32 + row) * EP + wc * 32 + li] = acc[r];
  }
  __syncthreads();
  const bool plain = a.y_rows_per_b >= a.rows;
  const bool vec_ok = ((a.ldy & 3) == 0) && ((a.y_stride_b & 3) == 0) && ((reinterpret_cast<size_t>(a.Y) & 15) == 0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int i = tid + k * 256;
    const int r = i >> 4, c4 = (i & 15) * 4;
    const int gr = row0 + r, gc = col0 + c4;
    if (gr >= a.rows || gc >= a.ncols) continue;
    const float4 v = *reinterpret_cast<const float4*>(&sm[r * EP + c4]);
    float o[4] = {v.x, v.y, v.z, v.w};
    const long yoff = row_offset(gr, a.y_rows_per_b, a.y_stride_b, a.ldy, plain) + gc;
    float* dst = a.Y + yoff;
    if (vec_ok && gc + 3 < a.ncols) {
      if (a.bias) {
        const float4 bb = have_pre ? bias_pre : *reinterpret_cast<const float4*>(a.bias + gc);
        o[0] += bb.x; o[1] += bb.y; o[2] += bb.z; o[3] += bb.w;
      }
      if (a.accumulate) {
        const float* ab = a.acc_src ? a.acc_src : a.Y;
        const float4 old = SC1 ? ld4_sc1(ab, yoff) : *reinterpret_cast<const float4*>(ab + yoff);
        o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
      }
      if (SC1) st4_sc1(a.Y, yoff, make_float4(o[0], o[1], o[2], o[3]));
      else *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (gc + e < a.ncols) {
          float x = o[e] + (a.bias ? a.bias[gc + e] : 0.f);
          if (a.accumulate) x += (a.acc_src ? a.acc_src + yoff : dst)[e];
          dst[e] = x;
        }
    }
  }
}

// K-split variant for jobs without the LayerNorm prologue: the two 64-wide halves of K go through a 33 KB LDS
// image (4 workgroups/CU instead of 2); the second half's global loads are in flight while the first half is
// multiplied.  Same MFMA order per output element as gemm_tile (k ascending), hence bit-identical results.
constexpr int GPH = 66;       // LDS pitch of a K-half tile: (66*row) mod 64 = 2*row -> conflict-free ds_read_b64
// `ticket` (persistent queue): the workgroup's NEXT ticket is drawn by thread 0 right after the second K-half went to LDS --
// every load of the wave has been consumed by then, so the returning atomic (which retires in order with loads) stalls
// nothing: it is in flight over the second half's MFMAs and the epilogue and is collected at the end-of-tile drain.
template <bool STAMPS = false, bool SC1 = false>
__device__ __forceinline__ void gemm_tile_ksplit(const GemmArgs& a, const int bx, const int by, float* smh /*>= 2*GT*GPH floats*/,
                                                 int32_t* ticket = nullptr, int ticket_step = 0, int* ticket_out = nullptr) {
  float* Xh = smh;
  float* Wh = smh + GT * GPH;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int row0 = bx * GT, col0 = by * GT;
  const bool xplain = a.x_rows_per_b >= a.rows;
  float4 xv[4], wv[4];
  auto fetch_x = [&](int half, float4 (&dst)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      const int r = i >> 4, c4 = half * 64 + (i & 15) * 4;
      const int gr = row0 + r;
      dst[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < a.rows) {
        const long xoff = row_offset(gr, a.x_rows_per_b, a.x_stride_b, a.ldx, xplain) + c4;
        dst[k] = SC1 ? ld4_sc1(a.X, xoff) : *reinterpret_cast<const float4*>(a.X + xoff);
        if (a.X2 != nullptr && a.x2_Eb > 0) {              // bond row -> row of its destination atom
          const long r2 = (long)(gr / a.x2_Eb) * a.x2_N + (gr % a.x2_Eb) / a.x2_NLm1;
          const float4 t = SC1 ? ld4_sc1(a.X2, r2 * a.x2_ld + c4) : *reinterpret_cast<const float4*>(a.X2 + r2 * a.x2_ld + c4);
          dst[k].x += t.x; dst[k].y += t.y; dst[k].z += t.z; dst[k].w += t.w;
        } else if (a.X2 != nullptr) {
          const int bb = gr / a.x2_N, n = gr % a.x2_N;
          if (n >= a.x2_NP) {
            const long o2 = ((long)bb * (a.x2_N - a.x2_NP) + (n - a.x2_NP)) * 128 + c4;
            const float4 t = SC1 ? ld4_sc1(a.X2, o2) : *reinterpret_cast<const float4*>(a.X2 + o2);
            dst[k].x += t.x; dst[k].y += t.y; dst[k].z += t.z; dst[k].w += t.w;
          }
        }
      }
    }
  };
  auto fetch_w = [&](int half, float4 (&dst)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      const int r = i >> 4, c4 = half * 64 + (i & 15) * 4;
      const int gc = col0 + r;
      dst[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gc < a.ncols) dst[k] = *reinterpret_cast<const float4*>(a.W + (long)gc * 128 + c4);
    }
  };
  auto fetch = [&](int half) { fetch_x(half, xv); fetch_w(half, wv); };
  auto commit2 = [&](const float4 (&xs)[4], const float4 (&ws)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256;
      const int r = i >> 4, c4 = (i & 15) * 4;
      float2* d = reinterpret_cast<float2*>(&Xh[r * GPH + c4]);
      d[0] = make_float2(xs[k].x, xs[k].y);
      d[1] = make_float2(xs[k].z, xs[k].w);
      float2* e = reinterpret_cast<float2*>(&Wh[r * GPH + c4]);
      e[0] = make_float2(ws[k].x, ws[k].y);
      e[1] = make_float2(ws[k].z, ws[k].w);
    }
  };
  auto commit = [&]() { commit2(xv, wv); };
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, hh = lane >> 5;
  const float* xa = &Xh[(wr * 32 + li) * GPH + 2 * hh];
  const float* wb = &Wh[(wc * 32 + li) * GPH + 2 * hh];
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // (phase stamps only in the stand-alone kernel tools/gemm_clocks.py drives: compiled out of the batched launches)
  long long* dbg = (STAMPS && a.dbg) ? a.dbg + ((long)by * gridDim.x + bx) * 8 : nullptr;
#define GSTAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  GSTAMP(0);
#if DD_GEMM_EARLY_HALF || DD_GEMM_EARLY_BIAS
  // the epilogue's bias values, requested now (they used to be a dependent global load behind the output's LDS round trip)
  const int bias_gc = col0 + (tid & 15) * 4;
  const bool bias_ok = a.bias != nullptr && bias_gc + 3 < a.ncols;
  float4 bias_pre = make_float4(0.f, 0.f, 0.f, 0.f);
  if (bias_ok) bias_pre = *reinterpret_cast<const float4*>(a.bias + bias_gc);
#endif
  if (a.ln != nullptr) {
    // LayerNorm+ReLU prologue (MLP hidden activation): both K-halves of the rows are fetched first; a row's 128 channels
    // sit in one 16-lane DPP row (4 + 4 channels per lane), so mean / variance are 4-step row reductions
    float4 x1[4];
    fetch_x(0, xv);
    fetch_x(1, x1);
    fetch_w(0, wv);
    const int cl = (tid & 15) * 4;
    const float4 g0 = *reinterpret_cast<const float4*>(a.ln + cl), g1 = *reinterpret_cast<const float4*>(a.ln + 64 + cl);
    const float4 b0 = *reinterpret_cast<const float4*>(a.ln + 128 + cl), b1 = *reinterpret_cast<const float4*>(a.ln + 192 + cl);
    auto row_sum = [](float v) {
      v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
      return v;
    };
    auto norm4 = [](float4& v, float mean, float rstd, const float4& g, const float4& b) {
      v.x = fmaxf(fmaf((v.x - mean) * rstd, g.x, b.x), 0.f);
      v.y = fmaxf(fmaf((v.y - mean) * rstd, g.y, b.y), 0.f);
      v.z = fmaxf(fmaf((v.z - mean) * rstd, g.z, b.z), 0.f);
      v.w = fmaxf(fmaf((v.w - mean) * rstd, g.w, b.w), 0.f);
    };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float s = ((xv[k].x + xv[k].y) + (xv[k].z + xv[k].w)) + ((x1[k].x + x1[k].y) + (x1[k].z + x1[k].w));
      const float mean = row_sum(s) * (1.0f / 128.0f);
      float q = 0.f;
      { const float d0 = xv[k].x - mean, d1 = xv[k].y - mean, d2 = xv[k].z - mean, d3 = xv[k].w - mean;
        q = fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3); }
      { const float d0 = x1[k].x - mean, d1 = x1[k].y - mean, d2 = x1[k].z - mean, d3 = x1[k].w - mean;
        q += fmaf(d0, d0, d1 * d1) + fmaf(d2, d2, d3 * d3); }
      const float rstd = dd_rsqrt(row_sum(q) * (1.0f / 128.0f) + 1e-5f);
      norm4(xv[k], mean, rstd, g0, b0);
      norm4(x1[k], mean, rstd, g1, b1);
    }
    commit();
    __syncthreads();
    GSTAMP(1);
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = x1[k];
    fetch_w(1, wv);
  } else {
#if DD_GEMM_EARLY_HALF
    // both K halves are requested before the first barrier (round 6): the second half's rows used to be requested behind the first
    // half's LDS commit and barrier, ~0.6 us of their latency exposed after the first half's 32 MFMAs in the single-tile launches
    // on the layer's critical chain; 16 more registers per thread, the 33 KB LDS image is unchanged
    float4 x1[4], w1[4];
    fetch(0);
    fetch_x(1, x1);
    fetch_w(1, w1);
    commit();
    __syncthreads();
    GSTAMP(1);
#pragma unroll
    for (int k = 0; k < 4; ++k) { xv[k] = x1[k]; wv[k] = w1[k]; }
#else
    fetch(0);
    commit();
    __syncthreads();
    GSTAMP(1);
    fetch(1);
#endif
  }
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    float2 av = *reinterpret_cast<const float2*>(xa + 4 * kk);
    float2 bv = *reinterpret_cast<const float2*>(wb + 4 * kk);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
  }
  asm volatile("" : "+v"(acc));
  GSTAMP(2);
  __syncthreads();
  commit();
  if (ticket != nullptr && tid == 0)
    *ticket_out = __hip_atomic_fetch_add(ticket, ticket_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  GSTAMP(3);
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) {
    float2 av = *reinterpret_cast<const float2*>(xa + 4 * kk);
    float2 bv = *reinterpret_cast<const float2*>(wb + 4 * kk);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
  }
  asm volatile("" : "+v"(acc));
  GSTAMP(4);
  __syncthreads();                                     // operands dead: the tile buffer becomes the output stage
#if DD_GEMM_EARLY_HALF || DD_GEMM_EARLY_BIAS
  gemm_epilogue<SC1>(a, smh, acc, row0, col0, bias_ok, bias_pre);
#else
  gemm_epilogue<SC1>(a, smh, acc, row0, col0);
#endif
  if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); GSTAMP(5); }
#undef GSTAMP
}



}  // namespace dd
