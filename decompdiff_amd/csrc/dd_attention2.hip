// Fused segment attention, tiled variant (v2): 16 members per tile, matrix cores for every contraction.
//
// Same math and interface as dd_attention.hip (see there for the algebra).  One wavefront owns one softmax segment;
// all per-segment state lives in registers, LDS only holds read-only weight images shared by the workgroup.
//
// k pass -- lane l = (mm = l & 15, cg = l >> 4) holds member mm of the current 16-member tile and the 32 hidden
// channels c(nt, r) = 16*nt + 4*cg + r (nt < 8, r < 4; register 4*nt + r):
//   * gathered projection rows summed in registers (requested one tile ahead);
//   * type (x) Gaussian / angular part of the first Linear: MFMA with the table as A operand (channel-permuted LDS
//     image packed on the host) and the per-lane features as B -> lands in this very layout;
//   * LayerNorm + ReLU on the VALU, row reductions over the 4 lanes of a member with v_permlane{16,32}_swap;
//   * scores S[m][h] = sum_c z_k[m][c] * Q~[h][c]: 32 MFMAs per tile (A = z from the registers above, B = Q~ held in
//     32 VGPRs per segment), exact fp32 (k-ordered fmaf chain);  softmax over the segment in registers.
// v pass -- member-major layout: lane (mm, cg) holds members 4*cg + r and channels 16*nt + mm.  The same table MFMAs
//   with swapped operands deliver the activation in exactly the layout the aggregation
//   Z~[h][c] = sum_m aw[m][h] * z_v[m][c] consumes as an A operand, with aw straight from the softmax registers as B.
//   No transposes, no LDS scratch.  (Coordinate modes keep the k-pass layout: v16[m][h] = z_v[m] . W2xv[h].)
// Epilogue -- W2v . Z~ from a head-permuted LDS image, reduce-scatter over the 4 lanes of a head with two swaps.
//
// LDS plans (per mode, struct Lds), persistent bond-layer workgroups, the prefetch pipeline and the launch shapes are
// described at their definitions below and in DESIGN.md section 5.
#include <vector>
#include <type_traits>

#include "dd_kernels.hpp"
#include "dd_gemm_tile.hpp"

#ifndef DD_KNN_CONST_ADD
#define DD_KNN_CONST_ADD 1
#endif
// Round 6 (EXPERIMENTS.md R6-1): the VALU stream of a segment cut by a quarter.  Each switch builds the round-5 form when 0
// (tools/build_variant.sh); DD_LN_FOLD also selects the weight form the library expects (dd_weights_form(), packing.py).
#ifndef DD_LN_FOLD
#define DD_LN_FOLD 1       // LayerNorm + ReLU of the key / value MLPs as max(fma(P, rstd, beta'), 0) on mean-free, sign-folded rows
#endif
#ifndef DD_FAST_ANGLE
#define DD_FAST_ANGLE 1    // angle codes without library atan2f / IEEE sqrt / IEEE division
#endif
#ifndef DD_PEEL_FOLD
#define DD_PEEL_FOLD 0     // first step of the query fold as products instead of zero-initialised accumulators (measured: +1 %, R6-1)
#endif
#ifndef DD_PEEL_Z
#define DD_PEEL_Z 0        // the same for the aggregation chains (measured: +1.8 %, 48 more registers, R6-1)
#endif
#ifndef DD_COOP
#define DD_COOP 0          // persistent bond-layer workgroups: query fold and epilogue split by head over the 8 waves of a trip (bl_coop_body);
                           // measured equal to the per-wave fold / epilogue (EXPERIMENTS.md R6-2): built with -DDD_COOP=1 for the A/B only
#endif
#ifndef DD_COOP_WK_RELOAD
#define DD_COOP_WK_RELOAD 0   // bl_coop_body: the wave's 16 W2k rows re-read (L2) with every trip's prologue instead of held in 32 registers
#endif
#ifndef DD_COOP_RC
#define DD_COOP_RC 0          // bl_coop_body: the segment's own row (Rk) requested with the next trip's prologue instead of at its first tile
#endif
#ifndef DD_NE_COOP_EPI
#define DD_NE_COOP_EPI 1   // node_layer_with_edge blocks: no W2v image -- the epilogue as one MFMA chain per wave (W2v rows from L2, Z~ through LDS)
#endif
#ifndef DD_NE_PERSIST
#define DD_NE_PERSIST 0    // node_layer_with_edge as persistent workgroups (ne_persist_body) beside the persistent bond-layer ones: bit-identical,
                           // measured 10 % SLOWER (EXPERIMENTS.md R6-8); compiled with -DDD_NE_PERSIST=1 for the A/B only
#endif
#ifndef DD_TRIP_SYNC
#define DD_TRIP_SYNC 6     // persistent bond-layer workgroups: a workgroup barrier every n-th trip (1 <= n <= 7).  Round 2 kept the waves in
                           // lock-step (free-running waves measured 6 % slower then); round 6: n = 3 ... 6 -1.7 % at B = 8, -3.5 % at C-large, bit-identical (R6-7)
#endif
#ifndef DD_NODE_TRACE
#define DD_NODE_TRACE 0    // measurement variant (tools/build_variant.sh trace -DDD_NODE_TRACE=1): per-workgroup clocks of the fused node launch
#endif
#ifndef DD_GAUSS_CACHE
#define DD_GAUSS_CACHE 1   // node_layer_with_edge: the tile's Gaussian features kept from the k pass for the v pass
#endif
#ifndef DD_UNCOND_FETCH
#define DD_UNCOND_FETCH 1  // next tile's rows requested unconditionally (clipped members): no register copies at the tile joins
#endif

namespace dd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace v2 {

// Pitch of the head-permuted W2k / W2v images.  The fold and the epilogue read them with ds_read_b128 at
// (d*16 + mm) * WPITCH + 16*nt + 4*cg; a b128 read is served in four 16-lane groups, each holding 8 values of mm with
// one cg and the complementary 8 with the next cg, so the 16-byte bank slot (mm * WPITCH/4 + cg) mod 16 must be
// injective over such a group: WPITCH/4 = 2 (mod 16) is, 33 (pitch 132) collides once per group (measured: 35 % of
// all LDS cycles were bank conflicts).
constexpr int WPITCH = 136;
constexpr int WB_FLOATS = 128 * WPITCH;        // 69.6 KB: one head-permuted 128 x 128 weight image (W2k or W2v)
constexpr int TABP = 24 * 128;                 // Gaussian table stride per edge type (21 rows + 3 zero rows)
// sum / max over the 4 lanes {l, l^16, l^32, l^48}
__device__ __forceinline__ float quad_sum(float v) { v = swap16_sum(v, v); return swap32_sum(v, v); }
__device__ __forceinline__ float quad_max(float v) { return swap32_max(swap16_max(v)); }

// ---- cooperative staging (all loads of a thread issued before its LDS stores) -------------------
template <int NT>
__device__ __forceinline__ void stage_w2k_permuted(float* WB, const float* __restrict__ W2k) {
  // global row r = h*8 + d  ->  LDS row d*16 + h (pitch WPITCH, see there)
  constexpr int PER = (4096 + NT - 1) / NT;
  float4 tmp[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * NT;
    if (4096 % NT == 0 || i < 4096) tmp[k] = reinterpret_cast<const float4*>(W2k)[i];
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * NT;
    if (4096 % NT == 0 || i < 4096) {
      const int r = i >> 5, c4 = (i & 31) * 4;
      *reinterpret_cast<float4*>(&WB[((r & 7) * 16 + (r >> 3)) * WPITCH + c4]) = tmp[k];
    }
  }
}
template <int NT, int N4>
__device__ __forceinline__ void stage_plain(float* dst, const float* __restrict__ src) {
  constexpr int PER = (N4 + NT - 1) / NT;
  float4 tmp[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * NT;
    if (N4 % NT == 0 || i < N4) tmp[k] = reinterpret_cast<const float4*>(src)[i];
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * NT;
    if (N4 % NT == 0 || i < N4) reinterpret_cast<float4*>(dst)[i] = tmp[k];
  }
}

__device__ __forceinline__ void add_row(float (&P)[32], const float* __restrict__ row, int cg) {
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const float4 v = *reinterpret_cast<const float4*>(row + 16 * nt + 4 * cg);
    P[4 * nt] += v.x; P[4 * nt + 1] += v.y; P[4 * nt + 2] += v.z; P[4 * nt + 3] += v.w;
  }
}
__device__ __forceinline__ void load_row(float (&P)[32], const float* __restrict__ row, int cg) {
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const float4 v = *reinterpret_cast<const float4*>(row + 16 * nt + 4 * cg);
    P[4 * nt] = v.x; P[4 * nt + 1] = v.y; P[4 * nt + 2] = v.z; P[4 * nt + 3] = v.w;
  }
}

// LayerNorm(128)+ReLU of a member row spread over the 4 lanes {l, l^16, l^32, l^48}, 32 channels each.
// DD_LN_FOLD (packing.py::kernel_form_layer): the rows arrive mean-free with the sign of gamma folded in, |gamma| sits in the
// second Linear and beta' = beta / |gamma| in ln[128..]: the variance is E[P^2] and the activation one FMA + one max.
__device__ __forceinline__ void ln_relu32(float (&P)[32], const float* __restrict__ ln /*LDS: gamma[128], beta[128]*/, int cg) {
#if DD_LN_FOLD
  float v0 = P[0] * P[0], v1 = P[1] * P[1];
#pragma unroll
  for (int k = 2; k < 32; k += 2) { v0 = fmaf(P[k], P[k], v0); v1 = fmaf(P[k + 1], P[k + 1], v1); }
  const float rstd = dd_rsqrt(quad_sum(v0 + v1) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const float4 b = *reinterpret_cast<const float4*>(ln + 128 + 16 * nt + 4 * cg);
    P[4 * nt] = fmaxf(fmaf(P[4 * nt], rstd, b.x), 0.f);
    P[4 * nt + 1] = fmaxf(fmaf(P[4 * nt + 1], rstd, b.y), 0.f);
    P[4 * nt + 2] = fmaxf(fmaf(P[4 * nt + 2], rstd, b.z), 0.f);
    P[4 * nt + 3] = fmaxf(fmaf(P[4 * nt + 3], rstd, b.w), 0.f);
  }
#else
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) s += P[k];
  const float mean = quad_sum(s) * (1.0f / 128.0f);
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) { P[k] -= mean; v = fmaf(P[k], P[k], v); }
  const float rstd = dd_rsqrt(quad_sum(v) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const float4 g = *reinterpret_cast<const float4*>(ln + 16 * nt + 4 * cg);
    const float4 b = *reinterpret_cast<const float4*>(ln + 128 + 16 * nt + 4 * cg);
    P[4 * nt] = fmaxf(fmaf(P[4 * nt] * rstd, g.x, b.x), 0.f);
    P[4 * nt + 1] = fmaxf(fmaf(P[4 * nt + 1] * rstd, g.y, b.y), 0.f);
    P[4 * nt + 2] = fmaxf(fmaf(P[4 * nt + 2] * rstd, g.z, b.z), 0.f);
    P[4 * nt + 3] = fmaxf(fmaf(P[4 * nt + 3] * rstd, g.w, b.w), 0.f);
  }
#endif
}

// D[m][h] = sum_c z[m][c] * Bm[h][c] for one tile; two accumulators hide the 40-cycle dependent latency
__device__ __forceinline__ f32x4 mfma_rows(const float (&z)[32], const float (&Bm)[32]) {
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 32; s += 2) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(z[s], Bm[s], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(z[s + 1], Bm[s + 1], a1, 0, 0, 0);
  }
  return a0 + a1;
}

// sum over the 16 lanes of a DPP row (lanes sharing lane >> 4)
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// LayerNorm(128)+ReLU in the member-major ("T") layout: Tz[4*nt + r] = row of member 4*cg + r, channel 16*nt + mm.
__device__ __forceinline__ void ln_relu_T(float (&Tz)[32], const float* __restrict__ ln /*LDS*/, int mm) {
#if DD_LN_FOLD
  float be[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) be[nt] = ln[128 + 16 * nt + mm];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float v0 = Tz[r] * Tz[r], v1 = Tz[4 + r] * Tz[4 + r];
#pragma unroll
    for (int nt = 2; nt < 8; nt += 2) { v0 = fmaf(Tz[4 * nt + r], Tz[4 * nt + r], v0); v1 = fmaf(Tz[4 * nt + 4 + r], Tz[4 * nt + 4 + r], v1); }
    const float rstd = dd_rsqrt(row16_sum(v0 + v1) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) Tz[4 * nt + r] = fmaxf(fmaf(Tz[4 * nt + r], rstd, be[nt]), 0.f);
  }
#else
  float g[8], be[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) { g[nt] = ln[16 * nt + mm]; be[nt] = ln[128 + 16 * nt + mm]; }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) s += Tz[4 * nt + r];
    const float mean = row16_sum(s) * (1.0f / 128.0f);
    float v = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { Tz[4 * nt + r] -= mean; v = fmaf(Tz[4 * nt + r], Tz[4 * nt + r], v); }
    const float rstd = dd_rsqrt(row16_sum(v) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) Tz[4 * nt + r] = fmaxf(fmaf(Tz[4 * nt + r] * rstd, g[nt], be[nt]), 0.f);
  }
#endif
}

// sin and cos of x for 0 <= x <= ~10 (angle codes: x <= 3*pi): three-term Cody-Waite reduction by pi/2 and the
// cephes single-precision kernels on [-pi/4, pi/4]; absolute error < 1.5e-7.  Branch-free, no large-argument path
// (the library sincosf drags a Payne-Hanek loop and a private array into every caller).
__device__ __forceinline__ void sincos_small(float x, float& sn, float& cs) {
  const float kf = rintf(x * 0.63661977236758134f);
  float r = fmaf(-kf, 1.5703125f, x);                  // pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8
  r = fmaf(-kf, 4.837512969970703125e-4f, r);
  r = fmaf(-kf, 7.54978995489188e-8f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                        fmaf(-0.5f, z, 1.0f));
  const int q = (int)kf;
  const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  sn = (q & 2) ? -s0 : s0;
  cs = ((q + 1) & 2) ? -c0 : c0;
}

// The three angular codes g = cg, 4 + cg, 8 + cg (merged order, see below) of the angle between a and b:
//   code = [th, sin{1,2,3}th, sin{1,1/2,1/3}th, cos{1,2,3}th, cos{1,1/2,1/3}th, 0, 0, 0]
// (AngularEncoding, models/common.py:38-53: freq 1,2,3 then 1,1/2,1/3; th = atan2(|a x b|, a.b)).
// sin/cos of th come straight from the cross and dot products, the multiples from the addition formulas and the
// half angle from sin th = 2 sin(th/2) cos(th/2) with the well-conditioned root; only th/3 needs a sincos.
__device__ __forceinline__ void angle_codes(float nn /*|a x b|^2*/, float dt /*a.b*/, int cg, float (&out)[3]) {
#if DD_FAST_ANGLE
  // No library atan2f, no IEEE sqrt / division (v_sqrt / v_rsq / v_rcp are 1 ulp): sin th and cos th from the products,
  //   tan(phi) = sin th / (1 + |cos th|)  with phi = th/2 (cos th >= 0) or pi/2 - th/2, phi in [0, pi/4], atan as an odd
  //   degree-15 minimax polynomial (2e-8 on [0, 1]);  sin / cos of th/3 by doubling from th/6 in [0, pi/6] (cephes kernels).
  // All 11 codes within 5e-7 of the exact values (tools/angle_codes_check.py: no worse than atan2f + sinf / cosf of fp32 products).
  const float n = __builtin_amdgcn_sqrtf(nn);
  const float n2 = fmaf(dt, dt, nn);
  const bool ok = n2 > 0.f;
  const float r = ok ? dd_rsqrt(n2) : 0.f;
  const float s1 = n * r, c1 = ok ? dt * r : 1.0f;
  const float s2 = 2.0f * s1 * c1, c2 = fmaf(c1, c1, -s1 * s1);
  const float s3 = fmaf(s2, c1, c2 * s1), c3 = fmaf(c2, c1, -s2 * s1);
  const float den = 1.0f + fabsf(c1);
  const float x = s1 * __builtin_amdgcn_rcpf(den);
  const float u = x * x;
  float p = -4.0544654832e-03f;
  p = fmaf(p, u, 2.1862573855e-02f);
  p = fmaf(p, u, -5.5911747027e-02f);
  p = fmaf(p, u, 9.6421528094e-02f);
  p = fmaf(p, u, -1.3908611309e-01f);
  p = fmaf(p, u, 1.9946561855e-01f);
  p = fmaf(p, u, -3.3329860447e-01f);
  p = fmaf(p, u, 9.9999933550e-01f);
  const float phi = p * x;
  const bool acute = c1 >= 0.f;
  const float th = acute ? 2.0f * phi : fmaf(-2.0f, phi, 3.14159265358979f);
  const float big = __builtin_amdgcn_sqrtf(0.5f * den);            // the well-conditioned half-angle root
  const float small = 0.5f * s1 * __builtin_amdgcn_rcpf(big);
  const float sh = acute ? small : big, ch = acute ? big : small;
  const float a6 = th * (1.0f / 6.0f), z = a6 * a6;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, a6, a6);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                        fmaf(-0.5f, z, 1.0f));
  const float st = 2.0f * ps * pc, ct = fmaf(-2.0f * ps, ps, 1.0f);
#else
  const float n = sqrtf(nn);
  const float th = atan2f(n, dt);
  const float n2 = fmaf(n, n, dt * dt);
  const float r = n2 > 0.f ? dd_rsqrt(n2) : 0.f;
  const float s1 = n * r, c1 = n2 > 0.f ? dt * r : 1.0f;
  const float s2 = 2.0f * s1 * c1, c2 = fmaf(c1, c1, -s1 * s1);
  const float s3 = fmaf(s2, c1, c2 * s1), c3 = fmaf(c2, c1, -s2 * s1);
  float sh, ch;
  if (c1 >= 0.f) { ch = sqrtf(0.5f * (1.0f + c1)); sh = 0.5f * s1 / ch; }
  else           { sh = sqrtf(0.5f * (1.0f - c1)); ch = 0.5f * s1 / sh; }
  float st, ct;
  sincos_small(th * (1.0f / 3.0f), st, ct);
#endif
  // merged code order (packing.py _ANGLE_MERGE): [th, s1, s2, s3 | sin th/2, sin th/3, c1, c2 | c3, cos th/2, cos th/3, 0]
  out[0] = cg == 0 ? th : (cg == 1 ? s1 : (cg == 2 ? s2 : s3));      // g = 0..3
  out[1] = cg == 0 ? sh : (cg == 1 ? st : (cg == 2 ? c1 : c2));      // g = 4..7
  out[2] = cg == 0 ? c3 : (cg == 1 ? ch : (cg == 2 ? ct : 0.f));     // g = 8..11
}

// first-layer table part on the matrix cores: one k-step (4 table rows) for the 8 channel tiles.
//   TR = false: acc[nt][r] = P[member mm][channel 16nt+4cg+r]   (A = table, B = feature)
//   TR = true : acc[nt][r] = P[member 4cg+r][channel 16nt+mm]   (A = feature, B = table)
template <bool TR>
__device__ __forceinline__ void mfma_table_step(f32x4 (&acc)[8], const float* __restrict__ tb /*LDS row 4s+cg, + mm*4*/, float f) {
  const float4 w0 = *reinterpret_cast<const float4*>(tb);
  const float4 w1 = *reinterpret_cast<const float4*>(tb + 64);
  const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
  for (int nt = 0; nt < 8; ++nt)
    acc[nt] = TR ? __builtin_amdgcn_mfma_f32_16x16x4f32(f, w[nt], acc[nt], 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_16x16x4f32(w[nt], f, acc[nt], 0, 0, 0);
}

template <int MODE>
struct Lds {
  static constexpr bool TRIP = (MODE == M_BL);
  static constexpr bool KNN = (MODE == M_NE || MODE == M_PE);
  static constexpr bool RES = (MODE == M_NB || MODE == M_BL);  // no Gaussian tables: W2k and W2v images both resident
  static constexpr int WV = RES ? WB_FLOATS : 0;              // W2v image (aliases the W2k buffer unless RES)
  static constexpr int XPAD = KNN ? 128 : 0;                  // (node mode: the W2k image's place later holds the 8 x (16 x WPITCH + 16) exchange buffer)
  static constexpr int TAB = WB_FLOATS + XPAD;                // kNN modes: [k lo, k hi, v lo, v hi] tables of the two
                                                              // source types a workgroup meets (4 x 24 x 128)
  static constexpr int LNP = WB_FLOATS * (RES ? 2 : 1) + XPAD + (KNN ? 4 * TABP : 0);   // [4][128] gamma_k, beta_k, gamma_v, beta_v
  static constexpr int WAO = LNP + 512;                       // [2][12][128] angle weights (BL), MFMA operand layout
  static constexpr int TOTAL = WAO + (TRIP ? 2 * 12 * 128 : 0);
};

// consumer side of the in-launch hand-off of k_attn2_pos_g (see there; measurement build)
__device__ __forceinline__ void pos_wait_tiles(const int32_t* counter, int target_in) {
  const bool fast = target_in < 0;                        // (measurement: negative target = poll every 0.2 us, no back-off)
  const int target = fast ? -target_in : target_in;
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (spins == 0 || fast) __builtin_amdgcn_s_sleep(8);                  // 0.2, 0.4, 0.8 us, ... capped at 3.3 us between polls
      else if (spins == 1) __builtin_amdgcn_s_sleep(16);
      else if (spins == 2) __builtin_amdgcn_s_sleep(32);
      else if (spins == 3) __builtin_amdgcn_s_sleep(64);
      else __builtin_amdgcn_s_sleep(127);
      if (++spins > (1u << 20)) __builtin_trap();                   // > 3 s: the producers of this launch never ran -- fail loudly
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// node_layer_with_edge workgroups never mix protein and ligand centres (the Gaussian tables of only two edge types
// are then needed): per sample ceil(NP/NW) protein blocks followed by ceil(NL/NW) ligand blocks.
__host__ __device__ inline int ne_blocks_per_sample(int NP, int NL, int NW) { return (NP + NW - 1) / NW + (NL + NW - 1) / NW; }

// Body of one workgroup (NW waves = NW segments).  `block` is the workgroup index within this mode's range and
// `smem` the workgroup's LDS (>= Lds<MODE>::TOTAL floats), so several modes can share one launch.  All per-wave
// state lives in registers; LDS only holds the (read-only) weight images shared by the workgroup.
// PERSIST (modes without Gaussian tables): the workgroup stages its images once and every wave then pulls segments
// from the global counter a.work_counter until none are left -- no barriers after the first one.
// PAIR (coordinate modes): two waves per segment.  The launch is a pure latency chain (one segment per wave, 120
// workgroups), and its two halves only meet at the very end: wave 0 of a pair runs the query MLP, the fold, the k pass and
// the softmax, wave 1 meanwhile the v pass (activations and the 16 per-head values of every member); the attention
// weights cross through LDS behind one workgroup barrier and wave 1 forms the coordinate update.
// PAIR = 2 (QUAD, round 6): four waves per segment -- the two 16-member tiles of a kNN segment (odd / even tiles of a longer bond
// segment) go to different waves on both sides; the softmax's max and sum and the two halves of the coordinate update cross through
// LDS (3 more workgroup barriers).  With NW = 2 segments per workgroup the launch has 240 workgroups instead of 120.
template <int MODE, int MAXT, int NW, bool PERSIST = false, bool RAG = false, int PAIR = 0, bool STAMPS = true, typename ARGS = AttnArgs>
__device__ __forceinline__ void attn2_body(const ARGS& a, const int block, float* smem) {
  constexpr bool KNN = (MODE == M_NE || MODE == M_PE);
  constexpr bool POS = (MODE == M_PE || MODE == M_PB);
  constexpr bool TRIP = (MODE == M_BL);
  constexpr bool BOND = (MODE == M_NB || MODE == M_PB);
  static_assert(!PAIR || (POS && !PERSIST), "wave pairs: coordinate modes");
  constexpr bool QUAD = PAIR == 2;
  constexpr int NT = (QUAD ? 4 : (PAIR ? 2 : 1)) * NW * 64;
  using L = Lds<MODE>;
  constexpr int LNP = L::LNP, WAO = L::WAO;
  constexpr bool RES = L::RES;
  static_assert(!PERSIST || RES, "persistent workgroups need both weight images resident");
  float* WB = smem;
  float* WV = smem + L::WV;
  // (pairs: waves p and p + NW -- one k-side and one v-side wave on every SIMD, which want different units at a time)
  const int wave = PAIR ? ((threadIdx.x >> 6) % NW) : (threadIdx.x >> 6), lane0 = threadIdx.x & 63;
  const int role = PAIR ? ((threadIdx.x >> 6) / NW) : 0;  // 0: query / k pass / softmax, 1: v pass / coordinate update
  const bool kside = !PAIR || (QUAD ? role < 2 : role == 0), vside = !PAIR || (QUAD ? role >= 2 : role == 1);   // (QUAD: roles 0, 1 / 2, 3)
  const int tsel = role & 1;                              // QUAD: this wave's tiles are t = tsel, tsel + 2, ...
  auto mine = [&](int t) { return !QUAD || (t & 1) == tsel; };

  const int N = a.NP + a.NL, NLm1 = a.NL - 1, Eb = a.NL * NLm1;
  // Padded heterogeneous batch (a.nl_real != NULL): sample b's real atoms are the first np_real[b] / nl_real[b] rows of
  // its blocks.  Padding atoms own no segment and are no members: kNN lists hold real atoms only, and because padding
  // sits at the END of a ligand block the real members of a bond / triplet segment are a PREFIX of the dense member
  // order -- the member count M (and tile count T) simply becomes a per-segment value.  The persistent bond-layer
  // workgroups enumerate the real segments compactly through the prefix sums a.bl_prefix.
  // (RAG is a template parameter: the dense instantiations keep their register allocation -- the masked code costs
  // 1.5 % of the step when it shares a kernel with the dense path)
  constexpr bool COMPACT = RAG && TRIP && PERSIST;
  const int nseg = COMPACT ? a.bl_prefix[a.B] : ((MODE == M_NE) ? a.B * N : (TRIP ? a.B * Eb : a.B * a.NL));
  int M = KNN ? a.K : (TRIP ? a.NL - 2 : NLm1);
  int T = (M + 15) >> 4;
  // compact index r of a real bond-layer segment -> its id in the dense (padded) enumeration b*Eb + i*(NL-1) + j'
  auto bl_dense_seg = [&](int r) -> int {
    if (!COMPACT) return r;
    int bb = 0;
    while (a.bl_prefix[bb + 1] <= r) ++bb;               // B <= 64 samples, wave-uniform scalar loop
    const int nm1 = a.nl_real[bb] - 1, e = r - a.bl_prefix[bb];
    return bb * Eb + (e / nm1) * NLm1 + (e % nm1);
  };
  // node_layer_with_edge: block -> (sample, protein or ligand centres, first node)
  const int ne_bps = ne_blocks_per_sample(a.NP, a.NL, NW), ne_nbp = (a.NP + NW - 1) / NW;
  const int ne_b = block / ne_bps, ne_rb = block % ne_bps;
  const bool wg_protein = (MODE == M_NE) && ne_rb < ne_nbp;
  if (MODE == M_NE && RAG && ne_b < a.B) {               // a block of padding centres only: nothing to do (block-uniform)
    const int first = wg_protein ? ne_rb * NW : (ne_rb - ne_nbp) * NW;
    if (first >= (wg_protein ? (a.np_real ? a.np_real[ne_b] : a.NP) : a.nl_real[ne_b])) return;
  }
  // (STAMPS = false in the fused launches, which never run with a clock buffer: the stamp branches split the k pass of
  // tile 0 into five basic blocks the scheduler cannot move instructions across; -1.4 % step time, bit-identical)
  long long* dbg = (STAMPS && a.dbg_clock) ? a.dbg_clock + (long)block * 16 : nullptr;
#define DD_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  DD_STAMP(0);

  // W2k (head-permuted; W2v too when resident), LayerNorm parameters, angle weights
  auto stage_all = [&]() {
    stage_w2k_permuted<NT>(WB, a.W2k);
    if (RES) stage_w2k_permuted<NT>(WV, a.W2v);          // row o = h*8 + j  ->  LDS row j*16 + h
    if (threadIdx.x < 64) {
      reinterpret_cast<float4*>(smem + LNP)[threadIdx.x] = reinterpret_cast<const float4*>(a.lnk)[threadIdx.x];
      reinterpret_cast<float4*>(smem + LNP + 256)[threadIdx.x] = reinterpret_cast<const float4*>(a.lnv)[threadIdx.x];
    }
    if (TRIP) {
      for (int i = threadIdx.x; i < 12 * 32; i += NT) {
        reinterpret_cast<float4*>(smem + WAO)[i] = reinterpret_cast<const float4*>(a.Wakp)[i];
        reinterpret_cast<float4*>(smem + WAO + 12 * 128)[i] = reinterpret_cast<const float4*>(a.Wavp)[i];
      }
    }
    if (KNN) {
      // edge type = 2 * (source is protein) + (centre is protein); the centre kind is uniform over the workgroup
      const int tyl = wg_protein ? 1 : 0;
      constexpr int PER = (4 * 768) / NT;                  // float4 per thread (NT divides 3072 for 2, 4, 8, 12 waves)
      static_assert((4 * 768) % NT == 0, "table staging assumes NT | 3072");
      float4 tmp[PER];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const int i = threadIdx.x + k * NT, q = i / 768, w = i - q * 768;
        const float* src = (q < 2 ? a.Akp : a.Avp) + (tyl + 2 * (q & 1)) * TABP;
        tmp[k] = reinterpret_cast<const float4*>(src)[w];
      }
#pragma unroll
      for (int k = 0; k < PER; ++k) reinterpret_cast<float4*>(smem + L::TAB)[threadIdx.x + k * NT] = tmp[k];
    }
    __syncthreads();
  };
  // persistent batches: trip indices in sb[it & 7] (thread 0 draws the next one while the current is processed; sb[8] = number of
  // indices published so far, release/acquire), next batch's query prefetched over the epilogue
  int* sb = reinterpret_cast<int*>(smem + L::TOTAL);
  int it = 0;
  float4 qn0 = make_float4(0.f, 0.f, 0.f, 0.f), qn1 = qn0;
  // trip index -> first segment (compact index) and number of segments of the trip (see AttnArgs::trip_full)
  auto trip_range = [&](int tr, int& base, int& cnt) {
    if (tr < a.trip_full) { base = tr * NW; cnt = NW; }
    else { base = a.trip_full * NW + (tr - a.trip_full) * a.trip_q; cnt = a.trip_q; }
    cnt = nseg - base < cnt ? nseg - base : cnt;         // <= 0: no work left
  };
  if (PERSIST) {
    if (threadIdx.x == 0) { sb[0] = atomicAdd(a.work_counter, 1); sb[8] = 0; }
    stage_all();                                       // (ends with the barrier that also publishes sb[0])
  }

  for (;;) {
  // the lane id is laundered per iteration so that the (loop-invariant) per-lane address arithmetic and LDS weight
  // reads stay inside the iteration instead of being hoisted into a few hundred live registers
  int lane = lane0;
  asm volatile("" : "+v"(lane) :: "memory");
  const int mm = lane & 15, cg = lane >> 4;            // member slot / channel group; also (head, row-group)
  int seg;
  if (PERSIST) {
    // NW consecutive segments per trip (round 2: letting every wave pull segments on its own was measured 6 % slower; round 6: the
    // waves of a workgroup still take the segments of a trip together, but only meet at a barrier every DD_TRIP_SYNC trips)
    // (DD_TRIP_SYNC = n: a barrier every n-th trip only -- a wave's trip index then leads another's by at most n - 1, the eight trip
    //  slots sb[0..7] and the published count sb[8] cover n <= 7; every wave has waited for sb[8] >= it in the previous trip's epilogue)
    if (it > 0 && it % DD_TRIP_SYNC == 0) __syncthreads();   // a barrier per trip keeps the waves in phase
    int base, cnt;
    trip_range(__builtin_amdgcn_readfirstlane(sb[it & 7]), base, cnt);
    if (cnt <= 0) break;
    if (threadIdx.x == 0) {
      sb[(it + 1) & 7] = atomicAdd(a.work_counter, 1);
      __hip_atomic_store(&sb[8], it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    seg = wave < cnt ? bl_dense_seg(base + wave) : a.B * Eb;
  } else if (MODE == M_NE) {
    const int nd = wg_protein ? ne_rb * NW + wave : a.NP + (ne_rb - ne_nbp) * NW + wave;
    seg = (nd < (wg_protein ? a.NP : N) && ne_b < a.B) ? ne_b * N + nd : nseg;
  } else {
    seg = block * NW + wave;
  }
  bool active = seg < ((MODE == M_NE) ? a.B * N : (TRIP ? a.B * Eb : a.B * a.NL));

  int b = 0, si = 0, sj = 0, node = 0;
  if (active) {
    if (MODE == M_NE) { b = seg / N; node = seg % N; }
    else if (TRIP) { b = seg / Eb; const int e = seg % Eb; si = e / NLm1; const int jp = e % NLm1; sj = jp + (jp >= si ? 1 : 0); }
    else { b = seg / a.NL; si = seg % a.NL; node = a.NP + si; }
    if (RAG) {
      const int nlb = a.nl_real[b], npb = a.np_real ? a.np_real[b] : a.NP;
      if (MODE == M_NE) active = node < a.NP ? node < npb : node - a.NP < nlb;
      else if (TRIP) active = si < nlb && sj < nlb;
      else active = si < nlb;
      if (!KNN) { M = TRIP ? nlb - 2 : nlb - 1; T = (M + 15) >> 4; }
    }
  }
  const float* xb = a.x + (long)b * N * 3;
  const float* xl = xb + (long)a.NP * 3;
  const long nrow = (long)b * N + node;                // kNN list row (KNN modes)
  const long src_base = KNN ? (long)b * N : (long)b * a.NL;
  const long erow0 = (long)seg * NLm1;
  const long drow = (MODE == M_NE) ? (long)seg : (long)b * a.NL + si;
  const int tlo = si < sj ? si : sj, thi = si < sj ? sj : si;
  auto trip_k = [&](int mc) { int k = mc; if (k >= tlo) ++k; if (k >= thi) ++k; return k; };   // third atom of member mc

  // kNN: neighbour and distance of member 16t + mm (issued first: two dependent global round trips)
  int jm[MAXT];
  int jT[MAXT][4];                                     // neighbours of members 16t + 4cg + r (v pass, member-major)
  float dm[MAXT], ewm[MAXT][4];                        // ewm: edge weights of the same members
  if (KNN && active) {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int m = 16 * t + mm;
      jm[t] = a.nbr[nrow * a.K + (m < M ? m : M - 1)];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int mr = 16 * t + 4 * cg + r;
        ewm[t][r] = a.ew[nrow * a.K + (mr < M ? mr : 0)];
        jT[t][r] = a.nbr[nrow * a.K + (mr < M ? mr : M - 1)];
      }
    }
    const float cx = xb[3 * node], cy = xb[3 * node + 1], cz = xb[3 * node + 2];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const float rx = cx - xb[3 * jm[t]], ry = cy - xb[3 * jm[t] + 1], rz = cz - xb[3 * jm[t] + 2];
      dm[t] = sqrtf(rx * rx + ry * ry + rz * rz);
    }
  }

  if (!PERSIST) stage_all();
#if defined(DD_DEBUG_OPTIONS) && DD_DEBUG_OPTIONS          // (measurement build only: EXPERIMENTS.md R5-2)
  if (POS && a.wait_flags != nullptr) {
    // the projections of the new h (k / v source rows, destination rows, query hidden rows) are formed by the LEADING workgroups
    // of this very launch (k_attn2_pos_g): staged the weight images first, now wait for their tile counter (see pos_wait_tiles)
    pos_wait_tiles(a.wait_flags + a.wait_idx, a.wait_n);
  }
#endif
  DD_STAMP(1);

  // query first: the Q~ fold must not queue behind the prefetched gathers (loads return in order)
  float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
  if (PERSIST && it > 0) {
    q0 = qn0; q1 = qn1;                                // requested during the previous trip's epilogue
  } else if (POS && a.W2q != nullptr && kside) {
    // coordinate modes: the query MLP's second layer runs here (q = W2q . relu(LN(hidden)) + b2q, one 128x128
    // mat-vec per segment) instead of as a 16-tile GEMM launch on the critical chain.  Lane l owns outputs 2l, 2l+1.
    float* sc = smem + L::TOTAL + (QUAD ? wave + NW * tsel : wave) * 256;   // (QUAD: both k waves of a segment form the query, each in its own scratch)
    if (active) {
      float2 hv = *reinterpret_cast<const float2*>(a.qhid + drow * a.ld_qhid + 2 * lane);
      ln_relu2(hv.x, hv.y, a.lnq[2 * lane], a.lnq[2 * lane + 1], a.lnq[128 + 2 * lane], a.lnq[128 + 2 * lane + 1]);
      *reinterpret_cast<float2*>(sc + 2 * lane) = hv;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (active) {
      const float2* wT = reinterpret_cast<const float2*>(a.W2q) + lane;      // W2q^T [k][o]: row k is one coalesced 512 B read
      float a0 = 0.f, a1 = 0.f;
#pragma unroll 4
      for (int k4 = 0; k4 < 32; ++k4) {                 // 64 row reads in flight per trip: the mat-vec is latency-bound
        const float4 z = *reinterpret_cast<const float4*>(sc + 4 * k4);
        const float2 u0 = wT[(4 * k4 + 0) * 64], u1 = wT[(4 * k4 + 1) * 64], u2 = wT[(4 * k4 + 2) * 64], u3 = wT[(4 * k4 + 3) * 64];
        a0 = fmaf(u0.x, z.x, a0); a1 = fmaf(u0.y, z.x, a1);
        a0 = fmaf(u1.x, z.y, a0); a1 = fmaf(u1.y, z.y, a1);
        a0 = fmaf(u2.x, z.z, a0); a1 = fmaf(u2.y, z.z, a1);
        a0 = fmaf(u3.x, z.w, a0); a1 = fmaf(u3.y, z.w, a1);
      }
      *reinterpret_cast<float2*>(sc + 128 + 2 * lane) = make_float2(a0 + a.b2q[2 * lane], a1 + a.b2q[2 * lane + 1]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (active) {
      q0 = *reinterpret_cast<const float4*>(sc + 128 + mm * 8);
      q1 = *reinterpret_cast<const float4*>(sc + 128 + mm * 8 + 4);
    }
  } else if (active && kside) {
    q0 = *reinterpret_cast<const float4*>(a.q + (long)seg * 128 + mm * 8);
    q1 = *reinterpret_cast<const float4*>(a.q + (long)seg * 128 + mm * 8 + 4);
  }
  // ---- prefetch: rows of the k pass and the triplet geometry are requested now, ahead of the Q~ fold and the angle
  //      codes, so that the 2-3 us the gathers take (the tables exceed L2) are not exposed at the head of pass 1
  float Rc[32], Pf[32], Pf2[32];                          // segment row; gather row(s) of the next k-pass tile
  float tri_i[3] = {0.f, 0.f, 0.f}, tri_j[3] = {0.f, 0.f, 0.f}, tri_k[MAXT][3];
  auto fetch_k_rows = [&](int t) {
    const int m = 16 * t + mm;
    const int mc = m < M ? m : M - 1;
    if (KNN) {
      load_row(Pf, a.ks + (src_base + jm[t < MAXT ? t : 0]) * a.ld_ks, cg);
    } else if (!TRIP) {
      load_row(Pf, a.ks + (src_base + mc + (mc >= si ? 1 : 0)) * a.ld_ks, cg);
      load_row(Pf2, a.ke + (erow0 + mc) * a.ld_ke, cg);
    } else {
      const int k = trip_k(mc);
      load_row(Pf, a.ke + ((long)b * Eb + sj * NLm1 + (k - (k > sj ? 1 : 0))) * a.ld_ke, cg);
    }
  };
  // bond-graph and triplet segments: the segment's own row is re-read per tile (L1 / L2) instead of held in 32 registers
  // across the whole k pass -- the same sum (a + b = b + a).  The 4-tile node launch (ligands of 50-64 atoms) goes from 135
  // spilled registers to none (C-large: 205 -> 222 steps/s), the 3-tile one from 56 to none, the 2-tile one from 252 to
  // 229 registers (-0.4 % step time); results bit-identical (tools/lib_checksum.py).
  constexpr bool REREAD = !KNN && !POS;
  const float* rc_row = TRIP ? a.Rk + (long)seg * 128 : a.kd + drow * a.ld_kd;
  if (active && kside) {
    if (!REREAD) load_row(Rc, rc_row, cg);
    fetch_k_rows(QUAD ? tsel : 0);
    if (TRIP) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { tri_i[c] = xl[3 * si + c]; tri_j[c] = xl[3 * sj + c]; }
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int m = 16 * t + mm;
        const int k = trip_k(m < M ? m : M - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) tri_k[t][c] = xl[3 * k + c];
      }
    }
  }
  // the query is needed first: wait for it alone (exact count: straight-line code), and hand it to the rolled fold
  // loop below as plain register values -- a load result consumed inside a loop costs a conservative vmcnt(0)
  asm volatile("" : "+v"(q0.x), "+v"(q0.y), "+v"(q0.z), "+v"(q0.w), "+v"(q1.x), "+v"(q1.y), "+v"(q1.z), "+v"(q1.w));

  // ---- Q~ as the MFMA B operand: lane (h = mm, cg) holds Q~[h][c(kk, cg)] ----------------------------------
  float Qb[32];
#if !DD_PEEL_FOLD
#pragma unroll
  for (int k = 0; k < 32; ++k) Qb[k] = 0.f;
#endif
  if (active && kside) {
    // a real loop (8 LDS reads in flight per trip): fully unrolled, the scheduler hoists all 64 reads = 256 registers.
    // The 8 query values rotate through r0 so that no dynamic register indexing is needed.
    float r0 = q0.x, r1 = q0.y, r2 = q0.z, r3 = q0.w, r4 = q1.x, r5 = q1.y, r6 = q1.z, r7 = q1.w;
#if DD_PEEL_FOLD
    {                                                    // d = 0 peeled: products instead of 32 zeroing moves + 32 FMAs
      const float* wr = WB + mm * WPITCH + 4 * cg;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 w = *reinterpret_cast<const float4*>(wr + 16 * nt);
        Qb[4 * nt] = r0 * w.x; Qb[4 * nt + 1] = r0 * w.y; Qb[4 * nt + 2] = r0 * w.z; Qb[4 * nt + 3] = r0 * w.w;
      }
      r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5; r5 = r6; r6 = r7;
    }
#pragma nounroll
    for (int d = 1; d < 8; ++d) {
      const float qv = r0;
      r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5; r5 = r6; r6 = qv;
#else
#pragma nounroll
    for (int d = 0; d < 8; ++d) {                        // (unrolled by 2: fewer spills, but 2 % slower end to end)
      const float qv = r0;
      r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5; r5 = r6; r6 = r7; r7 = qv;
#endif
      const float* wr = WB + (d * 16 + mm) * WPITCH + 4 * cg;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 w = *reinterpret_cast<const float4*>(wr + 16 * nt);
        Qb[4 * nt] = fmaf(qv, w.x, Qb[4 * nt]);
        Qb[4 * nt + 1] = fmaf(qv, w.y, Qb[4 * nt + 1]);
        Qb[4 * nt + 2] = fmaf(qv, w.z, Qb[4 * nt + 2]);
        Qb[4 * nt + 3] = fmaf(qv, w.w, Qb[4 * nt + 3]);
      }
    }
#if !DD_LN_FOLD                                          // (the softmax scale 1 / sqrt(8) lives in the W2k image otherwise)
#pragma unroll
    for (int k = 0; k < 32; ++k) Qb[k] *= 0.35355339059327373f;
#endif
  }
  DD_STAMP(2);

  // ---- BL: angle codes of member 16t + mm, rows 4s + cg, kept in registers (MFMA feature operand) ------------
  float cod[MAXT][3];
  if (TRIP && active) {
    const float ax = tri_j[0] - tri_i[0], ay = tri_j[1] - tri_i[1], az = tri_j[2] - tri_i[2];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const float bx = tri_k[t][0] - tri_i[0], by = tri_k[t][1] - tri_i[1], bz = tri_k[t][2] - tri_i[2];
      const float dot = ax * bx + ay * by + az * bz;
      const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
      angle_codes(cx * cx + cy * cy + cz * cz, dot, cg, cod[t]);
    }
  }
  DD_STAMP(3);

  // first-layer table part of tile t (features of member 16t + mm), either orientation
  // (DD_GAUSS_CACHE, node mode: the Gaussians of a tile are formed in the k pass and kept for the v pass -- 5 registers per tile
  //  instead of 5 more expf per tile)
  constexpr bool GCACHE = DD_GAUSS_CACHE && KNN && !POS;
  float Fg[GCACHE ? MAXT : 1][5];
  auto table_part = [&](int t, int pass, f32x4 (&acc)[8], auto tr) {
    constexpr bool TR = decltype(tr)::value;
    if (KNN) {
      // Gaussian / type tables: F = 20 Gaussians + the per-type constant (24 rows = 6 k-steps).  A tile whose members
      // mix ligand and protein sources runs once per table with the other members' features zeroed.
      // DD_KNN_CONST_ADD: the sixth k-step only adds table row 20 (feature 1, rows 21-23 are zero rows) -- fma(1, w, acc) is
      // round(acc + w), and the steps of the OTHER table add exact zeros to a member, so the row of the member's own table added
      // on the VALU after the Gaussian steps gives the same bits: 8 (16 in a mixed tile) MFMAs less per tile pass.
      constexpr int KS = DD_KNN_CONST_ADD ? 5 : 6;
      float F[6];
      if (GCACHE) {
        if (pass == 0) {
#pragma unroll
          for (int s = 0; s < 5; ++s) Fg[t][s] = gauss_feat(dm[t], 4 * s + cg);
        }
#pragma unroll
        for (int s = 0; s < 5; ++s) F[s] = Fg[GCACHE ? t : 0][s];
      } else {
#pragma unroll
        for (int s = 0; s < 5; ++s) F[s] = gauss_feat(dm[t], 4 * s + cg);
      }
      F[5] = cg == 0 ? 1.0f : 0.0f;
      const bool hi = jm[t] < a.NP;
      const float* tab = smem + L::TAB + pass * 2 * TABP + cg * 128 + mm * 4;   // [lo, hi] tables of this pass
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const bool want = half ? hi : !hi;
        if (__builtin_amdgcn_ballot_w64(want) != 0ull) {
#pragma unroll
          for (int s = 0; s < KS; ++s) mfma_table_step<TR>(acc, tab + half * TABP + s * 512, want ? F[s] : 0.0f);
        }
      }
      if (DD_KNN_CONST_ADD) {
        // row 20 in the operand layout: channel 16 nt + i sits at (nt / 4) * 64 + i * 4 + nt % 4
        const float* row20 = smem + L::TAB + pass * 2 * TABP + 20 * 128;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // TR: acc[nt][r] = member 4cg + r, channel 16 nt + mm;  else: member mm, channel 16 nt + 4cg + r
          const bool hi_r = TR ? (jT[t][r] < a.NP) : hi;
          const float* p = row20 + (hi_r ? TABP : 0) + (TR ? mm : 4 * cg + r) * 4;
          const float4 c0 = *reinterpret_cast<const float4*>(p);
          const float4 c1 = *reinterpret_cast<const float4*>(p + 64);
          acc[0][r] += c0.x; acc[1][r] += c0.y; acc[2][r] += c0.z; acc[3][r] += c0.w;
          acc[4][r] += c1.x; acc[5][r] += c1.y; acc[6][r] += c1.z; acc[7][r] += c1.w;
        }
      }
    } else if (TRIP) {
      const float* tb = smem + WAO + pass * 12 * 128 + cg * 128 + mm * 4;
#pragma unroll
      for (int s = 0; s < 3; ++s) mfma_table_step<TR>(acc, tb + s * 512, cod[t][s]);
    }
  };

  // pre-activation of tile t for the k (pass = 0) or v (pass = 1) MLP: lane (mm, cg) holds member 16t + mm,
  // channels 16nt + 4cg + r  (P[4nt + r])
  auto build_pre = [&](int t, int pass, float (&P)[32]) {
    const int m = 16 * t + mm;
    const int mc = m < M ? m : M - 1;                  // clipped: out-of-range slots replay the last member
    const float* tab_d = pass ? a.vd : a.kd;  const int ld_d = pass ? a.ld_vd : a.ld_kd;
    const float* tab_s = pass ? a.vs : a.ks;  const int ld_s = pass ? a.ld_vs : a.ld_ks;
    const float* tab_e = pass ? a.ve : a.ke;  const int ld_e = pass ? a.ld_ve : a.ld_ke;
    if (pass == 0) {
      if (REREAD) {
#pragma unroll
        for (int k = 0; k < 32; ++k) P[k] = Pf[k];
        add_row(P, rc_row, cg);
#pragma unroll
        for (int k = 0; k < 32; ++k)
          if (BOND) P[k] += Pf2[k];
      } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          P[k] = Rc[k] + Pf[k];
          if (BOND) P[k] += Pf2[k];
        }
      }
      if (QUAD) { if (t + 2 < MAXT) fetch_k_rows(t + 2); }
      else if (t + 1 < MAXT && (DD_UNCOND_FETCH || t + 1 < T)) fetch_k_rows(t + 1);   // next tile's rows fly during this tile's arithmetic
                                                              // (unconditional: members are clipped, a tile beyond T re-reads row M - 1)
    } else if (KNN) {
      load_row(P, tab_d + drow * ld_d, cg);
      add_row(P, tab_s + (src_base + jm[t]) * ld_s, cg);
    } else if (!TRIP) {
      const int j = mc + (mc >= si ? 1 : 0);
      load_row(P, tab_d + drow * ld_d, cg);
      add_row(P, tab_s + (src_base + j) * ld_s, cg);
      add_row(P, tab_e + (erow0 + mc) * ld_e, cg);
    } else {
      const int k = trip_k(mc);
      const long kj = (long)b * Eb + sj * NLm1 + (k - (k > sj ? 1 : 0));
      load_row(P, (pass ? a.Rv : a.Rk) + (long)seg * 128, cg);
      add_row(P, tab_e + kj * ld_e, cg);
    }
    if (dbg && t == 0 && pass == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); DD_STAMP(11); }   // rows arrived
    if (KNN || TRIP) {
      f32x4 acc[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{P[4 * nt], P[4 * nt + 1], P[4 * nt + 2], P[4 * nt + 3]};
      table_part(t, pass, acc, std::false_type{});
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) { P[4 * nt] = acc[nt][0]; P[4 * nt + 1] = acc[nt][1]; P[4 * nt + 2] = acc[nt][2]; P[4 * nt + 3] = acc[nt][3]; }
    }
    if (dbg && t == 0 && pass == 0) { asm volatile("" : "+v"(P[0]), "+v"(P[31])); DD_STAMP(12); }           // table part done
    ln_relu32(P, smem + LNP + pass * 256, cg);
    if (dbg && t == 0 && pass == 0) { asm volatile("" : "+v"(P[0]), "+v"(P[31])); DD_STAMP(13); }           // LayerNorm done
  };

  // v-MLP hidden activation in the member-major layout the aggregation consumes without a transpose: lane (mm, cg)
  // holds members 16t + 4cg + r (r < 4), channels 16nt + mm  (Tz[4nt + r]).  fetch_T requests the rows of a tile
  // (issued one tile ahead), finish_T turns them into the activation.
  float Tc[8], Tr[32], Tr2[32];
  auto fetch_T = [&](int t) {
    const float* rs[4];
    const float* re[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = 16 * t + 4 * cg + r;
      const int mc = m < M ? m : M - 1;
      re[r] = nullptr;
      if (KNN) {
        rs[r] = a.vs + (src_base + jT[t < MAXT ? t : 0][r]) * a.ld_vs + mm;
      } else if (!TRIP) {
        const int j = mc + (mc >= si ? 1 : 0);
        rs[r] = a.vs + (src_base + j) * a.ld_vs + mm;
        re[r] = a.ve + (erow0 + mc) * a.ld_ve + mm;
      } else {
        const int k = trip_k(mc);
        rs[r] = a.ve + ((long)b * Eb + sj * NLm1 + (k - (k > sj ? 1 : 0))) * a.ld_ve + mm;
      }
    }
    if (t == 0) {
      const float* rc = TRIP ? a.Rv + (long)seg * 128 + mm : a.vd + drow * a.ld_vd + mm;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) Tc[nt] = rc[16 * nt];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        Tr[4 * nt + r] = rs[r][16 * nt];
        if (BOND) Tr2[4 * nt + r] = re[r][16 * nt];
      }
  };
  auto finish_T = [&](int t, float (&Tz)[32]) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        float v = Tc[nt] + Tr[4 * nt + r];
        if (BOND) v += Tr2[4 * nt + r];
        Tz[4 * nt + r] = v;
      }
    if (t + 1 < MAXT && (DD_UNCOND_FETCH || t + 1 < T)) fetch_T(t + 1);       // next tile's rows fly during this tile's arithmetic
    if (KNN || TRIP) {
      f32x4 acc[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{Tz[4 * nt], Tz[4 * nt + 1], Tz[4 * nt + 2], Tz[4 * nt + 3]};
      table_part(t, 1, acc, std::true_type{});
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) { Tz[4 * nt] = acc[nt][0]; Tz[4 * nt + 1] = acc[nt][1]; Tz[4 * nt + 2] = acc[nt][2]; Tz[4 * nt + 3] = acc[nt][3]; }
    }
    ln_relu_T(Tz, smem + LNP + 256, mm);
  };

  // ---- kNN node mode: the W2k image is dead once every wave has folded its query; W2v takes its place now.  The
  //      second barrier follows at once (the waves are still aligned here; one in front of the epilogue would make
  //      every wave wait for the slowest); the Gaussian tables of both passes are resident.
  constexpr bool NECO = DD_NE_COOP_EPI && MODE == M_NE && !PERSIST && NW == 8;   // cooperative epilogue, no W2v image (see below)
  if (KNN && !POS) {
    __syncthreads();                                   // (NECO: every wave has folded its query -- the image's place becomes the exchange buffer)
    if (!NECO) {
      stage_w2k_permuted<NT>(WV, a.W2v);               // row o = h*8 + j  ->  LDS row j*16 + h
      __syncthreads();
    }
  }
  DD_STAMP(4);

  // ---- pass 1: scores S[t][r] = score[member 16t + 4cg + r][head mm] ---------------------------------------
  f32x4 S[MAXT];
  float ssum = 0.f;                                    // sum_m alpha*w for head mm
  if (active && kside) {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t < T && mine(t)) {
        float P[32];
        build_pre(t, 0, P);
        S[t] = mfma_rows(P, Qb);
        if (dbg && t == 0) { asm volatile("" : "+v"(S[t])); DD_STAMP(14); }                                      // scores done
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * t + 4 * cg + r >= M) S[t][r] = -INFINITY;
      } else {
        S[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
    }
    DD_STAMP(5);
    if (!POS) fetch_T(0);                              // v-pass rows of tile 0 arrive during the softmax
    if constexpr (!QUAD) {
    // segment softmax per head: max-shift, exp, / sum  (scatter_softmax), then * e_w
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, S[t][r]);
    mx = quad_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (16 * t + 4 * cg + r < M) ? expf(S[t][r] - mx) : 0.f;
        S[t][r] = e;
        sum += e;
      }
    sum = quad_sum(sum);
    const float rsum = 1.0f / sum;                       // one division per head instead of one per member
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * t + 4 * cg + r;
        float w = 1.0f;
        if (KNN) w = ewm[t][r];
#if defined(DD_EXACT_MATH) && DD_EXACT_MATH
        const float aw = (m < M) ? (S[t][r] / sum) * w : 0.f;        // scatter_softmax divides per member
#else
        const float aw = (m < M) ? (S[t][r] * rsum) * w : 0.f;
#endif
        S[t][r] = aw;
        ssum += aw;
      }
    ssum = quad_sum(ssum);
    }
  } else {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // QUAD: the two k-side waves of a segment hold the scores of different tiles -- the per-head max and sum cross through LDS
  float* const exq = smem + L::TOTAL + 2 * NW * 256 + NW * MAXT * 256;   // [NW][2][16] max, [NW][2][16] sum, [NW][2][4] dx partials
  if constexpr (QUAD) {
    const bool ks = active && kside;
    float mx = -INFINITY;
    if (ks) {
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, S[t][r]);
      mx = quad_max(mx);
      if (cg == 0) exq[(wave * 2 + tsel) * 16 + mm] = mx;
    }
    __syncthreads();
    float sum = 0.f;
    if (ks) {
      mx = fmaxf(exq[(wave * 2) * 16 + mm], exq[(wave * 2 + 1) * 16 + mm]);
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = (16 * t + 4 * cg + r < M && mine(t)) ? expf(S[t][r] - mx) : 0.f;
          S[t][r] = e;
          sum += e;
        }
      sum = quad_sum(sum);
      if (cg == 0) exq[NW * 32 + (wave * 2 + tsel) * 16 + mm] = sum;
    }
    __syncthreads();
    if (ks) {
      sum = exq[NW * 32 + (wave * 2) * 16 + mm] + exq[NW * 32 + (wave * 2 + 1) * 16 + mm];   // (the same order in both waves)
      const float rsum = 1.0f / sum;
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * t + 4 * cg + r;
          float w = 1.0f;
          if (KNN) w = ewm[t][r];
          S[t][r] = (m < M && mine(t)) ? (S[t][r] * rsum) * w : 0.f;
        }
    }
  }
  DD_STAMP(6);
  DD_STAMP(7);

  // ---- pass 2 -----------------------------------------------------------------------------------------------
  if (POS) {
    // wave pairs: the weights go from the k side to the v side through LDS, [tile][r][lane] per pair, behind one barrier
    float* xw = smem + L::TOTAL + (QUAD ? 2 : 1) * NW * 256 + wave * (MAXT * 4 * 64);
    f32x4 Vt[MAXT];                                    // V[r] = v16[member 16t+4cg+r][head mm] - bias
    if (active && vside) {
      float Wv[32];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 w = *reinterpret_cast<const float4*>(a.W2v16 + mm * 128 + 16 * nt + 4 * cg);
        Wv[4 * nt] = w.x; Wv[4 * nt + 1] = w.y; Wv[4 * nt + 2] = w.z; Wv[4 * nt + 3] = w.w;
      }
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t < T && mine(t)) {
          float P[32];
          build_pre(t, 1, P);
          Vt[t] = mfma_rows(P, Wv);
        }
      }
    }
    if (PAIR) {
      if (active && kside) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
          if (mine(t)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) xw[(t * 4 + r) * 64 + lane] = S[t][r];
          }
      }
      __syncthreads();
      if (!QUAD && role == 0) { DD_STAMP(10); return; }
      if (active && vside) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
          if (mine(t)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) S[t][r] = xw[(t * 4 + r) * 64 + lane];
          }
      }
    }
    float dx = 0.f, dy = 0.f, dz = 0.f;
    if (active && vside) {
      const float bv = a.b2v16[mm];
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t < T && mine(t)) {
          const f32x4 V = Vt[t];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * t + 4 * cg + r;
            const int mc = m < M ? m : 0;
            int j;
            const float* xs;
            if (KNN) { j = jT[t][r]; xs = xb; }
            else { j = mc + (mc >= si ? 1 : 0); xs = xl; }
            const int ii = KNN ? node : si;
            const float coef = S[t][r] * (V[r] + bv);   // S holds alpha*w (0 for out-of-range members)
            dx = fmaf(coef, xs[3 * ii] - xs[3 * j], dx);
            dy = fmaf(coef, xs[3 * ii + 1] - xs[3 * j + 1], dy);
            dz = fmaf(coef, xs[3 * ii + 2] - xs[3 * j + 2], dz);
          }
        }
      }
      dx = wave_sum(dx) * (1.0f / 16.0f);
      dy = wave_sum(dy) * (1.0f / 16.0f);
      dz = wave_sum(dz) * (1.0f / 16.0f);
      if (QUAD && lane < 3) exq[NW * 64 + (wave * 2 + tsel) * 4 + lane] = lane == 0 ? dx : (lane == 1 ? dy : dz);
    }
    if (QUAD) __syncthreads();                         // (the two halves of the update; every wave of the workgroup arrives here)
    if (active && vside && (!QUAD || tsel == 0)) {
      if (QUAD) {                                      // even tiles + odd tiles, in this order
        dx = exq[NW * 64 + (wave * 2) * 4 + 0] + exq[NW * 64 + (wave * 2 + 1) * 4 + 0];
        dy = exq[NW * 64 + (wave * 2) * 4 + 1] + exq[NW * 64 + (wave * 2 + 1) * 4 + 1];
        dz = exq[NW * 64 + (wave * 2) * 4 + 2] + exq[NW * 64 + (wave * 2 + 1) * 4 + 2];
      }
      if (lane < 3) {
        const float v = lane == 0 ? dx : (lane == 1 ? dy : dz);
        if (MODE == M_PE || a.x_next == nullptr) {
          a.out[(long)seg * 3 + lane] = v;            // delta only; x is advanced by k_xupdate
        } else {
          const long xi = ((long)b * N + node) * 3 + lane;
          a.x_next[xi] = a.x[xi] + a.dxe[(long)seg * 3 + lane] + v;
        }
      }
    }
    DD_STAMP(10);
    return;
  }

  // aggregation  Z[nt][r] = Z~[head mm][channel 16nt + 4cg + r] = sum_m aw[m][mm] * z_v[m][c]:  A = z_v in the
  // member-major layout (row = channel 16nt + mm, k = member 16t + 4cg + ks), B = alpha*w of the same member as
  // produced by pass 1 (S[t][ks]) -- no transposes, no LDS
  f32x4 Z[8];
#if DD_PEEL_Z
  if (active && T > 0) {                                 // the first k-step of tile 0 starts the chains from an inline zero
    {
      float Tz[32];
      finish_T(0, Tz);
      const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) Z[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Tz[4 * nt], S[0][0], zero4, 0, 0, 0);
#pragma unroll
      for (int ks = 1; ks < 4; ++ks)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          Z[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Tz[4 * nt + ks], S[0][ks], Z[nt], 0, 0, 0);
    }
#pragma unroll
    for (int t = 1; t < MAXT; ++t) {
      if (t < T) {
        float Tz[32];
        finish_T(t, Tz);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int nt = 0; nt < 8; ++nt)
            Z[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Tz[4 * nt + ks], S[t][ks], Z[nt], 0, 0, 0);
      }
    }
  } else {                                               // (idle wave, or a bond pair without a third atom: NL = 2)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) Z[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#else
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) Z[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (active) {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t < T) {
        float Tz[32];
        finish_T(t, Tz);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int nt = 0; nt < 8; ++nt)
            Z[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Tz[4 * nt + ks], S[t][ks], Z[nt], 0, 0, 0);
      }
    }
  }
#endif
  DD_STAMP(8);

  // ---- epilogue: out[o] = W2v[o,:] . Z~[head(o),:] + b2v[o] * sum_m alpha*w ----------------------------------
  DD_STAMP(9);
  if (PERSIST) {                                       // next trip's query
    while (__hip_atomic_load(&sb[8], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < it + 1) {}
    int base2, cnt2;
    trip_range(sb[(it + 1) & 7], base2, cnt2);
    if (wave < cnt2) {
      const int nseg2 = bl_dense_seg(base2 + wave);
      qn0 = *reinterpret_cast<const float4*>(a.q + (long)nseg2 * 128 + mm * 8);
      qn1 = *reinterpret_cast<const float4*>(a.q + (long)nseg2 * 128 + mm * 8 + 4);
    }
  }
  // lin_node inside the node launch (a.lin_W != NULL; NE / NB blocks of the fused launch, round 6): this wave's 16 rows of W_lin
  // as the MFMA B operand, requested ahead of the epilogue (see below)
  constexpr bool LIN = (MODE == M_NE || MODE == M_NB) && !PERSIST;
  const bool lin = LIN && a.lin_W != nullptr;            // (block-uniform)
  float4 Bw[8];
  float lin_old[4] = {0.f, 0.f, 0.f, 0.f};              // rows 4 cg + r of the block, column 16 wave + mm: h (+ the pending W_lin . A_nb) + bias
  long lin_off[4] = {-1, -1, -1, -1};
  if (lin) {
    const float* bw = a.lin_W + (16 * wave + mm) * 128 + 4 * cg;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) Bw[nt] = *reinterpret_cast<const float4*>(bw + 16 * nt);
    if (cg < 2) {                                        // (requested here, ahead of the epilogue: no dependent load behind the barrier)
      const int col = 16 * wave + mm;
      const float bias = (MODE == M_NE) ? a.lin_b[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * cg + r;                        // segment of the block
        if (MODE == M_NE) {
          const int nd = wg_protein ? ne_rb * NW + i : a.NP + (ne_rb - ne_nbp) * NW + i;
          bool ok = nd < (wg_protein ? a.NP : N) && ne_b < a.B;
          if (RAG && ok) ok = nd < a.NP ? nd < (a.np_real ? a.np_real[ne_b] : a.NP) : nd - a.NP < a.nl_real[ne_b];
          if (ok) {
            lin_off[r] = ((long)ne_b * N + nd) * 128 + col;
            float v = a.out[lin_off[r]];
            if (nd >= a.NP && a.lin_add != nullptr) v += a.lin_add[((long)ne_b * a.NL + nd - a.NP) * 128 + col];
            lin_old[r] = v + bias;
          }
        } else {
          const int sg = block * NW + i;
          bool ok = sg < a.B * a.NL;
          if (RAG && ok) ok = sg % a.NL < a.nl_real[sg / a.NL];
          if (ok) lin_off[r] = (long)sg * 128 + col;
        }
      }
    }
  }
  float* const lin_rows = smem + L::TOTAL + 16;          // [NW][WPITCH]: the attention outputs of the block's NW segments
  if constexpr (NECO) {
    // Cooperative epilogue of a node block (as bl_coop_body's): every wave leaves its Z~ [16 heads][128] in the exchange buffer that
    // took the W2k image's place; wave w then forms out[s][16 w .. 16 w + 15] for the block's 8 segments as ONE 16 x 16 x 128 MFMA
    // chain -- rows i = 2 s + h' (the two heads 2w, 2w + 1 of segment s), columns = the 16 outputs of those heads, the two diagonal
    // 8-column blocks are the results -- with its 16 rows of W2v as the B operand straight from L2 (requested before the barrier).
    // No W2v image: one barrier instead of a 64 KB re-staging between two, and 8 KB instead of 64 KB of LDS reads per wave.
    constexpr int XSTR = 16 * WPITCH + 16;
    float* const XB = smem;
    float* const SS = lin_rows + NW * WPITCH;            // [8][16] sums of the attention weights
    float4 Bv[8];
    {
      const float* bv = a.W2v + (16 * wave + mm) * 128 + 4 * cg;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) Bv[nt] = *reinterpret_cast<const float4*>(bv + 16 * nt);
    }
    const int eh = mm >> 3;
    const float bias2 = a.b2v[16 * wave + mm];
    if (active) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
        *reinterpret_cast<float4*>(XB + wave * XSTR + mm * WPITCH + 16 * nt + 4 * cg) = make_float4(Z[nt][0], Z[nt][1], Z[nt][2], Z[nt][3]);
      if (cg == 0) SS[wave * 16 + mm] = ssum;
    }
    __syncthreads();
    const float* ar = XB + (mm >> 1) * XSTR + (2 * wave + (mm & 1)) * WPITCH + 4 * cg;
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < 8; nt += 2) {
      const float4 a0 = *reinterpret_cast<const float4*>(ar + 16 * nt), a1 = *reinterpret_cast<const float4*>(ar + 16 * nt + 16);
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, Bv[nt].x, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, Bv[nt + 1].x, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, Bv[nt].y, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, Bv[nt + 1].y, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, Bv[nt].z, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, Bv[nt + 1].z, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, Bv[nt].w, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, Bv[nt + 1].w, d1, 0, 0, 0);
    }
    const f32x4 d = d0 + d1;                             // rows 4 cg + r <-> (s = 2 cg + (r >> 1), h' = r & 1); useful: h' == eh
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int sI = 2 * cg + u;                         // segment of the block
      const int nd = wg_protein ? ne_rb * NW + sI : a.NP + (ne_rb - ne_nbp) * NW + sI;
      bool ok = nd < (wg_protein ? a.NP : N) && ne_b < a.B;
      if (RAG && ok) ok = nd < a.NP ? nd < (a.np_real ? a.np_real[ne_b] : a.NP) : nd - a.NP < a.nl_real[ne_b];
      const float val = fmaf(bias2, SS[sI * 16 + 2 * wave + eh], eh ? d[2 * u + 1] : d[2 * u]);
      if (lin) lin_rows[sI * WPITCH + 16 * wave + mm] = val;       // (rows of idle segments: finite garbage, never stored)
      else if (ok) a.out[((long)ne_b * N + nd) * 128 + 16 * wave + mm] = val;
    }
  } else
  if (active) {
    // lane (h = mm, cg): partial dot products over its 32 channels for the 8 outputs of head h
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* wr = WV + (j * 16 + mm) * WPITCH + 4 * cg;
      float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
      for (int nt = 0; nt < 8; nt += 2) {
        const float4 w0 = *reinterpret_cast<const float4*>(wr + 16 * nt);
        const float4 w1 = *reinterpret_cast<const float4*>(wr + 16 * nt + 16);
        acc0 = fmaf(w0.x, Z[nt][0], acc0); acc0 = fmaf(w0.y, Z[nt][1], acc0);
        acc0 = fmaf(w0.z, Z[nt][2], acc0); acc0 = fmaf(w0.w, Z[nt][3], acc0);
        acc1 = fmaf(w1.x, Z[nt + 1][0], acc1); acc1 = fmaf(w1.y, Z[nt + 1][1], acc1);
        acc1 = fmaf(w1.z, Z[nt + 1][2], acc1); acc1 = fmaf(w1.w, Z[nt + 1][3], acc1);
      }
      o[j] = acc0 + acc1;
      __builtin_amdgcn_sched_barrier(0);
    }
    // reduce over the 4 lanes of a head and scatter: lane cg ends with outputs j = cg and j = 4 + cg
    const float p0 = swap16_sum(o[0], o[1]), p1 = swap16_sum(o[2], o[3]);
    const float p2 = swap16_sum(o[4], o[5]), p3 = swap16_sum(o[6], o[7]);
    float q0 = swap32_sum(p0, p1), q1 = swap32_sum(p2, p3);
    q0 = fmaf(a.b2v[mm * 8 + cg], ssum, q0);
    q1 = fmaf(a.b2v[mm * 8 + 4 + cg], ssum, q1);
    const bool nb_assign = (MODE == M_NB) && a.out_assign;
    if (lin) {
      lin_rows[wave * WPITCH + mm * 8 + cg] = q0;
      lin_rows[wave * WPITCH + mm * 8 + 4 + cg] = q1;
    } else {
      float* dst;
      if (MODE == M_NB && !nb_assign) dst = a.out + ((long)b * N + node) * 128 + mm * 8 + cg;
      else dst = a.out + (long)seg * 128 + mm * 8 + cg;
      if (MODE == M_NE || nb_assign) {
        dst[0] = q0; dst[4] = q1;
      } else {
        dst[0] += q0; dst[4] += q1;
      }
    }
  }
  if constexpr (LIN) {
    if (lin) {
      // h' = h + W_lin . A + b_lin (uni_transformer_edge.py:276-277) for the block's NW nodes as ONE 16 x 16 x 128 MFMA chain per
      // wave: rows = the block's segments (rows 8-15 repeat 0-7), columns = outputs 16 wave .. 16 wave + 15.  NE blocks update
      // their rows of h in place (no launch reads h while the node launch runs); NB blocks store W_lin . A_nb to their own
      // buffer (a.out), which the consumers of the new h add to the ligand rows (GemmArgs::X2) and the next layer's NE blocks
      // fold into h (a.lin_add) -- one writer per row, a fixed order of the sums: deterministic.
      __syncthreads();
      const float* ar = lin_rows + (mm & 7) * WPITCH + 4 * cg;
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 8; nt += 2) {
        const float4 a0 = *reinterpret_cast<const float4*>(ar + 16 * nt), a1 = *reinterpret_cast<const float4*>(ar + 16 * nt + 16);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, Bw[nt].x, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, Bw[nt + 1].x, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, Bw[nt].y, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, Bw[nt + 1].y, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, Bw[nt].z, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, Bw[nt + 1].z, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, Bw[nt].w, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, Bw[nt + 1].w, d1, 0, 0, 0);
      }
      const f32x4 d = d0 + d1;                           // lane (column mm, cg): rows 4 cg + r
      if (cg < 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (lin_off[r] >= 0) a.out[lin_off[r]] = (MODE == M_NE) ? lin_old[r] + d[r] : d[r];
      }
    }
  }
  DD_STAMP(10);
  if (!PERSIST) break;
  ++it;
  }
#undef DD_STAMP
}

// ---- persistent bond-layer workgroups with a COOPERATIVE query fold and epilogue (round 6, EXPERIMENTS.md R6-2) ------------------
// In attn2_body every wave folds its own query (Q~ = q . blockdiag(W2k): each of the 16 K weights used once per segment) and runs
// its own epilogue (W2v . Z~: the same), streaming the whole 64 KB W2k / W2v image through the LDS port shared by 8 in-phase waves:
// 1 MiB of LDS reads per trip of 8 segments, ~8 k cycles at the port's 128 B/clk for ~3 k cycles of arithmetic.  Here the 8 waves
// of a trip split both by HEAD instead: wave w owns heads 2w, 2w + 1 of all 8 segments.
//   * fold: the 16 W2k rows of its heads live in 32 registers for the workgroup's lifetime (no W2k image in LDS at all); the 8
//     queries of the trip are published in LDS (4 KB, broadcast reads); the 8 x 2 x 128 products go to an exchange buffer that
//     takes the W2k image's place, and every wave reads the Q~ of ITS segment back in the MFMA B-operand layout (8 x 16 bytes).
//   * epilogue: every wave writes its Z~ (16 heads x 128 channels) to the same buffer; wave w forms out[s][16w .. 16w + 15] for the 8
//     segments as ONE 16 x 16 x 128 MFMA chain (rows = (segment, head of the pair), columns = the 16 outputs of its two heads;
//     the two diagonal 8-column blocks are the results), W2v as the B operand from a natural-order LDS image (8 KB per wave).
//   LDS traffic per trip: 64 (queries, broadcast) + 64 + 64 + 64 + 64 + 64 KB instead of 1 MiB.
//   * the NEXT trip's prologue (segment indices, query, gather rows of its first k-pass tile, triplet geometry) is issued in front
//     of the epilogue, so those loads fly during the epilogue, the trip barrier and the next fold.
// Exchange buffer addressing: (s, h, c) -> s * XSTR + h * WPITCH + c with XSTR = 16 * WPITCH + 16: writers (lane = head, fixed
// segment), the Q~ readers (the same) and the epilogue's A-operand readers (lane i = 2 * segment-in-pair... row i = 2s + h') all
// hit 16-byte bank slots 2 * (lane & 15) + (lane >> 4) (mod 16), the conflict-free pattern of the weight images (see WPITCH).
struct CoopLds {
  static constexpr int XSTR = 16 * WPITCH + 16;
  static constexpr int XB = 0;                          // exchange buffer [8 segments][16 heads][WPITCH] (+ 16 floats of skew per segment)
  static constexpr int WV = 8 * XSTR;                   // W2v image, natural row order, pitch WPITCH
  static constexpr int LNP = WV + WB_FLOATS;            // [4][128] LayerNorm parameters (k: gamma, beta; v: gamma, beta)
  static constexpr int WAO = LNP + 512;                 // [2][12][128] angle tables (MFMA operand layout)
  static constexpr int QS = WAO + 2 * 12 * 128;         // [8][128] queries of the trip
  static constexpr int SS = QS + 8 * 128;               // [8][16] sum of the attention weights per (segment, head)
  static constexpr int SG = SS + 8 * 16;                // ints: [8] dense segment ids of the trip, [8..10] trip indices (current, next)
  static constexpr int TOTAL = SG + 16;
};

template <int MAXT, bool RAG, bool STAMPS, typename ARGS>
__device__ __forceinline__ void bl_coop_body(const ARGS& a, float* smem) {
  constexpr int NW = 8, NT = NW * 64;
  using C = CoopLds;
  constexpr int XSTR = C::XSTR;
  const int wave = threadIdx.x >> 6, lane0 = threadIdx.x & 63;
  const int NLm1 = a.NL - 1, Eb = a.NL * NLm1;
  const int nseg = RAG ? a.bl_prefix[a.B] : a.B * Eb;
  float* const XB = smem + C::XB;
  float* const WV = smem + C::WV;
  float* const QS = smem + C::QS;
  float* const SS = smem + C::SS;
  int* const sg = reinterpret_cast<int*>(smem + C::SG);
  int* const sb = sg + 8;
  long long* const dbg0 = STAMPS ? a.dbg_clock : nullptr;
  long long* dbg = nullptr;
#define DD_STAMP(i) do { if (STAMPS && dbg && threadIdx.x == 0) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)

  // ---- once per workgroup: W2v image (natural order), LayerNorm parameters, angle tables; this wave's W2k rows in registers
  {
    constexpr int PER = 4096 / NT;
    float4 tmp[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) tmp[k] = reinterpret_cast<const float4*>(a.W2v)[threadIdx.x + k * NT];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = threadIdx.x + k * NT;
      *reinterpret_cast<float4*>(&WV[(i >> 5) * WPITCH + (i & 31) * 4]) = tmp[k];
    }
    if (threadIdx.x < 64) {
      reinterpret_cast<float4*>(smem + C::LNP)[threadIdx.x] = reinterpret_cast<const float4*>(a.lnk)[threadIdx.x];
      reinterpret_cast<float4*>(smem + C::LNP + 256)[threadIdx.x] = reinterpret_cast<const float4*>(a.lnv)[threadIdx.x];
    }
    for (int i = threadIdx.x; i < 12 * 32; i += NT) {
      reinterpret_cast<float4*>(smem + C::WAO)[i] = reinterpret_cast<const float4*>(a.Wakp)[i];
      reinterpret_cast<float4*>(smem + C::WAO + 12 * 128)[i] = reinterpret_cast<const float4*>(a.Wavp)[i];
    }
  }
  float Wk[32];                                          // W2k[16 wave + r][2 lane + {0, 1}]  (rows 8 h + d of heads 2 wave, 2 wave + 1)
  auto load_wk = [&](int lane) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 w = *reinterpret_cast<const float2*>(a.W2k + (16 * wave + r) * 128 + 2 * lane);
#if DD_LN_FOLD
      Wk[2 * r] = w.x; Wk[2 * r + 1] = w.y;
#else
      Wk[2 * r] = w.x * 0.35355339059327373f; Wk[2 * r + 1] = w.y * 0.35355339059327373f;
#endif
    }
  };
  load_wk(lane0);
  // (loop-invariant, and loaded HERE: a global load inside the epilogue would queue behind the next trip's gathers)
  const float ep_bias = a.b2v[(2 * wave + ((lane0 & 15) >> 3)) * 8 + (lane0 & 7)];
  if (threadIdx.x == 0) sb[0] = atomicAdd(a.work_counter, 1);
  __syncthreads();

  auto trip_range = [&](int tr, int& base, int& cnt) {
    if (tr < a.trip_full) { base = tr * NW; cnt = NW; }
    else { base = a.trip_full * NW + (tr - a.trip_full) * a.trip_q; cnt = a.trip_q; }
    cnt = nseg - base < cnt ? nseg - base : cnt;         // <= 0: no work left
  };
  auto bl_dense_seg = [&](int r) -> int {                // compact index of a real segment -> id in the dense (padded) enumeration
    if (!RAG) return r;
    int bb = 0;
    while (a.bl_prefix[bb + 1] <= r) ++bb;
    const int nm1 = a.nl_real[bb] - 1, e = r - a.bl_prefix[bb];
    return bb * Eb + (e / nm1) * NLm1 + (e % nm1);
  };

  // ---- state of the NEXT trip's segment, filled one trip ahead (fetch) --------------------------------------------------------
  int segN = 0, bN = 0, siN = 0, sjN = 0, MN = 0;
  bool actN = false;
  float2 q2 = make_float2(0.f, 0.f);
  float Pf[32];
#if DD_COOP_RC
  float Rc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) Rc[k] = 0.f;
#endif
  float tri_i[3] = {0.f, 0.f, 0.f}, tri_j[3] = {0.f, 0.f, 0.f}, tri_k[MAXT][3];
#pragma unroll
  for (int k = 0; k < 32; ++k) Pf[k] = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) tri_k[t][0] = tri_k[t][1] = tri_k[t][2] = 0.f;
  auto third_atom = [](int mc, int si, int sj) { const int lo = si < sj ? si : sj, hi = si < sj ? sj : si; int k = mc; if (k >= lo) ++k; if (k >= hi) ++k; return k; };
  auto fetch = [&](int tr, int lane) {
    int base, cnt;
    trip_range(tr, base, cnt);
    actN = wave < cnt;
    if (!actN) return;
    const int mm = lane & 15, cg = lane >> 4;
    segN = bl_dense_seg(base + wave);
    bN = segN / Eb;
    const int e = segN % Eb;
    siN = e / NLm1;
    const int jp = e % NLm1;
    sjN = jp + (jp >= siN ? 1 : 0);
    MN = RAG ? a.nl_real[bN] - 2 : a.NL - 2;
    q2 = *reinterpret_cast<const float2*>(a.q + (long)segN * 128 + 2 * lane);
    {                                                    // k-pass rows of tile 0: bond (k -> j) of member mm
      const int mc = mm < MN ? mm : (MN > 0 ? MN - 1 : 0);
      const int k = third_atom(mc, siN, sjN);
      load_row(Pf, a.ke + ((long)bN * Eb + sjN * NLm1 + (k - (k > sjN ? 1 : 0))) * a.ld_ke, cg);
    }
#if DD_COOP_RC
    load_row(Rc, a.Rk + (long)segN * 128, cg);
#endif
    const float* xl = a.x + ((long)bN * (a.NP + a.NL) + a.NP) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) { tri_i[c] = xl[3 * siN + c]; tri_j[c] = xl[3 * sjN + c]; }
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int m = 16 * t + mm;
      const int k = third_atom(m < MN ? m : MN - 1, siN, sjN);
#pragma unroll
      for (int c = 0; c < 3; ++c) tri_k[t][c] = xl[3 * k + c];
    }
  };
  fetch(__builtin_amdgcn_readfirstlane(sb[0]), lane0);

  for (int it = 0;; ++it) {
    // (the lane id is laundered per iteration: loop-invariant per-lane address arithmetic stays inside the iteration)
    int lane = lane0;
    asm volatile("" : "+v"(lane) :: "memory");
    const int mm = lane & 15, cg = lane >> 4;

    // ---- 1. publish this wave's query and segment id; barrier A (also: the previous trip's epilogue reads of XB are done)
    if (actN) *reinterpret_cast<float2*>(QS + wave * 128 + 2 * lane) = q2;
    if (lane == 0) sg[wave] = segN;
    __syncthreads();
    int base, cnt;
    trip_range(__builtin_amdgcn_readfirstlane(sb[it & 1]), base, cnt);
    if (cnt <= 0) break;
    if (threadIdx.x == 0) sb[(it + 1) & 1] = atomicAdd(a.work_counter, 1);
    if (STAMPS && dbg0) dbg = dbg0 + (long)__builtin_amdgcn_readfirstlane(sb[it & 1]) * 16;
    DD_STAMP(0);

    const int seg = segN, b = bN, si = siN, sj = sjN;
    const bool active = actN;
    const int M = MN, T = (M + 15) >> 4;

    // old values of the rows this wave will update in the epilogue (out[s][8 (2 wave + h) + j] for s = 2 (lane >> 4) + {0, 1}):
    // requested now, ahead of every gather of the trip (loads return in order)
    const int eh = mm >> 3, ej = mm & 7;                 // epilogue: column = output 8 eh + ej of head 2 wave + eh
    float old0 = 0.f, old1 = 0.f;
    float* dst0 = nullptr;
    float* dst1 = nullptr;
    {
      const int s0 = 2 * cg, s1 = 2 * cg + 1;
      if (s0 < cnt) { dst0 = a.out + (long)sg[s0] * 128 + (2 * wave + eh) * 8 + ej; old0 = *dst0; }
      if (s1 < cnt) { dst1 = a.out + (long)sg[s1] * 128 + (2 * wave + eh) * 8 + ej; old1 = *dst1; }
    }

    // ---- 2. cooperative fold: Q~[s][h][2 lane + {0, 1}] for h = 2 wave, 2 wave + 1 and the 8 segments of the trip
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      float qv[16];
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const float4 v = *reinterpret_cast<const float4*>(QS + s * 128 + 16 * wave + 4 * k4);    // (wave-uniform address: broadcast)
        qv[4 * k4] = v.x; qv[4 * k4 + 1] = v.y; qv[4 * k4 + 2] = v.z; qv[4 * k4 + 3] = v.w;
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float a0 = qv[8 * hh] * Wk[16 * hh], a1 = qv[8 * hh] * Wk[16 * hh + 1];
#pragma unroll
        for (int d = 1; d < 8; ++d) {
          a0 = fmaf(qv[8 * hh + d], Wk[16 * hh + 2 * d], a0);
          a1 = fmaf(qv[8 * hh + d], Wk[16 * hh + 2 * d + 1], a1);
        }
        *reinterpret_cast<float2*>(XB + s * XSTR + (2 * wave + hh) * WPITCH + 2 * lane) = make_float2(a0, a1);
      }
    }
    DD_STAMP(1);
    __syncthreads();                                     // barrier C
    DD_STAMP(2);
    // ---- 3. Q~ of this wave's segment as the MFMA B operand: lane (h = mm, cg) holds Q~[h][16 nt + 4 cg + r]
    float Qb[32];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const float4 v = *reinterpret_cast<const float4*>(XB + wave * XSTR + mm * WPITCH + 16 * nt + 4 * cg);
      Qb[4 * nt] = v.x; Qb[4 * nt + 1] = v.y; Qb[4 * nt + 2] = v.z; Qb[4 * nt + 3] = v.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                     // barrier D: XB is free for the epilogue's exchange
    DD_STAMP(3);

    // ---- 4. the segment itself: angle codes, k pass, softmax, v pass (as attn2_body<M_BL>) -----------------------------------
    const long erow_j = (long)b * Eb + (long)sj * NLm1;   // first bond row of destination j in the dst-major tables
    auto kj_row = [&](int m) {                           // row of bond (k -> j) for member m (clipped)
      const int k = third_atom(m < M ? m : M - 1, si, sj);
      return erow_j + (k - (k > sj ? 1 : 0));
    };
    float cod[MAXT][3];
    f32x4 S[MAXT];
    float ssum = 0.f;
    f32x4 Z[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) Z[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
      const float ax = tri_j[0] - tri_i[0], ay = tri_j[1] - tri_i[1], az = tri_j[2] - tri_i[2];
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const float bx = tri_k[t][0] - tri_i[0], by = tri_k[t][1] - tri_i[1], bz = tri_k[t][2] - tri_i[2];
        const float dot = ax * bx + ay * by + az * bz;
        const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
        angle_codes(cx * cx + cy * cy + cz * cz, dot, cg, cod[t]);
      }
      DD_STAMP(4);
      const float* rk_row = a.Rk + (long)seg * 128;
      // k pass: lane (mm, cg) = member 16 t + mm, channels 16 nt + 4 cg + r
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t < T) {
          float P[32];
#if DD_COOP_RC
#pragma unroll
          for (int k = 0; k < 32; ++k) P[k] = Pf[k] + Rc[k];
#else
#pragma unroll
          for (int k = 0; k < 32; ++k) P[k] = Pf[k];
          add_row(P, rk_row, cg);
#endif
          if (t + 1 < MAXT) load_row(Pf, a.ke + kj_row(16 * (t + 1) + mm) * a.ld_ke, cg);   // next tile's rows (clipped members)
          if (STAMPS && dbg && t == 0) { asm volatile("" : "+v"(P[0]), "+v"(P[31])); DD_STAMP(13); }      // segment row arrived
          f32x4 acc[8];
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{P[4 * nt], P[4 * nt + 1], P[4 * nt + 2], P[4 * nt + 3]};
          const float* tb = smem + C::WAO + cg * 128 + mm * 4;
#pragma unroll
          for (int s3 = 0; s3 < 3; ++s3) mfma_table_step<false>(acc, tb + s3 * 512, cod[t][s3]);
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) { P[4 * nt] = acc[nt][0]; P[4 * nt + 1] = acc[nt][1]; P[4 * nt + 2] = acc[nt][2]; P[4 * nt + 3] = acc[nt][3]; }
          ln_relu32(P, smem + C::LNP, cg);
          if (STAMPS && dbg && t == 0) { asm volatile("" : "+v"(P[0]), "+v"(P[31])); DD_STAMP(14); }      // table part + LayerNorm done
          S[t] = mfma_rows(P, Qb);
          if (STAMPS && dbg && t == 0) { asm volatile("" : "+v"(S[t])); DD_STAMP(15); }                     // scores of tile 0 done
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * t + 4 * cg + r >= M) S[t][r] = -INFINITY;
        } else {
          S[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
      }
      DD_STAMP(5);
      // v-pass rows (member-major layout: lane (mm, cg) holds members 16 t + 4 cg + r, channels 16 nt + mm), one tile ahead
      float Tc[8], Tr[32];
      auto fetch_T = [&](int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* rs = a.ve + kj_row(16 * t + 4 * cg + r) * a.ld_ve + mm;
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) Tr[4 * nt + r] = rs[16 * nt];
        }
      };
      {
        const float* rc = a.Rv + (long)seg * 128 + mm;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) Tc[nt] = rc[16 * nt];
      }
      fetch_T(0);
      // segment softmax per head (scatter_softmax): max-shift, exp, one reciprocal per head
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, S[t][r]);
      mx = quad_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = (16 * t + 4 * cg + r < M) ? expf(S[t][r] - mx) : 0.f;
          S[t][r] = e;
          sum += e;
        }
      sum = quad_sum(sum);
#if defined(DD_EXACT_MATH) && DD_EXACT_MATH
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { S[t][r] = (16 * t + 4 * cg + r < M) ? S[t][r] / sum : 0.f; ssum += S[t][r]; }
#else
      const float rsum = 1.0f / sum;
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { S[t][r] = (16 * t + 4 * cg + r < M) ? S[t][r] * rsum : 0.f; ssum += S[t][r]; }
#endif
      ssum = quad_sum(ssum);
      DD_STAMP(6);
      // v pass + aggregation  Z[nt][r] = Z~[head mm][channel 16 nt + 4 cg + r]
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t < T) {
          float Tz[32];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) Tz[4 * nt + r] = Tc[nt] + Tr[4 * nt + r];
          if (t + 1 < MAXT) fetch_T(t + 1);
          f32x4 acc[8];
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{Tz[4 * nt], Tz[4 * nt + 1], Tz[4 * nt + 2], Tz[4 * nt + 3]};
          const float* tb = smem + C::WAO + 12 * 128 + cg * 128 + mm * 4;
#pragma unroll
          for (int s3 = 0; s3 < 3; ++s3) mfma_table_step<true>(acc, tb + s3 * 512, cod[t][s3]);
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) { Tz[4 * nt] = acc[nt][0]; Tz[4 * nt + 1] = acc[nt][1]; Tz[4 * nt + 2] = acc[nt][2]; Tz[4 * nt + 3] = acc[nt][3]; }
          ln_relu_T(Tz, smem + C::LNP + 256, mm);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
              Z[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Tz[4 * nt + ks], S[t][ks], Z[nt], 0, 0, 0);
        }
      }
    }
    DD_STAMP(7);

    // ---- 5. the next trip's prologue: its loads fly during the epilogue, barrier A and the next fold ---------------------------
    fetch(__builtin_amdgcn_readfirstlane(sb[(it + 1) & 1]), lane);
#if DD_COOP_WK_RELOAD
    load_wk(lane);
#endif
    DD_STAMP(8);

    // ---- 6. cooperative epilogue ------------------------------------------------------------------------------------------------
    if (active) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
        *reinterpret_cast<float4*>(XB + wave * XSTR + mm * WPITCH + 16 * nt + 4 * cg) = make_float4(Z[nt][0], Z[nt][1], Z[nt][2], Z[nt][3]);
      if (cg == 0) SS[wave * 16 + mm] = ssum;
    }
    DD_STAMP(9);
    __syncthreads();                                     // barrier E
    DD_STAMP(10);
    {
      // rows i = 2 s + h' (i = lane & 15), k = channel 16 nt + 4 (lane >> 4) + r4;  columns = outputs 16 wave + (lane & 15)
      const float* ar = XB + (mm >> 1) * XSTR + (2 * wave + (mm & 1)) * WPITCH + 4 * cg;
      const float* br = WV + (16 * wave + mm) * WPITCH + 4 * cg;
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 8; nt += 2) {
        const float4 a0 = *reinterpret_cast<const float4*>(ar + 16 * nt), b0 = *reinterpret_cast<const float4*>(br + 16 * nt);
        const float4 a1 = *reinterpret_cast<const float4*>(ar + 16 * nt + 16), b1 = *reinterpret_cast<const float4*>(br + 16 * nt + 16);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, b0.x, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, b1.x, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, b0.y, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, b1.y, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, b0.z, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, b1.z, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, b0.w, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, b1.w, d1, 0, 0, 0);
      }
      const f32x4 d = d0 + d1;                           // rows 4 cg + r <-> (s = 2 cg + (r >> 1), h' = r & 1); useful: h' == eh
      DD_STAMP(11);
      const float bias = ep_bias;
      if (dst0) *dst0 = old0 + fmaf(bias, SS[(2 * cg) * 16 + 2 * wave + eh], eh ? d[1] : d[0]);
      if (dst1) *dst1 = old1 + fmaf(bias, SS[(2 * cg + 1) * 16 + 2 * wave + eh], eh ? d[3] : d[2]);
    }
    DD_STAMP(12);
  }
#undef DD_STAMP
}

// ---- persistent node_layer_with_edge workgroups (round 6, EXPERIMENTS.md R6-8) ------------------------------------------------------
// The NE blocks of attn2_body stage 112 KB of weight images for every 8 nodes (18 % of a block's time) and start and end as workgroups of
// their own.  Here a workgroup stages ONLY the Gaussian tables of its centre kind (protein or ligand) and the LayerNorm rows, once, and
// then pulls blocks of 8 nodes of that kind from a counter:
//   * no W2k image: the query fold is split by head over the 8 waves (as bl_coop_body: wave w keeps the 16 W2k rows of heads 2w, 2w + 1
//     in registers, the block's 8 queries are published in LDS, the products cross through the exchange buffer);
//   * no W2v image: the epilogue is one MFMA chain per wave with its 16 W2v rows straight from L2 (as the NE blocks' NECO form), Z~
//     through the same exchange buffer.  (Launches that carry lin_node, a.lin_W, keep the block form: launch_node_nw.)
// LDS: exchange buffer 70 KB + tables 48 KB + small = 127 KB.  Same arithmetic in the same order as the block form: bit-identical.
struct NepLds {
  static constexpr int XSTR = 16 * WPITCH + 16;
  static constexpr int XB = 0;
  static constexpr int TAB = 8 * XSTR;                  // [k lo, k hi, v lo, v hi] tables of this centre kind (4 x 24 x 128)
  static constexpr int LNP = TAB + 4 * TABP;            // [4][128]
  static constexpr int QS = LNP + 512;                  // [8][128] queries of the block
  static constexpr int ROWS = QS + 8 * 128;             // [8][WPITCH] attention outputs of the block (lin_node)
  static constexpr int SS = ROWS + 8 * WPITCH;          // [8][16]
  static constexpr int SB = SS + 128;                   // ints: block indices (current, next)
  static constexpr int TOTAL = SB + 16;
};

template <bool RAG, typename ARGS>
__device__ __forceinline__ void ne_persist_body(const ARGS& a, float* smem, const bool wg_protein, int32_t* counter) {
  constexpr int NW = 8, NT = NW * 64, MAXT = 2;
  using C = NepLds;
  constexpr int XSTR = C::XSTR;
  const int wave = threadIdx.x >> 6, lane0 = threadIdx.x & 63;
  const int N = a.NP + a.NL;
  const int nb_kind = wg_protein ? (a.NP + NW - 1) / NW : (a.NL + NW - 1) / NW;     // blocks of this kind per sample
  const int n_blocks = a.B * nb_kind;
  float* const XB = smem + C::XB;
  float* const QS = smem + C::QS;
  float* const SS = smem + C::SS;
  int* const sb = reinterpret_cast<int*>(smem + C::SB);
  const int M = a.K, T = (M + 15) >> 4;

  // ---- once per workgroup: tables of this centre kind, LayerNorm rows; this wave's W2k rows in registers
  {
    const int tyl = wg_protein ? 1 : 0;                  // edge type = 2 * (source is protein) + (centre is protein)
    constexpr int PER = (4 * 768) / NT;
    float4 tmp[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = threadIdx.x + k * NT, q = i / 768, w = i - q * 768;
      const float* src = (q < 2 ? a.Akp : a.Avp) + (tyl + 2 * (q & 1)) * TABP;
      tmp[k] = reinterpret_cast<const float4*>(src)[w];
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) reinterpret_cast<float4*>(smem + C::TAB)[threadIdx.x + k * NT] = tmp[k];
    if (threadIdx.x < 64) {
      reinterpret_cast<float4*>(smem + C::LNP)[threadIdx.x] = reinterpret_cast<const float4*>(a.lnk)[threadIdx.x];
      reinterpret_cast<float4*>(smem + C::LNP + 256)[threadIdx.x] = reinterpret_cast<const float4*>(a.lnv)[threadIdx.x];
    }
  }
  if (threadIdx.x == 0) sb[0] = atomicAdd(counter, 1);
  __syncthreads();

  for (int it = 0;; ++it) {
    int lane = lane0;
    asm volatile("" : "+v"(lane) :: "memory");
    const int mm = lane & 15, cg = lane >> 4;
    const int blk = __builtin_amdgcn_readfirstlane(sb[it & 1]);
    if (blk >= n_blocks) break;
    const int ne_b = blk / nb_kind, rb = blk % nb_kind;
    const int node = wg_protein ? rb * NW + wave : a.NP + rb * NW + wave;
    bool active = node < (wg_protein ? a.NP : N);
    if (RAG && active) active = wg_protein ? node < (a.np_real ? a.np_real[ne_b] : a.NP) : node - a.NP < a.nl_real[ne_b];
    const long seg = (long)ne_b * N + node;
    const float* xb = a.x + (long)ne_b * N * 3;
    const long src_base = (long)ne_b * N;

    // neighbours, edge weights, distances (two dependent round trips: first), the query, the rows of the first k-pass tile
    int jm[MAXT], jT[MAXT][4];
    float dm[MAXT], ewm[MAXT][4];
    float2 q2 = make_float2(0.f, 0.f);
    float Rc[32], Pf[32];
    if (active) {
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const int m = 16 * t + mm;
        jm[t] = a.nbr[seg * a.K + (m < M ? m : M - 1)];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mr = 16 * t + 4 * cg + r;
          ewm[t][r] = a.ew[seg * a.K + (mr < M ? mr : 0)];
          jT[t][r] = a.nbr[seg * a.K + (mr < M ? mr : M - 1)];
        }
      }
      q2 = *reinterpret_cast<const float2*>(a.q + seg * 128 + 2 * lane);
      const float cx = xb[3 * node], cy = xb[3 * node + 1], cz = xb[3 * node + 2];
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        const float rx = cx - xb[3 * jm[t]], ry = cy - xb[3 * jm[t] + 1], rz = cz - xb[3 * jm[t] + 2];
        dm[t] = sqrtf(rx * rx + ry * ry + rz * rz);
      }
      load_row(Rc, a.kd + seg * a.ld_kd, cg);
      load_row(Pf, a.ks + (src_base + jm[0]) * a.ld_ks, cg);
      *reinterpret_cast<float2*>(QS + wave * 128 + 2 * lane) = q2;
    }
    // this wave's 16 rows of W2k (heads 2 wave, 2 wave + 1), channels 2 lane, 2 lane + 1: 8 KB per wave and block from L2 (held in
    // registers for the whole workgroup they cost the k / v passes 32 registers: 442 spilled)
    float Wk[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float2 w = *reinterpret_cast<const float2*>(a.W2k + (16 * wave + r) * 128 + 2 * lane);
#if DD_LN_FOLD
      Wk[2 * r] = w.x; Wk[2 * r + 1] = w.y;
#else
      Wk[2 * r] = w.x * 0.35355339059327373f; Wk[2 * r + 1] = w.y * 0.35355339059327373f;
#endif
    }
    __syncthreads();                                     // barrier 1: queries published; the previous block's exchange reads are done
    if (threadIdx.x == 0) sb[(it + 1) & 1] = atomicAdd(counter, 1);

    // ---- cooperative fold: Q~[s][h][2 lane + {0, 1}] for h = 2 wave, 2 wave + 1 and the 8 segments of the block
#pragma unroll 2
    for (int s = 0; s < 8; ++s) {
      float qv[16];
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4) {
        const float4 v = *reinterpret_cast<const float4*>(QS + s * 128 + 16 * wave + 4 * k4);
        qv[4 * k4] = v.x; qv[4 * k4 + 1] = v.y; qv[4 * k4 + 2] = v.z; qv[4 * k4 + 3] = v.w;
      }
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        // (the same chain as attn2_body's fold: fma(q_d, w_d, acc) from acc = 0, d ascending)
        float a0 = fmaf(qv[8 * hh], Wk[16 * hh], 0.f), a1 = fmaf(qv[8 * hh], Wk[16 * hh + 1], 0.f);
#pragma unroll
        for (int d = 1; d < 8; ++d) {
          a0 = fmaf(qv[8 * hh + d], Wk[16 * hh + 2 * d], a0);
          a1 = fmaf(qv[8 * hh + d], Wk[16 * hh + 2 * d + 1], a1);
        }
        *reinterpret_cast<float2*>(XB + s * XSTR + (2 * wave + hh) * WPITCH + 2 * lane) = make_float2(a0, a1);
      }
    }
    __syncthreads();                                     // barrier 2
    float Qb[32];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const float4 v = *reinterpret_cast<const float4*>(XB + wave * XSTR + mm * WPITCH + 16 * nt + 4 * cg);
      Qb[4 * nt] = v.x; Qb[4 * nt + 1] = v.y; Qb[4 * nt + 2] = v.z; Qb[4 * nt + 3] = v.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();                                     // barrier 3: the exchange buffer is free for the epilogue

    // first-Linear table part (type (x) Gaussian) of tile t on the matrix cores, either orientation (see attn2_body::table_part)
    float Fg[MAXT][5];
    auto table_part = [&](int t, int pass, f32x4 (&acc)[8], auto tr) {
      constexpr bool TR = decltype(tr)::value;
      if (pass == 0) {
#pragma unroll
        for (int s = 0; s < 5; ++s) Fg[t][s] = gauss_feat(dm[t], 4 * s + cg);
      }
      const bool hi = jm[t] < a.NP;
      const float* tab = smem + C::TAB + pass * 2 * TABP + cg * 128 + mm * 4;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const bool want = half ? hi : !hi;
        if (__builtin_amdgcn_ballot_w64(want) != 0ull) {
#pragma unroll
          for (int s = 0; s < 5; ++s) mfma_table_step<TR>(acc, tab + half * TABP + s * 512, want ? Fg[t][s] : 0.0f);
        }
      }
      const float* row20 = smem + C::TAB + pass * 2 * TABP + 20 * 128;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool hi_r = TR ? (jT[t][r] < a.NP) : hi;
        const float* p = row20 + (hi_r ? TABP : 0) + (TR ? mm : 4 * cg + r) * 4;
        const float4 c0 = *reinterpret_cast<const float4*>(p);
        const float4 c1 = *reinterpret_cast<const float4*>(p + 64);
        acc[0][r] += c0.x; acc[1][r] += c0.y; acc[2][r] += c0.z; acc[3][r] += c0.w;
        acc[4][r] += c1.x; acc[5][r] += c1.y; acc[6][r] += c1.z; acc[7][r] += c1.w;
      }
    };

    f32x4 S[MAXT];
    float ssum = 0.f;
    f32x4 Z[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) Z[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
      // ---- k pass
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t < T) {
          float P[32];
#pragma unroll
          for (int k = 0; k < 32; ++k) P[k] = Rc[k] + Pf[k];
          if (t + 1 < MAXT) load_row(Pf, a.ks + (src_base + jm[t + 1]) * a.ld_ks, cg);
          f32x4 acc[8];
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{P[4 * nt], P[4 * nt + 1], P[4 * nt + 2], P[4 * nt + 3]};
          table_part(t, 0, acc, std::false_type{});
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) { P[4 * nt] = acc[nt][0]; P[4 * nt + 1] = acc[nt][1]; P[4 * nt + 2] = acc[nt][2]; P[4 * nt + 3] = acc[nt][3]; }
          ln_relu32(P, smem + C::LNP, cg);
          S[t] = mfma_rows(P, Qb);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * t + 4 * cg + r >= M) S[t][r] = -INFINITY;
        } else {
          S[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
      }
      // ---- v-pass rows of tile 0 (member-major layout), then the softmax
      float Tc[8], Tr[32];
      auto fetch_T = [&](int t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* rs = a.vs + (src_base + jT[t][r]) * a.ld_vs + mm;
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) Tr[4 * nt + r] = rs[16 * nt];
        }
      };
      {
        const float* rc = a.vd + seg * a.ld_vd + mm;
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) Tc[nt] = rc[16 * nt];
      }
      fetch_T(0);
      float mx = -INFINITY;
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, S[t][r]);
      mx = quad_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = (16 * t + 4 * cg + r < M) ? expf(S[t][r] - mx) : 0.f;
          S[t][r] = e;
          sum += e;
        }
      sum = quad_sum(sum);
      const float rsum = 1.0f / sum;
#pragma unroll
      for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = 16 * t + 4 * cg + r;
#if defined(DD_EXACT_MATH) && DD_EXACT_MATH
          const float aw = (m < M) ? (S[t][r] / sum) * ewm[t][r] : 0.f;
#else
          const float aw = (m < M) ? (S[t][r] * rsum) * ewm[t][r] : 0.f;
#endif
          S[t][r] = aw;
          ssum += aw;
        }
      ssum = quad_sum(ssum);
      // ---- v pass + aggregation
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t < T) {
          float Tz[32];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt) Tz[4 * nt + r] = Tc[nt] + Tr[4 * nt + r];
          if (t + 1 < MAXT) fetch_T(t + 1);
          f32x4 acc[8];
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{Tz[4 * nt], Tz[4 * nt + 1], Tz[4 * nt + 2], Tz[4 * nt + 3]};
          table_part(t, 1, acc, std::true_type{});
#pragma unroll
          for (int nt = 0; nt < 8; ++nt) { Tz[4 * nt] = acc[nt][0]; Tz[4 * nt + 1] = acc[nt][1]; Tz[4 * nt + 2] = acc[nt][2]; Tz[4 * nt + 3] = acc[nt][3]; }
          ln_relu_T(Tz, smem + C::LNP + 256, mm);
#pragma unroll
          for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nt = 0; nt < 8; ++nt)
              Z[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Tz[4 * nt + ks], S[t][ks], Z[nt], 0, 0, 0);
        }
      }
    }

    // ---- cooperative epilogue (as the NE blocks' NECO form)
    float4 Bv[8];
    {
      const float* bv = a.W2v + (16 * wave + mm) * 128 + 4 * cg;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) Bv[nt] = *reinterpret_cast<const float4*>(bv + 16 * nt);
    }
    const int eh = mm >> 3;
    const float bias2 = a.b2v[16 * wave + mm];
    if (active) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt)
        *reinterpret_cast<float4*>(XB + wave * XSTR + mm * WPITCH + 16 * nt + 4 * cg) = make_float4(Z[nt][0], Z[nt][1], Z[nt][2], Z[nt][3]);
      if (cg == 0) SS[wave * 16 + mm] = ssum;
    }
    __syncthreads();                                     // barrier 4
    {
      const float* ar = XB + (mm >> 1) * XSTR + (2 * wave + (mm & 1)) * WPITCH + 4 * cg;
      f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int nt = 0; nt < 8; nt += 2) {
        const float4 a0 = *reinterpret_cast<const float4*>(ar + 16 * nt), a1 = *reinterpret_cast<const float4*>(ar + 16 * nt + 16);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, Bv[nt].x, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, Bv[nt + 1].x, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, Bv[nt].y, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, Bv[nt + 1].y, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, Bv[nt].z, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, Bv[nt + 1].z, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, Bv[nt].w, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, Bv[nt + 1].w, d1, 0, 0, 0);
      }
      const f32x4 d = d0 + d1;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int sI = 2 * cg + u;
        const int nd = (wg_protein ? rb * NW : a.NP + rb * NW) + sI;
        bool ok = nd < (wg_protein ? a.NP : N);
        if (RAG && ok) ok = wg_protein ? nd < (a.np_real ? a.np_real[ne_b] : a.NP) : nd - a.NP < a.nl_real[ne_b];
        const float val = fmaf(bias2, SS[sI * 16 + 2 * wave + eh], eh ? d[2 * u + 1] : d[2 * u]);
        if (ok) a.out[((long)ne_b * N + nd) * 128 + 16 * wave + mm] = val;
      }
    }
  }
}

// stand-alone launch of the cooperative bond-layer workgroups (one launch per sub-layer: dd_debug_set_fusion(0), phase stamps)
template <int MAXT, bool RAG>
__global__ __launch_bounds__(512) void k_attn2_bl_coop(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[CoopLds::TOTAL];
  bl_coop_body<MAXT, RAG, true>(a, smem);
}

template <int MODE, int MAXT, int NW, bool RAG = false>
__global__ __launch_bounds__(NW * 64) void k_attn2(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[Lds<MODE>::TOTAL + ((MODE == M_PE || MODE == M_PB) ? NW * 256 : 0) +
                                                     ((MODE == M_NE || MODE == M_NB) ? NW * WPITCH + 16 + 128 : 0)];
  attn2_body<MODE, MAXT, NW, false, RAG>(a, blockIdx.x, smem);
}

constexpr int imax(int a, int b) { return a > b ? a : b; }

// The three sub-layers that read the *old* h / h_bond (NE, NB, BL) are independent: one launch.  The NE workgroups
// (long, one batch of NW segments each) come first, then the few NB ones; the BL workgroups are persistent and pull
// single segments from a global counter, so they fill whatever the coarse NE schedule leaves idle and all finish
// within one segment of each other.
template <int MAXT, int NW, bool RAG = false>
__global__ __launch_bounds__(NW * 64) void k_attn2_node(const AttnArgs ne, const AttnArgs nb, const AttnArgs bl, int n_ne, int n_nb,
                                                        int persist, int n_bl_first, const int32_t* wflags, int widx, int wn, int nep) {
  constexpr int SZ = imax(imax(imax(Lds<M_NE>::TOTAL, Lds<M_NB>::TOTAL), Lds<M_BL>::TOTAL) + 12, imax(DD_COOP && NW == 8 && MAXT == 2 ? CoopLds::TOTAL : 0, DD_NE_PERSIST && NW == 8 ? NepLds::TOTAL : 0));
  __shared__ __attribute__((aligned(16))) float smem[SZ];
  int blk = blockIdx.x;
  // this layer's projection / query rows come from the previous layer's tail queue on the other stream (no graph edge)
#if !DD_NODE_TRACE
  dd_wait_flags(wflags, widx, wn, -1, 0, DD_FLAG_ERR, 500);
#endif
  // Each body reads ITS argument block through the kernarg pointer with an offset the compiler cannot see through: with
  // the three by-value structs used directly, their scalar loads are hoisted in front of the branch below and stay live
  // across all bodies (104 SGPRs + 80 spilled to VGPR lanes, which in turn pushed the 256-register node body into
  // scratch); read in place, a body holds only its own pointers.
  typedef const __attribute__((address_space(4))) AttnArgs KArgs;
  auto args = [](unsigned k) -> KArgs& {
    unsigned off = k * (unsigned)sizeof(AttnArgs);       // kernel parameters: ne at 0, nb and bl behind it
    asm volatile("" : "+s"(off));
    return *reinterpret_cast<KArgs*>((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
  };
  static_assert(sizeof(AttnArgs) % 8 == 0, "kernel parameter layout");
  (void)ne; (void)nb; (void)bl;
  // n_bl_first > 0: the persistent bond-layer workgroups come first in dispatch order and keep their CUs for the whole
  // launch, the node blocks cycle through the remaining CUs -- both parts then end together (see launch_node_nw)
#if DD_NODE_TRACE   // (measurement variant: start / end shader clock, kind and hardware id of every workgroup -> tools/node_trace.py)
  long long* const trace = (wn == -12345) ? reinterpret_cast<long long*>(const_cast<int32_t*>(wflags)) + (long)blockIdx.x * 16 : nullptr;   // (a 128-byte line per workgroup: lines shared across XCDs lose updates)
  // (system-scope stores: a dirty line left in one XCD's L2 by an earlier launch would otherwise be written back over a later launch's entry)
#define DD_TRACE_ST(i, v) __hip_atomic_store(trace + (i), (long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
  if (trace && threadIdx.x == 0) {
    DD_TRACE_ST(0, __builtin_amdgcn_s_memrealtime());   // (the 100 MHz real-time counter: one time base for all XCDs)
    DD_TRACE_ST(3, (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4 /* HW_REG_HW_ID */) |
                       ((long long)(unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20 /* HW_REG_XCC_ID */) << 32));
  }
#define DD_TRACE_END(kind) do { if (trace && threadIdx.x == 0) { DD_TRACE_ST(1, __builtin_amdgcn_s_memrealtime()); DD_TRACE_ST(2, kind); } } while (0)
#else
#define DD_TRACE_END(kind) do { } while (0)
#endif
  if (n_bl_first > 0) {
    if (blk < n_bl_first) {
      if constexpr (DD_COOP && NW == 8 && MAXT == 2) bl_coop_body<MAXT, RAG, false>(args(2), smem);
      else attn2_body<M_BL, MAXT, NW, true, RAG, false, false>(args(2), blk, smem);
      DD_TRACE_END(2);
      return;
    }
    blk -= n_bl_first;
    if constexpr (DD_NE_PERSIST && NW == 8) {
      if (nep != 0) {
        // persistent node_layer_with_edge workgroups (nep = protein-kind count << 16 | ligand-kind count): the NB blocks come FIRST in
        // dispatch order, the persistent workgroups their CUs cannot hold yet start when they end and pull what is left
        const int np_p = nep >> 16, np_l = nep & 0xffff;
        if (blk < n_nb) { attn2_body<M_NB, MAXT, NW, false, RAG, false, false>(args(1), blk, smem); DD_TRACE_END(1); return; }
        blk -= n_nb;
        KArgs& na = args(0);
        if (blk < np_p) ne_persist_body<RAG>(na, smem, true, na.work_counter);
        else if (blk < np_p + np_l) ne_persist_body<RAG>(na, smem, false, na.work_counter + 1);
        DD_TRACE_END(0);
        return;
      }
    }
    if (blk < n_ne) { attn2_body<M_NE, 2, NW, false, RAG, false, false>(args(0), blk, smem); DD_TRACE_END(0); }
    else { attn2_body<M_NB, MAXT, NW, false, RAG, false, false>(args(1), blk - n_ne, smem); DD_TRACE_END(1); }
    return;
  }
  if (blk < n_ne) attn2_body<M_NE, 2, NW, false, RAG, false, false>(args(0), blk, smem);
  else if (blk < n_ne + n_nb) attn2_body<M_NB, MAXT, NW, false, RAG, false, false>(args(1), blk - n_ne, smem);
  else if (persist) {
    if constexpr (DD_COOP && NW == 8 && MAXT == 2) bl_coop_body<MAXT, RAG, false>(args(2), smem);
    else attn2_body<M_BL, MAXT, NW, true, RAG, false, false>(args(2), blk - n_ne - n_nb, smem);
  }
  else attn2_body<M_BL, MAXT, NW, false, RAG, false, false>(args(2), blk - n_ne - n_nb, smem);
}
// Same for the two coordinate sub-layers (both write their own delta buffer; x is updated afterwards).
// (NW segments = 2 NW waves per workgroup: attn2_body's PAIR)
template <int MAXT, int NW, bool RAG = false>
__global__ __launch_bounds__(NW * 128) void k_attn2_pos(const AttnArgs pe, const AttnArgs pb, int n_pe) {
  // + per-segment scratch of the in-kernel query MLP + the attention weights handed from the k wave to the v wave
  constexpr int SZ = imax(Lds<M_PE>::TOTAL, Lds<M_PB>::TOTAL) + NW * 256 + NW * MAXT * 256;
  __shared__ __attribute__((aligned(16))) float smem[SZ];
  const int blk = blockIdx.x;
  // projections of the new h / h_bond from the layer-tail queue on the other stream (no graph edge): poll, one acquire
  dd_wait_flags(pe.wait_flags, pe.wait_idx, pe.wait_n, -1, 0, DD_FLAG_ERR, 400);
  if (blk < n_pe) attn2_body<M_PE, 2, NW, false, RAG, true, false>(pe, blk, smem);
  else attn2_body<M_PB, MAXT, NW, false, RAG, true, false>(pb, blk - n_pe, smem);
  // x update (x += (dx_edge + dx_bond) on the ligand rows, uni_transformer_edge.py:285) by the workgroup that finishes
  // last: ~120 workgroups, so the ticket costs nothing and a launch on the critical chain is saved
  if (pe.work_counter == nullptr) return;
  __shared__ int s_last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    s_last = atomicAdd(pe.work_counter, 1) == (int)gridDim.x - 1 ? 1 : 0;
  }
  __syncthreads();
  if (s_last) {
    __threadfence();
    const int n = pe.B * pe.NL * 3, N = pe.NP + pe.NL;
    const volatile float* dxe = pe.out;
    const volatile float* dxb = pb.out;
    for (int idx = threadIdx.x; idx < n; idx += NW * 128) {
      const int b = idx / (pe.NL * 3), r = idx % (pe.NL * 3);
      const long xi = ((long)b * N + pe.NP) * 3 + r;
      pe.x_next[xi] = pe.x[xi] + dxe[idx] + dxb[idx];
    }
    if (threadIdx.x == 0) *pe.work_counter = 0;
  }
}

// QUAD variant (four waves per segment, NW = 2 segments per workgroup: attn2_body's PAIR = 2)
template <int MAXT, int NW, bool RAG = false>
__global__ __launch_bounds__(NW * 256) void k_attn2_pos_q(const AttnArgs pe, const AttnArgs pb, int n_pe) {
  // + two query-MLP scratch rows per segment + the weights' hand-over + the max / sum / update exchange
  constexpr int SZ = imax(Lds<M_PE>::TOTAL, Lds<M_PB>::TOTAL) + 2 * NW * 256 + NW * MAXT * 256 + NW * 64 + NW * 8 + 16;
  __shared__ __attribute__((aligned(16))) float smem[SZ];
  const int blk = blockIdx.x;
  if (blk < n_pe) attn2_body<M_PE, 2, NW, false, RAG, 2, false>(pe, blk, smem);
  else attn2_body<M_PB, MAXT, NW, false, RAG, 2, false>(pb, blk - n_pe, smem);
}

#if defined(DD_DEBUG_OPTIONS) && DD_DEBUG_OPTIONS
// ---- coordinate launch with the projections of the new h inside (k_attn2_pos_g) -------------------------------------------
// MEASUREMENT BUILD ONLY (dd_debug_set_option(30, 1)): bit-identical, but the launch takes 40 us where the two launches it
// replaces take 11 + 23 -- an in-launch hand-off (write-through stores, counter, poll, acquire, cold reads) costs more than the
// kernel boundary it removes, whatever the poll rate or the store type (EXPERIMENTS.md R5-2).
// The {P2, PL2} projection launch sits between lin_node and the coordinate attention on the step's critical chain
// (11 us + the launch boundary, per layer).  Its 64 x 64 tiles now run in the LEADING workgroups of the coordinate launch (two
// tiles per 512-thread workgroup, the GEMM launch's own tile code: bit-identical), while the attention workgroups -- dispatched
// behind them, one per CU either way -- stage their 127 KB of weight images; then one lane per attention workgroup polls the tile
// counter.  Producers never wait and are dispatched first, so the hand-off cannot deadlock.  Visibility (guide section 6,
// Guideline 16, form R1): tiles store write-through (sc1) and bump the counter after their stores have drained; the consumer
// does ONE agent-scope acquire after its poll, the workgroup barrier releases the other waves, plain loads follow.
// Jobs listed after the first `n_lead` ones (the heads' first Linear in the last layer: nobody in this launch reads them) run
// in TRAILING workgroups, behind the attention workgroups in dispatch order.
struct PosGemm { GemmArgs job[4]; int end[4]; int nbx[4]; int njobs; int mode; };   // mode: bit 0 plain stores + release fence (not sc1)

// tiles `first`, `first + 1` (< limit) of the job list: one per group of 256 threads, each in its own 33 KB LDS image
__device__ __forceinline__ void pos_gemm_tiles(const PosGemm& pg, int first, int limit, float* smem) {
  const int half = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
  int tile = first + half;
  const bool valid = tile < limit;
  if (!valid) tile = first;                              // (its rows are moved out of range below: barriers only)
  int j = 0, base = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
    if (j == i && tile >= pg.end[i] && i + 1 < pg.njobs) { base = pg.end[i]; j = i + 1; }
  j = __builtin_amdgcn_readfirstlane(j);
  const GemmArgs a = pg.job[j];                          // wave-uniform index into the kernel arguments: scalar loads
  const int lb = tile - base, nbx = pg.nbx[j];
  const int bx = valid ? lb % nbx : (1 << 20), by = lb / nbx;
  if (pg.mode & 1) gemm_tile_ksplit<false, false>(a, bx, by, smem + half * (2 * GT * GPH));
  else gemm_tile_ksplit<false, true>(a, bx, by, smem + half * (2 * GT * GPH));
}

template <int MAXT, int NW, bool RAG = false>
__global__ __launch_bounds__(NW * 128) void k_attn2_pos_g(const AttnArgs pe, const AttnArgs pb, int n_pe, const PosGemm pg, int n_g0,
                                                          int n_tiles0, int n_att, int n_tiles_all, int32_t* counter) {
  static_assert(NW == 4, "two 256-thread tile groups per workgroup");
  constexpr int SZ = imax(Lds<M_PE>::TOTAL, Lds<M_PB>::TOTAL) + NW * 256 + NW * MAXT * 256;
  static_assert(SZ >= 2 * 2 * GT * GPH, "two K-half tile images fit the attention workgroup's LDS");
  __shared__ __attribute__((aligned(16))) float smem[SZ];
  int blk = blockIdx.x;
  if (blk < n_g0 || blk >= n_g0 + n_att) {
    const bool lead = blk < n_g0;
    const int first = lead ? 2 * blk : n_tiles0 + 2 * (blk - n_g0 - n_att);
    pos_gemm_tiles(pg, first, lead ? n_tiles0 : n_tiles_all, smem);
    if (lead) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's write-through stores have left
      __syncthreads();                                              // ... and every other wave's of the workgroup
      if (threadIdx.x == 0) {
        const int n_done = n_tiles0 - first < 2 ? n_tiles0 - first : 2;
        if (pg.mode & 1) __hip_atomic_fetch_add(counter, n_done, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // (L2 write-back first)
        else __hip_atomic_fetch_add(counter, n_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    return;
  }
  blk -= n_g0;
  if (blk < n_pe) attn2_body<M_PE, 2, NW, false, RAG, true, false>(pe, blk, smem);
  else attn2_body<M_PB, MAXT, NW, false, RAG, true, false>(pb, blk - n_pe, smem);
}

#endif

#ifdef DD_ASM_ONLY      // (hipcc -S -DDD_ASM_ONLY: only the shipped small-ligand kernels, for instruction censuses -- 20 s instead of 2 min)
template __global__ void k_attn2_node<2, 8, false>(const AttnArgs, const AttnArgs, const AttnArgs, int, int, int, int, const int32_t*, int, int, int);
template __global__ void k_attn2_pos<2, 4, false>(const AttnArgs, const AttnArgs, int);
template __global__ void k_attn2_bl_coop<2, false>(const AttnArgs);
template __global__ void k_attn2_pos_q<2, 2, false>(const AttnArgs, const AttnArgs, int);
}  // namespace v2
}  // namespace dd
#else
template <int MODE, int MAXT, int NW>
static int launch_mode(const AttnArgs& a, int nseg, hipStream_t st) {
  if (nseg <= 0) return DD_OK;
  if (a.nl_real != nullptr) hipLaunchKernelGGL((k_attn2<MODE, MAXT, NW, true>), dim3((nseg + NW - 1) / NW), dim3(NW * 64), 0, st, a);
  else hipLaunchKernelGGL((k_attn2<MODE, MAXT, NW>), dim3((nseg + NW - 1) / NW), dim3(NW * 64), 0, st, a);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

}  // namespace v2

}  // namespace dd
// Which form of the packed attention MLPs these kernels expect: 0 canonical (reference values), 1 packing.kernel_form_layer
extern "C" int dd_weights_form(void) { return DD_LN_FOLD ? 1 : 0; }
namespace dd {

int g_attn_waves = 8;        // waves (= segments) per workgroup of the fused node launch
int g_attn_persist = 1;      // bond_layer workgroups of the fused launch are persistent (global batch counter)
#ifndef DD_BL_TAIL
#define DD_BL_TAIL 1
#endif
int g_bl_tail = DD_BL_TAIL;  // the last (partial) round of bond-layer trips spread evenly over the persistent workgroups
int g_bl_first = 1;          // bond-layer workgroups first in the node launch: 0 off, 1 measured split per shape, n>1 that many
int g_ne_persist = [] { const char* e = getenv("DD_NE_PERSIST"); return e ? (e[0] != '0') : 1; }();   // (A/B: DD_NE_PERSIST=0 keeps the block form)
int g_node_split_trial = -1; // >= 0 while autotune_node_split is timing a candidate (0 = node blocks first)
long long* g_node_trace = nullptr;   // DD_NODE_TRACE builds: [workgroups][16] clocks of the next fused node launches (dd_debug_set_clock_buffer(buf, 200))
namespace {
struct NodeSplit { int B, NP, NL, K, n_bl; };
std::vector<NodeSplit> g_node_splits;
}
int node_split_lookup(int B, int NP, int NL, int K) {          // -1: not measured yet, 0: node blocks first, n: n bond-layer WGs first
  for (const NodeSplit& e : g_node_splits)
    if (e.B == B && e.NP == NP && e.NL == NL && e.K == K) return e.n_bl;
  return -1;
}
void node_split_store(int B, int NP, int NL, int K, int n_bl) {
  for (NodeSplit& e : g_node_splits)
    if (e.B == B && e.NP == NP && e.NL == NL && e.K == K) { e.n_bl = n_bl; return; }
  g_node_splits.push_back(NodeSplit{B, NP, NL, K, n_bl});
}
bool node_split_applies(int B, int NL, int n_cu) {             // the bond layer has at least one trip per CU
  return g_attn_persist && g_bl_first == 1 && (B * NL * (NL - 1) + 7) / 8 >= n_cu;
}

int launch_attn2(int mode, const AttnArgs& a, hipStream_t st) {
  using namespace v2;
  const int N = a.NP + a.NL;
  const bool small = a.NL <= 33;                     // NL-1 <= 32 members -> 2 tiles
  const bool big = a.NL > 65;                        // up to 128 ligand atoms: 8 tiles (register spills accepted: rare sizes)
  if (a.NL > 129) return DD_ERR_UNSUPPORTED_SHAPE;
  switch (mode) {
    case M_NE: return launch_mode<M_NE, 2, 8>(a, a.B * v2::ne_blocks_per_sample(a.NP, a.NL, 8) * 8, st);   // (blocks * NW)
    case M_PE: return launch_mode<M_PE, 2, 8>(a, a.B * a.NL, st);
    case M_NB: return small ? launch_mode<M_NB, 2, 8>(a, a.B * a.NL, st)
                            : (big ? launch_mode<M_NB, 8, 8>(a, a.B * a.NL, st) : launch_mode<M_NB, 4, 8>(a, a.B * a.NL, st));
    case M_PB: return small ? launch_mode<M_PB, 2, 8>(a, a.B * a.NL, st)
                            : (big ? launch_mode<M_PB, 8, 8>(a, a.B * a.NL, st) : launch_mode<M_PB, 4, 8>(a, a.B * a.NL, st));
    case M_BL:
#if DD_COOP
      if (a.work_counter != nullptr && small) {        // the fused launch's persistent cooperative workgroups, on their own
                                                       // (2-tile body only: the longer bodies spill with the carried prefetch state)
        const int n_trips = (a.B * a.NL * (a.NL - 1) + 7) / 8;
        if (n_trips <= 0) return DD_OK;
        const int n_wg = n_trips < 256 ? n_trips : 256;
        AttnArgs c = a;
        c.trip_full = 1 << 27; c.trip_q = 0;
#define DD_LAUNCH_COOP(MAXT)                                                                                                  \
        do {                                                                                                                  \
          if (a.nl_real != nullptr) hipLaunchKernelGGL((k_attn2_bl_coop<MAXT, true>), dim3(n_wg), dim3(512), 0, st, c);       \
          else hipLaunchKernelGGL((k_attn2_bl_coop<MAXT, false>), dim3(n_wg), dim3(512), 0, st, c);                           \
        } while (0)
        DD_LAUNCH_COOP(2);
#undef DD_LAUNCH_COOP
        DD_CHECK_LAUNCH();
        return DD_OK;
      }
#endif
      return small ? launch_mode<M_BL, 2, 8>(a, a.B * a.NL * (a.NL - 1), st)
                   : (big ? launch_mode<M_BL, 8, 8>(a, a.B * a.NL * (a.NL - 1), st)
                          : launch_mode<M_BL, 4, 8>(a, a.B * a.NL * (a.NL - 1), st));
  }
  return DD_ERR_BAD_ARG;
}

// Fused launches (ligands with <= 33 atoms).  Returns DD_ERR_UNSUPPORTED_SHAPE for larger ligands; the caller then
// falls back to one launch per sub-layer.
template <int NW, int MAXT>
static int launch_node_nw(const AttnArgs& ne, const AttnArgs& nb, const AttnArgs& bl, hipStream_t st) {
  using namespace v2;
  const int N = ne.NP + ne.NL;
  const int n_ne = ne.B * ne_blocks_per_sample(ne.NP, ne.NL, NW), n_nb = (ne.B * ne.NL + NW - 1) / NW;
  int n_bl = (ne.B * ne.NL * (ne.NL - 1) + NW - 1) / NW;
  const int persist = (g_attn_persist && bl.work_counter != nullptr) ? 1 : 0;
  const int32_t* const wf = g_node_trace ? reinterpret_cast<const int32_t*>(g_node_trace) : ne.wait_flags;   // (DD_NODE_TRACE builds)
  const int wfn = g_node_trace ? -12345 : ne.wait_n;
  AttnArgs blt = bl;                                   // + the trip plan of the persistent workgroups (set_trips below)
  blt.trip_full = 1 << 27; blt.trip_q = 0;
  auto set_trips = [&](int n_wg) {
    if (!g_bl_tail || ne.nl_real != nullptr || n_wg <= 0) return;      // (padded batches: the segment count lives on the device)
    const int nseg = ne.B * ne.NL * (ne.NL - 1);
    const int full = nseg / (NW * n_wg) * n_wg, rem = nseg - full * NW;
    blt.trip_full = full;
    blt.trip_q = (rem + n_wg - 1) / n_wg;              // 0 when the rounds come out even: trip `full` then finds no work
  };
  if (persist) {
    static int n_cu = 0;
    if (n_cu == 0) {
      int dev = 0;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return DD_ERR_HIP;
      n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (n_bl > n_cu) n_bl = n_cu;                      // one workgroup per CU (LDS-limited)
    if (g_bl_first > 0 && n_bl == n_cu) {
      // The bond-layer workgroups first, on a fixed share of the CUs, the node blocks cycling through the rest: both
      // parts run in whole "rounds" (a node block ~23 us, a bond-layer trip ~17 us at NL = 30), so the best share is a
      // step function of the shape -- it is measured once per shape (dd_api.hip::autotune_node_split), not modelled.
      int want = g_bl_first > 1 ? g_bl_first : (g_node_split_trial >= 0 ? g_node_split_trial : node_split_lookup(ne.B, ne.NP, ne.NL, ne.K));
#if DD_NODE_TRACE
      { static const int env_bl = [] { const char* e = getenv("DD_BL_FIRST"); return e ? atoi(e) : 0; }(); if (env_bl > 0 && g_node_split_trial < 0) want = env_bl; }
#endif
      if (want > 0) {
        n_bl = want < 16 ? 16 : (want > n_cu - 16 ? n_cu - 16 : want);
        set_trips(n_bl);
        // persistent node_layer_with_edge workgroups on the other CUs, split by centre kind in proportion to the blocks of each kind
        int nep = 0, n_grid = n_ne + n_nb + n_bl;
        if (DD_NE_PERSIST && NW == 8 && g_ne_persist && ne.work_counter != nullptr && ne.lin_W == nullptr) {
          const int nbp = ne.B * ((ne.NP + NW - 1) / NW), nbl = ne.B * ((ne.NL + NW - 1) / NW), n_rest = n_cu - n_bl;
          int np_l = nbl > 0 ? (int)((long)n_rest * nbl / (nbp + nbl > 0 ? nbp + nbl : 1) + 0.5) : 0;
          if (nbl > 0 && np_l < 1) np_l = 1;
          if (np_l > nbl) np_l = nbl;
          int np_p = n_rest - np_l;
          if (np_p > nbp) np_p = nbp;
          if (np_p > 0 || np_l > 0) { nep = (np_p << 16) | np_l; n_grid = n_bl + n_nb + np_p + np_l; }
        }
        if (ne.nl_real != nullptr)
          hipLaunchKernelGGL((k_attn2_node<MAXT, NW, true>), dim3(n_grid), dim3(NW * 64), 0, st, ne, nb, blt, n_ne, n_nb,
                             persist, n_bl, wf, ne.wait_idx, wfn, nep);
        else
          hipLaunchKernelGGL((k_attn2_node<MAXT, NW>), dim3(n_grid), dim3(NW * 64), 0, st, ne, nb, blt, n_ne, n_nb, persist,
                             n_bl, wf, ne.wait_idx, wfn, nep);
        DD_CHECK_LAUNCH();
        return DD_OK;
      }
    }
  }
  if (persist) set_trips(n_bl);
  if (ne.nl_real != nullptr)
    hipLaunchKernelGGL((k_attn2_node<MAXT, NW, true>), dim3(n_ne + n_nb + n_bl), dim3(NW * 64), 0, st, ne, nb, blt, n_ne, n_nb, persist, 0, wf, ne.wait_idx, wfn, 0);
  else
    hipLaunchKernelGGL((k_attn2_node<MAXT, NW>), dim3(n_ne + n_nb + n_bl), dim3(NW * 64), 0, st, ne, nb, blt, n_ne, n_nb, persist, 0, wf, ne.wait_idx, wfn, 0);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_attn2_node(const AttnArgs& ne, const AttnArgs& nb, const AttnArgs& bl, hipStream_t st) {
  if (ne.NL > 129) return DD_ERR_UNSUPPORTED_SHAPE;
  if (ne.NL > 65) return launch_node_nw<8, 8>(ne, nb, bl, st);    // 65 .. 128 members per segment: 8 tiles (spills: rare sizes)
  if (ne.NL > 49) return launch_node_nw<8, 4>(ne, nb, bl, st);    // up to 64 members per segment: 4 tiles
  if (ne.NL > 33) return launch_node_nw<8, 3>(ne, nb, bl, st);    // up to 48 members: 3 tiles (fewer live registers than 4)
  return launch_node_nw<8, 2>(ne, nb, bl, st);             // (12- and 16-wave workgroups were tried: register spills)
}
template <int NW>
static int launch_pos_nw(const AttnArgs& pe, const AttnArgs& pb, hipStream_t st) {
  using namespace v2;
  const int n = (pe.B * pe.NL + NW - 1) / NW;
  if (pe.NL > 65) {                                    // 8 tiles: the weights' hand-over buffer fits the LDS for NW <= 4 only
    if constexpr (NW <= 4) {
      if (pe.nl_real != nullptr) hipLaunchKernelGGL((k_attn2_pos<8, NW, true>), dim3(2 * n), dim3(NW * 128), 0, st, pe, pb, n);
      else hipLaunchKernelGGL((k_attn2_pos<8, NW>), dim3(2 * n), dim3(NW * 128), 0, st, pe, pb, n);
      DD_CHECK_LAUNCH();
      return DD_OK;
    } else {
      return DD_ERR_UNSUPPORTED_SHAPE;
    }
  }
  if (pe.nl_real != nullptr) {
    if (pe.NL > 49) hipLaunchKernelGGL((k_attn2_pos<4, NW, true>), dim3(2 * n), dim3(NW * 128), 0, st, pe, pb, n);
    else if (pe.NL > 33) hipLaunchKernelGGL((k_attn2_pos<3, NW, true>), dim3(2 * n), dim3(NW * 128), 0, st, pe, pb, n);
    else hipLaunchKernelGGL((k_attn2_pos<2, NW, true>), dim3(2 * n), dim3(NW * 128), 0, st, pe, pb, n);
  } else if (pe.NL > 49) hipLaunchKernelGGL((k_attn2_pos<4, NW>), dim3(2 * n), dim3(NW * 128), 0, st, pe, pb, n);
  else if (pe.NL > 33) hipLaunchKernelGGL((k_attn2_pos<3, NW>), dim3(2 * n), dim3(NW * 128), 0, st, pe, pb, n);
  else hipLaunchKernelGGL((k_attn2_pos<2, NW>), dim3(2 * n), dim3(NW * 128), 0, st, pe, pb, n);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
// Coordinate launch with up to 4 projection jobs inside: jobs[0 .. n_lead) are what the attention workgroups wait for (their tiles
// run in the leading workgroups), the rest runs behind them.  `counter`: an int32 that is zero when the launch starts (one per
// layer: the forward's first launch zeroes the workspace counters).  DD_ERR_UNSUPPORTED_SHAPE: use the two launches instead.
int g_pos_g_mode = 0;        // measurement: bit 0 plain stores + release instead of write-through stores, bit 1 fast polls
int launch_attn2_pos_g(const AttnArgs& pe_in, const AttnArgs& pb_in, const GemmArgs* jobs, int njobs, int n_lead, int32_t* counter,
                       hipStream_t st) {
#if !(defined(DD_DEBUG_OPTIONS) && DD_DEBUG_OPTIONS)
  (void)pe_in; (void)pb_in; (void)jobs; (void)njobs; (void)n_lead; (void)counter; (void)st;
  return DD_ERR_UNSUPPORTED_SHAPE;                       // the default library does not compile the variant: the two launches
#else
  using namespace v2;
  if (njobs <= 0 || njobs > 4 || n_lead <= 0 || n_lead > njobs || counter == nullptr) return DD_ERR_BAD_ARG;
  if (pe_in.NL > 65 || pe_in.work_counter != nullptr) return DD_ERR_UNSUPPORTED_SHAPE;
  for (int i = 0; i < njobs; ++i)
    if (jobs[i].ln != nullptr || (jobs[i].ldy & 3) || (jobs[i].ncols & 3) || (reinterpret_cast<size_t>(jobs[i].Y) & 15))
      return DD_ERR_UNSUPPORTED_SHAPE;                   // (the write-through epilogue is the 16-byte path)
  PosGemm pg;
  pg.njobs = njobs;
  pg.mode = g_pos_g_mode & 1;
  int total = 0, tiles0 = 0;
  for (int i = 0; i < 4; ++i) {
    const GemmArgs& g = jobs[i < njobs ? i : 0];
    pg.job[i] = g;
    pg.nbx[i] = i < njobs ? (g.rows + GT - 1) / GT : 1;
    if (i < njobs) total += pg.nbx[i] * ((g.ncols + GT - 1) / GT);
    pg.end[i] = total;
    if (i + 1 == n_lead) tiles0 = total;
  }
  constexpr int NW = 4;
  const int n = (pe_in.B * pe_in.NL + NW - 1) / NW, n_att = 2 * n;
  const int n_g0 = (tiles0 + 1) / 2, n_g1 = (total - tiles0 + 1) / 2;
  AttnArgs pe = pe_in, pb = pb_in;
  pe.wait_flags = pb.wait_flags = counter; pe.wait_idx = pb.wait_idx = 0; pe.wait_n = pb.wait_n = (g_pos_g_mode & 2) ? -tiles0 : tiles0;
  const dim3 grid(n_g0 + n_att + n_g1), block(NW * 128);
  if (pe.nl_real != nullptr) {
    if (pe.NL > 49) hipLaunchKernelGGL((k_attn2_pos_g<4, NW, true>), grid, block, 0, st, pe, pb, n, pg, n_g0, tiles0, n_att, total, counter);
    else if (pe.NL > 33) hipLaunchKernelGGL((k_attn2_pos_g<3, NW, true>), grid, block, 0, st, pe, pb, n, pg, n_g0, tiles0, n_att, total, counter);
    else hipLaunchKernelGGL((k_attn2_pos_g<2, NW, true>), grid, block, 0, st, pe, pb, n, pg, n_g0, tiles0, n_att, total, counter);
  } else if (pe.NL > 49) hipLaunchKernelGGL((k_attn2_pos_g<4, NW>), grid, block, 0, st, pe, pb, n, pg, n_g0, tiles0, n_att, total, counter);
  else if (pe.NL > 33) hipLaunchKernelGGL((k_attn2_pos_g<3, NW>), grid, block, 0, st, pe, pb, n, pg, n_g0, tiles0, n_att, total, counter);
  else hipLaunchKernelGGL((k_attn2_pos_g<2, NW>), grid, block, 0, st, pe, pb, n, pg, n_g0, tiles0, n_att, total, counter);
  DD_CHECK_LAUNCH();
  return DD_OK;
#endif
}

int g_pos_waves = 4;         // waves per workgroup of the fused coordinate launch: 2, 4 or 8
#ifndef DD_POS_QUAD
#define DD_POS_QUAD 0       // measured SLOWER (EXPERIMENTS.md R6-5): 240 workgroups of 130 KB LDS leave no CU for the side stream's GEMMs
#endif
int g_pos_quad = DD_POS_QUAD;   // four waves per segment, two segments per workgroup (k_attn2_pos_q); no in-launch x update in this form
static int launch_pos_quad(const AttnArgs& pe, const AttnArgs& pb, hipStream_t st) {
  using namespace v2;
  constexpr int NW = 2;
  const int n = (pe.B * pe.NL + NW - 1) / NW;
  const dim3 grid(2 * n), block(NW * 256);
  if (pe.nl_real != nullptr) {
    if (pe.NL > 65) hipLaunchKernelGGL((k_attn2_pos_q<8, NW, true>), grid, block, 0, st, pe, pb, n);
    else if (pe.NL > 33) hipLaunchKernelGGL((k_attn2_pos_q<4, NW, true>), grid, block, 0, st, pe, pb, n);
    else hipLaunchKernelGGL((k_attn2_pos_q<2, NW, true>), grid, block, 0, st, pe, pb, n);
  } else if (pe.NL > 65) hipLaunchKernelGGL((k_attn2_pos_q<8, NW>), grid, block, 0, st, pe, pb, n);
  else if (pe.NL > 33) hipLaunchKernelGGL((k_attn2_pos_q<4, NW>), grid, block, 0, st, pe, pb, n);
  else hipLaunchKernelGGL((k_attn2_pos_q<2, NW>), grid, block, 0, st, pe, pb, n);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_attn2_pos(const AttnArgs& pe, const AttnArgs& pb, hipStream_t st) {
  if (pe.NL > 129) return DD_ERR_UNSUPPORTED_SHAPE;
  if (g_pos_quad && pe.work_counter == nullptr) return launch_pos_quad(pe, pb, st);
  if (pe.NL > 65 && g_pos_waves > 4) return launch_pos_nw<4>(pe, pb, st);
  if (g_pos_waves == 2) return launch_pos_nw<2>(pe, pb, st);
  if (g_pos_waves == 4) return launch_pos_nw<4>(pe, pb, st);
  return launch_pos_nw<8>(pe, pb, st);
}

}  // namespace dd
#endif  // DD_ASM_ONLY
