// Fused segment attention, tiled variant (v2): 16 members per tile, matrix cores for the
// per-head contractions.
//
// Same math and interface as dd_attention.hip (see there for the algebra); the work inside a
// wavefront is laid out for v_mfma_f32_16x16x4_f32 instead of one member at a time:
//
//   lane l = (mm = l & 15, cg = l >> 4)   holds member mm of the current 16-member tile and the 32
//   hidden channels  c(nt, r) = 16*nt + 4*cg + r  (nt < 8, r < 4; register kk = 4*nt + r).
//
//   * pre-activation / LayerNorm / ReLU: VALU, 32 channels per lane, row reductions over the 4
//     lanes of a member with v_permlane{16,32}_swap (no LDS, no 64-lane butterflies);
//   * scores   S[m][h] = sum_c z_k[m][c] * Q~[h][c]      : 32 MFMAs per tile (A = z straight from the
//     registers above, B = Q~ held in 32 VGPRs per segment), exact fp32 (k-ordered fmaf chain);
//   * softmax over the members of the segment: registers + 2 swaps per reduction;
//   * aggregation Z~[h][c] = sum_m aw[m][h] * z_v[m][c]   : 32 MFMAs per tile (A = z_v transposed
//     through a 5 KB LDS tile, B = alpha*w after an in-register 4x4 lane transpose);
//   * pos layers: v16[m][h] = z_v[m] . W2xv[h]            : 32 MFMAs per tile, consumed in registers.
//
// This removes the per-member 16-value cross-lane reductions and broadcasts of v1 (its dominant
// instruction count) and moves ~4k MACs per member from the VALU to the otherwise idle MFMA pipe.
#include "dd_kernels.hpp"

namespace dd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace v2 {

constexpr int WPITCH = 132;                    // pitch of the head-permuted W2k image
constexpr int WB_FLOATS = 128 * WPITCH;        // 67.6 KB weight buffer (W2k -> Gaussian tables -> W2v^T)
constexpr int TABP = 24 * 128;                 // Gaussian table stride per edge type (21 rows + 3 zero rows)
constexpr int ZS_PITCH = 80;                   // half tile [16 members][64 channels] + pad
constexpr int ZS_FLOATS = 16 * ZS_PITCH;

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void swap16_pair(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __uint_as_float(r0); b = __uint_as_float(r1);
}
__device__ __forceinline__ void swap32_pair(float& a, float& b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  const unsigned r0 = r[0], r1 = r[1];
  a = __uint_as_float(r0); b = __uint_as_float(r1);
}
// sum / max over the 4 lanes {l, l^16, l^32, l^48}
__device__ __forceinline__ float quad_sum(float v) { v = swap16_sum(v, v); return swap32_sum(v, v); }
__device__ __forceinline__ float quad_max(float v) { return swap32_max(swap16_max(v)); }

// ---- cooperative staging (all loads of a thread issued before its LDS stores) -------------------
template <int NT>
__device__ __forceinline__ void stage_w2k_permuted(float* WB, const float* __restrict__ W2k) {
  // global row r = h*8 + d  ->  LDS row d*16 + h (pitch 132): the 16 heads of a d land in 16 different bank slots
  constexpr int PER = (4096 + NT - 1) / NT;
  float4 tmp[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < 4096) tmp[k] = reinterpret_cast<const float4*>(W2k)[i];
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < 4096) {
      const int r = i >> 5, c4 = (i & 31) * 4;
      *reinterpret_cast<float4*>(&WB[((r & 7) * 16 + (r >> 3)) * WPITCH + c4]) = tmp[k];
    }
  }
}
template <int NT>
__device__ __forceinline__ void stage_plain(float* dst, const float* __restrict__ src, int n4) {
  constexpr int PER = (4096 + NT - 1) / NT;
  float4 tmp[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < n4) tmp[k] = reinterpret_cast<const float4*>(src)[i];
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * NT;
    if (i < n4) reinterpret_cast<float4*>(dst)[i] = tmp[k];
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
__device__ __forceinline__ void ln_relu32(float (&P)[32], const float* __restrict__ ln /*LDS: gamma[128], beta[128]*/, int cg) {
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) s += P[k];
  const float mean = quad_sum(s) * (1.0f / 128.0f);
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) { P[k] -= mean; v = fmaf(P[k], P[k], v); }
  const float rstd = __builtin_amdgcn_rsqf(quad_sum(v) * (1.0f / 128.0f) + 1e-5f);
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    const float4 g = *reinterpret_cast<const float4*>(ln + 16 * nt + 4 * cg);
    const float4 b = *reinterpret_cast<const float4*>(ln + 128 + 16 * nt + 4 * cg);
    P[4 * nt] = fmaxf(fmaf(P[4 * nt] * rstd, g.x, b.x), 0.f);
    P[4 * nt + 1] = fmaxf(fmaf(P[4 * nt + 1] * rstd, g.y, b.y), 0.f);
    P[4 * nt + 2] = fmaxf(fmaf(P[4 * nt + 2] * rstd, g.z, b.z), 0.f);
    P[4 * nt + 3] = fmaxf(fmaf(P[4 * nt + 3] * rstd, g.w, b.w), 0.f);
  }
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

template <int MODE, int MAXT, int NW>
struct Lds {
  static constexpr bool KNN = (MODE == M_NE || MODE == M_PE);
  static constexpr bool POS = (MODE == M_PE || MODE == M_PB);
  static constexpr bool TRIP = (MODE == M_BL);
  static constexpr int LNP = WB_FLOATS;                       // [4][128]  gamma_k, beta_k, gamma_v, beta_v
  static constexpr int WAO = LNP + 512;                       // [2][16][128] angle weights (BL), MFMA A-operand layout
  static constexpr int SCR0 = WAO + (TRIP ? 2 * 16 * 128 : 0);
  static constexpr int FE_SZ = KNN ? 16 * 20 : (TRIP ? MAXT * 256 : 0);
  static constexpr int FE = POS ? 0 : ZS_FLOATS;              // feature scratch sits behind the transpose tile
  static constexpr int UNI = POS ? FE_SZ : (ZS_FLOATS + FE_SZ > 16 * 132 ? ZS_FLOATS + FE_SZ : 16 * 132);
  static constexpr int SS = UNI;                              // [16] sum_m alpha*w per head
  static constexpr int SCRW = UNI + 16;
  static constexpr int TOTAL = SCR0 + NW * SCRW;
};

// Body of one workgroup (NW waves = NW segments).  `block` is the workgroup index within this mode's range and
// `smem` the workgroup's LDS (>= Lds<MODE,MAXT,NW>::TOTAL floats), so several modes can share one launch.
template <int MODE, int MAXT, int NW>
__device__ __forceinline__ void attn2_body(const AttnArgs& a, const int block, float* smem) {
  constexpr bool KNN = (MODE == M_NE || MODE == M_PE);
  constexpr bool POS = (MODE == M_PE || MODE == M_PB);
  constexpr bool TRIP = (MODE == M_BL);
  constexpr bool BOND = (MODE == M_NB || MODE == M_PB);
  constexpr int NT = NW * 64;
  using L = Lds<MODE, MAXT, NW>;
  constexpr int LNP = L::LNP, WAO = L::WAO, SCR0 = L::SCR0, FE = L::FE, SS = L::SS, SCRW = L::SCRW;
  (void)BOND;
  float* WB = smem;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int mm = lane & 15, cg = lane >> 4;            // member slot / channel group; also (head, row-group)
  float* scr = smem + SCR0 + wave * SCRW;

  const int N = a.NP + a.NL, NLm1 = a.NL - 1, Eb = a.NL * NLm1;
  const int nseg = (MODE == M_NE) ? a.B * N : (TRIP ? a.B * Eb : a.B * a.NL);
  const int M = KNN ? a.K : (TRIP ? a.NL - 2 : NLm1);
  const int T = (M + 15) >> 4;
  const int seg = block * NW + wave;
  const bool active = seg < nseg;
  long long* dbg = a.dbg_clock ? a.dbg_clock + (long)block * 16 : nullptr;
#define DD_STAMP(i) do { if (dbg && threadIdx.x == 0) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
  DD_STAMP(0);

  int b = 0, si = 0, sj = 0, node = 0;
  if (active) {
    if (MODE == M_NE) { b = seg / N; node = seg % N; }
    else if (TRIP) { b = seg / Eb; const int e = seg % Eb; si = e / NLm1; const int jp = e % NLm1; sj = jp + (jp >= si ? 1 : 0); }
    else { b = seg / a.NL; si = seg % a.NL; node = a.NP + si; }
  }
  const float* xb = a.x + (long)b * N * 3;
  const float* xl = xb + (long)a.NP * 3;
  const long nrow = (long)b * N + node;                // kNN list row (KNN modes)
  const long src_base = KNN ? (long)b * N : (long)b * a.NL;
  const long erow0 = (long)seg * NLm1;

  // ---- stage: W2k (head-permuted), LayerNorm parameters, angle weights -----------------------------------
  stage_w2k_permuted<NT>(WB, a.W2k);
  if (threadIdx.x < 64) {
    reinterpret_cast<float4*>(smem + LNP)[threadIdx.x] = reinterpret_cast<const float4*>(a.lnk)[threadIdx.x];
    reinterpret_cast<float4*>(smem + LNP + 256)[threadIdx.x] = reinterpret_cast<const float4*>(a.lnv)[threadIdx.x];
  }
  if (TRIP) {
    for (int i = threadIdx.x; i < 16 * 32; i += NT) {
      reinterpret_cast<float4*>(smem + WAO)[i] = reinterpret_cast<const float4*>(a.Wakp)[i];
      reinterpret_cast<float4*>(smem + WAO + 16 * 128)[i] = reinterpret_cast<const float4*>(a.Wavp)[i];
    }
  }
  __syncthreads();
  DD_STAMP(1);

  // ---- Q~ as the MFMA B operand: lane (h = mm, cg) holds Q~[h][c(kk, cg)] ----------------------------------
  float Qb[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) Qb[k] = 0.f;
  if (active) {
    const float4 q0 = *reinterpret_cast<const float4*>(a.q + (long)seg * 128 + mm * 8);
    const float4 q1 = *reinterpret_cast<const float4*>(a.q + (long)seg * 128 + mm * 8 + 4);
    const float qd[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      const float* wr = WB + (d * 16 + mm) * WPITCH + 4 * cg;
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 w = *reinterpret_cast<const float4*>(wr + 16 * nt);
        Qb[4 * nt] = fmaf(qd[d], w.x, Qb[4 * nt]);
        Qb[4 * nt + 1] = fmaf(qd[d], w.y, Qb[4 * nt + 1]);
        Qb[4 * nt + 2] = fmaf(qd[d], w.z, Qb[4 * nt + 2]);
        Qb[4 * nt + 3] = fmaf(qd[d], w.w, Qb[4 * nt + 3]);
      }
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) Qb[k] *= 0.35355339059327373f;
  }
  DD_STAMP(2);

  // ---- BL: angle codes of all members (lane m computes member m), kept in LDS ---------------------------------
  if (TRIP && active && lane < M) {
    const int lo = si < sj ? si : sj, hi = si < sj ? sj : si;
    int k = lane;
    if (k >= lo) ++k;
    if (k >= hi) ++k;
    const float ax = xl[3 * sj] - xl[3 * si], ay = xl[3 * sj + 1] - xl[3 * si + 1], az = xl[3 * sj + 2] - xl[3 * si + 2];
    const float bx = xl[3 * k] - xl[3 * si], by = xl[3 * k + 1] - xl[3 * si + 1], bz = xl[3 * k + 2] - xl[3 * si + 2];
    const float dot = ax * bx + ay * by + az * bz;
    const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
    const float th = atan2f(sqrtf(cx * cx + cy * cy + cz * cz), dot);
    float* c = scr + FE + 16 * lane;
    c[0] = th;
    c[1] = sinf(th);            c[7] = cosf(th);
    c[2] = sinf(th * 2.0f);     c[8] = cosf(th * 2.0f);
    c[3] = sinf(th * 3.0f);     c[9] = cosf(th * 3.0f);
    c[4] = c[1];                c[10] = c[7];
    c[5] = sinf(th * 0.5f);     c[11] = cosf(th * 0.5f);
    c[6] = sinf(th * (1.0f / 3.0f)); c[12] = cosf(th * (1.0f / 3.0f));
    c[13] = 0.f; c[14] = 0.f; c[15] = 0.f;
  }
  DD_STAMP(3);

  // pre-activation of tile t for the k (pass = 0) or v (pass = 1) MLP, in the lane layout described above
  auto build_pre = [&](int t, int pass, float (&P)[32]) {
    const int m = 16 * t + mm;
    const int mc = m < M ? m : M - 1;                  // clipped: out-of-range slots replay the last member
    const float* tab_d = pass ? a.vd : a.kd;  const int ld_d = pass ? a.ld_vd : a.ld_kd;
    const float* tab_s = pass ? a.vs : a.ks;  const int ld_s = pass ? a.ld_vs : a.ld_ks;
    const float* tab_e = pass ? a.ve : a.ke;  const int ld_e = pass ? a.ld_ve : a.ld_ke;
    if (KNN) {
      const int j = a.nbr[nrow * a.K + mc];
      const float rx = xb[3 * node] - xb[3 * j], ry = xb[3 * node + 1] - xb[3 * j + 1], rz = xb[3 * node + 2] - xb[3 * j + 2];
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      const long drow = (MODE == M_NE) ? (long)seg : (long)b * a.NL + si;
      load_row(P, tab_d + drow * ld_d, cg);
      add_row(P, tab_s + (src_base + j) * ld_s, cg);
      // Gaussian / type part on the matrix cores:  P^T[c][m] += sum_g A_ty[g][c] * F[m][g]  with F = 20 Gaussians and
      // the per-type constant (24 rows, 6 k-steps).  Lane (mm, cg) supplies F[mm][4s+cg] and the table entries of row
      // 4s+cg; the result lands directly in the P layout.  A tile whose members mix ligand and protein sources runs
      // once per table with the other members' features zeroed.
      float F[6];
#pragma unroll
      for (int s = 0; s < 5; ++s) F[s] = gauss_feat(d, 4 * s + cg);
      F[5] = cg == 0 ? 1.0f : 0.0f;
      const bool hi = j < a.NP;
      const float* tab = WB + (node < a.NP ? 1 : 0) * TABP + cg * 128 + mm * 4;
      f32x4 acc[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{P[4 * nt], P[4 * nt + 1], P[4 * nt + 2], P[4 * nt + 3]};
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const bool want = half ? hi : !hi;
        if (__builtin_amdgcn_ballot_w64(want) != 0ull) {
          const float* tb = tab + half * 2 * TABP;
#pragma unroll
          for (int s = 0; s < 6; ++s) {
            const float f = want ? F[s] : 0.0f;
            const float4 w0 = *reinterpret_cast<const float4*>(tb + s * 512);
            const float4 w1 = *reinterpret_cast<const float4*>(tb + s * 512 + 64);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, f, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, f, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, f, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, f, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, f, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, f, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, f, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, f, acc[7], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) { P[4 * nt] = acc[nt][0]; P[4 * nt + 1] = acc[nt][1]; P[4 * nt + 2] = acc[nt][2]; P[4 * nt + 3] = acc[nt][3]; }
    } else if (!TRIP) {
      const int j = mc + (mc >= si ? 1 : 0);
      load_row(P, tab_d + ((long)b * a.NL + si) * ld_d, cg);
      add_row(P, tab_s + (src_base + j) * ld_s, cg);
      add_row(P, tab_e + (erow0 + mc) * ld_e, cg);
    } else {
      const int lo = si < sj ? si : sj, hi = si < sj ? sj : si;
      int k = mc;
      if (k >= lo) ++k;
      if (k >= hi) ++k;
      const long kj = (long)b * Eb + sj * NLm1 + (k - (k > sj ? 1 : 0));
      load_row(P, (pass ? a.Rv : a.Rk) + (long)seg * 128, cg);
      add_row(P, tab_e + kj * ld_e, cg);
      // angle part on the matrix cores (13 codes padded to 16 = 4 k-steps), same scheme as the Gaussian tables
      const float* cd = scr + FE + 16 * mc + cg;
      const float* tb = smem + WAO + pass * 16 * 128 + cg * 128 + mm * 4;
      f32x4 acc[8];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[nt] = f32x4{P[4 * nt], P[4 * nt + 1], P[4 * nt + 2], P[4 * nt + 3]};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float f = cd[4 * s];
        const float4 w0 = *reinterpret_cast<const float4*>(tb + s * 512);
        const float4 w1 = *reinterpret_cast<const float4*>(tb + s * 512 + 64);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, f, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, f, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, f, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, f, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, f, acc[4], 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, f, acc[5], 0, 0, 0);
        acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, f, acc[6], 0, 0, 0);
        acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, f, acc[7], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) { P[4 * nt] = acc[nt][0]; P[4 * nt + 1] = acc[nt][1]; P[4 * nt + 2] = acc[nt][2]; P[4 * nt + 3] = acc[nt][3]; }
    }
    ln_relu32(P, smem + LNP + pass * 256, cg);
  };

  // ---- Gaussian tables for pass 1 --------------------------------------------------------------------------
  __syncthreads();                                     // all waves done with W2k
  if (KNN) stage_plain<NT>(WB, a.Akp, 4 * 24 * 32);
  __syncthreads();
  DD_STAMP(4);

  // ---- pass 1: scores S[t][r] = score[member 16t + 4cg + r][head mm] ---------------------------------------
  f32x4 S[MAXT];
  float ssum = 0.f;                                    // sum_m alpha*w for head mm
  if (active) {
    if (TRIP) wave_lds_sync();                         // angle codes visible
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t < T) {
        float P[32];
        build_pre(t, 0, P);
        S[t] = mfma_rows(P, Qb);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * t + 4 * cg + r >= M) S[t][r] = -INFINITY;
      } else {
        S[t] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
    }
    DD_STAMP(5);
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
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = 16 * t + 4 * cg + r;
        float w = 1.0f;
        if (KNN) w = a.ew[nrow * a.K + (m < M ? m : 0)];
        const float aw = (m < M) ? (S[t][r] / sum) * w : 0.f;
        S[t][r] = aw;
        ssum += aw;
      }
    ssum = quad_sum(ssum);
  } else {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  DD_STAMP(6);
  if (KNN) {
    __syncthreads();
    stage_plain<NT>(WB, a.Avp, 4 * 24 * 32);
    __syncthreads();
  }
  DD_STAMP(7);

  // ---- pass 2 -----------------------------------------------------------------------------------------------
  if (POS) {
    if (active) {
      float Wv[32];
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 w = *reinterpret_cast<const float4*>(a.W2v16 + mm * 128 + 16 * nt + 4 * cg);
        Wv[4 * nt] = w.x; Wv[4 * nt + 1] = w.y; Wv[4 * nt + 2] = w.z; Wv[4 * nt + 3] = w.w;
      }
      const float bv = a.b2v16[mm];
      float dx = 0.f, dy = 0.f, dz = 0.f;
#pragma unroll
      for (int t = 0; t < MAXT; ++t) {
        if (t < T) {
          float P[32];
          build_pre(t, 1, P);
          const f32x4 V = mfma_rows(P, Wv);             // V[r] = v16[member 16t+4cg+r][head mm] - bias
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = 16 * t + 4 * cg + r;
            const int mc = m < M ? m : 0;
            int j;
            const float* xs;
            if (KNN) { j = a.nbr[nrow * a.K + mc]; xs = xb; }
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

  f32x4 Z[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) Z[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (active) {
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      if (t < T) {
        // B operand: alpha*w transposed over the 4 lanes of a head so that lane kq holds members 4ks + kq
        float w0 = S[t][0], w1 = S[t][1], w2 = S[t][2], w3 = S[t][3];
        swap16_pair(w0, w1); swap16_pair(w2, w3);
        swap32_pair(w0, w2); swap32_pair(w1, w3);
        const float awT[4] = {w0, w1, w2, w3};
        float P[32];
        build_pre(t, 1, P);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(scr + mm * ZS_PITCH + 16 * q + 4 * cg) =
                make_float4(P[16 * half + 4 * q], P[16 * half + 4 * q + 1], P[16 * half + 4 * q + 2], P[16 * half + 4 * q + 3]);
          wave_lds_sync();
#pragma unroll
          for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
              const float av = scr[(4 * ks + cg) * ZS_PITCH + 16 * q + mm];   // z_v[member 4ks+cg][channel 16(4half+q)+mm]
              Z[4 * half + q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, awT[ks], Z[4 * half + q], 0, 0, 0);
            }
          }
          wave_lds_sync();
        }
      }
    }
  }
  DD_STAMP(8);

  // ---- epilogue: out[o] = W2v[o,:] . Z~[head(o),:] + b2v[o] * sum_m alpha*w ----------------------------------
  __syncthreads();                                     // pass-2 tables dead
  stage_plain<NT>(WB, a.W2vT, 4096);
  if (active) {
    wave_lds_sync();
    if (cg == 0) scr[SS + mm] = ssum;
    // Z[nt][r] = Z~[head mm][channel 16nt + 4cg + r]  ->  zt[h][c], pitch 132 (aliases the tile scratch)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
      *reinterpret_cast<float4*>(scr + mm * 132 + 16 * nt + 4 * cg) = make_float4(Z[nt][0], Z[nt][1], Z[nt][2], Z[nt][3]);
  }
  __syncthreads();
  DD_STAMP(9);
  if (active) {
    const int hsel = lane >> 2;
    float o0 = 0.f, o1 = 0.f;
    const float* zrow = scr + hsel * 132;
#pragma unroll 4
    for (int c4 = 0; c4 < 32; ++c4) {
      const float4 z = *reinterpret_cast<const float4*>(zrow + 4 * c4);
      const float2 w0 = *reinterpret_cast<const float2*>(&WB[(4 * c4 + 0) * 128 + 2 * lane]);
      const float2 w1 = *reinterpret_cast<const float2*>(&WB[(4 * c4 + 1) * 128 + 2 * lane]);
      const float2 w2 = *reinterpret_cast<const float2*>(&WB[(4 * c4 + 2) * 128 + 2 * lane]);
      const float2 w3 = *reinterpret_cast<const float2*>(&WB[(4 * c4 + 3) * 128 + 2 * lane]);
      o0 = fmaf(w0.x, z.x, o0); o1 = fmaf(w0.y, z.x, o1);
      o0 = fmaf(w1.x, z.y, o0); o1 = fmaf(w1.y, z.y, o1);
      o0 = fmaf(w2.x, z.z, o0); o1 = fmaf(w2.y, z.z, o1);
      o0 = fmaf(w3.x, z.w, o0); o1 = fmaf(w3.y, z.w, o1);
    }
    const float sh = scr[SS + hsel];
    const float2 bb = *reinterpret_cast<const float2*>(a.b2v + 2 * lane);
    o0 = fmaf(bb.x, sh, o0);
    o1 = fmaf(bb.y, sh, o1);
    float* dst;
    const bool nb_assign = (MODE == M_NB) && a.out_assign;
    if (MODE == M_NB && !nb_assign) dst = a.out + ((long)b * N + node) * 128 + 2 * lane;
    else dst = a.out + (long)seg * 128 + 2 * lane;
    if (MODE == M_NE || nb_assign) {
      *reinterpret_cast<float2*>(dst) = make_float2(o0, o1);
    } else {
      const float2 old = *reinterpret_cast<const float2*>(dst);
      *reinterpret_cast<float2*>(dst) = make_float2(old.x + o0, old.y + o1);
    }
  }
  DD_STAMP(10);
#undef DD_STAMP
}

template <int MODE, int MAXT, int NW>
__global__ __launch_bounds__(NW * 64) void k_attn2(const AttnArgs a) {
  __shared__ __attribute__((aligned(16))) float smem[Lds<MODE, MAXT, NW>::TOTAL];
  attn2_body<MODE, MAXT, NW>(a, blockIdx.x, smem);
}

constexpr int imax(int a, int b) { return a > b ? a : b; }

// The three sub-layers that read the *old* h / h_bond (NE, NB, BL) are independent: one launch, the longest
// workgroups (BL) first, the 30-workgroup NB hidden in the tail instead of costing a launch of its own.
template <int MAXT>
__global__ __launch_bounds__(512) void k_attn2_node(const AttnArgs ne, const AttnArgs nb, const AttnArgs bl, int n_bl, int n_ne) {
  constexpr int SZ = imax(imax(Lds<M_NE, 2, 8>::TOTAL, Lds<M_NB, MAXT, 8>::TOTAL), Lds<M_BL, MAXT, 8>::TOTAL);
  __shared__ __attribute__((aligned(16))) float smem[SZ];
  const int blk = blockIdx.x;
  if (blk < n_bl) attn2_body<M_BL, MAXT, 8>(bl, blk, smem);
  else if (blk < n_bl + n_ne) attn2_body<M_NE, 2, 8>(ne, blk - n_bl, smem);
  else attn2_body<M_NB, MAXT, 8>(nb, blk - n_bl - n_ne, smem);
}
// Same for the two coordinate sub-layers (both write their own delta buffer; x is updated afterwards).
template <int MAXT>
__global__ __launch_bounds__(512) void k_attn2_pos(const AttnArgs pe, const AttnArgs pb, int n_pe) {
  constexpr int SZ = imax(Lds<M_PE, 2, 8>::TOTAL, Lds<M_PB, MAXT, 8>::TOTAL);
  __shared__ __attribute__((aligned(16))) float smem[SZ];
  const int blk = blockIdx.x;
  if (blk < n_pe) attn2_body<M_PE, 2, 8>(pe, blk, smem);
  else attn2_body<M_PB, MAXT, 8>(pb, blk - n_pe, smem);
}

template <int MODE, int MAXT, int NW>
static int launch_mode(const AttnArgs& a, int nseg, hipStream_t st) {
  if (nseg <= 0) return DD_OK;
  hipLaunchKernelGGL((k_attn2<MODE, MAXT, NW>), dim3((nseg + NW - 1) / NW), dim3(NW * 64), 0, st, a);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

}  // namespace v2

int launch_attn2(int mode, const AttnArgs& a, hipStream_t st) {
  using namespace v2;
  const int N = a.NP + a.NL;
  const bool small = a.NL <= 33;                     // NL-1 <= 32 members -> 2 tiles
  switch (mode) {
    case M_NE: return launch_mode<M_NE, 2, 8>(a, a.B * N, st);
    case M_PE: return launch_mode<M_PE, 2, 8>(a, a.B * a.NL, st);
    case M_NB: return small ? launch_mode<M_NB, 2, 8>(a, a.B * a.NL, st) : launch_mode<M_NB, 4, 8>(a, a.B * a.NL, st);
    case M_PB: return small ? launch_mode<M_PB, 2, 8>(a, a.B * a.NL, st) : launch_mode<M_PB, 4, 8>(a, a.B * a.NL, st);
    case M_BL: return small ? launch_mode<M_BL, 2, 8>(a, a.B * a.NL * (a.NL - 1), st)
                            : launch_mode<M_BL, 4, 6>(a, a.B * a.NL * (a.NL - 1), st);
  }
  return DD_ERR_BAD_ARG;
}

// Fused launches (ligands with <= 33 atoms: every mode uses 8-wave workgroups).  Returns DD_ERR_UNSUPPORTED_SHAPE
// for larger ligands; the caller then falls back to one launch per sub-layer.
int launch_attn2_node(const AttnArgs& ne, const AttnArgs& nb, const AttnArgs& bl, hipStream_t st) {
  using namespace v2;
  if (ne.NL > 33) return DD_ERR_UNSUPPORTED_SHAPE;
  const int N = ne.NP + ne.NL;
  const int n_ne = (ne.B * N + 7) / 8, n_nb = (ne.B * ne.NL + 7) / 8, n_bl = (ne.B * ne.NL * (ne.NL - 1) + 7) / 8;
  hipLaunchKernelGGL((k_attn2_node<2>), dim3(n_bl + n_ne + n_nb), dim3(512), 0, st, ne, nb, bl, n_bl, n_ne);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_attn2_pos(const AttnArgs& pe, const AttnArgs& pb, hipStream_t st) {
  using namespace v2;
  if (pe.NL > 33) return DD_ERR_UNSUPPORTED_SHAPE;
  const int n = (pe.B * pe.NL + 7) / 8;
  hipLaunchKernelGGL((k_attn2_pos<2>), dim3(2 * n), dim3(512), 0, st, pe, pb, n);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

}  // namespace dd
