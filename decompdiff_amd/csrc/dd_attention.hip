// Fused "segment attention" kernels — the message-passing hot path.
//
// One wavefront owns one attention segment (all edges / triplets that share a softmax):
//   NE  node_layer_with_edge : segment = node i,           members = its K kNN edges        (uni_transformer_edge.py:42-74)
//   NB  node_layer_with_bond : segment = ligand atom i,    members = NL-1 incoming bonds    (same class, bond edges)
//   BL  bond_layer           : segment = bond edge (j->i), members = NL-2 triplets k->j->i  (uni_transformer_edge.py:125-167)
//   PE  pos_layer_with_edge  : segment = ligand atom i,    members = its K kNN edges        (uni_transformer_edge.py:188-210)
//   PB  pos_layer_with_bond  : segment = ligand atom i,    members = NL-1 incoming bonds
// replacing, per sub-layer, the reference's  cat -> 2x MLP(340/384/437) -> scatter_softmax ->
// scatter_sum  chain (torch ATen + torch_scatter + torch_sparse kernels) by a single launch:
//
//   * first Linear of the k/v MLPs: factorised (packing.py) -> gathered 512-byte rows of the
//     per-node / per-bond projection tables (coalesced, one float2 per lane) + a 20-term
//     Gaussian (or 13-term angular) contraction whose table is staged in LDS / registers;
//   * LayerNorm + ReLU per member with wave-wide butterflies;
//   * second Linear of k folded into the query:  (q_h . W2k_h) . z  ==  q_h . (W2k z)_h,
//     so a score costs 16x128 MACs instead of 128x128 (b2k cancels inside the softmax);
//   * second Linear of v applied after aggregation:  W2v_h . (sum_m a_m z_m) + b2v_h sum_m a_m;
//   * segment softmax / sum are wave-local (segments are contiguous and fixed-length), no atomics.
//
// Lane layout: lane l holds hidden channels 2l, 2l+1 of the member being processed.
// Workgroup = 8 waves = 8 segments sharing LDS-staged weights (W2k -> Gaussian tables ->
// W2v^T re-use one 64 KB buffer), 2 waves per SIMD.
#include "dd_kernels.hpp"

namespace dd {

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cooperative global -> LDS copy by the 512-thread workgroup.  All loads of a thread are issued
// before its first LDS store (8 independent 16-byte loads in flight per lane) — the naive
// load/store loop serialises one L2 round trip per iteration.
template <int NF4>
__device__ __forceinline__ void stage_weights(float* dst, const float* __restrict__ src) {
  constexpr int PER = (NF4 + 511) / 512;
  const float4* s4 = reinterpret_cast<const float4*>(src);
  float4* d4 = reinterpret_cast<float4*>(dst);
  float4 tmp[PER];
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * 512;
    if (i < NF4) tmp[k] = s4[i];
  }
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = threadIdx.x + k * 512;
    if (i < NF4) d4[i] = tmp[k];
  }
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int MODE, int MMAX>
__global__ __launch_bounds__(512) void k_attn(const AttnArgs a) {
  constexpr bool KNN = (MODE == M_NE || MODE == M_PE);
  constexpr bool POS = (MODE == M_PE || MODE == M_PB);
  constexpr bool TRIP = (MODE == M_BL);
  // per-wave scratch layout (floats)
  constexpr int SC = 0;                      // [MMAX][16] scores -> alpha*w
  constexpr int SRC = SC + MMAX * 16;        // [MMAX] int   source row / node id
  constexpr int TY = SRC + MMAX;             // [MMAX] int   edge type
  constexpr int WGT = TY + MMAX;             // [MMAX]       e_w
  constexpr int REL = WGT + MMAX;            // [MMAX][4]    rel_x (pos modes)
  constexpr int FEAT = REL + MMAX * 4;       // [MMAX][20] Gaussians | [MMAX][16] angle code
  constexpr int FEAT_SZ = KNN ? MMAX * 20 : (TRIP ? MMAX * 16 : 0);
  constexpr int ZT_SZ = POS ? 0 : 16 * 132;  // epilogue transposer (aliases everything above)
  constexpr int UNI = cmax(FEAT + FEAT_SZ, ZT_SZ);
  constexpr int SSUM = UNI;                  // [16]
  constexpr int SCR = UNI + 16;
  constexpr int WL = 128 * 128;

  __shared__ __attribute__((aligned(16))) float smem[WL + 8 * SCR];
  float* Wl = smem;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* scr = smem + WL + wave * SCR;
  int* scri = reinterpret_cast<int*>(scr);

  const int N = a.NP + a.NL, NLm1 = a.NL - 1, Eb = a.NL * NLm1;
  const int nseg = (MODE == M_NE) ? a.B * N : (TRIP ? a.B * Eb : a.B * a.NL);
  const int M = KNN ? a.K : (TRIP ? a.NL - 2 : NLm1);
  const int seg = blockIdx.x * 8 + wave;
  const bool active = seg < nseg;

  // ---- segment decode -------------------------------------------------------------------
  int b = 0, si = 0, sj = 0, node = 0;
  if (active) {
    if (MODE == M_NE) { b = seg / N; node = seg % N; }
    else if (TRIP) { b = seg / Eb; int e = seg % Eb; si = e / NLm1; int jp = e % NLm1; sj = jp + (jp >= si ? 1 : 0); }
    else { b = seg / a.NL; si = seg % a.NL; node = a.NP + si; }
  }
  const float* xb = a.x + (long)b * N * 3;
  const float* xl = xb + (long)a.NP * 3;

  // ---- phase A: Q~[h][c] = scale * sum_d q[h*8+d] * W2k[h*8+d][c] ------------------------------
  stage_weights<WL / 4>(Wl, a.W2k);
  __syncthreads();
  float Qa[16], Qb[16];
  if (active) {
    const float2 qv = *reinterpret_cast<const float2*>(a.q + (long)seg * 128 + 2 * lane);
#pragma unroll
    for (int h = 0; h < 16; ++h) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const int idx = h * 8 + d;
        const float qs = lane_bcast((idx & 1) ? qv.y : qv.x, idx >> 1);
        const float2 w = *reinterpret_cast<const float2*>(&Wl[idx * 128 + 2 * lane]);
        s0 = fmaf(qs, w.x, s0);
        s1 = fmaf(qs, w.y, s1);
      }
      Qa[h] = s0 * 0.35355339059327373f;   // 1/sqrt(8)
      Qb[h] = s1 * 0.35355339059327373f;
      asm volatile("" ::: "memory");       // keep the 128 LDS operand reads from being hoisted en bloc
    }
  }

  // ---- member geometry (lane m describes member m) ----------------------------------------
  if (active && lane < M) {
    const int m = lane;
    if (KNN) {
      const long nrow = (long)b * N + node;
      const int j = a.nbr[nrow * a.K + m];
      const float rx = xb[3 * node] - xb[3 * j], ry = xb[3 * node + 1] - xb[3 * j + 1], rz = xb[3 * node + 2] - xb[3 * j + 2];
      const float d = sqrtf(rx * rx + ry * ry + rz * rz);
      scri[SRC + m] = j;
      scri[TY + m] = 2 * (j < a.NP ? 1 : 0) + (node < a.NP ? 1 : 0);
      scr[WGT + m] = a.ew[nrow * a.K + m];
      scr[REL + 4 * m] = rx; scr[REL + 4 * m + 1] = ry; scr[REL + 4 * m + 2] = rz;
#pragma unroll 1
      for (int g = 0; g < 20; ++g) scr[FEAT + 20 * m + g] = gauss_feat(d, g);
    } else if (!TRIP) {
      const int j = m + (m >= si ? 1 : 0);
      scri[SRC + m] = j;
      scr[WGT + m] = 1.0f;
      scr[REL + 4 * m] = xl[3 * si] - xl[3 * j];
      scr[REL + 4 * m + 1] = xl[3 * si + 1] - xl[3 * j + 1];
      scr[REL + 4 * m + 2] = xl[3 * si + 2] - xl[3 * j + 2];
    } else {
      const int lo = si < sj ? si : sj, hi = si < sj ? sj : si;
      int k = m;
      if (k >= lo) ++k;
      if (k >= hi) ++k;
      scri[SRC + m] = b * Eb + sj * NLm1 + (k - (k > sj ? 1 : 0));   // global row of edge (k -> j)
      scr[WGT + m] = 1.0f;
      // angle at i between (j - i) and (k - i)   (uni_transformer_edge.py:132-137)
      const float ax = xl[3 * sj] - xl[3 * si], ay = xl[3 * sj + 1] - xl[3 * si + 1], az = xl[3 * sj + 2] - xl[3 * si + 2];
      const float bx = xl[3 * k] - xl[3 * si], by = xl[3 * k + 1] - xl[3 * si + 1], bz = xl[3 * k + 2] - xl[3 * si + 2];
      const float dot = ax * bx + ay * by + az * bz;
      const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
      const float th = atan2f(sqrtf(cx * cx + cy * cy + cz * cz), dot);
      float* c = scr + FEAT + 16 * m;
      c[0] = th;
      c[1] = sinf(th);            c[7] = cosf(th);
      c[2] = sinf(th * 2.0f);     c[8] = cosf(th * 2.0f);
      c[3] = sinf(th * 3.0f);     c[9] = cosf(th * 3.0f);
      c[4] = c[1];                c[10] = c[7];
      c[5] = sinf(th * 0.5f);     c[11] = cosf(th * 0.5f);
      c[6] = sinf(th * (1.0f / 3.0f)); c[12] = cosf(th * (1.0f / 3.0f));
      c[13] = 0.f; c[14] = 0.f; c[15] = 0.f;
    }
  }

  // ---- segment constants ----------------------------------------------------------------------
  float2 ck = make_float2(0.f, 0.f), cv = make_float2(0.f, 0.f);
  if (active) {
    if (TRIP) {
      const float dx = xl[3 * si] - xl[3 * sj], dy = xl[3 * si + 1] - xl[3 * sj + 1], dz = xl[3 * si + 2] - xl[3 * sj + 2];
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      const float gl = gauss_feat(d, lane < 20 ? lane : 0);
#pragma unroll
      for (int g = 0; g < 20; ++g) {
        const float gg = lane_bcast(gl, g);
        const float2 wk = *reinterpret_cast<const float2*>(a.Wg2k + g * 128 + 2 * lane);
        const float2 wv = *reinterpret_cast<const float2*>(a.Wg2v + g * 128 + 2 * lane);
        ck.x = fmaf(wk.x, gg, ck.x); ck.y = fmaf(wk.y, gg, ck.y);
        cv.x = fmaf(wv.x, gg, cv.x); cv.y = fmaf(wv.y, gg, cv.y);
      }
    } else {
      const long drow = (MODE == M_NE) ? (long)seg : (long)b * a.NL + si;
      ck = *reinterpret_cast<const float2*>(a.kd + drow * a.ld_kd + 2 * lane);
      cv = *reinterpret_cast<const float2*>(a.vd + drow * a.ld_vd + 2 * lane);
    }
  }
  const long src_base = (MODE == M_NE || MODE == M_PE) ? (long)b * N : (long)b * a.NL;
  const long erow0 = (long)seg * NLm1;      // NB/PB: bond rows of this dst atom
  // gathered first-Linear partial sums of member m (issued one member ahead of their use; the two rows
  // are summed only at the point of use so that the loads stay in flight across the previous member)
  struct Rows { float2 r, e; };
  auto gather = [&](int m, const float* tab_s, int ld_s, const float* tab_e, int ld_e) -> Rows {
    const int sidx = __builtin_amdgcn_readfirstlane(scri[SRC + m]);
    Rows o;
    o.e = make_float2(0.f, 0.f);
    if (KNN) {
      o.r = *reinterpret_cast<const float2*>(tab_s + (src_base + sidx) * ld_s + 2 * lane);
    } else if (!TRIP) {
      o.r = *reinterpret_cast<const float2*>(tab_s + (src_base + sidx) * ld_s + 2 * lane);
      o.e = *reinterpret_cast<const float2*>(tab_e + (erow0 + m) * ld_e + 2 * lane);
    } else {
      o.r = *reinterpret_cast<const float2*>(tab_e + (long)sidx * ld_e + 2 * lane);
    }
    return o;
  };

  // ---- stage pass-1 table ------------------------------------------------------------------------
  __syncthreads();                           // everyone is done with W2k
  if (KNN) { stage_weights<4 * 21 * 128 / 4>(Wl, a.Ak); }
  __syncthreads();

  // ---- pass 1: scores ------------------------------------------------------------------------------
  if (active) {
    const float2 g = *reinterpret_cast<const float2*>(a.lnk + 2 * lane);
    const float2 be = *reinterpret_cast<const float2*>(a.lnk + 128 + 2 * lane);
    float wa0[13], wa1[13];
    if (TRIP) {
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        const float2 w = *reinterpret_cast<const float2*>(a.Wak + t * 128 + 2 * lane);
        wa0[t] = w.x; wa1[t] = w.y;
      }
    }
    wave_lds_sync();
    Rows nxt;
    nxt.r = nxt.e = make_float2(0.f, 0.f);
    if (M > 0) nxt = gather(0, a.ks, a.ld_ks, a.ke, a.ld_ke);
    for (int m = 0; m < M; ++m) {
      float2 pre = make_float2(ck.x + (nxt.r.x + nxt.e.x), ck.y + (nxt.r.y + nxt.e.y));
      // consume the rows fetched during the previous member BEFORE issuing the next fetch: hipcc waits
      // vmcnt(0) at a loop-carried load's first use, which would otherwise also drain the new loads
      asm volatile("" : "+v"(pre.x), "+v"(pre.y) : : "memory");
      if (m + 1 < M) nxt = gather(m + 1, a.ks, a.ld_ks, a.ke, a.ld_ke);
      if (KNN) {
        const int ty = __builtin_amdgcn_readfirstlane(scri[TY + m]);
        const float* tab = Wl + ty * 21 * 128 + 2 * lane;
        const float4* G4 = reinterpret_cast<const float4*>(scr + FEAT + 20 * m);
#pragma unroll
        for (int g4 = 0; g4 < 5; ++g4) {
          const float4 gv = G4[g4];
          float2 t0 = *reinterpret_cast<const float2*>(tab + (4 * g4 + 0) * 128);
          float2 t1 = *reinterpret_cast<const float2*>(tab + (4 * g4 + 1) * 128);
          float2 t2 = *reinterpret_cast<const float2*>(tab + (4 * g4 + 2) * 128);
          float2 t3 = *reinterpret_cast<const float2*>(tab + (4 * g4 + 3) * 128);
          pre.x = fmaf(t0.x, gv.x, pre.x); pre.y = fmaf(t0.y, gv.x, pre.y);
          pre.x = fmaf(t1.x, gv.y, pre.x); pre.y = fmaf(t1.y, gv.y, pre.y);
          pre.x = fmaf(t2.x, gv.z, pre.x); pre.y = fmaf(t2.y, gv.z, pre.y);
          pre.x = fmaf(t3.x, gv.w, pre.x); pre.y = fmaf(t3.y, gv.w, pre.y);
        }
        const float2 tc = *reinterpret_cast<const float2*>(tab + 20 * 128);
        pre.x += tc.x; pre.y += tc.y;
      } else if (TRIP) {
        const float4* C4 = reinterpret_cast<const float4*>(scr + FEAT + 16 * m);
        const float4 c0 = C4[0], c1 = C4[1], c2 = C4[2], c3 = C4[3];
        const float cc[13] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x};
#pragma unroll
        for (int t = 0; t < 13; ++t) { pre.x = fmaf(wa0[t], cc[t], pre.x); pre.y = fmaf(wa1[t], cc[t], pre.y); }
      }
      ln_relu2(pre.x, pre.y, g.x, g.y, be.x, be.y);
      float p[16];
#pragma unroll
      for (int h = 0; h < 16; ++h) p[h] = fmaf(pre.y, Qb[h], pre.x * Qa[h]);
      const float s = reduce16(p, lane);
      if ((lane & 3) == 0) scr[SC + 16 * m + head_of_lane(lane)] = s;
    }
    wave_lds_sync();
    // ---- segment softmax over members, per head (scatter_softmax semantics: max-shift, exp, / sum)
    {
      const int h = lane & 15, part = lane >> 4;
      float mx = -INFINITY;
      for (int m = part; m < M; m += 4) mx = fmaxf(mx, scr[SC + 16 * m + h]);
      mx = swap32_max(swap16_max(mx));
      float sum = 0.f;
      for (int m = part; m < M; m += 4) {
        const float e = expf(scr[SC + 16 * m + h] - mx);
        scr[SC + 16 * m + h] = e;
        sum += e;
      }
      sum = swap16_sum(sum, sum);
      sum = swap32_sum(sum, sum);
      float tot = 0.f;
      for (int m = part; m < M; m += 4) {
        const float aw = (scr[SC + 16 * m + h] / sum) * scr[WGT + m];
        scr[SC + 16 * m + h] = aw;
        tot += aw;
      }
      tot = swap16_sum(tot, tot);
      tot = swap32_sum(tot, tot);
      if (part == 0) scr[SSUM + h] = tot;
    }
    wave_lds_sync();
  }

  // ---- stage pass-2 table ------------------------------------------------------------------------
  if (KNN) {
    __syncthreads();
    stage_weights<4 * 21 * 128 / 4>(Wl, a.Av);
    __syncthreads();
  }

  // ---- pass 2: values ------------------------------------------------------------------------------
  float Za[16], Zb[16];
#pragma unroll
  for (int h = 0; h < 16; ++h) { Za[h] = 0.f; Zb[h] = 0.f; }
  float dxa = 0.f, dya = 0.f, dza = 0.f;
  if (active) {
    const float2 g = *reinterpret_cast<const float2*>(a.lnv + 2 * lane);
    const float2 be = *reinterpret_cast<const float2*>(a.lnv + 128 + 2 * lane);
    float wa0[13], wa1[13];
    if (TRIP) {
#pragma unroll
      for (int t = 0; t < 13; ++t) {
        const float2 w = *reinterpret_cast<const float2*>(a.Wav + t * 128 + 2 * lane);
        wa0[t] = w.x; wa1[t] = w.y;
      }
    }
    float Va[16], Vb[16];
    float bv16 = 0.f;
    if (POS) {
#pragma unroll
      for (int h = 0; h < 16; ++h) {
        const float2 w = *reinterpret_cast<const float2*>(a.W2v16 + h * 128 + 2 * lane);
        Va[h] = w.x; Vb[h] = w.y;
      }
      bv16 = a.b2v16[head_of_lane(lane)];
    }
    Rows nxt;
    nxt.r = nxt.e = make_float2(0.f, 0.f);
    if (M > 0) nxt = gather(0, a.vs, a.ld_vs, a.ve, a.ld_ve);
    for (int m = 0; m < M; ++m) {
      float2 pre = make_float2(cv.x + (nxt.r.x + nxt.e.x), cv.y + (nxt.r.y + nxt.e.y));
      asm volatile("" : "+v"(pre.x), "+v"(pre.y) : : "memory");
      if (m + 1 < M) nxt = gather(m + 1, a.vs, a.ld_vs, a.ve, a.ld_ve);
      if (KNN) {
        const int ty = __builtin_amdgcn_readfirstlane(scri[TY + m]);
        const float* tab = Wl + ty * 21 * 128 + 2 * lane;
        const float4* G4 = reinterpret_cast<const float4*>(scr + FEAT + 20 * m);
#pragma unroll
        for (int g4 = 0; g4 < 5; ++g4) {
          const float4 gv = G4[g4];
          float2 t0 = *reinterpret_cast<const float2*>(tab + (4 * g4 + 0) * 128);
          float2 t1 = *reinterpret_cast<const float2*>(tab + (4 * g4 + 1) * 128);
          float2 t2 = *reinterpret_cast<const float2*>(tab + (4 * g4 + 2) * 128);
          float2 t3 = *reinterpret_cast<const float2*>(tab + (4 * g4 + 3) * 128);
          pre.x = fmaf(t0.x, gv.x, pre.x); pre.y = fmaf(t0.y, gv.x, pre.y);
          pre.x = fmaf(t1.x, gv.y, pre.x); pre.y = fmaf(t1.y, gv.y, pre.y);
          pre.x = fmaf(t2.x, gv.z, pre.x); pre.y = fmaf(t2.y, gv.z, pre.y);
          pre.x = fmaf(t3.x, gv.w, pre.x); pre.y = fmaf(t3.y, gv.w, pre.y);
        }
        const float2 tc = *reinterpret_cast<const float2*>(tab + 20 * 128);
        pre.x += tc.x; pre.y += tc.y;
      } else if (TRIP) {
        const float4* C4 = reinterpret_cast<const float4*>(scr + FEAT + 16 * m);
        const float4 c0 = C4[0], c1 = C4[1], c2 = C4[2], c3 = C4[3];
        const float cc[13] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x};
#pragma unroll
        for (int t = 0; t < 13; ++t) { pre.x = fmaf(wa0[t], cc[t], pre.x); pre.y = fmaf(wa1[t], cc[t], pre.y); }
      }
      ln_relu2(pre.x, pre.y, g.x, g.y, be.x, be.y);
      if (!POS) {
        const float4* A4 = reinterpret_cast<const float4*>(scr + SC + 16 * m);
        const float4 a0 = A4[0], a1 = A4[1], a2 = A4[2], a3 = A4[3];
        const float aw[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
        for (int h = 0; h < 16; ++h) { Za[h] = fmaf(aw[h], pre.x, Za[h]); Zb[h] = fmaf(aw[h], pre.y, Zb[h]); }
      } else {
        float p[16];
#pragma unroll
        for (int h = 0; h < 16; ++h) p[h] = fmaf(pre.y, Vb[h], pre.x * Va[h]);
        const float v16 = reduce16(p, lane) + bv16;
        const float coef = scr[SC + 16 * m + head_of_lane(lane)] * v16;
        dxa = fmaf(coef, scr[REL + 4 * m], dxa);
        dya = fmaf(coef, scr[REL + 4 * m + 1], dya);
        dza = fmaf(coef, scr[REL + 4 * m + 2], dza);
      }
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------------
  if (POS) {
    if (active) {
      // each head's coefficient is replicated on 4 lanes: keep one copy, sum the 16 heads, mean over heads
      if ((lane & 3) != 0) { dxa = 0.f; dya = 0.f; dza = 0.f; }
      dxa = wave_sum(dxa) * (1.0f / 16.0f);
      dya = wave_sum(dya) * (1.0f / 16.0f);
      dza = wave_sum(dza) * (1.0f / 16.0f);
      if (lane < 3) {
        const float v = lane == 0 ? dxa : (lane == 1 ? dya : dza);
        if (MODE == M_PE) {
          a.out[(long)seg * 3 + lane] = v;
        } else {
          const long xi = ((long)b * N + node) * 3 + lane;
          a.x_next[xi] = a.x[xi] + a.dxe[(long)seg * 3 + lane] + v;
        }
      }
    }
    return;
  }
  __syncthreads();                           // all waves finished reading the pass-2 table
  stage_weights<WL / 4>(Wl, a.W2vT);
  float ssum = 0.f;
  if (active) {
    wave_lds_sync();
    ssum = scr[SSUM + (lane >> 2)];          // SSUM lies outside the aliased union
    // Z~ -> LDS transposer zt[h][c], pitch 132 (aliases the score / geometry scratch, now dead)
#pragma unroll
    for (int h = 0; h < 16; ++h) *reinterpret_cast<float2*>(scr + h * 132 + 2 * lane) = make_float2(Za[h], Zb[h]);
  }
  __syncthreads();                           // W2v^T staged (block-wide) and zt visible
  if (active) {
    const int hsel = lane >> 2;              // outputs o = 2*lane, 2*lane+1 belong to head o>>3 = lane>>2
    float o0 = 0.f, o1 = 0.f;
    const float* zrow = scr + hsel * 132;
#pragma unroll 4
    for (int c4 = 0; c4 < 32; ++c4) {
      const float4 z = *reinterpret_cast<const float4*>(zrow + 4 * c4);
      const float2 w0 = *reinterpret_cast<const float2*>(&Wl[(4 * c4 + 0) * 128 + 2 * lane]);
      const float2 w1 = *reinterpret_cast<const float2*>(&Wl[(4 * c4 + 1) * 128 + 2 * lane]);
      const float2 w2 = *reinterpret_cast<const float2*>(&Wl[(4 * c4 + 2) * 128 + 2 * lane]);
      const float2 w3 = *reinterpret_cast<const float2*>(&Wl[(4 * c4 + 3) * 128 + 2 * lane]);
      o0 = fmaf(w0.x, z.x, o0); o1 = fmaf(w0.y, z.x, o1);
      o0 = fmaf(w1.x, z.y, o0); o1 = fmaf(w1.y, z.y, o1);
      o0 = fmaf(w2.x, z.z, o0); o1 = fmaf(w2.y, z.z, o1);
      o0 = fmaf(w3.x, z.w, o0); o1 = fmaf(w3.y, z.w, o1);
    }
    const float2 bb = *reinterpret_cast<const float2*>(a.b2v + 2 * lane);
    o0 = fmaf(bb.x, ssum, o0);
    o1 = fmaf(bb.y, ssum, o1);
    float* dst;
    if (MODE == M_NB) dst = a.out + ((long)b * N + node) * 128 + 2 * lane;
    else dst = a.out + (long)seg * 128 + 2 * lane;
    if (MODE == M_NE) {
      *reinterpret_cast<float2*>(dst) = make_float2(o0, o1);
    } else {
      const float2 old = *reinterpret_cast<const float2*>(dst);
      *reinterpret_cast<float2*>(dst) = make_float2(old.x + o0, old.y + o1);
    }
  }
}

template <int MODE, int MMAX>
static int launch_mode(const AttnArgs& a, int nseg, hipStream_t st) {
  if (nseg <= 0) return DD_OK;
  hipLaunchKernelGGL((k_attn<MODE, MMAX>), dim3((nseg + 7) / 8), dim3(512), 0, st, a);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

int launch_attn(int mode, const AttnArgs& a, hipStream_t st) {
  const int N = a.NP + a.NL;
  switch (mode) {
    case M_NE: return launch_mode<M_NE, 32>(a, a.B * N, st);
    case M_NB: return launch_mode<M_NB, 64>(a, a.B * a.NL, st);
    case M_BL: return launch_mode<M_BL, 64>(a, a.B * a.NL * (a.NL - 1), st);
    case M_PE: return launch_mode<M_PE, 32>(a, a.B * a.NL, st);
    case M_PB: return launch_mode<M_PB, 64>(a, a.B * a.NL, st);
  }
  return DD_ERR_BAD_ARG;
}

}  // namespace dd
