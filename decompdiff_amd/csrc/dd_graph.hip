// Graph construction and light per-node / per-edge kernels:
//   k_knn            torch_cluster.knn via torch_geometric.nn.knn_graph (uni_transformer_edge.py:353)
//   k_edge_weights   e_w = sigmoid(MLP_{20->128->1}(G(dist)))           (uni_transformer_edge.py:422-427)
//   k_embed_*        atom / bond embeddings + context composition        (decompdiff.py:219-256,279,296-297)
//   k_bl_assemble3   per-bond-edge partial sums of the bond_layer first Linear (packing.py docstring)
// One wavefront (64 lanes) per centre / edge row; feature rows are 128 floats = 2 per lane,
// so every gather is a single coalesced 512-byte row read.
#include "dd_kernels.hpp"

namespace dd {

// ------------------------------------------------------------------------------------ kNN
// One wave per centre.  Each lane owns candidates c = lane, lane+64, ... (<= 32 per lane,
// N <= 2048).  Key = (bits(d2) << 32) | index is monotone in (d2, index) because d2 >= 0: the K
// smallest keys in ascending order are the neighbours in ascending (distance, index) — the same
// total order the oracle's stable sort uses.  d2 = (dx*dx + dy*dy) + dz*dz with no FMA contraction.
// The per-lane candidate count is a template parameter (ceil(N / 64), not the maximum 16).

// atom n of sample b is real (not padding): protein rows n < NP count up to np_real[b], ligand rows up to nl_real[b]
__device__ __forceinline__ bool atom_is_real(int n, int NP, int npb, int nlb) { return n < NP ? n < npb : (n - NP) < nlb; }

// Positions of one sample, either one [N,3] block (the workspace copy: l = p + 3 NP) or the sampler's separate protein /
// ligand blocks (the fused head launch reads x_t where the step kernel left it: same values, other addresses).
struct PosView {
  const float* p;
  const float* l;
  int NP;
  __device__ __forceinline__ float get(int n, int c) const { return n < NP ? p[3 * n + c] : l[3 * (n - NP) + c]; }
};

// One wave: the K nearest neighbours of centre i, ascending (d2, index), to nbr_row[0..K) (global) and -- when srt is
// given -- to srt[0..K) (this wave's LDS slice: the edge-weight pass of the fused head launch reads them from there).
template <int CAND>
__device__ __forceinline__ void knn_wave(const PosView& pv, int i, int N, int K, int32_t* __restrict__ nbr_row, bool masked, int npb,
                                         int nlb, unsigned long long* sel /*LDS, 64 per wave*/, int32_t* srt /*LDS or nullptr*/) {
  const int lane = threadIdx.x & 63;
  const float cx = pv.get(i, 0), cy = pv.get(i, 1), cz = pv.get(i, 2);
  unsigned khi[CAND], klo[CAND];                         // key = (bits(d2), index): d2 >= 0, so unsigned order = float order
#pragma unroll
  for (int t = 0; t < CAND; ++t) {
    const int c = lane + 64 * t;
    khi[t] = ~0u; klo[t] = ~0u;
    if (c < N && c != i && (!masked || atom_is_real(c, pv.NP, npb, nlb))) {
      float dx = __fsub_rn(cx, pv.get(c, 0)), dy = __fsub_rn(cy, pv.get(c, 1)), dz = __fsub_rn(cz, pv.get(c, 2));
      float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
      khi[t] = __float_as_uint(d2); klo[t] = (unsigned)c;
    }
  }
  // Selection without K serial wave-min rounds:
  //  (1) radix select on the distance bits: the K-th smallest d2 (with multiplicity), one bit per trip from the top;
  //      a trip is CAND ballots + scalar popcounts, no cross-lane dependency chain;
  //  (2) every candidate below that distance is a neighbour, and of those AT it the `need` lowest indices (candidate
  //      c = lane + 64 t, so index order is (t, lane) order: ballot prefix counts);
  //  (3) the K selected keys go to LDS, one per slot, and lane j ranks key j among them: its position in ascending
  //      (d2, index) order -- the order K rounds of wave-min produced (and the oracle's stable sort).
  unsigned prefix = 0;
  int need = K;
  for (int bit = 31; bit >= 0; --bit) {
    int c0 = 0;                                          // keys that agree with the prefix above `bit` and have this bit clear
#pragma unroll
    for (int t = 0; t < CAND; ++t)
      c0 += __builtin_popcountll(__builtin_amdgcn_ballot_w64((khi[t] >> bit) == (prefix >> bit)));
    if (c0 < need) { need -= c0; prefix |= 1u << bit; }
  }
  int base = 0, ties = 0;
#pragma unroll
  for (int t = 0; t < CAND; ++t) {
    const bool eq = khi[t] == prefix;
    const unsigned long long meq = __builtin_amdgcn_ballot_w64(eq);
    const int tie_rank = ties + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(meq >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)meq, 0u));
    const bool take = khi[t] < prefix || (eq && tie_rank < need);
    const unsigned long long mtk = __builtin_amdgcn_ballot_w64(take);
    const int slot = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mtk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mtk, 0u));
    if (take && slot < 64) sel[slot] = ((unsigned long long)khi[t] << 32) | klo[t];
    base += __builtin_popcountll(mtk);
    ties += __builtin_popcountll(meq);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's LDS writes before its reads below (one wave owns the slice)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (lane < K) {
    const unsigned long long mine = sel[lane];
    int rank = 0;
    for (int j = 0; j < K; ++j) rank += sel[j] < mine ? 1 : 0;     // same address for every lane: an LDS broadcast
    const int32_t idx = (int32_t)(unsigned)(mine & 0xffffffffull);
    nbr_row[rank] = idx;
    if (srt) srt[rank] = idx;
  }
  if (srt) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

template <int CAND>
__global__ __launch_bounds__(256) void k_knn(const float* __restrict__ x, int B, int N, int K, int32_t* __restrict__ nbr, int NP,
                                             const int32_t* __restrict__ np_real, const int32_t* __restrict__ nl_real) {
  __shared__ unsigned long long sel[4][64];
  const int lane = threadIdx.x & 63;
  const int centre = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (centre >= B * N) return;
  const int b = centre / N, i = centre % N;
  const bool masked = nl_real != nullptr;
  const int npb = masked && np_real ? np_real[b] : NP, nlb = masked ? nl_real[b] : N - NP;
  if (masked && !atom_is_real(i, NP, npb, nlb)) {          // a padding atom is no centre: its list is never read
    if (lane < K) nbr[(long)centre * K + lane] = 0;
    return;
  }
  const float* xb = x + (long)b * N * 3;
  const PosView pv{xb, xb + 3 * (long)NP, NP};
  knn_wave<CAND>(pv, i, N, K, nbr + (long)centre * K, masked, npb, nlb, sel[threadIdx.x >> 6], nullptr);
}

// ------------------------------------------------------------------------------ edge weights
// One wave per node, 16 neighbours per tile in the attention kernels' k-pass layout
// (lane = (member mm, channel group cg), 32 hidden channels per lane).  hidden = b1 + W1 . G(d) is 5 k-steps of
// v_mfma_f32_16x16x4_f32 with the weights as A operands held in 40 registers; LayerNorm + ReLU and the 128 -> 1
// output layer reduce over the 4 lanes of a member with permlane swaps.
typedef float f32x4e __attribute__((ext_vector_type(4)));
// The 128-channel vectors of the edge-weight MLP (first-layer bias, LayerNorm gamma / beta, output weights) in LDS, once per
// workgroup: kept in registers they cost 128 VGPRs per lane for two uses each, which left the kernel at ONE wave per SIMD.
__device__ __forceinline__ void ew_stage(float* lw /*LDS, 512 floats*/, const float* __restrict__ b1, const float* __restrict__ ln,
                                         const float* __restrict__ w2) {
  const int t = threadIdx.x;                             // 256 threads: 128 float4
  if (t < 32) reinterpret_cast<float4*>(lw)[t] = reinterpret_cast<const float4*>(b1)[t];
  else if (t < 96) reinterpret_cast<float4*>(lw + 128)[t - 32] = reinterpret_cast<const float4*>(ln)[t - 32];   // gamma | beta
  else if (t < 128) reinterpret_cast<float4*>(lw + 384)[t - 96] = reinterpret_cast<const float4*>(w2)[t - 96];
}
// One wave: e_w of the K edges of node i.  nbr_row: the node's neighbour list (global memory, or the LDS copy knn_wave left).
__device__ __forceinline__ void ew_wave(const PosView& pv, int i, int K, const int32_t* nbr_row, const float* __restrict__ W1T,
                                        const float* lw /*LDS: ew_stage*/, const float* __restrict__ b2, float* __restrict__ ew_row) {
  const int lane = threadIdx.x & 63, mm = lane & 15, cg = lane >> 4;
  const float cx = pv.get(i, 0), cy = pv.get(i, 1), cz = pv.get(i, 2);
  float Wa[5][8];                                        // A operands: W1T[4s + cg][16nt + mm]
#pragma unroll
  for (int s = 0; s < 5; ++s)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) Wa[s][nt] = W1T[(4 * s + cg) * 128 + 16 * nt + mm];
  const float bias2 = b2[0];
  auto quad = [](float v) { v = swap16_sum(v, v); return swap32_sum(v, v); };
  for (int t0 = 0; t0 < K; t0 += 16) {
    // (the LDS offset is made opaque per trip: the vectors are loop-invariant, and hoisted out of the loop they are the
    // 128 registers per lane this layout is there to avoid)
    int o4 = 4 * cg;
    asm volatile("" : "+v"(o4));
    auto vec = [&](int which, int nt) { return *reinterpret_cast<const float4*>(lw + which * 128 + 16 * nt + o4); };   // channels 16nt + 4cg .. +3
    const int m = t0 + mm;
    const int j = nbr_row[m < K ? m : K - 1];
    const float dx = cx - pv.get(j, 0), dy = cy - pv.get(j, 1), dz = cz - pv.get(j, 2);
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    f32x4e acc[8];
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) { const float4 bb = vec(0, nt); acc[nt] = f32x4e{bb.x, bb.y, bb.z, bb.w}; }
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      const float fk = gauss_feat(d, 4 * s + cg);
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wa[s][nt], fk, acc[nt], 0, 0, 0);
    }
    float sum = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) sum += (acc[nt][0] + acc[nt][1]) + (acc[nt][2] + acc[nt][3]);
    const float mean = quad(sum) * (1.0f / 128.0f);
    float var = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { acc[nt][r] -= mean; var = fmaf(acc[nt][r], acc[nt][r], var); }
    const float rstd = dd_rsqrt(quad(var) * (1.0f / 128.0f) + 1e-5f);
    float dot = 0.f;
#pragma unroll
    for (int nt = 0; nt < 8; ++nt) {
      const float4 gm = vec(1, nt), bt = vec(2, nt), ww = vec(3, nt);
      dot = fmaf(fmaxf(fmaf(acc[nt][0] * rstd, gm.x, bt.x), 0.f), ww.x, dot);
      dot = fmaf(fmaxf(fmaf(acc[nt][1] * rstd, gm.y, bt.y), 0.f), ww.y, dot);
      dot = fmaf(fmaxf(fmaf(acc[nt][2] * rstd, gm.z, bt.z), 0.f), ww.z, dot);
      dot = fmaf(fmaxf(fmaf(acc[nt][3] * rstd, gm.w, bt.w), 0.f), ww.w, dot);
    }
    const float logit = quad(dot) + bias2;
    if (cg == 0 && m < K) ew_row[m] = 1.0f / (1.0f + expf(-logit));
  }
}
__global__ __launch_bounds__(256, 3) void k_edge_weights2(const float* __restrict__ x, const int32_t* __restrict__ nbr, int B,
                                                          int N, int K, const float* __restrict__ W1T,
                                                          const float* __restrict__ b1, const float* __restrict__ ln,
                                                          const float* __restrict__ w2, const float* __restrict__ b2,
                                                          float* __restrict__ ew, int NP, const int32_t* __restrict__ np_real,
                                                          const int32_t* __restrict__ nl_real) {
  __shared__ __attribute__((aligned(16))) float lw[512];
  ew_stage(lw, b1, ln, w2);
  __syncthreads();                                       // (before any wave leaves)
  const int node = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (node >= B * N) return;
  const int b = node / N, i = node % N;
  if (nl_real != nullptr && !atom_is_real(i, NP, np_real ? np_real[b] : NP, nl_real[b])) return;   // padding atom
  const float* xb = x + (long)b * N * 3;
  const PosView pv{xb, xb + 3 * (long)NP, NP};
  ew_wave(pv, i, K, nbr + (long)node * K, W1T, lw, b2, ew + (long)node * K);
}

// -------------------------------------------------------------------------------- embeddings
// protein_h[r, c] = sum_f W[c, f] * feat[r, f] + b[c]     (W padded to 128 rows: row 127 = 0, b[127] = 0)
__global__ void k_embed_protein(const float* __restrict__ feat, int rows, const float* __restrict__ W,
                                const float* __restrict__ b, float* __restrict__ out) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * 128) return;
  int r = idx >> 7, c = idx & 127;
  float acc = 0.f;
  const float* f = feat + (long)r * 29;
  const float* w = W + c * 29;
#pragma unroll
  for (int k = 0; k < 29; ++k) acc = fmaf(w[k], f[k], acc);
  out[idx] = acc + b[c];
}

// One launch at the head of a step: node embedding / context, bond embedding, and the per-forward work counters
// of the persistent kernels (64 ints) set to zero.
__device__ __forceinline__ void embed_block(const unsigned blk, const float* __restrict__ protein_h, const float* __restrict__ protein_pos,
                                                   const float* __restrict__ lig_pos, const int32_t* __restrict__ lig_v,
                                                   const float* __restrict__ lig_aux, const float* __restrict__ Wl,
                                                   const float* __restrict__ bl, int B, int NP, int NL, float* __restrict__ h,
                                                   float* __restrict__ xa, float* __restrict__ xb,
                                                   const int32_t* __restrict__ bond, long bond_rows,
                                                   const float* __restrict__ Wb, const float* __restrict__ bb,
                                                   float* __restrict__ hb, int32_t* __restrict__ counters, int node_blocks,
                                                   int32_t* __restrict__ advance, int nv) {
  // (measurement build: + the tile-queue schedule's flag words; the error word behind them is sticky: dd_queue_error reads
  // and clears it.  The default library has no such schedule and zeroes the work counters only.)
#if defined(DD_DEBUG_OPTIONS) && DD_DEBUG_OPTIONS
  constexpr int n_zero = DD_NUM_COUNTERS + DD_FLAG_ERR;
#else
  constexpr int n_zero = DD_NUM_COUNTERS;
#endif
  if (blk == 0 && counters)
    for (int i = threadIdx.x; i < n_zero; i += 256) counters[i] = 0;
  // the step index moves on with the first launch of a step's forward (nothing before the step kernels reads it): the
  // step kernels use *advance - 1, and no one-thread launch sits at the end of a step
  if (blk == 0 && threadIdx.x == 64 && advance) *advance += 1;
  if ((int)blk < node_blocks) {
    const int N = NP + NL;
    const long idx = (long)blk * 256 + threadIdx.x;
    if (idx >= (long)B * N * 128) return;
    const int c = idx & 127;
    const long node = idx >> 7;
    const int b = node / N, n = node % N;
    float val;
    if (n < NP) {
      val = protein_h[((long)b * NP + n) * 128 + c];
    } else {
      const long a = (long)b * NL + (n - NP);
      const float* w = Wl + c * (nv + 2);                // ligand_atom_emb row: nv class columns + 2 arm / scaffold indicators
      val = w[lig_v[a]] + w[nv] * lig_aux[2 * a] + w[nv + 1] * lig_aux[2 * a + 1] + bl[c];
    }
    h[idx] = val;
    if (c < 3) {
      const float p = n < NP ? protein_pos[((long)b * NP + n) * 3 + c] : lig_pos[((long)b * NL + (n - NP)) * 3 + c];
      xa[node * 3 + c] = p;
      xb[node * 3 + c] = p;
    }
  } else {
    const long idx = (long)(blk - node_blocks) * 256 + threadIdx.x;
    if (idx >= bond_rows * 128) return;
    const int c = idx & 127;
    const long e = idx >> 7;
    hb[idx] = Wb[c * 5 + bond[e]] + bb[c];
  }
}
__global__ __launch_bounds__(256) void k_embed_all(const float* __restrict__ protein_h, const float* __restrict__ protein_pos,
                                                   const float* __restrict__ lig_pos, const int32_t* __restrict__ lig_v,
                                                   const float* __restrict__ lig_aux, const float* __restrict__ Wl,
                                                   const float* __restrict__ bl, int B, int NP, int NL, float* __restrict__ h,
                                                   float* __restrict__ xa, float* __restrict__ xb,
                                                   const int32_t* __restrict__ bond, long bond_rows,
                                                   const float* __restrict__ Wb, const float* __restrict__ bb,
                                                   float* __restrict__ hb, int32_t* __restrict__ counters, int node_blocks,
                                                   int32_t* __restrict__ advance, int nv) {
  embed_block(blockIdx.x, protein_h, protein_pos, lig_pos, lig_v, lig_aux, Wl, bl, B, NP, NL, h, xa, xb, bond, bond_rows, Wb, bb, hb,
              counters, node_blocks, advance, nv);
}

// --------------------------------------------------------------------------- layer-0 rows from tables
// One thread per float4 of an output row.  Ligand atom a = b*NL + i with combination c = 8*(arm flag) + class:
//   l0_P[b*N + NP + i] = TN[c] (640), PL[a] = TL[c] (1280), l0_qn[b*N + NP + i] = TQN[c], qlnb[a] = TQL[c];
// bond e = b*Eb + t*(NL-1) + s' of type ty with destination atom t:  PB[e] = TB[ty] (640), qb[e] = TQB[c(t)][ty].
constexpr int L0_TN = 0, L0_TL = 16 * 640, L0_TB = L0_TL + 16 * 1280, L0_TQN = L0_TB + 5 * 640, L0_TQL = L0_TQN + 16 * 128,
              L0_TQB = L0_TQL + 16 * 128;
__device__ __forceinline__ void layer0_rows_block(const unsigned blk, const float* __restrict__ tab, const int32_t* __restrict__ lig_v,
                                                     const float* __restrict__ lig_aux, const int32_t* __restrict__ bond, int B,
                                                     int NP, int NL, float* __restrict__ l0_P, float* __restrict__ PL,
                                                     float* __restrict__ l0_qn, float* __restrict__ qlnb, float* __restrict__ PB,
                                                     float* __restrict__ qb, long atom_f4) {
  const long idx = (long)blk * 256 + threadIdx.x;
  const int N = NP + NL, Eb = NL * (NL - 1);
  auto combo = [&](long a) { return (lig_aux[2 * a + 1] > 0.5f ? 8 : 0) + lig_v[a]; };
  const float4* t4 = reinterpret_cast<const float4*>(tab);
  if (idx < atom_f4) {
    constexpr int PER = 160 + 320 + 32 + 32;             // float4 per atom: 640 + 1280 + 128 + 128 floats
    const long a = idx / PER;
    const int k = (int)(idx % PER);
    const int b = (int)(a / NL), i = (int)(a % NL);
    const int c = combo(a);
    const long node = (long)b * N + NP + i;
    if (k < 160) reinterpret_cast<float4*>(l0_P + node * 640)[k] = t4[(L0_TN + c * 640) / 4 + k];
    else if (k < 480) reinterpret_cast<float4*>(PL + a * 1280)[k - 160] = t4[(L0_TL + c * 1280) / 4 + (k - 160)];
    else if (k < 512) reinterpret_cast<float4*>(l0_qn + node * 128)[k - 480] = t4[(L0_TQN + c * 128) / 4 + (k - 480)];
    else reinterpret_cast<float4*>(qlnb + a * 128)[k - 512] = t4[(L0_TQL + c * 128) / 4 + (k - 512)];
    return;
  }
  constexpr int PERB = 160 + 32;                         // float4 per bond: 640 + 128 floats
  const long j = idx - atom_f4;
  const long e = j / PERB;
  if (e >= (long)B * Eb) return;
  const int k = (int)(j % PERB);
  const int ty = bond[e];
  if (k < 160) {
    reinterpret_cast<float4*>(PB + e * 640)[k] = t4[(L0_TB + ty * 640) / 4 + k];
  } else {
    const int b = (int)(e / Eb), t = (int)((e % Eb) / (NL - 1));
    const int c = combo((long)b * NL + t);
    reinterpret_cast<float4*>(qb + e * 128)[k - 160] = t4[(L0_TQB + (c * 5 + ty) * 128) / 4 + (k - 160)];
  }
}
__global__ __launch_bounds__(256) void k_layer0_rows(const float* __restrict__ tab, const int32_t* __restrict__ lig_v,
                                                     const float* __restrict__ lig_aux, const int32_t* __restrict__ bond, int B,
                                                     int NP, int NL, float* __restrict__ l0_P, float* __restrict__ PL,
                                                     float* __restrict__ l0_qn, float* __restrict__ qlnb, float* __restrict__ PB,
                                                     float* __restrict__ qb, long atom_f4) {
  layer0_rows_block(blockIdx.x, tab, lig_v, lig_aux, bond, B, NP, NL, l0_P, PL, l0_qn, qlnb, PB, qb, atom_f4);
}

// ------------------------------------------------------------------------------ head of a forward: two launches
// Everything the first attention layer waits for that only depends on the state the previous reverse step left:
// k_head_graph -- the kNN graph + edge weights, one wave per centre (the bodies of k_knn and k_edge_weights2, the list
// handed over through LDS; x_t is read from the sampler's protein / ligand position buffers, so the launch does not
// wait for the workspace copy the embedding blocks write); k_head_rows -- the embeddings / context (with the zeroed work
// counters and the step index) and the layer-0 rows (the bodies of k_embed_all and k_layer0_rows as two block ranges).
// Outputs bit-identical to the four separate launches.
struct HeadArgs {
  // graph
  int B, NP, NL, K;
  const float *protein_pos, *lig_pos;
  int32_t* nbr;
  float* ew;
  const float *EW_W1T, *EW_b1, *EW_ln, *EW_w2, *EW_b2;
  const int32_t *np_real, *nl_real;
  // embeddings
  const float* protein_h;
  const int32_t* lig_v;
  const float* lig_aux;
  const float *Wl, *bl;
  float *h, *xa, *xb;
  const int32_t* bond;
  long bond_rows;
  const float *Wb, *bb;
  float* hb;
  int32_t *counters, *advance;
  int nv;                   // atom classes (columns of the ligand embedding before the two indicators)
  // layer-0 rows (tab == nullptr: none)
  const float* tab;
  float *l0_P, *PL, *l0_qn, *qlnb, *PB, *qb;
  long atom_f4;
  int n_graph, n_embed_nodes, n_embed;
};
template <int CAND>
__global__ __launch_bounds__(256, 3) void k_head_graph(const HeadArgs a) {
  __shared__ unsigned long long sel[4][64];
  __shared__ int32_t srt[4][32];
  __shared__ __attribute__((aligned(16))) float lw[512];
  ew_stage(lw, a.EW_b1, a.EW_ln, a.EW_w2);
  __syncthreads();                                       // (before any wave leaves)
  const int N = a.NP + a.NL, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int centre = (int)blockIdx.x * 4 + w;
  if (centre >= a.B * N) return;
  const int b = centre / N, i = centre % N;
  const bool masked = a.nl_real != nullptr;
  const int npb = masked && a.np_real ? a.np_real[b] : a.NP, nlb = masked ? a.nl_real[b] : a.NL;
  if (masked && !atom_is_real(i, a.NP, npb, nlb)) {        // a padding atom is no centre: its list is never read
    if (lane < a.K) a.nbr[(long)centre * a.K + lane] = 0;
    return;
  }
  const PosView pv{a.protein_pos + (long)b * a.NP * 3, a.lig_pos + (long)b * a.NL * 3, a.NP};
  knn_wave<CAND>(pv, i, N, a.K, a.nbr + (long)centre * a.K, masked, npb, nlb, sel[w], srt[w]);
  ew_wave(pv, i, a.K, srt[w], a.EW_W1T, lw, a.EW_b2, a.ew + (long)centre * a.K);
}
// (a kernel of its own: the copy blocks need many waves per SIMD, the graph blocks ~200 registers per lane)
__global__ __launch_bounds__(256) void k_head_rows(const HeadArgs a) {
  const unsigned blk = blockIdx.x;
  if ((int)blk < a.n_embed) {
    embed_block(blk, a.protein_h, a.protein_pos, a.lig_pos, a.lig_v, a.lig_aux, a.Wl, a.bl, a.B, a.NP, a.NL, a.h, a.xa, a.xb, a.bond,
                a.bond_rows, a.Wb, a.bb, a.hb, a.counters, a.n_embed_nodes, a.advance, a.nv);
    return;
  }
  layer0_rows_block(blk - a.n_embed, a.tab, a.lig_v, a.lig_aux, a.bond, a.B, a.NP, a.NL, a.l0_P, a.PL, a.l0_qn, a.qlnb, a.PB, a.qb,
                    a.atom_f4);
}

// --------------------------------------------------------------------------- bond-layer assemble
// For bond edge e = (src s -> dst t) of sample b (dst-major id e = t*(NL-1) + s'):
//   Ek[e] = PB.k_hb[e] + Wg1k . G(d_e) + PL[s].k_hk + PL[t].k_hj        (b1 folded into PB bias)
//   Ev[e] = same with the v weights
//   q1[e] = PB.q_hb[e] + PL[t].q_hi
// PB row layout [640]: NB ke | NB ve | BL k_hb | BL v_hb | BL q_hb;  PL row layout [1280]: see packing.py.
//   Rk[e] = Wg2k . G(d_e),  Rv[e] = Wg2v . G(d_e)   (the G(d_ji) columns, constant over a triplet segment)
// 16 bonds per wave tile, lane = (bond mm, channel group cg), 32 channels per lane as in
// the attention kernel's k-pass layout.  The four Gaussian contractions  out^T[c][bond] += sum_g W[g][c] G[bond][g]
// (20 Gaussians = 5 k-steps of v_mfma_f32_16x16x4_f32) take the tables as A operands from a channel-permuted LDS image
// (packing.py BL_Wgp, 40 KB) and the per-lane Gaussians as B; the gathered projection rows enter as the accumulator.
typedef float f32x4a __attribute__((ext_vector_type(4)));
// (two waves per SIMD asked for: left alone, the compiler takes 216 VGPRs + 60 AGPRs = one wave per SIMD, i.e. ONE workgroup per
// CU and 2.1 rounds for the 545 workgroups of a C-small batch; -1.0 % step time, no spills, bit-identical)
__global__ __launch_bounds__(256, 2) void k_bl_assemble3(const float* __restrict__ x, const float* __restrict__ PB,
                                                      const float* __restrict__ PL, const float* __restrict__ Wgp /*[4,20,128]*/,
                                                      int B, int NP, int NL, float* __restrict__ Ek, float* __restrict__ Ev,
                                                      float* __restrict__ q1, float* __restrict__ Rk, float* __restrict__ Rv,
                                                      int blocks_per_output, const float* __restrict__ xprev,
                                                      const float* __restrict__ dxe, const float* __restrict__ dxb,
                                                      float* __restrict__ xout, const FlagWait fw) {
  // Deferred coordinate update (xprev != nullptr): the previous layer's x += dx_edge + dx_bond (uni_transformer_edge.py:285)
  // has not been applied yet -- every lane forms its two ligand positions from xprev + deltas (same association as
  // k_xupdate, so bit-identical) and workgroup 0 writes the updated rows for the kernels that follow.
  if (xprev != nullptr && blockIdx.x == 0) {
    const int n = B * NL * 3;
    for (int idx = threadIdx.x; idx < n; idx += 256) {
      const int bb = idx / (NL * 3), r = idx % (NL * 3);
      const long xi = ((long)bb * (NP + NL) + NP) * 3 + r;
      xout[xi] = xprev[xi] + dxe[idx] + dxb[idx];
    }
  }
  // workgroup = (output q: Ek, Ev, Rk, Rv, q1; four consecutive 16-bond tiles, one per wave): 5x the parallelism of a
  // tile-per-wave split over all outputs (only 435 tiles exist at B=8), and only that output's 10 KB table is staged
  __shared__ __attribute__((aligned(16))) float tab[DD_NGAUSS * 128];
  const int q = blockIdx.x / blocks_per_output;
  if (q < 4) {
    const float4* src = reinterpret_cast<const float4*>(Wgp + q * DD_NGAUSS * 128);
    for (int i = threadIdx.x; i < DD_NGAUSS * 32; i += 256) reinterpret_cast<float4*>(tab)[i] = src[i];
  }
  // PB / PL of this layer come from the layer-tail queue on the other stream: no graph edge, the counters are polled here
  // (the table above is staged meanwhile); ends with the barrier that also publishes the table
  if (fw.flags != nullptr) dd_wait_flags(fw.flags, fw.idx0, fw.n0, fw.idx1, fw.n1, DD_FLAG_ERR, 300);
  else __syncthreads();
  const int lane = threadIdx.x & 63, mm = lane & 15, cg = lane >> 4;
  const int Eb = NL * (NL - 1), N = NP + NL;
  const long nrows = (long)B * Eb;
  const long tile = (long)(blockIdx.x % blocks_per_output) * 4 + (threadIdx.x >> 6);
  if (tile * 16 >= nrows) return;
  const long e_raw = tile * 16 + mm;
  const bool valid = e_raw < nrows;
  const long e_glob = valid ? e_raw : nrows - 1;
  const int b = e_glob / Eb, e = e_glob % Eb;
  const int t = e / (NL - 1), sp = e % (NL - 1);
  const int s = sp + (sp >= t ? 1 : 0);
  const float* pb = PB + e_glob * 640 + 4 * cg;
  const float* ps = PL + ((long)b * NL + s) * 1280 + 4 * cg;
  const float* pt = PL + ((long)b * NL + t) * 1280 + 4 * cg;
  auto ld4 = [](const float* p) { return *reinterpret_cast<const float4*>(p); };
  if (q == 4) {                                        // q1 = q_hb[e] + q_hi[t]
    if (q1 != nullptr && valid) {
#pragma unroll
      for (int nt = 0; nt < 8; ++nt) {
        const float4 a0 = ld4(pb + 512 + 16 * nt), a1 = ld4(pt + 1152 + 16 * nt);
        *reinterpret_cast<float4*>(q1 + e_glob * 128 + 4 * cg + 16 * nt) = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
      }
    }
    return;
  }
  f32x4a acc[8];
#pragma unroll
  for (int nt = 0; nt < 8; ++nt) {
    if (q < 2) {
      const float4 a0 = ld4(pb + 256 + 128 * q + 16 * nt);          // k_hb | v_hb
      const float4 a1 = ld4(ps + 640 + 256 * q + 16 * nt);          // k_hk | v_hk
      const float4 a2 = ld4(pt + 768 + 256 * q + 16 * nt);          // k_hj | v_hj
      acc[nt] = f32x4a{(a0.x + a1.x) + a2.x, (a0.y + a1.y) + a2.y, (a0.z + a1.z) + a2.z, (a0.w + a1.w) + a2.w};
    } else {
      acc[nt] = f32x4a{0.f, 0.f, 0.f, 0.f};
    }
  }
  float dx, dy, dz;
  if (xprev != nullptr) {
    const float* xl = xprev + ((long)b * N + NP) * 3;
    const float* de = dxe + (long)b * NL * 3;
    const float* db = dxb + (long)b * NL * 3;
    dx = (xl[3 * t] + de[3 * t] + db[3 * t]) - (xl[3 * s] + de[3 * s] + db[3 * s]);
    dy = (xl[3 * t + 1] + de[3 * t + 1] + db[3 * t + 1]) - (xl[3 * s + 1] + de[3 * s + 1] + db[3 * s + 1]);
    dz = (xl[3 * t + 2] + de[3 * t + 2] + db[3 * t + 2]) - (xl[3 * s + 2] + de[3 * s + 2] + db[3 * s + 2]);
  } else {
    const float* xl = x + ((long)b * N + NP) * 3;
    dx = xl[3 * t] - xl[3 * s]; dy = xl[3 * t + 1] - xl[3 * s + 1]; dz = xl[3 * t + 2] - xl[3 * s + 2];
  }
  const float d = sqrtf(dx * dx + dy * dy + dz * dz);
  const float* tb = tab + cg * 128 + mm * 4;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const float fk = gauss_feat(d, 4 * k + cg);
    const float4 w0 = *reinterpret_cast<const float4*>(tb + k * 512);
    const float4 w1 = *reinterpret_cast<const float4*>(tb + k * 512 + 64);
    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, fk, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, fk, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, fk, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, fk, acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, fk, acc[4], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, fk, acc[5], 0, 0, 0);
    acc[6] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, fk, acc[6], 0, 0, 0);
    acc[7] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, fk, acc[7], 0, 0, 0);
  }
  float* dst = (q == 0 ? Ek : (q == 1 ? Ev : (q == 2 ? Rk : Rv))) + e_glob * 128 + 4 * cg;
  if (valid) {
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
      *reinterpret_cast<float4*>(dst + 16 * nt) = make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]);
  }
}

// x0-hat: ligand rows of the final coordinates
__global__ void k_extract_ligand(const float* __restrict__ x, int B, int NP, int NL, float* __restrict__ out) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * NL * 3) return;
  int b = idx / (NL * 3), r = idx % (NL * 3);
  out[idx] = x[((long)b * (NP + NL) + NP) * 3 + r];
}
int launch_extract_ligand(const float* x, int B, int NP, int NL, float* out, hipStream_t st) {
  int n = B * NL * 3;
  hipLaunchKernelGGL(k_extract_ligand, dim3((n + 255) / 256), dim3(256), 0, st, x, B, NP, NL, out);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

// x_next[ligand rows] = x + dx_edge + dx_bond   (x = x + delta_x * mask_ligand, uni_transformer_edge.py:285)
__global__ void k_xupdate(const float* __restrict__ x, const float* __restrict__ dxe, const float* __restrict__ dxb,
                          int B, int NP, int NL, float* __restrict__ x_next) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * NL * 3) return;
  int b = idx / (NL * 3), r = idx % (NL * 3);
  long xi = ((long)b * (NP + NL) + NP) * 3 + r;
  x_next[xi] = x[xi] + dxe[idx] + dxb[idx];
}
int launch_xupdate(const float* x, const float* dxe, const float* dxb, int B, int NP, int NL, float* x_next, hipStream_t st) {
  int n = B * NL * 3;
  hipLaunchKernelGGL(k_xupdate, dim3((n + 255) / 256), dim3(256), 0, st, x, dxe, dxb, B, NP, NL, x_next);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

int launch_knn(const float* x, int B, int N, int K, int32_t* nbr, hipStream_t st, int NP, const int32_t* np_real,
               const int32_t* nl_real) {
  const dim3 grid((B * N + 3) / 4), block(256);
  const int cand = (N + 63) / 64;
  if (NP < 0) { NP = N; np_real = nl_real = nullptr; }
  if (cand <= 2) hipLaunchKernelGGL(k_knn<2>, grid, block, 0, st, x, B, N, K, nbr, NP, np_real, nl_real);
  else if (cand <= 4) hipLaunchKernelGGL(k_knn<4>, grid, block, 0, st, x, B, N, K, nbr, NP, np_real, nl_real);
  else if (cand <= 6) hipLaunchKernelGGL(k_knn<6>, grid, block, 0, st, x, B, N, K, nbr, NP, np_real, nl_real);
  else if (cand <= 11) hipLaunchKernelGGL(k_knn<11>, grid, block, 0, st, x, B, N, K, nbr, NP, np_real, nl_real);
  else if (cand <= 16) hipLaunchKernelGGL(k_knn<16>, grid, block, 0, st, x, B, N, K, nbr, NP, np_real, nl_real);
  else hipLaunchKernelGGL(k_knn<DD_N_MAX / 64>, grid, block, 0, st, x, B, N, K, nbr, NP, np_real, nl_real);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_edge_weights(const float* x, const int32_t* nbr, int B, int N, int K, const float* W1T, const float* b1,
                        const float* ln, const float* w2, const float* b2, float* ew, hipStream_t st, int NP, const int32_t* np_real,
                        const int32_t* nl_real) {
  if (NP < 0) { NP = N; np_real = nl_real = nullptr; }
  hipLaunchKernelGGL(k_edge_weights2, dim3((B * N + 3) / 4), dim3(256), 0, st, x, nbr, B, N, K, W1T, b1, ln, w2, b2, ew, NP, np_real, nl_real);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_embed_all(const float* protein_h, const float* protein_pos, const float* lig_pos, const int32_t* lig_v,
                     const float* lig_aux, const float* Wl, const float* bl, int B, int NP, int NL, float* h, float* xa, float* xb,
                     const int32_t* bond, long bond_rows, const float* Wb, const float* bb, float* hb, int32_t* counters,
                     hipStream_t st, int32_t* advance, int nv) {
  const long nn = (long)B * (NP + NL) * 128, nb = bond_rows * 128;
  const int node_blocks = (int)((nn + 255) / 256), bond_blocks = (int)((nb + 255) / 256);
  hipLaunchKernelGGL(k_embed_all, dim3(node_blocks + bond_blocks), dim3(256), 0, st, protein_h, protein_pos, lig_pos, lig_v, lig_aux,
                     Wl, bl, B, NP, NL, h, xa, xb, bond, bond_rows, Wb, bb, hb, counters, node_blocks, advance, nv);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_head_all(const float* protein_h, const float* protein_pos, const float* lig_pos, const int32_t* lig_v,
                    const float* lig_aux, const float* Wl, const float* bl, int B, int NP, int NL, int K, float* h, float* xa, float* xb,
                    const int32_t* bond, long bond_rows, const float* Wb, const float* bb, float* hb, int32_t* counters,
                    int32_t* advance, int32_t* nbr, float* ew, const float* EW_W1T, const float* EW_b1, const float* EW_ln,
                    const float* EW_w2, const float* EW_b2, const int32_t* np_real, const int32_t* nl_real, const float* l0_tables,
                    float* l0_P, float* PL, float* l0_qn, float* qlnb, float* PB, float* qb, hipStream_t st, int parts, int nv) {
  if (K > 32 || (parts & 3) == 0 || (l0_tables != nullptr && nv != DD_NUM_V)) return DD_ERR_UNSUPPORTED_SHAPE;
  const int N = NP + NL;
  HeadArgs a;
  a.B = B; a.NP = NP; a.NL = NL; a.K = K;
  a.protein_pos = protein_pos; a.lig_pos = lig_pos; a.nbr = nbr; a.ew = ew;
  a.EW_W1T = EW_W1T; a.EW_b1 = EW_b1; a.EW_ln = EW_ln; a.EW_w2 = EW_w2; a.EW_b2 = EW_b2;
  a.np_real = np_real; a.nl_real = nl_real;
  a.protein_h = protein_h; a.lig_v = lig_v; a.lig_aux = lig_aux; a.Wl = Wl; a.bl = bl; a.h = h; a.xa = xa; a.xb = xb;
  a.bond = bond; a.bond_rows = bond_rows; a.Wb = Wb; a.bb = bb; a.hb = hb; a.counters = counters; a.advance = advance;
  a.nv = nv;
  a.tab = l0_tables; a.l0_P = l0_P; a.PL = PL; a.l0_qn = l0_qn; a.qlnb = qlnb; a.PB = PB; a.qb = qb;
  const long nn = (long)B * N * 128, nb = bond_rows * 128;
  a.n_embed_nodes = (int)((nn + 255) / 256);
  a.n_embed = a.n_embed_nodes + (int)((nb + 255) / 256);
  a.n_graph = (B * N + 3) / 4;
  a.atom_f4 = (long)B * NL * (160 + 320 + 32 + 32);
  const long bond_f4 = (long)B * NL * (NL - 1) * (160 + 32);
  const int n_l0 = l0_tables ? (int)((a.atom_f4 + bond_f4 + 255) / 256) : 0;
  const dim3 block(256);
  if (parts & 2) hipLaunchKernelGGL(k_head_rows, dim3((unsigned)(a.n_embed + n_l0)), block, 0, st, a);   // parts: 2 = embedding + layer-0 rows
  if (parts & 1) {                                                                                        //        1 = graph
    const dim3 grid((unsigned)a.n_graph);
    const int cand = (N + 63) / 64;
    if (cand <= 2) hipLaunchKernelGGL(k_head_graph<2>, grid, block, 0, st, a);
    else if (cand <= 4) hipLaunchKernelGGL(k_head_graph<4>, grid, block, 0, st, a);
    else if (cand <= 6) hipLaunchKernelGGL(k_head_graph<6>, grid, block, 0, st, a);
    else if (cand <= 11) hipLaunchKernelGGL(k_head_graph<11>, grid, block, 0, st, a);
    else if (cand <= 16) hipLaunchKernelGGL(k_head_graph<16>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(k_head_graph<DD_N_MAX / 64>, grid, block, 0, st, a);
  }
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_layer0_rows(const float* tables, const int32_t* lig_v, const float* lig_aux, const int32_t* bond, int B, int NP, int NL,
                       float* l0_P, float* PL, float* l0_qn, float* qlnb, float* PB, float* qb, hipStream_t st) {
  const long atom_f4 = (long)B * NL * (160 + 320 + 32 + 32), bond_f4 = (long)B * NL * (NL - 1) * (160 + 32);
  hipLaunchKernelGGL(k_layer0_rows, dim3((unsigned)((atom_f4 + bond_f4 + 255) / 256)), dim3(256), 0, st, tables, lig_v, lig_aux, bond,
                     B, NP, NL, l0_P, PL, l0_qn, qlnb, PB, qb, atom_f4);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_bl_assemble(const float* x, const float* PB, const float* PL, const float* Wgp, int B, int NP, int NL, float* Ek, float* Ev,
                       float* q1, float* Rk, float* Rv, hipStream_t st, const float* xprev, const float* dxe, const float* dxb,
                       float* xout, FlagWait fw) {
  if (Wgp == nullptr) return DD_ERR_BAD_ARG;
  const long rows = (long)B * NL * (NL - 1);
  const long tiles = (rows + 15) / 16;
  const int bpo = (int)((tiles + 3) / 4);                            // blocks per output (4 tiles each)
  hipLaunchKernelGGL(k_bl_assemble3, dim3((unsigned)(bpo * (q1 ? 5 : 4))), dim3(256), 0, st, x, PB, PL, Wgp, B, NP, NL, Ek, Ev, q1,
                     Rk, Rv, bpo, xprev, dxe, dxb, xout, fw);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

}  // namespace dd

extern "C" int dd_knn(const float* x, int B, int N, int K, int32_t* nbr, void* stream) {
  if (!x || !nbr || B <= 0 || N <= 1) return DD_ERR_BAD_ARG;
  if (N > DD_N_MAX || K > DD_KNN_MAX || K > N - 1 || K <= 0) return DD_ERR_UNSUPPORTED_SHAPE;
  return dd::launch_knn(x, B, N, K, nbr, (hipStream_t)stream);
}

// ... for a padded heterogeneous batch (the sampler's layout: NP + NL rows per sample, the real atoms are the first np_real[b]
// protein rows and the first nl_real[b] ligand rows): padding atoms are neither centres nor candidates.
extern "C" int dd_knn_masked(const float* x, int B, int NP, int NL, int K, const int32_t* np_real, const int32_t* nl_real, int32_t* nbr,
                             void* stream) {
  const int N = NP + NL;
  if (!x || !nbr || !nl_real || B <= 0 || NP < 0 || NL <= 0 || K <= 0 || K > DD_KNN_MAX || K > N - 1 || N > DD_N_MAX) return DD_ERR_BAD_ARG;
  return dd::launch_knn(x, B, N, K, nbr, (hipStream_t)stream, NP, np_real, nl_real);
}

extern "C" int dd_edge_weights(const float* x, const int32_t* nbr, int B, int N, int K, const float* W1T,
                               const float* b1, const float* ln, const float* w2, const float* b2, float* ew,
                               void* stream) {
  if (!x || !nbr || !W1T || !b1 || !ln || !w2 || !b2 || !ew) return DD_ERR_BAD_ARG;
  return dd::launch_edge_weights(x, nbr, B, N, K, W1T, b1, ln, w2, b2, ew, (hipStream_t)stream);
}

extern "C" int dd_embed_protein(const float* protein_v, int rows, const float* W, const float* b, float* protein_h,
                                void* stream) {
  if (!protein_v || !W || !b || !protein_h || rows <= 0) return DD_ERR_BAD_ARG;
  long n = (long)rows * 128;
  hipLaunchKernelGGL(dd::k_embed_protein, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     protein_v, rows, W, b, protein_h);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
