// Kernel argument structs and launcher prototypes shared by the translation units.
#pragma once
#include "dd_common.hpp"

namespace dd {

struct FlagWait { const int32_t* flags; int idx0, n0, idx1, n1; };       // up to two counters (idx < 0: none)
inline FlagWait no_wait() { return FlagWait{nullptr, -1, 0, -1, 0}; }

struct GemmArgs {
  const float* X; int x_rows_per_b; long x_stride_b; int ldx; int rows;
  const float* W; const float* bias; const float* ln;
  float* Y; int y_rows_per_b; long y_stride_b; int ldy; int ncols; int accumulate;
  // optional second addend for X rows that are ligand nodes: X[b*N + n] += X2[b*NL + n - NP] (n >= NP)
  const float* X2; int x2_N, x2_NP;
  // x2_Eb > 0 selects the other addend form: X row r is a bond edge, X2 row = its destination atom
  //   X[r] += X2[((r / x2_Eb) * x2_N + (r % x2_Eb) / x2_NLm1) * x2_ld]      (x2_N = ligand atoms per sample here)
  int x2_Eb, x2_NLm1, x2_ld;
  long long* dbg;          // profiling aid: s_memtime phase stamps, 8 per workgroup (single launches only)
  const float* acc_src;    // accumulate = 1: the addend is read from acc_src (same layout as Y) instead of Y itself (NULL: Y)
};
inline GemmArgs gemm_args(const float* X, int x_rows_per_b, long x_stride_b, int ldx, int rows, const float* W,
                          const float* bias, const float* ln, float* Y, int y_rows_per_b, long y_stride_b, int ldy,
                          int ncols, int accumulate) {
  GemmArgs g{X, x_rows_per_b, x_stride_b, ldx, rows, W, bias, ln, Y, y_rows_per_b, y_stride_b, ldy, ncols, accumulate,
             nullptr, 0, 0, 0, 0, 0, nullptr, nullptr};
  return g;
}
// Persistent layer-tail queue (dd_gemm.hip::k_gemm_tail): jobs in dependency order, counters in the workspace.
// Flag words (int32, zeroed by the first launch of every forward), per layer DD_FLAGS_PER_LAYER of them, EACH ON ITS OWN
// 128-byte line (DD_FLAG_STRIDE ints apart): agent-scope atomics on one word retire at ~12 ns each device-wide
// (tools/bench_atomics.hip) and every poll of a word on the same line queues with them -- with all of a layer's words in
// one line the queue took 330 us instead of 40.
enum { DD_FLAG_TICKET = 0, DD_FLAG_LIN = 1, DD_FLAG_TICKET2 = 2, DD_FLAG_PL1 = 3, DD_FLAG_P1Q = 4, DD_FLAG_PBQ = 5, DD_FLAG_PB1R = 6,
       DD_FLAG_NODE = 7, DD_FLAGS_PER_LAYER = 8 };
constexpr int DD_FLAG_LAYERS = 8;
constexpr int DD_FLAG_STRIDE = 32;
constexpr int DD_FLAG_ERR = DD_FLAG_LAYERS * DD_FLAGS_PER_LAYER * DD_FLAG_STRIDE;   // first spin that timed out (0 = none); sticky
#if defined(DD_DEBUG_OPTIONS) && DD_DEBUG_OPTIONS
constexpr int DD_NUM_FLAGS = DD_FLAG_ERR + DD_FLAG_STRIDE;
#else
constexpr int DD_NUM_FLAGS = 0;                                         // the default library's workspace carries no flag words
#endif
constexpr int DD_NUM_COUNTERS = 192;                                    // [0, 64): work counters of the persistent attention workgroups (per
                                                                        // layer), [64, 128): tile counters of the coordinate launches (k_attn2_pos_g), [128, 192): block counters of the persistent
                                                                        // node_layer_with_edge workgroups (two per layer: protein / ligand centres)
constexpr int DD_TAIL_CHUNK = 2;                                        // consecutive tiles drawn per ticket
constexpr int DD_TAIL_MAX_JOBS = 14;
struct TailJob {
  GemmArgs g;
  int nbx, tiles;                        // filled by launch_gemm_tail
  int wait0, wait0_n, wait1, wait1_n;    // flag index (-1: none) that must have reached the target before a tile starts
  int sig;                               // flag index bumped by every finished tile (-1: none)
};
struct TailArgs {
  TailJob job[DD_TAIL_MAX_JOBS];
  int end[DD_TAIL_MAX_JOBS];             // cumulative tile counts
  int njobs, total, ticket, persist;     // ticket: index of this launch's ticket counter; persist: workgroups loop over tickets
  int32_t* flags;
};
inline TailJob tail_job(const GemmArgs& g, int wait0 = -1, int wait0_n = 0, int sig = -1, int wait1 = -1, int wait1_n = 0) {
  TailJob q{g, 0, 0, wait0, wait0_n, wait1, wait1_n, sig};
  return q;
}
inline int gemm_tiles(int rows, int cols) { return ((rows + 63) / 64) * ((cols + 63) / 64); }
int launch_gemm_tail(TailArgs& ta, hipStream_t st);
int launch_gemm128(const GemmArgs& a, hipStream_t st);
// up to 6 independent projections in one launch (small ones ride along with the big ones)
int launch_gemm128_batch(const GemmArgs* jobs, int njobs, hipStream_t st);

int launch_embed_all(const float* protein_h, const float* protein_pos, const float* lig_pos, const int32_t* lig_v,
                     const float* lig_aux, const float* Wl, const float* bl, int B, int NP, int NL, float* h, float* xa, float* xb,
                     const int32_t* bond, long bond_rows, const float* Wb, const float* bb, float* hb, int32_t* counters,
                     hipStream_t st, int32_t* advance = nullptr, int nv = DD_NUM_V);
int launch_drift_armsca(const float* lig_pos, const int32_t* decomp_index, int B, int NL, float min_d, float max_d,
                        float* grad, int accumulate, int norm_B, hipStream_t st);
int launch_drift_arms_repul(const float* lig_pos, const int32_t* decomp_index, int B, int NL, float max_d, int mode, float* grad,
                            int accumulate, int norm_B, hipStream_t st);
// (NP, np_real, nl_real: padded heterogeneous batches -- padding atoms are neither centres nor candidates)
int launch_drift_clash(const float* lig_pos, const float* offset, const float* full_protein_pos, int B, int NL, int NF,
                       float sigma, float gamma, float* grad, int accumulate, const int32_t* nl_real, hipStream_t st);
int launch_knn(const float* x, int B, int N, int K, int32_t* nbr, hipStream_t st, int NP = -1, const int32_t* np_real = nullptr,
               const int32_t* nl_real = nullptr);
int launch_edge_weights(const float* x, const int32_t* nbr, int B, int N, int K, const float* W1T, const float* b1,
                        const float* ln, const float* w2, const float* b2, float* ew, hipStream_t st, int NP = -1,
                        const int32_t* np_real = nullptr, const int32_t* nl_real = nullptr);
int launch_bl_assemble(const float* x, const float* PB, const float* PL, const float* Wgp, int B, int NP, int NL, float* Ek, float* Ev,
                       float* q1, float* Rk, float* Rv, hipStream_t st, const float* xprev = nullptr, const float* dxe = nullptr,
                       const float* dxb = nullptr, float* xout = nullptr, FlagWait fw = FlagWait{nullptr, -1, 0, -1, 0});
// layer-0 rows of the ligand atoms / bonds gathered from dd_sampler.l0_tables (see include/decompdiff_hip.h)
int launch_layer0_rows(const float* tables, const int32_t* lig_v, const float* lig_aux, const int32_t* bond, int B, int NP, int NL,
                       float* l0_P, float* PL, float* l0_qn, float* qlnb, float* PB, float* qb, hipStream_t st);
// head of a forward: kNN graph + edge weights (parts & 1), embeddings / context + layer-0 rows (parts & 2; l0_tables may be NULL)
int launch_head_all(const float* protein_h, const float* protein_pos, const float* lig_pos, const int32_t* lig_v,
                    const float* lig_aux, const float* Wl, const float* bl, int B, int NP, int NL, int K, float* h, float* xa, float* xb,
                    const int32_t* bond, long bond_rows, const float* Wb, const float* bb, float* hb, int32_t* counters,
                    int32_t* advance, int32_t* nbr, float* ew, const float* EW_W1T, const float* EW_b1, const float* EW_ln,
                    const float* EW_w2, const float* EW_b2, const int32_t* np_real, const int32_t* nl_real, const float* l0_tables,
                    float* l0_P, float* PL, float* l0_qn, float* qlnb, float* PB, float* qb, hipStream_t st, int parts = 3,
                    int nv = DD_NUM_V);
int launch_extract_ligand(const float* x, int B, int NP, int NL, float* out, hipStream_t st);

enum { M_NE = 0, M_NB = 1, M_BL = 2, M_PE = 3, M_PB = 4 };

struct AttnArgs {
  int B, NP, NL, K;
  const float* x;          // [B,N,3] positions at layer input
  const int32_t* nbr;      // [B,N,K]
  const float* ew;         // [B,N,K]
  const float *kd, *ks, *vd, *vs, *ke, *ve;   // projection tables
  int ld_kd, ld_ks, ld_vd, ld_vs, ld_ke, ld_ve;
  const float* q;          // [segments,128]
  const float *Akp, *Avp;  // [4,24,128] MFMA A-operand layout (tiled kernel)
  const float *Wakp, *Wavp; // [12,128] merged angle-code columns (BL, tiled kernel)
  const float *lnk, *lnv;  // [2,128]
  const float* W2k;        // [128,128]
  const float* b2v;        // node modes
  const float* W2v;        // node modes, tiled kernel: [128 o][128 c]
  // coordinate modes, optional: second layer of the query MLP evaluated in the kernel (q is then ignored)
  const float *qhid, *lnq, *W2q, *b2q; int ld_qhid;   // hidden pre-activation rows [B*NL, ld_qhid], LN [2,128], W2q^T [128 k,128 o], [128]
  int* work_counter;       // persistent workgroups: next segment to process (zeroed before the launch)
  const float *W2v16, *b2v16;  // pos modes
  float* out;
  const float* dxe;        // PB: result of PE
  float* x_next;           // PB
  const float *Rk, *Rv;    // BL (tiled kernel): per-edge G(d_ji) partial sums [B*Eb,128]
  long long* dbg_clock;    // optional [n_blocks][16] s_memtime stamps of wave 0 (profiling aid), may be NULL
  int out_assign;          // NB: 1 = write (=) rows [B*NL,128] of `out` instead of accumulating into the node table
  // padded heterogeneous batches (dd_sampler.np_real / nl_real / bl_prefix), all NULL for dense batches
  const int32_t *np_real, *nl_real, *bl_prefix;
  // inputs produced by the layer-tail queue running on the other stream (dd_gemm.hip::k_gemm_tail): the launch starts
  // without a graph edge and polls flags[wait_idx] >= wait_n before its first read of them (NULL: ordinary stream order)
  const int32_t* wait_flags; int wait_idx, wait_n;
  // persistent bond-layer workgroups: work_counter hands out TRIPS.  Trips t < trip_full cover NW consecutive segments each
  // (whole rounds of all persistent workgroups); the remainder -- less than one round -- is spread evenly: trip t >= trip_full
  // covers trip_q (< NW) segments, so that the last round runs one wave per SIMD on every CU instead of full trips on some CUs
  // beside idle ones.  Set by launch_attn2_node (trip_q = 0 and trip_full = 1 << 27: plain NW-segment trips).
  int trip_full, trip_q;
  // lin_node inside the node launch (NE / NB blocks, fused launch only; NULL: the attention output goes to `out` as before):
  //   NE: out[row] += W_lin . A[row] + lin_b (+ lin_add[ligand row]: the previous layer's W_lin . A_nb, still pending) -- `out` = h;
  //   NB: out[ligand row] = W_lin . A_nb[row]   (W_lin [128 o][128 c], models/encoders/uni_transformer_edge.py:276-277)
  const float *lin_W, *lin_b, *lin_add;
};

int launch_attn2(int mode, const AttnArgs& a, hipStream_t st);   // one sub-layer: 16-member tiles, scores/aggregation on MFMA
int launch_attn2_node(const AttnArgs& ne, const AttnArgs& nb, const AttnArgs& bl, hipStream_t st);   // NE+NB+BL, one launch (ne.wait_*)
int launch_attn2_pos(const AttnArgs& pe, const AttnArgs& pb, hipStream_t st);                        // PE+PB, one launch
// ... with the projections that feed it computed by its leading workgroups (dd_attention2.hip::k_attn2_pos_g)
int launch_attn2_pos_g(const AttnArgs& pe, const AttnArgs& pb, const GemmArgs* jobs, int njobs, int n_lead, int32_t* counter,
                       hipStream_t st);
int launch_xupdate(const float* x, const float* dxe, const float* dxb, int B, int NP, int NL, float* x_next, hipStream_t st);

struct StepRowsArgs {
  const float* hid;        // [rows,128] first Linear of the head (pre-activation incl. bias)
  const float* logits_in;  // [rows,NC] head outputs supplied by the host instead of hid / W2 / b2 (k_step_rows only), or NULL
  const float* W2;         // [NC,128]
  const float* b2;         // [NC]
  int rows, NC, rows_per_sample;
  const float* tab;        // [4][T] log_alphas, log_1m_alphas, log_cumprod, log_1m_cumprod, then [NC] log prior
  int T;
  const int32_t* step_counter;   // run state [4]: steps done, t_start, seed lo, seed hi (dd_sampler_reset)
  int counter_bias;        // step index = step_counter[0] - counter_bias (1 when the forward's first launch advanced it)
  int32_t* state;          // [rows] current class, updated in place
  const float* uniforms;   // [n_steps, rows, NC] or NULL
  uint32_t stream_id;
  float* logits_out;       // [rows,NC] raw logits (pred_*), may be NULL
  float* traj_recon;       // [n_steps, rows, NC] log_softmax(logits), may be NULL
  float* traj_prob;        // [n_steps, rows, NC] posterior log-probs, may be NULL
  int32_t* traj_state;     // [n_steps, rows] sampled classes, may be NULL
};
struct StepPosArgs {
  int B, NL, T;
  const int32_t* step_counter;   // run state [4], see StepRowsArgs
  int counter_bias;
  const float* x0;          // [B*NL,3] predicted x0 (centred)
  // deferred tail of the forward: x0 = x0_prev[ligand rows] + x0_dxe + x0_dxb (the last layer's coordinate update,
  // same association as k_xupdate), also stored to x0_out; x0 is ignored then
  const float* x0_prev; const float* x0_dxe; const float* x0_dxb; float* x0_out; int NP;
  float* xt;                // [B*NL,3] current positions, updated in place
  const float* tab_pos;     // [3][T] c0, ct, logvar
  const float* tab_score;   // [T]
  const float* atom_std;    // [B*NL,3]
  const float* offset;      // [B,3]
  const float* grad_a; int scale_a;   // armsca gradient (may be NULL)
  const float* grad_c; int scale_c;   // clash gradient (may be NULL)
  const float* grad_r; int scale_r;   // arms_repul gradient (may be NULL); the three are added in this order
  const float* eps;         // [n_steps,B*NL,3] or NULL
  float* traj_pos;          // [n_steps,B*NL,3] or NULL
};
int launch_step_rows(const StepRowsArgs& a, hipStream_t st);
int launch_step_pos(const StepPosArgs& a, hipStream_t st);
int launch_advance(int32_t* ctr, hipStream_t st);
int launch_reset_run_state(int32_t* rs, int t_start, uint64_t seed, hipStream_t st);
int launch_step_all(const StepRowsArgs& rb, const StepRowsArgs& rv, const StepPosArgs& p, hipStream_t st);

}  // namespace dd
