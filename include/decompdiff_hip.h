/*
 * decompdiff_hip.h — C ABI of libdecompdiff_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the reverse-diffusion sampling hot path of bytedance/DecompDiff
 * (SURVEY.md §8b).  The reference is pure Python on PyTorch; the native ops its hot path
 * reaches live in third-party wheels (torch_scatter / torch_cluster / torch_sparse / ATen).
 * Each entry point below names the reference call site(s) it replaces.  All pointers are
 * DEVICE pointers unless marked HOST; the caller owns every buffer; nothing is allocated,
 * freed or synchronised inside (two documented exceptions: dd_sample_steps_graph waits for its
 * own replays before destroying the graph, dd_profile_step reads its events); every launch goes
 * to `stream` (a hipStream_t passed as void*).  Return value: 0 on success, negative dd_status on error (no exceptions cross
 * the ABI); dd_status_string() gives the text.
 *
 * Dense fixed-shape layout ("padded/masked fixed-size pocket+ligand graphs"):
 *   B samples; per sample NP protein atoms then NL ligand atoms (N = NP+NL, the context
 *   order of models/common.py:167-194); kNN lists [B,N,K]; the fully connected ligand bond
 *   graph is implicit and dst-major: edge e = dst*(NL-1) + src - (src>dst)
 *   (utils/transforms.py:331-337); triplets (k->j->i) are implicit (closed-form indices).
 */
#ifndef DECOMPDIFF_HIP_H
#define DECOMPDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DD_HIDDEN 128
#define DD_HEADS 16
#define DD_NGAUSS 20
#define DD_KNN_MAX 32
#define DD_NL_MAX 128     /* ligand atoms per sample supported by the fused kernels (tile counts 2 / 3 / 4 / 8 of 16 members) */
#define DD_N_MAX 2048     /* atoms per sample supported by the kNN kernel (candidates per lane: 2 / 4 / 6 / 11 / 16 / 32) */
#define DD_NUM_V 8        /* atom classes of ligand_atom_mode 'basic' (scripts/sample_diffusion_decomp.py:540); dd_sampler.num_v
                             selects 13 ('add_aromatic') or 23 ('full') instead (utils/transforms.py:15-64,138-151) */
#define DD_NUM_V_MAX 23
#define DD_NUM_B 5        /* bond classes  (configs/training.yml:33) */

typedef enum dd_status {
  DD_OK = 0,
  DD_ERR_BAD_ARG = -1,
  DD_ERR_UNSUPPORTED_SHAPE = -2,
  DD_ERR_WORKSPACE_TOO_SMALL = -3,
  DD_ERR_HIP = -4
} dd_status;

/* Packed-weight slots, per layer (decompdiff_amd/packing.py: LAYER_SLOTS, same order). */
typedef enum dd_wslot {
  DD_W_n1, DD_b_n1, DD_W_l1, DD_b_l1, DD_W_b1, DD_b_b1,
  DD_NE_Ak, DD_NE_Av, DD_NE_lnk, DD_NE_lnv, DD_NE_lnq, DD_NE_W2q, DD_NE_b2q, DD_NE_W2k, DD_NE_W2vT, DD_NE_b2v, DD_NE_W2v, DD_NE_Akp, DD_NE_Avp,
  DD_NB_lnk, DD_NB_lnv, DD_NB_lnq, DD_NB_W2q, DD_NB_b2q, DD_NB_W2k, DD_NB_W2vT, DD_NB_b2v, DD_NB_W2v,
  DD_BL_Wg1k, DD_BL_Wg1v, DD_BL_Wg2k, DD_BL_Wg2v, DD_BL_Wak, DD_BL_Wav,
  DD_BL_lnk, DD_BL_lnv, DD_BL_lnq, DD_BL_W2q, DD_BL_b2q, DD_BL_W2k, DD_BL_W2vT, DD_BL_b2v, DD_BL_W2v, DD_BL_Wgp, DD_BL_Wakp, DD_BL_Wavp,
  DD_W_lin, DD_b_lin,
  DD_W_n2, DD_b_n2, DD_W_l2, DD_b_l2, DD_W_b2, DD_b_b2,
  DD_PE_Ak, DD_PE_Av, DD_PE_lnk, DD_PE_lnv, DD_PE_lnq, DD_PE_W2q, DD_PE_b2q, DD_PE_W2k, DD_PE_W2v, DD_PE_b2v, DD_PE_Akp, DD_PE_Avp, DD_PE_W2qT,
  DD_PB_lnk, DD_PB_lnv, DD_PB_lnq, DD_PB_W2q, DD_PB_b2q, DD_PB_W2k, DD_PB_W2v, DD_PB_b2v, DD_PB_W2qT,
  DD_NUM_LAYER_SLOTS
} dd_wslot;

/* Global slots (packing.py: GLOBAL_SLOTS), stored after all layers. */
typedef enum dd_gslot {
  DD_G_W_pemb, DD_G_b_pemb, DD_G_W_lemb, DD_G_b_lemb, DD_G_W_bemb, DD_G_b_bemb,
  DD_G_EW_W1T, DD_G_EW_b1, DD_G_EW_ln, DD_G_EW_w2, DD_G_EW_b2,
  DD_G_VH_W1, DD_G_VH_b1, DD_G_VH_W2, DD_G_VH_b2,
  DD_G_BH_W1, DD_G_BH_b1, DD_G_BH_W2, DD_G_BH_b2,
  DD_NUM_GLOBAL_SLOTS
} dd_gslot;

/* One sampler run: everything `DecompScorePosNet3D.sample_diffusion`
 * (models/decompdiff.py:552-703) keeps alive across its loop. */
typedef struct dd_sampler {
  /* shapes */
  int32_t B, NP, NL, K, NF;        /* NF: full-protein atoms per sample (clash drift), 0 = none */
  int32_t num_layers;
  int32_t T;                       /* length of the schedule tables */
  int32_t t_start;                 /* timestep of step 0 (reference: num_timesteps-1) */
  /* model */
  const float* weights;            /* packed arena */
  const int64_t* slot_off;         /* HOST: offsets (floats) [num_layers*DD_NUM_LAYER_SLOTS + DD_NUM_GLOBAL_SLOTS] */
  const float* tab_pos;            /* [3][T]: posterior_mean_c0_coef, posterior_mean_ct_coef, posterior_logvar */
  const float* tab_v;              /* [4][T]: log_alphas_v, log_one_minus_alphas_v, log_alphas_cumprod_v, log_one_minus_alphas_cumprod_v (atoms),
                                      followed by [8] log prior_probs (DiscreteTransition.prior_probs, transitions.py:114-120; uniform = -log 8) */
  const float* tab_b;              /* [4][T] + [5]: same for bonds */
  const float* tab_score;          /* [T]: pos_score_coef (drift 'scale' option) */
  /* static inputs */
  const float* protein_pos;        /* [B,NP,3] centred (center_pos, decompdiff.py:20-32) */
  const float* protein_h;          /* [B,NP,128] protein embeddings + node indicator (dd_embed_protein) */
  const float* lig_aux;            /* [B,NL,2] */
  const float* atom_std;           /* [B,NL,3] prior_stds[ligand_decomp_batch] */
  const float* offset;             /* [B,3] protein centroid */
  const int32_t* decomp_index;     /* [B,NL] arm id or -1 (armsca drift); may be NULL */
  const float* full_protein_pos;   /* [B,NF,3] un-centred (clash drift); may be NULL */
  /* state, updated in place */
  float* lig_pos;                  /* [B,NL,3] centred x_t */
  int32_t* lig_v;                  /* [B,NL] */
  int32_t* lig_bond;               /* [B,NL*(NL-1)] */
  int32_t* step_counter;           /* [4] device ints, the run state: steps done so far in this run, t_start, seed lo,
                                      seed hi.  Written by dd_sampler_reset() from the t_start / seed fields of this
                                      struct; the kernels read the chain's start time and Philox key from HERE, so a
                                      captured step graph can be re-used for another chain of the same shape */
  /* drift (configs/sampling_drift.yml:31-37) */
  int32_t drift_armsca; float armsca_min_d, armsca_max_d; int32_t armsca_scale;
  int32_t drift_clash;  float clash_sigma, clash_gamma;    int32_t clash_scale;
  int32_t drift_norm_batch;        /* batch size the armsca loss is averaged over (guidance_funcs.py:78); 0 = B.
                                      Set when a larger (e.g. ragged) batch is run in several dense groups. */
  /* noise: injected (reference draw order, decompdiff.py:620,633,680) or Philox when NULL */
  const float* u_v;                /* [n_steps,B*NL,8]  uniforms for atom types   */
  const float* u_b;                /* [n_steps,B*Eb,5]  uniforms for bond types   */
  const float* eps;                /* [n_steps,B*NL,3]  normals for positions     */
  uint64_t seed;
  /* trajectories (may each be NULL): indexed by step */
  float* traj_pos;                 /* [n_steps,B*NL,3]  x_{t-1} + offset          */
  int32_t* traj_v;                 /* [n_steps,B*NL]                              */
  int32_t* traj_bond;              /* [n_steps,B*Eb]                              */
  float* traj_v0;                  /* [n_steps,B*NL,8]  log_softmax(v logits)     */
  float* traj_vt;                  /* [n_steps,B*NL,8]  posterior log-probs       */
  float* traj_bt;                  /* [n_steps,B*Eb,5]  bond posterior log-probs  */
  /* outputs of the last forward (always written) */
  float* pred_pos;                 /* [B,NL,3]  x0-hat (centred)                  */
  float* pred_v;                   /* [B,NL,8]  logits                            */
  float* pred_bond;                /* [B,Eb,5]  logits                            */
  /* scratch */
  float* workspace; size_t workspace_floats;
  /* heterogeneous batches ("padded/masked fixed-size pocket+ligand graphs"): samples with fewer atoms than NP / NL are
   * padded to the dense shape -- sample b's real protein atoms are rows 0 .. np_real[b]-1 of its protein block, its real
   * ligand atoms rows 0 .. nl_real[b]-1 of its ligand block (PyG collate of samples with different sizes:
   * utils/data.py:389-446, scripts/sample_diffusion_decomp.py:300-326).  Padding rows never enter a kNN list, a bond
   * or triplet segment, a softmax or a drift term; their own rows hold finite don't-care values.  NULL = dense. */
  const int32_t* np_real;          /* [B] or NULL */
  const int32_t* nl_real;          /* [B] or NULL */
  const int32_t* bl_prefix;        /* [B+1] prefix sums of nl_real*(nl_real-1) (compact enumeration of the real
                                      bond-layer segments); required when nl_real is given */
  /* Layer-0 tables (optional; all three NULL = off).  The first layer's projections and query rows are functions of the
   * embedded inputs alone: a ligand atom's rows depend on its (class, arm flag) -- 16 combinations -- a bond's on its
   * type (5) and its destination atom's combination, and the protein rows are the same in every step of a chain.  With
   * the tables set, a step gathers those rows instead of running the first layer's projection and query GEMMs
   * (models/common.py:85-105 MLP first Linear + the query MLPs of uni_transformer_edge.py:42-167).  Requires
   * lig_aux rows to be exactly (1,0) or (0,1) (utils/transforms.py arm / scaffold indicator) and a dense batch. */
  const float* l0_tables;          /* [DD_L0_TABLE_FLOATS], device; filled by dd_layer0_tables() once per weight set */
  float* l0_P;                     /* [B,N,640] layer-0 node projections; protein rows by dd_layer0_prepare() */
  float* l0_qn;                    /* [B,N,128] layer-0 node queries; protein rows by dd_layer0_prepare() */
  /* ABI 7: atom-class count of the checkpoint (0 = DD_NUM_V = 8; 13; 23): width of lig_v's range, of u_v, pred_v,
   * traj_v0 / traj_vt, of the v head's second Linear and (num_v + 2) of the ligand embedding's rows.  The layer-0 tables
   * exist for 8 classes only (l0_tables must be NULL otherwise). */
  int32_t num_v;
  /* ABI 8: arms_repul drift (utils/guidance_funcs.py:81-118; an EXTENSION of energy_drift_opt -- the reference defines the
   * energy but its sample_diffusion has no branch for it, models/decompdiff.py:643-675).  0 = off, 1 = mode 'min',
   * 2 = mode 'all'; evaluated at x_t like the other terms, scaled by pos_score_coef[t] when repul_scale is set. */
  int32_t drift_repul;
  float repul_max_d;
  int32_t repul_scale;
} dd_sampler;

/* Layout of l0_tables (floats): node projections [16][640], ligand projections [16][1280], bond projections [5][640],
 * node queries [16][128], node-with-bond queries [16][128], bond-layer queries [16 dst combinations][5 types][128].
 * Combination = 8 * (arm flag) + class. */
#define DD_L0_TABLE_FLOATS (16 * 640 + 16 * 1280 + 5 * 640 + 16 * 128 + 16 * 128 + 80 * 128)
/* Build the tables: `mini` is a sampler with B = 1, NP = 0, NL = 16, K = 15 whose lig_v[i] = i % 8, lig_aux[i] = (i < 8 ?
 * (1,0) : (0,1)) and lig_bond[dst * 15 + s] = s % 5, with the model's weights and its own workspace; the SAME embedding,
 * projection and query kernels the forward uses run on it, so a gathered row equals the GEMM's row bit for bit. */
int dd_layer0_tables(const dd_sampler* mini, float* tables, void* stream);
/* Per chain (after protein_h is in place): the protein rows of s->l0_P and s->l0_qn. */
int dd_layer0_prepare(const dd_sampler* s, void* stream);

const char* dd_status_string(int status);
int dd_abi_version(void);
/* ABI 9: form of the attention MLPs in the packed arena that the kernels of this build expect.
 *   0 = canonical: the reference's values in the slot layout of decompdiff_amd/packing.py::pack_layer;
 *   1 = kernel form (packing.py::kernel_form_layer, exact algebra): every key / value MLP of the five attention sub-layers
 *       (models/encoders/uni_transformer_edge.py:42-74,125-167,188-210; MLP = models/common.py:85-105) has the channel mean of
 *       each additive first-Linear part removed (LayerNorm ignores it), sign(gamma) folded into those parts, |gamma| into the
 *       columns of the second Linear (W2k, W2v), beta / |gamma| in the LayerNorm slot's second row and the softmax scale
 *       1 / sqrt(8) inside W2k.  Same slots, shapes and offsets. */
int dd_weights_form(void);

/* (Re)start a chain: step_counter[0..3] = {0, s->t_start, s->seed}.  Must be enqueued on `stream` before the first
 * dd_sample_steps* / dd_graph_launch / dd_reverse_step of a chain. */
int dd_sampler_reset(const dd_sampler* s, void* stream);

/* Floats of workspace needed by dd_forward/dd_sample_steps for these shapes. */
size_t dd_workspace_floats(int B, int NP, int NL, int K);

/* torch_geometric.nn.knn_graph -> torch_cluster.knn (call site uni_transformer_edge.py:353):
 * per-sample K nearest other atoms, ascending (d2, index); d2 = (dx*dx+dy*dy)+dz*dz in fp32. */
int dd_knn(const float* x /*[B,N,3]*/, int B, int N, int K, int32_t* nbr /*[B,N,K]*/, void* stream);
/* The same for a padded heterogeneous batch in the layout of dd_sampler.np_real / nl_real (NP + NL rows per sample, the real atoms
 * first in each block; np_real may be NULL = all NP rows real): padding atoms are neither centres nor candidates; the list of a
 * padding centre is zeros.  Every sample needs at least K + 1 real atoms.  (Round 5: the padded training network.) */
int dd_knn_masked(const float* x /*[B,NP+NL,3]*/, int B, int NP, int NL, int K, const int32_t* np_real, const int32_t* nl_real,
                  int32_t* nbr /*[B,NP+NL,K]*/, void* stream);

/* e_w = sigmoid(MLP(GaussianSmearing(dist)))  (uni_transformer_edge.py:422-427). */
int dd_edge_weights(const float* x, const int32_t* nbr, int B, int N, int K, const float* W1T /*[20,128]*/,
                    const float* b1, const float* ln /*[2,128]*/, const float* w2 /*[128]*/, const float* b2 /*[1]*/,
                    float* ew /*[B,N,K]*/, void* stream);

/* nn.Linear / MLP first-layer projections (ATen addmm call sites in models/common.py:85-105):
 * Y[r, 0:ncols] (+)= op(X[r, 0:128]) * W[ncols,128]^T + bias, fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Row r of X lives at X + (r / x_rows_per_b) * x_stride_b + (r % x_rows_per_b) * ldx (floats);
 * same for Y.  ln != NULL applies LayerNorm(gamma,beta)+ReLU to each X row first (the MLP's
 * hidden activation); accumulate != 0 adds into Y. */
int dd_gemm128(const float* X, int x_rows_per_b, long x_stride_b, int ldx, int rows, const float* W,
               const float* bias, const float* ln, float* Y, int y_rows_per_b, long y_stride_b, int ldy,
               int ncols, int accumulate, void* stream);

/* nn.Linear backward of the training step (scripts/train_diffusion_decomp.py; the ATen mm calls autograd makes behind
 * models/common.py:85-105), weight gradient:  out[M,128] (+)= A[rows,M]^T * X[rows,128]  (A = dY, M <= 128 output channels,
 * row r of A at A + r*lda, of X at X + r*ldx, of out at out + o*ldo).  The input gradient dX = dY * W is dd_gemm128 with the
 * transposed weight.  fp32 MFMA, slabs of rows reduced in a fixed order (bitwise reproducible, no atomics);
 * scratch: dd_gemm128_tn_scratch_floats(rows, M) floats. */
size_t dd_gemm128_tn_scratch_floats(long rows, int M);
int dd_gemm128_tn(const float* A, int lda, int M, const float* X, int ldx, long rows, float* scratch, float* out, int ldo,
                  int accumulate, void* stream);
/* ... with the bias gradient of the same Linear: bias_out[M] = column sums of A (dY^T 1; autograd's dy.sum(0)), formed from the
 * rows the kernel stages anyway, slabs added in the same fixed order.  Same scratch size.  (Added in round 5; ABI unchanged:
 * a new entry point, no struct layout touched.) */
int dd_gemm128_tn_bias(const float* A, int lda, int M, const float* X, int ldx, long rows, float* scratch, float* out, int ldo,
                       int accumulate, float* bias_out, void* stream);

/* LayerNorm(128) + ReLU of the reference's MLPs (models/common.py:85-105) for the training step, one kernel each way:
 *   forward : y[r] = relu(LN(x[r]) * gamma + beta) (eps 1e-5), stats[r] = (mean, 1/std) kept for the backward;
 *   backward: dx from dy (the ReLU mask is recomputed from x), dgamma[128] / dbeta[128] summed over the rows in a fixed order
 *             (no atomics); scratch: dd_ln_relu_scratch_floats(rows) floats.
 * Rows are contiguous [rows,128] fp32; x / y / dy / dx / gamma / beta 16-byte aligned (read as float4), stats 8-byte aligned
 * (checked: DD_ERR_BAD_ARG otherwise); rows == 0 is DD_OK both ways.  (Round 5; new entry points, ABI unchanged.) */
size_t dd_ln_relu_scratch_floats(long rows);
int dd_ln_relu_forward(const float* x, const float* gamma, const float* beta, float* y, float* stats, long rows, void* stream);
int dd_ln_relu_backward(const float* x, const float* stats, const float* gamma, const float* beta, const float* dy, float* dx,
                        float* scratch, float* dgamma, float* dbeta, long rows, void* stream);

/* Op-level message passing: the torch_scatter pairs of the reference's attention layers as stand-alone ops
 *     alpha = scatter_softmax((q[dst] * k / sqrt(8)).sum(-1), dst, dim=0);  out = scatter_sum(alpha[..., None] * v, dst, dim=0)
 * (uni_transformer_edge.py:63-68 NodeUpdateLayer, :158-164 BondUpdateLayer, :205-211 PosUpdateLayer).  16 heads x 8
 * channels.  Edges grouped by destination: seg_ptr[s] .. seg_ptr[s+1] are the edges of destination s (knn_graph's,
 * the dst-major bond list's and the SparseTensor triplets' order).  v is multiplied by e_w[edge] when e_w != NULL
 * (:55-56).  q is [n_seg,128], or per edge [E,128] with q_per_edge != 0 (rows of a segment are then identical and
 * the first is read).  Destinations without edges get zeros.  The sampling loop does not call these (its fused
 * kernels never materialise q / k / v); they serve hosts that keep the reference's Python layers. */
int dd_attn_aggregate_node(const float* q, int q_per_edge, const float* k /*[E,128]*/, const float* v /*[E,128]*/,
                           const float* e_w /*[E] or NULL*/, const int32_t* seg_ptr /*[n_seg+1]*/, int n_seg,
                           float* out /*[n_seg,128]*/, void* stream);
/* BondUpdateLayer form: q per triplet, no e_w. */
int dd_attn_aggregate_triplet(const float* q /*[E3,128]*/, const float* k, const float* v, const int32_t* seg_ptr, int n_seg,
                              float* out /*[n_seg,128]*/, void* stream);
/* PosUpdateLayer form: v16 [E,16] per head, rel_x [E,3]; out[s] = mean_heads(sum_e alpha * v16 * e_w * rel_x) [n_seg,3]. */
int dd_attn_aggregate_pos(const float* q /*[n_seg,128]*/, const float* k, const float* v16, const float* e_w, const float* rel_x,
                          const int32_t* seg_ptr, int n_seg, float* out /*[n_seg,3]*/, void* stream);

/* Stand-alone torch_scatter drop-ins over dim 0 of a [E,F] fp32 tensor whose rows are grouped by destination (CSR
 * segments seg_ptr [n_seg+1]); reference call sites: scatter_softmax / scatter_sum uni_transformer_edge.py:64,68,160,164,
 * 205,209, scatter_mean decompdiff.py:25 (center_pos), scatter_min guidance_funcs.py:52.
 * dd_segment_reduce: op 0 sum, 1 mean, 2 min, 3 max -> out [n_seg,F]; empty segments give 0 (torch_scatter's fill);
 * min / max also write the row index of the extremum to arg_out [n_seg,F] (may be NULL; E for an empty segment, the
 * smallest index on ties).  dd_segment_softmax: out[e,f] = exp(src[e,f] - max_seg) / sum_seg exp(...), [E,F]. */
int dd_segment_reduce(const float* src, const int32_t* seg_ptr, int n_seg, int F, int op, long E, float* out,
                      int64_t* arg_out, void* stream);
int dd_segment_softmax(const float* src, const int32_t* seg_ptr, int n_seg, int F, float* out, void* stream);

/* Embeddings (decompdiff.py:219-256, 296-297). protein_h is step-invariant. */
int dd_embed_protein(const float* protein_v /*[B*NP,29]*/, int rows, const float* W /*[128,29]*/, const float* b,
                     float* protein_h /*[rows,128]*/, void* stream);

/* Whole score network once: DecompScorePosNet3D.forward for the shipped config
 * (decompdiff.py:213-351 -> uni_transformer_edge.py:394-443).  Reads s->lig_pos/lig_v/lig_bond,
 * writes s->pred_pos/pred_v/pred_bond. */
int dd_forward(const dd_sampler* s, void* stream);

/* One reverse transition from network outputs the host computed itself (decompdiff.py:601-689 without the forward):
 * log_softmax + q_v_posterior + log_sample_categorical for atoms and bonds (logits_v [B*NL,8], logits_b [B*Eb,5]),
 * C0 posterior mean from x0 [B*NL,3] (centred), drift, noise; updates s->lig_pos / lig_v / lig_bond, writes the
 * trajectories at the current step index and advances s->step_counter -- exactly what a step of dd_sample_steps does
 * after its forward (bit-identical when fed dd_forward's pred_*). */
int dd_reverse_step(const dd_sampler* s, const float* logits_v, const float* logits_b, const float* x0, void* stream);

/* n_steps iterations of the reverse loop (decompdiff.py:575-689): forward, categorical
 * posteriors + Gumbel-argmax, Gaussian posterior mean, optional drift, noise, trajectories. */
int dd_sample_steps(const dd_sampler* s, int n_steps, void* stream);

/* Same loop with one step captured into a hipGraph and replayed n_steps times. */
int dd_sample_steps_graph(const dd_sampler* s, int n_steps, void* stream);

/* The same, split so that a host can interleave its own work (e.g. draining trajectory chunks to the host on a copy
 * stream) with the replays: create captures `steps_per_graph` (>= 1) consecutive steps of chain `s` on `stream` into one
 * executable graph; launch replays it n_graphs times on `stream` without waiting; destroy waits for nothing — the
 * caller synchronises the stream first.  The dd_sampler and every buffer it points to must stay alive and unchanged
 * until the graph is destroyed. */
int dd_graph_create(const dd_sampler* s, int steps_per_graph, void* stream, void** graph_out /*HOST*/);
int dd_graph_launch(void* graph, int n_graphs, void* stream);
int dd_graph_destroy(void* graph);

/* n independent chains advanced together: chain i's step graph is captured on streams[i] (HOST array of n distinct,
 * non-default streams) and the n graphs are replayed round-robin, so chains with different sizes (the sub-batches of
 * a heterogeneous PyG batch: sample_diffusion_decomp.py:300-326 collates samples with different ligand sizes when
 * num_atoms_mode is 'prior' / 'old' / 'stat') share the GPU instead of running back to back.  Every chain needs its
 * own workspace and state buffers.  Waits for all streams before returning. */
int dd_sample_steps_graph_multi(const dd_sampler* const* s /*HOST [n]*/, int n, int n_steps,
                                void* const* streams /*HOST [n]*/);

/* Drift guidance gradients at x_t (utils/guidance_funcs.py:24-78), analytic. grad [B,NL,3]. */
int dd_drift_armsca(const float* lig_pos, const int32_t* decomp_index, int B, int NL, float min_d, float max_d,
                    float* grad, int accumulate, void* stream);
int dd_drift_clash(const float* lig_pos, const float* offset, const float* full_protein_pos, int B, int NL, int NF,
                   float sigma, float gamma, float* grad, int accumulate, void* stream);
/* grad[B*NL,3] (+)= d/dx of compute_batch_arms_repul_loss (utils/guidance_funcs.py:94-118, :81-91) -- the torch.autograd.grad
 * of that energy: sum over the samples' arm pairs a1 <= a2 of relu(max_d - min distance) (mode 1 = 'min') or of the mean of
 * relu(max_d - distance) over all atom pairs (mode 2 = 'all'), divided by B.  decomp_index [B*NL]: -1 scaffold, >= 0 arm id. */
int dd_drift_arms_repul(const float* lig_pos, const int32_t* decomp_index, int B, int NL, float max_d, int mode, float* grad,
                        int accumulate, void* stream);

/* Build flags of the loaded library: bit 0 = measurement build (-DDD_DEBUG_OPTIONS=1: alternative launch schedules behind
 * dd_debug_set_option, see decompdiff_hip_debug.h), bit 1 = -DDD_EXACT_MATH=1 (correctly rounded 1/sqrt and softmax
 * division).  0 for the default library. */
int dd_build_flags(void);

/* Measurement, profiling and test-access entry points (dd_profile_step, dd_debug_*, dd_queue_error, dd_workspace_view)
 * are declared in decompdiff_hip_debug.h: exported by the same library, not part of the drop-in boundary. */

#ifdef __cplusplus
}
#endif
#endif /* DECOMPDIFF_HIP_H */
