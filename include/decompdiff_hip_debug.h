/*
 * decompdiff_hip_debug.h — measurement, profiling and test-access entry points of libdecompdiff_hip.so.
 *
 * NOT part of the drop-in boundary (include/decompdiff_hip.h): nothing here replaces a reference call site.  bench.py
 * (dd_profile_step), tools/ (dd_debug_set_option and friends; most keys exist only in the measurement build,
 * lib/libdecompdiff_hip_dbg.so, see dd_build_flags) and the GPU tests (dd_workspace_view, dd_debug_philox) use them.
 */
#ifndef DECOMPDIFF_HIP_DEBUG_H
#define DECOMPDIFF_HIP_DEBUG_H

#include "decompdiff_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid for bench.py: runs n_iters reverse steps with HIP events recorded on `stream`
 * around every launch and returns the mean milliseconds per step spent in each kernel class.
 * (Synchronises `stream`; not for use inside a capture.) */
typedef enum dd_prof_cat {
  DD_PROF_MISC, DD_PROF_GEMM, DD_PROF_ASSEMBLE, DD_PROF_ATTN_NE, DD_PROF_ATTN_NB, DD_PROF_ATTN_BL, DD_PROF_ATTN_PE,
  DD_PROF_ATTN_PB, DD_PROF_STEP,
  DD_PROF_EVENT_PAIR,   /* one empty start/stop event pair per step: what the event bracket itself adds to every launch */
  DD_NUM_PROF_CATS
} dd_prof_cat;
int dd_profile_step(const dd_sampler* s, int n_iters, float* ms_per_category /*HOST [DD_NUM_PROF_CATS]*/, void* stream);

/* Profiling aid: per-workgroup s_memtime phase stamps of one attention kernel class (see dd_api.hip). */
int dd_debug_set_clock_buffer(long long* buf, int mode);
/* Launch structure of the attention sub-layers: 1 (default) = fused multi-mode launches of the tiled kernels,
 * 0 = one launch per sub-layer (per-kernel timing; the cross-check variant), 3 = fused launches without the
 * second-stream overlap of the coordinate sub-layers with the next layer's projections (the default 1 has the overlap on).
 * All variants produce the same results up to fp32 summation order. */
int dd_debug_set_fusion(int mode);
/* Measurement aid: runtime switches for A/B timing in one process (key 0: as dd_debug_set_fusion; key 1: K-split
 * projection GEMM tiles on/off).  Results are identical for every setting up to fp32 summation order. */
int dd_debug_set_option(int key, int value);
/* Counter bumped by every dd_debug_set_* call (hosts that cache captured step graphs re-capture when it moved). */
int dd_debug_options_epoch(void);
/* Measured split of the fused node launch for a shape (dd_debug_set_option key 18 = 1): number of CUs kept by the
 * persistent bond-layer workgroups, 0 = node blocks first, -1 = not measured yet. */
int dd_debug_node_split(int B, int NP, int NL, int K);
/* File that keeps the measured splits across processes (per device model, CU count and build of this library; one text line
 * "B NP NL K n_bl" per shape; DD_NODE_SPLIT_CACHE=0 turns it off, DD_NODE_SPLIT_CACHE_DIR moves it from ~/.cache/decompdiff_amd):
 * the ranks of a node and later runs read it instead of timing ~30 forward passes per shape each.  Writes the path (empty
 * string: off) into out[cap]. */
int dd_debug_node_split_cache_path(char* out, int cap);
/* Launch structure in effect, for the accounting of measurement tools (bench.py's per-class FLOP counts): bit 0 = the {P2, PL2}
 * projections of the new h and the heads' first Linear run inside the coordinate launch (k_attn2_pos_g) instead of a GEMM launch. */
int dd_debug_schedule(void);
/* Health word of the in-launch hand-offs of the TILE-QUEUE schedule (dd_debug_set_option(8, 5); measurement build only --
 * the default library's schedule uses graph edges and never polls; EXPERIMENTS.md R3-1): the coordinate attention, the
 * next assemble and the next node attention start beside the persistent GEMM queue of their layer and poll its device
 * counters instead of waiting for a graph edge; every poll is bounded (~0.1 s).  *code = 0:
 * no poll gave up since the last forward started; otherwise the id of the first waiter that did (100+j / 200+j a queue
 * tile of job j, 300/301 assemble, 400 coordinate attention, 500 node attention) -- the results of that forward are then
 * invalid.  Synchronises `stream`.  (ABI 7.)  Exported by the measurement build ONLY (lib/libdecompdiff_hip_dbg.so,
 * dd_build_flags() & 1): the default library has neither the schedule nor its flag words in the workspace. */
#if defined(DD_DEBUG_OPTIONS) && DD_DEBUG_OPTIONS
int dd_queue_error(const dd_sampler* s, void* stream, int* code);
#endif

/* Test aid: the production (Philox4x32-10) noise of one step exactly as the step kernels draw it -- kind 1: uniforms
 * [rows,8] of the atom-type stream (transitions.py:79 rand_like), 2: uniforms [rows,5] of the bond-type stream,
 * 7: normals [rows] of the coordinate stream (decompdiff.py:680 randn_like). */
int dd_debug_philox(uint64_t seed, int step, long rows, int kind, float* out, void* stream);

/* Debug/test access to intermediate buffers of the last dd_forward (pointers into workspace). */
typedef struct dd_ws_view {
  float *x, *h, *hb, *ew, *A;
  int32_t* nbr;
  float* Anb;   /* [B,NL,128] node_layer_with_bond output of the last layer when the fused launch is used, else NULL */
  int32_t lin_in_node;   /* 1: lin_node ran inside the node launch -- `h` is the final h WITHOUT W_lin . A_nb of the last layer on the
                            ligand rows, `Anb` holds exactly that pending term (add it to h[:, NP:]), `A` is NULL (never materialised) */
} dd_ws_view;
int dd_workspace_view(const dd_sampler* s, dd_ws_view* out);

#ifdef __cplusplus
}
#endif
#endif /* DECOMPDIFF_HIP_DEBUG_H */
