// fp32 projection GEMM, K = 128:  Y[r, c] (+)= op(X[r, :]) . W[c, :] + bias[c]
//
// Replaces the ATen addmm call sites behind every nn.Linear of the reference's MLPs
// (models/common.py:85-105) after the first-Linear factorisation (packing.py).  Exact fp32
// on the matrix cores: v_mfma_f32_32x32x2_f32 is bitwise a k-ordered fmaf chain
// (cdna_hip_programming.md §3), so results do not depend on scheduling.
//
// Tile: 64 rows x 64 cols per 256-thread workgroup, whole K=128 staged in LDS once
// (2 x 64 x 130 floats = 66.6 KB -> 2 workgroups/CU); 4 waves as 2x2, one 32x32
// accumulator each, 64 MFMAs per wave.  LDS row pitch 130 floats makes the per-lane
// ds_read_b64 operand fetch (row = lane&31) conflict-free: bank = (2*row + const) mod 64.
#include "dd_kernels.hpp"
#include "dd_gemm_tile.hpp"

namespace dd {

__device__ __forceinline__ void gemm_tile(const GemmArgs& a, const int bx, const int by) {
  __shared__ __attribute__((aligned(16))) float smg[2 * GT * GP];
  float* Xs = smg;
  float* Ws = smg + GT * GP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = bx * GT, col0 = by * GT;
  const bool xplain = a.x_rows_per_b >= a.rows;

  // ---- stage X tile (64 rows x 128) and W tile (64 cols x 128): all 16 float4 global loads of a thread
  //      are issued before the first LDS store; two ds_write_b64 each (pitch 130 is only 8-byte aligned)
  {
    float4 xv[8], wv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + k * 256;
      const int r = i >> 5, c4 = (i & 31) * 4;
      const int gr = row0 + r, gc = col0 + r;
      xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      wv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < a.rows) {
        const float* src = a.X + row_offset(gr, a.x_rows_per_b, a.x_stride_b, a.ldx, xplain) + c4;
        xv[k] = *reinterpret_cast<const float4*>(src);
        if (a.X2 != nullptr) {
          const int bb = gr / a.x2_N, n = gr % a.x2_N;
          if (n >= a.x2_NP) {
            const float4 t = *reinterpret_cast<const float4*>(a.X2 + ((long)bb * (a.x2_N - a.x2_NP) + (n - a.x2_NP)) * 128 + c4);
            xv[k].x += t.x; xv[k].y += t.y; xv[k].z += t.z; xv[k].w += t.w;
          }
        }
      }
      if (gc < a.ncols) wv[k] = *reinterpret_cast<const float4*>(a.W + (long)gc * 128 + c4);
    }
    // optional LayerNorm+ReLU prologue (MLP hidden activation), on the registers: a row's 128 channels sit in the 32
    // lanes of a half wave (4 each), so mean / variance are half-wave reductions (4 DPP steps + one permlane swap)
    if (a.ln != nullptr) {
      const int c4 = (tid & 31) * 4;
      const float4 gm = *reinterpret_cast<const float4*>(a.ln + c4);
      const float4 bt = *reinterpret_cast<const float4*>(a.ln + 128 + c4);
      auto half_sum = [](float v) {
        v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v); v += dpp_mov<0x140>(v);
        return swap16_sum(v, v);
      };
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float mean = half_sum((xv[k].x + xv[k].y) + (xv[k].z + xv[k].w)) * (1.0f / 128.0f);
        const float dx = xv[k].x - mean, dy = xv[k].y - mean, dz = xv[k].z - mean, dw = xv[k].w - mean;
        const float var = half_sum(fmaf(dx, dx, dy * dy) + fmaf(dz, dz, dw * dw)) * (1.0f / 128.0f);
        const float rstd = dd_rsqrt(var + 1e-5f);
        xv[k].x = fmaxf(fmaf(dx * rstd, gm.x, bt.x), 0.f);
        xv[k].y = fmaxf(fmaf(dy * rstd, gm.y, bt.y), 0.f);
        xv[k].z = fmaxf(fmaf(dz * rstd, gm.z, bt.z), 0.f);
        xv[k].w = fmaxf(fmaf(dw * rstd, gm.w, bt.w), 0.f);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int i = tid + k * 256;
      const int r = i >> 5, c4 = (i & 31) * 4;
      float2* d = reinterpret_cast<float2*>(&Xs[r * GP + c4]);
      d[0] = make_float2(xv[k].x, xv[k].y);
      d[1] = make_float2(xv[k].z, xv[k].w);
      float2* e = reinterpret_cast<float2*>(&Ws[r * GP + c4]);
      e[0] = make_float2(wv[k].x, wv[k].y);
      e[1] = make_float2(wv[k].z, wv[k].w);
    }
  }
  __syncthreads();

  // ---- MFMA: wave (wr, wc) computes rows wr*32.., cols wc*32..
  const int wr = wave >> 1, wc = wave & 1;
  const int li = lane & 31, hh = lane >> 5;
  const float* xa = &Xs[(wr * 32 + li) * GP + 2 * hh];
  const float* wb = &Ws[(wc * 32 + li) * GP + 2 * hh];
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 8
  for (int kk = 0; kk < 32; ++kk) {
    float2 av = *reinterpret_cast<const float2*>(xa + 4 * kk);
    float2 bv = *reinterpret_cast<const float2*>(wb + 4 * kk);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
  }

  __syncthreads();                                     // operands dead: the tile buffer becomes the output stage
  gemm_epilogue<false>(a, smg, acc, row0, col0);
}

__global__ __launch_bounds__(256) void k_gemm128(GemmArgs a) { gemm_tile(a, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(256) void k_gemm128_ks(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float sm[2 * GT * GPH];
  gemm_tile_ksplit<true>(a, blockIdx.x, blockIdx.y, sm);
}

constexpr int DD_GEMM_BATCH_MAX = 6;
struct GemmBatch { GemmArgs job[DD_GEMM_BATCH_MAX]; int end[DD_GEMM_BATCH_MAX]; int nbx[DD_GEMM_BATCH_MAX]; int big[DD_GEMM_BATCH_MAX]; int njobs; };
// `nby` > 0 selects the XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
// with the plain order (row block fastest) the 64 x 128 input block of a row block is pulled into every L2 once per
// column tile.  Here a job's block range starts at a multiple of 8, XCD x = lb % 8 owns the row blocks x, x + 8, ...
// and walks their column tiles back to back: an input block is filled into one L2 once (the weights, 0.3 MB, are
// L2-resident everywhere).  Blocks past the last row block of an XCD exit.
template <bool KS>
__device__ __forceinline__ void gemm_job(const GemmArgs& a, int lb, int nbx, int nby, float* sm) {
  int bx = lb % nbx, by = lb / nbx;
  if (nby > 0) {
    const int k = lb >> 3;
    bx = (lb & 7) + 8 * (k / nby);
    by = k % nby;
    if (bx >= nbx) return;
  }
  if (KS) gemm_tile_ksplit<false>(a, bx, by, sm);
  else gemm_tile(a, bx, by);
}
template <bool KS>
__global__ __launch_bounds__(256) void k_gemm128_batch(GemmBatch gb) {
  __shared__ __attribute__((aligned(16))) float sm[KS ? 2 * GT * GPH : 4];
  int j = 0, base = 0;
  const int blk = blockIdx.x;
#pragma unroll
  for (int i = 0; i < DD_GEMM_BATCH_MAX - 1; ++i)
    if (j == i && blk >= gb.end[i] && i + 1 < gb.njobs) { base = gb.end[i]; j = i + 1; }
  const int lb = blk - base;
  // block-uniform job selection through a uniform index into the kernel arguments (scalar loads): ONE copy of the tile
  // code.  (One inlined copy per job made this kernel 70 KB -- more than the 64 KB instruction cache two CUs share.)
  const GemmArgs a = gb.job[j];
  gemm_job<KS>(a, lb, gb.nbx[j], gb.big[j], sm);
}


#if defined(DD_DEBUG_OPTIONS) && DD_DEBUG_OPTIONS
// ---------------------------------------------------------------------------------------------------------------------
// Persistent layer-tail queue: every dense GEMM between two node-attention launches -- lin_node, the projections of the
// coordinate sub-layers, the NEXT layer's projections and its query MLPs' second Linear (the ATen addmm chain of
// models/common.py:85-105 per uni_transformer_edge.py:259-287) -- as ONE launch.  The tiles of all jobs form one list in
// dependency order; a workgroup draws tickets from a device counter (the next ticket is drawn while the current tile is
// multiplied), waits -- one lane, relaxed polls -- until the counters its job names have reached their targets, runs the
// K-split 64 x 64 tile with write-through (sc1) stores and L1-bypassing (sc1) loads, and bumps the job's counters once
// the workgroup's stores have drained.  Tickets are handed out in list order and a tile only waits for counters fed by
// tiles EARLIER in the list, which were therefore drawn by workgroups that are resident and never wait for later
// tickets: progress does not depend on dispatch order, placement or the number of resident workgroups.  The same
// counters let the kernels around the queue start without a graph edge: the coordinate attention and the next assemble
// run on the other stream and poll them (dd_attention2.hip / dd_graph.hip: wait_flags).
// Spins are bounded: after ~0.1 s a waiter records an error code in flags[DD_FLAG_ERR] and goes on (the host reads the
// word at the end of a chain) instead of hanging the device.
template <bool SC1, bool WAITS, bool STATIC = false>
__global__ __launch_bounds__(256) void k_gemm_tail(TailArgs ta) {
  __shared__ __attribute__((aligned(16))) float sm[2 * GT * GPH + 4];
  int* s_t = reinterpret_cast<int*>(sm + 2 * GT * GPH);
  int32_t* flags = ta.flags;
  if (STATIC) {                                          // timing only: tiles dealt round-robin, no ticket counter
    for (int t0 = blockIdx.x * DD_TAIL_CHUNK; t0 < ta.total; t0 += gridDim.x * DD_TAIL_CHUNK)
      for (int t = t0; t < t0 + DD_TAIL_CHUNK && t < ta.total; ++t) {
        int j = 0, base = 0;
        for (int i = 0; i + 1 < ta.njobs; ++i)
          if (j == i && t >= ta.end[i]) { base = ta.end[i]; j = i + 1; }
        const TailJob q = ta.job[j];
        const int lb = t - base;
        if (WAITS && threadIdx.x == 0) {
          dd_poll_flag(flags, q.wait0, q.wait0_n, DD_FLAG_ERR, 100 + j);
          dd_poll_flag(flags, q.wait1, q.wait1_n, DD_FLAG_ERR, 200 + j);
        }
        __syncthreads();
        gemm_tile_ksplit<false, SC1>(q.g, lb % q.nbx, lb / q.nbx, sm);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && q.sig >= 0) __hip_atomic_fetch_add(flags + q.sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    return;
  }
  if (threadIdx.x == 0) s_t[0] = __hip_atomic_fetch_add(flags + ta.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  int t = s_t[0];
  while (t < ta.total) {
    int j = 0, base = 0;
    for (int i = 0; i + 1 < ta.njobs; ++i)
      if (j == i && t >= ta.end[i]) { base = ta.end[i]; j = i + 1; }
    const TailJob q = ta.job[j];                         // (uniform index: scalar loads from the kernel arguments)
    const int lb = t - base;
    if (WAITS && threadIdx.x == 0) {
      dd_poll_flag(flags, q.wait0, q.wait0_n, DD_FLAG_ERR, 100 + j);
      dd_poll_flag(flags, q.wait1, q.wait1_n, DD_FLAG_ERR, 200 + j);
    }
    __syncthreads();
    int nxt = 0;
    gemm_tile_ksplit<false, SC1>(q.g, lb % q.nbx, lb / q.nbx, sm, ta.persist ? flags + ta.ticket : nullptr, 1, &nxt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores ...
    __syncthreads();                                     // ... before ONE lane publishes (and before sm is reused)
    if (threadIdx.x == 0) {
      if (q.sig >= 0) __hip_atomic_fetch_add(flags + q.sig, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_t[0] = nxt;
    }
    if (!ta.persist) return;
    __syncthreads();
    t = s_t[0];
  }
}

int g_tail_variant = 0;   // dd_debug_set_option(25, v): timing-only variants of the queue (results invalid for v != 0)
int launch_gemm_tail(TailArgs& ta, hipStream_t st) {
  if (ta.njobs <= 0 || ta.njobs > DD_TAIL_MAX_JOBS || !ta.flags) return DD_ERR_BAD_ARG;
  int total = 0;
  for (int i = 0; i < ta.njobs; ++i) {
    TailJob& q = ta.job[i];
    q.nbx = (q.g.rows + GT - 1) / GT;
    q.tiles = q.nbx * ((q.g.ncols + GT - 1) / GT);
    total += q.tiles;
    ta.end[i] = total;
  }
  for (int i = ta.njobs; i < DD_TAIL_MAX_JOBS; ++i) ta.end[i] = total;
  ta.total = total;
  ta.persist = g_tail_variant == 7 ? 1 : 0;
  if (total <= 0) return DD_OK;
  static int slots = 0;
  if (slots == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    slots = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                ? 4 * prop.multiProcessorCount : 1024;
  }
  // one ticket per workgroup (grid = tiles): workgroups leave when their tile is done, so a launch with a large per-CU
  // footprint on the other stream (the coordinate attention: one workgroup per CU through its LDS) still finds CUs --
  // persistent workgroups (g_tail_variant 7: grid = resident slots, tickets drawn in a loop) hold theirs until the queue is empty
  const dim3 grid(g_tail_variant == 7 ? (total < slots ? total : slots) : total);
  if (g_tail_variant == 1) hipLaunchKernelGGL((k_gemm_tail<true, false>), grid, dim3(256), 0, st, ta);        // timing only: no waits
  else if (g_tail_variant == 2) hipLaunchKernelGGL((k_gemm_tail<false, true>), grid, dim3(256), 0, st, ta);   // timing only: plain memory ops
  else if (g_tail_variant == 3) hipLaunchKernelGGL((k_gemm_tail<false, false>), grid, dim3(256), 0, st, ta);
  else if (g_tail_variant == 4) hipLaunchKernelGGL((k_gemm_tail<true, false, true>), grid, dim3(256), 0, st, ta);    // static, no waits
  else if (g_tail_variant == 5) hipLaunchKernelGGL((k_gemm_tail<true, true, true>), grid, dim3(256), 0, st, ta);     // static + waits (serialized runs only)
  else if (g_tail_variant == 6) hipLaunchKernelGGL((k_gemm_tail<true, true, true>), dim3(total / DD_TAIL_CHUNK + 1), dim3(256), 0, st, ta);   // one chunk per workgroup
  else hipLaunchKernelGGL((k_gemm_tail<true, true>), grid, dim3(256), 0, st, ta);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

#endif

long long* g_gemm_dbg = nullptr;
int g_gemm_ksplit = 1;   // dd_debug_set_option(1, v): K-split tiles for jobs without a LayerNorm prologue

int g_gemm_xcd = 1;      // dd_debug_set_option(21, v): XCD-aware tile order in the batched projection launches

int launch_gemm128_batch(const GemmArgs* jobs, int njobs, hipStream_t st) {
  if (njobs <= 0 || njobs > DD_GEMM_BATCH_MAX) return DD_ERR_BAD_ARG;
  GemmBatch gb;
  gb.njobs = njobs;
  int total = 0;
  for (int i = 0; i < DD_GEMM_BATCH_MAX; ++i) {
    if (i < njobs) {
      gb.job[i] = jobs[i];
      const int t = GT;
      gb.nbx[i] = (jobs[i].rows + t - 1) / t;
      const int nby = (jobs[i].ncols + t - 1) / t;
      const bool xcd = g_gemm_xcd && gb.nbx[i] >= 16 && nby >= 2;        // (nothing to share with one column tile)
      gb.big[i] = xcd ? nby : 0;
      total += xcd ? 8 * ((gb.nbx[i] + 7) / 8) * nby : gb.nbx[i] * nby;
    } else {
      gb.job[i] = jobs[0];
      gb.nbx[i] = 1;
      gb.big[i] = 0;
    }
    gb.end[i] = total;
  }
  if (total <= 0) return DD_OK;
  bool any_ln = false;
  for (int i = 0; i < njobs; ++i) any_ln = any_ln || jobs[i].ln != nullptr;
  (void)any_ln;                                        // (the K-split tile handles the LayerNorm prologue as well)
  if (!g_gemm_ksplit) hipLaunchKernelGGL(k_gemm128_batch<false>, dim3(total), dim3(256), 0, st, gb);
  else hipLaunchKernelGGL(k_gemm128_batch<true>, dim3(total), dim3(256), 0, st, gb);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

int launch_gemm128(const GemmArgs& a, hipStream_t st) {
  if (a.rows <= 0 || a.ncols <= 0) return DD_OK;
  dim3 grid((a.rows + GT - 1) / GT, (a.ncols + GT - 1) / GT);
  if (!g_gemm_ksplit) hipLaunchKernelGGL(k_gemm128, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(k_gemm128_ks, grid, dim3(256), 0, st, a);
  DD_CHECK_LAUNCH();
  return DD_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM of the training step (nn.Linear backward, dW = dY^T . X):
//     C[o, i] = sum_r A[r, o] * X[r, i]        A = dY [rows, M] (M <= 128 output channels), X [rows, 128], C [M, 128]
// The contraction runs over the ROWS (up to ~10^5 edges / triplets) and the output is one 128 x 128 block, the opposite
// shape of the projection GEMM above: a workgroup takes a slab of `rch` rows, stages 32 rows of A and X per trip in LDS
// (row-major, pitch 160: the two k halves of a v_mfma_f32_32x32x2_f32 operand fetch hit disjoint bank halves), wave w owns
// columns 32 w .. 32 w + 31 of X and all MT 32-row tiles of the output, and writes its partial block; k_gemm_tn_reduce adds the
// slabs in a fixed order (no atomics: bitwise reproducible).  Exact fp32 (k-ordered fmaf chains), like the forward GEMM.
constexpr int TNP = 160;
template <int MT>
__global__ __launch_bounds__(256, 2) void k_gemm_tn(const float* __restrict__ A, int lda, int M, const float* __restrict__ X, int ldx,
                                                    long rows, int rch, float* __restrict__ part /*[slabs][MT*32][128]*/,
                                                    float* __restrict__ bpart /*[slabs][128] column sums of A (the bias gradient), or NULL*/) {
  __shared__ __attribute__((aligned(16))) float As[32 * TNP];
  __shared__ __attribute__((aligned(16))) float Bs[32 * TNP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hh = lane >> 5;
  const long r0 = (long)blockIdx.x * rch, r1 = (r0 + rch < rows) ? r0 + rch : rows;
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
  const bool a_vec = ((lda & 3) == 0) && ((M & 3) == 0) && ((reinterpret_cast<size_t>(A) & 15) == 0);
  float4 xv[4], av[MT];
  float4 asum = make_float4(0.f, 0.f, 0.f, 0.f);       // this thread's share of the column sums of A (its MT rows per trip share a column group)
  // the rows of trip rb: requested one trip ahead (they fly while the previous 32 rows are multiplied)
  auto fetch = [&](long rb) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256, r = i >> 5, c4 = (i & 31) * 4;
      const long gr = rb + r;
      xv[k] = gr < r1 ? *reinterpret_cast<const float4*>(X + gr * ldx + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      const int i = tid + k * 256, r = i / (MT * 8), c4 = (i % (MT * 8)) * 4;
      const long gr = rb + r;
      av[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < r1) {
        const float* src = A + gr * lda + c4;
        if (a_vec && c4 + 3 < M) av[k] = *reinterpret_cast<const float4*>(src);
        else {
          if (c4 < M) av[k].x = src[0];
          if (c4 + 1 < M) av[k].y = src[1];
          if (c4 + 2 < M) av[k].z = src[2];
          if (c4 + 3 < M) av[k].w = src[3];
        }
      }
    }
  };
  fetch(r0);
  for (long rb = r0; rb < r1; rb += 32) {
    __syncthreads();                                     // the previous trip's operand reads are done
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = tid + k * 256, r = i >> 5, c4 = (i & 31) * 4;
      *reinterpret_cast<float4*>(&Bs[r * TNP + c4]) = xv[k];
    }
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      const int i = tid + k * 256, r = i / (MT * 8), c4 = (i % (MT * 8)) * 4;
      *reinterpret_cast<float4*>(&As[r * TNP + c4]) = av[k];
      asum.x += av[k].x; asum.y += av[k].y; asum.z += av[k].z; asum.w += av[k].w;
    }
    __syncthreads();
    if (rb + 32 < r1) fetch(rb + 32);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float b = Bs[(2 * kk + hh) * TNP + 32 * wave + li];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float a = As[(2 * kk + hh) * TNP + 32 * mt + li];
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mt], 0, 0, 0);
      }
    }
  }
  float* dst = part + (long)blockIdx.x * (MT * 32) * 128;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;   // C/D map: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
      dst[(32 * mt + row) * 128 + 32 * wave + li] = acc[mt][r];
    }
  if (bpart != nullptr) {
    // column sums of this slab's rows of A: thread t owns column group t % (8 MT) in every trip; the 256 / (8 MT) threads of a
    // group are added in ascending thread order (fixed association), through the operand buffer (dead by now)
    constexpr int G = MT * 8;
    __syncthreads();
    reinterpret_cast<float4*>(As)[tid] = asum;
    __syncthreads();
    if (tid < G) {
      float4 t = reinterpret_cast<const float4*>(As)[tid];
#pragma unroll
      for (int j = 1; j < 256 / G; ++j) {
        const float4 u = reinterpret_cast<const float4*>(As)[tid + j * G];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      float* d = bpart + (long)blockIdx.x * 128 + 4 * tid;
      d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
    }
  }
}
__global__ __launch_bounds__(256) void k_gemm_tn_reduce(const float* __restrict__ part, int slabs, int mpad, int M, float* __restrict__ out,
                                                        int ldo, int accumulate, const float* __restrict__ bpart = nullptr,
                                                        float* __restrict__ bias_out = nullptr, int n_main = 0) {
  if (bpart != nullptr && (int)blockIdx.x >= n_main) {     // the bias gradient: M column sums, slabs dealt to the 4 waves as below
    __shared__ float redb[4][64];
    const int o = ((int)blockIdx.x - n_main) * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s = 0.f;
    if (o < M)
      for (int w = q; w < slabs; w += 4) s += bpart[(long)w * 128 + o];
    redb[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && o < M) bias_out[o] = ((redb[0][threadIdx.x] + redb[1][threadIdx.x]) + redb[2][threadIdx.x]) + redb[3][threadIdx.x];
    return;
  }
  // 64 output elements per workgroup, the slabs dealt to 4 waves (slab w -> wave w mod 4, ascending), the four partial sums
  // added in wave order: a fixed association, reproducible bit for bit
  __shared__ float red[4][64];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  float s = 0.f;
  if (e < M * 128) {
    const int o = e >> 7, i = e & 127;
    for (int w = q; w < slabs; w += 4) s += part[((long)w * mpad + o) * 128 + i];
  }
  red[q][threadIdx.x & 63] = s;
  __syncthreads();
  if (q == 0 && e < M * 128) {
    const float t = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    float* d = out + (long)(e >> 7) * ldo + (e & 127);
    *d = accumulate ? *d + t : t;
  }
}

int gemm_tn_rows_per_slab(long rows) {
  // ~512 slabs (two resident workgroups on each of the 256 CUs), at least 64 and at most 4096 rows each, whole 32-row trips
  long rch = ((rows + 511) / 512 + 31) / 32 * 32;
  if (rch < 64) rch = 64;
  return (int)rch;
}
}  // namespace dd

extern "C" int dd_gemm128(const float* X, int x_rows_per_b, long x_stride_b, int ldx, int rows, const float* W,
                          const float* bias, const float* ln, float* Y, int y_rows_per_b, long y_stride_b, int ldy,
                          int ncols, int accumulate, void* stream) {
  if (!X || !W || !Y || x_rows_per_b <= 0 || y_rows_per_b <= 0 || (ldx & 3) != 0) return DD_ERR_BAD_ARG;
  dd::GemmArgs a = dd::gemm_args(X, x_rows_per_b, x_stride_b, ldx, rows, W, bias, ln, Y, y_rows_per_b, y_stride_b, ldy, ncols, accumulate);
  a.dbg = dd::g_gemm_dbg;
  return dd::launch_gemm128(a, (hipStream_t)stream);
}

// nn.Linear backward, weight gradient (training step; ATen mm call sites of autograd behind models/common.py:85-105):
// out[M,128] (+)= A[rows,M]^T . X[rows,128].  `scratch` holds dd_gemm128_tn_scratch_floats(rows, M) floats.
extern "C" size_t dd_gemm128_tn_scratch_floats(long rows, int M) {
  if (rows <= 0 || M <= 0 || M > 128) return 0;
  const int rch = dd::gemm_tn_rows_per_slab(rows);
  const long slabs = (rows + rch - 1) / rch;
  const int mt = M <= 32 ? 1 : (M <= 64 ? 2 : 4);
  return (size_t)slabs * mt * 32 * 128 + (size_t)slabs * 128;      // (+ the slabs' column sums of A: dd_gemm128_tn_bias)
}
static int gemm128_tn_impl(const float* A, int lda, int M, const float* X, int ldx, long rows, float* scratch, float* out, int ldo,
                           int accumulate, float* bias_out, void* stream) {
  if (!A || !X || !scratch || !out || M <= 0 || M > 128 || rows <= 0 || (ldx & 3) != 0 || lda < M || ldo < 128 ||
      (reinterpret_cast<size_t>(X) & 15) != 0)
    return DD_ERR_BAD_ARG;
  const int rch = dd::gemm_tn_rows_per_slab(rows);
  const int slabs = (int)((rows + rch - 1) / rch);
  hipStream_t st = (hipStream_t)stream;
  const int mt = M <= 32 ? 1 : (M <= 64 ? 2 : 4), mpad = mt * 32;
  float* bpart = bias_out ? scratch + (size_t)slabs * mpad * 128 : nullptr;
  if (mt == 1) hipLaunchKernelGGL(dd::k_gemm_tn<1>, dim3(slabs), dim3(256), 0, st, A, lda, M, X, ldx, rows, rch, scratch, bpart);
  else if (mt == 2) hipLaunchKernelGGL(dd::k_gemm_tn<2>, dim3(slabs), dim3(256), 0, st, A, lda, M, X, ldx, rows, rch, scratch, bpart);
  else hipLaunchKernelGGL(dd::k_gemm_tn<4>, dim3(slabs), dim3(256), 0, st, A, lda, M, X, ldx, rows, rch, scratch, bpart);
  DD_CHECK_LAUNCH();
  const int n_main = (M * 128 + 63) / 64, n_bias = bias_out ? (M + 63) / 64 : 0;
  hipLaunchKernelGGL(dd::k_gemm_tn_reduce, dim3(n_main + n_bias), dim3(256), 0, st, scratch, slabs, mpad, M, out, ldo, accumulate,
                     (const float*)bpart, bias_out, n_main);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
extern "C" int dd_gemm128_tn(const float* A, int lda, int M, const float* X, int ldx, long rows, float* scratch, float* out, int ldo,
                             int accumulate, void* stream) {
  return gemm128_tn_impl(A, lda, M, X, ldx, rows, scratch, out, ldo, accumulate, nullptr, stream);
}
// ... and the bias gradient with it: bias_out[M] = column sums of A (= dY^T 1), formed from the rows the kernel stages anyway
// (ATen's dy.sum(0) was 640 launches and 9.5 ms of a training step)
extern "C" int dd_gemm128_tn_bias(const float* A, int lda, int M, const float* X, int ldx, long rows, float* scratch, float* out, int ldo,
                                  int accumulate, float* bias_out, void* stream) {
  if (!bias_out) return DD_ERR_BAD_ARG;
  return gemm128_tn_impl(A, lda, M, X, ldx, rows, scratch, out, ldo, accumulate, bias_out, stream);
}
