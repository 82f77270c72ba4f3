// Reverse-diffusion step kernels (models/decompdiff.py:601-689, models/transitions.py:65-161)
// and drift guidance with analytic gradients (utils/guidance_funcs.py:24-78).
//
//   k_step_rows   one wave per ligand atom / per bond row: type head second Linear
//                 (ShiftedSoftplus -> Linear, decompdiff.py:194-211), log_softmax, categorical
//                 posterior q(v_{t-1}|v_t, v0) in log space, Gumbel-argmax sample, trajectories.
//   k_step_pos    one thread per ligand coordinate: C0 posterior mean, drift, noise.
//   k_drift_*     gradients of the armsca_prox / clash energies at x_t.
// The current step index lives in device memory (step_counter) so that one captured hipGraph
// can be replayed for every step.
#include "dd_kernels.hpp"

namespace dd {

// Production noise (Philox4x32-10, counter = (row, row >> 32, t, stream id)): one uniform per (row, class) for the
// Gumbel draws -- classes 4k..4k+3 from the block of stream `sid | (k << 8)` (8 classes: two blocks, up to 23 classes of
// ligand_atom_mode 'full': six) -- and one
// Box-Muller normal per coordinate (stream 7).  Streams: 1 atom types, 2 bond types, 7 coordinates; distinct
// (row, t, stream) never share a counter.  `t` is the diffusion time index of the step (t_start - steps done), not the
// step number of the call: a chain resumed with start_step and the same seed continues the noise of the unsplit chain
// (tests/test_gpu_properties.py) instead of replaying the draws of its first steps.  dd_debug_philox exposes exactly these functions to the statistics test.
template <int NC>
__device__ __forceinline__ float philox_uniform(uint64_t seed, long row, int step, uint32_t sid, int c) {
  Philox ph(seed);
  uint32_t bits = 0u;
#pragma unroll
  for (int blk = 0; blk < (NC + 3) / 4; ++blk) {
    uint32_t r[4];
    ph.gen((uint32_t)row, (uint32_t)(row >> 32), (uint32_t)step, sid | ((uint32_t)blk << 8), r);
#pragma unroll
    for (int k = 0; k < 4; ++k) bits = (c == 4 * blk + k) ? r[k] : bits;
  }
  return u01(bits);
}
__device__ __forceinline__ float philox_normal(uint64_t seed, int idx, int step) {
  Philox ph(seed);
  uint32_t r[4];
  ph.gen((uint32_t)idx, 0u, (uint32_t)step, 7u, r);
  const float u1 = fmaxf(u01(r[0]), 5.9604645e-8f), u2 = u01(r[1]);
  return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}


// Run state in device memory (dd_sampler.step_counter, [4] int32: steps done, t_start, seed lo, seed hi; written by
// dd_sampler_reset): a captured step graph is replayed for every step AND re-used for the next chain of the same shape
// with another seed / start time -- nothing chain-specific is baked into its kernel arguments.
struct RunState { int step, t_start; uint64_t seed; };
__device__ __forceinline__ RunState load_run_state(const int32_t* __restrict__ rs, int counter_bias) {
  RunState r;
  r.step = rs[0] - counter_bias;
  r.t_start = rs[1];
  r.seed = (uint64_t)(uint32_t)rs[2] | ((uint64_t)(uint32_t)rs[3] << 32);
  return r;
}

__device__ __forceinline__ float log_add_exp(float a, float b) {
  float m = fmaxf(a, b);
  return m + logf(expf(a - m) + expf(b - m));
}

template <int NC>
__global__ __launch_bounds__(256) void k_step_rows(const StepRowsArgs a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.rows) return;
  const RunState rs = load_run_state(a.step_counter, a.counter_bias);
  const int step = rs.step, t = rs.t_start - step;
  // ShiftedSoftplus (common.py:66-72): softplus(x) - log 2, torch threshold 20
  float logit[NC];
  if (a.logits_in != nullptr) {                          // dd_reverse_step: the host supplies the head outputs
#pragma unroll
    for (int c = 0; c < NC; ++c) logit[c] = a.logits_in[row * NC + c];
  } else {
    float2 hv = *reinterpret_cast<const float2*>(a.hid + row * 128 + 2 * lane);
    hv.x = (hv.x > 20.f ? hv.x : log1pf(expf(hv.x))) - 0.6931471805599453f;
    hv.y = (hv.y > 20.f ? hv.y : log1pf(expf(hv.y))) - 0.6931471805599453f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float2 w = *reinterpret_cast<const float2*>(a.W2 + c * 128 + 2 * lane);
      logit[c] = wave_sum(fmaf(hv.y, w.y, hv.x * w.x)) + a.b2[c];
    }
  }
  if (lane != 0) return;
  // log_softmax
  float mx = logit[0];
#pragma unroll
  for (int c = 1; c < NC; ++c) mx = fmaxf(mx, logit[c]);
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) se += expf(logit[c] - mx);
  const float lse = mx + logf(se);
  const int tm1 = t - 1 < 0 ? 0 : t - 1;
  const float la_t = a.tab[t], l1ma_t = a.tab[a.T + t];
  const float lca = a.tab[2 * a.T + tm1], l1mca = a.tab[3 * a.T + tm1];
  const float* log_prior_tab = a.tab + 4 * a.T;         // [NC] log prior (uniform: -log NC), DiscreteTransition.prior_probs
  const int cur = a.state[row];
  float un[NC];
  float umax = -INFINITY;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float log_prior = log_prior_tab[c];
    const float lv0 = logit[c] - lse;
    const float lvt = (c == cur) ? 0.f : -69.07755278982137f;    // log(clamp(onehot, 1e-30))
    const float q0 = log_add_exp(lv0 + lca, l1mca + log_prior);  // q(v_{t-1} | v0)
    const float q1 = log_add_exp(lvt + la_t, l1ma_t + log_prior); // q(v_t | v_{t-1})
    un[c] = q0 + q1;
    umax = fmaxf(umax, un[c]);
    if (a.logits_out) a.logits_out[row * NC + c] = logit[c];
    if (a.traj_recon) a.traj_recon[((long)step * a.rows + row) * NC + c] = lv0;
  }
  float us = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) us += expf(un[c] - umax);
  const float ulse = umax + logf(us);
  // Gumbel-argmax (transitions.py:78-84)
  float u[NC];
  if (a.uniforms) {
#pragma unroll
    for (int c = 0; c < NC; ++c) u[c] = a.uniforms[((long)step * a.rows + row) * NC + c];
  } else {
#pragma unroll
    for (int c = 0; c < NC; ++c) u[c] = philox_uniform<NC>(rs.seed, row, t, a.stream_id, c);
  }
  int best = 0;
  float bestv = -INFINITY;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float lp = un[c] - ulse;
    const float g = -logf(-logf(u[c] + 1e-30f) + 1e-30f);
    const float s = g + lp;
    if (s > bestv) { bestv = s; best = c; }
    if (a.traj_prob) a.traj_prob[((long)step * a.rows + row) * NC + c] = lp;
  }
  a.state[row] = best;
  if (a.traj_state) a.traj_state[(long)step * a.rows + row] = best;
}

__global__ void k_step_pos(const StepPosArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = a.B * a.NL * 3;
  if (idx >= n) return;
  const RunState rs = load_run_state(a.step_counter, a.counter_bias);
  const int step = rs.step, t = rs.t_start - step;
  const int atom = idx / 3, c = idx % 3, b = atom / a.NL;
  const float xt = a.xt[idx];
  float x0;
  if (a.x0_prev != nullptr) {
    const int r = idx % (a.NL * 3);
    x0 = a.x0_prev[((long)b * (a.NP + a.NL) + a.NP) * 3 + r] + a.x0_dxe[idx] + a.x0_dxb[idx];
    a.x0_out[idx] = x0;
  } else {
    x0 = a.x0[idx];
  }
  float mean = a.tab_pos[t] * x0 + a.tab_pos[a.T + t] * xt;
  float g = 0.f;
  if (a.grad_a) g += a.scale_a ? a.grad_a[idx] * a.tab_score[t] : a.grad_a[idx];
  if (a.grad_c) g += a.scale_c ? a.grad_c[idx] * a.tab_score[t] : a.grad_c[idx];
  if (a.grad_r) g += a.scale_r ? a.grad_r[idx] * a.tab_score[t] : a.grad_r[idx];
  mean -= g;
  float e;
  if (a.eps) {
    e = a.eps[(long)step * n + idx];
  } else {
    e = philox_normal(rs.seed, idx, t);
  }
  const float nz = t == 0 ? 0.f : 1.f;
  const float nxt = mean + nz * expf(0.5f * a.tab_pos[2 * a.T + t]) * e * a.atom_std[idx];
  a.xt[idx] = nxt;
  if (a.traj_pos) a.traj_pos[(long)step * n + idx] = nxt + a.offset[b * 3 + c];
}

__global__ void k_advance(int32_t* step_counter) { *step_counter += 1; }
__global__ void k_reset_run_state(int32_t* rs, int t_start, uint32_t seed_lo, uint32_t seed_hi) {
  rs[0] = 0; rs[1] = t_start; rs[2] = (int32_t)seed_lo; rs[3] = (int32_t)seed_hi;
}

// ------------------------------------------------------------------------------ drift: armsca
// Per sample: d_arm = min over (arm atom a in arm, scaffold atom s) |x_a - x_s|;
// loss = sum_b mean_arms( relu(min_d - d) + relu(d - max_d) ) / B       (guidance_funcs.py:50-78)
// One workgroup per sample: 64 threads (NL <= 64: one wave, the reduction stays in registers) or 128 (NL <= 128: the two
// waves' winners meet in LDS).
__global__ __launch_bounds__(DD_NL_MAX) void k_drift_armsca(const float* __restrict__ pos, const int32_t* __restrict__ decomp,
                                                     int B, int NL, float min_d, float max_d, float* __restrict__ grad,
                                                     int accumulate, int norm_B) {
  __shared__ float px[DD_NL_MAX], py[DD_NL_MAX], pz[DD_NL_MAX];
  __shared__ int arm[DD_NL_MAX];
  __shared__ float gx[DD_NL_MAX], gy[DD_NL_MAX], gz[DD_NL_MAX];
  const int b = blockIdx.x, l = threadIdx.x;
  int my_arm = -2;
  if (l < NL) {
    px[l] = pos[((long)b * NL + l) * 3]; py[l] = pos[((long)b * NL + l) * 3 + 1]; pz[l] = pos[((long)b * NL + l) * 3 + 2];
    my_arm = decomp[(long)b * NL + l];
    arm[l] = my_arm;
    gx[l] = 0.f; gy[l] = 0.f; gz[l] = 0.f;
  }
  __syncthreads();
  // number of arms = max id + 1 (scatter_min output rows); a sample is "valid" iff it has arm and scaffold atoms
  int n_arms = 0, n_sca = 0, n_armatoms = 0;
  for (int i = 0; i < NL; ++i) {
    n_arms = arm[i] + 1 > n_arms ? arm[i] + 1 : n_arms;
    n_sca += arm[i] == -1;
    n_armatoms += arm[i] >= 0;
  }
  if (n_sca > 0 && n_armatoms > 0) {
    // One arm at a time, one lane per scaffold atom (the serial form -- one lane per ARM walking all NL x NL pairs -- kept
    // two lanes busy for 54 us per step).  Same arithmetic and the same winners: lane s finds the nearest atom of the arm
    // (ascending i, strict <: the first minimum, as scatter_min), then the wave takes the (distance, s) lexicographic
    // minimum -- the first scaffold atom at the smallest distance, which is what the ascending strict-< loop over s picked.
    for (int a = 0; a < n_arms; ++a) {
      float bcol = INFINITY;
      int bca = -1;
      if (l < NL && my_arm == -1) {
        for (int i = 0; i < NL; ++i) {
          if (arm[i] != a) continue;
          float dx = px[i] - px[l], dy = py[i] - py[l], dz = pz[i] - pz[l];
          float d = sqrtf(dx * dx + dy * dy + dz * dz);
          if (d < bcol) { bcol = d; bca = i; }
        }
      }
      unsigned key = __float_as_uint(bcol);                // d >= 0 (or +inf): unsigned order = float order
      int who = l;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const unsigned k2 = (unsigned)__shfl_xor((int)key, off);
        const int w2 = __shfl_xor(who, off);
        if (k2 < key || (k2 == key && w2 < who)) { key = k2; who = w2; }
      }
      int ba = __shfl(bca, who & 63);                      // this wave's winner and its arm atom
      if (blockDim.x > 64) {                               // 65 .. 128 ligand atoms: second wave's winner through LDS
        __shared__ unsigned wk[2];
        __shared__ int ww[2], wa[2];
        __syncthreads();                                   // (previous arm's readers are done)
        if ((l & 63) == 0) { wk[l >> 6] = key; ww[l >> 6] = who; wa[l >> 6] = ba; }
        __syncthreads();
        const int pick = (wk[1] < wk[0] || (wk[1] == wk[0] && ww[1] < ww[0])) ? 1 : 0;
        key = wk[pick]; who = ww[pick]; ba = wa[pick];
      }
      const float best = __uint_as_float(key);
      const int bs = who;
      if (l == 0 && ba >= 0) {
        float coef = 0.f;
        if (min_d - best > 0.f) coef -= 1.f;
        if (best - max_d > 0.f) coef += 1.f;
        coef /= ((float)n_arms * (float)norm_B);          // the loss is averaged over the caller's whole batch
        if (coef != 0.f) {
          float dx = px[ba] - px[bs], dy = py[ba] - py[bs], dz = pz[ba] - pz[bs];
          float inv = 1.0f / best;
          // (products rounded before the add, as the atomic adds of the serial form took them; arms in ascending order,
          //  as its lane-ordered atomics)
          const float tx = __fmul_rn(__fmul_rn(coef, dx), inv), ty = __fmul_rn(__fmul_rn(coef, dy), inv), tz = __fmul_rn(__fmul_rn(coef, dz), inv);
          const float ux = __fmul_rn(__fmul_rn(-coef, dx), inv), uy = __fmul_rn(__fmul_rn(-coef, dy), inv), uz = __fmul_rn(__fmul_rn(-coef, dz), inv);
          gx[ba] = __fadd_rn(gx[ba], tx); gy[ba] = __fadd_rn(gy[ba], ty); gz[ba] = __fadd_rn(gz[ba], tz);
          gx[bs] = __fadd_rn(gx[bs], ux); gy[bs] = __fadd_rn(gy[bs], uy); gz[bs] = __fadd_rn(gz[bs], uz);
        }
      }
    }
  }
  __syncthreads();
  if (l < NL) {
    float* g = grad + ((long)b * NL + l) * 3;
    if (accumulate) { g[0] += gx[l]; g[1] += gy[l]; g[2] += gz[l]; }
    else { g[0] = gx[l]; g[1] = gy[l]; g[2] = gz[l]; }
  }
}

// ------------------------------------------------------------------------------ drift: arms_repul
// compute_batch_arms_repul_loss (guidance_funcs.py:81-118), analytic gradient.  Per sample, for every arm pair a1 <= a2 (the
// reference's loop INCLUDES a1 == a2) with atoms on both sides, d_pq = |x_p - x_q| over p in a1, q in a2:
//   mode 1 'min': relu(max_d - min_pq d_pq)          -> d/dx_p = -(x_p - x_q)/d at the arg-min pair (first minimum in (p, q)
//                 order; torch splits the gradient evenly over exact ties), nothing for a1 == a2 (the minimum is a zero on
//                 the diagonal and torch's norm has a zero subgradient there);
//   mode 2 'all': mean_pq relu(max_d - d_pq)         -> d/dx_p = -(x_p - x_q) / (d n1 n2) for every pair inside max_d; the
//                 pairs of one arm appear as (p, q) and (q, p): twice the weight, and the diagonal (d = 0) gives nothing.
// The sum over pairs and samples is divided by the batch size.  The clamp passes its gradient where max_d - d >= 0
// (torch.clamp backward is inclusive).  One workgroup per sample, one thread per ligand atom.
__global__ __launch_bounds__(DD_NL_MAX) void k_drift_arms_repul(const float* __restrict__ pos, const int32_t* __restrict__ decomp, int B,
                                                                int NL, float max_d, int mode, float* __restrict__ grad,
                                                                int accumulate, int norm_B) {
  __shared__ float px[DD_NL_MAX], py[DD_NL_MAX], pz[DD_NL_MAX];
  __shared__ int arm[DD_NL_MAX];
  __shared__ unsigned long long best[DD_NL_MAX];           // mode 1: per atom p of a1 the key (bits(d), q) of its nearest atom of a2
  __shared__ float gx[DD_NL_MAX], gy[DD_NL_MAX], gz[DD_NL_MAX];
  const int b = blockIdx.x, l = threadIdx.x;
  int my_arm = -2;
  if (l < NL) {
    px[l] = pos[((long)b * NL + l) * 3]; py[l] = pos[((long)b * NL + l) * 3 + 1]; pz[l] = pos[((long)b * NL + l) * 3 + 2];
    my_arm = decomp[(long)b * NL + l];
    arm[l] = my_arm;
    gx[l] = 0.f; gy[l] = 0.f; gz[l] = 0.f;
  }
  __syncthreads();
  int n_arms = 0;                                          // mask.max() + 1 (guidance_funcs.py:99)
  for (int i = 0; i < NL; ++i) n_arms = arm[i] + 1 > n_arms ? arm[i] + 1 : n_arms;
  const float inv_B = 1.0f / (float)norm_B;
  if (mode == 2) {
    // every atom gathers its own gradient: sum over the atoms q of every arm (its own included), ascending q
    if (l < NL && my_arm >= 0) {
      int n_mine = 0;
      for (int i = 0; i < NL; ++i) n_mine += arm[i] == my_arm;
      float ax = 0.f, ay = 0.f, az = 0.f;
      for (int a2 = 0; a2 < n_arms; ++a2) {
        int n2 = 0;
        for (int i = 0; i < NL; ++i) n2 += arm[i] == a2;
        if (n2 == 0) continue;
        const float w = (a2 == my_arm ? 2.0f : 1.0f) / ((float)n_mine * (float)n2) * inv_B;
        for (int q = 0; q < NL; ++q) {
          if (arm[q] != a2 || q == l) continue;
          const float dx = px[l] - px[q], dy = py[l] - py[q], dz = pz[l] - pz[q];
          const float d = sqrtf(dx * dx + dy * dy + dz * dz);
          if (max_d - d >= 0.f && d > 0.f) {
            const float c = -w / d;
            ax = fmaf(c, dx, ax); ay = fmaf(c, dy, ay); az = fmaf(c, dz, az);
          }
        }
      }
      gx[l] = ax; gy[l] = ay; gz[l] = az;
    }
  } else {
    for (int a1 = 0; a1 < n_arms; ++a1) {
      for (int a2 = a1 + 1; a2 < n_arms; ++a2) {           // (a1 == a2: zero gradient, see above)
        unsigned long long key = ~0ull;
        if (l < NL && my_arm == a1) {
          for (int q = 0; q < NL; ++q) {
            if (arm[q] != a2) continue;
            const float dx = px[l] - px[q], dy = py[l] - py[q], dz = pz[l] - pz[q];
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);
            const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)q;   // d >= 0: bit order = float order
            key = k < key ? k : key;
          }
        }
        if (l < NL) best[l] = key;
        __syncthreads();
        if (l == 0) {
          unsigned long long bk = ~0ull;
          int bp = -1;
          for (int p = 0; p < NL; ++p)
            if (best[p] < bk) { bk = best[p]; bp = p; }     // strict <: the first p at the smallest distance
          if (bp >= 0 && (unsigned)(bk >> 32) != 0xffffffffu) {
            const float d = __uint_as_float((unsigned)(bk >> 32));
            const int q = (int)(unsigned)(bk & 0xffffffffull);
            if (max_d - d >= 0.f && d > 0.f) {
              const float c = -inv_B / d;
              const float dx = px[bp] - px[q], dy = py[bp] - py[q], dz = pz[bp] - pz[q];
              gx[bp] = fmaf(c, dx, gx[bp]); gy[bp] = fmaf(c, dy, gy[bp]); gz[bp] = fmaf(c, dz, gz[bp]);
              gx[q] = fmaf(-c, dx, gx[q]); gy[q] = fmaf(-c, dy, gy[q]); gz[q] = fmaf(-c, dz, gz[q]);
            }
          }
        }
        __syncthreads();
      }
    }
  }
  __syncthreads();
  if (l < NL) {
    float* g = grad + ((long)b * NL + l) * 3;
    if (accumulate) { g[0] += gx[l]; g[1] += gy[l]; g[2] += gz[l]; }
    else { g[0] = gx[l]; g[1] = gy[l]; g[2] = gz[l]; }
  }
}

// ------------------------------------------------------------------------------ drift: clash
// G_i = -sigma * log(1e-3 + sum_j exp(-|p_j - y_i|^2 / sigma)), y_i = x_i + offset_b
// loss = sum_b mean_i relu(gamma - G_i)   (guidance_funcs.py:24-42)
// dloss/dx_i = -(1/NL) [G_i < gamma] * 2/(1e-3+S) * sum_j e_j (y_i - p_j)
// One workgroup (256 threads) per ligand atom.
__global__ __launch_bounds__(256) void k_drift_clash(const float* __restrict__ pos, const float* __restrict__ offset,
                                                     const float* __restrict__ prot, int B, int NL, int NF, float sigma,
                                                     float gamma, float* __restrict__ grad, int accumulate,
                                                     const int32_t* __restrict__ nl_real) {
  __shared__ float red[4][4];
  const int atom = blockIdx.x, b = atom / NL;
  const int nlb = nl_real ? nl_real[b] : NL;             // the loss is the mean over the sample's REAL ligand atoms
  if (atom % NL >= nlb) {                                // padding atom of a heterogeneous batch
    if (threadIdx.x < 3 && !accumulate) grad[(long)atom * 3 + threadIdx.x] = 0.f;
    return;
  }
  const float yx = pos[(long)atom * 3] + offset[b * 3], yy = pos[(long)atom * 3 + 1] + offset[b * 3 + 1],
              yz = pos[(long)atom * 3 + 2] + offset[b * 3 + 2];
  const float* pb = prot + (long)b * NF * 3;
  float S = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
  const float inv_sigma = 1.0f / sigma;
  for (int j = threadIdx.x; j < NF; j += 256) {
    float dx = yx - pb[3 * j], dy = yy - pb[3 * j + 1], dz = yz - pb[3 * j + 2];
    float e = expf(-(dx * dx + dy * dy + dz * dz) * inv_sigma);
    S += e; sx = fmaf(e, dx, sx); sy = fmaf(e, dy, sy); sz = fmaf(e, dz, sz);
  }
  S = wave_sum(S); sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[wave][0] = S; red[wave][1] = sx; red[wave][2] = sy; red[wave][3] = sz; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float St = red[0][0] + red[1][0] + red[2][0] + red[3][0];
    float v = red[0][1 + threadIdx.x] + red[1][1 + threadIdx.x] + red[2][1 + threadIdx.x] + red[3][1 + threadIdx.x];
    float G = -sigma * logf(1e-3f + St);
    float g = 0.f;
    if (gamma - G > 0.f) g = -(1.0f / (float)nlb) * 2.0f / (1e-3f + St) * v;
    float* dst = grad + (long)atom * 3 + threadIdx.x;
    *dst = accumulate ? *dst + g : g;
  }
}


// ---------------------------------------------------------------------------------- one launch per reverse step
// Bond rows, atom rows and the coordinate update are independent given the head activations, so they share one
// launch (block ranges); the step counter is advanced by a 1-thread launch behind it (a last-block ticket was
// tried: ~1800 atomics on one address serialise to 45 us).  A row is one wave: the
// second head Linear uses the whole wave, the per-class posterior / Gumbel arithmetic runs one class per lane with
// the cross-class sums taken in class order through v_readlane (bit-identical to the serial k_step_rows).
template <int NC>
__device__ __forceinline__ void step_row(const StepRowsArgs& a, const long row, const int lane, const RunState rs) {
  const int step = rs.step, t = rs.t_start - step;
  float2 hv = *reinterpret_cast<const float2*>(a.hid + row * 128 + 2 * lane);
  hv.x = (hv.x > 20.f ? hv.x : log1pf(expf(hv.x))) - 0.6931471805599453f;
  hv.y = (hv.y > 20.f ? hv.y : log1pf(expf(hv.y))) - 0.6931471805599453f;
  float logit[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float2 w = *reinterpret_cast<const float2*>(a.W2 + c * 128 + 2 * lane);
    logit[c] = wave_sum(fmaf(hv.y, w.y, hv.x * w.x)) + a.b2[c];
  }
  const int c = lane < NC ? lane : NC - 1;               // lanes >= NC shadow the last class (results unused)
  float mine = logit[0], mx = logit[0];
#pragma unroll
  for (int k = 1; k < NC; ++k) { mine = (c == k) ? logit[k] : mine; mx = fmaxf(mx, logit[k]); }
  const float e = expf(mine - mx);
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < NC; ++k) se += lane_bcast(e, k);
  const float lse = mx + logf(se);
  const int tm1 = t - 1 < 0 ? 0 : t - 1;
  const float la_t = a.tab[t], l1ma_t = a.tab[a.T + t];
  const float lca = a.tab[2 * a.T + tm1], l1mca = a.tab[3 * a.T + tm1];
  const float log_prior = a.tab[4 * a.T + c];            // log prior of this lane's class (uniform: -log NC)
  const int cur = a.state[row];
  const float lv0 = mine - lse;
  const float lvt = (c == cur) ? 0.f : -69.07755278982137f;      // log(clamp(onehot, 1e-30))
  const float q0 = log_add_exp(lv0 + lca, l1mca + log_prior);    // q(v_{t-1} | v0)
  const float q1 = log_add_exp(lvt + la_t, l1ma_t + log_prior);  // q(v_t | v_{t-1})
  const float un = q0 + q1;
  float umax = -INFINITY;
#pragma unroll
  for (int k = 0; k < NC; ++k) umax = fmaxf(umax, lane_bcast(un, k));
  const float ue = expf(un - umax);
  float us = 0.f;
#pragma unroll
  for (int k = 0; k < NC; ++k) us += lane_bcast(ue, k);
  const float ulse = umax + logf(us);
  float u;
  if (a.uniforms) {
    u = a.uniforms[((long)step * a.rows + row) * NC + c];
  } else {
    u = philox_uniform<NC>(rs.seed, row, t, a.stream_id, c);
  }
  const float lp = un - ulse;
  const float sc = -logf(-logf(u + 1e-30f) + 1e-30f) + lp;      // Gumbel-argmax (transitions.py:78-84)
  int best = 0;
  float bestv = -INFINITY;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const float sk = lane_bcast(sc, k);
    if (sk > bestv) { bestv = sk; best = k; }
  }
  if (lane < NC) {
    if (a.logits_out) a.logits_out[row * NC + c] = mine;
    if (a.traj_recon) a.traj_recon[((long)step * a.rows + row) * NC + c] = lv0;
    if (a.traj_prob) a.traj_prob[((long)step * a.rows + row) * NC + c] = lp;
  }
  if (lane == 0) {
    a.state[row] = best;
    if (a.traj_state) a.traj_state[(long)step * a.rows + row] = best;
  }
}

__device__ __forceinline__ void step_pos_elem(const StepPosArgs& a, const int idx, const RunState rs) {
  const int n = a.B * a.NL * 3;
  if (idx >= n) return;
  const int step = rs.step, t = rs.t_start - step;
  const int atom = idx / 3, c = idx % 3, b = atom / a.NL;
  const float xt = a.xt[idx];
  float x0;
  if (a.x0_prev != nullptr) {
    const int r = idx % (a.NL * 3);
    x0 = a.x0_prev[((long)b * (a.NP + a.NL) + a.NP) * 3 + r] + a.x0_dxe[idx] + a.x0_dxb[idx];
    a.x0_out[idx] = x0;
  } else {
    x0 = a.x0[idx];
  }
  float mean = a.tab_pos[t] * x0 + a.tab_pos[a.T + t] * xt;
  float g = 0.f;
  if (a.grad_a) g += a.scale_a ? a.grad_a[idx] * a.tab_score[t] : a.grad_a[idx];
  if (a.grad_c) g += a.scale_c ? a.grad_c[idx] * a.tab_score[t] : a.grad_c[idx];
  if (a.grad_r) g += a.scale_r ? a.grad_r[idx] * a.tab_score[t] : a.grad_r[idx];
  mean -= g;
  float e;
  if (a.eps) {
    e = a.eps[(long)step * n + idx];
  } else {
    e = philox_normal(rs.seed, idx, t);
  }
  const float nz = t == 0 ? 0.f : 1.f;
  const float nxt = mean + nz * expf(0.5f * a.tab_pos[2 * a.T + t]) * e * a.atom_std[idx];
  a.xt[idx] = nxt;
  if (a.traj_pos) a.traj_pos[(long)step * n + idx] = nxt + a.offset[b * 3 + c];
}

template <int NV>
__global__ __launch_bounds__(256) void k_step_all(const StepRowsArgs rb, const StepRowsArgs rv, const StepPosArgs p, int nb_b, int nb_v) {
  const int blk = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const RunState rs = load_run_state(rb.step_counter, rb.counter_bias);
  if (blk < nb_b) {
    const long row = (long)blk * 4 + wave;
    if (row < rb.rows) step_row<DD_NUM_B>(rb, row, lane, rs);
  } else if (blk < nb_b + nb_v) {
    const long row = (long)(blk - nb_b) * 4 + wave;
    if (row < rv.rows) step_row<NV>(rv, row, lane, rs);
  } else {
    step_pos_elem(p, (blk - nb_b - nb_v) * 256 + threadIdx.x, rs);
  }
}

int launch_step_all(const StepRowsArgs& rb, const StepRowsArgs& rv, const StepPosArgs& p, hipStream_t st) {
  if (rb.NC != DD_NUM_B || (rv.NC != 8 && rv.NC != 13 && rv.NC != 23)) return DD_ERR_UNSUPPORTED_SHAPE;
  const int nb_b = (rb.rows + 3) / 4, nb_v = (rv.rows + 3) / 4, nb_p = (p.B * p.NL * 3 + 255) / 256;
  const dim3 grid(nb_b + nb_v + nb_p);
  // atom classes: 8 (ligand_atom_mode 'basic'), 13 ('add_aromatic'), 23 ('full') -- utils/transforms.py:15-64,138-151
  if (rv.NC == 8) hipLaunchKernelGGL(k_step_all<8>, grid, dim3(256), 0, st, rb, rv, p, nb_b, nb_v);
  else if (rv.NC == 13) hipLaunchKernelGGL(k_step_all<13>, grid, dim3(256), 0, st, rb, rv, p, nb_b, nb_v);
  else hipLaunchKernelGGL(k_step_all<23>, grid, dim3(256), 0, st, rb, rv, p, nb_b, nb_v);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

int launch_step_rows(const StepRowsArgs& a, hipStream_t st) {
  if (a.rows <= 0) return DD_OK;
  dim3 grid((a.rows + 3) / 4);
  if (a.NC == 8) hipLaunchKernelGGL(k_step_rows<8>, grid, dim3(256), 0, st, a);
  else if (a.NC == 13) hipLaunchKernelGGL(k_step_rows<13>, grid, dim3(256), 0, st, a);
  else if (a.NC == 23) hipLaunchKernelGGL(k_step_rows<23>, grid, dim3(256), 0, st, a);
  else if (a.NC == DD_NUM_B) hipLaunchKernelGGL(k_step_rows<DD_NUM_B>, grid, dim3(256), 0, st, a);
  else return DD_ERR_UNSUPPORTED_SHAPE;
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_step_pos(const StepPosArgs& a, hipStream_t st) {
  int n = a.B * a.NL * 3;
  hipLaunchKernelGGL(k_step_pos, dim3((n + 255) / 256), dim3(256), 0, st, a);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_reset_run_state(int32_t* rs, int t_start, uint64_t seed, hipStream_t st) {
  hipLaunchKernelGGL(k_reset_run_state, dim3(1), dim3(1), 0, st, rs, t_start, (uint32_t)seed, (uint32_t)(seed >> 32));
  DD_CHECK_LAUNCH();
  return DD_OK;
}
int launch_advance(int32_t* ctr, hipStream_t st) {
  hipLaunchKernelGGL(k_advance, dim3(1), dim3(1), 0, st, ctr);
  DD_CHECK_LAUNCH();
  return DD_OK;
}

}  // namespace dd

namespace dd {
__global__ __launch_bounds__(256) void k_debug_philox(uint64_t seed, int step, long rows, int kind, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (kind == 7) { if (i < rows) out[i] = philox_normal(seed, (int)i, step); return; }
  const int NC = kind == 1 ? DD_NUM_V : DD_NUM_B;
  if (i >= rows * NC) return;
  const long row = i / NC; const int c = (int)(i % NC);
  out[i] = kind == 1 ? philox_uniform<DD_NUM_V>(seed, row, step, 1u, c) : philox_uniform<DD_NUM_B>(seed, row, step, 2u, c);
}
int launch_drift_armsca(const float* lig_pos, const int32_t* decomp_index, int B, int NL, float min_d, float max_d,
                        float* grad, int accumulate, int norm_B, hipStream_t st) {
  hipLaunchKernelGGL(k_drift_armsca, dim3(B), dim3(NL > 64 ? 128 : 64), 0, st, lig_pos, decomp_index, B, NL, min_d, max_d, grad, accumulate,
                     norm_B > 0 ? norm_B : B);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
}  // namespace dd

extern "C" int dd_drift_armsca(const float* lig_pos, const int32_t* decomp_index, int B, int NL, float min_d,
                               float max_d, float* grad, int accumulate, void* stream) {
  if (!lig_pos || !decomp_index || !grad || B <= 0 || NL <= 0) return DD_ERR_BAD_ARG;
  if (NL > DD_NL_MAX) return DD_ERR_UNSUPPORTED_SHAPE;
  return dd::launch_drift_armsca(lig_pos, decomp_index, B, NL, min_d, max_d, grad, accumulate, B, (hipStream_t)stream);
}

namespace dd {
int launch_drift_arms_repul(const float* lig_pos, const int32_t* decomp_index, int B, int NL, float max_d, int mode, float* grad,
                            int accumulate, int norm_B, hipStream_t st) {
  hipLaunchKernelGGL(k_drift_arms_repul, dim3(B), dim3(DD_NL_MAX), 0, st, lig_pos, decomp_index, B, NL, max_d, mode, grad, accumulate,
                     norm_B > 0 ? norm_B : B);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
}  // namespace dd

extern "C" int dd_drift_arms_repul(const float* lig_pos, const int32_t* decomp_index, int B, int NL, float max_d, int mode,
                                   float* grad, int accumulate, void* stream) {
  if (!lig_pos || !decomp_index || !grad || B <= 0 || NL <= 0 || (mode != 1 && mode != 2)) return DD_ERR_BAD_ARG;
  if (NL > DD_NL_MAX) return DD_ERR_UNSUPPORTED_SHAPE;
  return dd::launch_drift_arms_repul(lig_pos, decomp_index, B, NL, max_d, mode, grad, accumulate, B, (hipStream_t)stream);
}

namespace dd {
int launch_drift_clash(const float* lig_pos, const float* offset, const float* full_protein_pos, int B, int NL, int NF,
                       float sigma, float gamma, float* grad, int accumulate, const int32_t* nl_real, hipStream_t st) {
  hipLaunchKernelGGL(k_drift_clash, dim3(B * NL), dim3(256), 0, st, lig_pos, offset, full_protein_pos, B, NL, NF, sigma, gamma,
                     grad, accumulate, nl_real);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
}  // namespace dd

extern "C" int dd_drift_clash(const float* lig_pos, const float* offset, const float* full_protein_pos, int B, int NL,
                              int NF, float sigma, float gamma, float* grad, int accumulate, void* stream) {
  if (!lig_pos || !offset || !full_protein_pos || !grad || B <= 0 || NL <= 0 || NF <= 0) return DD_ERR_BAD_ARG;
  return dd::launch_drift_clash(lig_pos, offset, full_protein_pos, B, NL, NF, sigma, gamma, grad, accumulate, nullptr,
                                (hipStream_t)stream);
}

// Test aid: the production noise of one step, exactly as the step kernels draw it.  kind 1: uniforms [rows,8] of the
// atom-type stream, 2: uniforms [rows,5] of the bond-type stream, 7: normals [rows] of the coordinate stream.
extern "C" int dd_debug_philox(uint64_t seed, int step, long rows, int kind, float* out, void* stream) {
  if (!out || rows <= 0 || (kind != 1 && kind != 2 && kind != 7)) return DD_ERR_BAD_ARG;
  const long n = kind == 7 ? rows : rows * (kind == 1 ? DD_NUM_V : DD_NUM_B);
  hipLaunchKernelGGL(dd::k_debug_philox, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, seed, step, rows, kind, out);
  DD_CHECK_LAUNCH();
  return DD_OK;
}
