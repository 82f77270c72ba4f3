"""Differentiable path of the score network and the training objective.

The sampling loop runs hand-fused HIP kernels that have no backward.  Training needs gradients for all 616 tensors,
so this module states the same network with autograd: dense layers are torch (rocBLAS) ops, the graph ops are the HIP
kernels of the op-level C ABI wrapped in ``torch.autograd.Function`` s with analytic backward passes that again run on
the HIP kernels (``scatter_sum`` <-> gather, ``scatter_softmax`` <-> y * (g - scatter_sum(y * g)[index])), and the
kNN graph comes from ``dd_knn``.  Like the fused kernels it uses the exact first-Linear factorisation
(W.[a;b;c] = W_a a + W_b b + W_c c: per-node / per-bond projection tables that the edges gather) instead of materialising
the 340- / 384- / 437-wide concatenations of the reference (uni_transformer_edge.py:48,148-149,194), which cuts the
activation memory autograd keeps by ~3x; everything else follows the reference layer for layer:

    forward            models/decompdiff.py:213-351 -> UniTransformerO2TwoUpdateGeneralBond (uni_transformer_edge.py:394-443)
    get_diffusion_loss models/decompdiff.py:419-550 (C0 parameterisation, 'mse' position loss, categorical KL for atom
                       and bond types, bond diffusion)

Everything runs on the HIP device; CPU tensors raise (no fallback).  ``DecompScorePosNet3D.get_diffusion_loss`` calls
into here; with autograd disabled (validation, scripts/train_diffusion_decomp.py validate()) the network output comes
from the fused ``dd_forward`` instead.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import os

import torch
import torch.nn.functional as F

from . import functional as FN
from . import hip_lib

GAUSS_OFFSETS = [0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10]
H, NH = 128, 16


# --------------------------------------------------------------------------------------------------------------------
# graph ops with autograd (forward and backward on the HIP kernels)
# --------------------------------------------------------------------------------------------------------------------
class _ScatterSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, plan):
        ctx.plan = plan
        return FN.scatter_sum(src, plan, dim=0, dim_size=plan.n)

    @staticmethod
    def backward(ctx, g):
        return g.index_select(0, ctx.plan.raw), None


class _ScatterSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src, plan):
        y = FN.scatter_softmax(src, plan, dim=0, dim_size=plan.n)
        ctx.save_for_backward(y)
        ctx.plan = plan
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        yg = y * g
        return yg - y * FN.scatter_sum(yg, ctx.plan, dim=0, dim_size=ctx.plan.n).index_select(0, ctx.plan.raw), None


class _Gather(torch.autograd.Function):
    """rows = table[plan.raw]; the backward is a segment sum on the HIP kernel (ATen's index_select backward is an atomic
    index_add_, advanced indexing a sort-based index_put: 17 + 10 ms of a 105 ms step before)."""

    @staticmethod
    def forward(ctx, table, plan):
        ctx.plan = plan
        ctx.rows = table.size(0)
        return table.index_select(0, plan.raw)

    @staticmethod
    def backward(ctx, g):
        if ctx.rows != ctx.plan.n:
            raise RuntimeError("gather plan built for another table height")
        return FN.scatter_sum(g.contiguous(), ctx.plan, dim=0, dim_size=ctx.plan.n), None


def seg_plan(index, dim_size, check=True):
    """One SegmentPlan per (index vector, table height) and network call: checks, sort (if needed) and segment pointers once.
    check=False: no host round trip (an index produced on the device inside a captured step, known to be in range)."""
    plan = FN.SegmentPlan(index, dim_size, check=check)
    plan.raw = index
    return plan


def scatter_sum(src, plan, dim_size=None):
    return _ScatterSum.apply(src, plan)


def scatter_softmax(src, plan, dim_size=None):
    return _ScatterSoftmax.apply(src, plan)


def gather(table, plan):
    return _Gather.apply(table, plan)


class _Linear128(torch.autograd.Function):
    """y = x W^T + b for K = 128 inputs on the library's own fp32 MFMA GEMMs, forward and backward: dd_gemm128 for y and for
    dX = dY W (the transposed weight as its W operand; 128 outputs), dd_gemm128_tn for dW = dY^T X (row slabs reduced in a fixed
    order).  No ATen GEMM behind the hidden-width layers of the training step (models/common.py:85-105)."""

    @staticmethod
    def forward(ctx, x, W, b):
        lib = hip_lib.load()
        xc = x.contiguous()
        Wc = W.contiguous()
        rows, out = xc.size(0), Wc.size(0)
        y = torch.empty(rows, out, device=x.device, dtype=torch.float32)
        bc = b.contiguous() if b is not None else None
        hip_lib.check(lib.dd_gemm128(hip_lib.ptr(xc), rows, 0, 128, rows, hip_lib.ptr(Wc), hip_lib.ptr(bc), None, hip_lib.ptr(y), rows, 0,
                                     out, out, 0, hip_lib.stream_ptr(x.device)), "dd_gemm128")
        ctx.save_for_backward(xc, Wc)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, Wc = ctx.saved_tensors
        lib = hip_lib.load()
        dy = dy.contiguous()
        rows, out = dy.shape
        st = hip_lib.stream_ptr(dy.device)
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            if out == 128:
                Wt = Wc.t().contiguous()                                   # [in = 128 "columns", K = out = 128]
                dx = torch.empty(rows, 128, device=dy.device, dtype=torch.float32)
                hip_lib.check(lib.dd_gemm128(hip_lib.ptr(dy), rows, 0, 128, rows, hip_lib.ptr(Wt), None, None, hip_lib.ptr(dx), rows, 0,
                                             128, 128, 0, st), "dd_gemm128")
            else:                                                          # narrow heads (16 / 8 / 5 outputs): K is not 128
                dx = dy @ Wc
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dW = torch.empty(out, 128, device=dy.device, dtype=torch.float32)
            scratch = torch.empty(int(lib.dd_gemm128_tn_scratch_floats(rows, out)), device=dy.device, dtype=torch.float32)
            if want_db:                                    # the bias gradient (column sums of dY) from the same launch pair
                db = torch.empty(out, device=dy.device, dtype=torch.float32)
                hip_lib.check(lib.dd_gemm128_tn_bias(hip_lib.ptr(dy), out, out, hip_lib.ptr(xc), 128, rows, hip_lib.ptr(scratch),
                                                     hip_lib.ptr(dW), 128, 0, hip_lib.ptr(db), st), "dd_gemm128_tn_bias")
            else:
                hip_lib.check(lib.dd_gemm128_tn(hip_lib.ptr(dy), out, out, hip_lib.ptr(xc), 128, rows, hip_lib.ptr(scratch),
                                                hip_lib.ptr(dW), 128, 0, st), "dd_gemm128_tn")
        if want_db and db is None:
            db = dy.sum(0)
        return dx, dW, db


class _LinearFeat(torch.autograd.Function):
    """y = f W^T for a narrow feature block f [rows, Kf] (Kf = 84 type x Gaussian columns, 20 Gaussians, 13 angle codes) into the
    128 hidden channels.  The forward is a thin ATen product (K = Kf); the backward -- where the contraction runs over 10^4-10^5
    rows -- is on the library's kernels: df = dY W through dd_gemm128 (K = 128, Kf columns), dW^T = f^T dY through dd_gemm128_tn."""

    @staticmethod
    def forward(ctx, f, W):
        fc, Wc = f.contiguous(), W.contiguous()
        ctx.save_for_backward(fc, Wc)
        return fc @ Wc.t()

    @staticmethod
    def backward(ctx, dy):
        fc, Wc = ctx.saved_tensors
        lib = hip_lib.load()
        dy = dy.contiguous()
        rows, kf = fc.shape
        st = hip_lib.stream_ptr(dy.device)
        df = dW = None
        if ctx.needs_input_grad[0]:
            Wt = Wc.t().contiguous()                                       # [Kf "columns", K = 128]
            df = torch.empty(rows, kf, device=dy.device, dtype=torch.float32)
            hip_lib.check(lib.dd_gemm128(hip_lib.ptr(dy), rows, 0, 128, rows, hip_lib.ptr(Wt), None, None, hip_lib.ptr(df), rows, 0,
                                         kf, kf, 0, st), "dd_gemm128")
        if ctx.needs_input_grad[1]:
            dWt = torch.empty(kf, 128, device=dy.device, dtype=torch.float32)
            scratch = torch.empty(int(lib.dd_gemm128_tn_scratch_floats(rows, kf)), device=dy.device, dtype=torch.float32)
            hip_lib.check(lib.dd_gemm128_tn(hip_lib.ptr(fc), kf, kf, hip_lib.ptr(dy), 128, rows, hip_lib.ptr(scratch), hip_lib.ptr(dWt),
                                            128, 0, st), "dd_gemm128_tn")
            dW = dWt.t()
        return df, dW


class _LnRelu(torch.autograd.Function):
    """relu(LayerNorm(x) * gamma + beta) over 128 channels (the middle of every MLP of the network, common.py:85-105) as one HIP
    kernel each way (dd_train.hip): the backward forms dx, dgamma and dbeta in one pass over x and dy and recomputes the ReLU mask,
    where autograd runs threshold_backward, native_layer_norm_backward (two passes) and a reduce."""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        lib = hip_lib.load()
        xc, g, b = x.contiguous(), gamma.contiguous(), beta.contiguous()
        rows = xc.size(0)
        y = torch.empty_like(xc)
        stats = torch.empty(rows, 2, device=x.device, dtype=torch.float32)
        hip_lib.check(lib.dd_ln_relu_forward(hip_lib.ptr(xc), hip_lib.ptr(g), hip_lib.ptr(b), hip_lib.ptr(y), hip_lib.ptr(stats), rows,
                                             hip_lib.stream_ptr(x.device)), "dd_ln_relu_forward")
        ctx.save_for_backward(xc, stats, g, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, stats, g, b = ctx.saved_tensors
        lib = hip_lib.load()
        dy = dy.contiguous()
        rows = xc.size(0)
        dx = torch.empty_like(xc)
        dg, db = torch.empty(128, device=dy.device, dtype=torch.float32), torch.empty(128, device=dy.device, dtype=torch.float32)
        scratch = torch.empty(int(lib.dd_ln_relu_scratch_floats(rows)), device=dy.device, dtype=torch.float32)
        hip_lib.check(lib.dd_ln_relu_backward(hip_lib.ptr(xc), hip_lib.ptr(stats), hip_lib.ptr(g), hip_lib.ptr(b), hip_lib.ptr(dy),
                                              hip_lib.ptr(dx), hip_lib.ptr(scratch), hip_lib.ptr(dg), hip_lib.ptr(db), rows,
                                              hip_lib.stream_ptr(dy.device)), "dd_ln_relu_backward")
        return dx, dg, db


def ln_relu(x, gamma, beta):
    """F.relu(F.layer_norm(x, (128,), gamma, beta, 1e-5)) for [rows, 128] fp32 device tensors on the fused kernels (others: ATen)."""
    if x.dim() != 2 or x.size(1) != H or x.dtype != torch.float32 or not x.is_cuda or x.size(0) == 0:
        return F.relu(F.layer_norm(x, (H,), gamma, beta, 1e-5))
    return _LnRelu.apply(x, gamma, beta)


def linear_feat(f, W):
    """F.linear(f, W) without bias for narrow feature blocks feeding the 128 hidden channels."""
    if f.dim() != 2 or W.size(0) != 128 or f.size(1) > 128 or f.dtype != torch.float32 or f.size(0) == 0:
        return F.linear(f, W)
    return _LinearFeat.apply(f, W)


def linear128(x, W, b=None):
    """F.linear for [rows, 128] inputs through the library's GEMMs (other widths: ATen)."""
    if x.dim() != 2 or x.size(1) != 128 or W.size(1) != 128 or x.dtype != torch.float32 or W.size(0) > 128 or x.size(0) == 0:
        return F.linear(x, W, b)
    return _Linear128.apply(x, W, b)


_CONST = {}


def _const(values, device, dtype=torch.float32):
    """Small constant vectors live on the device once per process (a torch.tensor(list, device=...) per call is a blocking
    host -> device copy: 20 of them per training step before)."""
    key = (tuple(values), str(device), dtype)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.tensor(list(values), dtype=dtype, device=device)
    return t


def _gauss(d):
    off = _const(GAUSS_OFFSETS, d.device, d.dtype)
    return torch.exp(-0.5 * (d.reshape(-1, 1) - off) ** 2)                # GaussianSmearing, coeff -0.5 (common.py:23)


def _angle_code(theta):
    f = _const([1.0, 2.0, 3.0, 1.0, 0.5, 1.0 / 3.0], theta.device, theta.dtype)
    a = theta.unsqueeze(-1)
    return torch.cat([a, torch.sin(a * f), torch.cos(a * f)], -1)           # AngularEncoding (common.py:46-54), 13 wide


class _P:
    """Parameter access by reference key (``refine_net.base_block.0.lin_node.weight`` ...)."""

    def __init__(self, model):
        self.p = dict(model.named_parameters())

    def lin(self, name, x):
        return linear128(x, self.p[name + ".weight"], self.p[name + ".bias"])

    def w(self, name):
        return self.p[name + ".weight"]

    def b(self, name):
        return self.p[name + ".bias"]

    def mlp_tail(self, name, pre):
        """LayerNorm -> ReLU -> second Linear of MLP(num_layer=2, norm=True) (common.py:85-105) on a pre-activation."""
        return self.lin(name + ".net.3", ln_relu(pre, self.p[name + ".net.1.weight"], self.p[name + ".net.1.bias"]))

    def mlp(self, name, x):
        return self.mlp_tail(name, self.lin(name + ".net.0", x))


def _attention(q_e, k, v, seg, n_seg=None, member_real=None):
    """alpha = scatter_softmax((q k / sqrt(d)).sum(-1)); out = scatter_sum(alpha v)  (uni_transformer_edge.py:63-68).
    `seg`: the SegmentPlan of the destination index.  `member_real` (padded batches): bool per member -- padding members get a
    score of -1e30 (their exponential is exactly 0: the softmax is the softmax over the real members, same sums) and a weight of
    exactly 0 afterwards (a segment without any real member then contributes nothing)."""
    hd = k.shape[1] // NH
    score = (q_e.view(-1, NH, hd) * k.view(-1, NH, hd)).sum(-1) / math.sqrt(hd)
    if member_real is not None:
        score = torch.where(member_real.unsqueeze(-1), score, torch.full_like(score, -1e30))
    alpha = scatter_softmax(score, seg)
    if member_real is not None:
        alpha = alpha * member_real.unsqueeze(-1).to(alpha.dtype)
    return alpha


def _edge_mlp_pre(P, name, W_off, dst_tab, src_tab, dst, src, extra):
    """first Linear of an edge MLP, factorised: W[:, a:b] applied per node once, gathered per edge."""
    by_src = gather(src_tab, src) if isinstance(src, FN.SegmentPlan) else src_tab.index_select(0, src)
    # (the bias joins the per-NODE table before the gather: its gradient is then a sum over N rows behind the gather's segment sum,
    #  not a reduction over all E edge rows)
    return gather(dst_tab + P.b(name + ".net.0"), dst) + by_src + extra


def check_batch_layout(batch_protein, batch_ligand, ligand_fc_bond_index, batch_ligand_bond=None):
    """The layout both network paths assume, checked once per call (never silently wrong triplets / gradients): sorted
    PyG batch vectors and, per sample, the dst-major fully connected bond list of FeaturizeLigandBond('fc')
    (utils/transforms.py:331-337) offset by the sample's first ligand row (utils/data.py:443-444).
    Returns (n_p, n_l) per sample as python lists."""
    for name, t in (("batch_protein", batch_protein), ("batch_ligand", batch_ligand)):
        if t.numel() == 0:
            raise ValueError("empty batch")
        if t.numel() > 1 and bool((t[1:] < t[:-1]).any().item()):
            raise NotImplementedError(f"{name} must be sorted (PyG Batch order)")
    B = int(batch_protein.max().item()) + 1
    n_p = torch.bincount(batch_protein, minlength=B).tolist()
    n_l = torch.bincount(batch_ligand, minlength=B).tolist()
    if len(n_l) != B or min(n_p) < 1 or min(n_l) < 2:
        raise NotImplementedError("every sample needs protein atoms and at least 2 ligand atoms")
    dev = ligand_fc_bond_index.device
    exp, off = [], 0
    for n in n_l:
        dst = torch.arange(n, device=dev).repeat_interleave(n - 1)
        sp = torch.arange(n - 1, device=dev).repeat(n)
        exp.append(torch.stack([sp + (sp >= dst).long(), dst], 0) + off)
        off += n
    exp = torch.cat(exp, 1)
    if ligand_fc_bond_index.shape != exp.shape or not torch.equal(ligand_fc_bond_index, exp):
        raise NotImplementedError("ligand_fc_bond_index must be the dst-major fully connected graph ('fc' mode)")
    if batch_ligand_bond is not None:
        want = torch.repeat_interleave(torch.arange(B, device=dev), torch.tensor([n * (n - 1) for n in n_l], device=dev))
        if batch_ligand_bond.shape != want.shape or not torch.equal(batch_ligand_bond, want):
            raise NotImplementedError("batch_ligand_bond does not match the fully connected bond lists")
    return n_p, n_l


def network_grouped(net, model, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, ligand_v_aux, batch_ligand,
                    ligand_fc_bond_index, ligand_bond_type, sizes=None) -> Dict[str, torch.Tensor]:
    """``net`` (``network`` or the fused forward) on a batch whose samples differ in size -- what the reference's training
    batches are (batch_size 4 of different complexes, configs/training.yml:62): samples of equal (protein, ligand) size
    form one dense sub-batch each, the outputs are put back in the caller's row order (differentiable: cat + index_select).
    Exact because every graph op of the network is segmented by sample."""
    n_p, n_l = sizes if sizes is not None else check_batch_layout(batch_protein, batch_ligand, ligand_fc_bond_index)
    B, dev = len(n_p), protein_pos.device
    if len(set(zip(n_p, n_l))) == 1:
        if net is network and sizes is not None:            # (layout established by check_batch_layout: no re-check, no sync)
            return net(model, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, ligand_v_aux, batch_ligand,
                       ligand_fc_bond_index, ligand_bond_type, checked_B=B)
        return net(model, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, ligand_v_aux, batch_ligand,
                   ligand_fc_bond_index, ligand_bond_type)
    if net is network and os.environ.get("DD_TRAIN_PAD", "1") != "0":
        out = network_padded(model, protein_pos, protein_v, ligand_pos, ligand_v, ligand_v_aux, ligand_bond_type, n_p, n_l)
        if out is not None:
            return out
    o_p = [0] + list(torch.tensor(n_p).cumsum(0).tolist())
    o_l = [0] + list(torch.tensor(n_l).cumsum(0).tolist())
    n_b = [n * (n - 1) for n in n_l]
    o_b = [0] + list(torch.tensor(n_b).cumsum(0).tolist())
    groups: Dict = {}
    for b in range(B):
        groups.setdefault((n_p[b], n_l[b]), []).append(b)
    outs, rows_l, rows_b = [], [], []
    ar = lambda a, n: torch.arange(a, a + n, device=dev)
    for (np_, nl_), members in groups.items():
        rp = torch.cat([ar(o_p[b], np_) for b in members])
        rl = torch.cat([ar(o_l[b], nl_) for b in members])
        rb = torch.cat([ar(o_b[b], nl_ * (nl_ - 1)) for b in members])
        g = len(members)
        _, _, fc = model._expected_layout(g, np_, nl_, dev)
        extra = {"checked_B": g} if net is network else {}      # (batch vectors and bond list built right here: no re-check, no sync)
        outs.append(net(model, protein_pos[rp], protein_v[rp], torch.arange(g, device=dev).repeat_interleave(np_),
                        ligand_pos[rl], ligand_v[rl], ligand_v_aux[rl], torch.arange(g, device=dev).repeat_interleave(nl_),
                        fc, ligand_bond_type[rb], **extra))
        rows_l.append(rl)
        rows_b.append(rb)
    inv_l = torch.empty(o_l[-1], dtype=torch.long, device=dev)
    inv_l[torch.cat(rows_l)] = torch.arange(o_l[-1], device=dev)
    inv_b = torch.empty(o_b[-1], dtype=torch.long, device=dev)
    inv_b[torch.cat(rows_b)] = torch.arange(o_b[-1], device=dev)
    cat = lambda k, inv: torch.cat([o[k] for o in outs], 0).index_select(0, inv)
    return {"pred_ligand_pos": cat("pred_ligand_pos", inv_l), "pred_ligand_v": cat("pred_ligand_v", inv_l),
            "pred_bond": cat("pred_bond", inv_b)}


def _pad_rows(n_p, n_l, NPm=None, NLm=None):
    """Row maps of a heterogeneous batch into its padded dense layout (numpy, host): real protein / ligand rows and real bonds
    (the caller's dst-major lists over n_l[b] atoms) -> rows of the [B*NPm] / [B*NLm] / [B*NLm(NLm-1)] padded arrays."""
    import numpy as np
    B = len(n_p)
    NPm, NLm = NPm or max(n_p), NLm or max(n_l)
    Ebm = NLm * (NLm - 1)
    rows_p = np.concatenate([b * NPm + np.arange(n) for b, n in enumerate(n_p)])
    rows_l = np.concatenate([b * NLm + np.arange(n) for b, n in enumerate(n_l)])
    rb = []
    for b, n in enumerate(n_l):
        dst = np.repeat(np.arange(n), n - 1)
        sp = np.tile(np.arange(n - 1), n)
        src = sp + (sp >= dst)
        rb.append(b * Ebm + dst * (NLm - 1) + (src - (src > dst)))
    return rows_p, rows_l, np.concatenate(rb)


def _far_positions(B, NPm, NLm, dev):
    """Positions of padding atoms: far from everything and from each other, never collinear (norms / cross products keep finite
    gradients): protein padding around x = +1000, ligand padding around x = -1000 (and beyond, 3 A apart)."""
    i_p = torch.arange(B * NPm, device=dev, dtype=torch.float32)
    far_p = torch.stack([1.0e3 + 3.0 * i_p, 5.0 * torch.sin(1.3 * i_p), 4.0 * torch.cos(2.1 * i_p)], -1)
    i_l = torch.arange(B * NLm, device=dev, dtype=torch.float32)
    far_l = torch.stack([-1.0e3 - 3.0 * i_l, 5.0 * torch.cos(1.7 * i_l), 4.0 * torch.sin(0.9 * i_l)], -1)
    return far_p, far_l


def network_padded(model, protein_pos, protein_v, ligand_pos, ligand_v, ligand_v_aux, ligand_bond_type, n_p, n_l):
    """A batch whose samples differ in size -- what the reference's loader yields (batch_size 4 different complexes,
    configs/training.yml:62) -- as ONE dense pass: every sample is padded to the batch's largest protein / ligand, padding atoms
    sit far away at distinct non-collinear positions (finite features everywhere), are excluded from the kNN graph and masked as
    members of the bond-graph and triplet attentions; the outputs' real rows are returned in the caller's order.  One network
    pass instead of one per distinct size (4 x fewer launches of a host-bound step) for max-size padding waste in the NL^3
    triplet count.  Returns None if a sample has fewer than K + 1 real atoms (the per-sample kNN degree would differ)."""
    dev = protein_pos.device
    B, NPm, NLm = len(n_p), max(n_p), max(n_l)
    N = NPm + NLm
    K = min(int(model.config.knn), N - 1)
    if min(a + b for a, b in zip(n_p, n_l)) < K + 1:
        return None
    rows_p, rows_l, rows_b = (torch.from_numpy(r).to(dev) for r in _pad_rows(n_p, n_l))
    far_p, far_l = _far_positions(B, NPm, NLm, dev)
    pp = far_p.index_copy(0, rows_p, protein_pos.to(torch.float32))
    lp = far_l.index_copy(0, rows_l, ligand_pos.to(torch.float32))
    pv = torch.zeros(B * NPm, protein_v.shape[1], device=dev, dtype=protein_v.dtype).index_copy(0, rows_p, protein_v)
    lv = torch.zeros(B * NLm, device=dev, dtype=ligand_v.dtype).index_copy(0, rows_l, ligand_v)
    la = torch.zeros(B * NLm, ligand_v_aux.shape[1], device=dev, dtype=ligand_v_aux.dtype).index_copy(0, rows_l, ligand_v_aux)
    bt = torch.zeros(B * NLm * (NLm - 1), device=dev, dtype=ligand_bond_type.dtype).index_copy(0, rows_b, ligand_bond_type)
    b_p, b_l, fc = model._expected_layout(B, NPm, NLm, dev)
    pad = dict(np_real=torch.tensor(n_p, dtype=torch.int32, device=dev), nl_real=torch.tensor(n_l, dtype=torch.int32, device=dev))
    out = network(model, pp, pv, b_p, lp, lv, la, b_l, fc, bt, checked_B=B, pad=pad)
    return {"pred_ligand_pos": out["pred_ligand_pos"].index_select(0, rows_l), "pred_ligand_v": out["pred_ligand_v"].index_select(0, rows_l),
            "pred_bond": out["pred_bond"].index_select(0, rows_b)}


# While a step is being CAPTURED every cached object the capture reads (index structures, segment plans) is appended here:
# the graph bakes in their device addresses, so the graph entry keeps them alive whatever the caches evict later
_PIN = None


def _pin(obj):
    if _PIN is not None:
        _PIN.append(obj)
    return obj


_STRUCT: Dict = {}
_STRUCT_MAX_BYTES = int(os.environ.get("DD_TRAIN_STRUCT_CACHE_MB", "512")) << 20     # device memory the cached index structures may hold
_STRUCT_MAX_ENTRIES = 16


def _tensor_bytes(obj) -> int:
    if torch.is_tensor(obj):
        return obj.numel() * obj.element_size()
    if isinstance(obj, dict):
        return sum(_tensor_bytes(v) for v in obj.values())
    if isinstance(obj, (list, tuple)):
        return sum(_tensor_bytes(v) for v in obj)
    return sum(_tensor_bytes(v) for v in vars(obj).values()) if hasattr(obj, "__dict__") else 0


def _structure(B, NP, NL, K, dev):
    """Index structure of a dense batch of B samples with NP protein + NL ligand atoms each (static across steps).
    Cached per shape, least recently used first out, bounded by bytes (the triplet index vectors grow as B * NL^3: ~0.65 GB
    at NL = 128, B = 8 -- a structure beyond the budget is built per call and not kept)."""
    key = (B, NP, NL, K, str(dev))
    S = _STRUCT.pop(key, None)
    if S is not None:
        _STRUCT[key] = S                                   # re-inserted: most recently used last
        return _pin(S)
    N = NP + NL
    is_lig = torch.cat([torch.zeros(NP, dtype=torch.bool), torch.ones(NL, dtype=torch.bool)]).repeat(B).to(dev)
    lig_rows = is_lig.nonzero().squeeze(1)
    # dst-major fully connected bond list of FeaturizeLigandBond('fc') (utils/transforms.py:331-337), per-sample offsets
    d = torch.arange(NL, device=dev).repeat_interleave(NL - 1)
    sp = torch.arange(NL - 1, device=dev).repeat(NL)
    fc1 = torch.stack([sp + (sp >= d).long(), d], 0)
    fc = torch.cat([fc1 + b * NL for b in range(B)], 1)
    bond_src, bond_dst = lig_rows[fc[0]], lig_rows[fc[1]]
    bsrc_loc, b_of_bond = fc[0] % NL, fc[0] // NL           # (padded batches: is the source atom of a bond real?)
    dst = torch.arange(B * N, device=dev).repeat_interleave(K)
    S = dict(is_lig=is_lig, lig_rows=lig_rows, fc=fc, bond_src=bond_src, bond_dst=bond_dst, dst=dst,
             dst_is_prot=(~is_lig.index_select(0, dst)).long(), base=(torch.arange(B, device=dev) * N).view(B, 1, 1),
             p_dst=seg_plan(dst, B * N), p_bdst=seg_plan(bond_dst, B * N), p_bsrc=seg_plan(bond_src, B * N), trip=None, p_ji=None,
             bsrc_loc=bsrc_loc, b_of_bond=b_of_bond)
    # ---- triplets k -> j -> i over the fully connected ligand bond graph (BondUpdateLayer.triplets, :103-123)
    NLm1, Ebs = NL - 1, NL * (NL - 1)
    if NL > 2:
        e_ji = torch.arange(Ebs, device=dev).repeat_interleave(NL - 2)                  # segment = bond (j -> i), dst-major
        i_loc, jp = e_ji // NLm1, e_ji % NLm1
        j_loc = jp + (jp >= i_loc).long()
        mc = torch.arange(NL - 2, device=dev).repeat(Ebs)
        lo, hi = torch.minimum(i_loc, j_loc), torch.maximum(i_loc, j_loc)
        k_loc = mc + (mc >= lo).long()
        k_loc = k_loc + (k_loc >= hi).long()
        e_kj = j_loc * NLm1 + (k_loc - (k_loc > j_loc).long())                          # bond (k -> j)
        boff = (torch.arange(B, device=dev) * Ebs).repeat_interleave(Ebs * (NL - 2))
        aoff = (torch.arange(B, device=dev) * N + NP).repeat_interleave(Ebs * (NL - 2))
        rep = lambda t: t.repeat(B)
        S["trip"] = dict(ji=rep(e_ji) + boff, kj=rep(e_kj) + boff, i=rep(i_loc) + aoff, j=rep(j_loc) + aoff, k=rep(k_loc) + aoff)
        S["tk_loc"], S["b_of_trip"] = rep(k_loc), boff // Ebs
        S["p_ji"] = seg_plan(S["trip"]["ji"], B * Ebs)
        # rows of bond (k -> j) gathered per triplet: the backward through a sorted segment sum (NL - 2 rows per bond) instead of
        # ATen's atomic index_add_.  (The per-ATOM gathers of the triplets -- (NL-1)(NL-2) rows per atom -- stay on index_add_:
        # a wave per segment is the wrong shape for 120 segments of 812 rows, measured 3 x slower than the atomics.)
        S["p_tkj"] = seg_plan(S["trip"]["kj"], B * Ebs)
    S["_bytes"] = _tensor_bytes(S)
    if S["_bytes"] <= _STRUCT_MAX_BYTES:
        _STRUCT[key] = S
        while len(_STRUCT) > _STRUCT_MAX_ENTRIES or sum(v["_bytes"] for v in _STRUCT.values()) > _STRUCT_MAX_BYTES:
            _STRUCT.pop(next(iter(_STRUCT)))               # least recently used first
    return _pin(S)


def _knn_src(x, B, N, K, base, NP=None, pad=None):
    """Sources of the kNN edges (dd_knn; edges grouped by centre in ascending order, neighbours by ascending distance).
    `pad` (padded batches): real atom counts per sample -- padding atoms are neither centres nor candidates (dd_knn_masked)."""
    xc = x.to(torch.float32).contiguous()
    if pad is not None:
        nbr = torch.empty(B, N, K, dtype=torch.int32, device=x.device)
        hip_lib.check(hip_lib.load().dd_knn_masked(hip_lib.ptr(xc), B, NP, N - NP, K, hip_lib.ptr(pad["np_real"]), hip_lib.ptr(pad["nl_real"]),
                                                   hip_lib.ptr(nbr), hip_lib.stream_ptr(x.device)), "dd_knn_masked")
        return (nbr.long() + base).reshape(-1)
    ext = FN.torch_ext()
    if ext is not None:
        nbr = ext.knn(xc.view(B, N, 3), K)
    else:
        nbr = torch.empty(B, N, K, dtype=torch.int32, device=x.device)
        hip_lib.check(hip_lib.load().dd_knn(hip_lib.ptr(xc), B, N, K, hip_lib.ptr(nbr), hip_lib.stream_ptr(x.device)), "dd_knn")
    return (nbr.long() + base).reshape(-1)


def network(model, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, ligand_v_aux, batch_ligand,
            ligand_fc_bond_index, ligand_bond_type, checked_B: Optional[int] = None, pad: Optional[Dict] = None) -> Dict[str, torch.Tensor]:
    """DecompScorePosNet3D.forward for the shipped configuration, differentiable w.r.t. the model's parameters.
    Dense batches (equal sizes per sample, sorted batch vectors, dst-major fc bond index -- `check_batch_layout`, run by
    `diffusion_loss`; samples of different sizes go through `network_grouped`).  `checked_B`: the caller has established the
    layout (number of samples given): no device -> host round trip in here -- what a captured training step needs.
    `pad` = dict(np_real, nl_real: int32 [B] on the device): a PADDED heterogeneous batch (`network_padded`) -- every sample's
    real atoms are the first rows of its protein / ligand block; padding atoms are excluded from the kNN graph and masked as
    members of the bond-graph and triplet attentions, so the real rows of the outputs equal the unpadded network's."""
    cfg = model.config
    P = _P(model)
    dev = protein_pos.device
    hip_lib.require_gpu(protein_pos, "protein_pos")
    B = int(checked_B) if checked_B is not None else int(batch_protein.max().item()) + 1
    NP, NL = batch_protein.numel() // B, batch_ligand.numel() // B
    N = NP + NL
    K = min(int(cfg.knn), N - 1)
    # ---- embeddings + context order [protein..., ligand...] per sample (compose_context, common.py:153-194)
    lig_feat = torch.cat([F.one_hot(ligand_v, model.num_classes).float(), ligand_v_aux.float()], -1)
    h_p = torch.cat([P.lin("protein_atom_emb", protein_v.float()), torch.zeros(B * NP, 1, device=dev)], -1)
    h_l = torch.cat([P.lin("ligand_atom_emb", lig_feat), torch.ones(B * NL, 1, device=dev)], -1)
    h = torch.cat([h_p.view(B, NP, H), h_l.view(B, NL, H)], 1).reshape(B * N, H)
    x = torch.cat([protein_pos.view(B, NP, 3), ligand_pos.view(B, NL, 3)], 1).reshape(B * N, 3)
    # ---- static index structure of the dense batch shape (context order, bond endpoints, triplets, segment plans): built once
    #      per (B, NP, NL, K, device) and reused by every step of that shape -- ~150 small index kernels and ~20 device -> host
    #      round trips per step otherwise
    S = _structure(B, NP, NL, K, dev)
    if checked_B is None and (ligand_fc_bond_index.shape != S["fc"].shape or not torch.equal(ligand_fc_bond_index, S["fc"])):
        raise NotImplementedError("ligand_fc_bond_index must be the dst-major fully connected graph ('fc' mode)")
    is_lig, lig_rows, bond_src, bond_dst, dst = S["is_lig"], S["lig_rows"], S["bond_src"], S["bond_dst"], S["dst"]
    p_dst, p_bdst, p_ji, trip = S["p_dst"], S["p_bdst"], S["p_ji"], S["trip"]
    h_bond = P.lin("ligand_bond_emb", F.one_hot(ligand_bond_type, model.num_bond_classes).float())
    Eb_tot = h_bond.shape[0]
    # ---- graph of the step (uni_transformer_edge.py:404-427): kNN among all atoms of a sample, fixed for all layers
    src = _knn_src(x.detach(), B, N, K, S["base"], NP=NP, pad=pad)
    real_b = real_t = None
    if pad is not None:                                    # members of the bond-graph / triplet segments that are real atoms
        nl_r = pad["nl_real"].long()
        real_b = S["bsrc_loc"] < nl_r.index_select(0, S["b_of_bond"])
        if trip is not None:
            real_t = S["tk_loc"] < nl_r.index_select(0, S["b_of_trip"])
    p_src = seg_plan(src, B * N, check=False)              # rows gathered by source: backward = sorted segment sum, no atomics
    etype = 2 * (~is_lig.index_select(0, src)).long() + S["dst_is_prot"]  # 0 ll, 1 l->p(dst p), 2 p->l, 3 pp
    etype_1h = F.one_hot(etype, 4).float()
    d0 = (x.index_select(0, dst) - x.index_select(0, src)).norm(dim=-1)
    e_w = torch.sigmoid(P.mlp("refine_net.edge_pred_layer", _gauss(d0)))
    NLm1, Ebs = NL - 1, NL * (NL - 1)
    mask_l = is_lig.float().unsqueeze(-1)
    for l in range(int(cfg.num_layers)):
        p = f"refine_net.base_block.{l}"
        rel = gather(x, p_dst) - x.index_select(0, src)
        dist = rel.norm(dim=-1)
        g = _gauss(dist)
        ef_type = torch.cat([(etype_1h.unsqueeze(-1) * g.unsqueeze(1)).reshape(-1, 80), etype_1h], -1)      # [E, 84]

        def node_layer_edge(name, hh, v_width):
            """kNN sub-layers: first Linear columns [0:84] edge feature, [84:212] h[dst], [212:340] h[src]."""
            outs = []
            for f_ in ("k", "v"):
                nm = f"{p}.{name}.{'h' if name.startswith('node') else 'x'}{f_}_func"
                W = P.w(nm + ".net.0")
                pre = _edge_mlp_pre(P, nm, None, linear128(hh, W[:, 84:212]), linear128(hh, W[:, 212:340]), p_dst, p_src,
                                    linear_feat(ef_type, W[:, 0:84]))
                outs.append(P.mlp_tail(nm, pre))
            return outs

        def node_layer_bond(name, hh, hb):
            """bond sub-layers: columns [0:128] h_bond[e], [128:256] h[dst], [256:384] h[src]."""
            outs = []
            for f_ in ("k", "v"):
                nm = f"{p}.{name}.{'h' if name.startswith('node') else 'x'}{f_}_func"
                W = P.w(nm + ".net.0")
                pre = _edge_mlp_pre(P, nm, None, linear128(hh, W[:, 128:256]), linear128(hh, W[:, 256:384]), p_bdst, S["p_bsrc"],
                                    linear128(hb, W[:, 0:128]))
                outs.append(P.mlp_tail(nm, pre))
            return outs

        # node_layer_with_edge (NodeUpdateLayer, :42-74)
        k_e, v_e = node_layer_edge("node_layer_with_edge", h, H)
        q_e = P.mlp(f"{p}.node_layer_with_edge.hq_func", h)
        alpha = _attention(gather(q_e, p_dst), k_e, v_e, p_dst)
        a_edge = scatter_sum((alpha.unsqueeze(-1) * (v_e * e_w).view(-1, NH, H // NH)).reshape(-1, H), p_dst)
        # node_layer_with_bond
        k_b, v_b = node_layer_bond("node_layer_with_bond", h, h_bond)
        q_b = P.mlp(f"{p}.node_layer_with_bond.hq_func", h)
        alpha = _attention(gather(q_b, p_bdst), k_b, v_b, p_bdst, member_real=real_b)
        a_bond = scatter_sum((alpha.unsqueeze(-1) * v_b.view(-1, NH, H // NH)).reshape(-1, H), p_bdst)
        # bond_layer (BondUpdateLayer, :125-167): kv = [h_bond[kj](128), G(d_kj)(20), G(d_ji)(20), angle(13), h[k], h[j]]
        if trip is not None:
            nm_b = f"{p}.bond_layer"
            d_bond = (gather(x, p_bdst) - x.index_select(0, bond_src)).norm(dim=-1)
            gb = _gauss(d_bond)
            x_i = x.index_select(0, trip["i"])
            v_ji, v_ki = x.index_select(0, trip["j"]) - x_i, x.index_select(0, trip["k"]) - x_i
            theta = torch.atan2(torch.cross(v_ji, v_ki, dim=-1).norm(dim=-1), (v_ji * v_ki).sum(-1))
            code = _angle_code(theta)
            kv = []
            for f_ in ("hk_func", "hv_func"):
                W = P.w(f"{nm_b}.{f_}.net.0")
                # every triplet (k -> j -> i) term is a per-BOND row: h[k] rides with bond (k -> j) (its source atom), h[j] and the bias
                # with bond (j -> i) -- two gathers of [Eb, 128] tables per MLP instead of four [E3, 128] ones, and no per-atom
                # index_add_ of 812 rows per atom in the backward
                per_kj = linear128(h_bond, W[:, 0:128]) + linear_feat(gb, W[:, 128:148]) + gather(linear128(h, W[:, 181:309]), S["p_bsrc"])
                per_ji = linear_feat(gb, W[:, 148:168]) + P.b(f"{nm_b}.{f_}.net.0") + gather(linear128(h, W[:, 309:437]), S["p_bsrc"])
                pre = gather(per_kj, S["p_tkj"]) + gather(per_ji, p_ji) + linear_feat(code, W[:, 168:181])
                kv.append(P.mlp_tail(f"{nm_b}.{f_}", pre))
            # hq depends on the (j -> i) bond only: evaluated per bond, gathered per triplet (exact)
            q_bond = P.mlp(f"{nm_b}.hq_func", torch.cat([h_bond, gather(h, p_bdst)], -1))
            alpha = _attention(gather(q_bond, p_ji), kv[0], kv[1], p_ji, member_real=real_t)
            d_hb = scatter_sum((alpha.unsqueeze(-1) * kv[1].view(-1, NH, H // NH)).reshape(-1, H), p_ji)
        else:
            d_hb = torch.zeros_like(h_bond)
        new_h_bond = h_bond + d_hb
        new_h = h + P.lin(f"{p}.lin_node", a_edge + a_bond)
        # pos_layer_with_edge / pos_layer_with_bond (PosUpdateLayer, :188-210), with the NEW h / h_bond
        k_pe, v_pe = node_layer_edge("pos_layer_with_edge", new_h, NH)
        q_pe = P.mlp(f"{p}.pos_layer_with_edge.xq_func", new_h)
        alpha = _attention(gather(q_pe, p_dst), k_pe, None, p_dst)
        dx_e = scatter_sum(((alpha * (v_pe * e_w)).unsqueeze(-1) * rel.unsqueeze(1)).reshape(-1, NH * 3), p_dst).view(-1, NH, 3).mean(1)
        k_pb, v_pb = node_layer_bond("pos_layer_with_bond", new_h, new_h_bond)
        q_pb = P.mlp(f"{p}.pos_layer_with_bond.xq_func", new_h)
        alpha = _attention(gather(q_pb, p_bdst), k_pb, None, p_bdst, member_real=real_b)
        rel_b = gather(x, p_bdst) - x.index_select(0, bond_src)
        dx_b = scatter_sum(((alpha * v_pb).unsqueeze(-1) * rel_b.unsqueeze(1)).reshape(-1, NH * 3), p_bdst).view(-1, NH, 3).mean(1)
        x = x + (dx_e + dx_b) * mask_l
        h, h_bond = new_h, new_h_bond
    softplus = lambda t: F.softplus(t) - math.log(2.0)                              # ShiftedSoftplus (common.py:66-72)
    final_h = h.index_select(0, lig_rows)
    out = {"pred_ligand_pos": x.index_select(0, lig_rows),
           "pred_ligand_v": P.lin("v_inference.2", softplus(P.lin("v_inference.0", final_h))),
           "pred_bond": P.lin("bond_inference.2", softplus(P.lin("bond_inference.0", h_bond)))}
    return out


# --------------------------------------------------------------------------------------------------------------------
# the objective
# --------------------------------------------------------------------------------------------------------------------
def _log_onehot(idx, k):
    return torch.log(F.one_hot(idx, k).float().clamp(min=1e-30))


def _log_add_exp(a, b):
    m = torch.max(a, b)
    return m + torch.log(torch.exp(a - m) + torch.exp(b - m))


class _Trans:
    """DiscreteTransition (models/transitions.py:97-161) over the model's schedule tables."""

    def __init__(self, mod):
        self.t = mod

    def pred(self, log_v0, t, batch):                      # q(v_t | v_0)
        return _log_add_exp(log_v0 + self.t.log_alphas_cumprod_v[t][batch].unsqueeze(-1),
                            self.t.log_one_minus_alphas_cumprod_v[t][batch].unsqueeze(-1) + self.t.prior_probs)

    def pred_one(self, log_vt_1, t, batch):                 # q(v_t | v_{t-1})
        return _log_add_exp(log_vt_1 + self.t.log_alphas_v[t][batch].unsqueeze(-1),
                            self.t.log_one_minus_alphas_v[t][batch].unsqueeze(-1) + self.t.prior_probs)

    def posterior(self, log_v0, log_vt, t, batch):          # q(v_{t-1} | v_t, v_0)
        tm1 = torch.where(t - 1 < 0, torch.zeros_like(t), t - 1)
        un = self.pred(log_v0, tm1, batch) + self.pred_one(log_vt, t, batch)
        return un - torch.logsumexp(un, dim=-1, keepdim=True)


def _v_loss(log_model, log_v0, log_true, t, batch, n, plan=None):
    kl = (log_true.exp() * (log_true - log_model)).sum(1)                  # categorical_kl (decompdiff.py:35-37)
    nll = -(log_v0.exp() * log_model).sum(1)                                # -log_categorical (decompdiff.py:40-41)
    mask = (t == 0).float()[batch]
    per = mask * nll + (1.0 - mask) * kl
    return FN_mean(per, plan if plan is not None else batch, n)


_PLANS: Dict = {}


def _static_plan(batch, B, n_l, bonds):
    """SegmentPlan of a batch vector whose content follows from the per-sample sizes (checked by check_batch_layout): built
    once per (sizes, device) -- a plan per call costs three device -> host round trips, which a captured step cannot have."""
    key = (tuple(n_l), bool(bonds), str(batch.device))
    plan = _PLANS.get(key)
    if plan is None or plan.E != batch.numel():
        if len(_PLANS) > 64:
            _PLANS.clear()
        plan = _PLANS[key] = seg_plan(batch.clone(), B)
    return _pin(plan)


def FN_mean(per_row, batch, n):
    """scatter_mean over samples, differentiable (sum through the HIP scatter, count is constant).  `batch`: index vector
    or its SegmentPlan."""
    plan = batch if isinstance(batch, FN.SegmentPlan) else seg_plan(batch, n)
    cnt = (plan.ptr[1:] - plan.ptr[:-1]).clamp(min=1).to(per_row.dtype)
    return scatter_sum(per_row.unsqueeze(-1), plan).squeeze(-1) / cnt


def sample_time(model, num_graphs, device, method=None):
    """sample_time (decompdiff.py:374-400), both methods; the draw is made on the CPU generator (torch.manual_seed), as the
    reference's CPU run makes it.

    'symmetric' (:391-397): t and T-1-t pairs, uniform weights.
    'importance' (:375-389): while any time step has been recorded 10 times or fewer (`Lt_count`), fall back to 'symmetric';
    afterwards draw t ~ sqrt(Lt_history + 1e-10) + 1e-4 (entry 0 overwritten with entry 1), normalised, by
    torch.multinomial, and return the drawn steps' probabilities.  NOTE the reference never writes `Lt_history` /
    `Lt_count` (two registered buffers, models/decompdiff.py:146-147; no update anywhere in the repository), so its
    'importance' method always takes the fall-back; the buffers are part of the state_dict here as there, and a host that
    maintains them itself gets the importance branch exactly as written upstream."""
    method = method or model.sample_time_method
    if method == "importance":
        if not bool((model.Lt_count > 10).all()):
            return sample_time(model, num_graphs, device, method="symmetric")
        lt_sqrt = torch.sqrt(model.Lt_history.detach().float().cpu() + 1e-10) + 0.0001
        lt_sqrt[0] = lt_sqrt[1]                                # overwrite the decoder term with L1 (:380)
        pt_all = lt_sqrt / lt_sqrt.sum()
        ts = torch.multinomial(pt_all, num_samples=num_graphs, replacement=True)
        return ts.to(device), pt_all.gather(0, ts).to(device)
    if method == "symmetric":
        ts = torch.randint(0, model.num_timesteps, size=(num_graphs // 2 + 1,))
        ts = torch.cat([ts, model.num_timesteps - ts - 1], 0)[:num_graphs]
        return ts.to(device), torch.ones(num_graphs, device=device) / model.num_timesteps
    raise ValueError(method)                                   # (:399-400)


def prepare_batch(model, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, ligand_v_aux, batch_ligand,
                  prior_centers, prior_stds, prior_num_atoms, batch_prior, ligand_decomp_batch, ligand_fc_bond_index,
                  ligand_fc_bond_type, batch_ligand_bond, time_step=None) -> Dict:
    """Host side of get_diffusion_loss (decompdiff.py:419-480): layout checks, the time steps and the three noise draws on the
    CPU generator in the reference's order (time steps, position noise, atom-type Gumbel uniforms, bond-type Gumbel uniforms),
    moved to the device -- everything a step needs that involves the host.  The device side is `objective`."""
    dev = protein_pos.device
    sizes = check_batch_layout(batch_protein, batch_ligand, ligand_fc_bond_index, batch_ligand_bond)
    B = len(sizes[0])
    if time_step is None:
        time_step, _ = sample_time(model, B, dev)
    time_step = time_step.to(dev)
    assert len(ligand_decomp_batch) == int(prior_num_atoms.sum().item())
    pos_noise = torch.zeros(ligand_pos.shape).normal_().to(dev)
    u_v = torch.rand(ligand_v.shape[0], model.num_classes).to(dev)
    u_b = torch.rand(ligand_fc_bond_type.shape[0], model.num_bond_classes).to(dev)
    return dict(sizes=sizes, B=B, time_step=time_step, pos_noise=pos_noise, u_v=u_v, u_b=u_b,
                protein_pos=protein_pos, protein_v=protein_v, batch_protein=batch_protein, ligand_pos=ligand_pos, ligand_v=ligand_v,
                ligand_v_aux=ligand_v_aux, batch_ligand=batch_ligand, prior_centers=prior_centers, prior_stds=prior_stds,
                ligand_decomp_batch=ligand_decomp_batch, ligand_fc_bond_index=ligand_fc_bond_index,
                ligand_fc_bond_type=ligand_fc_bond_type, batch_ligand_bond=batch_ligand_bond)


# tensors of a prepared batch that change from step to step (a captured step copies them into its static buffers)
PREP_TENSORS = ("time_step", "pos_noise", "u_v", "u_b", "protein_pos", "protein_v", "batch_protein", "ligand_pos", "ligand_v",
                "ligand_v_aux", "batch_ligand", "prior_centers", "prior_stds", "ligand_decomp_batch", "ligand_fc_bond_index",
                "ligand_fc_bond_type", "batch_ligand_bond")


def objective(model, prep: Dict, network_fn=None) -> Dict:
    """Device side of get_diffusion_loss (decompdiff.py:455-550) on a prepared batch: forward diffusion of the state, the
    score network, the three losses.  No device -> host round trip when the batch is dense (one size for all samples) and
    `network_fn` is the differentiable network: this is what `GraphedTrainStep` captures."""
    dev = prep["protein_pos"].device
    sizes, B, time_step = prep["sizes"], prep["B"], prep["time_step"]
    protein_pos, protein_v, batch_protein = prep["protein_pos"], prep["protein_v"], prep["batch_protein"]
    ligand_pos, ligand_v, ligand_v_aux, batch_ligand = prep["ligand_pos"], prep["ligand_v"], prep["ligand_v_aux"], prep["batch_ligand"]
    ligand_decomp_batch, ligand_fc_bond_index = prep["ligand_decomp_batch"], prep["ligand_fc_bond_index"]
    ligand_fc_bond_type, batch_ligand_bond = prep["ligand_fc_bond_type"], prep["batch_ligand_bond"]
    a = model.alphas_cumprod.index_select(0, time_step)
    centers = prep["prior_centers"][ligand_decomp_batch]
    stds = prep["prior_stds"][ligand_decomp_batch]
    a_pos = a[batch_ligand].unsqueeze(-1)
    pos_pert = a_pos.sqrt() * (ligand_pos - centers) + (1.0 - a_pos).sqrt() * prep["pos_noise"] * stds + centers
    tv, tb = _Trans(model.atom_type_trans), _Trans(model.bond_type_trans)
    log_v0 = _log_onehot(ligand_v, model.num_classes)
    log_qv = tv.pred(log_v0, time_step, batch_ligand)
    v_pert = (-torch.log(-torch.log(prep["u_v"] + 1e-30) + 1e-30) + log_qv).argmax(-1)
    log_vt = _log_onehot(v_pert, model.num_classes)
    log_b0 = _log_onehot(ligand_fc_bond_type, model.num_bond_classes)
    log_qb = tb.pred(log_b0, time_step, batch_ligand_bond)
    b_pert = (-torch.log(-torch.log(prep["u_b"] + 1e-30) + 1e-30) + log_qb).argmax(-1)
    log_bt = _log_onehot(b_pert, model.num_bond_classes)
    # center_pos (decompdiff.py:20-32)
    if model.center_pos_mode == "protein":
        if len(set(sizes[0])) == 1:
            offset = protein_pos.view(B, sizes[0][0], 3).mean(1)
        else:                                                               # scatter_mean(protein_pos, batch_protein)
            cnt = torch.tensor(sizes[0], dtype=protein_pos.dtype, device=dev).view(B, 1)
            offset = torch.zeros(B, 3, dtype=protein_pos.dtype, device=dev).index_add_(0, batch_protein, protein_pos) / cnt
    elif model.center_pos_mode == "none":
        offset = torch.zeros(B, 3, device=dev)
    else:
        raise NotImplementedError(model.center_pos_mode)
    p_pos = protein_pos - offset[batch_protein]
    x_t = pos_pert - offset[batch_ligand]
    x_0 = ligand_pos - offset[batch_ligand]
    net = network_fn or network
    preds = network_grouped(net, model, p_pos, protein_v, batch_protein, x_t, v_pert, ligand_v_aux, batch_ligand,
                            ligand_fc_bond_index, b_pert, sizes=sizes)
    pred_pos, pred_v = preds["pred_ligand_pos"], preds["pred_ligand_v"]
    log_v_recon = F.log_softmax(pred_v, dim=-1)
    plan_l, plan_b = _static_plan(batch_ligand, B, sizes[1], False), _static_plan(batch_ligand_bond, B, sizes[1], True)
    kl_v = _v_loss(tv.posterior(log_v_recon, log_vt, time_step, batch_ligand), log_v0,
                   tv.posterior(log_v0, log_vt, time_step, batch_ligand), time_step, batch_ligand, B, plan_l)
    log_b_recon = F.log_softmax(preds["pred_bond"], dim=-1)
    kl_b = _v_loss(tb.posterior(log_b_recon, log_bt, time_step, batch_ligand_bond), log_b0,
                   tb.posterior(log_b0, log_bt, time_step, batch_ligand_bond), time_step, batch_ligand_bond, B, plan_b)
    if model.loss_pos_type != "mse":
        raise ValueError(model.loss_pos_type)
    loss_pos = FN_mean((((pred_pos - x_0) ** 2) / (stds ** 2)).sum(-1), plan_l, B).mean()
    return {"losses": {"pos": loss_pos, "v": kl_v.mean(), "bond": kl_b.mean()},
            "x0": x_0, "pred_ligand_pos": pred_pos, "pred_ligand_v": pred_v, "pred_pos_noise": pred_pos - x_t,
            "ligand_v_recon": F.softmax(pred_v, dim=-1), "ligand_b_recon": F.softmax(preds["pred_bond"], dim=-1),
            "time_step": time_step}


def pad_prepared(model, prep: Dict, bucket=(32, 4)) -> Optional[Dict]:
    """A prepared heterogeneous batch in a FIXED-SHAPE padded layout (sizes rounded up to `bucket`): every per-atom / per-bond tensor
    of `prep` scattered into [B*NPm] / [B*NLm] / [B*NLm(NLm-1)] arrays, the per-atom prior centres / stds gathered, row weights
    (1 real, 0 padding) and the real counts.  Runs outside any capture (its index maps have batch-dependent lengths); the result
    feeds `objective_padded`, whose tensors all have shapes that depend on (B, NPm, NLm) only.  None if a sample has fewer than
    K + 1 real atoms."""
    n_p, n_l = prep["sizes"]
    B = len(n_p)
    up = lambda n, q: -(-n // q) * q
    NPm, NLm = up(max(n_p), int(bucket[0])), up(max(n_l), int(bucket[1]))
    dev = prep["protein_pos"].device
    K = min(int(model.config.knn), NPm + NLm - 1)
    if min(a + b for a, b in zip(n_p, n_l)) < K + 1:
        return None
    rows_p, rows_l, rows_b = (torch.from_numpy(r).to(dev) for r in _pad_rows(n_p, n_l, NPm, NLm))
    Ebm = NLm * (NLm - 1)
    far_p, far_l = _far_positions(B, NPm, NLm, dev)
    z = lambda n, like, fill=0: torch.full((n,) + tuple(like.shape[1:]), fill, device=dev, dtype=like.dtype)
    dec = prep["ligand_decomp_batch"]
    out = dict(B=B, NPm=NPm, NLm=NLm, time_step=prep["time_step"],
               protein_pos=far_p.index_copy(0, rows_p, prep["protein_pos"].float()),
               protein_v=z(B * NPm, prep["protein_v"]).index_copy(0, rows_p, prep["protein_v"]),
               ligand_pos=far_l.index_copy(0, rows_l, prep["ligand_pos"].float()),
               ligand_v=z(B * NLm, prep["ligand_v"]).index_copy(0, rows_l, prep["ligand_v"]),
               ligand_v_aux=z(B * NLm, prep["ligand_v_aux"]).index_copy(0, rows_l, prep["ligand_v_aux"]),
               bond_type=z(B * Ebm, prep["ligand_fc_bond_type"]).index_copy(0, rows_b, prep["ligand_fc_bond_type"]),
               pos_noise=z(B * NLm, prep["pos_noise"]).index_copy(0, rows_l, prep["pos_noise"]),
               u_v=z(B * NLm, prep["u_v"], 0.5).index_copy(0, rows_l, prep["u_v"]),
               u_b=z(B * Ebm, prep["u_b"], 0.5).index_copy(0, rows_b, prep["u_b"]),
               centers=torch.zeros(B * NLm, 3, device=dev).index_copy(0, rows_l, prep["prior_centers"][dec].float()),
               stds=torch.ones(B * NLm, 3, device=dev).index_copy(0, rows_l, prep["prior_stds"][dec].float()),
               w_p=torch.zeros(B * NPm, device=dev).index_fill(0, rows_p, 1.0),
               w_l=torch.zeros(B * NLm, device=dev).index_fill(0, rows_l, 1.0),
               w_b=torch.zeros(B * Ebm, device=dev).index_fill(0, rows_b, 1.0),
               np_real=torch.tensor(n_p, dtype=torch.int32, device=dev), nl_real=torch.tensor(n_l, dtype=torch.int32, device=dev))
    # (counts clamped to 1 like torch_scatter's scatter_mean: a one-atom ligand has no bonds and an empty pocket no atoms -- 0 / 0
    #  would poison the step, the ragged path and the reference give 0)
    out["cnt_p"], out["cnt_l"] = out["np_real"].float().clamp(min=1.0), out["nl_real"].float().clamp(min=1.0)
    out["cnt_b"] = (out["nl_real"].float() * (out["nl_real"].float() - 1.0)).clamp(min=1.0)
    out["rows_l"], out["rows_b"] = rows_l, rows_b           # (for callers that want the real rows back; not used by the objective)
    return out


PAD_TENSORS = ("time_step", "protein_pos", "protein_v", "ligand_pos", "ligand_v", "ligand_v_aux", "bond_type", "pos_noise", "u_v", "u_b",
               "centers", "stds", "w_p", "w_l", "w_b", "np_real", "nl_real", "cnt_p", "cnt_l", "cnt_b")


def objective_padded(model, pp: Dict) -> Dict:
    """`objective` on a padded batch (`pad_prepared`): the same forward diffusion, network and losses with every per-sample mean
    taken over the real rows (weights 0 / 1, real counts).  Shapes depend on (B, NPm, NLm) only and nothing touches the host, so
    mixed-size batches of one shape bucket share ONE captured graph (`GraphedTrainStep`)."""
    B, NPm, NLm = pp["B"], pp["NPm"], pp["NLm"]
    dev = pp["protein_pos"].device
    time_step = pp["time_step"]
    b_p, b_l, fc = model._expected_layout(B, NPm, NLm, dev)
    b_b = torch.arange(B, device=dev).repeat_interleave(NLm * (NLm - 1))
    a = model.alphas_cumprod.index_select(0, time_step)
    a_pos = a[b_l].unsqueeze(-1)
    ligand_pos, centers, stds = pp["ligand_pos"], pp["centers"], pp["stds"]
    pos_pert = a_pos.sqrt() * (ligand_pos - centers) + (1.0 - a_pos).sqrt() * pp["pos_noise"] * stds + centers
    tv, tb = _Trans(model.atom_type_trans), _Trans(model.bond_type_trans)
    log_v0 = _log_onehot(pp["ligand_v"], model.num_classes)
    v_pert = (-torch.log(-torch.log(pp["u_v"] + 1e-30) + 1e-30) + tv.pred(log_v0, time_step, b_l)).argmax(-1)
    log_vt = _log_onehot(v_pert, model.num_classes)
    log_b0 = _log_onehot(pp["bond_type"], model.num_bond_classes)
    b_pert = (-torch.log(-torch.log(pp["u_b"] + 1e-30) + 1e-30) + tb.pred(log_b0, time_step, b_b)).argmax(-1)
    log_bt = _log_onehot(b_pert, model.num_bond_classes)
    if model.center_pos_mode == "protein":                 # mean of the REAL protein rows of each sample
        offset = (pp["protein_pos"] * pp["w_p"].unsqueeze(-1)).view(B, NPm, 3).sum(1) / pp["cnt_p"].unsqueeze(-1)
    elif model.center_pos_mode == "none":
        offset = torch.zeros(B, 3, device=dev)
    else:
        raise NotImplementedError(model.center_pos_mode)
    p_pos = pp["protein_pos"] - offset[b_p]
    x_t = pos_pert - offset[b_l]
    x_0 = ligand_pos - offset[b_l]
    preds = network(model, p_pos, pp["protein_v"], b_p, x_t, v_pert, pp["ligand_v_aux"], b_l, fc, b_pert, checked_B=B,
                    pad=dict(np_real=pp["np_real"], nl_real=pp["nl_real"]))
    pred_pos, pred_v = preds["pred_ligand_pos"], preds["pred_ligand_v"]
    plan_l, plan_b = _static_plan(b_l, B, [NLm] * B, False), _static_plan(b_b, B, [NLm] * B, True)

    def sample_mean(per_row, w, plan, cnt):                # scatter_mean over the real rows of every sample
        return scatter_sum((per_row * w).unsqueeze(-1), plan).squeeze(-1) / cnt

    def v_loss(log_model, log_v0_, log_true, batch, w, plan, cnt):
        kl = (log_true.exp() * (log_true - log_model)).sum(1)
        nll = -(log_v0_.exp() * log_model).sum(1)
        mask = (time_step == 0).float()[batch]
        return sample_mean(mask * nll + (1.0 - mask) * kl, w, plan, cnt)

    log_v_recon = F.log_softmax(pred_v, dim=-1)
    kl_v = v_loss(tv.posterior(log_v_recon, log_vt, time_step, b_l), log_v0, tv.posterior(log_v0, log_vt, time_step, b_l), b_l,
                  pp["w_l"], plan_l, pp["cnt_l"])
    log_b_recon = F.log_softmax(preds["pred_bond"], dim=-1)
    kl_b = v_loss(tb.posterior(log_b_recon, log_bt, time_step, b_b), log_b0, tb.posterior(log_b0, log_bt, time_step, b_b), b_b,
                  pp["w_b"], plan_b, pp["cnt_b"])
    if model.loss_pos_type != "mse":
        raise ValueError(model.loss_pos_type)
    loss_pos = sample_mean((((pred_pos - x_0) ** 2) / (stds ** 2)).sum(-1), pp["w_l"], plan_l, pp["cnt_l"]).mean()
    return {"losses": {"pos": loss_pos, "v": kl_v.mean(), "bond": kl_b.mean()}, "pred_ligand_pos": pred_pos, "pred_ligand_v": pred_v,
            "pred_bond": preds["pred_bond"], "x0": x_0, "time_step": time_step}


def diffusion_loss(model, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, ligand_v_aux, batch_ligand,
                   prior_centers, prior_stds, prior_num_atoms, batch_prior, ligand_decomp_batch, ligand_fc_bond_index,
                   ligand_fc_bond_type, batch_ligand_bond, time_step=None, network_fn=None) -> Dict:
    """get_diffusion_loss (decompdiff.py:419-550) = `prepare_batch` (host: checks, time steps, noise on the CPU generator in the
    reference's order, so a seeded call reproduces the reference's CPU run) + `objective` (device)."""
    prep = prepare_batch(model, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, ligand_v_aux, batch_ligand,
                         prior_centers, prior_stds, prior_num_atoms, batch_prior, ligand_decomp_batch, ligand_fc_bond_index,
                         ligand_fc_bond_type, batch_ligand_bond, time_step=time_step)
    return objective(model, prep, network_fn=network_fn)



# --------------------------------------------------------------------------------------------------------------------
# One training iteration as ONE captured graph (dense batches of a fixed shape)
# --------------------------------------------------------------------------------------------------------------------
class GraphedTrainStep:
    """loss + backward + optimizer step of scripts/train_diffusion_decomp.py's inner loop (get_diffusion_loss, the weighted sum of
    its three losses, backward, optimizer.step) replayed as one hipGraph per batch shape.

    The eager step is bound by the host: ~6 000 launches of 5-15 us kernels (`profiles/round4_train_step.txt`).  Everything of
    a step that needs the host -- layout checks, the time-step and noise draws on the CPU generator (the reference's order, so
    seeded runs stay comparable) -- is `prepare_batch`; the rest (`objective`, backward, the optimizer's update) touches the
    device only and is captured with torch.cuda.graph after `warmup` eager iterations of that shape (on a side stream, as
    torch's capture rules ask).  Later iterations copy the prepared tensors into the graph's static inputs and replay it.
    Batches whose samples differ in size go through the padded layout of their shape bucket (`pad_prepared` / `objective_padded`:
    sizes rounded up to `bucket`), so all batches of a bucket share one graph.

    The optimizer must be built with ``capturable=True`` (torch.optim.Adam / AdamW): its step counter then lives on the device.
    A captured graph holds the ADDRESSES of the parameters, their gradients and the optimizer state: in-place updates
    (optimizer steps, ``load_state_dict``) are seen by the next replay, but anything that gives the parameters new storage
    (``model.to(...)``, re-creating the optimizer) needs a new ``GraphedTrainStep``.  At most ``max_graphs`` graphs are kept (least
    recently used first out); each holds the activations of one iteration of its shape (~5 GB at B = 4, 300 + 30 atoms).
    ``step(**kw)`` takes get_diffusion_loss's keyword arguments and returns {"loss", "losses": {pos, v, bond}} (detached).

    ``max_grad_norm`` (the reference clips: ``clip_grad_norm_(model.parameters(), config.train.max_grad_norm)`` between backward and
    ``optimizer.step()``, scripts/train_diffusion_decomp.py:195; configs/training.yml: 8.0): the same call (foreach, no host
    sync) inside the captured region and the eager steps.  Learning rate: with ``capturable=True`` a Python-float ``lr`` is baked
    into the graph at capture time -- pass ``lr=torch.tensor(...)`` (a scheduler then updates it in place and replays see it); if
    a float ``lr`` of a param group changes (ReduceLROnPlateau), the graphs captured with the old value are dropped and re-captured.
    Everything a capture read from the module-level caches (index structures, segment plans) is pinned in its graph entry."""

    def __init__(self, model, optimizer, loss_weights=(1.0, 100.0, 100.0), warmup: int = 3, max_graphs: int = 4, bucket=(32, 4),
                 max_grad_norm=None):
        if not optimizer.defaults.get("capturable", False):
            raise ValueError("GraphedTrainStep needs an optimizer built with capturable=True (its state must live on the device)")
        self.model, self.opt, self.w = model, optimizer, tuple(float(x) for x in loss_weights)
        self.warmup, self.max_graphs, self.bucket = int(warmup), int(max_graphs), (int(bucket[0]), int(bucket[1]))
        self._seen: Dict = {}
        self._graphs: Dict = {}
        self._side = None
        self.replays = self.eager_steps = 0
        self.max_grad_norm = None if max_grad_norm is None else float(max_grad_norm)
        self._params = [p for g in optimizer.param_groups for p in g["params"]]

    def _clip(self):
        if self.max_grad_norm is not None:
            torch.nn.utils.clip_grad_norm_(self._params, self.max_grad_norm, foreach=True)

    def _float_lrs(self):
        return tuple(g["lr"] if not torch.is_tensor(g["lr"]) else None for g in self.opt.param_groups)

    def _total(self, res):
        lo = res["losses"]
        return self.w[0] * lo["pos"] + self.w[1] * lo["v"] + self.w[2] * lo["bond"]

    def _eager(self, prep, fn=None):
        self.opt.zero_grad(set_to_none=True)
        res = (fn or objective)(self.model, prep)
        loss = self._total(res)
        loss.backward()
        self._clip()
        self.opt.step()
        self.eager_steps += 1
        return {"loss": loss.detach(), "losses": {k: v.detach() for k, v in res["losses"].items()}}

    def step(self, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, ligand_v_aux, batch_ligand, prior_centers, prior_stds,
             prior_num_atoms, batch_prior, ligand_decomp_batch, ligand_fc_bond_index, ligand_fc_bond_type, batch_ligand_bond,
             time_step=None, **unused):
        model = self.model
        model.__dict__["_packed"] = None                   # parameters are about to change: never reuse a packed copy
        prep = prepare_batch(model, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, ligand_v_aux, batch_ligand,
                             prior_centers, prior_stds, prior_num_atoms, batch_prior, ligand_decomp_batch, ligand_fc_bond_index,
                             ligand_fc_bond_type, batch_ligand_bond, time_step=time_step)
        n_p, n_l = prep["sizes"]
        dense = len(set(zip(n_p, n_l))) == 1
        dev = protein_pos.device
        if os.environ.get("DD_TRAIN_GRAPH", "1") == "0":
            return self._eager(prep)
        if dense:
            key = (tuple(n_p), tuple(n_l), int(prior_centers.shape[0]), str(dev))
            data, names, fn = prep, PREP_TENSORS, objective
        else:
            # samples of different sizes: the padded layout of their shape bucket (sizes rounded up to `bucket`) -- all batches
            # whose largest protein / ligand fall into one bucket share a graph
            data = pad_prepared(model, prep, self.bucket)
            if data is None:
                return self._eager(prep)
            key = ("padded", data["B"], data["NPm"], data["NLm"], str(dev))
            names, fn = PAD_TENSORS, objective_padded
        ent = self._graphs.get(key)
        if ent is not None and ent["lrs"] != self._float_lrs():   # a float lr changed since the capture: it is baked into the graph
            self._graphs.pop(key)
            ent = None
        if ent is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(device=dev)
            cur = torch.cuda.current_stream(dev)
            self._side.wait_stream(cur)
            if n < self.warmup:                            # eager iterations of this shape first (real steps, on the side stream)
                with torch.cuda.stream(self._side):
                    out = self._eager(data, fn)
                cur.wait_stream(self._side)
                return out
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            static = dict(data)
            for k in names:
                static[k] = data[k].clone()
            self.opt.zero_grad(set_to_none=True)
            graph = torch.cuda.CUDAGraph()
            global _PIN
            pins, _PIN = [], []
            try:
                with torch.cuda.graph(graph, stream=self._side):
                    res = fn(model, static)
                    loss = self._total(res)
                    loss.backward()
                    self._clip()
                    self.opt.step()
                pins = _PIN
            finally:
                _PIN = None
            cur.wait_stream(self._side)
            ent = self._graphs[key] = dict(graph=graph, static=static, names=names, loss=loss.detach(), keep=pins, lrs=self._float_lrs(),
                                           losses={k: v.detach() for k, v in res["losses"].items()})
            # (the capture itself computed nothing: this iteration's update happens in the replay below)
        else:
            self._graphs[key] = self._graphs.pop(key)      # most recently used last
        for k in ent["names"]:
            ent["static"][k].copy_(data[k], non_blocking=True)
        ent["graph"].replay()
        self.replays += 1
        return {"loss": ent["loss"].clone(), "losses": {k: v.clone() for k, v in ent["losses"].items()}}
