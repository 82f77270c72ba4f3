"""Re-layout of the reference checkpoint tensors into the HIP kernels' packed form.

Pure data movement (slices / transposes / concatenations of fp32 tensors), no arithmetic
except folding first-Linear biases into the projection biases.  Column layouts of the
first Linear of every edge MLP follow SURVEY.md Appendix C
(/root/reference/models/encoders/uni_transformer_edge.py:48,148-149,194,268-269):

* 340-column MLPs (kNN edges)  : [0:80] type⊗Gaussian (col = type*20+g) | [80:84] type one-hot |
                                 [84:212] h[dst] | [212:340] h[src]
* 384-column MLPs (bond edges) : [0:128] h_bond[e] | [128:256] h[dst] | [256:384] h[src]
* bond_layer kv (437)          : [0:128] h_bond[kj] | [128:148] G(d_kj) | [148:168] G(d_ji) |
                                 [168:181] angle code | [181:309] h[k] | [309:437] h[j]
* bond_layer q (256)           : [0:128] h_bond[ji] | [128:256] h[i]

Because the first Linear is linear in a concatenation, ``W·[a;b;c] = W_a·a + W_b·b + W_c·c``
is evaluated as per-node / per-bond projections (dense GEMMs) that the fused attention
kernels gather and sum per edge / per triplet (exact algebra; fp32 summation order differs
from the reference by ~1e-7 relative).

The packed arena is one flat fp32 tensor; ``slots`` maps ``(layer, name)`` to (offset, shape).
The same slot names are declared in include/decompdiff_hip.h (enum dd_wslot).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import torch

H = 128
NG = 20          # Gaussian centres
NGP = 21         # + the per-type constant column
NA = 13          # angular code width
NH = 16

# Order is the C enum order (include/decompdiff_hip.h: enum dd_wslot). Do not reorder.
LAYER_SLOTS = [
    # --- projections on the *old* h / h_bond ---------------------------------------
    "W_n1", "b_n1",          # [640,128] all nodes: NE kd|ks|vd|vs|q1
    "W_l1", "b_l1",          # [1280,128] ligand nodes: NB kd|ks|vd|vs|q1 , BL k_hk|k_hj|v_hk|v_hj|q_hi
    "W_b1", "b_b1",          # [640,128] bond edges: NB ke|ve , BL k_hb|v_hb|q_hb
    # --- node_layer_with_edge (NE) ---------------------------------------------------
    "NE_Ak", "NE_Av",        # [4,21,128] Gaussian/type tables
    "NE_lnk", "NE_lnv", "NE_lnq",   # [2,128] gamma;beta
    "NE_W2q", "NE_b2q",      # [128,128],[128]
    "NE_W2k",                # [128,128]  (d, c)
    "NE_W2vT", "NE_b2v",     # [128(c),128(o)], [128]
    "NE_W2v",                # [128(o),128(c)] (tiled kernel: head-permuted LDS image)
    "NE_Akp", "NE_Avp",      # [4,24,128] the tables as MFMA A operands: rows padded with zeros, channels permuted
    # --- node_layer_with_bond (NB) ---------------------------------------------------
    "NB_lnk", "NB_lnv", "NB_lnq", "NB_W2q", "NB_b2q", "NB_W2k", "NB_W2vT", "NB_b2v", "NB_W2v",
    # --- bond_layer (BL) ---------------------------------------------------------------
    "BL_Wg1k", "BL_Wg1v",    # [20,128]  G(d_kj) columns
    "BL_Wg2k", "BL_Wg2v",    # [20,128]  G(d_ji) columns
    "BL_Wak", "BL_Wav",      # [13,128]  angle-code columns
    "BL_lnk", "BL_lnv", "BL_lnq", "BL_W2q", "BL_b2q", "BL_W2k", "BL_W2vT", "BL_b2v", "BL_W2v",
    "BL_Wgp",                # [4,20,128] Wg1k | Wg1v | Wg2k | Wg2v in the MFMA A-operand layout (assemble kernel)
    "BL_Wakp", "BL_Wavp",    # [12,128] angle-code columns, MFMA A-operand layout, duplicate codes merged (see _ANGLE_MERGE)
    # --- lin_node --------------------------------------------------------------------------
    "W_lin", "b_lin",
    # --- projections on the *new* h / h_bond -------------------------------------------
    "W_n2", "b_n2",          # [256,128] all nodes: PE ks|vs
    "W_l2", "b_l2",          # [1024,128] ligand nodes: PE kd|vd|q1 , PB kd|ks|vd|vs|q1
    "W_b2", "b_b2",          # [256,128] bond edges: PB ke|ve
    # --- pos_layer_with_edge (PE) ----------------------------------------------------------
    "PE_Ak", "PE_Av", "PE_lnk", "PE_lnv", "PE_lnq", "PE_W2q", "PE_b2q", "PE_W2k",
    "PE_W2v", "PE_b2v",      # [16,128],[16]
    "PE_Akp", "PE_Avp",      # [4,24,128]
    "PE_W2qT",               # [128(k),128(o)] (query MLP second layer evaluated inside the coordinate kernel)
    # --- pos_layer_with_bond (PB) ----------------------------------------------------------
    "PB_lnk", "PB_lnv", "PB_lnq", "PB_W2q", "PB_b2q", "PB_W2k", "PB_W2v", "PB_b2v", "PB_W2qT",
]
GLOBAL_SLOTS = [
    "W_pemb", "b_pemb",      # [128,29]/[128]  protein_atom_emb padded to 128 rows (row 127 = node indicator 0)
    "W_lemb", "b_lemb",      # [128,10]/[128]  ligand_atom_emb padded (row 127: weight 0, bias 1)
    "W_bemb", "b_bemb",      # [128,5]/[128]   ligand_bond_emb
    "EW_W1T", "EW_b1", "EW_ln", "EW_w2", "EW_b2",   # edge_pred_layer: [20,128],[128],[2,128],[128],[1]
    "VH_W1", "VH_b1", "VH_W2", "VH_b2",             # v_inference: [128,128],[128],[8,128],[8]
    "BH_W1", "BH_b1", "BH_W2", "BH_b2",             # bond_inference: [128,128],[128],[5,128],[5]
]


def _mlp(sd, name):
    return (sd[name + ".net.0.weight"], sd[name + ".net.0.bias"],
            torch.stack([sd[name + ".net.1.weight"], sd[name + ".net.1.bias"]], 0),
            sd[name + ".net.3.weight"], sd[name + ".net.3.bias"])


def _gauss_table(W1):
    """[128,340] first-Linear weight → [4,21,128] table: [t][g<20][c]=W1[c,t*20+g], [t][20][c]=W1[c,80+t]."""
    t = torch.empty(4, NGP, H, dtype=W1.dtype)
    for ty in range(4):
        t[ty, :NG] = W1[:, ty * NG:(ty + 1) * NG].t()
        t[ty, NG] = W1[:, 80 + ty]
    return t


# channel order of the MFMA A-operand tables: position half*64 + mm*4 + r holds channel 16*(4*half + r) + mm,
# so lane (mm, cg) of the tiled attention kernel reads its 8 channel tiles of row g with two 16-byte LDS reads
_MFMA_PERM = torch.tensor([16 * (4 * half + r) + mm for half in range(2) for mm in range(16) for r in range(4)])


def _mfma_rows(W, rows):
    """[..., g, 128] -> [..., rows, 128]: zero rows appended (k padded to a multiple of 4), channels permuted."""
    out = torch.zeros(*W.shape[:-2], rows, H, dtype=W.dtype)
    out[..., :W.shape[-2], :] = W[..., _MFMA_PERM]
    return out


# AngularEncoding emits [th, sin(f th), cos(f th)] with f = [1, 2, 3, 1, 1/2, 1/3]: sin th and cos th appear twice.
# Their weight columns are added on the host, which leaves 11 distinct codes = 3 MFMA k-steps instead of 4:
#   [th, sin th, sin 2th, sin 3th | sin th/2, sin th/3, cos th, cos 2th | cos 3th, cos th/2, cos th/3, 0]
_ANGLE_MERGE = [(0,), (1, 4), (2,), (3,), (5,), (6,), (7, 10), (8,), (9,), (11,), (12,)]


def _angle_rows(W):
    """[13,128] -> [12,128] merged angle-code rows in the MFMA operand layout."""
    m = torch.stack([sum(W[i] for i in grp) for grp in _ANGLE_MERGE], 0)
    return _mfma_rows(m, 12)


def pack_layer(sd: Dict[str, torch.Tensor], prefix: str) -> "OrderedDict[str, torch.Tensor]":
    z = lambda n: torch.zeros(n)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    ne = {f: _mlp(sd, f"{prefix}.node_layer_with_edge.{f}") for f in ("hk_func", "hv_func", "hq_func")}
    nb = {f: _mlp(sd, f"{prefix}.node_layer_with_bond.{f}") for f in ("hk_func", "hv_func", "hq_func")}
    bl = {f: _mlp(sd, f"{prefix}.bond_layer.{f}") for f in ("hk_func", "hv_func", "hq_func")}
    pe = {f: _mlp(sd, f"{prefix}.pos_layer_with_edge.{f}") for f in ("xk_func", "xv_func", "xq_func")}
    pb = {f: _mlp(sd, f"{prefix}.pos_layer_with_bond.{f}") for f in ("xk_func", "xv_func", "xq_func")}

    k, v, q = ne["hk_func"], ne["hv_func"], ne["hq_func"]
    out["W_n1"] = torch.cat([k[0][:, 84:212], k[0][:, 212:340], v[0][:, 84:212], v[0][:, 212:340], q[0]], 0)
    out["b_n1"] = torch.cat([k[1], z(H), v[1], z(H), q[1]])
    nk, nv, nq = nb["hk_func"], nb["hv_func"], nb["hq_func"]
    bk, bv, bq = bl["hk_func"], bl["hv_func"], bl["hq_func"]
    out["W_l1"] = torch.cat([nk[0][:, 128:256], nk[0][:, 256:384], nv[0][:, 128:256], nv[0][:, 256:384], nq[0],
                             bk[0][:, 181:309], bk[0][:, 309:437], bv[0][:, 181:309], bv[0][:, 309:437],
                             bq[0][:, 128:256]], 0)
    out["b_l1"] = torch.cat([nk[1], z(H), nv[1], z(H), nq[1], z(H), z(H), z(H), z(H), z(H)])
    out["W_b1"] = torch.cat([nk[0][:, 0:128], nv[0][:, 0:128], bk[0][:, 0:128], bv[0][:, 0:128], bq[0][:, 0:128]], 0)
    out["b_b1"] = torch.cat([z(H), z(H), bk[1], bv[1], bq[1]])

    out["NE_Ak"], out["NE_Av"] = _gauss_table(k[0]), _gauss_table(v[0])
    out["NE_lnk"], out["NE_lnv"], out["NE_lnq"] = k[2], v[2], q[2]
    out["NE_W2q"], out["NE_b2q"] = q[3], q[4]
    out["NE_W2k"] = k[3]
    out["NE_W2vT"], out["NE_b2v"] = v[3].t().contiguous(), v[4]
    out["NE_W2v"] = v[3]
    out["NE_Akp"], out["NE_Avp"] = _mfma_rows(out["NE_Ak"], 24), _mfma_rows(out["NE_Av"], 24)

    out["NB_lnk"], out["NB_lnv"], out["NB_lnq"] = nk[2], nv[2], nq[2]
    out["NB_W2q"], out["NB_b2q"], out["NB_W2k"] = nq[3], nq[4], nk[3]
    out["NB_W2vT"], out["NB_b2v"] = nv[3].t().contiguous(), nv[4]
    out["NB_W2v"] = nv[3]

    out["BL_Wg1k"], out["BL_Wg1v"] = bk[0][:, 128:148].t().contiguous(), bv[0][:, 128:148].t().contiguous()
    out["BL_Wg2k"], out["BL_Wg2v"] = bk[0][:, 148:168].t().contiguous(), bv[0][:, 148:168].t().contiguous()
    out["BL_Wak"], out["BL_Wav"] = bk[0][:, 168:181].t().contiguous(), bv[0][:, 168:181].t().contiguous()
    out["BL_lnk"], out["BL_lnv"], out["BL_lnq"] = bk[2], bv[2], bq[2]
    out["BL_W2q"], out["BL_b2q"], out["BL_W2k"] = bq[3], bq[4], bk[3]
    out["BL_W2vT"], out["BL_b2v"] = bv[3].t().contiguous(), bv[4]
    out["BL_W2v"] = bv[3]
    out["BL_Wgp"] = torch.stack([_mfma_rows(out[k], 20) for k in ("BL_Wg1k", "BL_Wg1v", "BL_Wg2k", "BL_Wg2v")], 0)
    out["BL_Wakp"], out["BL_Wavp"] = _angle_rows(out["BL_Wak"]), _angle_rows(out["BL_Wav"])

    out["W_lin"], out["b_lin"] = sd[f"{prefix}.lin_node.weight"], sd[f"{prefix}.lin_node.bias"]

    xk, xv, xq = pe["xk_func"], pe["xv_func"], pe["xq_func"]
    yk, yv, yq = pb["xk_func"], pb["xv_func"], pb["xq_func"]
    out["W_n2"] = torch.cat([xk[0][:, 212:340], xv[0][:, 212:340]], 0)
    out["b_n2"] = z(2 * H)
    out["W_l2"] = torch.cat([xk[0][:, 84:212], xv[0][:, 84:212], xq[0],
                             yk[0][:, 128:256], yk[0][:, 256:384], yv[0][:, 128:256], yv[0][:, 256:384], yq[0]], 0)
    out["b_l2"] = torch.cat([xk[1], xv[1], xq[1], yk[1], z(H), yv[1], z(H), yq[1]])
    out["W_b2"] = torch.cat([yk[0][:, 0:128], yv[0][:, 0:128]], 0)
    out["b_b2"] = z(2 * H)

    out["PE_Ak"], out["PE_Av"] = _gauss_table(xk[0]), _gauss_table(xv[0])
    out["PE_lnk"], out["PE_lnv"], out["PE_lnq"] = xk[2], xv[2], xq[2]
    out["PE_W2q"], out["PE_b2q"], out["PE_W2k"] = xq[3], xq[4], xk[3]
    out["PE_W2v"], out["PE_b2v"] = xv[3], xv[4]
    out["PE_Akp"], out["PE_Avp"] = _mfma_rows(out["PE_Ak"], 24), _mfma_rows(out["PE_Av"], 24)
    out["PE_W2qT"] = xq[3].t().contiguous()
    out["PB_lnk"], out["PB_lnv"], out["PB_lnq"] = yk[2], yv[2], yq[2]
    out["PB_W2q"], out["PB_b2q"], out["PB_W2k"] = yq[3], yq[4], yk[3]
    out["PB_W2v"], out["PB_b2v"] = yv[3], yv[4]
    out["PB_W2qT"] = yq[3].t().contiguous()
    assert list(out.keys()) == LAYER_SLOTS
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Kernel form of the attention MLPs (round 6).  Every key / value MLP of the five attention sub-layers is
#     z = relu(LayerNorm(P) * gamma + beta),  P = sum of first-Linear parts (projection rows, Gaussian / angle tables, bias),
# followed by a second Linear that is linear in z (W2k inside the folded query, W2v after the aggregation).  Three exact
# identities move work from the kernels' VALU stream to the host:
#   1. LayerNorm ignores a per-row constant, so the channel mean of every additive part is removed here
#      (W1 <- (I - 11^T/128) W1): the rows the kernels sum are mean-free and the variance is E[P^2] -- no mean pass.
#   2. relu(g * xh + b) = |g| * relu(sign(g) * xh + b / |g|): the sign goes into the first-Linear rows, |g| into the COLUMNS of
#      the second Linear, and the kernels evaluate max(fma(P, rstd, beta'), 0) -- one FMA + one max per element.
#      (gamma == 0 is carried as |g| = 1e-30: beta' = beta * 1e30 stays finite for |beta| < 3e8 and |g| * relu(.) gives relu(beta).)
#   3. The softmax scale 1/sqrt(8) of the folded query goes into W2k.
# The slots keep their shapes and order; the canonical (reference-valued) form is what pack_layer returns and what
# tests/dense_spec.py consumes.  libdecompdiff_hip.so reports which form its kernels expect (dd_weights_form()).
_SCALE = 0.35355339059327373       # 1 / sqrt(head dim 8)
_GAMMA_FLOOR = 1e-30

# per MLP: first-Linear row blocks [(W slot, b slot, first row)], channel-last tables, the LayerNorm slot, second Linears
# [(slot, channel axis, extra factor)], derived MFMA operand slots are rebuilt afterwards
_KERNEL_FORM_MLPS = [
    dict(rows=[("W_n1", "b_n1", 0), ("W_n1", "b_n1", 128)], tabs=["NE_Ak"], ln="NE_lnk", w2=[("NE_W2k", 1, _SCALE)]),
    dict(rows=[("W_n1", "b_n1", 256), ("W_n1", "b_n1", 384)], tabs=["NE_Av"], ln="NE_lnv", w2=[("NE_W2v", 1, 1.0), ("NE_W2vT", 0, 1.0)]),
    dict(rows=[("W_l1", "b_l1", 0), ("W_l1", "b_l1", 128), ("W_b1", "b_b1", 0)], tabs=[], ln="NB_lnk", w2=[("NB_W2k", 1, _SCALE)]),
    dict(rows=[("W_l1", "b_l1", 256), ("W_l1", "b_l1", 384), ("W_b1", "b_b1", 128)], tabs=[], ln="NB_lnv",
         w2=[("NB_W2v", 1, 1.0), ("NB_W2vT", 0, 1.0)]),
    dict(rows=[("W_l1", "b_l1", 640), ("W_l1", "b_l1", 768), ("W_b1", "b_b1", 256)], tabs=["BL_Wg1k", "BL_Wg2k", "BL_Wak"],
         ln="BL_lnk", w2=[("BL_W2k", 1, _SCALE)]),
    dict(rows=[("W_l1", "b_l1", 896), ("W_l1", "b_l1", 1024), ("W_b1", "b_b1", 384)], tabs=["BL_Wg1v", "BL_Wg2v", "BL_Wav"],
         ln="BL_lnv", w2=[("BL_W2v", 1, 1.0), ("BL_W2vT", 0, 1.0)]),
    dict(rows=[("W_n2", "b_n2", 0), ("W_l2", "b_l2", 0)], tabs=["PE_Ak"], ln="PE_lnk", w2=[("PE_W2k", 1, _SCALE)]),
    dict(rows=[("W_n2", "b_n2", 128), ("W_l2", "b_l2", 128)], tabs=["PE_Av"], ln="PE_lnv", w2=[("PE_W2v", 1, 1.0)]),
    dict(rows=[("W_l2", "b_l2", 384), ("W_l2", "b_l2", 512), ("W_b2", "b_b2", 0)], tabs=[], ln="PB_lnk", w2=[("PB_W2k", 1, _SCALE)]),
    dict(rows=[("W_l2", "b_l2", 640), ("W_l2", "b_l2", 768), ("W_b2", "b_b2", 128)], tabs=[], ln="PB_lnv", w2=[("PB_W2v", 1, 1.0)]),
]


def kernel_form_layer(out: "OrderedDict[str, torch.Tensor]") -> "OrderedDict[str, torch.Tensor]":
    """Canonical packed layer (pack_layer) -> the form the round-6 attention kernels consume (see above).  Returns a new dict;
    arithmetic in float64, rounded once to fp32."""
    o = OrderedDict((k, v.clone().double()) for k, v in out.items())
    for m in _KERNEL_FORM_MLPS:
        g, be = o[m["ln"]][0].clone(), o[m["ln"]][1].clone()
        s = torch.where(g < 0, -torch.ones_like(g), torch.ones_like(g))
        a = g.abs().clamp_min(_GAMMA_FLOOR)
        for (wn, bn, r0) in m["rows"]:
            W, b = o[wn], o[bn]
            blk = W[r0:r0 + H]
            W[r0:r0 + H] = (blk - blk.mean(0, keepdim=True)) * s[:, None]
            bb = b[r0:r0 + H]
            b[r0:r0 + H] = (bb - bb.mean()) * s
        for tn in m["tabs"]:
            t = o[tn]
            o[tn] = (t - t.mean(-1, keepdim=True)) * s
        o[m["ln"]] = torch.stack([torch.ones_like(g), be / a], 0)
        for (w2n, axis, fac) in m["w2"]:
            shape = [1, H] if axis == 1 else [H, 1]
            o[w2n] = o[w2n] * (a * fac).reshape(shape)
    res = OrderedDict((k, v.float()) for k, v in o.items())
    # derived MFMA-operand images of the transformed tables
    res["NE_Akp"], res["NE_Avp"] = _mfma_rows(res["NE_Ak"], 24), _mfma_rows(res["NE_Av"], 24)
    res["PE_Akp"], res["PE_Avp"] = _mfma_rows(res["PE_Ak"], 24), _mfma_rows(res["PE_Av"], 24)
    res["BL_Wgp"] = torch.stack([_mfma_rows(res[k], 20) for k in ("BL_Wg1k", "BL_Wg1v", "BL_Wg2k", "BL_Wg2v")], 0)
    res["BL_Wakp"], res["BL_Wavp"] = _angle_rows(res["BL_Wak"]), _angle_rows(res["BL_Wav"])
    assert list(res.keys()) == LAYER_SLOTS
    return res


def pack_global(sd, cfg) -> "OrderedDict[str, torch.Tensor]":
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    assert cfg.node_indicator and cfg.hidden_dim == H

    def pad_emb(wname, ind):
        W, b = sd[wname + ".weight"], sd[wname + ".bias"]
        Wp = torch.zeros(H, W.size(1))
        bp = torch.zeros(H)
        Wp[:H - 1], bp[:H - 1] = W, b
        bp[H - 1] = ind                       # node indicator column (decompdiff.py:245-256)
        return Wp, bp
    out["W_pemb"], out["b_pemb"] = pad_emb("protein_atom_emb", 0.0)
    out["W_lemb"], out["b_lemb"] = pad_emb("ligand_atom_emb", 1.0)
    out["W_bemb"], out["b_bemb"] = sd["ligand_bond_emb.weight"], sd["ligand_bond_emb.bias"]
    ew = _mlp(sd, "refine_net.edge_pred_layer")
    out["EW_W1T"], out["EW_b1"], out["EW_ln"] = ew[0].t().contiguous(), ew[1], ew[2]
    out["EW_w2"], out["EW_b2"] = ew[3].reshape(-1), ew[4].reshape(1)
    out["VH_W1"], out["VH_b1"] = sd["v_inference.0.weight"], sd["v_inference.0.bias"]
    out["VH_W2"], out["VH_b2"] = sd["v_inference.2.weight"], sd["v_inference.2.bias"]
    out["BH_W1"], out["BH_b1"] = sd["bond_inference.0.weight"], sd["bond_inference.0.bias"]
    out["BH_W2"], out["BH_b2"] = sd["bond_inference.2.weight"], sd["bond_inference.2.bias"]
    assert list(out.keys()) == GLOBAL_SLOTS
    return out


def pack_model(sd: Dict[str, torch.Tensor], cfg, kernel_form: bool = False):
    """Return (arena fp32 [n], offsets int64 [n_layers*len(LAYER_SLOTS)+len(GLOBAL_SLOTS)], named views).
    kernel_form: the attention MLPs in the form the kernels consume (kernel_form_layer) instead of the canonical,
    reference-valued one (what tests/dense_spec.py reads); same slots, shapes and offsets."""
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    named: "OrderedDict[Tuple[int, str], torch.Tensor]" = OrderedDict()
    for l in range(cfg.num_layers):
        layer = pack_layer(sd, f"refine_net.base_block.{l}")
        if kernel_form:
            layer = kernel_form_layer(layer)
        for k, v in layer.items():
            named[(l, k)] = v.contiguous()
    for k, v in pack_global(sd, cfg).items():
        named[(-1, k)] = v.contiguous()
    offsets, chunks, pos = [], [], 0
    for key, v in named.items():
        offsets.append(pos)
        chunks.append(v.reshape(-1))
        pos += v.numel()
        pad = (-pos) % 64                      # keep every slot 256-byte aligned
        if pad:
            chunks.append(torch.zeros(pad))
            pos += pad
    arena = torch.cat(chunks).contiguous()
    return arena, torch.tensor(offsets, dtype=torch.int64), named
