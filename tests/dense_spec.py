"""Test helper: torch-CPU statement of the *restructured* (dense, factorised) math that the
HIP kernels implement, operating on decompdiff_amd.packing's packed weights.

It exists so that (a) the algebraic restructuring + weight packing are validated on CPU
against the oracle before any GPU run (tests/test_dense_spec.py) and (b) each HIP kernel
has a same-shape reference for its unit parity test.  It is test infrastructure: nothing
in the product path imports it.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

H, NHEAD, DH = 128, 16, 8
OFFS = torch.tensor([0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10])
FREQ = torch.tensor([1.0, 2.0, 3.0, 1.0, 0.5, 1.0 / 3.0])
SCALE = 1.0 / math.sqrt(DH)


def gauss(d):
    return torch.exp(-0.5 * (d.unsqueeze(-1) - OFFS) ** 2)


def ln_relu(x, ln):
    return F.relu(F.layer_norm(x, (H,), ln[0], ln[1], 1e-5))


def knn_dense(x, K):
    """x [B,N,3] → nbr [B,N,K] int64, ascending (d2, index), self excluded."""
    d = x[:, :, None, :] - x[:, None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    N = x.size(1)
    d2 = d2 + torch.diag(torch.full((N,), float("inf")))
    return torch.sort(d2, dim=-1, stable=True).indices[..., :K]


def gather_nodes(t, idx):
    """t [B,N,C], idx [B,...] → t[b, idx[b,...]]"""
    B = t.size(0)
    flat = idx.reshape(B, -1)
    out = torch.gather(t, 1, flat.unsqueeze(-1).expand(-1, -1, t.size(-1)))
    return out.reshape(*idx.shape, t.size(-1))


def edge_weights(x, nbr, g):
    xj = gather_nodes(x, nbr)
    d = (x[:, :, None, :] - xj).norm(dim=-1)
    pre = gauss(d) @ g["EW_W1T"] + g["EW_b1"]
    z = ln_relu(pre, g["EW_ln"])
    return torch.sigmoid((z * g["EW_w2"]).sum(-1) + g["EW_b2"])


def expand_q(q, W2k):
    """Q~[.., h, c] = SCALE * sum_d q[.., h*8+d] * W2k[h*8+d, c]"""
    qh = q.reshape(*q.shape[:-1], NHEAD, DH)
    Wh = W2k.reshape(NHEAD, DH, H)
    return SCALE * torch.einsum("...hd,hdc->...hc", qh, Wh)


def project_out(Zb, S, W2vT, b2v):
    """out[.., h*8+j] = sum_c W2v[h*8+j, c] * Zb[.., h, c] + b2v[h*8+j] * S[.., h]"""
    Wh = W2vT.t().reshape(NHEAD, DH, H)          # [h, j, c]
    out = torch.einsum("...hc,hjc->...hj", Zb, Wh) + b2v.reshape(NHEAD, DH) * S.unsqueeze(-1)
    return out.reshape(*Zb.shape[:-2], H)


def attend_node(zk, zv, Qt, w, W2vT, b2v):
    """zk,zv [...,M,128]; Qt [...,16,128]; w [...,M] → out [...,128]"""
    score = torch.einsum("...mc,...hc->...mh", zk, Qt)
    alpha = torch.softmax(score, dim=-2)
    aw = alpha * w.unsqueeze(-1)
    Zb = torch.einsum("...mh,...mc->...hc", aw, zv)
    return project_out(Zb, aw.sum(-2), W2vT, b2v)


def attend_pos(zk, zv, Qt, w, rel, W2v, b2v):
    score = torch.einsum("...mc,...hc->...mh", zk, Qt)
    alpha = torch.softmax(score, dim=-2)
    v16 = zv @ W2v.t() + b2v                                     # [...,M,16]
    coef = (alpha * w.unsqueeze(-1) * v16).sum(-1)               # [...,M]
    return (coef.unsqueeze(-1) * rel).sum(-2) / NHEAD


def knn_pre(P_dst_k, P_src_k, A, x, nbr, dst_nodes, NP):
    """pre-activation of a 340-column MLP on kNN edges.

    P_dst_k [B,D,128] rows for the D dst nodes `dst_nodes` (node ids, [D]); P_src_k [B,N,128];
    A [4,21,128]; returns pre [B,D,K,128], rel [B,D,K,3]
    """
    nb = nbr[:, dst_nodes]                                       # [B,D,K]
    xj = gather_nodes(x, nb)
    rel = x[:, dst_nodes][:, :, None, :] - xj
    d = rel.norm(dim=-1)
    ty = 2 * (nb < NP).long() + (dst_nodes < NP).long()[None, :, None]
    G = torch.cat([gauss(d), torch.ones_like(d).unsqueeze(-1)], -1)      # [B,D,K,21]
    At = A[ty]                                                            # [B,D,K,21,128]
    pre = P_dst_k[:, :, None, :] + gather_nodes(P_src_k, nb) + torch.einsum("bdkg,bdkgc->bdkc", G, At)
    return pre, rel


def bond_indices(NL):
    """dst-major fc edges: e = i*(NL-1) + j' ; returns src[j] per (i, j') and triplet tables."""
    i = torch.arange(NL).repeat_interleave(NL - 1)
    jp = torch.arange(NL - 1).repeat(NL)
    j = jp + (jp >= i).long()
    return i, j                                                   # dst, src per edge


def edge_id(dst, src, NL):
    return dst * (NL - 1) + src - (src > dst).long()


def layer_forward(L, h, hb, x, nbr, ew, NP, NL):
    """One AttentionLayerO2TwoUpdateNodeGeneral in dense/factorised form.

    h [B,N,128], hb [B,Eb,128] (dst-major fc), x [B,N,3], nbr [B,N,K], ew [B,N,K].
    """
    B, N, _ = h.shape
    K = nbr.size(-1)
    lig = torch.arange(NP, N)
    allv = torch.arange(N)
    hl = h[:, NP:]
    P = h @ L["W_n1"].t() + L["b_n1"]                      # [B,N,640]
    PL = hl @ L["W_l1"].t() + L["b_l1"]                    # [B,NL,1280]
    PB = hb @ L["W_b1"].t() + L["b_b1"]                    # [B,Eb,640]
    e_dst, e_src = bond_indices(NL)

    # ---- NE
    q = ln_relu(P[..., 512:640], L["NE_lnq"]) @ L["NE_W2q"].t() + L["NE_b2q"]
    Qt = expand_q(q, L["NE_W2k"])
    pk, rel = knn_pre(P[..., 0:128], P[..., 128:256], L["NE_Ak"], x, nbr, allv, NP)
    pv, _ = knn_pre(P[..., 256:384], P[..., 384:512], L["NE_Av"], x, nbr, allv, NP)
    A = attend_node(ln_relu(pk, L["NE_lnk"]), ln_relu(pv, L["NE_lnv"]), Qt, ew, L["NE_W2vT"], L["NE_b2v"])

    # ---- NB (ligand dst only)
    q = ln_relu(PL[..., 512:640], L["NB_lnq"]) @ L["NB_W2q"].t() + L["NB_b2q"]
    Qt = expand_q(q, L["NB_W2k"])
    pk = (PB[..., 0:128] + PL[:, e_dst, 0:128] + PL[:, e_src, 128:256]).reshape(B, NL, NL - 1, H)
    pv = (PB[..., 128:256] + PL[:, e_dst, 256:384] + PL[:, e_src, 384:512]).reshape(B, NL, NL - 1, H)
    ones = torch.ones(B, NL, NL - 1)
    A_nb = attend_node(ln_relu(pk, L["NB_lnk"]), ln_relu(pv, L["NB_lnv"]), Qt, ones, L["NB_W2vT"], L["NB_b2v"])
    A = A.clone()
    A[:, NP:] += A_nb

    # ---- BL
    xl = x[:, NP:]
    de = ((xl[:, e_dst] - xl[:, e_src]) ** 2).sum(-1).sqrt()             # [B,Eb]
    Ge = gauss(de)
    Ek = PB[..., 256:384] + Ge @ L["BL_Wg1k"] + PL[:, e_src, 640:768] + PL[:, e_dst, 768:896]
    Ev = PB[..., 384:512] + Ge @ L["BL_Wg1v"] + PL[:, e_src, 896:1024] + PL[:, e_dst, 1024:1152]
    q1 = PB[..., 512:640] + PL[:, e_dst, 1152:1280]
    q = ln_relu(q1, L["BL_lnq"]) @ L["BL_W2q"].t() + L["BL_b2q"]
    Qt = expand_q(q, L["BL_W2k"])                                         # [B,Eb,16,128]
    Rk, Rv = Ge @ L["BL_Wg2k"], Ge @ L["BL_Wg2v"]
    # triplet members: for edge (i<-j): k not in {i,j} ascending
    kk = torch.arange(NL)[None, :].expand(NL * (NL - 1), -1)
    keep = (kk != e_dst[:, None]) & (kk != e_src[:, None])
    kmem = kk[keep].reshape(-1, NL - 2)                                   # [Eb, NL-2]
    kj = edge_id(e_src[:, None].expand_as(kmem), kmem, NL)                # edge (k -> j)
    pi = xl[:, e_dst][:, :, None, :]
    v1 = xl[:, e_src][:, :, None, :] - pi
    v2 = xl[:, kmem] - pi
    a = (v1 * v2).sum(-1)
    bnorm = torch.cross(v1.expand_as(v2), v2, dim=-1).norm(dim=-1)
    th = torch.atan2(bnorm, a)
    code = torch.cat([th.unsqueeze(-1), torch.sin(th.unsqueeze(-1) * FREQ), torch.cos(th.unsqueeze(-1) * FREQ)], -1)
    pk = Ek[:, kj] + Rk[:, :, None, :] + code @ L["BL_Wak"]
    pv = Ev[:, kj] + Rv[:, :, None, :] + code @ L["BL_Wav"]
    ones = torch.ones(B, NL * (NL - 1), NL - 2)
    D = attend_node(ln_relu(pk, L["BL_lnk"]), ln_relu(pv, L["BL_lnv"]), Qt, ones, L["BL_W2vT"], L["BL_b2v"])
    hb_new = hb + D

    h_new = h + A @ L["W_lin"].t() + L["b_lin"]

    # ---- PE / PB on new_h, new_h_bond
    hl2 = h_new[:, NP:]
    P2 = h_new @ L["W_n2"].t() + L["b_n2"]                                # [B,N,256]
    PL2 = hl2 @ L["W_l2"].t() + L["b_l2"]                                 # [B,NL,1024]
    PB2 = hb_new @ L["W_b2"].t() + L["b_b2"]                              # [B,Eb,256]
    q = ln_relu(PL2[..., 256:384], L["PE_lnq"]) @ L["PE_W2q"].t() + L["PE_b2q"]
    Qt = expand_q(q, L["PE_W2k"])
    pk, rel = knn_pre(PL2[..., 0:128], P2[..., 0:128], L["PE_Ak"], x, nbr, lig, NP)
    pv, _ = knn_pre(PL2[..., 128:256], P2[..., 128:256], L["PE_Av"], x, nbr, lig, NP)
    dx_e = attend_pos(ln_relu(pk, L["PE_lnk"]), ln_relu(pv, L["PE_lnv"]), Qt, ew[:, NP:], rel, L["PE_W2v"], L["PE_b2v"])

    q = ln_relu(PL2[..., 896:1024], L["PB_lnq"]) @ L["PB_W2q"].t() + L["PB_b2q"]
    Qt = expand_q(q, L["PB_W2k"])
    pk = (PB2[..., 0:128] + PL2[:, e_dst, 384:512] + PL2[:, e_src, 512:640]).reshape(B, NL, NL - 1, H)
    pv = (PB2[..., 128:256] + PL2[:, e_dst, 640:768] + PL2[:, e_src, 768:896]).reshape(B, NL, NL - 1, H)
    relb = (xl[:, e_dst] - xl[:, e_src]).reshape(B, NL, NL - 1, 3)
    ones = torch.ones(B, NL, NL - 1)
    dx_b = attend_pos(ln_relu(pk, L["PB_lnk"]), ln_relu(pv, L["PB_lnv"]), Qt, ones, relb, L["PB_W2v"], L["PB_b2v"])
    x_new = x.clone()
    x_new[:, NP:] += dx_e + dx_b
    return h_new, hb_new, x_new, dict(A=A, D=D, dx_e=dx_e, dx_b=dx_b)


def forward_dense(named, cfg, xp, fp, xl, v, aux, bond, K=32):
    """Full score-network forward in dense layout.

    xp [B,NP,3] protein pos, fp [B,NP,29] protein features, xl [B,NL,3], v [B,NL] int64,
    aux [B,NL,2], bond [B,Eb] int64 (dst-major fc).  Returns (x0 [B,NL,3], v_logits [B,NL,8],
    b_logits [B,Eb,5], trace).
    """
    g = {k[1]: t for k, t in named.items() if k[0] == -1}
    B, NP, _ = xp.shape
    NL = xl.size(1)
    N = NP + NL
    hp = fp @ g["W_pemb"].t() + g["b_pemb"]
    lf = torch.cat([F.one_hot(v, 8).float(), aux], -1)
    hl = lf @ g["W_lemb"].t() + g["b_lemb"]
    h = torch.cat([hp, hl], 1)
    hb = F.one_hot(bond, 5).float() @ g["W_bemb"].t() + g["b_bemb"]
    x = torch.cat([xp, xl], 1)
    Kk = min(K, N - 1)
    nbr = knn_dense(x, Kk)
    ew = edge_weights(x, nbr, g)
    trace = [dict(nbr=nbr, ew=ew)]
    for l in range(cfg.num_layers):
        L = {k[1]: t for k, t in named.items() if k[0] == l}
        h, hb, x, tr = layer_forward(L, h, hb, x, nbr, ew, NP, NL)
        tr.update(h=h, hb=hb, x=x)
        trace.append(tr)
    sp = lambda t: F.softplus(t) - math.log(2.0)
    vl = sp(h[:, NP:] @ g["VH_W1"].t() + g["VH_b1"]) @ g["VH_W2"].t() + g["VH_b2"]
    bl = sp(hb @ g["BH_W1"].t() + g["BH_b1"]) @ g["BH_W2"].t() + g["BH_b2"]
    return x[:, NP:], vl, bl, trace
