"""ORACLE (test infrastructure only — never imported by the product path).

Plain PyTorch-CPU fp32 restatement of the reference's score network for the shipped
configuration (`uni_o2_bond`, kNN graph, bond diffusion, no prior nodes, no time
embedding): models/decompdiff.py:213-351 + models/encoders/uni_transformer_edge.py +
models/common.py.  It is functional over a flat ``state_dict`` (same key names as the
reference checkpoint) and follows the reference op by op — concatenated edge inputs,
un-factorised Linear layers, edge-list scatter ops — so it is the *specification* the
restructured HIP path is checked against, not an implementation of that restructuring.

Pinned against the reference itself: oracle/make_golden.py imports /root/reference
(with shims for the absent third-party wheels) in the survey container and writes
tests/golden/*.npz; tests/test_oracle_golden.py replays them through this file.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import ops

# models/common.py:18 — every encoder GaussianSmearing uses fix_offset=True
GAUSS_OFFSETS = [0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10]
GAUSS_COEFF = -0.5 / (1.0 - 0.0) ** 2                                  # common.py:23
ANGLE_FREQS = [1.0, 2.0, 3.0, 1.0, 1.0 / 2.0, 1.0 / 3.0]                # common.py:38-40
LOG2 = math.log(2.0)


def gaussian_smearing(dist):
    """GaussianSmearing.forward (models/common.py:29-31) with the fixed 20 offsets."""
    off = torch.tensor(GAUSS_OFFSETS, dtype=dist.dtype, device=dist.device)
    d = dist.reshape(-1, 1) - off.view(1, -1)
    return torch.exp(GAUSS_COEFF * torch.pow(d, 2))


def angular_encoding(angle):
    """AngularEncoding.forward (models/common.py:46-54): [θ, sin(θ f), cos(θ f)]."""
    f = torch.tensor(ANGLE_FREQS, dtype=angle.dtype, device=angle.device)
    a = angle.unsqueeze(-1)
    return torch.cat([a, torch.sin(a * f), torch.cos(a * f)], -1)


def linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def mlp(sd, name, x):
    """MLP(num_layer=2, norm=True, relu) — models/common.py:85-105."""
    y = linear(sd, name + ".net.0", x)
    y = F.layer_norm(y, (y.size(-1),), sd[name + ".net.1.weight"], sd[name + ".net.1.bias"], 1e-5)
    return linear(sd, name + ".net.3", F.relu(y))


def shifted_softplus(x):
    return F.softplus(x) - LOG2                                          # common.py:66-72


def outer_product_feat(edge_type, dist_feat):
    """outer_product(edge_attr, dist_feat) (common.py:116-123): column = type*20 + g."""
    return (edge_type.unsqueeze(-1) * dist_feat.unsqueeze(1)).reshape(edge_type.size(0), -1)


def attention_weights(q_dst, k, seg, n_seg, n_heads):
    """alpha = scatter_softmax((q*k/sqrt(d)).sum(-1), seg)
    (uni_transformer_edge.py:64, 160, 205)."""
    d = k.size(-1) // n_heads
    kk = k.view(-1, n_heads, d)
    qq = q_dst.view(-1, n_heads, d)
    alpha = ops.scatter_softmax((qq * kk / math.sqrt(d)).sum(-1), seg, dim=0, dim_size=n_seg)
    return alpha


def node_update(sd, name, h, edge_feat, edge_index, e_w, n_heads):
    """NodeUpdateLayer.forward with out_fc=False (uni_transformer_edge.py:42-74)."""
    N = h.size(0)
    src, dst = edge_index
    kv = torch.cat([edge_feat, h[dst], h[src]], -1)
    k = mlp(sd, name + ".hk_func", kv)
    v = mlp(sd, name + ".hv_func", kv)
    if e_w is not None:
        v = v * e_w.view(-1, 1)
    q = mlp(sd, name + ".hq_func", h)
    alpha = attention_weights(q[dst], k, dst, N, n_heads)
    d = v.size(-1) // n_heads
    m = alpha.unsqueeze(-1) * v.view(-1, n_heads, d)
    return ops.scatter_sum(m, dst, dim=0, dim_size=N).view(N, -1)


def pos_update(sd, name, h, rel_x, edge_feat, edge_index, e_w, n_heads):
    """PosUpdateLayer.forward (uni_transformer_edge.py:188-210)."""
    N = h.size(0)
    src, dst = edge_index
    kv = torch.cat([edge_feat, h[dst], h[src]], -1)
    k = mlp(sd, name + ".xk_func", kv)
    v = mlp(sd, name + ".xv_func", kv)                                   # [E, n_heads]
    if e_w is not None:
        v = v * e_w.view(-1, 1)
    v = v.unsqueeze(-1) * rel_x.unsqueeze(1)                             # [E, heads, 3]
    q = mlp(sd, name + ".xq_func", h)
    alpha = attention_weights(q[dst], k, dst, N, n_heads)
    m = alpha.unsqueeze(-1) * v
    return ops.scatter_sum(m, dst, dim=0, dim_size=N).mean(1)


def bond_update(sd, name, h, h_bond, pos, bond_index, n_heads):
    """BondUpdateLayer.forward with include_h_node=True (uni_transformer_edge.py:125-167)."""
    E = h_bond.size(0)
    i, j, idx_i, idx_j, idx_k, idx_kj, idx_ji = ops.bond_triplets(bond_index, h.size(0))
    dist = (pos[i] - pos[j]).pow(2).sum(-1).sqrt()
    pos_i = pos[idx_i]
    pos_ji, pos_ki = pos[idx_j] - pos_i, pos[idx_k] - pos_i
    a = (pos_ji * pos_ki).sum(-1)
    b = torch.cross(pos_ji, pos_ki, dim=-1).norm(dim=-1)
    angle = torch.atan2(b, a)
    r_feat = gaussian_smearing(dist)
    a_feat = angular_encoding(angle)
    kv = torch.cat([h_bond[idx_kj], r_feat[idx_kj], r_feat[idx_ji], a_feat, h[idx_k], h[idx_j]], -1)
    qin = torch.cat([h_bond[idx_ji], h[idx_i]], -1)
    k = mlp(sd, name + ".hk_func", kv)
    v = mlp(sd, name + ".hv_func", kv)
    q = mlp(sd, name + ".hq_func", qin)
    alpha = attention_weights(q, k, idx_ji, E, n_heads)
    d = v.size(-1) // n_heads
    m = alpha.unsqueeze(-1) * v.view(-1, n_heads, d)
    return ops.scatter_sum(m, idx_ji, dim=0, dim_size=E).view(E, -1)


def edge_types(edge_index, mask_ligand):
    """_build_edge_type with decomp_group_idx=None (uni_transformer_edge.py:361-392)."""
    src, dst = edge_index
    n_src, n_dst = mask_ligand[src] == 1, mask_ligand[dst] == 1
    t = torch.zeros(src.numel(), dtype=torch.long, device=src.device)
    t[n_src & n_dst] = 0
    t[n_src & ~n_dst] = 1
    t[~n_src & n_dst] = 2
    t[~n_src & ~n_dst] = 3
    return F.one_hot(t, 4)


def attention_layer(sd, name, h, x, edge_type, edge_index, h_bond, bond_index, mask_ligand_atom, e_w, n_heads,
                    trace=None):
    """AttentionLayerO2TwoUpdateNodeGeneral.forward (uni_transformer_edge.py:259-287)."""
    src, dst = edge_index
    rel_x = x[dst] - x[src]
    dist = torch.norm(rel_x, p=2, dim=-1, keepdim=True)
    dist_feat = outer_product_feat(edge_type.to(x.dtype), gaussian_smearing(dist))
    edge_feat = torch.cat([dist_feat, edge_type.to(x.dtype)], -1)
    a_edge = node_update(sd, name + ".node_layer_with_edge", h, edge_feat, edge_index, e_w, n_heads)
    a_bond = node_update(sd, name + ".node_layer_with_bond", h, h_bond, bond_index, None, n_heads)
    d_bond = bond_update(sd, name + ".bond_layer", h, h_bond, x, bond_index, n_heads)
    new_h_bond = h_bond + d_bond
    new_h = h + linear(sd, name + ".lin_node", a_edge + a_bond)
    dx_edge = pos_update(sd, name + ".pos_layer_with_edge", new_h, rel_x, edge_feat, edge_index, e_w, n_heads)
    b_src, b_dst = bond_index
    rel_bx = x[b_dst] - x[b_src]
    dx_bond = pos_update(sd, name + ".pos_layer_with_bond", new_h, rel_bx, new_h_bond, bond_index, None, n_heads)
    new_x = x + (dx_edge + dx_bond) * mask_ligand_atom[:, None]
    if trace is not None:
        trace.append(dict(a_edge=a_edge, a_bond=a_bond, d_bond=d_bond, h=new_h, h_bond=new_h_bond,
                          dx_edge=dx_edge, dx_bond=dx_bond, x=new_x))
    return new_h, new_h_bond, new_x


def refine_net(sd, cfg, h, x, bond_index, h_bond, mask_ligand, mask_ligand_atom, batch, trace=None):
    """UniTransformerO2TwoUpdateGeneralBond.forward, cutoff_mode='knn' (uni_transformer_edge.py:394-443)."""
    for _ in range(cfg.num_blocks):
        edge_index = ops.knn_graph(x, k=cfg.knn, batch=batch)
        etype = edge_types(edge_index, mask_ligand)
        src, dst = edge_index
        dist = torch.norm(x[dst] - x[src], p=2, dim=-1, keepdim=True)
        e_w = torch.sigmoid(mlp(sd, "refine_net.edge_pred_layer", gaussian_smearing(dist)))
        if trace is not None:
            trace.append(dict(edge_index=edge_index, e_w=e_w))
        for l in range(cfg.num_layers):
            h, h_bond, x = attention_layer(sd, f"refine_net.base_block.{l}", h, x, etype, edge_index, h_bond,
                                           bond_index, mask_ligand_atom, e_w, cfg.n_heads, trace)
    return dict(x=x, h=h, h_bond=h_bond)


def compose_context(h_protein, h_ligand, pos_protein, pos_ligand, batch_protein, batch_ligand, ligand_atom_mask=None):
    """compose_context + find_index_after_sorting (models/common.py:153-194)."""
    batch_ctx = torch.cat([batch_protein, batch_ligand], 0)
    sort_idx = torch.sort(batch_ctx, stable=True).indices
    n_p, n_l = batch_protein.numel(), batch_ligand.numel()
    is_lig = torch.cat([torch.zeros(n_p, dtype=torch.bool), torch.ones(n_l, dtype=torch.bool)]).to(batch_ctx.device)
    mask_ligand = is_lig[sort_idx]
    if ligand_atom_mask is None:
        mask_ligand_atom = mask_ligand
    else:
        mask_ligand_atom = torch.cat([torch.zeros(n_p, dtype=torch.bool, device=batch_ctx.device),
                                      ligand_atom_mask.bool()])[sort_idx]
    h_ctx = torch.cat([h_protein, h_ligand], 0)[sort_idx]
    pos_ctx = torch.cat([pos_protein, pos_ligand], 0)[sort_idx]
    # position of original row r in the sorted context
    inv = torch.empty_like(sort_idx)
    inv[sort_idx] = torch.arange(sort_idx.numel(), device=sort_idx.device)
    return (h_ctx, pos_ctx, batch_ctx[sort_idx], mask_ligand, mask_ligand_atom, inv[:n_p], inv[n_p:])


def forward(sd, cfg, protein_pos, protein_v, batch_protein, init_ligand_pos, init_ligand_v, init_ligand_v_aux,
            batch_ligand, ligand_fc_bond_index, init_ligand_fc_bond_type, ligand_atom_mask=None,
            num_classes=8, trace=None):
    """DecompScorePosNet3D.forward for the shipped config (models/decompdiff.py:213-351).

    ``time_step``, the ``*_group_idx`` and ``prior_*`` arguments of the reference are unused
    on this path (time_emb_dim=0, add_prior_node=False) and are therefore not taken.
    """
    lig_feat = torch.cat([F.one_hot(init_ligand_v, num_classes).float(), init_ligand_v_aux], -1)
    h_protein = linear(sd, "protein_atom_emb", protein_v)
    h_ligand = linear(sd, "ligand_atom_emb", lig_feat)
    if cfg.node_indicator:
        h_protein = torch.cat([h_protein, torch.zeros(len(h_protein), 1).to(h_protein)], -1)
        h_ligand = torch.cat([h_ligand, torch.ones(len(h_ligand), 1).to(h_protein)], -1)
    h_all, pos_all, batch_all, mask_ligand, mask_ligand_atom, _, l_idx = compose_context(
        h_protein, h_ligand, protein_pos, init_ligand_pos, batch_protein, batch_ligand, ligand_atom_mask)
    bond_index = l_idx[ligand_fc_bond_index]
    h_bond = linear(sd, "ligand_bond_emb", F.one_hot(init_ligand_fc_bond_type, cfg.num_bond_classes).float())
    out = refine_net(sd, cfg, h_all, pos_all, bond_index, h_bond, mask_ligand, mask_ligand_atom, batch_all, trace)
    final_h = out["h"][mask_ligand_atom]
    v_logits = linear(sd, "v_inference.2", shifted_softplus(linear(sd, "v_inference.0", final_h)))
    preds = dict(pred_ligand_pos=out["x"][mask_ligand_atom], pred_ligand_v=v_logits)
    if cfg.bond_diffusion:
        assert cfg.bond_net_type == "lin"
        preds["pred_bond"] = linear(sd, "bond_inference.2",
                                    shifted_softplus(linear(sd, "bond_inference.0", out["h_bond"])))
    return preds
