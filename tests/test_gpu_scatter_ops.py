"""Op-level message-passing kernels (include/decompdiff_hip.h: dd_attn_aggregate_{node,triplet,pos}) against the
reference formulas written with the oracle's scatter ops (uni_transformer_edge.py:63-68, :158-164, :205-211)."""
import ctypes

import numpy as np
import pytest
import torch

from decompdiff_amd import hip_lib
from oracle import ops

pytestmark = pytest.mark.gpu


def _segments(n_seg, sizes, seed):
    g = torch.Generator().manual_seed(seed)
    cnt = torch.tensor(sizes)[torch.randint(0, len(sizes), (n_seg,), generator=g)]
    ptr = torch.zeros(n_seg + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(cnt, 0)
    dst = torch.repeat_interleave(torch.arange(n_seg), cnt)
    return ptr, dst, g


def _ref_alpha(q_e, k, dst, n_seg):
    s = (q_e.view(-1, 16, 8) * k.view(-1, 16, 8) / np.sqrt(8)).sum(-1)
    return ops.scatter_softmax(s.double(), dst, 0, dim_size=n_seg)


@pytest.mark.parametrize("n_seg,sizes,use_ew,per_edge", [(257, [32], True, False), (300, [0, 1, 5, 29, 33, 64], False, False),
                                                         (123, [28], False, True), (64, [3, 70], True, True)])
def test_attn_aggregate_node(n_seg, sizes, use_ew, per_edge):
    lib = hip_lib.load()
    dev = torch.device("cuda:0")
    ptr, dst, g = _segments(n_seg, sizes, 1)
    E = int(ptr[-1])
    q = torch.randn(n_seg, 128, generator=g) * 2
    k, v = torch.randn(E, 128, generator=g) * 2, torch.randn(E, 128, generator=g)
    ew = torch.rand(E, generator=g) if use_ew else None
    alpha = _ref_alpha(q[dst], k, dst, n_seg)
    m = alpha.unsqueeze(-1) * (v * (ew.view(-1, 1) if use_ew else 1.0)).view(-1, 16, 8).double()
    want = ops.scatter_sum(m, dst, 0, dim_size=n_seg).view(n_seg, 128)
    qd = (q[dst] if per_edge else q).contiguous().to(dev)
    kd, vd, pd = k.to(dev), v.to(dev), ptr.to(dev)
    ed = ew.to(dev) if use_ew else None
    out = torch.full((n_seg, 128), float("nan"), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if per_edge and not use_ew:
        rc = lib.dd_attn_aggregate_triplet(hip_lib.ptr(qd), hip_lib.ptr(kd), hip_lib.ptr(vd), hip_lib.ptr(pd), n_seg, hip_lib.ptr(out), st)
    else:
        rc = lib.dd_attn_aggregate_node(hip_lib.ptr(qd), int(per_edge), hip_lib.ptr(kd), hip_lib.ptr(vd),
                                        hip_lib.ptr(ed) if use_ew else None, hip_lib.ptr(pd), n_seg, hip_lib.ptr(out), st)
    assert rc == 0
    torch.cuda.synchronize()
    err = float((out.cpu().double() - want).abs().max())
    print(f"aggregate_node n_seg={n_seg} E={E}: max err {err:.3g}")
    assert err < 2e-5


def test_attn_aggregate_pos():
    lib = hip_lib.load()
    dev = torch.device("cuda:0")
    n_seg = 240
    ptr, dst, g = _segments(n_seg, [32, 29, 0, 7], 2)
    E = int(ptr[-1])
    q, k = torch.randn(n_seg, 128, generator=g) * 2, torch.randn(E, 128, generator=g) * 2
    v16, ew, rel = torch.randn(E, 16, generator=g), torch.rand(E, generator=g), torch.randn(E, 3, generator=g) * 3
    alpha = _ref_alpha(q[dst], k, dst, n_seg)
    vv = (v16 * ew.view(-1, 1)).unsqueeze(-1).double() * rel.unsqueeze(1).double()
    want = ops.scatter_sum(alpha.unsqueeze(-1) * vv, dst, 0, dim_size=n_seg).mean(1)
    out = torch.full((n_seg, 3), float("nan"), device=dev)
    t = [x.to(dev) for x in (q, k, v16, ew, rel, ptr)]
    rc = lib.dd_attn_aggregate_pos(*[hip_lib.ptr(x) for x in t], n_seg, hip_lib.ptr(out),
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    err = float((out.cpu().double() - want).abs().max())
    print(f"aggregate_pos n_seg={n_seg} E={E}: max err {err:.3g}")
    assert err < 2e-5


def test_attn_aggregate_rejects_null():
    lib = hip_lib.load()
    assert lib.dd_attn_aggregate_node(None, 0, None, None, None, None, 4, None, None) != 0


def test_functional_wrappers_match_the_reference_layer_math():
    """decompdiff_amd.functional: knn_graph in torch_cluster's order and the two aggregation wrappers fed exactly like
    NodeUpdateLayer / PosUpdateLayer feed torch_scatter (q gathered by dst, v pre-multiplied or not)."""
    from decompdiff_amd import functional as F2
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, N, K = 3, 70, 32
    x = torch.randn(B * N, 3, generator=g) * 4
    batch = torch.arange(B).repeat_interleave(N)
    want = ops.knn_graph(x, K, batch)
    got = F2.knn_graph(x.to(dev), K, batch.to(dev)).cpu()
    assert got.shape == want.shape and torch.equal(got, want)
    src, dst = got
    E = src.numel()
    q, k, v = torch.randn(B * N, 128, generator=g), torch.randn(E, 128, generator=g), torch.randn(E, 128, generator=g)
    ew = torch.rand(E, 1, generator=g)
    alpha = _ref_alpha(q[dst], k, dst, B * N)
    want_n = ops.scatter_sum(alpha.unsqueeze(-1) * (v * ew).view(-1, 16, 8).double(), dst, 0, dim_size=B * N).view(-1, 128)
    got_n = F2.scatter_attention(q.to(dev), k.to(dev), v.to(dev), dst.to(dev), B * N, e_w=ew.to(dev)).cpu().double()
    assert float((got_n - want_n).abs().max()) < 2e-5
    v16 = torch.randn(E, 16, generator=g)
    rel = x[dst] - x[src]
    want_p = ops.scatter_sum(alpha.unsqueeze(-1) * ((v16 * ew).unsqueeze(-1) * rel.unsqueeze(1)).double(), dst, 0, dim_size=B * N).mean(1)
    got_p = F2.scatter_attention_pos(q.to(dev), k.to(dev), v16.to(dev), rel.to(dev), dst.to(dev), B * N, e_w=ew.to(dev)).cpu().double()
    assert float((got_p - want_p).abs().max()) < 2e-5
    with pytest.raises(hip_lib.HipLibraryError):
        F2.knn_graph(x, K, batch)                    # CPU tensors: no fallback


# ---------------------------------------------------------------------------------------------------------------------
# plain torch_scatter drop-ins (decompdiff_amd.functional / torch.ops.decompdiff_amd.*) against the oracle's restatement of
# the published semantics, on the shapes of the reference's call sites
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_seg,sizes,trail", [(330, [32], (16,)), (240, [0, 1, 29], (16, 8)), (7, [300, 301], (3,)),
                                               (50, [5, 9], (30,)), (9, [4], (130,)), (40, [3, 64, 65], ())])
def test_scatter_ops_match_torch_scatter_semantics(n_seg, sizes, trail):
    from decompdiff_amd import functional as Fn
    import decompdiff_amd.torch_ops  # noqa: F401  (registers torch.ops.decompdiff_amd.*)
    dev = torch.device("cuda:0")
    ptr, dst, g = _segments(n_seg, sizes, 3)
    E = int(ptr[-1])
    src = torch.randn((E,) + trail, generator=g) * 3
    srcd, dstd = src.to(dev), dst.to(dev)
    want_sum = ops.scatter_sum(src.double(), dst, 0, dim_size=n_seg)
    want_mean = ops.scatter_mean(src.double(), dst, 0, dim_size=n_seg)
    want_sm = ops.scatter_softmax(src.double(), dst, 0, dim_size=n_seg)
    got_sum = torch.ops.decompdiff_amd.scatter_sum(srcd, dstd, 0, n_seg)
    got_mean = Fn.scatter_mean(srcd, dstd, dim=0, dim_size=n_seg)
    got_sm = torch.ops.decompdiff_amd.scatter_softmax(srcd, dstd, 0, n_seg)
    e1 = float((got_sum.cpu().double() - want_sum).abs().max())
    e2 = float((got_mean.cpu().double() - want_mean).abs().max())
    e3 = float((got_sm.cpu().double() - want_sm).abs().max()) if E else 0.0
    print(f"scatter n_seg={n_seg} E={E} trail={trail}: sum {e1:.3g} mean {e2:.3g} softmax {e3:.3g}")
    assert got_sum.shape == want_sum.shape and got_sm.shape == src.shape
    assert e1 < 1e-4 and e2 < 1e-5 and e3 < 1e-6
    # min: values exact (no arithmetic), arg = a row holding the minimum; empty destinations 0 / E (torch_scatter's fill)
    vmin, amin = torch.ops.decompdiff_amd.scatter_min(srcd, dstd, 0, n_seg)
    vmin, amin = vmin.cpu(), amin.cpu()
    big = torch.full((n_seg,) + trail, float("inf")).scatter_reduce(0, dst.view([-1] + [1] * len(trail)).expand_as(src), src,
                                                                     "amin", include_self=True)
    empty = torch.isinf(big)
    assert torch.equal(vmin[~empty], big[~empty]) and bool((vmin[empty] == 0).all()) and bool((amin[empty] == E).all())
    flat_src, flat_arg, flat_v = src.reshape(E, -1), amin.reshape(n_seg, -1), vmin.reshape(n_seg, -1)
    cols = torch.arange(flat_arg.shape[1]).expand_as(flat_arg)
    ok = ~empty.reshape(n_seg, -1)
    assert torch.equal(flat_src[flat_arg[ok], cols[ok]], flat_v[ok])
    assert torch.equal(dst[flat_arg[ok]], torch.arange(n_seg).view(-1, 1).expand_as(flat_arg)[ok])


def test_scatter_ops_unsorted_index_and_reference_call_shapes():
    """An unsorted index (torch_scatter accepts any order) and the exact calls of the reference: center_pos's
    scatter_mean (decompdiff.py:25), the drift's scatter_min over arm ids (guidance_funcs.py:52)."""
    from decompdiff_amd import functional as Fn
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    idx = torch.randint(0, 11, (500,), generator=g)
    src = torch.randn(500, 16, generator=g)
    got = Fn.scatter_softmax(src.to(dev), idx.to(dev), dim=0, dim_size=11).cpu()
    want = ops.scatter_softmax(src.double(), idx, 0, dim_size=11)
    assert float((got.double() - want).abs().max()) < 1e-6
    v, a = Fn.scatter_min(src.to(dev), idx.to(dev), dim=0)
    assert torch.equal(src[a.cpu(), torch.arange(16).expand(11, 16)], v.cpu())
    # center_pos: offset = scatter_mean(protein_pos, batch_protein, dim=0)
    pos = torch.randn(900, 3, generator=g) * 20
    batch = torch.arange(3).repeat_interleave(300)
    off = Fn.scatter_mean(pos.to(dev), batch.to(dev), dim=0).cpu()
    assert float((off.double() - ops.scatter_mean(pos.double(), batch, 0)).abs().max()) < 1e-5
    # armsca: min_dist_all, _ = scatter_min(pairwise_dist [n_arm_atoms, n_sca], arm_index, dim=0)
    d = torch.rand(16, 14, generator=g) * 5
    arm = torch.tensor([0] * 8 + [1] * 8)
    m, _ = Fn.scatter_min(d.to(dev), arm.to(dev), dim=0)
    assert torch.equal(m.cpu(), torch.stack([d[:8].min(0).values, d[8:].min(0).values]))


def test_scatter_attention_accepts_non_contiguous_and_half_inputs():
    """The wrappers convert to fp32-contiguous copies; those copies must stay alive until the launch is enqueued (a freed
    temporary would hand its block to the next conversion and k / v would alias)."""
    from decompdiff_amd import functional as Fn
    dev = torch.device("cuda:0")
    n_seg, per = 200, 32
    g = torch.Generator().manual_seed(2)
    q, k, v = (torch.randn(n, 256, generator=g) for n in (n_seg, n_seg * per, n_seg * per))
    dst = torch.arange(n_seg).repeat_interleave(per)
    qd, kd, vd = (t.to(dev)[:, ::2] for t in (q, k, v))                   # non-contiguous views
    base = Fn.scatter_attention(qd.contiguous(), kd.contiguous(), vd.contiguous(), dst.to(dev), n_seg)
    got = Fn.scatter_attention(qd, kd, vd, dst.to(dev), n_seg)
    assert torch.equal(got, base)
    got_h = Fn.scatter_attention(qd.half(), kd.half(), vd.half(), dst.to(dev), n_seg)
    ref_h = Fn.scatter_attention(qd.half().float(), kd.half().float(), vd.half().float(), dst.to(dev), n_seg)
    assert torch.equal(got_h, ref_h)


def test_torch_ops_registered_with_reference_signatures():
    import decompdiff_amd.torch_ops as T
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(2 * 40, 3, generator=g) * 10).to(dev)
    batch = torch.arange(2, device=dev).repeat_interleave(40)
    ei = torch.ops.decompdiff_amd.knn_graph(x, 8, batch)
    assert ei.shape == (2, 80 * 8) and ei.dtype == torch.long
    assert torch.equal(ei, __import__("decompdiff_amd").functional.knn_graph(x, 8, batch))
    T.patch_reference_imports(force=True)
    import torch_scatter                                           # the module the reference imports
    s = torch.randn(80 * 8, 16, device=dev)
    a = torch_scatter.scatter_softmax(s, ei[1], dim=0, dim_size=80)
    tot = torch_scatter.scatter_sum(a, ei[1], dim=0, dim_size=80)
    assert float((tot - 1).abs().max()) < 1e-5
    with pytest.raises(Exception):                                  # CPU tensors: no fallback
        torch.ops.decompdiff_amd.scatter_sum(torch.randn(4, 2), torch.tensor([0, 0, 1, 1]), 0, 2)


def test_torch_ops_backward_matches_autograd_of_the_oracle_ops():
    """register_autograd formulas of torch.ops.decompdiff_amd.scatter_{sum,mean,softmax,min} (the reference differentiates
    through them: guidance_funcs.py:52-60 takes torch.autograd.grad through scatter_min; training goes through
    scatter_softmax / scatter_sum) against torch autograd through the oracle's CPU restatement, fp64."""
    import decompdiff_amd.torch_ops  # noqa: F401
    dev = torch.device("cuda:0")
    for n_seg, sizes, trail in ((40, [0, 3, 17], (16,)), (25, [5, 9], (3,)), (12, [4, 33], ())):
        ptr, dst, g = _segments(n_seg, sizes, 11)
        E = int(ptr[-1])
        src = torch.randn((E,) + trail, generator=g) * 2
        w = torch.randn((n_seg,) + trail, generator=g)
        w_e = torch.randn((E,) + trail, generator=g)
        for name in ("sum", "mean", "softmax", "min"):
            a = src.clone().double().requires_grad_(True)
            b = src.clone().to(dev).requires_grad_(True)
            if name == "softmax":
                (ops.scatter_softmax(a, dst, 0, dim_size=n_seg) * w_e.double()).sum().backward()
                (torch.ops.decompdiff_amd.scatter_softmax(b, dst.to(dev), 0, n_seg) * w_e.to(dev)).sum().backward()
            elif name == "min":
                big = torch.full((n_seg,) + trail, float("inf"), dtype=torch.float64).scatter_reduce(
                    0, dst.view([-1] + [1] * len(trail)).expand_as(a), a, "amin", include_self=True)
                big = torch.where(torch.isinf(big), torch.zeros_like(big), big)
                (big * w.double()).sum().backward()
                (torch.ops.decompdiff_amd.scatter_min(b, dst.to(dev), 0, n_seg)[0] * w.to(dev)).sum().backward()
            else:
                f_ref = ops.scatter_sum if name == "sum" else ops.scatter_mean
                f_op = torch.ops.decompdiff_amd.scatter_sum if name == "sum" else torch.ops.decompdiff_amd.scatter_mean
                (f_ref(a, dst, 0, dim_size=n_seg) * w.double()).sum().backward()
                (f_op(b, dst.to(dev), 0, n_seg) * w.to(dev)).sum().backward()
            err = float((b.grad.cpu().double() - a.grad).abs().max()) if E else 0.0
            assert b.grad.shape == src.shape and err < 2e-5, (name, n_seg, trail, err)


def test_compiled_extension_ops_equal_the_ctypes_binding():
    """The ops of the compiled torch extension (torch.ops.decompdiff_hip.*, csrc/torch_ext.cpp) launch the same C-ABI entry
    points as the ctypes binding: functional.* through either route is bit-identical."""
    from decompdiff_amd import functional as Fn
    ops = Fn.torch_ext()
    assert ops is not None, "compiled extension missing on the GPU box"
    dev = torch.device("cuda:0")
    ptr, dst, g = _segments(300, [0, 5, 32], 9)
    E = int(ptr[-1])
    src = (torch.randn(E, 16, generator=g) * 2).to(dev)
    q, k, v = (torch.randn(n, 128, generator=g).to(dev) for n in (300, E, E))
    v16, rel, ew = torch.randn(E, 16, generator=g).to(dev), torch.randn(E, 3, generator=g).to(dev), torch.rand(E, generator=g).to(dev)
    x = torch.randn(3 * 70, 3, generator=g).to(dev)
    batch = torch.arange(3, device=dev).repeat_interleave(70)
    dstd = dst.to(dev)

    def run():
        return [Fn.knn_graph(x, 16, batch), Fn.scatter_sum(src, dstd, 0, dim_size=300), Fn.scatter_mean(src, dstd, 0, dim_size=300),
                *Fn.scatter_min(src, dstd, 0, dim_size=300), Fn.scatter_softmax(src, dstd, 0, dim_size=300),
                Fn.scatter_attention(q, k, v, dstd, 300, ew), Fn.scatter_attention_pos(q, k, v16, rel, dstd, 300, ew)]
    via_ext = run()
    saved = Fn._ext_state["ops"]
    try:
        Fn._ext_state["ops"] = None                          # force the ctypes route
        via_ctypes = run()
    finally:
        Fn._ext_state["ops"] = saved
    for a, b in zip(via_ext, via_ctypes):
        assert a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)
