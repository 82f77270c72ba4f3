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
