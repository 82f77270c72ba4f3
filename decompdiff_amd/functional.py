"""Op-level drop-ins over the C ABI (SURVEY.md §8b "op-level boundary"): the third-party functional calls the
reference's layers sit on, with the same argument meaning, for hosts that keep the reference's Python modules and
swap only these ops.  Everything runs on the HIP device; CPU tensors raise (no fallback).

=====================================================  ==========================================================
reference call                                          here
=====================================================  ==========================================================
``torch_geometric.nn.knn_graph(x, k, batch, flow)``     :func:`knn_graph`         (uni_transformer_edge.py:353)
``scatter_softmax(...); scatter_sum(alpha * v, ...)``   :func:`scatter_attention` (uni_transformer_edge.py:63-68,158-164)
the same with ``v.unsqueeze(-1) * rel_x`` and ``mean``  :func:`scatter_attention_pos` (uni_transformer_edge.py:199-211)
=====================================================  ==========================================================
"""
from __future__ import annotations

from typing import Optional

import torch

from . import hip_lib

# ------------------------------------------------------------------------------------------------------------------
# The compiled torch extension (csrc/torch_ext.cpp -> lib/decompdiff_torch_ext.so: TORCH_LIBRARY(decompdiff_hip, ...),
# HIP-key kernels that launch the same C-ABI entry points on torch's current stream).  When it is present the ops below
# go through the dispatcher (torch.ops.decompdiff_hip.*); otherwise -- extension not built, or another library build
# selected with DD_HIP_LIB (the extension is linked against the default one) -- through the ctypes binding of the same
# entry points.  DD_TORCH_EXT=0 forces ctypes.  Either way the arithmetic is the HIP library's: no CPU path.
# ------------------------------------------------------------------------------------------------------------------
import os as _os

_EXT_PATH = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "lib", "decompdiff_torch_ext.so")
_ext_state = {"tried": False, "ops": None}


def torch_ext():
    """torch.ops.decompdiff_hip (compiled extension) or None."""
    if not _ext_state["tried"]:
        _ext_state["tried"] = True
        if _os.environ.get("DD_TORCH_EXT", "1") != "0" and not _os.environ.get("DD_HIP_LIB") and _os.path.exists(_EXT_PATH):
            try:
                torch.ops.load_library(_EXT_PATH)
                if int(torch.ops.decompdiff_hip.abi_version()) == hip_lib.ABI_VERSION:
                    _ext_state["ops"] = torch.ops.decompdiff_hip
            except (OSError, RuntimeError):                  # toolchain skew: the ctypes binding serves the same entry points
                _ext_state["ops"] = None
    return _ext_state["ops"]


def knn_graph(x: torch.Tensor, k: int, batch: Optional[torch.Tensor] = None, loop: bool = False,
              flow: str = "source_to_target") -> torch.Tensor:
    """``edge_index [2,E]`` (row 0 = neighbour / source, row 1 = centre / target), grouped by centre in ascending
    order, neighbours by ascending distance, self loops excluded, candidates restricted to the same ``batch`` id — the
    order torch_cluster returns.  Samples must be contiguous and of equal size (PyG ``Batch`` of one pocket); a sample
    with fewer than ``k`` other atoms contributes all of them."""
    if flow != "source_to_target" or loop:
        raise NotImplementedError("knn_graph: flow='source_to_target', loop=False (the reference's call)")
    hip_lib.require_gpu(x, "x")
    n = x.size(0)
    if batch is None:
        B, N = 1, n
    else:
        B = int(batch.max().item()) + 1
        N = n // B
        if n % B or not torch.equal(batch, torch.arange(B, device=x.device).repeat_interleave(N)):
            raise NotImplementedError("knn_graph: batch must be sorted with equal counts per sample")
    kk = min(int(k), N - 1)
    if kk <= 0:
        return torch.empty(2, 0, dtype=torch.long, device=x.device)
    xc = x.detach().to(torch.float32).contiguous()
    ext = torch_ext()
    if ext is not None:
        nbr = ext.knn(xc.view(B, N, 3), kk)
    else:
        nbr = torch.empty(B, N, kk, dtype=torch.int32, device=x.device)
        hip_lib.check(hip_lib.load().dd_knn(hip_lib.ptr(xc), B, N, kk, hip_lib.ptr(nbr), hip_lib.stream_ptr(x.device)), "dd_knn")
    base = (torch.arange(B, device=x.device) * N).view(B, 1, 1)
    src = (nbr.long() + base).reshape(-1)
    dst = torch.arange(n, device=x.device).repeat_interleave(kk)
    return torch.stack([src, dst], 0)


def _seg_ptr(index: torch.Tensor, dim_size: int) -> torch.Tensor:
    if index.numel() > 1 and bool((index[1:] < index[:-1]).any().item()):
        raise NotImplementedError("scatter_attention: edges must be grouped by destination (sorted index), as knn_graph, "
                                  "the dst-major bond list and the SparseTensor triplets are")
    ptr = torch.zeros(dim_size + 1, dtype=torch.int32, device=index.device)
    ptr[1:] = torch.bincount(index, minlength=dim_size).cumsum(0)
    return ptr


def scatter_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, index: torch.Tensor, dim_size: int,
                      e_w: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``scatter_sum(scatter_softmax((q_e * k / sqrt(8)).sum(-1), index)[..., None] * (v * e_w), index, dim_size)``
    flattened to ``[dim_size,128]`` (16 heads x 8).  ``q`` is ``[dim_size,128]`` (gathered as ``q[index]`` by the
    reference's node / coordinate layers) or per edge ``[E,128]`` (bond layer: rows of a segment are identical)."""
    for name, t in (("q", q), ("k", k), ("v", v)):
        hip_lib.require_gpu(t, name)
    E = k.size(0)
    per_edge = q.size(0) == E and q.size(0) != dim_size
    f = lambda t: t.detach().to(torch.float32).reshape(t.size(0), -1).contiguous()
    out = torch.empty(dim_size, 128, device=k.device)
    ew = None if e_w is None else e_w.detach().to(torch.float32).reshape(-1).contiguous()
    # converted copies are bound to locals that outlive the launch: a temporary freed before the kernel is enqueued
    # would hand its block to the next same-size allocation (k and v would alias)
    qf, kf, vf, seg = f(q), f(k), f(v), _seg_ptr(index, dim_size)
    ext = torch_ext()
    if ext is not None:
        return ext.attn_aggregate_node(qf, bool(per_edge), kf, vf, ew, seg)
    hip_lib.check(hip_lib.load().dd_attn_aggregate_node(hip_lib.ptr(qf), int(per_edge), hip_lib.ptr(kf), hip_lib.ptr(vf),
                                                        hip_lib.ptr(ew), hip_lib.ptr(seg), dim_size,
                                                        hip_lib.ptr(out), hip_lib.stream_ptr(k.device)), "dd_attn_aggregate_node")
    return out


def scatter_attention_pos(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, rel_x: torch.Tensor, index: torch.Tensor,
                          dim_size: int, e_w: Optional[torch.Tensor] = None) -> torch.Tensor:
    """PosUpdateLayer's aggregation: ``v`` is ``[E,16]`` (one scalar per head), ``rel_x`` ``[E,3]``; returns
    ``scatter_sum(alpha[..., None] * (v * e_w)[..., None] * rel_x[:, None], index).mean(1)`` — ``[dim_size,3]``."""
    for name, t in (("q", q), ("k", k), ("v", v), ("rel_x", rel_x)):
        hip_lib.require_gpu(t, name)
    f = lambda t: t.detach().to(torch.float32).reshape(t.size(0), -1).contiguous()
    out = torch.empty(dim_size, 3, device=k.device)
    ew = None if e_w is None else e_w.detach().to(torch.float32).reshape(-1).contiguous()
    qf, kf, vf, rf, seg = f(q), f(k), f(v), f(rel_x), _seg_ptr(index, dim_size)      # (alive past the launch, see above)
    ext = torch_ext()
    if ext is not None:
        return ext.attn_aggregate_pos(qf, kf, vf, ew, rf, seg)
    hip_lib.check(hip_lib.load().dd_attn_aggregate_pos(hip_lib.ptr(qf), hip_lib.ptr(kf), hip_lib.ptr(vf), hip_lib.ptr(ew),
                                                       hip_lib.ptr(rf), hip_lib.ptr(seg), dim_size,
                                                       hip_lib.ptr(out), hip_lib.stream_ptr(k.device)), "dd_attn_aggregate_pos")
    return out


# ------------------------------------------------------------------------------------------------------------------
# Plain torch_scatter drop-ins, same signatures as the wheel the reference imports:
#     from torch_scatter import scatter_softmax, scatter_sum, scatter_mean, scatter_min
# over dim 0 with a 1-D index broadcast along it (every call site of the reference: uni_transformer_edge.py:64,68,160,
# 164,205,209; decompdiff.py:25; guidance_funcs.py:52).  Rows need not be sorted: an unsorted index is stable-sorted
# first (torch_scatter accumulates in an unspecified order; here the order is fixed -> deterministic results).
# ------------------------------------------------------------------------------------------------------------------
_OPS = {"sum": 0, "mean": 1, "min": 2, "max": 3}


def _prep_scatter(src: torch.Tensor, index: torch.Tensor, dim: int, dim_size: Optional[int]):
    hip_lib.require_gpu(src, "src")
    if dim not in (0, -src.dim()):
        raise NotImplementedError("decompdiff_amd scatter ops reduce over dim 0 (every call site of the reference does)")
    if index.dim() != 1:
        if index.shape != src.shape:
            raise ValueError("index must be 1-D or have the shape of src")
        flat = index.reshape(index.size(0), -1)
        if not bool((flat == flat[:, :1]).all().item()):
            raise NotImplementedError("index must be constant along the trailing dims (broadcast of a 1-D index)")
        index = flat[:, 0]
    if index.numel() != src.size(0):
        raise ValueError("index and src disagree along dim 0")
    E = src.size(0)
    n = int(dim_size) if dim_size is not None else (int(index.max().item()) + 1 if E else 0)
    if E and (int(index.min().item()) < 0 or int(index.max().item()) >= n):
        raise IndexError("scatter index out of range")
    perm = None
    if E > 1 and bool((index[1:] < index[:-1]).any().item()):
        index, perm = torch.sort(index, stable=True)
    x = src.detach().to(torch.float32).reshape(E, -1)
    x = (x[perm] if perm is not None else x).contiguous()
    ptr = torch.zeros(n + 1, dtype=torch.int32, device=src.device)
    if E:
        ptr[1:] = torch.bincount(index, minlength=n).cumsum(0)
    return x, ptr, n, perm


class SegmentPlan:
    """The checks and the segment pointer of one (index, dim_size) pair, made ONCE (three device -> host round trips and a
    bincount) for callers that scatter over the same index many times -- a training step uses each of its five index
    vectors 12-36 times (decompdiff_amd/training.py).  Pass it as `index` to scatter_sum / scatter_mean / scatter_softmax."""

    def __init__(self, index: torch.Tensor, dim_size: Optional[int] = None, check: bool = True):
        hip_lib.require_gpu(index, "index")
        if index.dim() != 1:
            raise ValueError("SegmentPlan: 1-D index")
        E = index.numel()
        if not check:
            # No device -> host round trip (usable while a stream is being captured): the caller vouches for 0 <= index < dim_size
            # (e.g. the sources of a kNN graph); the index is sorted unconditionally, the segment pointer comes from a search.
            if dim_size is None:
                raise ValueError("SegmentPlan(check=False) needs dim_size")
            n = int(dim_size)
            self.index, self.perm = torch.sort(index, stable=True)
            self.ptr = torch.searchsorted(self.index, torch.arange(n + 1, device=index.device, dtype=index.dtype)).to(torch.int32)
            self.n, self.E = n, E
            self.inv = torch.empty_like(self.perm)
            self.inv[self.perm] = torch.arange(E, device=index.device)
            return
        n = int(dim_size) if dim_size is not None else (int(index.max().item()) + 1 if E else 0)
        if E and (int(index.min().item()) < 0 or int(index.max().item()) >= n):
            raise IndexError("scatter index out of range")
        self.perm = None
        self.index = index
        if E > 1 and bool((index[1:] < index[:-1]).any().item()):
            self.index, self.perm = torch.sort(index, stable=True)
        self.ptr = torch.zeros(n + 1, dtype=torch.int32, device=index.device)
        if E:
            self.ptr[1:] = torch.bincount(self.index, minlength=n).cumsum(0)
        self.n, self.E = n, E
        self.inv = None
        if self.perm is not None:
            self.inv = torch.empty_like(self.perm)
            self.inv[self.perm] = torch.arange(E, device=index.device)


def _prep(src, index, dim, dim_size):
    """(rows [E,F] fp32 contiguous in segment order, ptr, n, perm) from an index vector or a SegmentPlan."""
    if isinstance(index, SegmentPlan):
        hip_lib.require_gpu(src, "src")
        if src.size(0) != index.E or (dim_size is not None and int(dim_size) != index.n):
            raise ValueError("SegmentPlan does not match src / dim_size")
        x = src.detach().to(torch.float32).reshape(index.E, -1)
        x = (x[index.perm] if index.perm is not None else x).contiguous()
        return x, index.ptr, index.n, index.perm
    return _prep_scatter(src, index, dim, dim_size)


def _segment_reduce(src, index, dim, dim_size, op, out=None):
    x, ptr, n, perm = _prep(src, index, dim, dim_size)
    E, F = x.shape
    ext = torch_ext()
    if ext is not None:
        res, arg = ext.segment_reduce(x, ptr, _OPS[op])
        if op not in ("min", "max"):
            arg = None
    else:
        res = torch.empty(n, F, device=src.device)
        arg = torch.empty(n, F, dtype=torch.int64, device=src.device) if op in ("min", "max") else None
    if ext is None and n and F:
        hip_lib.check(hip_lib.load().dd_segment_reduce(hip_lib.ptr(x) if E else None, hip_lib.ptr(ptr), n, F, _OPS[op], E,
                                                       hip_lib.ptr(res), hip_lib.ptr(arg), hip_lib.stream_ptr(src.device)),
                      "dd_segment_reduce")
    if arg is not None and perm is not None:                # row ids of the caller's (unsorted) order
        valid = arg < E
        arg = torch.where(valid, perm[arg.clamp(max=max(E - 1, 0))], arg)
    shape = (n,) + tuple(src.shape[1:])
    res = res.view(shape).to(src.dtype)
    if out is not None:
        if op in ("sum", "mean"):
            raise NotImplementedError("out= (accumulation into a given tensor) is not used by the reference's call sites")
    return res, (arg.view(shape) if arg is not None else None)


def scatter_sum(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: Optional[torch.Tensor] = None,
                dim_size: Optional[int] = None) -> torch.Tensor:
    """``torch_scatter.scatter_sum`` (= ``scatter_add``) over dim 0."""
    return _segment_reduce(src, index, dim if src.dim() > 1 or dim != -1 else 0, dim_size, "sum", out)[0]


def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: Optional[torch.Tensor] = None,
                 dim_size: Optional[int] = None) -> torch.Tensor:
    """``torch_scatter.scatter_mean`` over dim 0 (sum / max(count, 1))."""
    return _segment_reduce(src, index, dim if src.dim() > 1 or dim != -1 else 0, dim_size, "mean", out)[0]


def scatter_min(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: Optional[torch.Tensor] = None,
                dim_size: Optional[int] = None):
    """``torch_scatter.scatter_min`` over dim 0 -> ``(values, argmin)``; empty destinations hold 0 / ``src.size(0)``."""
    return _segment_reduce(src, index, dim if src.dim() > 1 or dim != -1 else 0, dim_size, "min", out)


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim: int = -1, out: Optional[torch.Tensor] = None,
                dim_size: Optional[int] = None):
    """``torch_scatter.scatter_max`` over dim 0 -> ``(values, argmax)``."""
    return _segment_reduce(src, index, dim if src.dim() > 1 or dim != -1 else 0, dim_size, "max", out)


def scatter_softmax(src: torch.Tensor, index: torch.Tensor, dim: int = -1, dim_size: Optional[int] = None) -> torch.Tensor:
    """``torch_scatter.composite.scatter_softmax`` over dim 0: per destination and trailing element, softmax over the rows
    scattered to it (max-shifted, no eps: torch_scatter >= 2.1)."""
    x, ptr, n, perm = _prep(src, index, dim if src.dim() > 1 or dim != -1 else 0, dim_size)
    E, F = x.shape
    ext = torch_ext()
    res = ext.segment_softmax(x, ptr) if ext is not None else torch.empty_like(x)
    if ext is None and E and F:
        hip_lib.check(hip_lib.load().dd_segment_softmax(hip_lib.ptr(x), hip_lib.ptr(ptr), n, F, hip_lib.ptr(res),
                                                        hip_lib.stream_ptr(src.device)), "dd_segment_softmax")
    if perm is not None:
        if isinstance(index, SegmentPlan):
            inv = index.inv
        else:
            inv = torch.empty_like(perm)
            inv[perm] = torch.arange(E, device=perm.device)
        res = res[inv]
    return res.view(src.shape).to(src.dtype)
