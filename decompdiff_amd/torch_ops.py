"""Torch-level registration of the op-level boundary (BASELINE.json north_star: "exposed as torch
extensions"): every drop-in of :mod:`decompdiff_amd.functional` becomes a dispatcher op ``torch.ops.decompdiff_amd.*``
(``torch.library.custom_op``, device type "cuda" = HIP on ROCm), with shape-propagating fake implementations so the ops
trace under ``torch.compile`` / ``make_fx``.  The kernels stay behind the C ABI (ctypes, no torch headers in the library);
there is no CPU implementation — CPU tensors raise.

    import decompdiff_amd.torch_ops                         # registers the ops
    edge_index = torch.ops.decompdiff_amd.knn_graph(x, 32, batch)
    alpha = torch.ops.decompdiff_amd.scatter_softmax(score, dst, 0, n)

:func:`patch_reference_imports` installs module objects named ``torch_scatter`` and ``torch_cluster`` exposing these ops
under the names the reference imports (``from torch_scatter import scatter_softmax, scatter_sum``:
models/encoders/uni_transformer_edge.py:9; ``scatter_mean``: models/decompdiff.py:9; ``scatter_min``:
utils/guidance_funcs.py:9), for hosts that keep the reference's Python layers.
"""
from __future__ import annotations

import sys
import types
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import functional as F

_NS = "decompdiff_amd"


@torch.library.custom_op(f"{_NS}::knn_graph", mutates_args=(), device_types="cuda")
def knn_graph(x: Tensor, k: int, batch: Optional[Tensor] = None, loop: bool = False, flow: str = "source_to_target") -> Tensor:
    return F.knn_graph(x, k, batch, loop, flow)


@knn_graph.register_fake
def _(x, k, batch=None, loop=False, flow="source_to_target"):
    ctx = torch.library.get_ctx()
    return x.new_empty((2, ctx.new_dynamic_size()), dtype=torch.long)


@torch.library.custom_op(f"{_NS}::scatter_attention", mutates_args=(), device_types="cuda")
def scatter_attention(q: Tensor, k: Tensor, v: Tensor, index: Tensor, dim_size: int, e_w: Optional[Tensor] = None) -> Tensor:
    return F.scatter_attention(q, k, v, index, dim_size, e_w)


@scatter_attention.register_fake
def _(q, k, v, index, dim_size, e_w=None):
    return k.new_empty((dim_size, 128), dtype=torch.float32)


@torch.library.custom_op(f"{_NS}::scatter_attention_pos", mutates_args=(), device_types="cuda")
def scatter_attention_pos(q: Tensor, k: Tensor, v: Tensor, rel_x: Tensor, index: Tensor, dim_size: int,
                          e_w: Optional[Tensor] = None) -> Tensor:
    return F.scatter_attention_pos(q, k, v, rel_x, index, dim_size, e_w)


@scatter_attention_pos.register_fake
def _(q, k, v, rel_x, index, dim_size, e_w=None):
    return k.new_empty((dim_size, 3), dtype=torch.float32)


def _n_out(index, dim_size):
    return dim_size if dim_size is not None else torch.library.get_ctx().new_dynamic_size()


@torch.library.custom_op(f"{_NS}::scatter_sum", mutates_args=(), device_types="cuda")
def scatter_sum(src: Tensor, index: Tensor, dim: int = -1, dim_size: Optional[int] = None) -> Tensor:
    return F.scatter_sum(src, index, dim, None, dim_size)


@scatter_sum.register_fake
def _(src, index, dim=-1, dim_size=None):
    return src.new_empty((_n_out(index, dim_size),) + tuple(src.shape[1:]))


@torch.library.custom_op(f"{_NS}::scatter_mean", mutates_args=(), device_types="cuda")
def scatter_mean(src: Tensor, index: Tensor, dim: int = -1, dim_size: Optional[int] = None) -> Tensor:
    return F.scatter_mean(src, index, dim, None, dim_size)


@scatter_mean.register_fake
def _(src, index, dim=-1, dim_size=None):
    return src.new_empty((_n_out(index, dim_size),) + tuple(src.shape[1:]))


@torch.library.custom_op(f"{_NS}::scatter_min", mutates_args=(), device_types="cuda")
def scatter_min(src: Tensor, index: Tensor, dim: int = -1, dim_size: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    return F.scatter_min(src, index, dim, None, dim_size)


@scatter_min.register_fake
def _(src, index, dim=-1, dim_size=None):
    shape = (_n_out(index, dim_size),) + tuple(src.shape[1:])
    return src.new_empty(shape), src.new_empty(shape, dtype=torch.long)


@torch.library.custom_op(f"{_NS}::scatter_softmax", mutates_args=(), device_types="cuda")
def scatter_softmax(src: Tensor, index: Tensor, dim: int = -1, dim_size: Optional[int] = None) -> Tensor:
    return F.scatter_softmax(src, index, dim, dim_size)


@scatter_softmax.register_fake
def _(src, index, dim=-1, dim_size=None):
    return torch.empty_like(src)


# ------------------------------------------------------------------------------------------------------------------
# Autograd formulas (the reference differentiates through these ops: drift guidance calls torch.autograd.grad through
# scatter_min, utils/guidance_funcs.py:52-60; training goes through scatter_softmax / scatter_sum / scatter_mean).
# Backward passes reuse the registered ops (so they run on the HIP kernels too); the index is never differentiated.
# ------------------------------------------------------------------------------------------------------------------
def _index_1d(index):
    return index if index.dim() == 1 else index.reshape(index.size(0), -1)[:, 0]


def _setup_sum(ctx, inputs, output):
    src, index, dim, dim_size = inputs
    ctx.save_for_backward(_index_1d(index))


def _bw_sum(ctx, g):
    (index,) = ctx.saved_tensors
    return g.index_select(0, index), None, None, None


scatter_sum.register_autograd(_bw_sum, setup_context=_setup_sum)


def _setup_mean(ctx, inputs, output):
    src, index, dim, dim_size = inputs
    idx = _index_1d(index)
    ctx.save_for_backward(idx, torch.bincount(idx, minlength=output.size(0)).clamp(min=1))


def _bw_mean(ctx, g):
    index, cnt = ctx.saved_tensors
    scale = (1.0 / cnt.to(g.dtype)).view((-1,) + (1,) * (g.dim() - 1))
    return (g * scale).index_select(0, index), None, None, None


scatter_mean.register_autograd(_bw_mean, setup_context=_setup_mean)


def _setup_softmax(ctx, inputs, output):
    src, index, dim, dim_size = inputs
    idx = _index_1d(index)
    ctx.save_for_backward(output, idx)
    ctx.n_seg = int(dim_size) if dim_size is not None else (int(idx.max().item()) + 1 if idx.numel() else 0)


def _bw_softmax(ctx, g):
    y, index = ctx.saved_tensors
    yg = y * g
    tot = torch.ops.decompdiff_amd.scatter_sum(yg.contiguous(), index, 0, ctx.n_seg)
    return yg - y * tot.index_select(0, index), None, None, None


scatter_softmax.register_autograd(_bw_softmax, setup_context=_setup_softmax)


def _setup_min(ctx, inputs, output):
    src, index, dim, dim_size = inputs
    ctx.save_for_backward(output[1])
    ctx.src_shape = tuple(src.shape)


def _bw_min(ctx, g_val, g_arg):
    (arg,) = ctx.saved_tensors
    E = ctx.src_shape[0]
    a = arg.reshape(arg.size(0), -1)
    g = g_val.reshape(arg.size(0), -1)
    valid = a < E                                                        # empty destinations point at row E
    grad = torch.zeros((E + 1, a.size(1)), dtype=g.dtype, device=g.device)
    grad.scatter_add_(0, torch.where(valid, a, torch.full_like(a, E)), g * valid.to(g.dtype))
    return grad[:E].reshape(ctx.src_shape), None, None, None


scatter_min.register_autograd(_bw_min, setup_context=_setup_min)


def patch_reference_imports(force: bool = False) -> None:
    """Make ``import torch_scatter`` / ``from torch_geometric.nn import knn_graph``-style imports of a host that keeps the
    reference's Python resolve to these ops.  Existing real packages are left alone unless ``force``."""
    ops = torch.ops.decompdiff_amd

    def _with_out(op):
        def f(src, index, dim=-1, out=None, dim_size=None):
            if out is not None:
                raise NotImplementedError("out= is not used by the reference's call sites")
            return op(src, index, dim, dim_size)
        return f
    if force or "torch_scatter" not in sys.modules:
        m = types.ModuleType("torch_scatter")
        m.scatter_sum = m.scatter_add = _with_out(ops.scatter_sum)
        m.scatter_mean = _with_out(ops.scatter_mean)
        m.scatter_min = _with_out(ops.scatter_min)
        m.scatter_softmax = lambda src, index, dim=-1, dim_size=None: ops.scatter_softmax(src, index, dim, dim_size)
        m.__doc__ = "decompdiff_amd drop-ins (HIP, gfx950) for the torch_scatter functions the reference imports"
        sys.modules["torch_scatter"] = m
    if force or "torch_cluster" not in sys.modules:
        c = types.ModuleType("torch_cluster")
        c.knn_graph = lambda x, k, batch=None, loop=False, flow="source_to_target", **kw: ops.knn_graph(x, k, batch, loop, flow)
        sys.modules["torch_cluster"] = c
