"""ORACLE TOOLING (survey container only — needs /root/reference; never shipped to or run on
the GPU box, never imported by the product path).

Import shims that let the *reference itself* (`/root/reference`, pure Python) be imported
and executed here, where its third-party dependencies are absent (no network):
``torch_scatter``, ``torch_geometric``/``torch_cluster``, ``torch_sparse``, ``rdkit``,
``openbabel``, ``easydict``, ``lmdb``.  The numerical ops are the
restatements in oracle/ops.py (published semantics of those packages); chemistry
toolkits are inert mocks because the sampling hot path never calls them.

Used only by oracle/make_golden.py to generate tests/golden/*.  Nothing from the
reference is copied: it is imported from where it lies.
"""
from __future__ import annotations

import sys
import types
from unittest import mock

import torch

from . import ops

REFERENCE_ROOT = "/root/reference"


class _Storage:
    def __init__(self, row, col, value):
        self._row, self._col, self._value = row, col, value

    def row(self):
        return self._row

    def col(self):
        return self._col

    def value(self):
        return self._value


class SparseTensor:
    """Minimal torch_sparse.SparseTensor: COO kept sorted by (row, col) like the real CSR.

    Supports exactly what uni_transformer_edge.py:108-122 touches: the constructor,
    row-selection by an index tensor, ``set_value(None).sum(dim=1)`` and ``storage``.
    """

    def __init__(self, row, col, value=None, sparse_sizes=None, _sorted=False):
        if not _sorted:
            key = row * (int(sparse_sizes[1]) if sparse_sizes is not None else int(col.max()) + 1) + col
            perm = torch.argsort(key, stable=True)
            row, col = row[perm], col[perm]
            value = value[perm] if value is not None else None
        self.storage = _Storage(row, col, value)
        self.sizes = tuple(int(s) for s in sparse_sizes)

    def __getitem__(self, index):
        row, col, value = self.storage.row(), self.storage.col(), self.storage.value()
        n_rows = self.sizes[0]
        counts = torch.bincount(row, minlength=n_rows)
        ptr = torch.zeros(n_rows + 1, dtype=torch.long)
        ptr[1:] = torch.cumsum(counts, 0)
        new_rows, new_cols, new_vals = [], [], []
        for out_r, r in enumerate(index.tolist()):
            s, e = int(ptr[r]), int(ptr[r + 1])
            new_rows.append(torch.full((e - s,), out_r, dtype=torch.long))
            new_cols.append(col[s:e])
            if value is not None:
                new_vals.append(value[s:e])
        cat = lambda xs: torch.cat(xs) if xs else torch.zeros(0, dtype=torch.long)
        return SparseTensor(cat(new_rows), cat(new_cols), cat(new_vals) if value is not None else None,
                            sparse_sizes=(index.numel(), self.sizes[1]), _sorted=True)

    def set_value(self, value, layout=None):
        assert value is None
        return SparseTensor(self.storage.row(), self.storage.col(), None, sparse_sizes=self.sizes, _sorted=True)

    def sum(self, dim):
        assert dim == 1 and self.storage.value() is None
        return torch.bincount(self.storage.row(), minlength=self.sizes[0]).to(torch.float)


class EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = EasyDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class _Data:
    """Minimal ``torch_geometric.data.Data``: an attribute/key store with PyG's documented collate hooks
    (``__inc__``: ``'batch' in key`` -> max+1, ``'index' in key`` -> num_nodes, else 0; ``__cat_dim__``: -1 for
    ``'index'`` keys, else 0).  The reference's ``ProteinLigandData`` (utils/data.py:343-446) subclasses it and
    overrides ``__inc__`` for its own keys — those overrides are what the harness fixtures pin."""

    def __init__(self, *a, **kw):
        object.__setattr__(self, "_store", {})
        for k, v in kw.items():
            self._store[k] = v

    def __getattr__(self, k):
        try:
            return object.__getattribute__(self, "_store")[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self._store[k] = v

    def __getitem__(self, k):
        return self._store[k]

    def __setitem__(self, k, v):
        self._store[k] = v

    def __contains__(self, k):
        return k in self._store

    def keys(self):
        return list(self._store.keys())

    @property
    def num_nodes(self):
        return None

    def clone(self):
        import copy
        out = self.__class__()
        for k, v in self._store.items():
            out._store[k] = v.clone() if torch.is_tensor(v) else copy.deepcopy(v)
        return out

    def to(self, device):
        for k, v in list(self._store.items()):
            if torch.is_tensor(v):
                self._store[k] = v.to(device)
        return self

    def __inc__(self, key, value, *a, **k):
        if "batch" in key and torch.is_tensor(value):
            return int(value.max()) + 1
        if "index" in key or key == "face":
            return self.num_nodes
        return 0

    def __cat_dim__(self, key, value, *a, **k):
        return -1 if ("index" in key or key == "face") else 0


class _Batch(_Data):
    """``Batch.from_data_list`` (documented PyG collate): tensors are concatenated along ``__cat_dim__`` after adding
    the running sum of the items' ``__inc__``; python numbers become a tensor, everything else a list; each
    ``follow_batch`` key gets a ``<key>_batch`` assignment vector."""

    @classmethod
    def from_data_list(cls, data_list, follow_batch=None, exclude_keys=None):
        out = cls()
        exclude = set(exclude_keys or [])
        for key in data_list[0].keys():
            if key in exclude:
                continue
            vals = [d[key] for d in data_list]
            v0 = vals[0]
            if torch.is_tensor(v0):
                cat_dim = data_list[0].__cat_dim__(key, v0)
                inc_total, pieces, sizes = 0, [], []
                for d, v in zip(data_list, vals):
                    if v.dim() == 0:
                        v = v.unsqueeze(0)
                    if not (isinstance(inc_total, int) and inc_total == 0):
                        v = v + inc_total
                    pieces.append(v)
                    sizes.append(v.size(cat_dim))
                    inc = d.__inc__(key, d[key])
                    if torch.is_tensor(inc):
                        inc = int(inc)
                    inc_total = inc_total + (inc or 0)
                out[key] = torch.cat(pieces, dim=cat_dim)
                if follow_batch and key in follow_batch:
                    out[key + "_batch"] = torch.repeat_interleave(torch.arange(len(vals)), torch.tensor(sizes))
            elif isinstance(v0, (int, float)) and not isinstance(v0, bool):
                out[key] = torch.tensor(vals)
            else:
                out[key] = vals
        return out


class _Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _not_available(*a, **k):
    raise NotImplementedError("not needed on the sampling hot path")


def install():
    """Register the shim modules and put the reference on sys.path."""
    _module("torch_scatter", scatter_sum=ops.scatter_sum, scatter_add=ops.scatter_sum,
            scatter_mean=ops.scatter_mean, scatter_softmax=ops.scatter_softmax,
            scatter_min=ops.scatter_min, scatter_max=ops.scatter_max)
    tg = _module("torch_geometric")
    tg.nn = _module("torch_geometric.nn", knn_graph=ops.knn_graph, radius_graph=_not_available,
                    radius=_not_available, knn=_not_available)
    tg.data = _module("torch_geometric.data", Data=_Data, Batch=_Batch)
    tg.loader = _module("torch_geometric.loader", DataLoader=_Data)
    tg.transforms = _module("torch_geometric.transforms", Compose=_Compose)
    _module("torch_sparse", SparseTensor=SparseTensor)
    _module("easydict", EasyDict=EasyDict)
    for name in ["rdkit", "rdkit.Chem", "rdkit.Chem.rdchem", "rdkit.Chem.AllChem", "rdkit.Chem.Lipinski",
                 "rdkit.Chem.rdMolAlign", "rdkit.Chem.ChemicalFeatures", "rdkit.RDConfig", "rdkit.RDLogger",
                 "rdkit.Geometry", "rdkit.Chem.rdMolTransforms", "rdkit.Chem.Descriptors", "rdkit.Chem.QED",
                 "rdkit.Chem.rdForceFieldHelpers", "rdkit.Chem.Draw", "rdkit.Chem.rdchem.BondType",
                 "rdkit.DataStructs", "rdkit.Chem.Scaffolds", "rdkit.Chem.Scaffolds.MurckoScaffold",
                 "openbabel", "openbabel.openbabel", "lmdb"]:
        if name not in sys.modules:
            sys.modules[name] = mock.MagicMock(name=name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


def load_reference_model(cfg_dict, state=None, protein_dim=29, ligand_dim=10, num_classes=8, prior_atom_types=None,
                         prior_bond_types=None):
    """Construct the reference's DecompScorePosNet3D and load synthetic weights (strict)."""
    install()
    from models.decompdiff import DecompScorePosNet3D           # noqa: the REFERENCE's module
    model = DecompScorePosNet3D(EasyDict(cfg_dict), protein_atom_feature_dim=protein_dim,
                                ligand_atom_feature_dim=ligand_dim, num_classes=num_classes,
                                prior_atom_types=prior_atom_types, prior_bond_types=prior_bond_types)
    if state is not None:
        full = model.state_dict()
        missing = [k for k in state if k not in full]
        assert not missing, missing
        full.update(state)
        model.load_state_dict(full, strict=True)
    return model
