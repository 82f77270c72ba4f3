"""Attribute-style model config for the sampling hot path.

Defaults are the values of the reference's shipped training config
(/root/reference/configs/training.yml:16-57, `model:` section).  The reference
reads them as attributes of an EasyDict (utils/misc.py:26-28) and uses
``getattr(config, name, default)`` for optional keys (models/decompdiff.py:85-90),
so any object with these attributes (EasyDict, SimpleNamespace, this class) can be
handed to :class:`decompdiff_amd.DecompScorePosNet3D`.
"""
from __future__ import annotations

SHIPPED_MODEL_CONFIG = dict(
    model_mean_type="C0",
    beta_schedule="sigmoid",
    beta_start=1.0e-7,
    beta_end=2.0e-3,
    v_beta_schedule="cosine",
    v_beta_s=0.01,
    num_diffusion_timesteps=1000,
    v_mode="categorical",
    v_net_type="mlp",
    loss_pos_type="mse",
    sample_time_method="symmetric",
    bond_diffusion=True,
    bond_net_type="lin",
    num_bond_classes=5,
    prior_types=False,
    h_node_in_bond_net=True,
    add_prior_node=False,
    time_emb_dim=0,
    time_emb_mode="simple",
    center_pos_mode="protein",
    node_indicator=True,
    model_type="uni_o2_bond",
    num_blocks=1,
    num_layers=6,
    hidden_dim=128,
    n_heads=16,
    edge_feat_dim=4,
    num_r_gaussian=20,
    knn=32,
    act_fn="relu",
    norm=True,
    cutoff_mode="knn",
    r_max=10.0,
    x2h_out_fc=False,
    sync_twoup=False,
)

# Feature widths used by scripts/sample_diffusion_decomp.py:537-541.
PROTEIN_ATOM_FEATURE_DIM = 27 + 2
LIGAND_ATOM_FEATURE_DIM = 8 + 2
NUM_ATOM_CLASSES = 8


class ModelConfig:
    """Minimal attribute bag (stand-in for easydict.EasyDict, which is not installed)."""

    def __init__(self, **overrides):
        values = dict(SHIPPED_MODEL_CONFIG)
        values.update(overrides)
        for k, v in values.items():
            setattr(self, k, v)

    def to_dict(self):
        return dict(self.__dict__)

    def __repr__(self):
        return f"ModelConfig({self.__dict__})"


def shipped_config(**overrides) -> ModelConfig:
    return ModelConfig(**overrides)
