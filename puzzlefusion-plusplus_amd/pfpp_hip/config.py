"""Minimal attribute-style config objects with the keys the reference's Hydra yaml tree defines
(config/denoiser/model.yaml, config/denoiser/encoder.yaml, config/ae/vq_vae.yaml,
config/verifier/model.yaml, config/auto_aggl.yaml).  Any object exposing the same attributes
(an OmegaConf DictConfig, a SimpleNamespace) can be passed to the modules instead."""
from __future__ import annotations

from types import SimpleNamespace as NS
from typing import Any, Mapping


def to_namespace(obj: Any) -> Any:
    """nested dict / yaml mapping -> attribute access (what `cfg.model.embed_dim` needs)"""
    if isinstance(obj, Mapping):
        return NS(**{k: to_namespace(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_namespace(v) for v in obj)
    return obj


def load_yaml(path: str) -> Any:
    import yaml

    with open(path) as fh:
        return to_namespace(yaml.safe_load(fh))


def ae_config(**over) -> NS:
    d = dict(n_embeddings=1024, embedding_dim=16, num_point=25, num_dim=64, local_decode_pts=40, beta=0.25)
    d.update(over)
    return NS(**d)


def denoiser_model_config(**over) -> NS:
    d = dict(num_dim=64, num_point=25, out_channels=7, std=1, multires=10, embed_dim=512, num_layers=6, num_heads=8,
             dropout_rate=0.1, DDPM_TRAIN_STEPS=1000, DDPM_BETA_SCHEDULE="linear", timestep_spacing="leading",
             PREDICT_TYPE="epsilon", BETA_START=1e-4, BETA_END=2e-2, num_inference_steps=20,
             multiple_ref_parts=True, encoder_weights_path=None, max_len=20)
    d.update(over)
    return NS(**d)


def denoiser_config(**over) -> NS:
    """cfg with .model and .ae, as passed to Denoiser / DenoiserTransformer / VQVAE"""
    return NS(model=denoiser_model_config(**over.pop("model", {})), ae=ae_config(**over.pop("ae", {})), **over)


def verifier_config(**over) -> NS:
    m = dict(num_bins=6, embed_dim=256, num_layers=6, num_heads=8)
    m.update(over.pop("model", {}))
    return NS(model=NS(**m), max_iters=over.pop("max_iters", 6), threshold=over.pop("threshold", 0.9), **over)


def auto_aggl_config(**over) -> NS:
    """cfg.denoiser.*, cfg.verifier.*, cfg.ae.* (config/auto_aggl.yaml:8-17)"""
    return NS(denoiser=denoiser_config(), verifier=verifier_config(), ae=NS(ae=ae_config()), **over)
