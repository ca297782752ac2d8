"""`puzzlefusion_plusplus.denoiser.model.modules.encoder.VQVAE` is the Hydra `_target_` of
config/denoiser/encoder.yaml:3; in the reference it duplicates vqvae/model/modules/vq_vae.py."""
from puzzlefusion_plusplus.vqvae.model.modules.vq_vae import VQVAE  # noqa: F401
