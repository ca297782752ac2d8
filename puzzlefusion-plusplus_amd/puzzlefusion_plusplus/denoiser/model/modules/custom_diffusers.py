"""`PiecewiseScheduler` import path of the reference (denoiser/model/modules/custom_diffusers.py)."""
from pfpp_hip.scheduler import PiecewiseScheduler, piecewise_betas  # noqa: F401


def betas_for_alpha_bar(num_diffusion_timesteps=1000, max_beta=0.999, alpha_transform_type="piece_wise"):
    if alpha_transform_type != "piece_wise":
        raise ValueError("only the piece_wise schedule is used by PuzzleFusion++")
    return piecewise_betas(num_diffusion_timesteps, max_beta)
