"""Noise schedule buffers (reference ldm/models/diffusion/ddpm.py:11-54): betas in fp64,
cumulative products, twelve fp32 buffers with the reference's names."""
import numpy as np
import torch
import torch.nn as nn

from ldm.modules.diffusionmodules.util import make_beta_schedule


class DDPM(nn.Module):
    def __init__(self, beta_schedule="linear", timesteps=1000, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
        super().__init__()
        self.v_posterior = 0
        self.register_schedule(beta_schedule, timesteps, linear_start, linear_end, cosine_s)

    def register_schedule(self, beta_schedule="linear", timesteps=1000, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
        betas = make_beta_schedule(beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        post_var = (1 - self.v_posterior) * betas * (1.0 - ac_prev) / (1.0 - ac) + self.v_posterior * betas
        buffers = {
            "betas": betas,
            "alphas_cumprod": ac,
            "alphas_cumprod_prev": ac_prev,
            "sqrt_alphas_cumprod": np.sqrt(ac),
            "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
            "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
            "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
            "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
            "posterior_variance": post_var,
            "posterior_log_variance_clipped": np.log(np.maximum(post_var, 1e-20)),
            "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
            "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
        }
        for name, val in buffers.items():
            self.register_buffer(name, torch.tensor(val, dtype=torch.float32))
