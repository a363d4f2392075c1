"""Host-side helpers of the diffusion modules (schedules, embeddings, layer factories).

Same names and semantics as the reference's ldm/modules/diffusionmodules/util.py; the device
versions of timestep_embedding / FourierEmbedder live in gligen_amd/csrc/misc.hip."""
import math

import numpy as np
import torch
import torch.nn as nn


class FourierEmbedder:
    """sin/cos features at frequencies temperature^(k/num_freqs), concatenated as
    [k][sin, cos][coord] (reference util.py:12-26)."""

    def __init__(self, num_freqs=64, temperature=100):
        self.num_freqs = num_freqs
        self.temperature = temperature
        self.freq_bands = temperature ** (torch.arange(num_freqs) / num_freqs)

    @torch.no_grad()
    def __call__(self, x, cat_dim=-1):
        feats = [fn(float(f) * x) for f in self.freq_bands for fn in (torch.sin, torch.cos)]
        return torch.cat(feats, cat_dim)


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """The fp64 "linear" beta schedule (reference util.py:30-34) -- the only one a shipped GLIGEN / SD-1.4 config selects
    (configs/*.yaml: beta_schedule is never set, ddpm.py:11 defaults to "linear"). `cosine_s` stays in the signature for callers
    that pass it through; another schedule name is an error here rather than an untested branch."""
    if schedule != "linear":
        raise ValueError(f"schedule '{schedule}' is not built: every shipped config uses 'linear' (reference util.py:30-52)")
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    return betas.numpy()


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """Sub-sampled timesteps, shifted by +1 (reference util.py:55-69)."""
    if ddim_discr_method == "uniform":
        stride = num_ddpm_timesteps // num_ddim_timesteps
        steps = np.arange(0, num_ddpm_timesteps, stride)
    elif ddim_discr_method == "quad":
        steps = (np.linspace(0, np.sqrt(num_ddpm_timesteps * 0.8), num_ddim_timesteps) ** 2).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps = steps + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps}")
    return steps


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """alphas / alphas_prev / sigmas of the sub-sampled chain (reference util.py:72-83)."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([alphacums[0]] + alphacums[ddim_timesteps[:-1]].tolist())
    sigmas = eta * np.sqrt((1 - alphas_prev) / (1 - alphas) * (1 - alphas / alphas_prev))
    if verbose:
        print(f"ddim alphas {alphas}; alphas_prev {alphas_prev}; sigmas {sigmas}")
    return sigmas, alphas, alphas_prev


def extract_into_tensor(a, t, x_shape):
    out = a.gather(-1, t)
    return out.reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    """[cos | sin] sinusoidal embedding (reference util.py:160-180)."""
    if repeat_only:
        return timesteps[:, None].expand(-1, dim)
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half).to(timesteps.device)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def zero_module(module):
    for p in module.parameters():
        p.detach().zero_()
    return module


class GroupNorm32(nn.GroupNorm):
    """Parameter holder: 32 groups, eps 1e-5, statistics in fp32 (reference util.py:223-226)."""


def normalization(channels):
    return GroupNorm32(32, channels)


def conv_nd(dims, *args, **kwargs):
    if dims != 2:
        raise ValueError(f"unsupported dimensions: {dims} (the MI355X engine implements 2D convolutions)")
    return nn.Conv2d(*args, **kwargs)


def linear(*args, **kwargs):
    return nn.Linear(*args, **kwargs)


def noise_like(shape, device, repeat=False):
    if repeat:
        return torch.randn((1, *shape[1:]), device=device).repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)
