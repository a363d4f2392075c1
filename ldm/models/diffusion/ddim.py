"""DDIM sampler (eta = 0) with classifier-free guidance — drop-in for the reference's
ldm/models/diffusion/ddim.py (same constructor, make_schedule, sample / ddim_sampling signatures and
in-place updates of `input`); gligen_inference.py --no_plms selects it with 250 steps
(reference gligen_inference.py:385-387).

The reference only ever builds the schedule with its default ddim_eta = 0 (ddim.py:27,59-62), for which
sigma_t = 0 and p_sample_ddim (ddim.py:111-134) reduces to
    pred_x0 = (x - sqrt(1 - a_t) e_t) / sqrt(a_t);   x_prev = sqrt(a_prev) pred_x0 + sqrt(1 - a_prev) e_t
i.e. the PLMS update with e' = e_t and no multistep history. The device loop is therefore the PLMS loop of
Engine::sample_plms with gl_plms_args.ddim = 1 (one hipGraph-replayed [cond ; uncond] evaluation per step + the
fused CFG / x_prev kernel); any other `model` callable goes through PLMSSampler._sample_generic with
multistep off.
"""
import torch

from ldm.models.diffusion.plms import PLMSSampler


class DDIMSampler(PLMSSampler):
    multistep = False

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=False):
        if ddim_eta != 0:
            raise NotImplementedError("DDIMSampler: only ddim_eta = 0 (the value the reference uses) is implemented")
        return super().make_schedule(ddim_num_steps, ddim_discretize=ddim_discretize, ddim_eta=0.0, verbose=verbose)

    @torch.no_grad()
    def ddim_sampling(self, shape, input, uc, guidance_scale=1, mask=None, x0=None):
        return self.plms_sampling(shape, input, uc, guidance_scale, mask=mask, x0=x0)
