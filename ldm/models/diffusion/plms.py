"""PLMS sampler with classifier-free guidance — drop-in for the reference's
ldm/models/diffusion/plms.py (same constructor, make_schedule, sample signature and in-place
updates of `input`).

With the native UNetModel the whole loop runs on the device (Engine::sample_plms): the
[cond ; uncond] pair is one batched UNet evaluation captured in a hipGraph, followed by a fused
CFG + Adams-Bashforth + x_prev kernel (reference plms.py:116-158). Any other `model` callable
(e.g. a recording mock in host-logic tests) goes through `_sample_generic`, which restates the
reference loop around `model(input)` calls.
"""
import numpy as np
import torch

from ldm.modules.diffusionmodules.util import make_ddim_sampling_parameters, make_ddim_timesteps


def _np(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


class PLMSSampler(object):
    multistep = True  # False in ldm.models.diffusion.ddim.DDIMSampler: e' = e_t, one model evaluation per step

    def __init__(self, diffusion, model, schedule="linear", alpha_generator_func=None, set_alpha_scale=None):
        super().__init__()
        self.diffusion = diffusion
        self.model = model
        self.device = diffusion.betas.device
        self.ddpm_num_timesteps = diffusion.num_timesteps
        self.schedule = schedule
        self.alpha_generator_func = alpha_generator_func
        self.set_alpha_scale = set_alpha_scale
        self.use_graph = True

    def register_buffer(self, name, attr):
        if type(attr) == torch.Tensor:
            attr = attr.to(self.device)
        setattr(self, name, attr)

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0.0, verbose=False):
        if ddim_eta != 0:
            raise ValueError("ddim_eta must be 0 for PLMS")
        self.ddim_timesteps = make_ddim_timesteps(ddim_discr_method=ddim_discretize, num_ddim_timesteps=ddim_num_steps,
                                                  num_ddpm_timesteps=self.ddpm_num_timesteps, verbose=verbose)
        ac = self.diffusion.alphas_cumprod
        assert ac.shape[0] == self.ddpm_num_timesteps, "alphas have to be defined for each timestep"
        f32 = lambda x: x.clone().detach().to(torch.float32).to(self.device)
        self.register_buffer("betas", f32(self.diffusion.betas))
        self.register_buffer("alphas_cumprod", f32(ac))
        self.register_buffer("alphas_cumprod_prev", f32(self.diffusion.alphas_cumprod_prev))
        self.register_buffer("sqrt_alphas_cumprod", f32(np.sqrt(ac.cpu())))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", f32(np.sqrt(1.0 - ac.cpu())))
        sigmas, alphas, alphas_prev = make_ddim_sampling_parameters(alphacums=ac.cpu(), ddim_timesteps=self.ddim_timesteps,
                                                                    eta=ddim_eta, verbose=verbose)
        self.register_buffer("ddim_sigmas", sigmas)
        self.register_buffer("ddim_alphas", alphas)
        self.register_buffer("ddim_alphas_prev", alphas_prev)
        self.register_buffer("ddim_sqrt_one_minus_alphas", np.sqrt(1.0 - alphas))

    @torch.no_grad()
    def sample(self, S, shape, input, uc=None, guidance_scale=1, mask=None, x0=None):
        self.make_schedule(ddim_num_steps=S)
        return self.plms_sampling(shape, input, uc, guidance_scale, mask=mask, x0=x0)

    @torch.no_grad()
    def plms_sampling(self, shape, input, uc=None, guidance_scale=1, mask=None, x0=None):
        if input["x"] is None:
            input["x"] = torch.randn(shape, device=self.device)
        if hasattr(self.model, "engine") and hasattr(self.model, "set_conditioning"):
            return self._sample_native(shape, input, uc, guidance_scale, mask, x0)
        return self._sample_generic(shape, input, uc, guidance_scale, mask, x0)

    # ---- device loop ---------------------------------------------------------------------
    def _sample_native(self, shape, input, uc, guidance_scale, mask, x0):
        model = self.model
        time_range = np.flip(self.ddim_timesteps).copy()
        S = len(time_range)
        alphas = None
        if self.alpha_generator_func is not None:
            alphas = np.asarray(self.alpha_generator_func(S), dtype=np.float32)
        cfg = uc is not None and guidance_scale != 1
        context = input["context"]
        g = input.get("grounding_input")
        if g is None:
            g = model.grounding_tokenizer_input.get_null_input()
        if cfg:
            g_null = model.grounding_tokenizer_input.get_null_input()
            if model.engine.unet_cfg["grounding_kind"] == "tokens":  # spatial-map tokenizers: concatenate their outputs
                tok = lambda kw: model.position_net.tokens(engine=model.engine, **kw)
                g, g_null = {"tokens": tok(g)}, {"tokens": tok(g_null)}
            ctx2 = torch.cat([context, uc.to(context)], dim=0)
            g2 = {k: torch.cat([g[k], g_null[k].to(g[k])], dim=0) for k in g}
        else:
            ctx2, g2 = context, g
        model.set_conditioning(ctx2, g2)
        # The reference's set_alpha_scale matches GatedSelfAttentionDense / GatedCrossAttentionDense by exact type
        # (gligen_inference.py:24-28): a gatedSA2 model keeps scale = 1 on every step, whatever the alpha schedule says.
        # The schedule still decides when the SD first conv is swapped in (plms.py:88-89).
        scales = alphas
        if alphas is None or model.fuser_type == "gatedSA2":
            model.push_fuser_scales()   # whatever the modules carry (1 unless set by hand), possibly one value per fuser
            scales = None

        img = input["x"].to(device=model.engine.device, dtype=torch.float32).contiguous().clone()
        a_t = _np(self.ddim_alphas).astype(np.float32)[::-1].copy()          # index = S - i - 1
        a_prev = _np(self.ddim_alphas_prev).astype(np.float32)[::-1].copy()
        extra = {}
        # Device-generator draws, call for call as the reference makes them, so that equal seeds give equal noise:
        # per step one randn_like(x0) inside q_sample when inpainting (plms.py:96-99 -> ldm.py:19-22), then the
        # sigma_t * randn_like(x) of get_x_prev_and_pred_x0 (plms.py:138; sigma is 0, the draw still advances the
        # generator) -- twice on the first PLMS step (plms.py:144,159), once otherwise and for DDIM (ddim.py:131).
        noise = []
        if mask is not None:
            assert x0 is not None
            x0 = x0.to(device=img.device, dtype=torch.float32)
            mask = mask.to(device=img.device, dtype=torch.float32)
        for i in range(S):
            if mask is not None:
                noise.append(torch.randn_like(x0))
            for _ in range(2 if (i == 0 and self.multistep) else 1):
                torch.randn_like(img)
        if mask is not None:
            if tuple(mask.shape[1:]) != (1, *img.shape[2:]) or tuple(x0.shape[1:]) != tuple(img.shape[1:]):
                raise ValueError(f"inpainting: mask {tuple(mask.shape)} / x0 {tuple(x0.shape)} do not match the latent {tuple(img.shape)}")
            extra = dict(mask=mask, x0=x0, noise=torch.stack(noise),
                         sqrt_ac=self.diffusion.sqrt_alphas_cumprod.cpu().numpy()[time_range],
                         sqrt_1mac=self.diffusion.sqrt_one_minus_alphas_cumprod.cpu().numpy()[time_range])
        gate_off = alphas is not None and bool((alphas == 0).any())
        if gate_off and not model.first_conv_restorable:
            model.restore_first_conv_from_SD()  # prints the reference's "not restorable" notice
        sd_conv = None
        if gate_off and model.first_conv_restorable and not model.__dict__.get("_first_conv_restored"):
            sd_conv = model.load_sd_first_conv()  # swapped in on the device at the first gated-off step
        restore_at = int(np.argmax(alphas == 0)) if sd_conv is not None else -1
        # channels concatenated to x in front of the first conv (reference openaimodel.py:442-447): the masked latent + mask
        # of an inpainting model, or the GroundingDownsampler output of a spatial-map model. The reference's uncond call
        # passes the same tensors (plms.py:118), so one [B] tensor serves both halves of the [cond ; uncond] evaluation; it
        # stays bound after the SD first conv is swapped in (the packed conv keeps its 4 + k layout, those weights are zero).
        if model.inpaint_mode:
            first_conv_extra = input.get("inpainting_extra_input")
            if first_conv_extra is None:
                raise ValueError("inpaint_mode model needs input['inpainting_extra_input']")
        else:
            first_conv_extra = model.first_conv_extra(input)
        model.engine.sample_plms(img, time_range, a_t, a_prev, scales, guidance_scale if cfg else 1.0,
                                 inpaint_extra=first_conv_extra, use_graph=self.use_graph,
                                 sd_first_conv=sd_conv, restore_at=restore_at, ddim=not self.multistep, **extra)
        if sd_conv is not None:
            model.restore_first_conv_from_SD()  # bring the module parameters in line with the engine
        if alphas is not None and self.set_alpha_scale is not None:
            self.set_alpha_scale(model, float(alphas[-1]))  # leave the modules as the reference loop does
        input["x"] = img
        input["timesteps"] = torch.full((shape[0],), int(time_range[-1]), device=img.device, dtype=torch.long)
        return img

    # ---- reference-shaped host loop for arbitrary model callables -------------------------------
    def _sample_generic(self, shape, input, uc, guidance_scale, mask, x0):
        b = shape[0]
        img = input["x"]
        time_range = np.flip(self.ddim_timesteps)
        S = len(time_range)
        alphas = self.alpha_generator_func(S) if self.alpha_generator_func is not None else None
        history = []

        def denoiser(inp):
            e = self.model(inp)
            if uc is not None and guidance_scale != 1:
                e_u = self.model(dict(x=inp["x"], timesteps=inp["timesteps"], context=uc,
                                      inpainting_extra_input=inp["inpainting_extra_input"],
                                      grounding_extra_input=inp["grounding_extra_input"]))
                e = e_u + guidance_scale * (e - e_u)
            return e

        def step_back(x, e, index):
            a_t, a_prev = float(self.ddim_alphas[index]), float(self.ddim_alphas_prev[index])  # noqa: E501
            pred_x0 = (x - float(np.sqrt(1.0 - a_t)) * e) / float(np.sqrt(a_t))
            return float(np.sqrt(a_prev)) * pred_x0 + float(np.sqrt(1.0 - a_prev)) * e

        for i, step in enumerate(time_range):
            if alphas is not None:
                self.set_alpha_scale(self.model, alphas[i])
                if alphas[i] == 0:
                    self.model.restore_first_conv_from_SD()
            index = S - i - 1
            ts = torch.full((b,), int(step), device=self.device, dtype=torch.long)
            ts_next = torch.full((b,), int(time_range[min(i + 1, S - 1)]), device=self.device, dtype=torch.long)
            if mask is not None:
                assert x0 is not None
                img = self.diffusion.q_sample(x0, ts) * mask + (1.0 - mask) * img
                input["x"] = img
            x = input["x"].clone()
            input["timesteps"] = ts
            e_t = denoiser(input)
            if not self.multistep:
                e_prime = e_t
            elif len(history) == 0:
                input["x"] = step_back(x, e_t, index)
                input["timesteps"] = ts_next
                e_prime = (e_t + denoiser(input)) / 2
            elif len(history) == 1:
                e_prime = (3 * e_t - history[-1]) / 2
            elif len(history) == 2:
                e_prime = (23 * e_t - 16 * history[-1] + 5 * history[-2]) / 12
            else:
                e_prime = (55 * e_t - 59 * history[-1] + 37 * history[-2] - 9 * history[-3]) / 24
            img = step_back(x, e_prime, index)
            input["x"] = img
            history = (history + [e_t])[-3:]
        return img
