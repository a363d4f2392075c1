"""Shared test plumbing: golden loading, seeded models (product side) and oracle inputs."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gligen_amd import synthetic as syn  # noqa: E402

GINPUT = {
    "text": "grounding_input.text_grounding_tokinzer_input.GroundingNetInput",
    "text_image": "grounding_input.text_image_grounding_tokinzer_input.GroundingNetInput",
    "keypoint": "grounding_input.keypoint_grounding_tokinzer_input.GroundingNetInput",
}


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {k: z[k] for k in z.files}
    if "meta" in out:
        out["meta"] = json.loads(str(out["meta"]))
    return out


def golden_shapes(name):
    return json.load(open(os.path.join(GOLDEN, "state_dict_shapes.json")))[name]


def oracle_cfg(cfg, kind):
    return dict(model_channels=cfg["model_channels"], channel_mult=cfg["channel_mult"], num_res_blocks=cfg["num_res_blocks"],
                attention_resolutions=cfg["attention_resolutions"], num_heads=cfg["num_heads"], grounding_kind=kind,
                fuser_type=cfg.get("fuser_type", "gatedSA"))


def unet_inputs(meta):
    """The exact inputs oracle/make_golden.py:unet_case fed to the reference."""
    B, hw, kind = meta["B"], meta["hw"], meta["kind"]
    batch = syn.make_batch(kind, B, n_valid=meta["n_valid"], seed=1, max_objs=meta.get("max_objs", 30))
    x = syn.make_latent(B, 4, hw, hw, seed=1)
    ctx = syn.make_context(B, seed=1)
    t = torch.tensor([981, 441][:B] if B <= 2 else [981] * B, dtype=torch.long)
    extra = None
    if meta["inpaint"]:
        from oracle.gligen_oracle import draw_masks_from_boxes
        mask = draw_masks_from_boxes(batch["boxes"], hw)
        z0 = syn.make_latent(B, 4, hw, hw, seed=2)
        extra = torch.cat([z0 * mask, mask], dim=1)
    return batch, x, ctx, t, extra


def grounding_kwargs(kind, batch):
    if kind == "text":
        return dict(boxes=batch["boxes"], masks=batch["masks"], positive_embeddings=batch["text_embeddings"])
    return dict(batch)


def build_product_unet(cfg, kind, inpaint=False, seed=1234, device=None):
    """This repo's drop-in UNetModel with seeded weights (+ its grounding_tokenizer_input)."""
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.util import instantiate_from_config
    model = UNetModel(**dict(cfg, grounding_tokenizer=syn.GROUNDING_TOKENIZERS[kind], inpaint_mode=inpaint)).eval()
    syn.fill_module_(model, seed)
    model.grounding_tokenizer_input = instantiate_from_config(dict(target=GINPUT[kind]))
    if device is not None:
        model = model.to(device)
    return model


def build_product_vae(ddconfig, seed=4321, device=None):
    from ldm.models.autoencoder import AutoencoderKL
    ae = AutoencoderKL(ddconfig=ddconfig, embed_dim=4, scale_factor=0.18215).eval()
    syn.fill_module_(ae, seed)
    if device is not None:
        ae = ae.to(device)
    return ae


def mse(a, b):
    a = torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).float().cpu()
    b = torch.as_tensor(np.asarray(b) if not torch.is_tensor(b) else b).float().cpu()
    return float(((a - b) ** 2).mean())


def scripted_randn_like(noise, multistep=True):
    """A stand-in for torch.randn_like that replays recorded q_sample draws in the call order of the reference's sampling
    loop (plms.py:96-99 then the sigma_t * randn_like(x) of get_x_prev_and_pred_x0, twice on the first PLMS step):
    noise[i] for the q_sample slot of step i, zeros for the sigma = 0 draws."""
    state = dict(step=0, slot=0)

    def fn(like, *a, **k):
        i, slot = state["step"], state["slot"]
        n_dummy = 2 if (i == 0 and multistep) else 1
        out = noise[i].to(like) if slot == 0 else torch.zeros_like(like)
        if slot == 0:
            assert tuple(out.shape) == tuple(like.shape), (out.shape, like.shape)
        state["slot"] += 1
        if state["slot"] > n_dummy:
            state["step"], state["slot"] = i + 1, 0
        return out
    return fn


def block_backward_inputs(meta):
    """The inputs of the training-slice golden (oracle/make_golden.py: block_backward_case), regenerated from the same seeded
    CPU generator in the same order: x, objs, context, target."""
    import torch
    g = torch.Generator().manual_seed(4242)
    B, N, C, Ng, D, T = meta["B"], meta["hw"] ** 2, meta["C"], meta["Ng"], meta["ctx_dim"], meta["ctx_T"]
    x = torch.randn(B, N, C, generator=g)
    objs = torch.randn(B, Ng, D, generator=g) * 0.5
    context = torch.randn(B, T, D, generator=g)
    target = torch.randn(B, N, C, generator=g)
    return x, objs, context, target


def resblock_backward_inputs(meta):
    """The inputs of the ResBlock training-slice goldens (oracle/make_golden.py: resblock_backward_case), regenerated from the same
    seeded CPU generator in the same order: x, emb, target."""
    import torch
    g = torch.Generator().manual_seed(4343)
    B, hw = meta["B"], meta["hw"]
    x = torch.randn(B, meta["Cin"], hw, hw, generator=g)
    emb = torch.randn(B, meta["emb_dim"], generator=g)
    target = torch.randn(B, meta["Cout"], hw, hw, generator=g)
    return x, emb, target


def st_backward_inputs(meta):
    """The inputs of the SpatialTransformer training-slice golden (oracle/make_golden.py: st_backward_case): x, objs, context, target."""
    import torch
    g = torch.Generator().manual_seed(4444)
    B, hw, C, Ng, D, T = meta["B"], meta["hw"], meta["C"], meta["Ng"], meta["ctx_dim"], meta["ctx_T"]
    x = torch.randn(B, C, hw, hw, generator=g)
    objs = torch.randn(B, Ng, D, generator=g) * 0.5
    context = torch.randn(B, T, D, generator=g)
    target = torch.randn(B, C, hw, hw, generator=g)
    return x, objs, context, target
