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


# ---- fabricated checkpoint pieces shared by the CPU loader tests and the GPU file -> image test ----------------------------
def _fake_omegaconf_pickle(path, payload):
    """Write `payload` (a dict of state_dicts + a nested config) the way the reference's trainer does
    (trainer.py:176,472-480: config_dict = vars(OmegaConf DictConfig)), with stand-in classes living in modules NAMED
    omegaconf.* so that the pickle stream references 'omegaconf.dictconfig DictConfig' etc. exactly like a real checkpoint."""
    import sys
    import types

    mods = {n: types.ModuleType(n) for n in ("omegaconf", "omegaconf.dictconfig", "omegaconf.listconfig", "omegaconf.nodes", "omegaconf.base")}

    def cls(mod, name):
        c = type(name, (), {"__module__": mod, "__getstate__": lambda self: dict(self.__dict__),
                            "__setstate__": lambda self, st: self.__dict__.update(st)})
        setattr(mods[mod], name, c)
        return c

    DictConfig, ListConfig = cls("omegaconf.dictconfig", "DictConfig"), cls("omegaconf.listconfig", "ListConfig")
    AnyNode, Meta = cls("omegaconf.nodes", "AnyNode"), cls("omegaconf.base", "ContainerMetadata")

    def wrap(v):
        if isinstance(v, dict):
            n = DictConfig()
            n.__dict__.update(_content={k: wrap(x) for k, x in v.items()}, _metadata=Meta(), _parent=None, _flags_cache=None)
            return n
        if isinstance(v, (list, tuple)):
            n = ListConfig()
            n.__dict__.update(_content=[wrap(x) for x in v], _metadata=Meta(), _parent=None, _flags_cache=None)
            return n
        n = AnyNode()
        n.__dict__.update(_val=v, _metadata=Meta(), _parent=None)
        return n

    cfg = payload.pop("config")
    payload["config_dict"] = dict(_content={k: wrap(v) for k, v in cfg.items()}, _metadata=Meta(), _parent=None, _flags_cache=None)
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        torch.save(payload, path)
    finally:
        for k, v in saved.items():
            if v is None:
                del sys.modules[k]
            else:
                sys.modules[k] = v


def _fabricated_clip(tmp_path, hidden=768, proj=768):
    import json as _json
    from transformers import CLIPConfig, CLIPImageProcessor, CLIPModel, CLIPProcessor, CLIPTextConfig, CLIPTokenizer, CLIPVisionConfig
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b); cs.append(256 + n); n += 1
    chars = [chr(c) for c in cs]
    vocab = {c: i for i, c in enumerate(chars)}
    vocab.update({c + "</w>": len(chars) + i for i, c in enumerate(chars)})
    vocab["<|startoftext|>"], vocab["<|endoftext|>"] = len(vocab), len(vocab) + 1
    (tmp_path / "vocab.json").write_text(_json.dumps(vocab))
    (tmp_path / "merges.txt").write_text("#version: 0.2\n")
    tok = CLIPTokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"))
    tcfg = CLIPTextConfig(vocab_size=49408, hidden_size=hidden, intermediate_size=256, num_hidden_layers=2, num_attention_heads=8,
                          max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=proj, eos_token_id=vocab["<|endoftext|>"],
                          bos_token_id=vocab["<|startoftext|>"], pad_token_id=vocab["<|endoftext|>"])
    vcfg = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, image_size=224, patch_size=32,
                            projection_dim=proj)
    torch.manual_seed(0)
    model = CLIPModel(CLIPConfig(text_config=tcfg.to_dict(), vision_config=vcfg.to_dict(), projection_dim=proj)).eval()
    return model, CLIPProcessor(image_processor=CLIPImageProcessor(), tokenizer=tok), tok
