"""GLIGEN inference entry point for MI355X — same functions, flags and flow as the reference's
gligen_inference.py (load_ckpt :70-86, prepare_batch :146-187, prepare_batch_kp :199-218,
run :343-446, flags :451-463), with the denoising loop and the decode executed by libgligen_amd.so.

Differences that follow from running offline / on the native engine:
  * `meta` may carry precomputed CLIP features (`text_embeddings`, `image_embeddings`, `context`,
    `uc`) — without them the HF CLIP weights are loaded exactly as the reference does (needs the
    hub cache); `--synthetic` builds seeded random-weight models and features so the whole path
    runs with no checkpoint at all;
  * checkpoints embed a pickled OmegaConf config: it is read with omegaconf when installed, else
    through a minimal unpickling shim (`_load_pickled_config`).
"""
import argparse
import os
import pickle
from functools import partial

import numpy as np
import torch
from PIL import Image

from ldm.models.diffusion.ddim import DDIMSampler
from ldm.models.diffusion.plms import PLMSSampler
from ldm.util import instantiate_from_config

device = "cuda"


def set_alpha_scale(model, alpha_scale):
    """Set the external gate multiplier on every fuser (exact-type match, as the reference)."""
    from ldm.modules.attention import GatedCrossAttentionDense, GatedSelfAttentionDense
    for module in model.modules():
        if type(module) == GatedCrossAttentionDense or type(module) == GatedSelfAttentionDense:
            module.scale = alpha_scale


def alpha_generator(length, type=None):
    """Per-step gate schedule: `type` = fractions of (alpha = 1, linear decay, alpha = 0) stages."""
    if type is None:
        type = [1, 0, 0]
    assert len(type) == 3
    assert type[0] + type[1] + type[2] == 1
    n_on = int(type[0] * length)
    n_decay = int(type[1] * length)
    n_off = length - n_on - n_decay
    decay = list(np.arange(start=0, stop=1, step=1 / n_decay)[::-1]) if n_decay != 0 else []
    alphas = [1] * n_on + decay + [0] * n_off
    assert len(alphas) == length
    return alphas


# ---- checkpoint loading ---------------------------------------------------------------------
class _Node(dict):
    """Stand-in for omegaconf container classes met while unpickling a checkpoint's config_dict."""

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {})


def _plain(obj):
    """omegaconf node graph (or our shim of it) -> plain python containers."""
    content = getattr(obj, "_content", None) if not isinstance(obj, (dict, list)) else None
    if content is None and isinstance(obj, _Node):
        content = obj.__dict__.get("_content")
    if content is not None:
        obj = content
    val = getattr(obj, "_val", None) if not isinstance(obj, (dict, list, str, int, float, bool, type(None))) else None
    if val is not None:
        return _plain(val)
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    return obj


class _ShimUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module.startswith("omegaconf"):
            return type(name, (_Node,), {})
        return super().find_class(module, name)


class _ShimPickle:
    Unpickler = _ShimUnpickler
    __name__ = "pickle"

    @staticmethod
    def load(f, **kw):
        return _ShimUnpickler(f, **kw).load()


def read_ckpt(ckpt_path):
    try:
        import omegaconf  # noqa: F401
        return torch.load(ckpt_path, map_location="cpu", weights_only=False)
    except ImportError:
        return torch.load(ckpt_path, map_location="cpu", weights_only=False, pickle_module=_ShimPickle)


def load_ckpt(ckpt_path):
    saved_ckpt = read_ckpt(ckpt_path)
    config = _plain(saved_ckpt["config_dict"]["_content"])
    model = instantiate_from_config(config["model"]).to(device).eval()
    autoencoder = instantiate_from_config(config["autoencoder"]).to(device).eval()
    text_encoder = instantiate_from_config(config["text_encoder"]).to(device).eval()
    diffusion = instantiate_from_config(config["diffusion"]).to(device)
    model.load_state_dict(saved_ckpt["model"])
    autoencoder.load_state_dict(saved_ckpt["autoencoder"])
    text_encoder.load_state_dict(saved_ckpt["text_encoder"])
    diffusion.load_state_dict(saved_ckpt["diffusion"])
    return model, autoencoder, text_encoder, diffusion, config


# ---- batch preparation ---------------------------------------------------------------------------
def project(x, projection_matrix):
    """penultimate CLIP feature (B,768) -> un-normalised CLIP embedding; matrix is Linear.weight (out,in)."""
    return x @ torch.transpose(projection_matrix, 0, 1)


_CLIP = {}


def _clip():
    if not _CLIP:
        from transformers import CLIPModel, CLIPProcessor
        version = "openai/clip-vit-large-patch14"
        _CLIP["model"] = CLIPModel.from_pretrained(version).to(device)
        _CLIP["processor"] = CLIPProcessor.from_pretrained(version)
    return _CLIP["model"], _CLIP["processor"]


@torch.no_grad()
def get_clip_feature(model, processor, input, is_image=False):
    """Text: pooler_output before projection. Image: image_embeds re-projected with the text
    projection matrix and scaled to norm 28.7 (reference gligen_inference.py:104-128)."""
    if input is None:
        return None
    if is_image:
        image = Image.open(input).convert("RGB")
        inputs = processor(images=[image], return_tensors="pt", padding=True)
        inputs["pixel_values"] = inputs["pixel_values"].to(device)
        inputs["input_ids"] = torch.tensor([[0, 1, 2, 3]]).to(device)
        feature = model(**inputs).image_embeds
        feature = project(feature, torch.load("projection_matrix").to(device).T).squeeze(0)
        return ((feature / feature.norm()) * 28.7).unsqueeze(0)
    inputs = processor(text=input, return_tensors="pt", padding=True)
    inputs["input_ids"] = inputs["input_ids"].to(device)
    inputs["pixel_values"] = torch.ones(1, 3, 224, 224).to(device)
    inputs["attention_mask"] = inputs["attention_mask"].to(device)
    return model(**inputs).text_model_output.pooler_output


def complete_mask(has_mask, max_objs):
    mask = torch.ones(1, max_objs)
    if has_mask is None:
        return mask
    if type(has_mask) == int or type(has_mask) == float:
        return mask * has_mask
    for idx, value in enumerate(has_mask):
        mask[0, idx] = value
    return mask


def batch_to_device(batch, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


@torch.no_grad()
def prepare_batch(meta, batch=1, max_objs=30):
    phrases, images = meta.get("phrases"), meta.get("images")
    n = len(meta["locations"])
    images = [None] * n if images is None else images
    phrases = [None] * n if phrases is None else phrases
    text_features, image_features = meta.get("text_embeddings"), meta.get("image_embeddings")
    if text_features is None and image_features is None:  # the reference path: encode with CLIP ViT-L/14
        model, processor = _clip()
        text_features = [get_clip_feature(model, processor, p, is_image=False) for p in phrases]
        image_features = [get_clip_feature(model, processor, im, is_image=True) for im in images]
    text_features = [None] * n if text_features is None else text_features
    image_features = [None] * n if image_features is None else image_features

    boxes, masks = torch.zeros(max_objs, 4), torch.zeros(max_objs)
    text_masks, image_masks = torch.zeros(max_objs), torch.zeros(max_objs)
    text_embeddings, image_embeddings = torch.zeros(max_objs, 768), torch.zeros(max_objs, 768)
    for idx, (box, tf, imf) in enumerate(zip(meta["locations"], text_features, image_features)):
        boxes[idx] = torch.as_tensor(box, dtype=torch.float32)
        masks[idx] = 1
        if tf is not None:
            text_embeddings[idx] = torch.as_tensor(tf).reshape(-1).float().cpu()
            text_masks[idx] = 1
        if imf is not None:
            image_embeddings[idx] = torch.as_tensor(imf).reshape(-1).float().cpu()
            image_masks[idx] = 1
    out = {
        "boxes": boxes.unsqueeze(0).repeat(batch, 1, 1),
        "masks": masks.unsqueeze(0).repeat(batch, 1),
        "text_masks": text_masks.unsqueeze(0).repeat(batch, 1) * complete_mask(meta.get("text_mask"), max_objs),
        "image_masks": image_masks.unsqueeze(0).repeat(batch, 1) * complete_mask(meta.get("image_mask"), max_objs),
        "text_embeddings": text_embeddings.unsqueeze(0).repeat(batch, 1, 1),
        "image_embeddings": image_embeddings.unsqueeze(0).repeat(batch, 1, 1),
    }
    return batch_to_device(out, device)


@torch.no_grad()
def prepare_batch_kp(meta, batch=1, max_persons_per_image=8):
    points = torch.zeros(max_persons_per_image * 17, 2)
    idx = 0
    for person in meta["locations"]:
        for kp in person:
            points[idx, 0], points[idx, 1] = kp[0], kp[1]
            idx += 1
    masks = ((points.mean(dim=1) != 0) * 1).float()
    out = {"points": points.unsqueeze(0).repeat(batch, 1, 1), "masks": masks.unsqueeze(0).repeat(batch, 1)}
    return batch_to_device(out, device)


def draw_masks_from_boxes(boxes, size):
    """Inpainting mask: 1 outside, 0 inside int(box*size) rectangles (reference inpaint_mask_func.py:16-41)."""
    masks = []
    for per_image in boxes:
        m = torch.ones(size, size)
        for bx in per_image:
            x0, y0, x1, y1 = (int(v) for v in bx * size)
            m[y0:y1, x0:x1] = 0
        masks.append(m)
    return torch.stack(masks).unsqueeze(1)


# ---- synthetic models (no checkpoints offline) ---------------------------------------------------------
_KIND_TO_INPUT = {
    "text": "grounding_input.text_grounding_tokinzer_input.GroundingNetInput",
    "text_image": "grounding_input.text_image_grounding_tokinzer_input.GroundingNetInput",
    "keypoint": "grounding_input.keypoint_grounding_tokinzer_input.GroundingNetInput",
}


def synthetic_config(kind="text", inpaint=False, image_size=64):
    from gligen_amd import synthetic as syn
    return {
        "diffusion": dict(target="ldm.models.diffusion.ldm.LatentDiffusion", params=dict(linear_start=0.00085, linear_end=0.012, timesteps=1000)),
        "model": dict(target="ldm.modules.diffusionmodules.openaimodel.UNetModel",
                      params=dict(syn.UNET_CFG, image_size=image_size, inpaint_mode=inpaint, grounding_tokenizer=syn.GROUNDING_TOKENIZERS[kind])),
        "autoencoder": dict(target="ldm.models.autoencoder.AutoencoderKL", params=dict(scale_factor=0.18215, embed_dim=4, ddconfig=syn.VAE_DDCONFIG)),
        "grounding_tokenizer_input": dict(target=_KIND_TO_INPUT[kind]),
        "inpaint_mode": inpaint,
    }


def load_synthetic(kind="text", inpaint=False, image_size=64, seed=1234, fast=False):
    """Seeded random-weight stand-ins for (model, autoencoder, diffusion, config).
    fast=True materialises the 1.07 B parameters directly on the device (same init distributions,
    drawn from the device generator) instead of filling them on the CPU and copying."""
    from gligen_amd import synthetic as syn
    config = synthetic_config(kind, inpaint, image_size)
    if fast:
        with torch.device("meta"):
            model = instantiate_from_config(config["model"]).eval()
            autoencoder = instantiate_from_config(config["autoencoder"]).eval()
        model = syn.fill_module_on_device_(model.to_empty(device=device), seed)
        autoencoder = syn.fill_module_on_device_(autoencoder.to_empty(device=device), seed + 1)
    else:
        model = syn.fill_module_(instantiate_from_config(config["model"]).eval(), seed).to(device)
        autoencoder = syn.fill_module_(instantiate_from_config(config["autoencoder"]).eval(), seed + 1).to(device)
    diffusion = instantiate_from_config(config["diffusion"]).to(device)
    return model, autoencoder, diffusion, config


# ---- run ---------------------------------------------------------------------------------------------
@torch.no_grad()
def generate(model, autoencoder, diffusion, batch, context, uc, *, steps=50, guidance_scale=7.5, alpha_type=None,
             starting_noise=None, inpainting_mask=None, z0=None, use_graph=True, no_plms=False):
    """The sampling core of run() (reference gligen_inference.py:389-431) on already-encoded inputs."""
    # reference gligen_inference.py:385-390: DDIM (250 steps) with --no_plms, else PLMS (50 steps)
    sampler_cls = DDIMSampler if no_plms else PLMSSampler
    sampler = sampler_cls(diffusion, model, alpha_generator_func=partial(alpha_generator, type=alpha_type), set_alpha_scale=set_alpha_scale)
    sampler.use_graph = use_graph
    inpainting_extra_input = None
    if inpainting_mask is not None:
        inpainting_extra_input = torch.cat([z0 * inpainting_mask, inpainting_mask], dim=1)
    grounding_input = model.grounding_tokenizer_input.prepare(batch)
    input = dict(x=starting_noise, timesteps=None, context=context, grounding_input=grounding_input,
                 inpainting_extra_input=inpainting_extra_input, grounding_extra_input=None)
    B = context.shape[0]
    shape = (B, model.in_channels, model.image_size, model.image_size)
    latents = sampler.sample(S=steps, shape=shape, input=input, uc=uc, guidance_scale=guidance_scale, mask=inpainting_mask, x0=z0)
    return autoencoder.decode(latents)


def save_images(samples, output_folder):
    os.makedirs(output_folder, exist_ok=True)
    start = len(os.listdir(output_folder))
    ids = list(range(start, start + samples.shape[0]))
    print(ids)
    for image_id, sample in zip(ids, samples):
        sample = torch.clamp(sample, min=-1, max=1) * 0.5 + 0.5
        arr = sample.cpu().numpy().transpose(1, 2, 0) * 255
        Image.fromarray(arr.astype(np.uint8)).save(os.path.join(output_folder, str(int(image_id)) + ".png"))


@torch.no_grad()
def run(meta, config, starting_noise=None, models=None):
    """config: argparse namespace / dict with batch_size, guidance_scale, negative_prompt, no_plms, folder."""
    args = dict(config) if isinstance(config, dict) else vars(config)
    if models is None:
        model, autoencoder, text_encoder, diffusion, ckpt_config = load_ckpt(meta["ckpt"])
    else:
        model, autoencoder, text_encoder, diffusion, ckpt_config = models
    model.grounding_tokenizer_input = instantiate_from_config(ckpt_config["grounding_tokenizer_input"])
    if "grounding_downsampler_input" in ckpt_config:
        raise NotImplementedError("spatial-map checkpoints (hed/canny/depth/normal/sem) are outside the MI355X hot path")
    B = args["batch_size"]
    batch = prepare_batch_kp(meta, B) if "keypoint" in meta["ckpt"] else prepare_batch(meta, B)
    if "context" in meta:  # precomputed CLIP last_hidden_state (B,77,768)
        context, uc = meta["context"].to(device), meta["uc"].to(device)
    else:
        context = text_encoder.encode([meta["prompt"]] * B)
        uc = text_encoder.encode(B * [args.get("negative_prompt") or ""])
    mask = z0 = None
    if "input_image" in meta:
        assert ckpt_config.get("inpaint_mode"), "input_image is given, the ckpt must be the inpaint model, are you using the correct ckpt?"
        mask = draw_masks_from_boxes(batch["boxes"], model.image_size).to(device)
        if "z0" in meta:
            z0 = meta["z0"].to(device)
        else:
            img = torch.from_numpy(np.asarray(Image.open(meta["input_image"]).convert("RGB").resize((512, 512)))).permute(2, 0, 1)
            z0 = autoencoder.encode((img.float().unsqueeze(0).to(device) / 255 - 0.5) / 0.5)
    no_plms = bool(args.get("no_plms"))
    samples = generate(model, autoencoder, diffusion, batch, context, uc, steps=250 if no_plms else 50, guidance_scale=args["guidance_scale"],
                       alpha_type=meta.get("alpha_type"), starting_noise=starting_noise, inpainting_mask=mask, z0=z0, no_plms=no_plms)
    save_images(samples, os.path.join(args["folder"], meta["save_folder_name"]))
    return samples


def _synthetic_meta(kind, B):
    from gligen_amd import synthetic as syn
    boxes, _ = syn.make_boxes(1, 8, seed=0)
    emb = syn.make_embeddings(1, 8, seed=0)[0, :8]
    meta = dict(ckpt=f"synthetic_{kind}", prompt="synthetic", save_folder_name=f"synthetic_{kind}",
                context=syn.make_context(B, seed=0), uc=syn.make_context(B, seed=1))
    if kind == "keypoint":
        pts = syn.make_batch("keypoint", 1)["points"][0, :34].reshape(2, 17, 2)
        meta["locations"] = pts.tolist()
    else:
        meta.update(locations=boxes[0, :8].tolist(), text_embeddings=list(emb))
        if kind == "text_image":
            meta["image_embeddings"] = list(syn.make_embeddings(1, 8, seed=7)[0, :8])
    return meta


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--folder", type=str, default="generation_samples", help="root folder for output")
    parser.add_argument("--batch_size", type=int, default=5, help="")
    parser.add_argument("--no_plms", action="store_true", help="use DDIM instead. WARNING: I did not test the code yet")
    parser.add_argument("--guidance_scale", type=float, default=7.5, help="")
    parser.add_argument("--negative_prompt", type=str,
                        default="longbody, lowres, bad anatomy, bad hands, missing fingers, extra digit, fewer digits, cropped, worst quality, low quality", help="")
    parser.add_argument("--synthetic", type=str, default=None, choices=["text", "text_image", "keypoint"],
                        help="run with seeded random weights and features (no checkpoint / CLIP needed)")
    parser.add_argument("--ckpt", type=str, default=None, help="GLIGEN checkpoint (diffusion_pytorch_model.bin)")
    args = parser.parse_args()
    if args.synthetic:
        model, autoencoder, diffusion, cfg = load_synthetic(args.synthetic)
        run(_synthetic_meta(args.synthetic, args.batch_size), args, models=(model, autoencoder, None, diffusion, cfg))
    else:
        if not args.ckpt:
            parser.error("--ckpt or --synthetic is required (the reference's demo meta_list needs downloaded checkpoints)")
        meta = dict(ckpt=args.ckpt, prompt="a teddy bear sitting next to a bird", phrases=["a teddy bear", "a bird"],
                    locations=[[0.0, 0.09, 0.33, 0.76], [0.55, 0.11, 1.0, 0.8]], alpha_type=[0.3, 0.0, 0.7],
                    save_folder_name="generation_box_text")
        run(meta, args)
