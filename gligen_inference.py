"""GLIGEN inference entry point for MI355X — same functions, flags and flow as the reference's
gligen_inference.py (load_ckpt :70-86, prepare_batch :146-187, prepare_batch_kp :199-218,
run :343-446, flags :451-463), with the denoising loop and the decode executed by libgligen_amd.so.

Differences that follow from running offline / on the native engine:
  * `meta` may carry precomputed CLIP features (`text_embeddings`, `image_embeddings`, `context`,
    `uc`) — without them the HF CLIP weights are loaded exactly as the reference does (needs the
    hub cache); `--synthetic` builds seeded random-weight models and features so the whole path
    runs with no checkpoint at all;
  * launched under torch.distributed.run (one process per GPU) the batch is sharded across the ranks: x_T is drawn once
    from `--seed` for the whole batch and sliced, so an N-GPU run writes the same images as a 1-GPU run (no data-path
    collective; the reference has no multi-GPU inference at all, gligen_inference.py:21);
  * checkpoints embed a pickled OmegaConf config: it is read with omegaconf when installed, else
    through a minimal unpickling shim (`_load_pickled_config`).
"""
import argparse
import os
import pickle
import time
from functools import partial

import numpy as np
import torch
from PIL import Image

from ldm.models.diffusion.ddim import DDIMSampler
from ldm.models.diffusion.plms import PLMSSampler
from ldm.util import instantiate_from_config

device = "cuda"
# run(): a per-GPU batch is split into two half-batches in flight (generate_lanes) from this many images on; below it one UNet
# evaluation over the whole batch is faster than two half-size ones side by side (profiles/r6/batch_sweep.txt)
SPLIT_BATCH_AT = 32


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
    """omegaconf node graph (real omegaconf classes, or the _Node stand-ins the shim unpickler builds) -> plain python
    containers. Containers keep their children in `_content`, value nodes (AnyNode, StringNode, ...) their value in `_val`."""
    state = getattr(obj, "__dict__", None)
    if isinstance(state, dict) and not isinstance(obj, type):
        if "_content" in state:
            return _plain(state["_content"])
        if "_val" in state:
            return _plain(state["_val"])
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


def crop_and_resize(image):
    """center crop to a square, resize to 512 x 512 (reference gligen_inference.py:189-193)."""
    w, h = image.size
    c = min(w, h)
    left, top = int(round((w - c) / 2.0)), int(round((h - c) / 2.0))
    return image.crop((left, top, left + c, top + c)).resize((512, 512))


def _pil_to_unit_tensor(image):
    arr = np.asarray(image)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    return (torch.from_numpy(arr.copy()).permute(2, 0, 1).float() / 255 - 0.5) / 0.5


def _prepare_batch_map(meta, batch, meta_key, batch_key):
    """prepare_batch_hed / _canny / _depth / _normal (reference gligen_inference.py:221-300): the map as an RGB image in
    [-1, 1], repeated over the batch, mask = 1. meta[meta_key] is a file name or an already-loaded tensor [3,512,512]."""
    src = meta[meta_key]
    img = src.float() if torch.is_tensor(src) else _pil_to_unit_tensor(crop_and_resize(Image.open(src).convert("RGB")))
    out = {batch_key: img.unsqueeze(0).repeat(batch, 1, 1, 1), "mask": torch.ones(batch, 1)}
    return batch_to_device(out, device)


def prepare_batch_hed(meta, batch=1):
    return _prepare_batch_map(meta, batch, "hed_image", "hed_edge")


def prepare_batch_canny(meta, batch=1):
    return _prepare_batch_map(meta, batch, "canny_image", "canny_edge")


def prepare_batch_depth(meta, batch=1):
    return _prepare_batch_map(meta, batch, "depth", "depth")


def prepare_batch_normal(meta, batch=1):
    return _prepare_batch_map(meta, batch, "normal", "normal")


@torch.no_grad()
def prepare_batch_sem(meta, batch=1):
    """ADE class-index image -> 152 one-hot planes at 512 x 512, nearest resize (reference gligen_inference.py:318-338;
    the colour visualisation it also writes is a side effect, not an input of the model)."""
    src = meta["sem"]
    if torch.is_tensor(src):
        sem = src.long()
    else:
        im = Image.open(src).convert("L")
        w, h = im.size
        c = min(w, h)
        left, top = int(round((w - c) / 2.0)), int(round((h - c) / 2.0))
        im = im.crop((left, top, left + c, top + c)).resize((512, 512), Image.NEAREST)
        sem = torch.from_numpy(np.asarray(im).copy()).long()
    planes = torch.zeros(152, 512, 512).scatter_(0, sem.unsqueeze(0), 1.0)
    out = {"sem": planes.unsqueeze(0).repeat(batch, 1, 1, 1), "mask": torch.ones(batch, 1)}
    return batch_to_device(out, device)


# run() picks the batch builder by checkpoint-name substring, in this order (reference gligen_inference.py:363-376)
_PREPARE_BY_NAME = (("keypoint", prepare_batch_kp), ("hed", prepare_batch_hed), ("canny", prepare_batch_canny), ("depth", prepare_batch_depth),
                    ("normal", prepare_batch_normal), ("sem", prepare_batch_sem))


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
             starting_noise=None, inpainting_mask=None, z0=None, use_graph=True, no_plms=False, grounding_extra_input=None,
             grounding_input=None):
    """The sampling core of run() (reference gligen_inference.py:389-431) on already-encoded inputs."""
    # reference gligen_inference.py:385-390: DDIM (250 steps) with --no_plms, else PLMS (50 steps)
    sampler_cls = DDIMSampler if no_plms else PLMSSampler
    sampler = sampler_cls(diffusion, model, alpha_generator_func=partial(alpha_generator, type=alpha_type), set_alpha_scale=set_alpha_scale)
    sampler.use_graph = use_graph
    inpainting_extra_input = None
    if inpainting_mask is not None:
        inpainting_extra_input = torch.cat([z0 * inpainting_mask, inpainting_mask], dim=1)
    if grounding_input is None:
        grounding_input = model.grounding_tokenizer_input.prepare(batch)
    input = dict(x=starting_noise, timesteps=None, context=context, grounding_input=grounding_input,
                 inpainting_extra_input=inpainting_extra_input, grounding_extra_input=grounding_extra_input)
    B = context.shape[0]
    shape = (B, model.in_channels, model.image_size, model.image_size)
    latents = sampler.sample(S=steps, shape=shape, input=input, uc=uc, guidance_scale=guidance_scale, mask=inpainting_mask, x0=z0)
    return autoencoder.decode(latents)


def _lane_clone(module):
    """A second execution context for the same weights: a shallow copy of the module (parameters and sub-modules shared)
    whose native engine is a FORK of the module's own (gl_ctx_fork): the packed weights are shared, the arena, conditioning,
    gates, captured hipGraph, stream and the restorable first conv are the lane's. Forking needs the parent's engine, so it
    is built here if it was not yet -- every lane exists before any lane samples."""
    import copy
    parent_engine = module.engine
    m = copy.copy(module)
    m.__dict__ = dict(module.__dict__)
    for key in ("_cond_key", "_cond_held", "_ds_cache", "_lanes"):
        if key in m.__dict__:
            m.__dict__[key] = None
    m.__dict__["_engine"] = parent_engine.fork()
    m.__dict__["_fork_of"] = parent_engine
    return m


def _lanes_of(model, autoencoder, lanes, device):
    """The (UNet clone, VAE clone, stream) triples of lanes 1 .. lanes - 1, cached on the model and dropped with its engine
    (UNetModel._drop_engine: .to() / load_state_dict / a first-conv that no longer matches). A clone made when the model's first
    conv was in another state than now (GLIGEN vs restored SD conv) is stale: its fork carries the old conv."""
    state = (model.first_conv_type, bool(model.__dict__.get("_first_conv_restored")))
    ctxs = model.__dict__.get("_lanes")
    stale = lambda c: (c[3] != state or c[0].__dict__.get("_engine") is None or c[1].__dict__.get("_engine") is None
                       or c[0].__dict__.get("_fork_of") is not model.__dict__.get("_engine")           # the parent engine was rebuilt
                       or c[1].__dict__.get("_fork_of") is not autoencoder.__dict__.get("_engine"))   # (weights reloaded / moved)
    if ctxs is None or any(stale(c) for c in ctxs):
        for c in ctxs or ():
            for m in c[:2]:
                if m.__dict__.get("_engine") is not None:
                    m.__dict__["_engine"].close()
        ctxs = model.__dict__["_lanes"] = []
    while len(ctxs) < lanes - 1:
        ctxs.append((_lane_clone(model), _lane_clone(autoencoder), torch.cuda.Stream(device=device), state))
    return ctxs


@torch.no_grad()
def generate_lanes(model, autoencoder, diffusion, batch, context, uc, *, lanes=2, starting_noise=None, grounding_extra_input=None,
                   grounding_input=None, **kw):
    """generate() with the batch split over `lanes` execution contexts that run concurrently on their own HIP streams, so that
    one half's kernel tails, launch gaps and memory-bound kernels overlap the other half's matrix work (what bench.py's
    --lanes measures: +15 % images/s at 2). Every sample's trajectory is independent, so the images are generate()'s up to the
    rounding of the tile / split-K configurations the GEMMs pick at the sub-batch size (another fp32 summation order).
    The lanes share ONE set of packed weights (engine forks); each has its own arena, conditioning, graph and first-conv copy,
    and every lane's engine exists before the first lane samples -- a lane that swaps the SD first conv in at a gated-off step
    (alpha_type [0.3, 0, 0.7]) changes its own engine and the shared nn.Module, never another lane's packed weights."""
    import copy
    n = context.shape[0]
    if lanes < 2 or n < 2 * lanes:
        return generate(model, autoencoder, diffusion, batch, context, uc, starting_noise=starting_noise,
                        grounding_extra_input=grounding_extra_input, grounding_input=grounding_input, **kw)
    model.engine, autoencoder.engine            # (built before the first fork)
    ctxs = _lanes_of(model, autoencoder, lanes, context.device)
    cuts = [n * i // lanes for i in range(lanes + 1)]
    cut = lambda t, i: t[cuts[i]:cuts[i + 1]] if torch.is_tensor(t) and t.shape[0] == n else t
    outs, main = [], torch.cuda.current_stream(context.device)
    tokenizer = model.grounding_tokenizer_input
    try:
        for i in range(lanes):
            m, ae, stream = (model, autoencoder, main) if i == 0 else ctxs[i - 1][:3]
            sub = {k: cut(v, i) for k, v in batch.items()}
            # the tokenizer input object remembers the batch its null (unconditional) input has to have: one per lane
            m.grounding_tokenizer_input = copy.copy(tokenizer)
            gi = None
            if grounding_input is not None:
                m.grounding_tokenizer_input.prepare(sub)          # shapes of this lane's null input
                gi = {k: cut(v, i) for k, v in grounding_input.items()}
            if i:
                m.first_conv_type = model.first_conv_type
                stream.wait_stream(main)          # the inputs were produced on the caller's stream
            with torch.cuda.stream(stream):
                outs.append(generate(m, ae, diffusion, sub, cut(context, i), cut(uc, i), starting_noise=cut(starting_noise, i),
                                     grounding_extra_input=cut(grounding_extra_input, i), grounding_input=gi, **kw))
    finally:
        model.grounding_tokenizer_input = tokenizer
    for c in ctxs[:lanes - 1]:
        main.wait_stream(c[2])
    # a lane that restored the SD first conv did it for the shared module: every context is in that state now
    if any(c[0].__dict__.get("_first_conv_restored") for c in ctxs[:lanes - 1]) or model.__dict__.get("_first_conv_restored"):
        model.first_conv_type = "SD"
        model.__dict__["_first_conv_restored"] = True
        state = (model.first_conv_type, True)
        model.__dict__["_lanes"] = [(c[0], c[1], c[2], state) for c in ctxs]
        for c in ctxs:
            c[0].first_conv_type = "SD"
            c[0].__dict__["_first_conv_restored"] = True
    return torch.cat(outs, dim=0)


@torch.no_grad()
def generate_stream(model, autoencoder, diffusion, batch, context, uc, noises, *, lanes=2, first_lane=0, **kw):
    """Several WHOLE batches of the same prompt (one starting noise each), issued round-robin to `lanes` execution contexts -- what
    bench.py times: while one batch's evaluation is in its kernel tails and memory-bound kernels, the other's matrix work fills the
    chip. Each batch keeps the benchmark's shape (the tile table, the captured graph and the timed row-local / two-GEMM choice are
    per shape), unlike generate_lanes, which splits ONE batch into halves. Returns the list of decoded batches in issue order.
    Batch i goes to lane (first_lane + i) % lanes."""
    import copy
    lanes = max(1, int(lanes))
    model.engine, autoencoder.engine
    dev = context.device
    main = torch.cuda.current_stream(dev)
    # every lane -- the first too -- runs on a stream of its own that waits ONCE for what the caller's stream had produced when this
    # call began (the inputs); a lane that waited on the caller's stream per batch would queue behind the other lane's whole batch
    s0 = model.__dict__.get("_lane0_stream")
    if s0 is None or s0.device != torch.device(dev):
        s0 = model.__dict__["_lane0_stream"] = torch.cuda.Stream(device=dev)
    ctxs = [(model, autoencoder, s0)] + [c[:3] for c in _lanes_of(model, autoencoder, lanes, dev)[:lanes - 1]]
    start = main.record_event()
    for c in ctxs:
        c[2].wait_event(start)
    tokenizer = model.grounding_tokenizer_input
    outs = []
    try:
        for i, x_T in enumerate(noises):
            m, ae, stream = ctxs[(first_lane + i) % lanes]
            if m is not model:
                m.grounding_tokenizer_input = copy.copy(tokenizer)
                m.first_conv_type = model.first_conv_type
            with torch.cuda.stream(stream):
                outs.append(generate(m, ae, diffusion, batch, context, uc, starting_noise=x_T, **kw))
    finally:
        model.grounding_tokenizer_input = tokenizer
    for c in ctxs:
        main.wait_stream(c[2])
    return outs


def save_images(samples, output_folder, first_id=None, ids=None):
    os.makedirs(output_folder, exist_ok=True)
    if ids is None:
        start = len(os.listdir(output_folder)) if first_id is None else first_id
        ids = list(range(start, start + samples.shape[0]))
    assert len(ids) == samples.shape[0]
    print(ids)
    for image_id, sample in zip(ids, samples):
        sample = torch.clamp(sample, min=-1, max=1) * 0.5 + 0.5
        arr = sample.cpu().numpy().transpose(1, 2, 0) * 255
        Image.fromarray(arr.astype(np.uint8)).save(os.path.join(output_folder, str(int(image_id)) + ".png"))


def _shard(t, lo, hi):
    return t[lo:hi] if torch.is_tensor(t) else t


@torch.no_grad()
def run(meta, config, starting_noise=None, models=None):
    """config: argparse namespace / dict with batch_size, guidance_scale, negative_prompt, no_plms, folder (+ optional
    steps, seed). Under torch.distributed (WORLD_SIZE > 1) batch_size is the GLOBAL batch: every rank samples its
    contiguous slice (gligen_amd.dist.shard_range) with replicated weights and writes its own images."""
    from gligen_amd import dist as gdist
    args = dict(config) if isinstance(config, dict) else vars(config)
    if models is None:
        model, autoencoder, text_encoder, diffusion, ckpt_config = load_ckpt(meta["ckpt"])
    else:
        model, autoencoder, text_encoder, diffusion, ckpt_config = models
    model.grounding_tokenizer_input = instantiate_from_config(ckpt_config["grounding_tokenizer_input"])
    grounding_downsampler_input = None
    if "grounding_downsampler_input" in ckpt_config:
        grounding_downsampler_input = instantiate_from_config(ckpt_config["grounding_downsampler_input"])
    B = args["batch_size"]
    rank, _, world = gdist.env_world()
    lo, hi = gdist.shard_range(B, rank, world)
    folder = os.path.join(args["folder"], meta["save_folder_name"])
    if hi == lo:
        # more ranks than samples (e.g. the CLI's default batch_size 5 on 8 GPUs): this rank has nothing to sample, but it
        # still joins the two barriers the other ranks wait in before they number their images
        os.makedirs(folder, exist_ok=True)
        gdist.barrier()
        gdist.barrier()
        return torch.empty((0, 3, 8 * model.image_size, 8 * model.image_size))
    prepare = next((fn for key, fn in _PREPARE_BY_NAME if key in meta["ckpt"]), prepare_batch)
    batch = {k: _shard(v, lo, hi) for k, v in prepare(meta, B).items()}
    if "grounding_tokens" in meta:   # spatial-map modalities: ConvNeXt tokens computed elsewhere (like precomputed CLIP features)
        batch["tokens"] = _shard(meta["grounding_tokens"].to(device), lo, hi)
    if "context" in meta:  # precomputed CLIP last_hidden_state (B,77,768)
        context, uc = meta["context"].to(device)[lo:hi], meta["uc"].to(device)[lo:hi]
    else:
        context = text_encoder.encode([meta["prompt"]] * (hi - lo))
        uc = text_encoder.encode((hi - lo) * [args.get("negative_prompt") or ""])
    if world > 1 and args.get("seed") is None:
        # no --seed on a sharded run: the reference draws a fresh x_T per invocation, and the ranks must still slice ONE draw --
        # rank 0 picks this run's seed and every rank takes it (without a process group: the old fixed seed 0)
        import torch.distributed as tdist
        if tdist.is_initialized():
            args["seed"] = gdist.broadcast_int(int(torch.seed() % (1 << 31)))
    if world > 1 or args.get("seed") is not None:
        # one seeded draw for the whole batch, sliced: an N-GPU run reproduces the 1-GPU images. The device generator is
        # seeded too, identically on every rank: the inpainting loop's per-step q_sample noise (randn_like(z0), batch 1 for
        # the single encoded input image) comes from it, so that noise is the same on all ranks and run to run. (With a
        # per-sample z0 batch that draw is batch-shaped and a sharded run is reproducible but not equal to the 1-GPU run.)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(int(args.get("seed") or 0))
        if starting_noise is None:
            gen = torch.Generator().manual_seed(int(args.get("seed") or 0))
            starting_noise = torch.randn((B, model.in_channels, model.image_size, model.image_size), generator=gen)
    if starting_noise is not None:
        starting_noise = starting_noise.to(device)[lo:hi] if starting_noise.shape[0] == B else starting_noise.to(device)
    mask = z0 = None
    if "input_image" in meta:
        assert ckpt_config.get("inpaint_mode"), "input_image is given, the ckpt must be the inpaint model, are you using the correct ckpt?"
        mask = draw_masks_from_boxes(batch["boxes"], model.image_size).to(device)
        if "z0" in meta:
            z0 = meta["z0"].to(device)
        else:
            img = torch.from_numpy(np.asarray(Image.open(meta["input_image"]).convert("RGB").resize((512, 512)))).permute(2, 0, 1)
            z0 = autoencoder.encode((img.float().unsqueeze(0).to(device) / 255 - 0.5) / 0.5)
    grounding_extra_input = None
    if grounding_downsampler_input is not None:
        grounding_extra_input = grounding_downsampler_input.prepare(batch)
    grounding_input = None
    if "tokens" in batch:
        model.grounding_tokenizer_input.prepare(batch)   # remembers the shapes its null input needs
        grounding_input = {"tokens": batch["tokens"]}
    no_plms = bool(args.get("no_plms"))
    steps = int(args.get("steps") or (250 if no_plms else 50))
    # batches of 32 and more run as two half-batches in flight (generate_lanes); below that one evaluation over the whole batch is the
    # faster schedule (measured, profiles/r6/batch_sweep.txt: 8 images 6.69 images/s as one batch against 6.24 as two halves, 16 images
    # 7.08 against 6.91, 32 images 7.11 against 7.19). Not for inpainting: its per-step q_sample noise comes from the device generator,
    # whose draws would interleave differently
    lanes = args.get("lanes")
    lanes = 2 if lanes is None else min(2, max(1, int(lanes)))      # (--lanes 0 and 1 both mean one batch at a time; one batch is split in two at most)
    if mask is not None or (hi - lo) < SPLIT_BATCH_AT:
        lanes = 1
    if lanes > 1 and starting_noise is None:   # x_T as the sampler would draw it (plms.py:71), before the batch is split
        starting_noise = torch.randn((hi - lo, model.in_channels, model.image_size, model.image_size), device=device)
    repeat = max(1, int(args.get("repeat") or 1))
    n_lanes_req = 3 if args.get("lanes") is None else max(1, int(args.get("lanes")))    # whole batches in flight (--repeat): 3, as bench.py
    if repeat > 1 and mask is None and meta.get("alpha_type") in (None, [1, 0, 0], [1.0, 0.0, 0.0]):
        # `repeat` batches of this prompt (seed, seed + 1, ...): whole batches in flight on the lanes, as bench.py runs them
        gkw = dict(steps=steps, guidance_scale=args["guidance_scale"], alpha_type=meta.get("alpha_type"), no_plms=no_plms,
                   grounding_extra_input=grounding_extra_input, grounding_input=grounding_input)
        shape = (hi - lo, model.in_channels, model.image_size, model.image_size)
        seed0 = int(args.get("seed") or 0)
        seeded = world > 1 or args.get("seed") is not None
        def noise(r):
            if r == 0 and starting_noise is not None:
                return starting_noise
            if not seeded:      # no --seed on one GPU: fresh x_T per batch, as the reference's sampler draws it (plms.py:71)
                return torch.randn(shape, device=device)
            # seeded (or sharded: every rank must slice the SAME global draw): batch r of the run is seed + r
            return torch.randn((B,) + shape[1:], generator=torch.Generator().manual_seed(seed0 + r))[lo:hi].to(device)
        n_lanes_req = min(n_lanes_req, repeat)
        for _ in range(max(0, int(args.get("warmup") or 0))):      # one lane at a time: tile tuning and the timed kernel choices want the chip to themselves
            for ln in range(n_lanes_req):
                generate_stream(model, autoencoder, diffusion, batch, context, uc, [noise(0)], lanes=n_lanes_req, first_lane=ln, **gkw)
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = generate_stream(model, autoencoder, diffusion, batch, context, uc, [noise(r) for r in range(repeat)], lanes=n_lanes_req, **gkw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        samples = torch.cat(outs, dim=0)
        print(f"[gligen_amd] {samples.shape[0]} images in {dt:.3f} s = {samples.shape[0] / dt:.3f} images/s on this GPU "
              f"(sampling + decode of {repeat} batches of {hi - lo}, {min(n_lanes_req, repeat)} in flight)", flush=True)
    else:
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        t0 = time.perf_counter()
        samples = generate_lanes(model, autoencoder, diffusion, batch, context, uc, lanes=lanes, steps=steps,
                                 guidance_scale=args["guidance_scale"], alpha_type=meta.get("alpha_type"), starting_noise=starting_noise,
                                 inpainting_mask=mask, z0=z0, no_plms=no_plms, grounding_extra_input=grounding_extra_input,
                                 grounding_input=grounding_input)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f"[gligen_amd] {samples.shape[0]} images in {dt:.3f} s = {samples.shape[0] / dt:.3f} images/s on this GPU "
                  f"(sampling + decode, first call of a shape includes tile tuning and graph capture; {lanes} sub-batch(es) in flight)", flush=True)
    if world > 1:
        os.makedirs(folder, exist_ok=True)
        gdist.barrier()
        start = len(os.listdir(folder))     # every rank counts before any rank writes
        gdist.barrier()
        # image id = position in the GLOBAL run: batch r of `--repeat` occupies [start + r B, start + (r + 1) B), this rank's slice of it
        # [lo, hi) -- ranks never overlap, and the numbering is that of a 1-GPU run of the same command
        n_rounds = samples.shape[0] // (hi - lo)
        save_images(samples, folder, ids=[start + r * B + lo + i for r in range(n_rounds) for i in range(hi - lo)])
    else:
        save_images(samples, folder)
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


# The reference's demo prompts (gligen_inference.py:466-637), same fields; checkpoints are looked up where the reference
# expects them (../gligen_checkpoints/...). The spatial-map entries additionally need ConvNeXt grounding tokens.
KEYPOINTS_COCO_18150 = [
    [[0.7598, 0.2542], [0.7431, 0.2104], [0.8118, 0.2021], [0.0, 0.0], [0.9514, 0.1813], [0.7806, 0.2917], [0.0, 0.0], [0.6785, 0.5125],
     [0.0, 0.0], [0.5389, 0.6479], [0.6785, 0.6750], [0.7973, 0.7042], [0.0, 0.0], [0.6181, 0.7375], [0.9764, 0.8458], [0.0, 0.0], [0.0, 0.0]],
    [[0.2681, 0.4313], [0.2514, 0.3979], [0.0, 0.0], [0.0785, 0.3854], [0.0, 0.0], [0.0910, 0.5583], [0.0, 0.0], [0.1243, 0.8479],
     [0.0, 0.0], [0.0, 0.0], [0.0, 0.0], [0.0, 0.0], [0.0, 0.0], [0.2410, 0.8146], [0.1202, 0.6146], [0.0, 0.0], [0.2743, 0.7188]],
]
CKPT_DIR = "../gligen_checkpoints"
meta_list = [
    dict(ckpt=f"{CKPT_DIR}/checkpoint_generation_text.pth", prompt="a teddy bear sitting next to a bird", phrases=["a teddy bear", "a bird"],
         locations=[[0.0, 0.09, 0.33, 0.76], [0.55, 0.11, 1.0, 0.8]], alpha_type=[0.3, 0.0, 0.7], save_folder_name="generation_box_text"),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_inpainting_text.pth", input_image="inference_images/dalle2_museum.jpg", prompt="a corgi and a cake",
         phrases=["corgi", "cake"], locations=[[0.25, 0.28, 0.42, 0.52], [0.14, 0.58, 0.58, 0.92]], save_folder_name="inpainting_box_text"),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_generation_text_image.pth", prompt="an alarm clock sitting on the beach", images=["inference_images/clock.png"],
         phrases=["alarm clock"], locations=[[0.0, 0.09, 0.53, 0.76]], alpha_type=[1.0, 0.0, 0.0], save_folder_name="generation_box_image"),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_generation_text_image.pth", prompt="a brick house in the woods, anime, oil painting",
         phrases=["a brick house", "placehoder"], images=["inference_images/placeholder.png", "inference_images/style_golden.jpg"],
         locations=[[0.4, 0.2, 1.0, 0.8], [0.0, 1.0, 0.0, 1.0]], alpha_type=[1, 0, 0], text_mask=[1, 0], image_mask=[0, 1],
         save_folder_name="generation_box_text_style"),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_inpainting_text_image.pth", input_image="inference_images/beach.jpg", prompt="a bigben on the beach",
         images=["inference_images/bigben.jpg"], locations=[[0.18, 0.08, 0.62, 0.75]], save_folder_name="inpainting_box_image"),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_generation_hed.pth", prompt="a man is eating breakfast", hed_image="inference_images/hed_man_eat.png",
         save_folder_name="hed", alpha_type=[0.9, 0, 0.1]),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_generation_canny.pth", prompt="A Humanoid Robot Designed for Companionship",
         canny_image="inference_images/canny_robot.png", alpha_type=[0.9, 0, 0.1], save_folder_name="canny"),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_generation_normal.pth", prompt="a large tree with no leaves in front of a building",
         normal="inference_images/normal_tree_building.jpg", alpha_type=[0.7, 0, 0.3], save_folder_name="normal"),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_generation_depth.pth", prompt="a Vibrant colorful Bird Sitting on Tree Branch",
         depth="inference_images/depth_bird.png", alpha_type=[0.7, 0, 0.3], save_folder_name="depth"),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_generation_sem.pth", prompt="a living room filled with lots of furniture and plants",
         sem="inference_images/sem_ade_living_room.png", alpha_type=[0.7, 0, 0.3], save_folder_name="sem"),
    dict(ckpt=f"{CKPT_DIR}/checkpoint_generation_keypoint.pth", prompt="A young man and a small boy are talking", locations=KEYPOINTS_COCO_18150,
         alpha_type=[0.3, 0.0, 0.7], save_folder_name="keypoint"),
]


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--folder", type=str, default="generation_samples", help="root folder for output")
    parser.add_argument("--batch_size", type=int, default=5, help="images per prompt (the GLOBAL batch under torch.distributed.run)")
    parser.add_argument("--no_plms", action="store_true", help="use DDIM instead. WARNING: I did not test the code yet")
    parser.add_argument("--guidance_scale", type=float, default=7.5, help="")
    parser.add_argument("--negative_prompt", type=str,
                        default="longbody, lowres, bad anatomy, bad hands, missing fingers, extra digit, fewer digits, cropped, worst quality, low quality", help="")
    parser.add_argument("--synthetic", type=str, default=None, choices=["text", "text_image", "keypoint"],
                        help="run with seeded random weights and features (no checkpoint / CLIP needed)")
    parser.add_argument("--inpaint", action="store_true", help="with --synthetic text: the inpainting model (9-channel first conv, encode + blend)")
    parser.add_argument("--ckpt", type=str, default=None, help="run only the meta_list entries whose checkpoint path contains this string")
    parser.add_argument("--seed", type=int, default=None, help="seed of x_T (one draw for the whole batch, sliced across ranks)")
    parser.add_argument("--lanes", type=int, default=None, help="with --repeat: whole batches in flight (default 3, the bench's schedule); without: per-GPU batches of 32 and more run as two half-batches in flight unless this is 1")
    parser.add_argument("--steps", type=int, default=None, help="override the sampler's step count (reference: 50 PLMS / 250 DDIM)")
    parser.add_argument("--repeat", type=int, default=1, help="batches of --batch_size per prompt (seed, seed + 1, ...): whole batches run --lanes at a time, as bench.py times them")
    parser.add_argument("--warmup", type=int, default=0, help="with --repeat: untimed rounds first (tile tuning, graph capture), so the printed images/s is the steady state")
    args = parser.parse_args(argv)

    # one process per GPU: `python -m torch.distributed.run --nproc-per-node N gligen_inference.py ...` shards --batch_size
    from gligen_amd import dist as gdist
    global device
    rank, local_rank, world = gdist.init_from_env()
    if world > 1:
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(device)
        if args.seed is None:
            args.seed = 0
    if args.synthetic:
        model, autoencoder, diffusion, cfg = load_synthetic(args.synthetic, inpaint=args.inpaint)
        meta = _synthetic_meta(args.synthetic, args.batch_size)
        if args.inpaint:
            from gligen_amd import synthetic as syn
            meta.update(input_image="synthetic", z0=autoencoder.encode(torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(8)).to(device) * 2 - 1))
        run(meta, args, models=(model, autoencoder, None, diffusion, cfg))
    else:
        for meta in meta_list:   # the reference runs every entry (gligen_inference.py:640-642)
            if args.ckpt is None or args.ckpt in meta["ckpt"]:
                run(meta, args)
    gdist.shutdown()


if __name__ == "__main__":
    main()
