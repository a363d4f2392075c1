"""Wall time of one training iteration (gl_unet_train_step: forward + loss + backward) of the correctness path, per configuration:
small UNet at a 16 x 16 latent, the shipped topology at 16 x 16 and at the real 64 x 64 latent. Synthetic inputs, seeded weights.
   PYTHONPATH=. python tools/train_bench.py [--full64]"""
import json
import sys
import time

import torch

from gligen_amd import synthetic as syn
from gligen_amd.engine import Engine


def run(name, cfg, B, hw, reps, eng, checkpoint=False, cache=False):
    model_shapes = None
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    m = UNetModel(**dict(cfg, grounding_tokenizer=syn.GROUNDING_TOKENIZERS["text"], inpaint_mode=False))
    sd = {k: v.float().to(eng.device).contiguous() for k, v in syn.seeded_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 1234).items()}
    del m
    b = syn.make_batch("text", B, n_valid=3, seed=5)
    batch = dict(x=syn.make_latent(B, 4, hw, hw, seed=6), timesteps=torch.tensor([981, 441, 300, 77][:B]).float(), context=syn.make_context(B, seed=6),
                 boxes=b["boxes"], masks=b["masks"], positive_embeddings=b["text_embeddings"], target=syn.make_latent(B, 4, hw, hw, seed=7))
    grads = {k: torch.zeros_like(v) for k, v in sd.items() if ".fuser." in k or k.startswith("position_net.")}
    eng = Engine(0, arena_gb=160.0)                      # a fresh arena per line: its high water is this configuration's
    if cache:
        eng.train_weight_cache(True)                     # the frozen parameters' bf16 operand copies kept across iterations (gl_train_weight_cache)
    eng.unet_train_step(cfg, sd, batch, grads=grads, checkpoint=checkpoint, use_weight_cache=cache)     # warm-up (GEMM tile selection; fills the cache)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        loss, _, _ = eng.unet_train_step(cfg, sd, batch, grads=grads, checkpoint=checkpoint, use_weight_cache=cache)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    print(json.dumps(dict(config=name, B=B, latent=hw, checkpoint=bool(checkpoint), weight_cache=bool(cache), weight_cache_gb=round(eng.train_weight_cache(True) / 2 ** 30, 2) if cache else 0,
                          s_per_iteration=round(dt, 4), loss=float(loss), arena_high_water_gb=round(eng.arena_high_water() / 2 ** 30, 2),
                          trainable_values=sum(int(g.numel()) for g in grads.values()))), flush=True)
    if cache:
        eng.train_weight_cache(False)


if __name__ == "__main__":
    eng = Engine(0, arena_gb=160.0)
    if "--b4only" in sys.argv:          # the profiled line (tools/gpu_run.sh train with TRAIN_PROF=1): the bench line's train_step shape
        run("shipped topology", syn.UNET_CFG, 4, 64, 2, eng, checkpoint=True)
        run("shipped topology", syn.UNET_CFG, 4, 64, 2, eng, checkpoint=True, cache=True)
        sys.exit(0)
    run("small UNet", syn.UNET_CFG_SMALL, 2, 16, 3, eng)
    run("shipped topology", syn.UNET_CFG, 1, 16, 2, eng)
    if "--full64" in sys.argv:
        run("shipped topology", syn.UNET_CFG, 1, 64, 1, eng)
        run("shipped topology", syn.UNET_CFG, 1, 64, 1, eng, checkpoint=True)
        run("shipped topology", syn.UNET_CFG, 4, 64, 1, eng, checkpoint=True)
