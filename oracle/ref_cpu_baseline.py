"""CPU baseline with the REFERENCE's own modules (test / measurement infrastructure, never on the product path).

bench.py runs this in a subprocess where /root/reference is mounted (the build container; the GPU box has no copy and
falls back to the oracle port): 1 warm-up + 3 timed UNetModel.forward (B = 1, 64 x 64 latent, fp32, all host cores)
and 1 AutoencoderKL.decode, as SURVEY.md §8(d) specifies, extrapolated to 102 forwards + 1 decode per 512 x 512 image.
Weights are the modules' default initialisation (timing does not depend on their values). Prints ONE JSON line.

    cd /tmp && python /root/repo/oracle/ref_cpu_baseline.py [--kind text|text_image|keypoint]
"""
import argparse
import importlib.util
import json
import os
import sys
import time

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path = [p for p in sys.path if os.path.abspath(p or ".") != REPO]
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import torch  # noqa: E402


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="text", choices=["text", "text_image", "keypoint"])
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    syn = _load("gl_synthetic", os.path.join(REPO, "gligen_amd", "synthetic.py"))
    from ldm.models.autoencoder import AutoencoderKL
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    from ldm.util import instantiate_from_config
    assert sys.modules["ldm.util"].__file__.startswith(REF), "must import the reference's ldm package"
    ginput = {"text": "grounding_input.text_grounding_tokinzer_input.GroundingNetInput",
              "text_image": "grounding_input.text_image_grounding_tokinzer_input.GroundingNetInput",
              "keypoint": "grounding_input.keypoint_grounding_tokinzer_input.GroundingNetInput"}[args.kind]
    torch.manual_seed(0)
    model = UNetModel(**dict(syn.UNET_CFG, grounding_tokenizer=syn.GROUNDING_TOKENIZERS[args.kind])).eval()
    gin = instantiate_from_config(dict(target=ginput))
    g = gin.prepare(syn.make_batch(args.kind, 1, n_valid=8))
    inp = dict(x=syn.make_latent(1, 4, 64, 64), timesteps=torch.tensor([501]), context=syn.make_context(1), grounding_input=g,
               inpainting_extra_input=None, grounding_extra_input=None)
    ts = []
    with torch.no_grad():
        for _ in range(1 + args.reps):
            t0 = time.perf_counter()
            model(inp)
            ts.append(time.perf_counter() - t0)
        del model
        ae = AutoencoderKL(ddconfig=syn.VAE_DDCONFIG, embed_dim=4, scale_factor=0.18215).eval()
        z = syn.make_latent(1, 4, 64, 64)
        t0 = time.perf_counter()
        ae.decode(z)
        t_dec = time.perf_counter() - t0
    t_unet = sum(ts[1:]) / args.reps
    print(json.dumps({"value": 1.0 / (102 * t_unet + t_dec), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "reference",
                      "sample": f"the reference's UNetModel ({args.kind} tokenizer) and AutoencoderKL imported from /root/reference, fp32 torch-CPU: "
                                f"1 warm-up + {args.reps} timed forwards (B=1, 64x64 latent; mean {t_unet:.2f} s, warm-up {ts[0]:.2f} s) + 1 decode "
                                f"({t_dec:.2f} s), extrapolated to 102 forwards + 1 decode per 512x512 image",
                      "host_cpus": os.cpu_count()}))


if __name__ == "__main__":
    main()
