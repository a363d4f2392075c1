"""Text encoder front-end (reference ldm/modules/encoders/modules.py:144-173): CLIP ViT-L/14 text
tower from HF transformers, last_hidden_state (B,77,768). It runs once per prompt and is not part
of the MI355X hot path; weights come from the GLIGEN checkpoint (load_ckpt) and the tokenizer from
the local HF cache."""
import torch
import torch.nn as nn


class AbstractEncoder(nn.Module):
    def encode(self, *args, **kwargs):
        raise NotImplementedError


class FrozenCLIPEmbedder(AbstractEncoder):
    """Uses the CLIP transformer encoder for text (from Hugging Face)."""

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77):
        super().__init__()
        from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
        try:
            self.tokenizer = CLIPTokenizer.from_pretrained(version)
            self.transformer = CLIPTextModel.from_pretrained(version)
        except Exception:  # offline: architecture from constants, weights arrive via load_state_dict
            self.tokenizer = None
            cfg = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                 num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=768)
            self.transformer = CLIPTextModel(cfg)
        self.device = device
        self.max_length = max_length
        self.freeze()

    def load_state_dict(self, state_dict, strict=True, **kw):
        """GLIGEN checkpoints were written under transformers 4.x, whose CLIPTextModel wraps the tower in `.text_model`
        (keys `transformer.text_model.embeddings...`); transformers 5.x dropped that level (`transformer.embeddings...`). Keys are
        renamed to whatever the installed class uses, so an existing checkpoint loads unchanged under either version."""
        own = self.state_dict().keys()
        wrapped_here = any(k.startswith("transformer.text_model.") for k in own)
        out = {}
        for k, v in state_dict.items():
            if k.endswith("position_ids") and k not in own:
                continue                          # a registered buffer in old versions only
            if wrapped_here and k.startswith("transformer.") and not k.startswith("transformer.text_model."):
                k = "transformer.text_model." + k[len("transformer."):]
            elif not wrapped_here and k.startswith("transformer.text_model."):
                k = "transformer." + k[len("transformer.text_model."):]
            out[k] = v
        return super().load_state_dict(out, strict=strict, **kw)

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, text, return_pooler_output=False):
        if self.tokenizer is None:
            raise RuntimeError("CLIP tokenizer files are not available offline; pass precomputed context embeddings instead")
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        out = self.transformer(input_ids=enc["input_ids"].to(self.device))
        if return_pooler_output:
            return out.last_hidden_state, out.pooler_output
        return out.last_hidden_state

    def encode(self, text, return_pooler_output=False):
        return self(text, return_pooler_output)
