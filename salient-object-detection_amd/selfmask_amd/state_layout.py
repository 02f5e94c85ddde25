"""The 267-tensor ``state_dict`` contract of the SelfMask MaskFormer and a synthetic checkpoint generator.

The key names / shapes are the drop-in contract with the reference so that ``selfmask_nq20.pt`` loads unchanged
(reference: networks/maskformer/maskformer.py:12-72, networks/vision_transformer.py:191-258,
networks/maskformer/transformer_decoder.py:229-258).

No dataset or checkpoint exists offline, so golden vectors, tests and the bench use a *synthetic* checkpoint filled
from ``numpy.random.Generator(PCG64(seed))`` (bit-stable across numpy versions) in key order.
"""
from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np

EMBED_DIM = 384
N_HEADS = 6
HEAD_DIM = 64
MLP_HIDDEN = 1536
ENC_DEPTH = 12
TRAINED_IMG = 224  # PatchEmbed is always built for 224x224 (vision_transformer.py:213-218)


def state_shapes(
        n_queries: int = 20,
        patch_size: int = 16,
        n_decoder_layers: int = 6,
        use_binary_classifier: bool = True,
) -> "OrderedDict[str, Tuple[int, ...]]":
    """Key -> shape in the exact ``nn.Module.state_dict()`` order of the reference MaskFormer."""
    D, H = EMBED_DIM, MLP_HIDDEN
    n0 = (TRAINED_IMG // patch_size) ** 2
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["query_embed"] = (n_queries, D)
    s["encoder.cls_token"] = (1, 1, D)
    s["encoder.pos_embed"] = (1, n0 + 1, D)
    s["encoder.patch_embed.proj.weight"] = (D, 3, patch_size, patch_size)
    s["encoder.patch_embed.proj.bias"] = (D,)
    for i in range(ENC_DEPTH):
        p = f"encoder.blocks.{i}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (H, D)
        s[p + "mlp.fc1.bias"] = (H,)
        s[p + "mlp.fc2.weight"] = (D, H)
        s[p + "mlp.fc2.bias"] = (D,)
    s["encoder.norm.weight"] = (D,)
    s["encoder.norm.bias"] = (D,)
    for j in range(n_decoder_layers):
        p = f"decoder.layers.{j}."
        for attn in ("self_attn", "multihead_attn"):
            s[p + attn + ".in_proj_weight"] = (3 * D, D)
            s[p + attn + ".in_proj_bias"] = (3 * D,)
            s[p + attn + ".out_proj.weight"] = (D, D)
            s[p + attn + ".out_proj.bias"] = (D,)
        s[p + "linear1.weight"] = (H, D)
        s[p + "linear1.bias"] = (H,)
        s[p + "linear2.weight"] = (D, H)
        s[p + "linear2.bias"] = (D,)
        for n in ("norm1", "norm2", "norm3"):
            s[p + n + ".weight"] = (D,)
            s[p + n + ".bias"] = (D,)
    s["decoder.norm.weight"] = (D,)
    s["decoder.norm.bias"] = (D,)
    if use_binary_classifier:
        s["ffn.layers.0.weight"] = (D, D)
        s["ffn.layers.0.bias"] = (D,)
        s["ffn.layers.1.weight"] = (D, D)
        s["ffn.layers.1.bias"] = (D,)
        s["ffn.layers.2.weight"] = (1, D)
        s["ffn.layers.2.bias"] = (1,)
    else:
        s["ffn.layers.0.weight"] = (D, D)
        s["ffn.layers.0.bias"] = (D,)
        s["ffn.layers.1.weight"] = (D, D)
        s["ffn.layers.1.bias"] = (D,)
        s["ffn.layers.2.weight"] = (D, D)
        s["ffn.layers.2.bias"] = (D,)
        s["linear_classifier.weight"] = (2, D)
        s["linear_classifier.bias"] = (2,)
        s["norm.weight"] = (D,)
        s["norm.bias"] = (D,)
    return s


# per-family fill rule: (kind, scale)
_STYLES = {
    # "soft": DINO-like init scales -> near-uniform softmax, moderate logits
    "soft": dict(linear=0.02, attn_in=0.02, bias=0.02, ln_w=0.1, ln_b=0.1, query=1.0, pos=0.02, patch=0.02),
    # "peaky": larger attention projections -> softmax far from uniform, some sigmoids saturate
    "peaky": dict(linear=0.03, attn_in=0.08, bias=0.05, ln_w=0.1, ln_b=0.1, query=1.0, pos=0.05, patch=0.03),
    # "calib": as "soft" but the shared decoder.norm gain is 0.25 so mask logits stay within ~+-16 (the range a
    # sigmoid-trained model works in); the fp32 noise floor there is ~2e-5, which makes the absolute 1e-4 logit
    # gate meaningful (with unit gain the fp32 reference itself sits 0.8-1.6e-4 from its own fp64 evaluation).
    "calib": dict(linear=0.02, attn_in=0.04, bias=0.02, ln_w=0.1, ln_b=0.1, query=1.0, pos=0.02, patch=0.02,
                  dec_norm_gain=0.25),
}


def synthetic_state_dict_numpy(
        seed: int = 0,
        style: str = "soft",
        n_queries: int = 20,
        patch_size: int = 16,
        n_decoder_layers: int = 6,
        use_binary_classifier: bool = True,
) -> "OrderedDict[str, np.ndarray]":
    """Deterministic fp32 weights for every key, generated in key order from one PCG64 stream."""
    st = _STYLES[style]
    rng = np.random.Generator(np.random.PCG64(seed))
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, shp in state_shapes(n_queries, patch_size, n_decoder_layers, use_binary_classifier).items():
        z = rng.standard_normal(shp)
        if k == "query_embed":
            v = st["query"] * z
        elif k.endswith("cls_token") or k.endswith("pos_embed"):
            v = st["pos"] * z
        elif "patch_embed.proj.weight" in k:
            v = st["patch"] * z
        elif k == "decoder.norm.weight":
            v = st.get("dec_norm_gain", 1.0) * (1.0 + st["ln_w"] * z)
        elif k == "decoder.norm.bias":
            v = st.get("dec_norm_gain", 1.0) * st["ln_b"] * z
        elif ("norm" in k.split(".")[-2]) and k.endswith(".weight"):
            v = 1.0 + st["ln_w"] * z
        elif ("norm" in k.split(".")[-2]) and k.endswith(".bias"):
            v = st["ln_b"] * z
        elif k.endswith("in_proj_weight") or k.endswith("attn.qkv.weight"):
            v = st["attn_in"] * z
        elif k.endswith("weight"):
            v = st["linear"] * z
        else:  # biases (incl. in_proj_bias)
            v = st["bias"] * z
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def synthetic_state_dict(seed: int = 0, style: str = "soft", **kw) -> "Dict[str, 'torch.Tensor']":
    import torch
    return OrderedDict((k, torch.from_numpy(v)) for k, v in synthetic_state_dict_numpy(seed, style, **kw).items())


def synthetic_images(seed: int, shape) -> np.ndarray:
    """Normalised-image-like input: N(0,1) fp32, the distribution torchvision Normalize() produces roughly."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal(shape).astype(np.float32)
