"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (unidepth_amd/).

Synthetic, *sensitised* UniDepthV2 checkpoints built from a config dict alone, so they can be
regenerated bit-identically on the GPU box (where /root/reference does not exist) from a seed.

Why sensitised: with the reference's default init (trunc_normal std=0.02, LayerScale 1.0) the final
depth is almost input independent, so parity on it proves nothing.  Here every
Linear/Conv weight is randn * fan_in^-0.5, biases randn * 0.1, pos_embed/cls randn * 0.5 (SURVEY
recipe S1) and, in addition, all LayerNorm affines, LayerScale gammas, RCU gammas and latents_pos are
randomised so that weight-folding bugs in the engine's repacker cannot hide.

The key set / shapes restate the reference's state_dict layout
(unidepth/models/unidepthv2/unidepthv2.py:418-460 build(), decoder.py:468-524 Decoder.build(),
backbones/dinov2.py:115-264) and are checked against the real reference with strict=True in
tests/test_oracle_pins.py (runs only where /root/reference exists).
"""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))

ARCH = {
    # name: (embed_dim, depth, heads, default output_idx)   -- backbones/dinov2.py:388-427, encoder.py:139-193
    "dinov2_vits14": (384, 12, 6, [3, 6, 9, 12]),
    "dinov2_vitb14": (768, 12, 12, [3, 6, 9, 12]),
    "dinov2_vitl14": (1024, 24, 16, [5, 12, 18, 24]),
}


def load_config(name: str) -> dict:
    """name in {vits14, vitb14, vitl14}; configs are our own compact restatement of the model section of
    the reference's configs/config_v2_<name>.json (only the keys the inference path reads)."""
    with open(os.path.join(_HERE, "configs", f"config_v2_{name}.json")) as f:
        return json.load(f)


def arch_of(config: dict):
    enc = config["model"]["pixel_encoder"]
    D, depth, heads, out_idx = ARCH[enc["name"]]
    out_idx = enc.get("output_idx", out_idx)
    C = config["model"]["pixel_decoder"]["hidden_dim"]
    return dict(D=D, depth=depth, heads=heads, output_idx=list(out_idx), C=C,
                dec_heads=config["model"]["num_heads"], expansion=config["model"]["expansion"],
                out_dim=config["model"]["pixel_decoder"]["out_dim"],
                dec_depths=list(config["model"]["pixel_decoder"]["depths"]),
                kernel_size=config["model"]["pixel_decoder"].get("kernel_size", 7))


def v2_param_shapes(config: dict) -> "OrderedDict[str, tuple]":
    a = arch_of(config)
    D, C, E = a["D"], a["C"], a["expansion"]
    ks = a["kernel_size"]
    s: "OrderedDict[str, tuple]" = OrderedDict()
    pe = "pixel_encoder."
    s[pe + "cls_token"] = (1, 1, D)
    s[pe + "pos_embed"] = (1, 37 * 37 + 1, D)
    s[pe + "register_tokens"] = (1, 1, D)
    s[pe + "mask_token"] = (1, D)
    s[pe + "patch_embed.proj.weight"] = (D, 3, 14, 14)
    s[pe + "patch_embed.proj.bias"] = (D,)
    for i in range(a["depth"]):
        b = f"{pe}blocks.{i}."
        s[b + "norm1.weight"] = (D,); s[b + "norm1.bias"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D); s[b + "attn.qkv.bias"] = (3 * D,)
        s[b + "attn.proj.weight"] = (D, D); s[b + "attn.proj.bias"] = (D,)
        s[b + "ls1.gamma"] = (D,)
        s[b + "norm2.weight"] = (D,); s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (4 * D, D); s[b + "mlp.fc1.bias"] = (4 * D,)
        s[b + "mlp.fc2.weight"] = (D, 4 * D); s[b + "mlp.fc2.bias"] = (D,)
        s[b + "ls2.gamma"] = (D,)
    s[pe + "norm.weight"] = (D,); s[pe + "norm.bias"] = (D,)

    pd = "pixel_decoder."
    s[pd + "level_embeds"] = (1, 1, 4, C)
    for grp in ("input_adapter", "camera_token_adapter"):
        for j in range(4):
            s[f"{pd}{grp}.input_adapters.{j}.weight"] = (C, D)
            s[f"{pd}{grp}.input_adapters.{j}.bias"] = (C,)

    def mlp(prefix, cin, exp, cout):
        s[prefix + "norm.weight"] = (cin,); s[prefix + "norm.bias"] = (cin,)
        s[prefix + "proj1.weight"] = (cin * exp, cin); s[prefix + "proj1.bias"] = (cin * exp,)
        s[prefix + "proj2.weight"] = (cout, cin * exp); s[prefix + "proj2.bias"] = (cout,)

    def attn_block(prefix, layer_scale):
        mlp(prefix + "mlp.", C, E, C)
        s[prefix + "kv.weight"] = (2 * C, C)
        s[prefix + "q.weight"] = (C, C)
        s[prefix + "norm_attnx.weight"] = (C,); s[prefix + "norm_attnx.bias"] = (C,)
        s[prefix + "norm_attnctx.weight"] = (C,); s[prefix + "norm_attnctx.bias"] = (C,)
        s[prefix + "out.weight"] = (C, C)
        if layer_scale:
            s[prefix + "ls1.gamma"] = (C,); s[prefix + "ls2.gamma"] = (C,)

    cl = pd + "camera_layer."
    s[cl + "latents_pos"] = (1, 4, C)
    attn_block(cl + "aggregate1.", True)
    attn_block(cl + "aggregate2.", True)
    mlp(cl + "project.", C, 1, C)
    mlp(cl + "out_pinhole.", C, 1, 1)

    dl = pd + "depth_layer."
    mult = 2
    next_dim = None
    for i, nl in enumerate(a["dec_depths"]):
        cur = min(C, mult * C // 2 ** i)
        next_dim = mult * C // 2 ** (i + 1)
        outd = max(next_dim, a["out_dim"])
        for c in range(nl):
            p = f"{dl}ups.{i}.convs.{c}."
            s[p + "gamma"] = (1, cur, 1, 1)
            s[p + "conv1.weight"] = (cur, cur, ks, ks); s[p + "conv1.bias"] = (cur,)
            s[p + "conv2.weight"] = (cur, cur, ks, ks); s[p + "conv2.bias"] = (cur,)
        s[f"{dl}ups.{i}.up.0.weight"] = (outd, cur, 1, 1); s[f"{dl}ups.{i}.up.0.bias"] = (outd,)
        k = max(1, 2 * i)
        s[f"{dl}process_features.{i}.weight"] = (C, cur, k, k)
        s[f"{dl}process_features.{i}.bias"] = (cur,)
    last = len(a["dec_depths"]) - 1
    outd = max(next_dim, a["out_dim"])
    s[f"{dl}depth_mlp.{last}.0.weight"] = (next_dim,); s[f"{dl}depth_mlp.{last}.0.bias"] = (next_dim,)
    s[f"{dl}depth_mlp.{last}.1.weight"] = (outd, next_dim); s[f"{dl}depth_mlp.{last}.1.bias"] = (outd,)
    s[dl + "confidence_mlp.0.weight"] = (next_dim,); s[dl + "confidence_mlp.0.bias"] = (next_dim,)
    s[dl + "confidence_mlp.1.weight"] = (outd, next_dim); s[dl + "confidence_mlp.1.bias"] = (outd,)
    for j in range(4):
        attn_block(f"{dl}prompt_camera.{j}.layers.0.", False)
    s[dl + "to_latents.weight"] = (C, C); s[dl + "to_latents.bias"] = (C,)
    for br in ("depth", "confidence"):
        s[f"{dl}to_{br}_lr.weight"] = (outd // 2, outd, 3, 3); s[f"{dl}to_{br}_lr.bias"] = (outd // 2,)
        s[f"{dl}to_{br}_hr.0.weight"] = (32, outd // 2, 3, 3); s[f"{dl}to_{br}_hr.0.bias"] = (32,)
        s[f"{dl}to_{br}_hr.2.weight"] = (1, 32, 1, 1); s[f"{dl}to_{br}_hr.2.bias"] = (1,)
    return s


def make_synthetic_checkpoint(config: dict, seed: int = 123) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic sensitised checkpoint; iteration order = v2_param_shapes order, one CPU generator."""
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in v2_param_shapes(config).items():
        leaf = k.rsplit(".", 1)[-1]
        t = torch.randn(shp, generator=g, dtype=torch.float32)
        is_norm = (".norm" in k or "depth_mlp.2.0." in k or "confidence_mlp.0." in k) and len(shp) == 1
        if k.endswith("pos_embed") or k.endswith("cls_token") or k.endswith("latents_pos"):
            t = t * 0.5
        elif k.endswith("register_tokens") or k.endswith("mask_token") or k.endswith("level_embeds"):
            t = t * 0.02  # unused by infer()
        elif leaf == "gamma":
            t = 1.0 + 0.1 * t
        elif is_norm and leaf == "weight":
            t = 1.0 + 0.1 * t
        elif is_norm and leaf == "bias":
            t = 0.1 * t
        elif leaf == "weight":
            if "process_features" in k:          # ConvTranspose2d [Cin, Cout, k, k]: fan_in = Cin
                fan_in = shp[0]
            else:
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
            t = t * fan_in ** -0.5
        elif leaf == "bias":
            t = 0.1 * t
        else:
            raise KeyError(k)
        sd[k] = t.contiguous()
    return sd


def make_outlier_checkpoint(config: dict, seed: int = 321) -> "OrderedDict[str, torch.Tensor]":
    """Sensitised checkpoint with the activation statistics that real DINOv2 checkpoints have and randn * fan_in^-1/2 ones lack
    (no real checkpoint is reachable offline): (a) two "massive activation" channels that sit at about +-300 in EVERY token of the
    residual stream from block depth/6 on (fc2 bias), (b) two heavy-tailed channels whose per-token values reach a few hundred
    from block depth/3 on (fc2 rows x 40), (c) one block whose attention logits are 16 x larger (q and k rows x 4: near one-hot
    softmax rows, scores in the hundreds).  Stresses the engine's fp16 storage of LayerNorm outputs, q/k, attention outputs and
    GELU(fc1), its fp32 softmax statistics (deferred max rescale) and the fp32 residual stream."""
    a = arch_of(config)
    D, depth = a["D"], a["depth"]
    sd = make_synthetic_checkpoint(config, seed)
    pe = "pixel_encoder."
    b1, b2, b3 = max(1, depth // 6), max(2, depth // 3), max(3, depth // 2)
    ch_const, ch_heavy = [7, D // 3 + 1], [D // 2 + 5, D - 9]
    for j, ch in enumerate(ch_const):
        sd[f"{pe}blocks.{b1}.mlp.fc2.bias"][ch] += (300.0 if j == 0 else -300.0) / sd[f"{pe}blocks.{b1}.ls2.gamma"][ch]
    for ch in ch_heavy:
        sd[f"{pe}blocks.{b2}.mlp.fc2.weight"][ch] *= 40.0
    qkv_w, qkv_b = sd[f"{pe}blocks.{b3}.attn.qkv.weight"], sd[f"{pe}blocks.{b3}.attn.qkv.bias"]
    qkv_w[: 2 * D] *= 4.0
    qkv_b[: 2 * D] *= 4.0
    return sd


def outlier_image(B: int, H: int, W: int, seed: int = 5) -> torch.Tensor:
    """uint8 image with saturated flat regions (0 / 255 blocks) next to noise: flat patches make many identical tokens, i.e. tied
    attention scores and zero-variance neighbourhoods for the convolutions."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)
    x[:, :, : H // 3, : W // 2] = 255
    x[:, :, H // 2:, W // 3: 2 * W // 3] = 0
    return x
