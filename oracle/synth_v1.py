"""TEST INFRASTRUCTURE ONLY -- seeded synthetic UniDepthV1 (ConvNeXt-L) checkpoints, regenerated bit-identically from a seed on any box.

Key set / shapes restate the reference's state_dict (unidepth/models/unidepthv1/unidepthv1.py:412-444 build(),
unidepthv1/decoder.py:468-533 Decoder.build(), backbones/convnext.py:301-445, layers/{attention,mlp,upsample,convnext}.py) and are
checked with strict=True against the real reference in tests/test_oracle_v1_pins.py (authoring container only).
Sensitised like the V2 checkpoints (oracle/synth.py): weights randn * fan_in^-1/2, biases randn * 0.1, all norm affines / layer
scales randomised, so folding mistakes in the engine's repacker show up."""
from __future__ import annotations

import json
import os
from collections import OrderedDict

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_config_v1(name: str = "cnvnxtl") -> dict:
    with open(os.path.join(_HERE, "configs", f"config_v1_{name}.json")) as f:
        return json.load(f)


VIT = {"dinov2_vitl14": (1024, 24, 16, [5, 12, 18, 24])}       # encoder.py:171-186: embed dim, blocks, heads, default output_idx


def v1_param_shapes(config: dict) -> "OrderedDict[str, tuple]":
    C = config["model"]["pixel_decoder"]["hidden_dim"]
    E = config["model"]["expansion"]
    dec_depths = list(config["model"]["pixel_decoder"]["depths"])
    s: "OrderedDict[str, tuple]" = OrderedDict()
    pe = "pixel_encoder."
    name = config["model"]["pixel_encoder"]["name"]
    if name in VIT:
        # UniDepthV1 on DINOv2 ViT-L/14 (configs/config_v1_vitl14.json; backbones/dinov2.py:115-264): same encoder keys as the V2 checkpoints
        D, depth, _, ends = VIT[name]
        ends = list(config["model"]["pixel_encoder"].get("output_idx", ends))
        s[pe + "cls_token"] = (1, 1, D)
        s[pe + "pos_embed"] = (1, 37 * 37 + 1, D)
        s[pe + "register_tokens"] = (1, 1, D)
        s[pe + "mask_token"] = (1, D)
        s[pe + "patch_embed.proj.weight"] = (D, 3, 14, 14)
        s[pe + "patch_embed.proj.bias"] = (D,)
        for i in range(depth):
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
        embed_dims = [D] * depth
        return _v1_decoder_shapes(s, config, ends, embed_dims, C, E, dec_depths)
    from .restate_v1 import convnext_arch
    a = convnext_arch(config)
    depths, dims = a["depths"], a["dims"]
    s[pe + "mask_token"] = (1, dims[0], 1, 1)
    s[pe + "stem.0.weight"] = (dims[0], 3, 4, 4); s[pe + "stem.0.bias"] = (dims[0],)
    s[pe + "stem.1.weight"] = (dims[0],); s[pe + "stem.1.bias"] = (dims[0],)
    for st, (dep, d) in enumerate(zip(depths, dims)):
        if st > 0:
            s[f"{pe}stages.{st}.downsample.0.weight"] = (dims[st - 1],); s[f"{pe}stages.{st}.downsample.0.bias"] = (dims[st - 1],)
            s[f"{pe}stages.{st}.downsample.1.weight"] = (d, dims[st - 1], 2, 2); s[f"{pe}stages.{st}.downsample.1.bias"] = (d,)
        for b in range(dep):
            p = f"{pe}stages.{st}.blocks.{b}."
            s[p + "gamma"] = (d,)
            s[p + "conv_dw.weight"] = (d, 1, 7, 7); s[p + "conv_dw.bias"] = (d,)
            s[p + "norm.weight"] = (d,); s[p + "norm.bias"] = (d,)
            s[p + "mlp.fc1.weight"] = (4 * d, d); s[p + "mlp.fc1.bias"] = (4 * d,)
            s[p + "mlp.fc2.weight"] = (d, 4 * d); s[p + "mlp.fc2.bias"] = (d,)

    ends = a["output_idx"]
    embed_dims = [d for dep, d in zip(depths, dims) for _ in range(dep)]
    return _v1_decoder_shapes(s, config, ends, embed_dims, C, E, dec_depths)


def _v1_decoder_shapes(s, config, ends, embed_dims, C, E, dec_depths):
    pd = "pixel_decoder."
    in_dims = [embed_dims[e - 1] for e in ends]                       # decoder.py:480
    tok_dims = [embed_dims[-i - 1] for i in range(len(ends))]         # decoder.py:478 (the LAST four blocks, deepest first)
    s[pd + "level_embeds"] = (len(in_dims), C)
    for grp, dd in (("input_adapter", in_dims), ("token_adapter", tok_dims)):
        for j, d in enumerate(dd):
            s[f"{pd}{grp}.input_adapters.{j}.0.weight"] = (d,); s[f"{pd}{grp}.input_adapters.{j}.0.bias"] = (d,)
            s[f"{pd}{grp}.input_adapters.{j}.1.weight"] = (C, d); s[f"{pd}{grp}.input_adapters.{j}.1.bias"] = (C,)

    def mlp(prefix, cin, exp, cout=None):
        cout = cin if cout is None else cout
        s[prefix + "norm.weight"] = (cin,); s[prefix + "norm.bias"] = (cin,)
        s[prefix + "proj1.weight"] = (int(cin * exp), cin); s[prefix + "proj1.bias"] = (int(cin * exp),)
        s[prefix + "proj2.weight"] = (cout, int(cin * exp)); s[prefix + "proj2.bias"] = (cout,)

    def attn_block(prefix, dim):
        mlp(prefix + "mlp.", dim, E)
        s[prefix + "kv.weight"] = (2 * dim, dim); s[prefix + "kv.bias"] = (2 * dim,)
        s[prefix + "q.weight"] = (dim, dim); s[prefix + "q.bias"] = (dim,)
        s[prefix + "norm_attnx.weight"] = (dim,); s[prefix + "norm_attnx.bias"] = (dim,)
        s[prefix + "norm_attnctx.weight"] = (dim,); s[prefix + "norm_attnctx.bias"] = (dim,)
        s[prefix + "out.weight"] = (dim, dim); s[prefix + "out.bias"] = (dim,)
        s[prefix + "ls1.gamma"] = (dim,); s[prefix + "ls2.gamma"] = (dim,)

    cl = pd + "camera_layer."
    s[cl + "latents_pos"] = (1, 4, C)
    attn_block(cl + "aggregate.", C)
    for i in range(2):
        attn_block(f"{cl}layers.{i}.", C)
    mlp(cl + "in_features.", C, 2)
    mlp(cl + "out.", C, 2, 1)
    s[cl + "cls_project.0.weight"] = (C,); s[cl + "cls_project.0.bias"] = (C,)
    s[cl + "cls_project.1.weight"] = (C // 2, C); s[cl + "cls_project.1.bias"] = (C // 2,)
    s[cl + "cls_project.3.weight"] = (C, C // 2); s[cl + "cls_project.3.bias"] = (C,)

    dl = pd + "depth_layer."
    for nm, cout in (("project_rays16", C), ("project_rays8", C // 2), ("project_rays4", C // 4)):
        mlp(f"{dl}{nm}.", 81, E, cout)
    mlp(dl + "to_latents.", C, 2)
    s[dl + "features_channel_cat.weight"] = (C, C * len(in_dims)); s[dl + "features_channel_cat.bias"] = (C,)
    for nm, d in (("up8", C), ("up4", C // 2), ("up2", C // 4)):
        for c in range(2):
            p = f"{dl}{nm}.convs.{c}."
            s[p + "gamma"] = (d,)
            s[p + "dwconv.weight"] = (d, 1, 7, 7); s[p + "dwconv.bias"] = (d,)
            s[p + "norm.weight"] = (d,); s[p + "norm.bias"] = (d,)
            s[p + "pwconv1.weight"] = (E * d, d); s[p + "pwconv1.bias"] = (E * d,)
            s[p + "pwconv2.weight"] = (d, E * d); s[p + "pwconv2.bias"] = (d,)
        s[f"{dl}{nm}.up.0.weight"] = (d // 2, d, 1, 1); s[f"{dl}{nm}.up.0.bias"] = (d // 2,)
        s[f"{dl}{nm}.up.2.weight"] = (d // 2, d // 2, 3, 3); s[f"{dl}{nm}.up.2.bias"] = (d // 2,)
    for nm, d, n in (("layers_16", C, dec_depths[0]), ("layers_8", C // 2, dec_depths[1]), ("layers_4", C // 4, dec_depths[2])):
        for i in range(n):
            attn_block(f"{dl}{nm}.{i}.", d)
    attn_block(dl + "aggregate_16.", C)
    attn_block(dl + "prompt_camera.", C)
    s[dl + "out2.weight"] = (1, C // 8, 3, 3); s[dl + "out2.bias"] = (1,)
    s[dl + "out4.weight"] = (1, C // 4, 3, 3); s[dl + "out4.bias"] = (1,)
    s[dl + "out8.weight"] = (1, C // 2, 3, 3); s[dl + "out8.bias"] = (1,)
    s[pd + "level_embed_layer.0.weight"] = (C, C); s[pd + "level_embed_layer.0.bias"] = (C,)
    s[pd + "level_embed_layer.2.weight"] = (C, C); s[pd + "level_embed_layer.2.bias"] = (C,)
    s[pd + "level_embed_layer.3.weight"] = (C,); s[pd + "level_embed_layer.3.bias"] = (C,)
    return s


def make_synthetic_checkpoint_v1(config: dict, seed: int = 211, encoder_only: bool = False) -> "OrderedDict[str, torch.Tensor]":
    g = torch.Generator().manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, shp in v1_param_shapes(config).items():
        if encoder_only and not k.startswith("pixel_encoder."):
            continue
        leaf = k.rsplit(".", 1)[-1]
        t = torch.randn(shp, generator=g, dtype=torch.float32)
        is_norm = len(shp) == 1 and (".norm" in k or "encoder.norm." in k or "stem.1." in k or "downsample.0." in k or ".input_adapters." in k and k.split(".")[-2] == "0"
                                     or "cls_project.0." in k or "level_embed_layer.3." in k)
        if k.endswith("latents_pos") or k.endswith("level_embeds") or k.endswith("pos_embed") or k.endswith("cls_token"):
            t = t * 0.5
        elif k.endswith("register_tokens"):
            t = t * 0.02                            # unused (num_register_tokens = 0)
        elif k.endswith("mask_token"):
            t = t * 0.02                            # unused by infer()
        elif leaf == "gamma":
            t = (0.5 if k.startswith("pixel_encoder.") else 1.0) * (1.0 + 0.1 * t)     # 36 encoder blocks: keep the residual stream moderate
        elif is_norm and leaf == "weight":
            t = 1.0 + 0.1 * t
        elif is_norm and leaf == "bias":
            t = 0.1 * t
        elif leaf == "weight":
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
