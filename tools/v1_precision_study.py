"""CPU study: where does UniDepthV1's fp16-operand error come from?  

Runs the fp32 oracle (oracle/restate_v1.py) with the operands of selected GEMMs rounded to fp16 -- what the engine's MFMA path
does -- and reports the depth ARel against the unrounded oracle.  Every F.linear / F.conv2d is matched to its weight's state_dict
name; a rule list {substring: mode} (longest match wins) picks the mode:
  'x'  exact (fp32 operands) = what a 3-term split-fp16 product (hi*hi + lo*hi + hi*lo) delivers
  'h'  both operands fp16          (the engine in round 2)
  'a'  A (activations) exact, W fp16
  'w'  A fp16, W exact                 (2-term split of the weights only)
Attention internals (softmax(q k^T) v, Nystrom matmuls) follow the mode of the enclosing block's name + ".attn".
The default rule set "engine" mirrors unidepth_amd/unidepthv1.py as built in round 2 (fp32 camera transformer, fp32 depth-wise convs).

    python tools/v1_precision_study.py            # a few minutes on 8 vCPU
"""
import contextlib
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restate_v1, synth_v1                                     # noqa: E402

_orig = dict(linear=F.linear, conv2d=F.conv2d, sdpa=F.scaled_dot_product_attention)
STATE = {"names": {}, "rules": {}, "attn": "x"}


def r16(t):
    return t.half().float()


def mode_of(name):
    best, bm = -1, "x"
    for k, m in STATE["rules"].items():
        if k in name and len(k) > best:
            best, bm = len(k), m
    return bm


def _rounders(mode):
    ra = r16 if mode in ("h", "w") else (lambda t: t)
    rw = r16 if mode in ("h", "a") else (lambda t: t)
    return ra, rw


def _pixel_mean(x, chan_dim):
    """mean over everything but the batch (dim 0) and the channel dim, kept for broadcasting"""
    dims = [d for d in range(x.dim()) if d not in (0, chan_dim % x.dim())]
    return x.mean(dim=dims, keepdim=True) if dims and x.dim() > 2 else x.mean(dim=0, keepdim=True)


def linear(x, w, b=None):
    m = mode_of(STATE["names"].get(id(w), "?"))
    if m == "m":        # single fp16 weights + MEAN-FIELD correction: the per-image mean activation against the weights' rounding error
        y = _orig["linear"](r16(x), r16(w), b)
        return y + _orig["linear"](_pixel_mean(x, -1), w - r16(w))
    ra, rw = _rounders(m)
    return _orig["linear"](ra(x), rw(w), b)


def conv2d(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
    if groups != 1:                                   # depth-wise 7x7: fp32 VALU in the engine
        return _orig["conv2d"](x, w, b, stride, padding, dilation, groups)
    m = mode_of(STATE["names"].get(id(w), "?"))
    if m == "m":
        y = _orig["conv2d"](r16(x), r16(w), b, stride, padding, dilation, groups)
        return y + _orig["conv2d"](_pixel_mean(x, 1).expand_as(x), w - r16(w), None, stride, padding, dilation, groups)
    ra, rw = _rounders(m)
    return _orig["conv2d"](ra(x), rw(w), b, stride, padding, dilation, groups)


def sdpa(q, k, v, *a, **kw):
    m = STATE["attn"]
    if m == "x":
        return _orig["sdpa"](q, k, v, *a, **kw)
    # fp16 q, k, v operands; probabilities rounded to fp16 before P V (flash kernel / softmax kernel with fp16 output)
    q, k, v = r16(q), r16(k), r16(v)
    p = torch.softmax((q @ k.transpose(-1, -2)) * q.shape[-1] ** -0.5, dim=-1)
    return r16(p) @ v


def nystrom(q, k, v, num_landmarks=128):
    m = STATE["attn"]
    if m == "x":
        return _nys_orig(q, k, v, num_landmarks)
    import math
    q, k, v = r16(q), r16(k), r16(v)
    ql, kl = r16(restate_v1.segment_means(q, num_landmarks)), r16(restate_v1.segment_means(k, num_landmarks))
    s = 1.0 / math.sqrt(k.shape[-1])
    k1 = r16(torch.softmax(q @ kl.transpose(-1, -2) * s, dim=-1))
    k2 = torch.softmax(ql @ kl.transpose(-1, -2) * s, dim=-1)
    k3 = r16(torch.softmax(ql @ k.transpose(-1, -2) * s, dim=-1)) @ v
    return k1 @ r16(restate_v1.iterative_pinv(k2) @ k3)


_nys_orig = restate_v1.nystrom_attention


@contextlib.contextmanager
def patched():
    F.linear, F.conv2d, F.scaled_dot_product_attention = linear, conv2d, sdpa
    restate_v1.nystrom_attention = nystrom
    try:
        yield
    finally:
        F.linear, F.conv2d, F.scaled_dot_product_attention = _orig["linear"], _orig["conv2d"], _orig["sdpa"]
        restate_v1.nystrom_attention = _nys_orig


class Study(restate_v1.OracleV1):
    def __init__(self, cfg, sd):
        super().__init__(cfg, sd)
        STATE["names"] = {id(v): k for k, v in self.w.items()}

    def _attn_block(self, x, p, heads, **kw):
        old, STATE["attn"] = STATE["attn"], mode_of(p + "attn")
        try:
            return super()._attn_block(x, p, heads, **kw)
        finally:
            STATE["attn"] = old


# the engine as built in round 2: every MFMA GEMM has fp16 operands; fp32 islands: the camera transformer except its two big GEMM groups
ENGINE_R2 = {"": "h", "camera_layer": "x", "camera_layer.in_features": "h", "camera_layer.aggregate.kv": "h", "token_adapter": "x"}


def main():
    torch.set_num_threads(int(os.environ.get("NT", "8")))
    vit = os.environ.get("ARCH", "cnvnxtl") == "vitl14"            # ARCH=vitl14: UniDepthV1 on the DINOv2 ViT-L/14 backbone
    cfg = synth_v1.load_config_v1("vitl14" if vit else "cnvnxtl")
    sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 212 if vit else 211)
    cases = [(1, 240, 320), (1, 480, 640), (2, 200, 360)][: int(os.environ.get("NC", "3"))]
    refs, ins = {}, {}
    for c in cases:
        ins[c] = torch.randint(0, 256, (c[0], 3, c[1], c[2]), dtype=torch.uint8, generator=torch.Generator().manual_seed(5))
        refs[c] = restate_v1.OracleV1(cfg, sd).infer(ins[c])

    STATE_NAMES = list(sd.keys())

    def run(tag, rules):
        if os.environ.get("ONLY") and os.environ["ONLY"] not in tag:          # ONLY=substring: run the matching rule sets only
            return
        STATE["rules"] = rules
        res = []
        for c in cases:
            with patched():
                out = Study(cfg, sd).infer(ins[c])
            d = ((out["depth"] - refs[c]["depth"]).abs() / refs[c]["depth"].abs().clamp_min(1e-6)).mean().item()
            kk = ((out["intrinsics"] - refs[c]["intrinsics"]).abs() / refs[c]["intrinsics"].abs().clamp_min(1.0)).max().item()
            res.append(f"{d:.2e}/{kk:.1e}")
        print(f"{tag:64s} depth ARel / K max-rel: " + "  ".join(res), flush=True)

    def with_(base, **kw):
        r = dict(base)
        r.update({k.replace("__", "."): v for k, v in kw.items()})
        return r

    if len(sys.argv) > 1 and sys.argv[1] == "placement_vit":
        W_ALL = {**ENGINE_R2, "": "w", "camera_layer.in_features": "w", "camera_layer.aggregate.kv": "w"}
        run("split weights everywhere", W_ALL)
        run("ViT qkv + fc1 (LayerNorm-fed) single fp16, rest split", {**W_ALL, "attn.qkv": "h", "mlp.fc1": "h"})
        run("ViT fc1 single fp16, rest split", {**W_ALL, "mlp.fc1": "h"})
        run("ViT qkv single fp16, rest split", {**W_ALL, "attn.qkv": "h"})
        run("ViT proj + fc2 single fp16, rest split", {**W_ALL, "attn.proj": "h", "mlp.fc2": "h"})
        run("whole ViT encoder single fp16, decoder split", {**W_ALL, "pixel_encoder": "h"})
        return
    if len(sys.argv) > 1 and sys.argv[1] == "placement":
        # round 3: WHERE the weights have to be exact.  'w' = split-fp16 weights (A fp16, W ~fp32), 'h' = single fp16 weights.
        W_ALL = {**ENGINE_R2, "": "w", "camera_layer.in_features": "w", "camera_layer.aggregate.kv": "w"}
        run("split weights everywhere (the engine since 9.1)", W_ALL)
        run("encoder single fp16, decoder split", {**W_ALL, "pixel_encoder": "h"})
        run("encoder stage 2 single fp16, rest split", {**W_ALL, "pixel_encoder.stages.2": "h"})
        run("encoder stage 2 fc1 single fp16, rest split", {**W_ALL, **{f"pixel_encoder.stages.2.blocks.{i}.mlp.fc1": "h" for i in range(27)}})
        run("encoder stage 2 fc2 single fp16, rest split", {**W_ALL, **{f"pixel_encoder.stages.2.blocks.{i}.mlp.fc2": "h" for i in range(27)}})
        run("encoder stage 2 blocks 0-13 single fp16, rest split", {**W_ALL, **{f"pixel_encoder.stages.2.blocks.{i}.": "h" for i in range(14)}})
        run("encoder stages 2+3 single fp16, rest split", {**W_ALL, "pixel_encoder.stages.2": "h", "pixel_encoder.stages.3": "h"})
        run("encoder split, decoder single fp16", {**ENGINE_R2, "pixel_encoder": "w"})
        # mean-field correction instead of the second term: y = A fp16(W)^T + mean_pixels(A) (W - fp16(W))^T  (a GEMV per image)
        M_ALL = {k: ("m" if v == "w" else v) for k, v in W_ALL.items()}
        run("mean-field correction everywhere instead of the split", M_ALL)
        run("mean-field: ConvNeXt fc2 only, fc1 single, rest split", {**W_ALL, **{f"pixel_encoder.stages.{s}.blocks.{i}.mlp.fc1": "h" for s, d in enumerate((3, 3, 27, 3)) for i in range(d)},
                                                                       **{f"pixel_encoder.stages.{s}.blocks.{i}.mlp.fc2": "m" for s, d in enumerate((3, 3, 27, 3)) for i in range(d)}})
        run("mean-field: encoder (fc1 single), decoder split", {**W_ALL, "pixel_encoder": "m", **{f"pixel_encoder.stages.{s}.blocks.{i}.mlp.fc1": "h" for s, d in enumerate((3, 3, 27, 3)) for i in range(d)}})
        # GEMMs whose A operand is a LayerNorm output (zero-mean rows): ConvNeXt fc1 / pwconv1, the transformer blocks' q / kv / mlp.proj1
        FC1 = {f"pixel_encoder.stages.{s}.blocks.{i}.mlp.fc1": "h" for s, d in enumerate((3, 3, 27, 3)) for i in range(d)}
        run("every encoder fc1 single fp16, rest split", {**W_ALL, **FC1})
        LNFED = {k: "h" for k in STATE_NAMES if k.startswith("pixel_decoder.depth_layer") and
                 (k.endswith(".mlp.proj1.weight") or k.endswith(".q.weight") or k.endswith(".kv.weight") or k.endswith(".pwconv1.weight"))}
        run("encoder fc1 + the depth decoder's LayerNorm-fed GEMMs single fp16", {**W_ALL, **FC1, **LNFED})
        return
    E = ENGINE_R2
    run("engine r2 (all MFMA GEMMs fp16 x fp16)", E)
    run("encoder exact", with_(E, pixel_encoder="x"))
    run("decoder exact", with_({"": "x"}, pixel_encoder="h"))
    UPS = {f"depth_layer.up{s}": "x" for s in (8, 4, 2)}
    OUTS = {f"depth_layer.out{s}": "x" for s in (8, 4, 2)}
    run("out convs exact", {**E, **OUTS})
    run("out convs + up* exact", {**E, **OUTS, **UPS})
    W_ALL = {**E, "": "w", "camera_layer.in_features": "w", "camera_layer.aggregate.kv": "w"}
    run("W exact everywhere", W_ALL)
    run("W exact everywhere, out convs + up* exact", {**W_ALL, **OUTS, **UPS})
    DEC_W = {**W_ALL, "pixel_encoder": "h"}
    run("W exact in the decoder only", DEC_W)
    run("W exact in decoder, out convs + up* exact", {**DEC_W, **OUTS, **UPS})
    run("W exact in decoder, outs + up* + adapters + to_latents exact",
        {**DEC_W, **OUTS, **UPS, "input_adapter": "x", "to_latents": "x", "features_channel_cat": "x"})
    LIN = {**OUTS, **UPS, "input_adapter": "x", "to_latents": "x", "features_channel_cat": "x", "project_rays": "x",
           "camera_layer.in_features": "x", "camera_layer.aggregate.kv": "x"}
    run("all decoder Linear / conv exact, attention internals fp16",
        {**E, **LIN, "depth_layer.aggregate_16": "x", "depth_layer.prompt_camera": "x", "depth_layer.layers_": "x",
         "depth_layer.aggregate_16.attn": "h", "depth_layer.prompt_camera.attn": "h", "depth_layer.layers_16.0.attn": "h",
         "depth_layer.layers_16.1.attn": "h", "depth_layer.layers_8.0.attn": "h", "depth_layer.layers_8.1.attn": "h",
         "depth_layer.layers_4.0.attn": "h", "depth_layer.layers_4.1.attn": "h"})


if __name__ == "__main__":
    main()
