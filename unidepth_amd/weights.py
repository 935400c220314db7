"""Load-time weight repacking for the HIP engine (reference state_dict -> fp16 GEMM operands).

Input: the reference's state_dict layout (unidepthv2.py:418-460, decoder.py:468-524).
Everything here is exact algebra done once in fp32 on the host, then rounded to fp16:
  * LayerNorm affines are folded into the consuming Linear:  W' = W diag(gamma),  b' = b + W beta
    (so the device LayerNorm kernel only normalises and its output can be shared by several consumers);
  * LayerScale / RCU gammas are folded into the producing Linear/Conv:  W' = diag(g) W,  b' = g * b;
  * decoder attention heads (C/8 wide: 32/48/64) are zero-padded to 64 so one attention kernel serves all;
  * 3x3 conv filters become [Cout, (tap, cin)] rows, ConvTranspose2d(k=s) filters [(a, c, cout), cin] rows;
  * K is zero-padded to a multiple of 64 (MFMA K-tile).
"""
from __future__ import annotations

import torch

LOG2E = 1.4426950408889634

ARCH = {  # backbones/dinov2.py:388-427, encoder.py:139-193
    "dinov2_vits14": (384, 12, 6, [3, 6, 9, 12]),
    "dinov2_vitb14": (768, 12, 12, [3, 6, 9, 12]),
    "dinov2_vitl14": (1024, 24, 16, [5, 12, 18, 24]),
}


def arch_of(config: dict) -> dict:
    enc = config["model"]["pixel_encoder"]
    if enc["name"] not in ARCH:
        raise NotImplementedError(f"pixel_encoder {enc['name']!r}: only the DINOv2 ViT-S/B/L backbones of UniDepthV2 are implemented")
    D, depth, heads, out_idx = ARCH[enc["name"]]
    # pack() folds the encoder's final LayerNorm into the adapters and takes each level's last block output: the only setting the
    # released V2 configs use (encoder.py:139-193 builds the backbone with use_norm / stacking from these keys)
    if not enc.get("use_norm", False) or enc.get("stacking_fn", "last") != "last":   # reference default: use_norm=False (encoder.py:150)
        raise NotImplementedError("pixel_encoder: only use_norm=true, stacking_fn='last' (all released V2 configs) is implemented")
    dec = config["model"]["pixel_decoder"]
    if dec.get("kernel_size", 7) != 3 or list(dec["depths"]) != [2, 2, 2]:
        raise NotImplementedError("pixel_decoder: only kernel_size=3, depths=[2,2,2] (all released V2 configs) is implemented")
    return dict(D=D, depth=depth, heads=heads, output_idx=list(enc.get("output_idx", out_idx)), C=dec["hidden_dim"],
                dec_heads=config["model"]["num_heads"], expansion=config["model"]["expansion"], out_dim=dec["out_dim"])


def _rup(x, m):
    return (x + m - 1) // m * m


def _padk(w: torch.Tensor) -> torch.Tensor:
    n, k = w.shape
    kp = _rup(k, 64)
    if kp == k:
        return w
    out = w.new_zeros(n, kp)
    out[:, :k] = w
    return out


def _fold_ln(w, b, gamma, beta):
    b0 = b if b is not None else w.new_zeros(w.shape[0])
    return w * gamma[None, :], b0 + w @ beta


def _pad_head_rows(w, heads, hd):           # [heads*hd, K] -> [heads*64, K]
    out = w.new_zeros(heads, 64, w.shape[1])
    out[:, :hd] = w.view(heads, hd, -1)
    return out.reshape(heads * 64, -1)


def _pad_head_vec(v, heads, hd):
    out = v.new_zeros(heads, 64)
    out[:, :hd] = v.view(heads, hd)
    return out.reshape(-1)


def _pad_head_cols(w, heads, hd):           # [N, heads*hd] -> [N, heads*64]
    out = w.new_zeros(w.shape[0], heads, 64)
    out[:, :, :hd] = w.view(w.shape[0], heads, hd)
    return out.reshape(w.shape[0], heads * 64)


def _conv3_rows(w):                          # [Cout, Cin, 3, 3] -> [Cout, 9*Cin], k = (ky*3 + kx)*Cin + ci
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)


def pack_vit_blocks(f: dict, D: int, depth: int, heads: int, put16, put32) -> None:
    """The DINOv2 encoder's GEMM operands (patch embedding + `depth` blocks) from the reference's `pixel_encoder.*` tensors: norm1 / norm2
    affines folded into qkv / fc1, LayerScale into proj / fc2, softmax scale and log2(e) into the q rows.  Shared by UniDepthV2 and by
    UniDepthV1 on the ViT-L backbone; `put16(name, w, wsum=False)` stores a GEMM operand (its row sums too when asked)."""
    pe = "pixel_encoder."
    put16("patch.w", f[pe + "patch_embed.proj.weight"].reshape(D, -1))
    put32("patch.b", f[pe + "patch_embed.proj.bias"])
    for i in range(depth):
        b = f"{pe}blocks.{i}."
        w, bb = _fold_ln(f[b + "attn.qkv.weight"], f[b + "attn.qkv.bias"], f[b + "norm1.weight"], f[b + "norm1.bias"])
        # softmax scale and the log2(e) of the kernel's exp2 live in the q projection (exact algebra; UdAttention.q_prescaled)
        qc = (D // heads) ** -0.5 * LOG2E
        w, bb = w.clone(), bb.clone()
        w[:D] *= qc; bb[:D] *= qc
        put16(f"enc.{i}.qkv.w", w, wsum=True); put32(f"enc.{i}.qkv.b", bb)
        g1 = f[b + "ls1.gamma"]
        put16(f"enc.{i}.proj.w", f[b + "attn.proj.weight"] * g1[:, None]); put32(f"enc.{i}.proj.b", f[b + "attn.proj.bias"] * g1)
        w, bb = _fold_ln(f[b + "mlp.fc1.weight"], f[b + "mlp.fc1.bias"], f[b + "norm2.weight"], f[b + "norm2.bias"])
        put16(f"enc.{i}.fc1.w", w, wsum=True); put32(f"enc.{i}.fc1.b", bb)
        g2 = f[b + "ls2.gamma"]
        put16(f"enc.{i}.fc2.w", f[b + "mlp.fc2.weight"] * g2[:, None]); put32(f"enc.{i}.fc2.b", f[b + "mlp.fc2.bias"] * g2)


def pack(config: dict, sd: dict, device) -> dict:
    a = arch_of(config)
    D, C, H = a["D"], a["C"], a["dec_heads"]
    hd = C // H
    f = {k: v.detach().to(torch.float32).cpu() for k, v in sd.items()}
    out: dict = {}

    def put16(name, w, wsum=False):
        w16 = _padk(w).to(torch.float16).contiguous()
        out[name] = w16.to(device)
        if wsum:
            # row sums of the ROUNDED weights: a LayerNorm folded into this GEMM's epilogue computes rstd (x W^T - mean 1 W^T), and the
            # cancellation against the MFMA's x W^T is only exact with the very operand the matrix pipe multiplies (UdGemm.wsum)
            out[name + "sum"] = w16.to(torch.float64).sum(dim=1).to(torch.float32).contiguous().to(device)

    def put32(name, v):
        out[name] = v.to(torch.float32).contiguous().to(device)

    pe = "pixel_encoder."
    pack_vit_blocks(f, D, a["depth"], a["heads"], put16, put32)
    gn, bn = f[pe + "norm.weight"], f[pe + "norm.bias"]
    put32("enc.norm.g", gn); put32("enc.norm.b", bn)     # only the module seams need the affine itself (it is folded into the adapters)

    pd = "pixel_decoder."
    for j in range(4):
        w, bb = _fold_ln(f[f"{pd}input_adapter.input_adapters.{j}.weight"], f[f"{pd}input_adapter.input_adapters.{j}.bias"], gn, bn)
        put16(f"dec.adapter.{j}.w", w); put32(f"dec.adapter.{j}.b", bb)
        # camera tokens feed the fp32 camera head (see UdLinearF32 in include/unidepth_hip.h): weights stay fp32
        w, bb = _fold_ln(f[f"{pd}camera_token_adapter.input_adapters.{j}.weight"], f[f"{pd}camera_token_adapter.input_adapters.{j}.bias"], gn, bn)
        put32(f"dec.camadapter.{j}.w", w); put32(f"dec.camadapter.{j}.b", bb)

    def mlp(src, dst, ls=None, pad_out_to=None):
        w, bb = _fold_ln(f[src + "proj1.weight"], f[src + "proj1.bias"], f[src + "norm.weight"], f[src + "norm.bias"])
        put16(dst + "fc1.w", w); put32(dst + "fc1.b", bb)
        w2, b2 = f[src + "proj2.weight"], f[src + "proj2.bias"]
        if ls is not None:
            w2, b2 = w2 * ls[:, None], b2 * ls
        if pad_out_to is not None and w2.shape[0] < pad_out_to:
            wz = w2.new_zeros(pad_out_to, w2.shape[1]); wz[: w2.shape[0]] = w2
            bz = b2.new_zeros(pad_out_to); bz[: b2.shape[0]] = b2
            w2, b2 = wz, bz
        put16(dst + "fc2.w", w2); put32(dst + "fc2.b", b2)

    def attn_block(src, dst, layer_scale):
        wq, bq = _fold_ln(f[src + "q.weight"], None, f[src + "norm_attnx.weight"], f[src + "norm_attnx.bias"])
        qc = hd ** -0.5 * LOG2E                                              # q pre-scaled for the attention kernel (UdAttention.q_prescaled)
        put16(dst + "q.w", _pad_head_rows(wq * qc, H, hd)); put32(dst + "q.b", _pad_head_vec(bq * qc, H, hd))
        wkv, bkv = _fold_ln(f[src + "kv.weight"], None, f[src + "norm_attnctx.weight"], f[src + "norm_attnctx.bias"])
        wk, wv, bk, bv = wkv[:C], wkv[C:], bkv[:C], bkv[C:]                     # rows [K | V], heads-major (attention.py:119-121)
        put16(dst + "kv.w", torch.cat([_pad_head_rows(wk, H, hd), _pad_head_rows(wv, H, hd)], 0))
        put32(dst + "kv.b", torch.cat([_pad_head_vec(bk, H, hd), _pad_head_vec(bv, H, hd)], 0))
        wo = _pad_head_cols(f[src + "out.weight"], H, hd)
        ls1 = f[src + "ls1.gamma"] if layer_scale else None
        if ls1 is not None:
            wo = wo * ls1[:, None]
        put16(dst + "out.w", wo)
        mlp(src + "mlp.", dst, ls=f[src + "ls2.gamma"] if layer_scale else None)

    # ---- camera head: fp32 weights, natural head width (no padding), same folds
    def mlp32(src, dst, ls=None):
        w, bb = _fold_ln(f[src + "proj1.weight"], f[src + "proj1.bias"], f[src + "norm.weight"], f[src + "norm.bias"])
        put32(dst + "fc1.w", w); put32(dst + "fc1.b", bb)
        w2, b2 = f[src + "proj2.weight"], f[src + "proj2.bias"]
        if ls is not None:
            w2, b2 = w2 * ls[:, None], b2 * ls
        put32(dst + "fc2.w", w2); put32(dst + "fc2.b", b2)

    def attn_block32(src, dst):
        wq, bq = _fold_ln(f[src + "q.weight"], None, f[src + "norm_attnx.weight"], f[src + "norm_attnx.bias"])
        put32(dst + "q.w", wq); put32(dst + "q.b", bq)
        wkv, bkv = _fold_ln(f[src + "kv.weight"], None, f[src + "norm_attnctx.weight"], f[src + "norm_attnctx.bias"])
        put32(dst + "kv.w", wkv); put32(dst + "kv.b", bkv)
        put32(dst + "out.w", f[src + "out.weight"] * f[src + "ls1.gamma"][:, None])
        # one [q | k | v] projection for the one-launch camera head (UdCameraHead: both read the same normalised rows)
        put32(dst + "qkv.w", torch.cat([wq, wkv], 0)); put32(dst + "qkv.b", torch.cat([bq, bkv], 0))
        mlp32(src + "mlp.", dst, ls=f[src + "ls2.gamma"])

    cl = pd + "camera_layer."
    mlp32(cl + "project.", "cam.project.")
    attn_block32(cl + "aggregate1.", "cam.agg1.")
    attn_block32(cl + "aggregate2.", "cam.agg2.")
    mlp32(cl + "out_pinhole.", "cam.out.")
    put32("cam.pos", f[cl + "latents_pos"].reshape(4, C))

    dl = pd + "depth_layer."
    for j in range(4):
        attn_block(f"{dl}prompt_camera.{j}.layers.0.", f"dh.{j}.", False)
    put16("dh.to_latents.w", f[dl + "to_latents.weight"]); put32("dh.to_latents.b", f[dl + "to_latents.bias"])
    chans = []
    for i in range(3):
        cur = min(C, 2 * C // 2 ** i)
        nxt = 2 * C // 2 ** (i + 1)
        outd = max(nxt, a["out_dim"])
        chans.append((cur, outd))
        k = max(1, 2 * i)
        wt = f[f"{dl}process_features.{i}.weight"]                                # [Cin, Cout, k, k]
        put16(f"dh.convt.{i}.w", wt.permute(2, 3, 1, 0).reshape(k * k * cur, C)) # n = (a*k + c)*Cout + o
        put32(f"dh.convt.{i}.b", f[f"{dl}process_features.{i}.bias"])
        for c in range(2):
            p = f"{dl}ups.{i}.convs.{c}."
            g = f[p + "gamma"].reshape(-1)
            put16(f"dh.ups.{i}.{c}.conv1.w", _conv3_rows(f[p + "conv1.weight"])); put32(f"dh.ups.{i}.{c}.conv1.b", f[p + "conv1.bias"])
            put16(f"dh.ups.{i}.{c}.conv2.w", _conv3_rows(f[p + "conv2.weight"]) * g[:, None]); put32(f"dh.ups.{i}.{c}.conv2.b", f[p + "conv2.bias"] * g)
        put16(f"dh.ups.{i}.up.w", f[f"{dl}ups.{i}.up.0.weight"].reshape(outd, cur)); put32(f"dh.ups.{i}.up.b", f[f"{dl}ups.{i}.up.0.bias"])
    nd = 2 * C // 8            # channels of the x8 feature map
    od = max(nd, a["out_dim"])
    # LN -> Linear -> 3x3 reflect conv has no non-linearity after the normalisation (decoder.py:186-188,199-212,297-298):
    # the per-pixel Linear (with the LN affine folded in) is composed INTO the conv filter at load time,
    #   W'[o, tap, ci] = sum_c Wconv[o, c, tap] Wlin[c, ci],   b'[o] = bconv[o] + sum_{c,tap} Wconv[o, c, tap] blin[c]
    # (a constant bias map stays constant under reflect padding), so the device runs one conv on the normalised x8 map.
    lr_w, lr_b = [], []
    for br, mlp_pre in (("depth", f"{dl}depth_mlp.2"), ("confidence", f"{dl}confidence_mlp")):
        wl, bl = _fold_ln(f[mlp_pre + ".1.weight"], f[mlp_pre + ".1.bias"], f[mlp_pre + ".0.weight"], f[mlp_pre + ".0.bias"])   # [od, nd]
        wc, bc_ = f[f"{dl}to_{br}_lr.weight"].double(), f[f"{dl}to_{br}_lr.bias"].double()                                    # [o2, od, 3, 3]
        wcomp = torch.einsum("ocyx,ci->oyxi", wc, wl.double()).reshape(wc.shape[0], -1)                                           # k = tap*nd + ci
        bcomp = bc_ + torch.einsum("ocyx,c->o", wc, bl.double())
        lr_w.append(_padk(wcomp.float())); lr_b.append(bcomp.float())
    out["dh.lr.w"] = torch.stack(lr_w, 0).to(torch.float16).contiguous().to(device)
    put32("dh.lr.b", torch.stack(lr_b, 0))
    hr_w = torch.stack([_padk(_conv3_rows(f[f"{dl}to_{br}_hr.0.weight"])) for br in ("depth", "confidence")], 0)
    out["dh.hr.w"] = hr_w.to(torch.float16).contiguous().to(device)
    put32("dh.hr.b1", torch.stack([f[f"{dl}to_{br}_hr.0.bias"] for br in ("depth", "confidence")], 0))
    put32("dh.hr.w2", torch.stack([f[f"{dl}to_{br}_hr.2.weight"].reshape(32) for br in ("depth", "confidence")], 0))
    out["dh.hr.b2"] = [float(f[f"{dl}to_{br}_hr.2.bias"].reshape(())) for br in ("depth", "confidence")]
    # the 4 camera-prompt blocks (and the 4 adapters) are independent -> stacked [4, N, K] copies for grouped launches
    for name in ("q.w", "q.b", "kv.w", "kv.b", "out.w", "fc1.w", "fc1.b", "fc2.w", "fc2.b"):
        out["dhg." + name] = torch.stack([out.pop(f"dh.{j}.{name}") for j in range(4)], 0).contiguous()
    for name in ("w", "b"):
        out["dec.adapterg." + name] = torch.stack([out.pop(f"dec.adapter.{j}.{name}") for j in range(4)], 0).contiguous()
    out["meta"] = dict(chans=chans, nd=nd, od=od, hd=hd)
    # host copies needed per input shape
    out["host.pos_embed"] = f[pe + "pos_embed"]
    out["host.cls_token"] = f[pe + "cls_token"].reshape(-1)
    return out
