"""UniDepthV1 components on the engine (SURVEY.md 8f next-1 / BASELINE.json configs[3]) against the oracle restatement
(oracle/restate_v1.py, pinned to the reference's own ConvNeXt code): kernel-level checks of the new ConvNeXt-side ops and the whole
ConvNeXt-L encoder.  Bars: fp32 ops <= 2e-5 rel-L2; encoder features (fp16 MFMA operands, fp32 residual stream) <= 3e-3."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import restate_v1, synth_v1  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidepth_amd import ops as _ops
    return _ops


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("B,H,W,C", [(2, 13, 21, 192), (1, 7, 9, 1536), (1, 30, 8, 384), (1, 3, 50, 64)])
def test_dwconv7(ops, B, H, W, C):
    import ctypes
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, H, W, C, generator=g).cuda()
    w = torch.randn(C, 1, 7, 7, generator=g).cuda() / 7
    b = torch.randn(C, generator=g).cuda()
    y = torch.zeros(B, H, W, C, device="cuda")
    wt = w.reshape(C, 49).t().contiguous()
    d = ops.mk(ops.UdDwConv7, x=x, w=wt, bias=b, y=y, B=B, H=H, W=W, C=C, ldx=C, ldy=C)
    ops.check(ops.lib.ud_dwconv7_nhwc_f32(ctypes.byref(d), ops.cur_stream()))
    ref = F.conv2d(x.permute(0, 3, 1, 2), w, b, padding=3, groups=C).permute(0, 2, 3, 1)
    torch.cuda.synchronize()
    assert rel(y, ref) < 2e-6


@pytest.mark.parametrize("B,H,W,C", [(2, 9, 14, 192), (1, 7, 8, 768)])
def test_layernorm_patchify2_and_conv(ops, B, H, W, C):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, C, generator=g).cuda() * 2 + 0.3
    Ho, Wo = H // 2, W // 2
    out = torch.zeros(B * Ho * Wo, 4 * C, dtype=torch.half, device="cuda")
    ops.check(ops.lib.ud_layernorm_patchify2(x.data_ptr(), out.data_ptr(), B, H, W, C, 4 * C, 1e-6, ops.cur_stream()))
    xn = F.layer_norm(x, (C,), eps=1e-6)
    ref = xn[:, : 2 * Ho, : 2 * Wo].reshape(B, Ho, 2, Wo, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(B * Ho * Wo, 4 * C)
    torch.cuda.synchronize()
    assert rel(out.float(), ref) < 1e-3


def test_patchify4_max_mean(ops):
    g = torch.Generator().manual_seed(2)
    img = torch.randn(2, 3, 22, 30, generator=g).cuda()
    Ho, Wo = 5, 7
    out = torch.zeros(2 * Ho * Wo, 64, dtype=torch.half, device="cuda")
    ops.check(ops.lib.ud_patchify4_nchw(img.data_ptr(), out.data_ptr(), 2, 22, 30, 64, ops.cur_stream()))
    ref = F.unfold(img[:, :, : 4 * Ho, : 4 * Wo], kernel_size=4, stride=4).transpose(1, 2).reshape(2 * Ho * Wo, 48)
    torch.cuda.synchronize()
    assert rel(out[:, :48].float(), ref) < 1e-3 and (out[:, 48:] == 0).all()
    a, b = torch.randn(4096, generator=g).cuda(), torch.randn(4096, generator=g).cuda()
    d = torch.zeros(4096, device="cuda")
    ops.check(ops.lib.ud_max_f32(d.data_ptr(), a.data_ptr(), 4096, 1, ops.cur_stream()))
    ops.check(ops.lib.ud_max_f32(d.data_ptr(), b.data_ptr(), 4096, 0, ops.cur_stream()))
    assert torch.equal(d, torch.maximum(a, b))
    x = torch.randn(3, 266, 1536, generator=g).cuda()
    m = torch.zeros(3, 1536, device="cuda")
    ops.check(ops.lib.ud_spatial_mean_f32(x.data_ptr(), m.data_ptr(), 3, 266, 1536, 1536, ops.cur_stream()))
    torch.cuda.synchronize()
    assert rel(m, x.mean(dim=1)) < 1e-6


def test_layernorm_affine_fp32(ops):
    g = torch.Generator().manual_seed(3)
    x = torch.randn(300, 192, generator=g).cuda() * 3
    gm, bt = torch.randn(192, generator=g).cuda(), torch.randn(192, generator=g).cuda()
    y = torch.zeros(300, 192, device="cuda")
    ops.layernorm(x=x, y=y, rows=300, D=192, ldx=192, ldy=192, eps=1e-6, rows_per_img=300, in_rows_per_img=300, out_rows_per_img=300,
                  out_f32=1, gamma=gm, beta=bt)
    torch.cuda.synchronize()
    assert rel(y, F.layer_norm(x, (192,), gm, bt, 1e-6)) < 2e-6


@pytest.mark.parametrize("B,H,W", [(1, 128, 160), (2, 462, 616)])
def test_convnext_encoder_vs_oracle(B, H, W):
    """ConvNeXt-L encoder (36 blocks) on the engine vs the oracle: stage-wise max_stack features, the four class tokens the decoder
    reads, and a sample of individual block outputs through the reference-signature seam pixel_encoder()."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidepth_amd import UniDepthV1
    cfg = synth_v1.load_config_v1()
    sd = synth_v1.make_synthetic_checkpoint_v1(cfg, 211, encoder_only=True)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 3, H, W, generator=g)
    orc = restate_v1.OracleConvNeXt(cfg, sd)
    outs, cls = orc.encode(x)
    feats = orc.stage_features(outs)
    model = UniDepthV1(cfg).load_state_dict(sd).to("cuda").eval()
    ef, ec = model.stage_features(x.cuda())
    torch.cuda.synchronize()
    res = {}
    for j in range(4):
        assert ef[j].shape == feats[j].shape
        res[f"stage{j}"] = rel(ef[j], feats[j])
        res[f"cls{j}"] = rel(ec[j], cls[-j - 1])
    print(f"convnext {B}x{H}x{W}", {k: f"{v:.1e}" for k, v in res.items()}, "|x| max", float(feats[3].abs().max()))
    assert all(v <= 3e-3 for v in res.values()), res
    if H <= 128:
        eo, ecls = model.pixel_encoder(x.cuda())
        assert len(eo) == 36 and all(o is not None for o in eo)
        for i in (0, 2, 5, 17, 32, 35):
            assert rel(eo[i], outs[i]) <= 3e-3 and rel(ecls[i], cls[i]) <= 3e-3, i
    with pytest.raises(NotImplementedError):
        model.infer(torch.zeros(1, 3, 64, 64, dtype=torch.uint8))
