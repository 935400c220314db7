"""Parity hardening (VERDICT r1 "Next round" #3): the headline batch on ALL its images, tap-level comparisons against the oracle
(SURVEY.md 8c tap list and tolerances), outlier-sensitised weights / inputs, the module seams (8b/B2), bicubic post-processing,
and the warm-state stream-safety case of the in-flight pipeline.  Everything goes through the C-ABI library."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import cases, restate, synth  # noqa: E402


@pytest.fixture(scope="module")
def engine_cls():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidepth_amd import UniDepthV2
    return UniDepthV2


def _arel(a, b):
    return ((a - b).abs() / b.abs().clamp_min(1e-6)).mean().item()


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _outputs_ok(out, ref, tag, depth_bar=1e-3):
    o = {k: v.float().cpu() for k, v in out.items()}
    st = {"depth": _arel(o["depth"], ref["depth"]), "conf": _arel(o["confidence"], ref["confidence"]),
          "K": ((o["intrinsics"] - ref["intrinsics"]).abs() / ref["intrinsics"].abs().clamp_min(1.0)).max().item(),
          "feat": _rel(o["depth_features"], ref["depth_features"]), "rays": (o["rays"] - ref["rays"]).abs().max().item()}
    print(tag, {k: f"{v:.2e}" for k, v in st.items()})
    for k in o:
        assert torch.isfinite(o[k]).all(), (tag, k)
    assert st["depth"] <= depth_bar and st["conf"] <= 2 * depth_bar and st["K"] <= 2e-3 and st["feat"] <= 3e-3 and st["rays"] <= 2e-3, (tag, st)
    return st


# ------------------------------------------------------------------------------------------- headline shapes, every image
def test_headline_bs8_every_image_vs_oracle(engine_cls):
    """BASELINE.json configs[1] (ViT-L/14, 518x518, bs=8): the full batch through the fp32 oracle (about 15 s of host time), every
    image held to the north_star bar, not just image 0."""
    cfg = synth.load_config("vitl14")
    sd = synth.make_synthetic_checkpoint(cfg, 125)
    g = torch.Generator().manual_seed(11)
    rgb = torch.randint(0, 256, (8, 3, 518, 518), dtype=torch.uint8, generator=g)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    out = model.infer(rgb.cuda())
    torch.cuda.synchronize()
    ref = restate.OracleV2(cfg, sd).infer(rgb)
    worst = 0.0
    for b in range(8):
        d = _arel(out["depth"][b].cpu(), ref["depth"][b])
        k = ((out["intrinsics"][b].cpu() - ref["intrinsics"][b]).abs() / ref["intrinsics"][b].abs().clamp_min(1.0)).max().item()
        worst = max(worst, d)
        assert d <= 1e-3 and k <= 2e-3, (b, d, k)
    print("bs8 worst per-image depth ARel", f"{worst:.2e}")
    _outputs_ok(out, ref, "vitl_518_bs8")


def test_vitl_644x966_bs2_vs_oracle(engine_cls):
    """BASELINE.json configs[4] shape (644x966 -> 644x952, 3128 tokens) at batch 2."""
    cfg = synth.load_config("vitl14")
    sd = synth.make_synthetic_checkpoint(cfg, 125)
    g = torch.Generator().manual_seed(13)
    rgb = torch.randint(0, 256, (2, 3, 644, 966), dtype=torch.uint8, generator=g)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    out = model.infer(rgb.cuda())
    torch.cuda.synchronize()
    ref = restate.OracleV2(cfg, sd).infer(rgb)
    _outputs_ok(out, ref, "vitl_644x966_bs2")


# ------------------------------------------------------------------------------------------- taps
FEATURE_TAPS = ["tokens0", "blocks.0.attn.qkv", "block0", "block5", "block11", "block17", "block23", "feat0", "feat1", "feat2", "feat3",
                "input_adapter.0", "input_adapter.3", "prompt_camera.0", "prompt_camera.1", "prompt_camera.2", "prompt_camera.3",
                "to_latents", "ups.0", "ups.1"]


def test_config5_mixed_list_bs32_vs_oracle(engine_cls):
    """BASELINE.json configs[4] at its stated size on one GPU: 16 x 644x966 + 16 x 518x518 through dist.infer_mixed (ViT-L/14).  Every
    output is checked for shape / finiteness, one image per shape against the fp32 oracle at the north-star bar, and the reversed list
    must return every image's bits unchanged (bucketing and batch position do not leak into an image's result)."""
    from unidepth_amd.dist import infer_mixed
    case = cases.CASES["vitl_518x518_b1"]
    cfg = synth.load_config(case["arch"])
    sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    g = torch.Generator().manual_seed(21)
    imgs = [torch.randint(0, 256, (3, 644, 966), dtype=torch.uint8, generator=g) for _ in range(16)] + \
           [torch.randint(0, 256, (3, 518, 518), dtype=torch.uint8, generator=g) for _ in range(16)]
    outs = infer_mixed(model, [im.cuda() for im in imgs], keys=("depth", "intrinsics", "confidence"), inflight=2)
    torch.cuda.synchronize()
    assert len(outs) == 32
    for im, o in zip(imgs, outs):
        assert o["depth"].shape == (1, im.shape[1], im.shape[2]) and torch.isfinite(o["depth"]).all() and (o["depth"] > 0).all()
        assert o["intrinsics"].shape == (3, 3) and torch.isfinite(o["confidence"]).all()
    orc = restate.OracleV2(cfg, sd)
    for i in (3, 20):
        ref = orc.infer(imgs[i][None])
        d = ((outs[i]["depth"].cpu() - ref["depth"][0]).abs() / ref["depth"][0]).mean().item()
        k = ((outs[i]["intrinsics"].cpu() - ref["intrinsics"][0]).abs() / ref["intrinsics"][0].abs().clamp_min(1.0)).max().item()
        print(f"config5 image {i} {tuple(imgs[i].shape[1:])}: depth ARel {d:.2e}, K {k:.2e}")
        assert d <= 1e-3 and k <= 2e-3, (i, d, k)
    outs_r = infer_mixed(model, [im.cuda() for im in imgs[::-1]], keys=("depth",), inflight=2)[::-1]
    torch.cuda.synchronize()
    for a, b in zip(outs, outs_r):
        assert torch.equal(a["depth"], b["depth"])


@pytest.mark.parametrize("arch,H,W,B,seed", [("vits14", 462, 616, 1, 123), ("vitl14", 518, 518, 2, 125)])
def test_taps_vs_oracle(engine_cls, arch, H, W, B, seed):
    """intermediate tensors of the engine against the oracle's, rel-L2 <= 3e-3 for encoder / decoder feature taps,
    <= 2e-3 relative on the 4 pinhole parameters, <= 2e-3 absolute (mean) on the pre-exp log-depth, then the 7 outputs."""
    cfg = synth.load_config(arch)
    sd = synth.make_synthetic_checkpoint(cfg, seed)
    g = torch.Generator().manual_seed(21)
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)
    orc = restate.OracleV2(cfg, sd)
    orc.keep_taps = True
    ref = orc.infer(rgb)
    rt = orc.taps
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    out, taps = model.infer_with_taps(rgb.cuda())
    torch.cuda.synchronize()
    plain = model.infer(rgb.cuda())                      # the segmented replay must not change anything
    for k in out:
        assert torch.equal(out[k], plain[k]), k
    depth = len([k for k in sd if k.endswith("attn.qkv.weight")])
    res = {}
    for name in FEATURE_TAPS:
        if name.startswith("block") and name[5:].isdigit() and int(name[5:]) >= depth:
            continue
        if name.startswith("feat"):
            j = int(name[4:])
            got = model.debug_taps()["features"][j]
            want = rt[name]
        elif name == "to_latents":
            got, want = out["depth_features"], rt[name]
        else:
            got, want = taps[name], rt[name]
        want = want.reshape(got.shape) if want.numel() == got.numel() else want
        res[name] = _rel(got.float(), want)
    # taps the engine keeps in normalised form (LayerNorm statistics only; the affine lives in the consumer's weights)
    res["rays_embedding_normed"] = _rel(taps["rays_embedding_normed"], F.layer_norm(rt["rays_embedding"], rt["rays_embedding"].shape[-1:], eps=1e-5))
    u2 = rt["ups.2"]
    res["ups.2_normed"] = _rel(taps["ups.2_normed"], F.layer_norm(u2.permute(0, 2, 3, 1), (u2.shape[1],), eps=1e-5).permute(0, 3, 1, 2))
    print(arch, {k: f"{v:.1e}" for k, v in res.items()})
    for name, v in res.items():
        # the ray embedding holds sin(angle * 2^k * pi) up to k = log2(max(h, w) / 2): an intrinsics error of 7e-4 (inside its 2e-3 bar)
        # moves the top band's phase by ~1e-2, so this tap inherits the camera head's error times the band frequency (the kernel
        # itself is held to 1e-3 on identical rays in tests/test_kernels_gpu.py::test_camera_rays_embed)
        bar = 1.5e-2 if name == "rays_embedding_normed" else 3e-3
        assert v <= bar, (name, v)
    kk = ((taps["intrinsics4"].cpu() - rt["intrinsics4"]).abs() / rt["intrinsics4"].abs()).max().item()
    assert kk <= 2e-3, kk
    dl = (taps["logdepth"].cpu() - rt["logdepth"]).abs()
    dc = (taps["logconf"].cpu() - rt["logconf"]).abs()
    print("logdepth |d| mean %.2e max %.2e   logconf mean %.2e" % (dl.mean().item(), dl.max().item(), dc.mean().item()))
    assert dl.mean().item() <= 2e-3 and dc.mean().item() <= 2e-3
    _outputs_ok(out, ref, arch + "_taps")


# ------------------------------------------------------------------------------------------- outliers
@pytest.mark.parametrize("arch,H,W", [("vits14", 462, 616), ("vitl14", 518, 518)])
def test_outlier_sensitised_checkpoint(engine_cls, arch, H, W):
    """Massive-activation channels (|x| ~ 300 in every token), heavy-tailed channels, one block with 16x attention logits, and a
    saturated input (oracle/synth.py make_outlier_checkpoint / outlier_image): the engine's fp16-stored activations must neither
    overflow nor lose the 1e-3 depth bar.  Real DINOv2 checkpoints have such channels; randn * fan_in^-1/2 ones do not."""
    cfg = synth.load_config(arch)
    sd = synth.make_outlier_checkpoint(cfg, 321)
    rgb = synth.outlier_image(1, H, W)
    orc = restate.OracleV2(cfg, sd)
    orc.keep_taps = True
    ref = orc.infer(rgb)
    last = max(int(k[5:]) for k in orc.taps if k.startswith("block") and k[5:].isdigit())
    xs = orc.taps[f"block{last}"]
    print("residual stream |x| max %.0f, std %.1f" % (xs.abs().max().item(), xs.std().item()))
    assert xs.abs().max().item() > 150.0                      # the stress is really there
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    out, taps = model.infer_with_taps(rgb.cuda(), names=[f"block{last}"])
    torch.cuda.synchronize()
    assert _rel(taps[f"block{last}"], xs) <= 3e-3
    _outputs_ok(out, ref, arch + "_outliers")


# ------------------------------------------------------------------------------------------- module seams
def test_module_seams_swap_halves_with_oracle(engine_cls):
    cfg = synth.load_config("vits14")
    sd = synth.make_synthetic_checkpoint(cfg, 123)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 252, 336, generator=g)           # network-resolution, already normalised image
    orc = restate.OracleV2(cfg, sd)
    feats, cls = orc.encode(x)
    ref = orc.decode(feats, cls, 252, 336)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    assert model.embed_dim == 384 and model.patch_size == 14 and len(model.embed_dims) == 12
    # engine encoder vs oracle encoder
    outs, toks = model.pixel_encoder(x.cuda())
    idx = [i - 1 for i in model.depths]
    assert all((outs[i] is None) == (i not in idx) for i in range(12))
    for j, i in enumerate(idx):
        assert outs[i].shape == feats[j].shape and toks[i].shape == cls[j].shape
        assert _rel(outs[i], feats[j]) <= 3e-3 and _rel(toks[i], cls[j]) <= 3e-3
    # ORACLE encoder output -> engine decoder
    dec = model.pixel_decoder({"image": x.cuda(), "features": [f.cuda() for f in feats], "tokens": [c.cuda() for c in cls]}, [])
    assert dec["rays"].shape == (2, 252 * 336, 3) and dec["radius"].shape == (2, 1, 252, 336)
    assert _arel(dec["radius"].cpu(), ref["radius"]) <= 1e-3
    assert _arel(dec["confidence"].cpu(), ref["confidence"]) <= 2e-3
    assert _rel(dec["depth_features"], ref["depth_features"]) <= 3e-3
    assert ((dec["intrinsics"].cpu() - ref["intrinsics"]).abs() / ref["intrinsics"].abs().clamp_min(1.0)).max().item() <= 2e-3
    # engine encoder output -> ORACLE decoder
    ef, ec = model.encode(x.cuda())
    ref2 = orc.decode([f.cpu() for f in ef], [c.cpu() for c in ec], 252, 336)
    assert _arel(ref2["radius"], ref["radius"]) <= 1e-3
    # export entries (export.py:27-45, :57-76): both halves at network resolution; GT rays replace the predicted ones
    from unidepth_amd.export import UniDepthV2ONNX, UniDepthV2ONNXcam
    m1 = UniDepthV2ONNX(cfg).load_state_dict(sd).to("cuda")
    pts, conf, K = m1.forward(x.cuda())
    rays_ref = ref["rays"]
    assert _rel(pts, rays_ref * ref["radius"]) <= 2e-3 and pts.shape == (2, 3, 252, 336)
    gt = F.normalize(torch.randn(2, 3, 252, 336, generator=g), dim=1)
    gt[:, 2] = gt[:, 2].abs() + 0.2
    gt = F.normalize(gt, dim=1)
    ref3 = orc.decode(feats, cls, 252, 336, rays_gt=gt)
    pts3, conf3, K3 = UniDepthV2ONNXcam(cfg).load_state_dict(sd).to("cuda").forward(x.cuda(), gt.cuda())
    assert _rel(pts3, gt * ref3["radius"]) <= 2e-3
    assert _arel(conf3.cpu(), ref3["confidence"]) <= 2e-3


# ------------------------------------------------------------------------------------------- post-processing modes, argument checks
def test_bicubic_interpolation_mode(engine_cls):
    """interpolation_mode='bicubic' (unidepthv2.py:80-89 passes it to F.interpolate(align_corners=False)) on a padded AND resized input."""
    case = cases.CASES["vits_375x1242_b1"]
    cfg = synth.load_config(case["arch"])
    sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
    rgb, _ = cases.case_inputs(case)
    orc = restate.OracleV2(cfg, sd)
    orc.interpolation_mode = "bicubic"
    ref = orc.infer(rgb)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    model.interpolation_mode = "bicubic"
    out = model.infer(rgb.cuda())
    torch.cuda.synchronize()
    _outputs_ok(out, ref, "vits_375x1242_bicubic")
    bil = restate.OracleV2(cfg, sd).infer(rgb)
    assert _arel(ref["depth"], bil["depth"]) > 1e-3          # the two modes really differ on this case
    model.interpolation_mode = "nearest"
    with pytest.raises(ValueError):
        model.infer(rgb.cuda())


def test_batch_camera_one_model_per_image(engine_cls):
    """A BatchCamera with one camera PER IMAGE, of different models (the reference's wrapper: unproject concatenates every member's own
    unproject, utils/camera.py:1166-1171; entered at unidepthv2.py:267-274): every image must get exactly the rays a single-camera call with its
    camera produces, and a depth within the cross-batch-size tolerance of that call; a wrapper whose members share one closed-form class
    takes the batched kernel and must agree with the [B,3,3] K tensor form bit for bit."""
    from unidepth_amd import cameras as C
    cfg = synth.load_config("vits14")
    sd = synth.make_synthetic_checkpoint(cfg, 19)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    g = torch.Generator().manual_seed(4)
    H, W = 300, 400
    rgb = torch.randint(0, 256, (4, 3, H, W), dtype=torch.uint8, generator=g).cuda()
    cams = [C.Pinhole(params=torch.tensor([[310.0, 305.0, 200.0, 150.0]])),
            C.EUCM(torch.tensor([[250.0, 251.0, 198.0, 152.0, 0.55, 1.05]])),
            C.OPENCV(torch.tensor([[300.0, 302.0, 201.0, 149.0, -0.2, 0.05, -0.01, 0, 0, 0, 1e-3, -1e-3, 0, 0, 0, 0]])),
            C.MEI(torch.tensor([[260.0, 262.0, 199.0, 151.0, -0.1, 0.02, 1e-3, -1e-3, 0.9]]))]
    out = model.infer(rgb, C.BatchCamera(cams))
    torch.cuda.synchronize()
    assert out["rays"].shape == (4, 3, H, W)
    for i, c in enumerate(cams):
        one = model.infer(rgb[i:i + 1], c)
        torch.cuda.synchronize()
        assert torch.equal(out["rays"][i], one["rays"][0]), i
        arel = ((out["depth"][i] - one["depth"][0]).abs() / one["depth"][0]).mean().item()
        assert arel < 2e-3, (i, arel)                                   # bs = 4 and bs = 1 plans differ in GEMM tile shapes, not in the camera
    # another ORDER of the same models: same plan (images are sorted by model inside infer(): no 2.6 GB plan per ordering), and,
    # infer() being batch-permutation equivariant, the same bits per image
    n_plans = len(model._plans)
    perm = [2, 0, 3, 1]
    outp = model.infer(rgb[perm], C.BatchCamera([cams[i] for i in perm]))
    torch.cuda.synchronize()
    assert len(model._plans) == n_plans
    for k in out:
        if out[k].shape[0] == 4:
            assert torch.equal(outp[k], out[k][perm]), k
    pins = [C.Pinhole(params=torch.tensor([[300.0 + 5 * i, 301.0, 200.0, 150.0]])) for i in range(4)]
    a = model.infer(rgb, C.BatchCamera(pins))
    b = model.infer(rgb, torch.cat([p.K for p in pins]).cuda())
    torch.cuda.synchronize()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    with pytest.raises(AssertionError):
        model.infer(rgb[:3], C.BatchCamera(cams))                       # four cameras, three images


def test_camera_batch_mismatch_is_rejected(engine_cls):
    cfg = synth.load_config("vits14")
    sd = synth.make_synthetic_checkpoint(cfg, 123)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    rgb = torch.zeros(4, 3, 252, 336, dtype=torch.uint8, device="cuda")
    K = torch.tensor(cases.DEMO_K).repeat(2, 1, 1)
    with pytest.raises(AssertionError):
        model.infer(rgb, K)                                   # 2 cameras for 4 images: neither broadcast nor per image


# ------------------------------------------------------------------------------------------- pipeline, warm state, temporaries
def test_pipeline_warm_state_with_temporaries(engine_cls):
    """with warm plans there is no host sync inside infer(), so the host runs ahead; inputs built as temporaries
    on the caller's stream must stay alive until the side stream has copied them (record_stream).  >= 3 distinct same-shape
    micro-batches per slot, each a temporary dropped right after submit()."""
    from unidepth_amd.pipeline import InferPipeline
    cfg = synth.load_config("vits14")
    sd = synth.make_synthetic_checkpoint(cfg, 31)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    g = torch.Generator().manual_seed(9)
    host = [torch.randint(0, 256, (2, 3, 252, 336), dtype=torch.uint8, generator=g) for _ in range(8)]
    refs = [{k: v.clone() for k, v in model.infer(h.cuda()).items()} for h in host]
    pipe = InferPipeline(model, depth=2)
    for h in host[:2]:
        pipe.submit(h.cuda())                                  # warm both slots
    pipe.sync()
    for rep in range(3):
        outs = []
        for h in host:
            tmp = torch.stack([h[0].cuda(), h[1].cuda()])      # temporary: its only reference dies at the end of this iteration
            outs.append(pipe.submit(tmp))
            del tmp
            junk = torch.full((2, 3, 252, 336), 7, dtype=torch.uint8, device="cuda")   # what the allocator would hand out next
            del junk
        for o in outs:
            pipe.wait(o)
        pipe.sync()
        for o, r in zip(outs, refs):
            for k in r:
                assert torch.equal(o[k], r[k]), (rep, k)
