"""End-to-end parity of the HIP engine (through the C-ABI) against the CPU oracle and the reference's golden
vectors, on seeded sensitised checkpoints.  Bars (BASELINE.json north_star / SURVEY.md 8c):
  depth ARel <= 1e-3 vs the fp32 oracle; intrinsics max-rel <= 2e-3; depth_features rel-L2 <= 3e-3;
  confidence/radius/points ARel-type <= 2e-3; rays max-abs <= 2e-3.  (fp16 MFMA operands, fp32 accumulate/residual.)"""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cases, restate, synth  # noqa: E402


@pytest.fixture(scope="module")
def engine_cls():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from unidepth_amd import UniDepthV2
    return UniDepthV2


def _engine_camera(cam):
    """oracle/cases.py camera spec -> what the engine's infer() takes: K tensor (on the GPU) or a unidepth_amd.cameras object."""
    if cam is None:
        return None
    if isinstance(cam, tuple):
        from unidepth_amd import cameras
        return getattr(cameras, cam[0])(cam[1])
    return cam.cuda()


def _arel(a, b):
    return ((a - b).abs() / b.abs().clamp_min(1e-6)).mean().item()


def _check(out, ref, tag):
    o = {k: v.float().cpu() for k, v in out.items()}
    assert set(o) == set(ref)
    for k in ref:
        assert o[k].shape == ref[k].shape, (tag, k, o[k].shape, ref[k].shape)
        assert torch.isfinite(o[k]).all(), (tag, k)
    stats = {
        "depth_arel": _arel(o["depth"], ref["depth"]),
        "radius_arel": _arel(o["radius"], ref["radius"]),
        "conf_arel": _arel(o["confidence"], ref["confidence"]),
        "K_maxrel": ((o["intrinsics"] - ref["intrinsics"]).abs() / ref["intrinsics"].abs().clamp_min(1.0)).max().item(),
        "feat_rel": ((o["depth_features"] - ref["depth_features"]).norm() / ref["depth_features"].norm()).item(),
        "rays_maxabs": (o["rays"] - ref["rays"]).abs().max().item(),
        "points_rel": ((o["points"] - ref["points"]).norm() / ref["points"].norm()).item(),
    }
    print(tag, {k: f"{v:.2e}" for k, v in stats.items()})
    assert stats["depth_arel"] <= 1e-3, (tag, stats)
    assert stats["radius_arel"] <= 1e-3, (tag, stats)
    assert stats["conf_arel"] <= 2e-3, (tag, stats)
    assert stats["K_maxrel"] <= 2e-3, (tag, stats)
    assert stats["feat_rel"] <= 3e-3, (tag, stats)
    assert stats["rays_maxabs"] <= 2e-3, (tag, stats)
    assert stats["points_rel"] <= 2e-3, (tag, stats)
    return stats


@pytest.mark.parametrize("name", list(cases.CASES))
def test_infer_matches_oracle_and_golden(engine_cls, name, golden_dir):
    case = cases.CASES[name]
    cfg = synth.load_config(case["arch"])
    sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
    rgb, cam = cases.case_inputs(case)
    orc = restate.OracleV2(cfg, sd)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    if case.get("resolution_level") is not None:
        orc.resolution_level = model.resolution_level = case["resolution_level"]
    ref = orc.infer(rgb, cam)
    out = model.infer(rgb.cuda(), _engine_camera(cam))
    torch.cuda.synchronize()
    stats = _check(out, ref, name)
    if cam is not None:                                             # GT-camera rays are plain fp32 geometry, not network output
        assert stats["rays_maxabs"] <= 2e-5, (name, stats)
    # golden digest produced by the real reference (fixtures)
    got = cases.digest({k: v.float().cpu() for k, v in out.items()})
    want = np.load(os.path.join(golden_dir, name + ".npz"))
    d = np.abs(got["depth"] - want["depth"]) / want["depth"]
    assert d.mean() <= 1e-3, (name, d.mean())
    if cam is not None:
        assert np.abs(got["rays"] - want["rays"]).max() <= 2e-5, name
    # second call on the cached plan must reproduce the first bit-for-bit (no state leaks between calls)
    out2 = model.infer(rgb.cuda(), _engine_camera(cam))
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(out[k], out2[k]), k
    assert out["depth"].data_ptr() != out2["depth"].data_ptr()       # fresh, caller-owned outputs


def test_infer_headline_batch8_properties(engine_cls):
    """BASELINE.json configs[1] (ViT-L/14, 518x518, bs=8): too slow for a full CPU oracle pass in a unit test ->
    size-independent properties: batch permutation equivariance (images are independent), agreement of image 0 with the
    bs=1 oracle, geometric identities between outputs."""
    case = cases.CASES["vitl_518x518_b1"]
    cfg = synth.load_config(case["arch"])
    sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    g = torch.Generator().manual_seed(11)
    rgb = torch.randint(0, 256, (8, 3, 518, 518), dtype=torch.uint8, generator=g)
    rgb[0] = cases.case_inputs(case)[0][0]
    out = model.infer(rgb.cuda())
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4])
    outp = model.infer(rgb[perm].cuda())
    torch.cuda.synchronize()
    for k in out:
        assert torch.equal(out[k][perm], outp[k]), k                     # bit-exact: same kernels, same per-image order
    ref = restate.OracleV2(cfg, sd).infer(rgb[:1])
    assert _arel(out["depth"][:1].cpu(), ref["depth"]) <= 1e-3
    # identities: points = rays * radius (no resample here), depth = points_z, |rays| = 1
    assert (out["points"][:, 2:] - out["depth"]).abs().max() == 0
    assert ((out["points"].norm(dim=1, keepdim=True) - out["radius"]).abs() / out["radius"]).max() < 1e-5
    assert (out["rays"].norm(dim=1) - 1).abs().max() < 1e-5
    assert out["depth"].shape == (8, 1, 518, 518) and out["depth_features"].shape == (8, 512, 37, 37)


def test_resolution_level_and_float_input(engine_cls):
    case = cases.CASES["vits_462x616_b1"]
    cfg = synth.load_config(case["arch"])
    sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
    rgb, _ = cases.case_inputs(case)
    orc = restate.OracleV2(cfg, sd)
    orc.resolution_level = 2
    ref = orc.infer(rgb.float())
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    model.resolution_level = 2
    out = model.infer(rgb.float().cuda())
    torch.cuda.synchronize()
    _check(out, ref, "vits_level2_float")
    model.resolution_level = 11
    with pytest.raises(AssertionError):
        model.infer(rgb.cuda())


@pytest.mark.parametrize("arch,H,W,B,normalize,ndim3", [("vits14", 240, 320, 1, True, True),      # up-resized input, [3,H,W] call form
                                                      ("vits14", 462, 616, 2, False, False),    # caller-normalised float input
                                                      ("vits14", 640, 200, 1, True, False),     # tall: aspect padding left / right (60 / 60)
                                                      ("vits14", 70, 90, 2, True, False),       # tiny input, up-resized ~5.6x; passed as a CPU tensor
                                                      ("vitl14", 644, 966, 1, True, False)])    # BASELINE configs[4] shape (3128 tokens)
def test_infer_more_shapes_and_call_forms(engine_cls, arch, H, W, B, normalize, ndim3):
    cfg = synth.load_config(arch)
    sd = synth.make_synthetic_checkpoint(cfg, 31)
    g = torch.Generator().manual_seed(17)
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=g)
    if not normalize:
        rgb = (rgb.float() / 255.0 - 0.45) / 0.25
    x = rgb[0] if ndim3 else rgb
    ref = restate.OracleV2(cfg, sd).infer(x, None, normalize=normalize)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    out = model.infer(x if H < 100 else x.cuda(), None, normalize=normalize)     # the reference moves a CPU input to its device itself
    torch.cuda.synchronize()
    assert out["depth"].is_cuda and out["depth"].shape[-2:] == (H, W)
    _check(out, ref, f"{arch}_{H}x{W}_b{B}_norm{int(normalize)}")


def test_infer_mixed_shapes_single_gpu(engine_cls):
    """Mixed-resolution list (BASELINE configs[4] pattern, small model): bucketed micro-batches must reproduce per-image infer()
    -- exactly: the kernels' summation orders do not depend on batch size or batch position."""
    from unidepth_amd.dist import infer_mixed
    cfg = synth.load_config("vits14")
    sd = synth.make_synthetic_checkpoint(cfg, 31)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    g = torch.Generator().manual_seed(5)
    shapes = [(240, 320), (196, 252), (240, 320), (240, 320), (196, 252)]
    imgs = [torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g).cuda() for h, w in shapes]
    outs = infer_mixed(model, imgs, keys=("depth", "intrinsics", "confidence"), max_batch=2)
    torch.cuda.synchronize()
    assert len(outs) == len(imgs)
    for im, o in zip(imgs, outs):
        ref = model.infer(im)
        assert o["depth"].shape == (1, im.shape[1], im.shape[2])
        d = ((o["depth"] - ref["depth"][0]).abs() / ref["depth"][0]).mean().item()
        assert d <= 1e-5, d
        assert ((o["intrinsics"] - ref["intrinsics"][0]).abs() / ref["intrinsics"][0].abs().clamp_min(1.0)).max().item() <= 1e-5


def test_pipeline_two_calls_in_flight(engine_cls):
    """unidepth_amd.pipeline: calls on separate HIP streams with separate buffer slots reproduce sequential infer() bit for bit
    (shared read-only weights, no shared activation state)."""
    from unidepth_amd.pipeline import InferPipeline
    cfg = synth.load_config("vits14")
    sd = synth.make_synthetic_checkpoint(cfg, 31)
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    g = torch.Generator().manual_seed(9)
    batches = [torch.randint(0, 256, (2, 3, 252, 336), dtype=torch.uint8, generator=g).cuda() for _ in range(5)]
    refs = [model.infer(b) for b in batches]
    torch.cuda.synchronize()
    pipe = InferPipeline(model, depth=2)
    outs = [pipe.submit(b) for b in batches]
    for o in outs:
        pipe.wait(o)
    pipe.sync()
    for o, r in zip(outs, refs):
        for k in r:
            assert torch.equal(o[k], r[k]), k


def test_interrupted_replay_leaves_no_state_behind(engine_cls):
    """the LayerNorm fold hands row statistics from a producer launch (proj / fc2: partial sums, a ticket per row
    tile, the last arriver reduces) to the consumer launch after it (qkv / fc1).  A replay that stops BETWEEN the two -- an exception in
    a tap replay, a caller abandoning a call -- must not poison later replays of the same plan: every completed launch leaves its
    tickets at zero (atomicInc with the arrival count as bound) and every statistic a consumer reads is rewritten by the producer in
    front of it.  Here the plan of a folded ViT-L bs=8 call is (1) replayed in full, (2) replayed up to just after a producer, in the
    middle of the encoder, (3) replayed from the middle of the encoder to just after a later producer (consumers fed with stale
    statistics, result discarded), and then replayed in full again: the outputs must be the SAME BITS as (1).  (bs = 8: the fold needs all
    four block GEMMs on the large-tile kernel, which the picker grants from M = 11008 rows on for ViT-L, not at bs = 4.)"""
    case = cases.CASES["vitl_518x518_b1"]
    cfg = synth.load_config(case["arch"])
    sd = synth.make_synthetic_checkpoint(cfg, case["ckpt_seed"])
    model = engine_cls(cfg).load_state_dict(sd).to("cuda").eval()
    rgb = torch.randint(0, 256, (8, 3, 518, 518), dtype=torch.uint8, generator=torch.Generator().manual_seed(5)).cuda()
    out0 = model.infer(rgb)
    torch.cuda.synchronize()
    plan = next(reversed(model._plans.values()))
    tags = [m[1] for m in plan.prog.meta]
    # the fold must be on for this plan (large-tile kernel, bs >= 4 for ViT-L), else the test tests nothing
    assert plan.ln_fold and tags.count("enc.ln") <= 2, "the LayerNorm fold is expected to be active at bs=8 ViT-L (only block 0's norm1 stays a launch)"
    prods = [i for i, t in enumerate(tags) if t in ("enc.proj", "enc.fc2")]
    assert len(prods) == 48
    stop_a = prods[7] + 1          # just after the proj of block 3: its consumer (fc1) never runs
    start_b, stop_b = prods[20] + 2, prods[30] + 1
    plan.prog.run(0, stop_a)
    plan.prog.run(start_b, stop_b)       # starts on a consumer whose producer did not run in this replay
    torch.cuda.synchronize()
    assert int(plan.row_tickets.abs().sum()) == 0, "a completed launch must leave its tickets at zero"
    out1 = model.infer(rgb)
    torch.cuda.synchronize()
    for k in out0:
        assert torch.equal(out0[k], out1[k]), k


def test_camera_head_barrier_timeout_is_loud(engine_cls):
    """VERDICT r5 weak #9 / ADVICE r5 (medium): the one-launch camera head needs its grid co-resident.  With the barrier time-out forced
    (spin limit 1: every workgroup that is not the last to arrive gives up) the call must NOT return plausible numbers: its intrinsics, rays
    and depth are NaN, the NEXT infer() raises and names the cause, and the call after that runs the per-layer form and is right again."""
    cfg = synth.load_config("vits14")
    model = engine_cls(cfg).load_state_dict(synth.make_synthetic_checkpoint(cfg, 41)).to("cuda").eval()
    rgb = torch.randint(0, 256, (2, 3, 240, 320), dtype=torch.uint8, generator=torch.Generator().manual_seed(9)).cuda()
    good = {k: v.clone() for k, v in model.infer(rgb).items()}
    torch.cuda.synchronize()
    model.clear_plans()
    model._cam_spin_limit = 1
    bad = model.infer(rgb)
    torch.cuda.synchronize()
    plan = next(reversed(model._plans.values()))
    assert plan.cam_one_launch and int(plan.cam_sync[2]) == 1 and int(plan.cam_fail[0]) == 1
    assert torch.isnan(bad["intrinsics"]).any() and torch.isnan(bad["depth"]).all() and torch.isnan(bad["rays"]).all()
    with pytest.raises(RuntimeError, match="camera head timed out"):
        model.infer(rgb)
    model._cam_spin_limit = 0
    again = model.infer(rgb)
    torch.cuda.synchronize()
    plan = next(reversed(model._plans.values()))
    assert not plan.cam_one_launch and "cam.head" not in [m[1] for m in plan.prog.meta]
    k1, k2 = good["intrinsics"].double(), again["intrinsics"].double()
    assert ((k1 - k2).abs() / k2.abs().clamp_min(1.0)).max().item() < 1e-5
    assert _arel(good["depth"].float().cpu(), again["depth"].float().cpu()) < 1e-3


def test_camera_heads_of_three_requests_in_flight(engine_cls):
    """Three requests in flight on three HIP streams (pipeline slots), i.e. three 128-workgroup spinning grids that would not fit the 256 CUs
    together: the library orders camera-head launches of one device behind each other (ud_camera_head_f32), so none of them times out and every
    request returns what it returns alone."""
    from unidepth_amd.pipeline import InferPipeline
    cfg = synth.load_config("vits14")
    model = engine_cls(cfg).load_state_dict(synth.make_synthetic_checkpoint(cfg, 41)).to("cuda").eval()
    rgbs = [torch.randint(0, 256, (2, 3, 240, 320), dtype=torch.uint8, generator=torch.Generator().manual_seed(s)).cuda() for s in (1, 2, 3)]
    alone = [{k: v.clone() for k, v in model.infer(r).items()} for r in rgbs]
    torch.cuda.synchronize()
    pipe = InferPipeline(model, depth=3)
    t0 = time.perf_counter()
    for rep in range(8):
        outs = [pipe.submit(r) for r in rgbs]
    pipe.sync()
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 5.0, "a camera-head barrier must never wait for its time-out"
    for a, o in zip(alone, outs):
        for k in a:
            assert torch.equal(a[k], o[k]), k
    for plan in model._plans.values():
        assert plan.cam_sync[:3].tolist() == [0, 0, 0] and int(plan.cam_fail[0]) == 0


@pytest.mark.parametrize("arch,B,H,W", [("vits14", 2, 240, 320), ("vitl14", 3, 518, 518)])
def test_camera_branch_one_launch_and_per_layer_forms_agree(engine_cls, monkeypatch, arch, B, H, W):
    """The token adapters + CameraHead run as ONE launch (ud_camera_head_f32) where the kernel's limits allow it and as the 26 per-layer
    launches it replaces otherwise: the same fp32 arithmetic in another summation order.  Both forms of one model agree to fp32 round-off on
    the intrinsics (1e-5); depth, behind the ray embedding's 2^k pi bands and a sensitised decoder in fp16, moves by a few 1e-4 with them (well
    inside the 1e-3 bar).  The one-launch form leaves its barrier words zero."""
    from unidepth_amd import ops
    cfg = synth.load_config(arch)
    model = engine_cls(cfg).load_state_dict(synth.make_synthetic_checkpoint(cfg, 41)).to("cuda").eval()
    rgb = torch.randint(0, 256, (B, 3, H, W), dtype=torch.uint8, generator=torch.Generator().manual_seed(9)).cuda()
    one = {k: v.clone() for k, v in model.infer(rgb).items()}
    torch.cuda.synchronize()
    plan = next(reversed(model._plans.values()))
    assert [m[1] for m in plan.prog.meta].count("cam.head") == 1 and plan.cam_sync[:3].tolist() == [0, 0, 0]
    monkeypatch.setattr(ops, "camera_head_supported", lambda d: False)
    model.clear_plans()
    per = model.infer(rgb)
    torch.cuda.synchronize()
    plan = next(reversed(model._plans.values()))
    assert "cam.head" not in [m[1] for m in plan.prog.meta]
    k1, k2 = one["intrinsics"].double(), per["intrinsics"].double()
    assert ((k1 - k2).abs() / k2.abs().clamp_min(1.0)).max().item() < 1e-5
    assert _arel(one["depth"].float().cpu(), per["depth"].float().cpu()) < 1e-3
