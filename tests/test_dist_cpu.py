"""world_size-2 gloo tests (CPU) of the data-parallel wrapper: sharding, uneven shards, gather order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeModel:
    """Stands in for the engine: outputs are a deterministic function of each image only (like the real model)."""

    def infer(self, rgb, camera=None):
        x = rgb.float()
        B = x.shape[0]
        return {"depth": x.mean(dim=1, keepdim=True) + 1.0, "confidence": x[:, :1] * 2.0,
                "intrinsics": x.reshape(B, -1)[:, :9].reshape(B, 3, 3).clone(), "rays": x[:1]}


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, B, q, algo=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unidepth_amd.dist import infer_data_parallel, shard_bounds
    g = torch.Generator().manual_seed(0)
    rgb = torch.randint(0, 256, (B, 3, 6, 5), dtype=torch.uint8, generator=g)
    out = infer_data_parallel(_FakeModel(), rgb, keys=("depth", "intrinsics"), gather_algo=algo)
    ref = _FakeModel().infer(rgb)
    ok = all(torch.equal(out[k], ref[k]) for k in ("depth", "intrinsics")) and out["depth"].shape[0] == B
    bounds = shard_bounds(B, world)
    ok = ok and bounds[0][0] == 0 and bounds[-1][1] == B and all(b[1] == nb[0] for b, nb in zip(bounds, bounds[1:]))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 5, 1])
def test_data_parallel_gather_gloo(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.parametrize("world,B", [(2, 5), (3, 7)])
def test_data_parallel_direct_all_pairs_gather_gloo(world, B):
    """The all-pairs form of the exchange step (world-1 sends + receives in one group instead of all_gather_into_tensor; what a
    fully-connected xGMI node wants) returns the same bits, incl. uneven shards and an odd world size."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q, "direct")) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def _worker_local(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unidepth_amd.dist import infer_data_parallel, shard_bounds
    g = torch.Generator().manual_seed(0)
    rgb = torch.randint(0, 256, (B, 3, 6, 5), dtype=torch.uint8, generator=g)
    lo, hi = shard_bounds(B, world)[rank]
    ref = infer_data_parallel(_FakeModel(), rgb, keys=("depth", "intrinsics", "confidence"))
    out = infer_data_parallel(_FakeModel(), keys=("depth", "intrinsics", "confidence"), rgb_local=rgb[lo:hi].clone(), n_images=B)   # only this rank's images
    ok = all(torch.equal(out[k], ref[k]) for k in ref) and out["depth"].shape[0] == B
    try:
        infer_data_parallel(_FakeModel(), rgb_local=torch.zeros(hi - lo + 1, 3, 6, 5, dtype=torch.uint8), n_images=B)     # wrong shard size: a clear error, on every rank alike
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 8), (2, 5), (3, 2)])
def test_data_parallel_local_shards_equal_global_batch_gloo(world, B):
    """infer_data_parallel(rgb_local=, n_images=): every rank hands over ONLY its shard (no redundant copy of the other ranks' images) and
    gets the same gathered outputs as with the global batch on every rank -- even shards, uneven shards, and a rank with no image at all."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_local, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def _worker_cabi(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes as C
    from unidepth_amd import _lib
    from unidepth_amd import dist as dmod
    g = torch.Generator().manual_seed(0)
    rgb = torch.randint(0, 256, (B, 3, 6, 5), dtype=torch.uint8, generator=g)
    keys = ("depth", "intrinsics", "confidence")
    ref = {a: dmod.infer_data_parallel(_FakeModel(), rgb, keys=keys, gather_algo=a) for a in ("collective", "direct")}     # torch.distributed route
    # the library's communicator, with RCCL replaced by stand-ins on this GPU-less box: the unique id must reach every rank through the
    # torch.distributed bootstrap, and the exchange must be called once per gather with this rank's padded block
    seen = {"uid": None, "init": None, "calls": []}

    def fake_unique_id(buf):
        C.memmove(buf, bytes(range(128)), 128)
        return 0

    def fake_init(uid, w, r):
        seen["uid"], seen["init"] = bytes(uid[:128]) if isinstance(uid, (bytes, bytearray)) else bytes(C.string_at(uid, 128)), (w, r)
        return 0

    def fake_allgather(buf, mine, direct):
        seen["calls"].append((mine.numel() * mine.element_size(), bool(direct), mine.is_contiguous()))
        dist.all_gather(list(buf.chunk(world)), mine)

    _lib.lib.ud_rccl_unique_id, _lib.lib.ud_rccl_init, _lib.lib.ud_rccl_finalize = fake_unique_id, fake_init, lambda: 0
    dmod._cabi_allgather, dmod._on_device = fake_allgather, lambda t: True
    assert not dmod.cabi_exchange_ready()
    dmod.init_cabi_exchange()
    ok = dmod.cabi_exchange_ready() and seen["uid"] == bytes(range(128)) and seen["init"] == (world, rank)
    for a in ("collective", "direct"):
        seen["calls"].clear()
        out = dmod.infer_data_parallel(_FakeModel(), rgb, keys=keys, gather_algo=a)
        ok = ok and all(torch.equal(out[k], ref[a][k]) for k in keys) and out["depth"].shape[0] == B
        per = -(-B // world)
        row_bytes = (6 * 5 + 9 + 6 * 5) * 4                  # depth + intrinsics + confidence packed per image, fp32
        ok = ok and seen["calls"] == [(per * row_bytes, a == "direct", True)]
    dmod.finalize_cabi_exchange()
    ok = ok and not dmod.cabi_exchange_ready()
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 8), (2, 5), (3, 7)])
def test_data_parallel_cabi_exchange_route_equals_torch_route_gloo(world, B):
    """with the library's communicator initialised (dist.init_cabi_exchange: torch.distributed carries only the unique id)
    the gathers go through ud_rccl_allgather_outputs -- ONE call per exchange step with this rank's padded packed block -- and return the
    same bits as the torch.distributed route, for both exchange forms, even and uneven shards.  (RCCL itself is replaced by a gloo stand-in
    here; the real entry runs in tests/test_rccl_gpu.py.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cabi, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]


def test_shard_bounds():
    from unidepth_amd.dist import shard_bounds
    assert shard_bounds(64, 8) == [(i * 8, i * 8 + 8) for i in range(8)]
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(1, 2) == [(0, 1), (1, 1)]


class _FakeShapeModel:
    """Engine stand-in for mixed-shape lists: per-image outputs, depth at the input resolution."""
    shape_constraints = {"ratio_bounds": [0.5, 2.5], "pixels_min": 200000.0, "pixels_max": 600000.0}

    def __init__(self):
        self.calls = []

    def infer(self, rgb, camera=None):
        x = rgb.float()
        B = x.shape[0]
        self.calls.append(tuple(x.shape))
        return {"depth": x.mean(dim=1, keepdim=True) + 1.0, "confidence": x[:, :1] * 2.0,
                "intrinsics": x.reshape(B, -1)[:, :9].reshape(B, 3, 3).clone()}


def _mixed_images():
    g = torch.Generator().manual_seed(3)
    shapes = [(6, 5), (4, 7), (6, 5), (6, 5), (4, 7), (8, 8), (6, 5)]
    return [torch.randint(0, 256, (3, h, w), dtype=torch.uint8, generator=g) for h, w in shapes]


def _worker_mixed(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unidepth_amd.dist import infer_mixed
    imgs = _mixed_images()
    model = _FakeShapeModel()
    out = infer_mixed(model, imgs, keys=("depth", "intrinsics"), max_batch=2)
    ok = len(out) == len(imgs)
    for im, o in zip(imgs, out):
        ref = _FakeShapeModel().infer(im[None])
        ok = ok and torch.equal(o["depth"], ref["depth"][0]) and torch.equal(o["intrinsics"], ref["intrinsics"][0])
    ok = ok and all(c[0] <= 2 for c in model.calls) and 0 < len(model.calls) < len(imgs)      # shared the work, batched by shape
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_infer_mixed_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mixed, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_plan_mixed_balance_and_single_process():
    from unidepth_amd.dist import infer_mixed, plan_mixed
    shapes = [(644, 966)] * 16 + [(518, 518)] * 16                   # BASELINE.json configs[4]
    costs = [2.7] * 16 + [1.0] * 16
    micro, owner = plan_mixed(shapes, costs, 8, 8)
    load = [0.0] * 8
    for (s, idx), r in zip(micro, owner):
        assert len({shapes[i] for i in idx}) == 1 and len(idx) <= 8
        load[r] += sum(costs[i] for i in idx)
    assert sorted(i for _, idx in micro for i in idx) == list(range(32))
    assert max(load) <= 1.15 * sum(costs) / 8                         # within 15 % of a perfectly even split
    micro1, owner1 = plan_mixed(shapes, costs, 1, 8)                  # one GPU: full micro-batches, bucketed
    assert [len(idx) for _, idx in micro1] == [8, 8, 8, 8] and set(owner1) == {0}
    imgs = _mixed_images()
    model = _FakeShapeModel()
    out = infer_mixed(model, imgs, keys=("depth", "intrinsics"), max_batch=8)        # no process group: runs everything here
    for im, o in zip(imgs, out):
        ref = _FakeShapeModel().infer(im[None])
        assert torch.equal(o["depth"], ref["depth"][0])
    assert sorted(model.calls) == sorted([(4, 3, 6, 5), (2, 3, 4, 7), (1, 3, 8, 8)])


def test_infer_mixed_camera_objects_run_alone():
    """K tensors are batched per shape bucket; an image with a camera OBJECT gets its own call (one camera per infer() call for
    the object models, as in the reference) and receives that object."""
    from unidepth_amd.dist import infer_mixed, plan_mixed

    class _Cam:                                       # stands in for a unidepth_amd.cameras / reference camera object
        pass

    class _Model(_FakeShapeModel):
        def __init__(self):
            super().__init__()
            self.cams = []

        def infer(self, rgb, camera=None):
            self.cams.append(camera)
            return super().infer(rgb)

    imgs = _mixed_images()                            # shapes: (6,5) x4, (4,7) x2, (8,8) x1
    K = torch.eye(3)
    obj = _Cam()
    cams = [K, None, K, obj, None, None, K]           # image 3 has an object; image 1 (no camera) keeps bucket (4,7) camera-free
    model = _Model()
    out = infer_mixed(model, imgs, cameras=cams, keys=("depth",), max_batch=8)
    assert len(out) == len(imgs) and all(o["depth"].shape[-2:] == im.shape[-2:] for o, im in zip(out, imgs))
    assert sum(1 for c in model.cams if c is obj) == 1
    solo_calls = [c for c, sh in zip(model.cams, model.calls) if c is obj]
    assert len(solo_calls) == 1 and (1, 3, 6, 5) in model.calls
    # bucket (6,5) minus the solo image = images 0, 2, 6: all carry K -> stacked [3,3,3]; bucket (4,7) has no cameras
    stacked = [c for c in model.cams if isinstance(c, torch.Tensor)]
    assert len(stacked) == 1 and stacked[0].shape == (3, 3, 3)
    micro, owner = plan_mixed([(6, 5)] * 3, [1.0] * 3, 2, 8, solo=[1])
    assert sorted(map(tuple, (idx for _, idx in micro))) == [(0, 2), (1,)] or sorted(len(idx) for _, idx in micro) == [1, 1, 1]


def test_plan_mixed_keeps_k_and_cameraless_images_apart():
    """A shape bucket with some K tensors and some camera-less images must not silently drop the intrinsics (round-1 bug):
    the two kinds run as separate infer() calls."""
    from unidepth_amd.dist import infer_mixed, plan_mixed

    class _Model(_FakeShapeModel):
        def __init__(self):
            super().__init__()
            self.cams = []

        def infer(self, rgb, camera=None):
            self.cams.append(None if camera is None else camera.clone())
            return super().infer(rgb)

    imgs = _mixed_images()                            # shapes: (6,5) x4 (images 0, 2, 3, 6), (4,7) x2, (8,8) x1
    K = [torch.eye(3) * (i + 1) for i in range(len(imgs))]
    cams = [K[0], None, None, K[3], None, None, None]            # bucket (6,5): images 0, 3 have K; images 2, 6 do not
    model = _Model()
    out = infer_mixed(model, imgs, cameras=cams, keys=("depth",), max_batch=8)
    assert len(out) == len(imgs)
    with_cam = [(c, sh) for c, sh in zip(model.cams, model.calls) if c is not None]
    assert len(with_cam) == 1 and with_cam[0][1] == (2, 3, 6, 5)
    assert torch.equal(with_cam[0][0], torch.stack([K[0], K[3]]))                 # both intrinsics arrive, in image order
    assert (2, 3, 6, 5) in [sh for c, sh in zip(model.cams, model.calls) if c is None]   # the camera-less half of the bucket
    micro, _ = plan_mixed([(6, 5)] * 4, [1.0] * 4, 1, 8, with_k=[0, 3])
    assert sorted(tuple(idx) for _, idx in micro) == [(0, 3), (1, 2)]
    with pytest.raises(ValueError):
        infer_mixed(model, imgs, cameras=cams[:3], keys=("depth",))


class _FakeRayModel:
    """One GT camera for the whole batch -> `rays` has batch 1 whatever the shard size (decoder.py:400), like the engine."""

    def infer(self, rgb, camera=None):
        x = rgb.float()
        B = x.shape[0]
        nr = 1 if (camera is not None and camera.reshape(-1, 3, 3).shape[0] == 1) else B
        return {"depth": x.mean(dim=1, keepdim=True) + 1.0, "intrinsics": x.reshape(B, -1)[:, :9].reshape(B, 3, 3).clone(),
                "rays": torch.full((nr, 3, 6, 5), 0.25)}


def _worker_rays(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unidepth_amd.dist import infer_data_parallel
    g = torch.Generator().manual_seed(0)
    rgb = torch.randint(0, 256, (3, 3, 6, 5), dtype=torch.uint8, generator=g)          # B=3, world=2 -> shards of 2 and 1 image
    K = torch.eye(3)[None]
    out = infer_data_parallel(_FakeRayModel(), rgb, camera=K, keys=("depth", "rays", "intrinsics"))
    ref = _FakeRayModel().infer(rgb, K)
    ok = all(torch.equal(out[k], ref[k]) for k in ("depth", "rays", "intrinsics"))
    out2 = infer_data_parallel(_FakeRayModel(), rgb, camera=None, keys=("rays", "depth"))   # per-image rays: gathered like the rest
    ok = ok and out2["rays"].shape[0] == 3 and torch.equal(out2["depth"], ref["depth"])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_single_camera_rays_uneven_shards_gloo():
    """with one GT camera and shards of 2 + 1 images the ranks used to disagree on whether `rays` joins the packed
    collective (decision taken from the LOCAL shard size) -> mismatched collectives.  Now decided from the call's arguments."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rays, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_gpus_flag_is_not_a_noop():
    """`python bench.py --gpus N` must either run N ranks or fail loudly: on a box with fewer GPUs it prints an error line and
    exits non-zero instead of silently benchmarking one rank."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "UD_BENCH_SHARE_GPU")}
    if torch.cuda.device_count() >= 2:
        pytest.skip("a multi-GPU box runs the real thing (tests/test_rccl_gpu.py)")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert "error" in json.loads(line)
