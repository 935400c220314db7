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


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from unidepth_amd.dist import infer_data_parallel, shard_bounds
    g = torch.Generator().manual_seed(0)
    rgb = torch.randint(0, 256, (B, 3, 6, 5), dtype=torch.uint8, generator=g)
    out = infer_data_parallel(_FakeModel(), rgb, keys=("depth", "intrinsics"))
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


def test_shard_bounds():
    from unidepth_amd.dist import shard_bounds
    assert shard_bounds(64, 8) == [(i * 8, i * 8 + 8) for i in range(8)]
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(1, 2) == [(0, 1), (1, 1)]
