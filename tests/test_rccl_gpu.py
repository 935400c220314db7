"""Multi-rank path on real hardware: `python bench.py --gpus 2` must launch two ranks itself and report n_gpus = 2 with an
`rccl` block.  With >= 2 visible GPUs this runs over RCCL (backend "nccl"); on a 1-GPU box the same launch
path is exercised with both ranks sharing cuda:0 and gloo as the exchange backend (functional only, never a measurement)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(extra_env):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-kernel-timing", "--arch", "vits14", "--size", "252", "336", "--batch", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_bench_two_ranks_rccl():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL over xGMI)")
    d = _run_bench({})
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4
    assert d["rccl"]["world_size"] == 2 and d["rccl"]["ranks_seen"] == [0, 1] and d["rccl"]["all_blocks_valid"]
    assert d["rccl"]["backend"].startswith("rccl") and d["rccl"]["gathered_bytes_per_rank_per_step"] == 2 * (2 * 252 * 336 + 9) * 4
    # two distinct devices, both forms of the exchange step ran and agree bit for bit, RCCL's own log was captured
    assert d["rccl"]["distinct_devices"] == 2 and d["rccl"]["algos_bit_identical"]
    assert set(d["rccl"]["gather_alone_ms_by_algo"]) == {"collective", "direct"}
    assert d["rccl"]["rccl_log"] and d["rccl"]["rccl_log"].get("lines", 0) > 0


def test_bench_self_launch_two_ranks_shared_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if torch.cuda.device_count() >= 2:
        pytest.skip("covered by the RCCL test on this box")
    d = _run_bench({"UD_BENCH_SHARE_GPU": "1", "UD_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["rccl"]["ranks_seen"] == [0, 1] and d["rccl"]["all_blocks_valid"]


def test_cabi_allgather_single_rank_communicator():
    """ud_rccl_* (include/unidepth_hip.h; SURVEY.md 8b `rccl_allgather_outputs`): unique id -> communicator of ONE rank on this GPU -> the
    all-gather of a packed output block, both forms (ncclAllGather and the all-pairs send / receive group), on a side stream; double init and
    a call after finalize are errors.  The multi-rank exchange itself needs >= 2 GPUs (test_bench_two_ranks_rccl)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ctypes as C
    from unidepth_amd import _lib
    L = _lib.lib
    uid = C.create_string_buffer(128)
    assert L.ud_rccl_unique_id(uid) == 0, L.ud_last_error()
    assert L.ud_rccl_init(uid, 1, 0) == 0, L.ud_last_error()
    try:
        assert L.ud_rccl_init(uid, 1, 0) != 0                                   # one communicator per process
        src = torch.randn(3, 1000, device="cuda")
        st = torch.cuda.Stream()
        for direct in (0, 1):
            dst = torch.zeros_like(src)
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                rc = L.ud_rccl_allgather_outputs(src.data_ptr(), dst.data_ptr(), src.numel() * 4, direct, st.cuda_stream)
            assert rc == 0, L.ud_last_error()
            st.synchronize()
            assert torch.equal(dst, src), direct
        assert L.ud_rccl_allgather_outputs(None, src.data_ptr(), 16, 0, None) != 0
    finally:
        assert L.ud_rccl_finalize() == 0
    assert L.ud_rccl_allgather_outputs(src.data_ptr(), src.data_ptr(), 16, 0, None) != 0   # no communicator any more


def test_dist_module_routes_gathers_through_the_cabi_communicator():
    """unidepth_amd.dist with the library's communicator (init_cabi_exchange: torch.distributed, here a one-rank gloo group, only carries the
    unique id): all_gather_batch of device tensors goes through ud_rccl_allgather_outputs for both exchange forms and returns its input; after
    finalize_cabi_exchange the torch.distributed route is back.  The two-rank form of the same call runs in test_bench_two_ranks_rccl."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import socket
    import torch.distributed as dist
    from unidepth_amd import dist as ud_dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        calls = []
        real = ud_dist._cabi_allgather
        ud_dist._cabi_allgather = lambda buf, mine, direct: (calls.append(bool(direct)), real(buf, mine, direct))[1]
        ud_dist.init_cabi_exchange()
        assert ud_dist.cabi_exchange_ready()
        x = torch.randn(5, 37, device="cuda")
        for algo in ("collective", "direct"):
            y = ud_dist.all_gather_batch(x, [5], algo=algo)
            torch.cuda.synchronize()
            assert torch.equal(x, y)
        assert calls == [False, True]
        ud_dist.finalize_cabi_exchange()
        assert not ud_dist.cabi_exchange_ready()
        ud_dist._cabi_allgather = real
    finally:
        ud_dist.finalize_cabi_exchange()
        dist.destroy_process_group()
