"""Batch data-parallel infer() across the GPUs of one node: one process per GPU, images are independent (no
cross-sample op anywhere on the path), weights replicated, ONE exchange step -- an all-gather of the requested outputs
over RCCL/xGMI (`backend="nccl"` is RCCL on ROCm).  The reference has no distributed inference path; its own helper for
variable-length gathers (unidepth/utils/distributed.py:153-176: size all-gather -> pad -> gather -> trim) is the pattern
followed for uneven shards."""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_images: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous split, ceil(n/world) per rank, trailing ranks may be short or empty."""
    per = -(-n_images // world)
    return [(min(r * per, n_images), min((r + 1) * per, n_images)) for r in range(world)]


# How the one exchange step is issued.  xGMI on an MI355X node is a full mesh of point-to-point links (7 x ~153 GB/s per GPU, no
# switch): a RING all-gather forwards every block over world-1 hops, each hop bound by ONE link, so an S-byte-per-rank gather costs
# ~ (world-1) S / 153 GB/s; the DIRECT form -- every rank sends its block to each peer over that peer's own link, all links busy at
# once -- costs ~ S / 153 GB/s (0.65 ms against 4.6 ms for 100 MB).  "collective" leaves the choice to RCCL
# (all_gather_into_tensor: ring / tree / its own direct kernels per message size); "direct" spells the all-pairs exchange out as
# world-1 sends + world-1 receives inside one ncclGroupStart / End (dist.batch_isend_irecv) -- no copy through an intermediate rank.
GATHER_ALGOS = ("collective", "direct")
DEFAULT_GATHER_ALGO = "collective"


def all_gather_direct(out: torch.Tensor, mine: torch.Tensor, group=None) -> None:
    """out [world * n, ...] <- every rank's `mine` [n, ...] (same shape on all ranks), as an all-pairs send / receive group."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = mine.shape[0]
    out[rank * n:(rank + 1) * n].copy_(mine)
    ops = []
    for k in range(1, world):                          # peer order rotated per rank: every step of the group pairs distinct links
        dst, src = (rank + k) % world, (rank - k) % world
        ops.append(dist.P2POp(dist.isend, mine, dist.get_global_rank(group, dst) if group is not None else dst, group))
        ops.append(dist.P2POp(dist.irecv, out[src * n:(src + 1) * n], dist.get_global_rank(group, src) if group is not None else src, group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()


# ---- the exchange step behind the C-ABI (include/unidepth_hip.h ud_rccl_*): when the library's communicator exists, the gathers of this module
# run through ud_rccl_allgather_outputs -- the same entry a non-Python host binds -- and torch.distributed is the bootstrap only (it carries the
# 128-byte unique id from rank 0 to the others).  One communicator per process (csrc/rccl.cpp), spanning the default group.
_cabi = {"world": 0, "rank": -1}


def init_cabi_exchange(group=None) -> None:
    """Create the library's RCCL communicator over the ranks of `group` (default: the world).  The calling thread's current HIP device is the
    one the communicator binds to.  torch.distributed (any backend) only broadcasts the unique id."""
    import ctypes as C
    from ._lib import check, lib
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    uid = C.create_string_buffer(128)
    if rank == 0:
        check(lib.ud_rccl_unique_id(uid), "ud_rccl_unique_id")
    box = [bytes(uid.raw) if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    check(lib.ud_rccl_init(box[0], world, rank), "ud_rccl_init")
    _cabi.update(world=world, rank=rank)


def finalize_cabi_exchange() -> None:
    from ._lib import check, lib
    if _cabi["world"]:
        check(lib.ud_rccl_finalize(), "ud_rccl_finalize")
    _cabi.update(world=0, rank=-1)


def cabi_exchange_ready(group=None) -> bool:
    return _cabi["world"] > 0 and _cabi["world"] == dist.get_world_size(group) and _cabi["rank"] == dist.get_rank(group)


def _on_device(t: torch.Tensor) -> bool:
    return t.is_cuda


def _cabi_allgather(buf: torch.Tensor, mine: torch.Tensor, direct: bool) -> None:
    """buf [world * n, ...] <- every rank's contiguous `mine` [n, ...] through ud_rccl_allgather_outputs on the current HIP stream."""
    from ._lib import check, lib
    check(lib.ud_rccl_allgather_outputs(mine.data_ptr(), buf.data_ptr(), mine.numel() * mine.element_size(), int(direct),
                                        torch.cuda.current_stream(mine.device).cuda_stream), "ud_rccl_allgather_outputs")


def all_gather_batch(t: torch.Tensor, counts: List[int], group=None, algo: Optional[str] = None) -> torch.Tensor:
    """All-gather tensors whose dim 0 differs per rank (counts[r] rows on rank r): pad to max, one exchange (`algo`: see GATHER_ALGOS),
    trim.  Device tensors go through the library's own communicator when it exists (init_cabi_exchange: the C-ABI exchange step),
    otherwise through torch.distributed on whatever backend the group has (RCCL on GPUs, gloo on CPU for the tests)."""
    algo = algo or DEFAULT_GATHER_ALGO
    if algo not in GATHER_ALGOS:
        raise ValueError(f"gather algo {algo!r}: one of {GATHER_ALGOS}")
    world = dist.get_world_size(group)
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    buf = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if mx == 0:
        return buf
    if _on_device(t) and cabi_exchange_ready(group):
        _cabi_allgather(buf, pad, algo == "direct")
    elif algo == "direct":
        all_gather_direct(buf, pad, group)
    else:
        dist.all_gather_into_tensor(buf, pad, group=group)
    if all(c == mx for c in counts):
        return buf
    return torch.cat([buf[r * mx: r * mx + counts[r]] for r in range(world)], dim=0)


def _camera_count(camera) -> int:
    """Number of cameras an infer() `camera` argument carries: 0 = none, K tensor [..., 3, 3] -> its batch, camera object -> rows
    of `.params` (a BatchCamera-like wrapper -> its `.cameras`)."""
    if camera is None:
        return 0
    if isinstance(camera, torch.Tensor):
        return int(camera.reshape(-1, 3, 3).shape[0])
    if getattr(camera, "cameras", None):
        return len(camera.cameras)
    params = getattr(camera, "params", None)
    return int(params.shape[0]) if params is not None and getattr(params, "ndim", 1) > 1 else 1


def infer_data_parallel(model, rgb: Optional[torch.Tensor] = None, camera=None, keys: Optional[Iterable[str]] = ("depth", "confidence", "intrinsics"),
                        group=None, gather_algo: Optional[str] = None, *, rgb_local: Optional[torch.Tensor] = None, n_images: Optional[int] = None,
                        **kw) -> Dict[str, torch.Tensor]:
    """Rank r runs infer() on its contiguous shard of a global batch of B images (shard_bounds) and all ranks return the gathered global
    outputs for `keys` (None = all seven).  Two ways to hand over the images:
      * `rgb` [B,3,H,W]: every rank passes the SAME global batch (any device) and slices its shard out of it;
      * `rgb_local` [b_r,3,H,W] + `n_images` = B: every rank passes ONLY its own shard (b_r = the size shard_bounds gives rank r; a rank
        without images passes [0,3,H,W]) -- no redundant host-to-device copy of the other ranks' images (8x at 8 GPUs).
    A single camera broadcasts; a per-image camera batch [B,3,3] is global in both forms (it is tiny) and sharded like the images."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if (rgb is None) == (rgb_local is None):
        raise ValueError("infer_data_parallel: pass either the global batch `rgb` or this rank's shard `rgb_local` (+ n_images)")
    if rgb_local is not None and n_images is None:
        raise ValueError("infer_data_parallel: rgb_local needs n_images (the size of the global batch)")
    B = rgb.shape[0] if rgb is not None else int(n_images)
    bounds = shard_bounds(B, world)
    lo, hi = bounds[rank]
    counts = [b - a for a, b in bounds]
    if rgb_local is not None:
        if rgb_local.ndim != 4 or rgb_local.shape[0] != hi - lo:
            raise ValueError(f"infer_data_parallel: rank {rank} holds images [{lo}, {hi}) of {B}: rgb_local must be [{hi - lo},3,H,W], "
                             f"got {tuple(rgb_local.shape)}")
        mine = rgb_local
    else:
        mine = rgb[lo:hi]
    cam = camera
    per_image_cam = isinstance(camera, torch.Tensor) and camera.ndim == 3 and camera.shape[0] == B and B > 1
    if per_image_cam:
        cam = camera[lo:hi]
    if hi > lo:
        out = model.infer(mine, cam, **kw)
    else:                                  # empty shard: run one (blank, if the images live elsewhere) image to learn shapes, contribute zero rows
        probe = rgb[:1] if rgb is not None else torch.zeros((1,) + tuple(rgb_local.shape[1:]), dtype=rgb_local.dtype, device=rgb_local.device)
        out = {k: v[:0] for k, v in model.infer(probe, camera[:1] if per_image_cam else camera, **kw).items()}
    keys = list(out.keys()) if keys is None else list(keys)
    res = {}
    # a single GT camera (one K / one camera object for B > 1 images) yields ONE ray map, identical on every rank: it is not
    # exchanged.  The decision must be the same on every rank (it changes the packed width and the number of collectives), so it
    # is taken from the call's arguments, never from the local shard size.
    n_cam = _camera_count(camera)
    rays_shared = n_cam == 1 and B > 1
    packable = [k for k in keys if not (k == "rays" and rays_shared)]
    if len(packable) > 1 and all(out[k].dtype == out[packable[0]].dtype for k in packable):
        # ONE collective for all requested outputs: per-image rows are concatenated, gathered, and split again
        widths = [int(torch.Size(out[k].shape[1:]).numel()) for k in packable]
        packed = torch.cat([out[k].reshape(out[k].shape[0], wdt) for k, wdt in zip(packable, widths)], dim=1)
        g = all_gather_batch(packed.contiguous(), counts, group, gather_algo)
        off = 0
        for k, wdt in zip(packable, widths):
            res[k] = g[:, off:off + wdt].reshape((g.shape[0],) + tuple(out[k].shape[1:]))
            off += wdt
        keys = [k for k in keys if k not in packable]
    for k in keys:
        t = out[k]
        if k == "rays" and rays_shared:
            if t.shape[0] == 0:            # empty shard: the shared ray map comes from a one-image probe (the rays depend on the camera only)
                probe = rgb[:1] if rgb is not None else torch.zeros((1,) + tuple(rgb_local.shape[1:]), dtype=rgb_local.dtype, device=rgb_local.device)
                t = model.infer(probe, camera, **kw)["rays"]
            res[k] = t[:1]                 # identical on every rank, nothing to exchange
            continue
        res[k] = all_gather_batch(t.contiguous(), counts, group, gather_algo)
    return res


# --------------------------------------------------------------------------------------------------------------------
# Mixed-resolution batches (BASELINE.json configs[4]: e.g. 16 x 644x966 + 16 x 518x518 over 8 GPUs).  One infer() call takes
# one image shape (as in the reference, whose infer() cannot mix shapes either), so a list of images is bucketed by shape,
# cut into micro-batches and the micro-batches are spread over the ranks by estimated cost: a 3128-token image costs ~2.7x a
# 1369-token one, a contiguous split of the list would leave ranks idle.
# --------------------------------------------------------------------------------------------------------------------
def image_cost(model, H: int, W: int) -> float:
    """Relative cost of one image = algorithmic FLOP of encoder + decoder at the network resolution the shape policy picks
    (SURVEY.md 8(d) model: per encoder layer 24 N D^2 + 4 N^2 D; decoder ~ 0.35 of the encoder at 1369 tokens, linear in N)."""
    from .unidepthv2 import get_paddings, get_resize_factor
    sc = model.shape_constraints
    _, (Hp, Wp) = get_paddings((H, W), sc["ratio_bounds"])
    bounds = model._pixels_bounds() if hasattr(model, "_pixels_bounds") else (sc["pixels_min"], sc["pixels_max"])
    _, (Hn, Wn) = get_resize_factor((Hp, Wp), bounds)
    n = (Hn // 14) * (Wn // 14) + 1
    arch = getattr(model, "_arch", None) or {"D": 1024, "depth": 24}
    D, depth = arch["D"], arch["depth"]
    enc = depth * (24.0 * n * D * D + 4.0 * n * n * D)
    dec = 258.0e6 * n * (D / 1024.0)
    return enc + dec


def plan_mixed(shapes: List[Tuple[int, int]], costs: List[float], world: int, max_batch: int = 8, solo: Iterable[int] = (),
               with_k: Iterable[int] = ()):
    """Deterministic plan (identical on every rank): bucket image indices by (H, W), cut buckets into micro-batches of at most
    `max_batch` images, assign micro-batches to ranks longest-first onto the least-loaded rank (ties -> lowest rank).
    Images listed in `solo` (those with a camera OBJECT: one camera per infer() call, as in the reference) get a micro-batch
    of their own; images listed in `with_k` (a [3,3] K tensor each) are bucketed apart from the camera-less images of the same
    shape, because one infer() call either takes intrinsics for all of its images or predicts them for all.
    Returns (micro_batches, owner): micro_batches[j] = (shape, [image indices]); owner[j] = rank."""
    solo = set(solo)
    with_k = set(with_k)
    buckets: Dict[Tuple[int, int, int], List[int]] = {}
    for i, s in enumerate(shapes):
        if i not in solo:
            buckets.setdefault((int(s[0]), int(s[1]), int(i in with_k)), []).append(i)
    micro = []
    # micro-batch granularity: about three micro-batches per rank (a third of the per-rank cost share each) keeps the greedy
    # assignment within a few % of even, while micro-batches stay as large as that allows (larger batches run faster per image)
    share = sum(costs) / (3.0 * world) if world > 1 else float("inf")
    for s in sorted(buckets):
        idx = buckets[s]
        cap = max(1, min(max_batch, int(share / max(costs[idx[0]], 1e-30)))) if world > 1 else max_batch
        nmb = -(-len(idx) // cap)
        per = -(-len(idx) // nmb)                      # even micro-batches rather than full ones + a small remainder
        micro += [((s[0], s[1]), idx[k:k + per]) for k in range(0, len(idx), per)]
    micro += [(tuple(shapes[i]), [i]) for i in sorted(solo)]
    order = sorted(range(len(micro)), key=lambda j: (-sum(costs[i] for i in micro[j][1]), j))
    load = [0.0] * world
    owner = [0] * len(micro)
    for j in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[j] = r
        load[r] += sum(costs[i] for i in micro[j][1])
    return micro, owner


def infer_mixed(model, images: List[torch.Tensor], cameras: Optional[List[Optional[torch.Tensor]]] = None,
                keys: Iterable[str] = ("depth", "confidence", "intrinsics"), max_batch: int = 8, group=None, inflight: int = 2,
                **kw) -> List[Dict[str, torch.Tensor]]:
    """infer() over a list of [3,H,W] images of arbitrary, mixed shapes; `cameras[i]` is None, a [3,3] K tensor (batched with
    the other K tensors of its shape bucket) or a camera object (unidepth_amd.cameras / the reference's classes: that image then
    runs as its own call).  Single process (no initialised process group): the micro-batches run back to back on this GPU.  Under torch.distributed every rank passes the SAME list, runs the micro-batches
    the plan gives it and all ranks return the complete, ordered result list; one all-gather per shape bucket (the outputs of a
    bucket have one shape, so all requested keys travel packed in one message)."""
    keys = list(keys)
    distributed = dist.is_available() and dist.is_initialized()
    world, rank = (dist.get_world_size(group), dist.get_rank(group)) if distributed else (1, 0)
    shapes = [(int(im.shape[-2]), int(im.shape[-1])) for im in images]
    costs = [image_cost(model, h, w) for h, w in shapes]
    if cameras is not None and len(cameras) != len(images):
        raise ValueError(f"infer_mixed: {len(cameras)} cameras for {len(images)} images (pass None for images without intrinsics)")
    solo = [i for i, c in enumerate(cameras or []) if c is not None and not isinstance(c, torch.Tensor)]
    with_k = [i for i, c in enumerate(cameras or []) if isinstance(c, torch.Tensor)]
    micro, owner = plan_mixed(shapes, costs, world, max_batch, solo, with_k)
    results: List[Optional[Dict[str, torch.Tensor]]] = [None] * len(images)
    mine: Dict[Tuple[int, int], List[Tuple[List[int], Dict[str, torch.Tensor]]]] = {}
    # consecutive micro-batches of a rank overlap on separate HIP streams (pipeline.py) when the engine supports buffer slots
    pipe = None
    if inflight > 1 and hasattr(model, "_plans") and images and images[0].is_cuda:
        from .pipeline import InferPipeline
        pipe = InferPipeline(model, depth=inflight)
    prev_max_plans = getattr(model, "max_plans", None)
    if hasattr(model, "reserve_plans"):
        # every (micro-batch size, shape, camera mode) of this rank may land on every pipeline slot: keep the whole cycle cached for the
        # duration of THIS call, bounded by UNIDEPTH_MIXED_MAX_PLANS (a list with dozens of distinct shapes must not pin dozens of ~2.6 GB
        # plans); the previous bound comes back when the call returns (_restore_plans) and the LRU evicts down to it
        sigs = {(len(idx), s, bool(cameras is not None and any(cameras[i] is not None for i in idx))) for (s, idx), r in zip(micro, owner) if r == rank}
        cap = int(os.environ.get("UNIDEPTH_MIXED_MAX_PLANS", "24"))
        need = len(sigs) * max(1, inflight) + 2
        if need > max(cap, prev_max_plans or 0):
            import warnings
            warnings.warn(f"infer_mixed: this rank cycles through {need} (micro-batch, shape, slot) plans but the cache is capped at "
                          f"{max(cap, prev_max_plans or 0)} (UNIDEPTH_MIXED_MAX_PLANS): plans will be evicted and rebuilt (a device synchronise + "
                          "a multi-GB allocation each) -- raise the cap or pass fewer distinct shapes per call", RuntimeWarning, stacklevel=2)
        model.reserve_plans(min(need, max(cap, prev_max_plans or 0)))

    def _restore_plans():
        if prev_max_plans is not None and hasattr(model, "trim_plans"):
            model.trim_plans(prev_max_plans)
    try:
        submitted = []
        for (s, idx), r in zip(micro, owner):
            if r != rank:
                continue
            rgb = torch.stack([images[i] for i in idx])
            cam = None
            if cameras is not None and len(idx) == 1 and idx[0] in solo:
                cam = cameras[idx[0]]
            elif cameras is not None and any(isinstance(cameras[i], torch.Tensor) for i in idx):
                assert all(isinstance(cameras[i], torch.Tensor) for i in idx), "plan_mixed keeps K and camera-less images apart"
                cam = torch.stack([cameras[i].reshape(3, 3) for i in idx])
            if pipe is not None:
                out = pipe.submit(rgb, cam, **kw)
                submitted.append(out)
            else:
                out = model.infer(rgb, cam, **kw)
            mine.setdefault(s, []).append((idx, {k: out[k] for k in keys}))
            if not distributed:
                for b, i in enumerate(idx):
                    results[i] = {k: out[k][b] for k in keys}
        for out in submitted:                              # only now: a wait on the caller's stream would order later submissions behind it
            pipe.wait(out)
    finally:
        _restore_plans()               # also when infer() or the pipeline raised: the enlarged bound must not pin multi-GB plans
    if not distributed:
        return results  # type: ignore[return-value]
    for s in sorted({m[0] for m in micro}):
        # rank r owns these images of the bucket, in plan order
        per_rank = [[i for (ss, idx), r in zip(micro, owner) if ss == s and r == q for i in idx] for q in range(world)]
        counts = [len(p) for p in per_rank]
        H, W = s
        widths = {"depth": H * W, "confidence": H * W, "radius": H * W, "points": 3 * H * W, "rays": 3 * H * W, "intrinsics": 9}
        tail = {"depth": (1, H, W), "confidence": (1, H, W), "radius": (1, H, W), "points": (3, H, W), "rays": (3, H, W), "intrinsics": (3, 3)}
        for k in keys:
            if k not in widths:
                raise KeyError(f"infer_mixed: output '{k}' has no shape rule (depth_features depends on the network grid)")
        dev = next(iter(mine[s][0][1].values())).device if s in mine else (images[0].device if images[0].is_cuda else torch.device("cpu"))
        if counts[rank]:
            packed = torch.cat([torch.cat([o[k].reshape(o[k].shape[0], -1).float() for k in keys], dim=1) for _, o in mine[s]], dim=0)
        else:
            packed = torch.zeros((0, sum(widths[k] for k in keys)), dtype=torch.float32, device=dev)
        g = all_gather_batch(packed.contiguous(), counts, group)
        row = 0
        for q in range(world):
            for i in per_rank[q]:
                off = 0
                results[i] = {}
                for k in keys:
                    results[i][k] = g[row, off:off + widths[k]].reshape(tail[k])
                    off += widths[k]
                row += 1
    return results  # type: ignore[return-value]
