"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo"
on CPU for tests).  PyTorch is only the launcher / collective provider here; the data path is the C ABI.

* extract + match shard by frame (no data-path collective): `frame_shard`, `max_over_ranks`
* local / global BA shards by landmark: `shard_by_keyframe_segment` (a landmark follows the keyframe segment that owns its keyframes; per
  damping trial only the separator blocks, what the segments leave on them and the solution cross ranks) or `shard_by_landmark` (l % world:
  one all-reduce of the whole reduced camera system per trial); `make_allreduce_callback` = the `svgpu_allreduce_fn` the library calls back into
"""
from __future__ import annotations

import ctypes as C

import numpy as np

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


def frame_shard(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of n_items for `rank` (first n % world ranks get one extra)."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a python float (the benchmark's time reduction)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_by_landmark(scene: dict, rank: int, world: int) -> dict:
    """Observation shard of a flat BA problem: all observations of landmark l go to rank l % world.
    Poses, points and intrinsics stay complete on every rank (what svgpu_local_ba_sharded expects)."""
    keep = (np.asarray(scene["obs_point"]) % world) == rank
    out = dict(scene)
    for k in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma_sq", "obs_huber"):
        out[k] = np.ascontiguousarray(np.asarray(scene[k])[keep])
    out["_obs_index"] = np.flatnonzero(keep)
    return out


def shard_by_keyframe_segment(scene: dict, rank: int, world: int) -> dict:
    """Observation shard along the keyframe segments of the `world`-rank solve (optimize.partition_keyframe_segments): every landmark goes
    to the rank that owns the piece of the keyframe graph its keyframes lie in, so per damping trial only the separator blocks cross ranks.
    Falls back to l % world when the problem gets no segmented plan.  The partition is cached in the scene dict (`_kfseg`)."""
    from . import optimize
    key = ("_kfseg", world)
    if scene.get("_kfseg_key") != key:
        scene["_kfseg"] = optimize.partition_keyframe_segments(scene, world)
        scene["_kfseg_key"] = key
    lm_rank, info = scene["_kfseg"]
    keep = lm_rank[np.asarray(scene["obs_point"])] == rank
    out = {k: v for k, v in scene.items() if not k.startswith("_kfseg")}
    for k in ("obs_pose", "obs_point", "obs_uvr", "obs_inv_sigma_sq", "obs_huber"):
        out[k] = np.ascontiguousarray(np.asarray(scene[k])[keep])
    out["_obs_index"] = np.flatnonzero(keep)
    out["_partition"] = info
    return out


class _CudaBuf:
    """Minimal __cuda_array_interface__ carrier so that torch can wrap a raw device pointer without copying."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def make_allreduce_callback(group=None, host_buffers: bool = False):
    """Returns (callback, keepalive): `callback` is a C function pointer of type svgpu_allreduce_fn that sums
    `count` doubles in place across the ranks of `group` with torch.distributed.all_reduce.
    The buffer is always device memory.  With the nccl backend the collective is enqueued on the library's stream; with a
    host-side backend (gloo) the payload is staged through the host."""
    import torch
    import torch.distributed as dist

    def _cb(user, buf, count, stream):
        try:
            if dist.get_backend(group) == "nccl":
                t = torch.as_tensor(_CudaBuf(buf, count), device="cuda")
                ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
                with torch.cuda.stream(ext):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            elif host_buffers:  # CPU-side plumbing tests only: the caller (not the library) passes host memory
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(count,))
                dist.all_reduce(torch.from_numpy(a), op=dist.ReduceOp.SUM, group=group)
            else:
                # the library always hands over DEVICE memory (svgpu.h): with a host-side backend (gloo: CPU-side tests of the
                # multi-rank control flow on one GPU) the payload makes the round trip through the host, ordered on the stream
                t = torch.as_tensor(_CudaBuf(buf, count), device="cuda")
                ext = torch.cuda.ExternalStream(stream) if stream else torch.cuda.current_stream()
                with torch.cuda.stream(ext):
                    h = t.cpu()          # synchronises with `ext` (everything enqueued before the callback has finished)
                    dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
                    t.copy_(h)
                    ext.synchronize()
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            import sys
            print("allreduce callback failed:", e, file=sys.stderr)
            return 1

    cb = ALLREDUCE_FN(_cb)
    return cb, (cb, _cb)


def init_comm(ctx, group=None) -> None:
    """Create the context's own RCCL communicator (svgpu_comm_init): rank 0 draws the ncclUniqueId, torch.distributed (any
    backend) only carries those 128 bytes to the other ranks.  Afterwards svgpu_local_ba_sharded / svgpu_global_ba_sharded
    run with allreduce = NULL: RCCL calls on the library's stream, no Python in the damping loop."""
    import torch
    import torch.distributed as dist
    from ._lib import lib
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ident = np.zeros(128, np.uint8)
    if rank == 0:
        ctx.check(lib().svgpu_comm_unique_id(C.c_void_p(ident.ctypes.data)), "svgpu_comm_unique_id")
    t = torch.from_numpy(ident)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0, group=group)
    ident = t.cpu().numpy().copy()
    ctx.check(lib().svgpu_comm_init(ctx.handle, rank, world, C.c_void_p(ident.ctypes.data)), "svgpu_comm_init")
