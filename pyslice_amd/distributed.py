"""Frame sharding and the two exchanges of the multi-GPU path (one process per GPU).

The reference has no distributed path.  Every (frame, probe) exit wave is
independent (calculators.py:172-186), so MD frames are sharded in contiguous blocks over the ranks
and nothing is exchanged while frames are propagated.  Two collectives exist, both at the end:

  * gather_frames      -- assemble (P, T, nx, ny) from the (P, T_r, nx, ny) shards (WFData);
  * frames_to_probes   -- all-to-all re-shard from frame-sharded to probe-sharded so every rank
                          holds complete time series for its probes and can run the TACAW time FFT
                          locally (tacaw_data.py:94-96 couples all frames of one probe/pixel).

Backend: "nccl" (RCCL over xGMI) on GPUs; the same code runs under "gloo" on CPU tensors, which is
how tests/test_distributed.py covers it without a GPU.  xGMI is point-to-point, so the all-to-all
(each pair its own link) is the natural pattern; no ring all-reduce is used anywhere.
"""
from __future__ import annotations

from typing import List, Optional

try:
    import torch
    import torch.distributed as dist
except ImportError:  # pragma: no cover
    torch = None
    dist = None


def rank_world():
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, world: int, rank: int):
    """Contiguous block [lo, hi) of `n` items for `rank` (first n % world ranks get one extra)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_frames(n_frames: int, world: int, rank: int) -> List[int]:
    lo, hi = shard_bounds(n_frames, world, rank)
    return list(range(lo, hi))


def _as_real(t):
    """Collectives move real views (gloo has no complex support; RCCL moves bytes either way).  Under gloo (CPU
    rehearsal of the multi-process path, possibly with device-resident data) tensors are staged through the host."""
    t = torch.view_as_real(t) if t.is_complex() else t
    if dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu(), True, t.device
    return t, True, None


def _restore(t, meta):
    if t is None:
        return None
    was_complex, dev = meta
    if dev is not None:
        t = t.to(dev)
    return torch.view_as_complex(t.contiguous()) if was_complex else t


def gather_frames(local, n_frames: int, dst: Optional[int] = 0):
    """(P, T_r, ...) shards -> (P, T, ...) on `dst` (None: on every rank).  Returns None elsewhere."""
    rank, world = rank_world()
    if world == 1:
        return local
    cplx = local.is_complex()
    local, _, dev = _as_real(local)
    return _restore(_gather_frames_real(local, n_frames, dst, rank, world), (cplx, dev))


def _gather_frames_real(local, n_frames, dst, rank, world):
    P = local.shape[0]
    rest = tuple(local.shape[2:])
    counts = [shard_bounds(n_frames, world, r)[1] - shard_bounds(n_frames, world, r)[0] for r in range(world)]
    tmax = max(counts)
    # equal-size buffers (frame axis first so that shards are contiguous), padded to the largest shard
    send = torch.zeros((tmax, P) + rest, dtype=local.dtype, device=local.device)
    send[: local.shape[1]] = local.transpose(0, 1)
    if dst is None:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(bufs, send)
    else:
        bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, bufs, dst=dst)
        if rank != dst:
            return None
    full = torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)       # (T, P, ...)
    return full.transpose(0, 1).contiguous()


def frames_to_probes(local, n_frames: int):
    """All-to-all: (P, T_r, ...) frame shard -> (P_r, T, ...) probe shard (complete time series)."""
    rank, world = rank_world()
    if world == 1:
        return local
    cplx = local.is_complex()
    local, _, dev = _as_real(local)
    return _restore(_frames_to_probes_real(local, n_frames, rank, world), (cplx, dev))


def _frames_to_probes_real(local, n_frames, rank, world):
    P = local.shape[0]
    rest = tuple(local.shape[2:])
    tcounts = [shard_bounds(n_frames, world, r)[1] - shard_bounds(n_frames, world, r)[0] for r in range(world)]
    pb = [shard_bounds(P, world, r) for r in range(world)]
    my_p = pb[rank][1] - pb[rank][0]
    send = [local[pb[r][0]:pb[r][1]].contiguous() for r in range(world)]          # to rank r: its probes, my frames
    recv = [torch.empty((my_p, tcounts[r]) + rest, dtype=local.dtype, device=local.device) for r in range(world)]
    dist.all_to_all(recv, send) if dist.get_backend() != "gloo" else _all_to_all_p2p(recv, send, rank, world)
    return torch.cat(recv, dim=1).contiguous()                                      # (P_r, T, ...)


def _all_to_all_p2p(recv, send, rank, world):
    """gloo has no all_to_all for uneven lists on every build: pairwise isend/irecv instead."""
    recv[rank].copy_(send[rank])
    reqs = []
    for r in range(world):
        if r == rank:
            continue
        if send[r].numel():
            reqs.append(dist.isend(send[r], dst=r))
        if recv[r].numel():
            reqs.append(dist.irecv(recv[r], src=r))
    for q in reqs:
        q.wait()


def gather_probes(local, n_probes: int, dst: Optional[int] = 0):
    """(P_r, ...) probe shards -> (P, ...) on dst (None: everywhere)."""
    rank, world = rank_world()
    if world == 1:
        return local
    cplx = local.is_complex()
    local, _, dev = _as_real(local)
    return _restore(_gather_probes_real(local, n_probes, dst, rank, world), (cplx, dev))


def _gather_probes_real(local, n_probes, dst, rank, world):
    counts = [shard_bounds(n_probes, world, r)[1] - shard_bounds(n_probes, world, r)[0] for r in range(world)]
    pmax = max(counts)
    send = torch.zeros((pmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    send[: local.shape[0]] = local
    if dst is None:
        bufs = [torch.empty_like(send) for _ in range(world)]
        dist.all_gather(bufs, send)
    else:
        bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        dist.gather(send, bufs, dst=dst)
        if rank != dst:
            return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
