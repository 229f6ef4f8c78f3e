"""Frame sharding and the two exchanges of the multi-GPU path (one process per GPU).

The reference has no distributed path.  Every (frame, probe) exit wave is
independent (calculators.py:172-186), so MD frames are sharded in contiguous blocks over the ranks
and nothing is exchanged while frames are propagated.  Two exchanges exist, both at the end:

  * gather_frames      -- assemble (P, T, nx, ny) from the (P, T_r, nx, ny) shards (WFData);
  * reduce_probes      -- streaming TACAW (the (P,T,nx,ny) array is never held anywhere: BASELINE config C5): every rank has
                          folded ITS frames into partial sums A_r[p,f,k]; the transform is linear in the frames, so the
                          result is sum_r A_r -- a reduce-scatter over probes done as a direct exchange (every pair of
                          GPUs its own xGMI link, each rank sums the world-1 slices it receives for its probes), followed
                          by gather_probes of the finished intensities.  No ring, no all-reduce of the full accumulator;
  * frames_to_probes   -- all-to-all re-shard from frame-sharded to probe-sharded so every rank
                          holds complete time series for its probes and can run the TACAW time FFT
                          locally (tacaw_data.py:94-96 couples all frames of one probe/pixel),
                          followed by gather_probes of the intensities.

Every exchange receives STRAIGHT INTO THE DESTINATION: the result array is allocated once in its final
(P, T, ...) layout and each peer's contribution -- for one probe a contiguous run of T_r frames -- is
received into its slice by a point-to-point operation; senders send slices of their shard in place.
The operations of one exchange are issued as grouped launches (torch.distributed.batch_isend_irecv =
ncclGroupStart/End on RCCL) of at most MAX_GROUP_OPS operations -- one launch at C4 -- so every pair of GPUs uses its own
xGMI link concurrently.  No padded
copies, no list of per-rank buffers, no concatenation: the only allocation is the result itself
(`alloc_log` records it; tests/test_distributed.py asserts it).  Footprint on rank 0 of BASELINE
config C4 (64 probes x 256 frames x 1024^2 on 8 GPUs, gather="rank0", output="device"): 137.4 GB result
+ 17.2 GB own shard (the engine's buffer) = 154.6 GB of 288 GB; TACAW: 17.2 GB probe shard + 8.6 GB
intensities per rank, 68.7 GB gathered on rank 0.

Backend: "nccl" (RCCL over xGMI) on GPUs; the same code runs under "gloo" on CPU tensors, which is
how tests/test_distributed.py covers it without a GPU.  No ring all-reduce is used anywhere.
"""
from __future__ import annotations

from typing import List, Optional

try:
    import torch
    import torch.distributed as dist
except ImportError:  # pragma: no cover
    torch = None
    dist = None

# bytes allocated by the exchanges since the last reset (test hook: the gathers must allocate the result and nothing else)
alloc_log: List[int] = []


def _alloc(shape, dtype, device):
    t = torch.empty(tuple(int(s) for s in shape), dtype=dtype, device=device)
    alloc_log.append(t.numel() * t.element_size())
    return t


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
    """Exchanges move real views (gloo has no complex support; RCCL moves bytes either way).  Under gloo with
    device-resident data (CPU rehearsal of the multi-process path on a GPU box) the shard is staged through the host."""
    cplx = t.is_complex()
    t = torch.view_as_real(t) if cplx else t
    if dist.get_backend() == "gloo" and t.is_cuda:
        return t.cpu(), cplx, t.device
    return t, cplx, None


def _restore(t, cplx, dev):
    if t is None:
        return None
    if dev is not None:
        t = t.to(dev)
    return torch.view_as_complex(t) if cplx else t


def _chunk(t):
    """a view usable as a point-to-point operand (inner dimensions dense); copies only if the caller's shard is strided"""
    return t if t.is_contiguous() else t.contiguous()


MAX_GROUP_OPS = 1024      # point-to-point operations per grouped launch (RCCL bounds the operations of one group)


def _exchange(sends, recvs, world):
    """sends / recvs: [(tensor, peer, ordinal)].  Grouped launches (every pair of GPUs on its own link at once) of at most
    MAX_GROUP_OPS operations: round k carries the ordinals [k*B, (k+1)*B) -- the ordinal is the probe's position in the
    sender-to-receiver stream, the same number on both sides, so every rank cuts the rounds at the same places."""
    per = max(1, MAX_GROUP_OPS // (2 * max(1, world - 1)))
    top = max([j for _, _, j in sends] + [j for _, _, j in recvs] + [-1])
    # every rank must run the same number of rounds only if it takes part in them: a round with no operation is skipped locally
    for lo in range(0, top + 1, per):
        ops = [dist.P2POp(dist.irecv, t, peer) for t, peer, j in recvs if lo <= j < lo + per and t.numel()]
        ops += [dist.P2POp(dist.isend, t, peer) for t, peer, j in sends if lo <= j < lo + per and t.numel()]
        if not ops:
            continue
        for req in dist.batch_isend_irecv(ops):
            req.wait()


def gather_frames(local, n_frames: int, dst: Optional[int] = 0):
    """(P, T_r, ...) shards -> (P, T, ...) on `dst` (None: on every rank).  Returns None elsewhere."""
    rank, world = rank_world()
    if world == 1:
        return local
    local, cplx, dev = _as_real(local)
    P = local.shape[0]
    bounds = [shard_bounds(n_frames, world, r) for r in range(world)]
    receivers = range(world) if dst is None else [dst]
    full = None
    sends, recvs = [], []
    if rank in receivers:
        full = _alloc((P, n_frames) + tuple(local.shape[2:]), local.dtype, local.device)
        lo, hi = bounds[rank]
        full[:, lo:hi].copy_(local)
        for r in range(world):
            if r != rank:
                lo, hi = bounds[r]
                recvs += [(full[p, lo:hi], r, p) for p in range(P)]
    for d in receivers:
        if d != rank:
            sends += [(_chunk(local[p]), d, p) for p in range(P)]
    _exchange(sends, recvs, world)
    return _restore(full, cplx, dev)


def frames_to_probes(local, n_frames: int):
    """All-to-all: (P, T_r, ...) frame shard -> (P_r, T, ...) probe shard (complete time series)."""
    rank, world = rank_world()
    if world == 1:
        return local
    local, cplx, dev = _as_real(local)
    P = local.shape[0]
    tb = [shard_bounds(n_frames, world, r) for r in range(world)]
    pb = [shard_bounds(P, world, r) for r in range(world)]
    p0, p1 = pb[rank]
    mine = _alloc((p1 - p0, n_frames) + tuple(local.shape[2:]), local.dtype, local.device)
    lo, hi = tb[rank]
    mine[:, lo:hi].copy_(local[p0:p1])
    sends, recvs = [], []
    for r in range(world):
        if r == rank:
            continue
        lo, hi = tb[r]
        recvs += [(mine[p - p0, lo:hi], r, p - p0) for p in range(p0, p1)]                      # my probes, rank r's frames
        sends += [(_chunk(local[p]), r, p - pb[r][0]) for p in range(pb[r][0], pb[r][1])]       # rank r's probes, my frames
    _exchange(sends, recvs, world)
    return _restore(mine, cplx, dev)


def gather_probes(local, n_probes: int, dst: Optional[int] = 0):
    """(P_r, ...) probe shards -> (P, ...) on dst (None: everywhere).  Returns None elsewhere."""
    rank, world = rank_world()
    if world == 1:
        return local
    local, cplx, dev = _as_real(local)
    pb = [shard_bounds(n_probes, world, r) for r in range(world)]
    receivers = range(world) if dst is None else [dst]
    full = None
    sends, recvs = [], []
    if rank in receivers:
        full = _alloc((n_probes,) + tuple(local.shape[1:]), local.dtype, local.device)
        full[pb[rank][0]:pb[rank][1]].copy_(local)
        recvs = [(full[pb[r][0]:pb[r][1]], r, 0) for r in range(world) if r != rank]
    sends = [(_chunk(local), d, 0) for d in receivers if d != rank]
    _exchange(sends, recvs, world)
    return _restore(full, cplx, dev)


def broadcast_from(t, src: int = 0):
    """in-place broadcast of a tensor from rank `src` (the streaming TACAW reference pattern: small, once per run)"""
    rank, world = rank_world()
    if world == 1:
        return t
    r, cplx, dev = _as_real(t)
    dist.broadcast(r, src=src)
    if dev is not None and rank != src:
        torch.view_as_real(t).copy_(r) if cplx else t.copy_(r)
    return t


def reduce_probes(acc, n_probes: int, temp_bytes: float = 32e9):
    """Sum-reduce-scatter over probes, in place.  acc: (P, ...) partial sums of this rank (real or complex, device or host).
    On return acc[p0:p1] holds the sum over ALL ranks for this rank's probe range [p0, p1) = shard_bounds(P, world, rank);
    the rest of acc is unchanged (stale partial sums).  Returns (p0, p1).

    Direct exchange: in round s every rank sends the slice of rank (rank + s) % world and receives its own slice from rank
    (rank - s) % world -- as many rounds at once as `temp_bytes` of receive buffers allow (all world-1 of them when they fit:
    one grouped launch, every pair of GPUs on its own link), then adds what it received.  Per link: one probe shard of the
    accumulator; C5 (256 probes x 1024 bins x 128^2 stored pixels on 8 GPUs): 4.3 GB per pair, 30 GB of receive buffers."""
    rank, world = rank_world()
    pb = [shard_bounds(n_probes, world, r) for r in range(world)]
    p0, p1 = pb[rank]
    if world == 1:
        return p0, p1
    real, cplx, dev = _as_real(acc)          # under gloo with device data: a host copy (rehearsal path)
    mine = real[p0:p1]
    shard_bytes = max(1, mine.numel() * mine.element_size())
    # the same number of shifts per round on every rank: sized for the LARGEST probe shard
    biggest = max(b - a for a, b in pb) * (real[0].numel() if real.shape[0] else 0) * real.element_size()
    per_round = int(max(1, min(world - 1, temp_bytes // max(1, biggest))))
    for s0 in range(1, world, per_round):
        shifts = range(s0, min(world, s0 + per_round))
        temp = _alloc((len(shifts),) + tuple(mine.shape), mine.dtype, mine.device)
        sends, recvs = [], []
        for i, s in enumerate(shifts):
            to, frm = (rank + s) % world, (rank - s) % world
            a, b = pb[to]
            if b > a:
                sends.append((_chunk(real[a:b]), to, 0))
            if p1 > p0:
                recvs.append((temp[i], frm, 0))
        _exchange(sends, recvs, world)
        if p1 > p0:
            for i in range(len(shifts)):         # fixed order: deterministic sums; in place: no allocation beyond the receive buffers
                mine.add_(temp[i])
        del temp
    if dev is not None and p1 > p0:          # gloo rehearsal with device data: write the reduced slice back
        target = torch.view_as_real(acc) if cplx else acc
        target[p0:p1].copy_(mine)
    return p0, p1
