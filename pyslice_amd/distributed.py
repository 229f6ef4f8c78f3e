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

from typing import List, NamedTuple, Optional

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


class Op(NamedTuple):
    """One point-to-point operation of an exchange plan.  `ordinal` is the operation's position in the stream between its two
    ranks (the same number on both sides), `round` the grouped launch it belongs to, `where` names the tensor slice."""
    kind: str          # "send" | "recv"
    peer: int
    ordinal: int
    numel: int         # elements moved (0-element operations are dropped from a plan: on both sides, the sizes agree)
    round: int
    where: tuple


def _group_size(world: int) -> int:
    """ordinals per grouped launch: every rank exchanges with at most world - 1 peers in both directions"""
    return max(1, MAX_GROUP_OPS // (2 * max(1, world - 1)))


# ---- plans: pure functions of (rank, world, sizes) -- tests/test_distributed.py builds them for every rank of a world of 8 and
# checks that every send has its receive in the same round with the same ordinal and size (a mismatched grouped launch on RCCL is a
# hang, not an error) ----------------------------------------------------------------------------------------------------------

def plan_gather_frames(rank: int, world: int, n_probes: int, n_frames: int, inner: int, dst: Optional[int] = 0) -> List[Op]:
    """(P, T_r, inner) frame shards -> (P, T, inner) on `dst` (None: everywhere): per probe one contiguous run of T_r frames"""
    per = _group_size(world)
    bounds = [shard_bounds(n_frames, world, r) for r in range(world)]
    receivers = range(world) if dst is None else [dst]
    ops: List[Op] = []
    if rank in receivers:
        for r in range(world):
            if r != rank:
                lo, hi = bounds[r]
                ops += [Op("recv", r, p, (hi - lo) * inner, p // per, ("full", p, lo, hi)) for p in range(n_probes)]
    lo, hi = bounds[rank]
    for d in receivers:
        if d != rank:
            ops += [Op("send", d, p, (hi - lo) * inner, p // per, ("local", p)) for p in range(n_probes)]
    return [o for o in ops if o.numel]


def plan_frames_to_probes(rank: int, world: int, n_probes: int, n_frames: int, inner: int) -> List[Op]:
    """all-to-all (P, T_r, inner) -> (P_r, T, inner): my probes' frames of rank r in, rank r's probes' frames of mine out"""
    per = _group_size(world)
    tb = [shard_bounds(n_frames, world, r) for r in range(world)]
    pb = [shard_bounds(n_probes, world, r) for r in range(world)]
    p0, p1 = pb[rank]
    mylo, myhi = tb[rank]
    ops: List[Op] = []
    for r in range(world):
        if r == rank:
            continue
        lo, hi = tb[r]
        ops += [Op("recv", r, p - p0, (hi - lo) * inner, (p - p0) // per, ("mine", p - p0, lo, hi)) for p in range(p0, p1)]
        ops += [Op("send", r, p - pb[r][0], (myhi - mylo) * inner, (p - pb[r][0]) // per, ("local", p)) for p in range(pb[r][0], pb[r][1])]
    return [o for o in ops if o.numel]


def plan_gather_probes(rank: int, world: int, n_probes: int, inner: int, dst: Optional[int] = 0) -> List[Op]:
    """(P_r, inner) probe shards -> (P, inner) on `dst` (None: everywhere): one operation per pair"""
    pb = [shard_bounds(n_probes, world, r) for r in range(world)]
    receivers = range(world) if dst is None else [dst]
    ops: List[Op] = []
    if rank in receivers:
        ops += [Op("recv", r, 0, (pb[r][1] - pb[r][0]) * inner, 0, ("full", pb[r][0], pb[r][1])) for r in range(world) if r != rank]
    ops += [Op("send", d, 0, (pb[rank][1] - pb[rank][0]) * inner, 0, ("local",)) for d in receivers if d != rank]
    return [o for o in ops if o.numel]


def plan_reduce_probes(rank: int, world: int, n_probes: int, inner: int, itemsize: int, temp_bytes: float = 32e9):
    """sum-reduce-scatter over probes as a direct exchange: shift s sends the slice of rank (rank + s) % world and receives this
    rank's slice from rank (rank - s) % world; `per_round` shifts share one grouped launch and one set of receive buffers (sized
    for the LARGEST probe shard, so every rank cuts the rounds at the same shifts).  Returns (ops, per_round)."""
    pb = [shard_bounds(n_probes, world, r) for r in range(world)]
    p0, p1 = pb[rank]
    biggest = max(b - a for a, b in pb) * inner * itemsize
    per_round = int(max(1, min(world - 1, temp_bytes // max(1, biggest))))
    ops: List[Op] = []
    for s in range(1, world):
        rnd, slot = (s - 1) // per_round, (s - 1) % per_round
        to, frm = (rank + s) % world, (rank - s) % world
        a, b = pb[to]
        ops.append(Op("send", to, 0, (b - a) * inner, rnd, ("acc", a, b)))
        ops.append(Op("recv", frm, 0, (p1 - p0) * inner, rnd, ("temp", slot)))
    return [o for o in ops if o.numel], per_round


def _run(ops: List[Op], resolve):
    """the grouped launches of a plan, round by round (a round without operations is skipped locally); resolve(where) -> tensor"""
    for rnd in sorted({o.round for o in ops}):
        batch = [o for o in ops if o.round == rnd]
        p2p = [dist.P2POp(dist.irecv, resolve(o.where), o.peer) for o in batch if o.kind == "recv"]
        p2p += [dist.P2POp(dist.isend, resolve(o.where), o.peer) for o in batch if o.kind == "send"]
        for req in dist.batch_isend_irecv(p2p):
            req.wait()


def gather_frames(local, n_frames: int, dst: Optional[int] = 0):
    """(P, T_r, ...) shards -> (P, T, ...) on `dst` (None: on every rank).  Returns None elsewhere."""
    rank, world = rank_world()
    if world == 1:
        return local
    local, cplx, dev = _as_real(local)
    P = local.shape[0]
    inner = 1
    for d in local.shape[2:]:
        inner *= int(d)
    full = None
    if dst is None or rank == dst:
        full = _alloc((P, n_frames) + tuple(local.shape[2:]), local.dtype, local.device)
        lo, hi = shard_bounds(n_frames, world, rank)
        full[:, lo:hi].copy_(local)
    _run(plan_gather_frames(rank, world, P, n_frames, inner, dst),
         lambda w: full[w[1], w[2]:w[3]] if w[0] == "full" else _chunk(local[w[1]]))
    return _restore(full, cplx, dev)


def frames_to_probes(local, n_frames: int):
    """All-to-all: (P, T_r, ...) frame shard -> (P_r, T, ...) probe shard (complete time series)."""
    rank, world = rank_world()
    if world == 1:
        return local
    local, cplx, dev = _as_real(local)
    P = local.shape[0]
    inner = 1
    for d in local.shape[2:]:
        inner *= int(d)
    p0, p1 = shard_bounds(P, world, rank)
    mine = _alloc((p1 - p0, n_frames) + tuple(local.shape[2:]), local.dtype, local.device)
    lo, hi = shard_bounds(n_frames, world, rank)
    mine[:, lo:hi].copy_(local[p0:p1])
    _run(plan_frames_to_probes(rank, world, P, n_frames, inner),
         lambda w: mine[w[1], w[2]:w[3]] if w[0] == "mine" else _chunk(local[w[1]]))
    return _restore(mine, cplx, dev)


def gather_probes(local, n_probes: int, dst: Optional[int] = 0):
    """(P_r, ...) probe shards -> (P, ...) on dst (None: everywhere).  Returns None elsewhere."""
    rank, world = rank_world()
    if world == 1:
        return local
    local, cplx, dev = _as_real(local)
    inner = 1
    for d in local.shape[1:]:
        inner *= int(d)
    full = None
    if dst is None or rank == dst:
        full = _alloc((n_probes,) + tuple(local.shape[1:]), local.dtype, local.device)
        a, b = shard_bounds(n_probes, world, rank)
        full[a:b].copy_(local)
    _run(plan_gather_probes(rank, world, n_probes, inner, dst),
         lambda w: full[w[1]:w[2]] if w[0] == "full" else _chunk(local))
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

    Direct exchange (plan_reduce_probes): in shift s every rank sends the slice of rank (rank + s) % world and receives its own
    slice from rank (rank - s) % world -- as many shifts at once as `temp_bytes` of receive buffers allow (all world-1 of them
    when they fit: one grouped launch, every pair of GPUs on its own link), then adds what it received.  Per link: one probe shard
    of the accumulator; C5 (256 probes x 1024 bins x 128^2 stored pixels on 8 GPUs): 4.3 GB per pair, 30 GB of receive buffers.
    A device that cannot hold the receive buffers halves `temp_bytes` -- on EVERY rank, by agreement (an all-reduce of one flag),
    because the rounds must be cut at the same shifts everywhere."""
    rank, world = rank_world()
    p0, p1 = shard_bounds(n_probes, world, rank)
    if world == 1:
        return p0, p1
    real, cplx, dev = _as_real(acc)          # under gloo with device data: a host copy (rehearsal path)
    mine = real[p0:p1]
    inner = 1
    for d in real.shape[1:]:
        inner *= int(d)
    while True:
        ops, per_round = plan_reduce_probes(rank, world, n_probes, inner, real.element_size(), temp_bytes)
        temp, ok = None, 1
        try:
            temp = _alloc((per_round,) + tuple(mine.shape), mine.dtype, mine.device)
        except RuntimeError:                 # out of memory (torch.cuda.OutOfMemoryError is a RuntimeError)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=real.device if dist.get_backend() != "gloo" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            break
        del temp
        if per_round == 1:
            raise MemoryError("reduce_probes: no room for one probe shard of receive buffer")
        temp_bytes = max(1.0, temp_bytes / 2)
    for rnd in sorted({o.round for o in ops}):
        batch = [o for o in ops if o.round == rnd]
        _run(batch, lambda w: temp[w[1]] if w[0] == "temp" else _chunk(real[w[1]:w[2]]))
        if p1 > p0:
            for o in sorted((o for o in batch if o.kind == "recv"), key=lambda o: o.where[1]):
                mine.add_(temp[o.where[1]])  # fixed order (the shifts of the round): deterministic sums, no further allocation
    del temp
    if dev is not None and p1 > p0:          # gloo rehearsal with device data: write the reduced slice back
        target = torch.view_as_real(acc) if cplx else acc
        target[p0:p1].copy_(mine)
    return p0, p1
