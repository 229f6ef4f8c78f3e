"""world_size-2 gloo tests of the multi-GPU exchanges (frame shards -> WFData gather, frame->probe
all-to-all for TACAW, probe gather) on CPU tensors."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, P, T, npix, q, max_ops=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyslice_amd import distributed as D
        if max_ops:
            D.MAX_GROUP_OPS = max_ops           # force several grouped launches per exchange
        rng = np.random.default_rng(0)
        full = torch.from_numpy((rng.standard_normal((P, T, npix)) + 1j * rng.standard_normal((P, T, npix))).astype(np.complex64))
        frames = D.shard_frames(T, world, rank)
        local = full[:, frames[0]:frames[-1] + 1].contiguous() if frames else full[:, :0]
        ok = True
        full_bytes = full.numel() * full.element_size()
        D.alloc_log.clear()
        g = D.gather_frames(local, T, dst=0)
        ok &= (g is None) if rank != 0 else bool(torch.equal(g, full))
        # the exchange allocates the result in its final layout on the receiver and nothing else anywhere
        ok &= D.alloc_log == ([full_bytes] if rank == 0 else [])
        D.alloc_log.clear()
        ga = D.gather_frames(local, T, dst=None)
        ok &= bool(torch.equal(ga, full)) and D.alloc_log == [full_bytes]
        # a shard that is a strided view (the engine's buffer holds more frame slots than the shard uses)
        padded = torch.zeros((P, local.shape[1] + 2, npix), dtype=local.dtype)
        padded[:, :local.shape[1]] = local
        ok &= bool(torch.equal(D.gather_frames(padded[:, :local.shape[1]], T, dst=None), full))
        D.alloc_log.clear()
        mine = D.frames_to_probes(local, T)
        lo, hi = D.shard_bounds(P, world, rank)
        ok &= bool(torch.equal(mine, full[lo:hi]))
        ok &= D.alloc_log == [(hi - lo) * T * npix * 8]
        # TACAW on the probe shard == TACAW on the full array restricted to those probes
        if mine.shape[0]:
            inten = (torch.fft.fftshift(torch.fft.fft(mine - mine.mean(dim=1, keepdim=True), dim=1), dim=1).abs() ** 2)
        else:       # a rank can own zero probes (P < world); MKL rejects empty FFTs
            inten = torch.zeros(mine.shape, dtype=torch.float32)
        D.alloc_log.clear()
        gi = D.gather_probes(inten, P, dst=0)
        ok &= D.alloc_log == ([P * T * npix * 4] if rank == 0 else [])
        if rank == 0:
            want = (torch.fft.fftshift(torch.fft.fft(full - full.mean(dim=1, keepdim=True), dim=1), dim=1).abs() ** 2)
            ok &= bool(torch.allclose(gi, want, rtol=1e-5, atol=1e-5))
        else:
            ok &= gi is None
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P,T,world,max_ops", [(4, 6, 2, None), (3, 5, 2, None), (1, 2, 2, None), (7, 5, 3, 4)])
def test_frame_shard_exchanges_gloo(P, T, world, max_ops):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, T, 12, q, max_ops)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def _reduce_worker(rank, world, port, P, F, K, temp_bytes, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyslice_amd import distributed as D
        # partial sums of every rank (the streaming TACAW accumulators of a frame shard): complex64 (P,F,K), float64 (P,K,2), (P,K)
        parts = []
        for r in range(world):
            rng = np.random.default_rng(100 + r)
            parts.append((torch.from_numpy((rng.standard_normal((P, F, K)) + 1j * rng.standard_normal((P, F, K))).astype(np.complex64)),
                          torch.from_numpy(rng.standard_normal((P, K, 2))), torch.from_numpy(rng.standard_normal((P, K)))))
        mine = [t.clone() for t in parts[rank]]
        lo, hi = D.shard_bounds(P, world, rank)
        ok = True
        for i, t in enumerate(mine):
            D.alloc_log.clear()
            p0, p1 = D.reduce_probes(t, P, temp_bytes=temp_bytes)
            ok &= (p0, p1) == (lo, hi)
            # summed in rank order after this rank's own part: compare against the same order (exact in float64, 1 ulp-ish in float32)
            want = parts[rank][i][lo:hi].clone()
            for s in range(1, world):
                want = want + parts[(rank - s) % world][i][lo:hi]
            ok &= bool(torch.equal(t[lo:hi], want))
            # receive buffers only, never more than the budget allows (one shard at least)
            shard = (hi - lo) * (t[0].numel() if P else 0) * t.element_size()
            biggest = max(b - a for a, b in (D.shard_bounds(P, world, r) for r in range(world))) * t[0].numel() * t.element_size()
            per_round = int(max(1, min(world - 1, temp_bytes // max(1, biggest))))
            ok &= all(a <= per_round * max(shard, 1) for a in D.alloc_log)
            # the rest of the array is left alone
            if lo > 0:
                ok &= bool(torch.equal(t[:lo], parts[rank][i][:lo]))
        # the reference pattern travels from rank 0 to everybody
        ref = parts[0][0][:, 0].clone() if rank == 0 else torch.zeros((P, K), dtype=torch.complex64)
        D.broadcast_from(ref, src=0)
        ok &= bool(torch.equal(ref, parts[0][0][:, 0]))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P,world,temp_bytes", [(4, 2, 32e9), (3, 2, 32e9), (1, 2, 32e9), (7, 3, 32e9), (7, 3, 1.0)])
def test_reduce_scatter_over_probes_gloo(P, world, temp_bytes):
    """frame-sharded streaming TACAW: the ranks' partial sums are summed by a direct exchange, probe-sharded result
    (uneven shards, a rank without probes, one round of all peers or -- tiny receive budget -- one peer per round)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reduce_worker, args=(r, world, port, P, 5, 6, temp_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok in res), res


def test_shard_bounds_cover():
    from pyslice_amd.distributed import shard_bounds, shard_frames
    for n in (1, 2, 7, 256):
        for w in (1, 2, 3, 8):
            cover = []
            for r in range(w):
                cover += shard_frames(n, w, r)
            assert cover == list(range(n))
            sizes = [shard_bounds(n, w, r)[1] - shard_bounds(n, w, r)[0] for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def test_bench_starts_its_own_ranks_and_relays_a_failure():
    """`python bench.py --gpus 2` without a launcher starts two rank processes itself (fresh interpreters, before
    anything touches a GPU).  Here there is no GPU, so both ranks fail: the parent must return their failure
    instead of hanging in the rendezvous, and must not print a result line."""
    import subprocess
    import sys
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the run itself is covered by the -m gpu suite")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--grid", "64", "--slices", "2",
                        "--probes", "1", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout


def _check_pairing(plans, world, max_group):
    """every send of rank s to rank r has exactly one receive on r from s: same round, ordinal and element count, in the same
    order (a pair's operations are matched by order); no rank enters a round with a peer that the peer skips; no grouped launch
    holds more than the bound"""
    for s in range(world):
        for r in range(world):
            if s == r:
                continue
            sends = [(o.round, o.ordinal, o.numel) for o in plans[s] if o.kind == "send" and o.peer == r]
            recvs = [(o.round, o.ordinal, o.numel) for o in plans[r] if o.kind == "recv" and o.peer == s]
            assert sends == recvs, (s, r, sends[:3], recvs[:3])
            assert sends == sorted(sends), (s, r)                       # rounds ascend along a pair's stream
            assert all(n > 0 for _, _, n in sends)
    for rank in range(world):
        rounds = {}
        for o in plans[rank]:
            rounds[o.round] = rounds.get(o.round, 0) + 1
        assert all(n <= max_group for n in rounds.values()), (rank, max(rounds.values()))


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_exchange_plans_pair_up_world8(world):
    """The exchanges are grouped point-to-point launches; on RCCL a send without its receive in the matching group is a hang, not
    an error, and no multi-GPU hardware has run them.  The plans are pure functions of (rank, world, sizes): build them for every
    rank -- BASELINE C4's and C5's shapes and uneven shards (fewer probes than ranks, frame counts that do not divide, receive
    budgets that force several rounds) -- and check that they pair up, stay inside the group bound, and move the bytes DESIGN.md
    section 5 states."""
    from pyslice_amd import distributed as D
    npix = 1024 * 1024
    cases = [(64, 256, 2 * npix), (3, 11, 10), (1, 5, 7), (17, 9, 3), (256, 2000, 4)]
    for P, T, inner in cases:
        for dst in (0, None, world - 1):
            plans = [D.plan_gather_frames(r, world, P, T, inner, dst) for r in range(world)]
            _check_pairing(plans, world, D.MAX_GROUP_OPS)
            # what the destination receives is everything but its own shard
            for d in (range(world) if dst is None else [dst]):
                lo, hi = D.shard_bounds(T, world, d)
                assert sum(o.numel for o in plans[d] if o.kind == "recv") == P * (T - (hi - lo)) * inner
        plans = [D.plan_frames_to_probes(r, world, P, T, inner) for r in range(world)]
        _check_pairing(plans, world, D.MAX_GROUP_OPS)
        for r in range(world):
            p0, p1 = D.shard_bounds(P, world, r)
            lo, hi = D.shard_bounds(T, world, r)
            assert sum(o.numel for o in plans[r] if o.kind == "recv") == (p1 - p0) * (T - (hi - lo)) * inner
            assert sum(o.numel for o in plans[r] if o.kind == "send") == (P - (p1 - p0)) * (hi - lo) * inner
        for dst in (0, None):
            _check_pairing([D.plan_gather_probes(r, world, P, T * inner, dst) for r in range(world)], world, D.MAX_GROUP_OPS)
        for temp_bytes in (32e9, 1.0, 3.5 * ((P + world - 1) // world) * inner * 8):
            both = [D.plan_reduce_probes(r, world, P, inner, 8, temp_bytes) for r in range(world)]
            assert len({pr for _, pr in both}) == 1                      # every rank cuts the rounds at the same shifts
            _check_pairing([ops for ops, _ in both], world, D.MAX_GROUP_OPS)
            per_round = both[0][1]
            for r, (ops, _) in enumerate(both):
                p0, p1 = D.shard_bounds(P, world, r)
                assert sum(o.numel for o in ops if o.kind == "recv") == (world - 1) * (p1 - p0) * inner
                assert all(o.where[1] < per_round for o in ops if o.kind == "recv")          # receive slots inside the buffer
    # a small group bound forces several grouped launches per exchange (the probe ordinals cut the rounds)
    old = D.MAX_GROUP_OPS
    try:
        D.MAX_GROUP_OPS = 4 * (world - 1)
        plans = [D.plan_frames_to_probes(r, world, 19, 13, 3) for r in range(world)]
        _check_pairing(plans, world, D.MAX_GROUP_OPS)
        assert max(o.round for ops in plans for o in ops) >= 1
    finally:
        D.MAX_GROUP_OPS = old
    if world == 8:
        # DESIGN.md section 5, BASELINE C4 (64 probes x 256 frames x 1024^2 complex64 on 8 GPUs)
        gf = D.plan_gather_frames(0, 8, 64, 256, 2 * npix, 0)
        assert len(gf) == 64 * 7 and sum(o.numel for o in gf) * 4 == 64 * 224 * npix * 8          # 120.3 GB into rank 0, one group
        assert {o.round for o in gf} == {0}
        assert all(o.numel * 4 == 32 * npix * 8 for o in gf)                                       # 268 MB per transfer
        fp = D.plan_frames_to_probes(3, 8, 64, 256, 2 * npix)
        assert sum(o.numel for o in fp if o.kind == "send") * 4 == 56 * 32 * npix * 8             # 15.0 GB leave every rank
        # C5: 256 probes x 1024 bins x 128^2 stored pixels, complex64 accumulators
        ops, per_round = D.plan_reduce_probes(5, 8, 256, 1024 * 128 * 128 * 2, 4, 32e9)
        assert per_round == 7 and {o.round for o in ops} == {0}
        assert all(o.numel * 4 == 32 * 1024 * 128 * 128 * 8 for o in ops)                          # 4.3 GB per link
