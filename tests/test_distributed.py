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


def _worker(rank, world, port, P, T, npix, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyslice_amd import distributed as D
        rng = np.random.default_rng(0)
        full = torch.from_numpy((rng.standard_normal((P, T, npix)) + 1j * rng.standard_normal((P, T, npix))).astype(np.complex64))
        frames = D.shard_frames(T, world, rank)
        local = full[:, frames[0]:frames[-1] + 1].contiguous() if frames else full[:, :0]
        ok = True
        g = D.gather_frames(local, T, dst=0)
        ok &= (g is None) if rank != 0 else bool(torch.equal(g, full))
        ga = D.gather_frames(local, T, dst=None)
        ok &= bool(torch.equal(ga, full))
        mine = D.frames_to_probes(local, T)
        lo, hi = D.shard_bounds(P, world, rank)
        ok &= bool(torch.equal(mine, full[lo:hi]))
        # TACAW on the probe shard == TACAW on the full array restricted to those probes
        if mine.shape[0]:
            inten = (torch.fft.fftshift(torch.fft.fft(mine - mine.mean(dim=1, keepdim=True), dim=1), dim=1).abs() ** 2)
        else:       # a rank can own zero probes (P < world); MKL rejects empty FFTs
            inten = torch.zeros(mine.shape, dtype=torch.float32)
        gi = D.gather_probes(inten, P, dst=0)
        if rank == 0:
            want = (torch.fft.fftshift(torch.fft.fft(full - full.mean(dim=1, keepdim=True), dim=1), dim=1).abs() ** 2)
            ok &= bool(torch.allclose(gi, want, rtol=1e-5, atol=1e-5))
        else:
            ok &= gi is None
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("P,T", [(4, 6), (3, 5), (1, 2)])
def test_frame_shard_exchanges_gloo(P, T):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, T, 12, q)) for r in range(world)]
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
