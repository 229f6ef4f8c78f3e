"""Randomised parity sweep of msl_tacaw: random frame counts 2 .. 1024 (half of them drawn from the 2-3-5-smooth numbers, so that
every mixed-radix kernel family comes up), random pixel counts 1 .. 40 000 (odd, tiny, ragged tiles), 1-3 probes, strong-mean
pixels; per pixel against the float64 transform of the same float32 samples.  usage: python tools/fuzz_tacaw.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def smooth(n):
    for p in (2, 3, 5):
        while n % p == 0:
            n //= p
    return n == 1


def main():
    import torch
    from pyslice_amd import _native
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    smooth_T = [t for t in range(2, 1025) if smooth(t)]
    dev = torch.device("cuda", 0)
    eng = _native.Engine(2, 2, 1, 1.0, 1.0, 1.0, 1.0, 0.0, n_probes=1, n_frames=0, device=0)
    worst, bad, t0 = 0.0, 0, time.time()
    for case in range(n_cases):
        T = int(rng.choice(smooth_T)) if rng.random() < 0.5 else int(rng.integers(2, 1025))
        npix = int(rng.choice([1, 2, 31, 33, 63, 64, 65, 255, 257, int(rng.integers(1, 40001)), int(rng.integers(1, 40001))]))
        P = int(rng.integers(1, 4))
        big = rng.standard_normal((P, 1, npix)) + 1j * rng.standard_normal((P, 1, npix))
        big[:, :, ::3] = 0.0
        x = (big * 1e2 + (rng.standard_normal((P, T, npix)) + 1j * rng.standard_normal((P, T, npix))) * 1e-2).astype(np.complex64)
        src = torch.from_numpy(x).to(dev)
        dst = torch.full((P, T, npix), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        eng.tacaw(src.data_ptr(), dst.data_ptr(), P, T, npix)
        eng.synchronize()
        got = dst.cpu().numpy().astype(np.float64)
        x64 = x.astype(np.complex128)
        want = np.abs(np.fft.fftshift(np.fft.fft(x64 - x64.mean(axis=1, keepdims=True), axis=1), axes=1)) ** 2
        err = float((np.linalg.norm(got - want, axis=1) / np.maximum(np.linalg.norm(want, axis=1), 1e-300)).max())
        ok = err < 5e-5 and got.min() >= 0.0 and got[:, T // 2].max() == 0.0
        worst = max(worst, err)
        bad += not ok
        print(f"{case:3d} T={T:4d} {'smooth' if smooth(T) else '      '} npix={npix:5d} P={P}: max per-pixel rel-L2 {err:.2e}{'' if ok else '   <-- FAIL'}", flush=True)
    eng.close()
    print(f"{n_cases} cases, worst per-pixel rel-L2 {worst:.2e}, {bad} failures, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
