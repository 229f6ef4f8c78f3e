"""A run of BASELINE C5's shape on ONE MI355X (its share of the 8-GPU job): 2048^2 grid, 400 slices, 16 probes, the
k-window and detector bin DESIGN.md picks for C5 (window 512 x 512, bin 4 x 4 -> 128 x 128 stored pixels), streaming TACAW
through a ring of 8 frame slots.  Prints the time per frame, the slice-step rate and the device memory footprint.

    python tools/c5_shaped_run.py [--frames 16] [--probes 16] [--slices 400]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--probes", type=int, default=16)
    ap.add_argument("--slices", type=int, default=400)
    ap.add_argument("--grid", type=int, default=2048)
    ap.add_argument("--tile", type=int, default=8)
    a = ap.parse_args()
    import torch
    import pyslice_amd as ps
    from pyslice_amd.synthetic import synthetic_trajectory, stem_probe_grid
    free0, total = torch.cuda.mem_get_info(0)
    t0 = time.time()
    tr = synthetic_trajectory(a.grid, a.slices, a.frames, seed=0)
    side = int(round(a.probes ** 0.5))
    pp = [tuple(p) for p in stem_probe_grid(side)]
    print(f"trajectory: {tr.n_atoms} atoms x {a.frames} frames ({time.time() - t0:.1f} s to synthesise)", flush=True)
    calc = ps.MultisliceCalculator(device=0, progress=False, stream_tile=a.tile, k_window=(512, 512), k_bin=(4, 4), output="device")
    t0 = time.time()
    calc.setup(tr, aperture=30.0, voltage_eV=100e3, probe_positions=pp)
    t_setup = time.time() - t0
    t0 = time.time()
    tac = calc.run_streaming_tacaw()
    calc._engine.synchronize()
    dt = time.time() - t0
    free1, _ = torch.cuda.mem_get_info(0)
    steps = len(pp) * a.frames * a.slices
    npix = a.grid * a.grid
    print(f"grid {a.grid}^2, {a.slices} slices, {len(pp)} probes, {a.frames} frames, ring of {calc._engine.n_frames} slots, "
          f"frame batch {calc._engine.frame_batch}")
    print(f"setup {t_setup:.2f} s; run {dt:.2f} s = {dt / a.frames * 1e3:.1f} ms per frame; {steps / dt:,.0f} slice-steps/s "
          f"(potential + slice loop + exit FFT + binning + fold)")
    print(f"device memory in use: {(free0 - free1) / 1e9:.1f} GB of {total / 1e9:.0f} GB")
    gb = lambda n: f"{n / 1e9:.2f} GB"
    P, nz, na = len(pp), a.slices, tr.n_atoms
    print("  transmission stacks (2 orientations): " + gb(2 * nz * npix * 8))
    print("  phase tables of the potential build  : " + gb(2 * na * (a.grid // 2 + 1) * 8) + "  (columns 0 .. n/2: the quadrant kernel)")
    print("  work buffers (psi0, psi, psiT)        : " + gb(3 * P * a.grid * (a.grid + 16) * 8))
    print("  staging window + ring + accumulators : " + gb(P * 512 * 512 * 8 + P * calc._engine.n_frames * 128 * 128 * 8 + P * a.frames * 128 * 128 * 12))
    print(f"  the reference's layout for the same run (P,T,nx,ny) complex128: {gb(P * a.frames * npix * 16)}; "
          f"full C5 share of one GPU (256 probes x 128 frames): {gb(256 * 128 * npix * 16)}")
    inten = tac.intensity
    print(f"result: intensity {tuple(inten.shape)} {inten.dtype}, total_diffraction {tac.total_diffraction.shape}; "
          f"sum over bins == total: {float(inten.sum(dim=1).double().sum()):.6e} vs {tac.total_diffraction.sum():.6e}")


if __name__ == "__main__":
    main()
