"""CPU-side checks: the C-ABI library loads and exports every symbol include/mslice.h declares,
host logic (grid, slice edges, sharding) agrees with the oracle, and the product fails loudly
without a GPU (no CPU fallback)."""
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pyslice_amd import build_native, _native
    build_native.build()
    return _native.load()


def test_header_symbols_exported(lib):
    from pyslice_amd import _native
    hdr = open(os.path.join(REPO, "include", "mslice.h")).read()
    declared = set(re.findall(r"\b(msl_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in mslice.h but not exported"
    assert declared == set(_native.EXPORTS)
    assert lib.msl_abi_version() == 3


def test_struct_layouts_match_header(lib):
    """ctypes mirrors of msl_config / msl_counters have the C sizes (guards silent ABI drift)."""
    import ctypes as C
    from pyslice_amd import _native
    assert C.sizeof(_native.MslConfig) == 3 * 4 + 4 + 5 * 8 + 5 * 4 + 7 * 4     # 3 ints (+pad), 5 doubles, 5+7 ints (frame_batch took one reserved slot)
    assert C.sizeof(_native.MslCounters) == 12 * 8


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from pyslice_amd import _native
    with pytest.raises(RuntimeError, match="no HIP device"):
        _native.Engine(64, 64, 4, 0.1, 0.1, 0.5, 0.037, 1e-3)
    import pyslice_amd as ps
    xs = np.linspace(0, 6.4, 64, endpoint=False)
    with pytest.raises(RuntimeError):
        ps.Potential(xs, xs, np.array([0.0, 0.5]), np.zeros((1, 3)), [5])
    with pytest.raises(NotImplementedError):
        ps.MultisliceCalculator(force_cpu=True)


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "pyslice_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "multislice_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_host_grid_and_edges_match_oracle(golden):
    from oracle import multislice_oracle as orc
    import pyslice_amd as ps
    from pyslice_amd.potentials import slice_edges
    g = golden("g1_grid")
    for box in g["boxes"]:
        tr = ps.Trajectory(np.array([5, 7]), np.zeros((1, 2, 3)), np.zeros((1, 2, 3)), box, 0.005)
        got = ps.gridFromTrajectory(tr, 0.1, 0.5)
        want = orc.grid_from_box(box, 0.1, 0.5)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        lo, hi = slice_edges(got[2])
        olo, ohi = orc.slice_edges(want[2])
        assert np.array_equal(lo, olo) and np.array_equal(hi, ohi)


def test_host_constants_match_golden(golden):
    from pyslice_amd.multislice import interaction_sigma, wavelength
    g = golden("g2_wavelength")
    for e, l, s in zip(g["eV"], g["wavelength"], g["sigma"]):
        assert wavelength(e) == l and interaction_sigma(e) == s


def test_cache_key_matches_reference_directory_names(golden):
    """MultisliceCalculator._generate_cache_key names the directory the reference names for the same run
    (calculators.py:78-94, 139): reference and build can share psi_data/."""
    import pyslice_amd as ps
    g = golden("g11_cache")
    for c in ("a", "b"):
        pp = [tuple(float(v) for v in p) for p in g[f"probe_positions_{c}"]] if bool(g[f"has_positions_{c}"]) else None
        pos = g[f"positions_{c}"]
        tr = ps.Trajectory(g[f"Z_{c}"], pos, np.zeros_like(pos), g[f"box_{c}"], 0.005)
        calc = ps.MultisliceCalculator.__new__(ps.MultisliceCalculator)        # no device needed for the key
        key = calc._generate_cache_key(tr, float(g[f"aperture_{c}"]), float(g[f"eV_{c}"]), 0.5, 0.1, pp)
        assert "torch_" + key == str(g[f"dir_name_{c}"])


def test_kirkland_table_and_names():
    import pyslice_amd as ps
    t = ps.loadKirkland()
    assert t.shape == (103, 3, 4) and t.dtype == np.float64
    assert ps.getZfromElementName("B") == 5 and ps.getZfromElementName("N") == 7 and ps.getZfromElementName("Au") == 79


def test_trajectory_validation_messages():
    import pyslice_amd as ps
    with pytest.raises(ValueError, match="positions must be"):
        ps.Trajectory(np.array([5]), np.zeros((1, 1, 2)), np.zeros((1, 1, 3)), np.eye(3), 0.005)
    with pytest.raises(ValueError, match="Atom count mismatch"):
        ps.Trajectory(np.array([5, 7]), np.zeros((1, 1, 3)), np.zeros((1, 1, 3)), np.eye(3), 0.005)
    with pytest.raises(ValueError, match="box_matrix"):
        ps.Trajectory(np.array([5]), np.zeros((1, 1, 3)), np.zeros((1, 1, 3)), np.eye(2), 0.005)


def test_synthetic_workload_grids():
    from pyslice_amd.synthetic import box_for_grid, synthetic_trajectory, stem_probe_grid
    import pyslice_amd as ps
    for n, nz in [(256, 50), (512, 100), (1024, 200), (2048, 400)]:
        tr = ps.Trajectory(np.array([5]), np.zeros((1, 1, 3)), np.zeros((1, 1, 3)), box_for_grid(n, nz), 0.005)
        xs, ys, zs, *_ = ps.gridFromTrajectory(tr)
        assert (len(xs), len(ys), len(zs)) == (n, n, nz)
    tr = synthetic_trajectory(64, 6, 3, seed=2)
    assert tr.positions.shape[0] == 3 and (tr.positions[:, :, 2] >= 0).all()
    assert stem_probe_grid(8).shape == (64, 2)


def test_m0_is_only_touched_by_the_addtid_exchange(tmp_path):
    """The lane<->register exchange stores with ds_write_addtid_b32 (inline asm), whose address base is M0.  The compiler
    does not know about that: the scheme is sound only while nothing else in the device code reads or writes M0, and every
    write of M0 is followed by the wait state the add-tid instruction needs.  Checked on the generated ISA."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from concurrent.futures import ThreadPoolExecutor
    from pyslice_amd.build_native import SOURCES

    def device_asm(src):
        out = tmp_path / (os.path.splitext(src)[0] + ".s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "--cuda-device-only", "-S", "-o", str(out),
                        os.path.join(REPO, "pyslice_amd", "csrc", src)], check=True, capture_output=True)
        return [l.strip() for l in open(out)]

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:            # every translation unit of the library
        lines = [l for part in pool.map(device_asm, SOURCES) for l in part]
    m0 = [i for i, l in enumerate(lines) if re.search(r"\bm0\b", l) and not l.startswith(";")]
    assert m0, "no add-tid exchange in the build?"
    for i in m0:
        assert lines[i].startswith("s_mov_b32 m0,"), lines[i]
        assert lines[i + 1].startswith("s_nop"), (lines[i], lines[i + 1])
    assert sum(l.startswith("ds_write_addtid_b32") for l in lines) > 0
    # no register kernel of the library spills or falls below two waves per SIMD (the generic LDS line kernels, the fallback for
    # lengths no register kernel serves, are the only ones allowed private memory)
    text = "\n".join(lines)
    shipped = ["rowT2_pass_kernelILi16E", "rowTW_pass_kernelILb1ELb1E",
               "rowTB_pass_kernelILi32E", "rowTB_pass_kernelILi16E", "rowTB2_pass_kernelILb0ELb0E", "rowTC2_pass_kernel",
               "ifftTB_kernelILi32ELb0E", "ifftTB_kernelILi32ELb1E", "ifftTB_kernelILi16ELb0E", "ifftTB_two_kernel", "ifftTB2_kernel", "ifftTW_kernel",
               "ifftT2_kernelILi16ELb0E", "ifftT2_kernelILi16ELb1E",
               "structure_factor_quad_kernel", "structure_factor_stream_kernel", "structure_factor_stream_bf16_kernel", "structure_factor_edge_kernel", "col_pass_kernelILi32ELi16ELb1E",
               "time_cz_kernelILi16ELi32ELb1E", "time_cz_kernelILi32ELi16ELb1E", "time_cz_kernelILi16ELi32ELb0E", "time_cz_kernelILi32ELi16ELb0E", "tacaw_fold_kernel", "row_pass_pf_kernelILi32E", "row_pass2_kernelILi32E"]
    names = re.findall(r"\.amdhsa_kernel (\S+)", text)
    assert not [n for n in names if re.search(r"rowTP|rowT3|rowTC_pass|structure_factor_mfma|structure_factor_nyquist", n)], "superseded kernels are back"
    for name in shipped:
        m = re.search(r"\.amdhsa_kernel (_ZN3msl\d+" + re.escape(name) + r"\S*)(.*?)\.end_amdhsa_kernel", text, re.S)
        assert m, name
        body = m.group(2)
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        assert scratch == 0, (name, scratch)
        assert vgpr <= (512 if name.startswith("structure_factor_stream") else 256), (name, vgpr)     # (one wave per SIMD by design)
    # the transposing pass of 256 / 1024-point lines (rowt_pass.h): every instantiation with compile-time pass flags fits the registers
    # of two (three) waves per SIMD without private memory; the run-time-flag form (first / last pass of natural-order stacks) spills
    # the t_k line by design and nothing else
    rowt = re.findall(r"\.amdhsa_kernel (_ZN3msl16rowT_pass_kernelILi(\d+)ELi16ELb([01])ELb([01])ELi(n?\d+)EEEvNS_7RowTJobE)(.*?)\.end_amdhsa_kernel", text, re.S)
    assert len(rowt) == 12, [r[0] for r in rowt]
    for name, R, ip, op, fl, body in rowt:
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        assert vgpr <= (256 if R == "32" else 168), (name, vgpr)
        assert (scratch == 0) if fl != "n1" else (scratch <= 400), (name, scratch)
    # ... and waits for its prefetched line with the iteration's 16 stores still in flight: the first run of eight descending vmcnt
    # waits (the leaf level consuming the eight 16-byte loads of a half line) ends at vmcnt(16) or above, not at vmcnt(0)
    hot = re.search(r"\n(_ZN3msl16rowT_pass_kernelILi32ELi16ELb1ELb1ELi3EEEvNS_7RowTJobE:.*?)s_endpgm", "\n".join(lines), re.S).group(1)
    waits = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", hot)]
    runs = [waits[i:i + 8] for i in range(len(waits) - 7) if all(waits[i + j] == waits[i] - j for j in range(8))]
    assert runs and runs[0][-1] >= 16, (waits[:40], runs[:1])
    # the per-lane / wave-split time transforms (105 instantiations): one wave per SIMD where the line needs the 512-register file,
    # never private memory
    timek = re.findall(r"\.amdhsa_kernel (_ZN3msl\d+time_(?:direct|split)_kernel\S*)(.*?)\.end_amdhsa_kernel", text, re.S)
    assert len(timek) == 105, len(timek)
    for name, body in timek:
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)) == 0, name
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)) <= 512, name


def test_default_frame_batch_rule():
    from pyslice_amd.calculators import default_frame_batch
    assert default_frame_batch(64, 200, 1024, 1024) == 4          # C3: about 256 images per launch
    assert default_frame_batch(1, 100, 512, 512) == 112           # C2: 48 GB of transmission stacks bound it (114 -> a multiple of 16)
    assert default_frame_batch(1, 50, 256, 256) == 256            # C1
    assert default_frame_batch(16, 400, 2048, 2048) == 1          # C5: one frame's stacks are 27 GB
    assert default_frame_batch(300, 10, 64, 64) == 1


def test_line_kernel_classes_and_sampling_hint(lib):
    """msl_line_kernel_class (no device needed) and the sampling hint built on it: the reference's own 501 x 491 grid
    (00_probe.py:7-8) has no direct kernel on either axis; a sampling at most 8 % finer puts both axes on one."""
    import types
    from pyslice_amd import _native
    from pyslice_amd.potentials import suggest_sampling
    assert [_native.line_kernel_class(n) for n in (256, 512, 1024, 2048)] == [2, 2, 2, 2]
    assert all(_native.line_kernel_class(n) == 1 for n in (135, 140, 144, 600, 700, 768, 960, 1000, 1400, 1500, 1728))
    assert all(_native.line_kernel_class(n) == 0 for n in (100, 143, 501, 491, 997, 1023, 1792, 2047, 4096))
    fast = _native.fast_lengths(129, 2048)
    assert len(fast) == 99 + 4 and all(any(n % p == 0 for p in (2, 3, 5, 7)) for n in fast)
    def smooth7(n):
        for p in (2, 3, 5, 7):
            while n % p == 0:
                n //= p
        return n == 1
    assert all(smooth7(n) for n in fast)
    # 501 x 491 sits just below 512: its convolution on the 1024-point transform is modelled no dearer than a finer direct grid
    tr = types.SimpleNamespace(box_matrix=np.diag([50.05, 49.05, 49.75]))
    assert (int(50.05 / 0.1) + 1, int(49.05 / 0.1) + 1) == (501, 491) and suggest_sampling(tr, 0.1) is None
    # 520 x 520 sits just above it (2048-point transform, 4 x the work per point): a sampling 1 % finer gives 525 x 525
    tr = types.SimpleNamespace(box_matrix=np.diag([51.95, 51.95, 49.75]))
    s, nx, ny = suggest_sampling(tr, 0.1)
    assert 0.085 <= s <= 0.1 and (int(51.95 / s) + 1, int(51.95 / s) + 1) == (nx, ny) == (525, 525)
    assert suggest_sampling(types.SimpleNamespace(box_matrix=np.diag([102.35, 102.35, 99.75])), 0.1) is None      # 1024 x 1024 already
