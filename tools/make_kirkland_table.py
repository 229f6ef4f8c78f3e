"""Build pyslice_amd/data/kirkland_abcd.npy from the public Kirkland parameter table.

The Kirkland (Advanced Computing in Electron Microscopy, App. C) scattering-factor
parameters are tabulated constants.  The reference reads them from a text file
(/root/reference/kirkland.txt, parsed at src/multislice/potentials.py:161-172: per element
a header line then 3 rows x 4 numbers in the order a1 b1 a2 b2 / a3 b3 c1 d1 / c2 d2 c3 d3).
This script converts that table once into a dense float64 array of shape (103, 3, 4) whose
last axis is (a_i, b_i, c_i, d_i) -- the layout both the oracle and the HIP library consume.

Run in the build container only (the text table does not travel):
    python tools/make_kirkland_table.py /root/reference/kirkland.txt
"""
import sys
import numpy as np


def parse(path):
    rows = []
    with open(path) as fh:
        lines = [ln.strip() for ln in fh if ln.strip()]
    i = 0
    while i < len(lines) and len(rows) < 103:
        if not lines[i].startswith("Z="):
            i += 1
            continue
        z = int(lines[i].split(",")[0].split("=")[1])
        assert z == len(rows) + 1, (z, len(rows))
        nums = []
        for k in range(1, 4):
            nums.extend(float(t) for t in lines[i + k].split())
        a1, b1, a2, b2, a3, b3, c1, d1, c2, d2, c3, d3 = nums
        rows.append([[a1, b1, c1, d1], [a2, b2, c2, d2], [a3, b3, c3, d3]])
        i += 4
    out = np.asarray(rows, dtype=np.float64)
    assert out.shape == (103, 3, 4), out.shape
    return out


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/kirkland.txt"
    tab = parse(src)
    np.save("pyslice_amd/data/kirkland_abcd.npy", tab)
    print("wrote pyslice_amd/data/kirkland_abcd.npy", tab.shape, tab[4])
