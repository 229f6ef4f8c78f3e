#!/usr/bin/env python3
"""Approximate VGPR liveness over a straight-line stretch of gfx950 assembly (a kernel body as hipcc -save-temps prints it).

    tools/asm_pressure.py file.s [first_line last_line]

Backward scan: an instruction's first operand is its destination unless the mnemonic is a store / write / compare; every
other v-register operand is a source.  Branches inside the stretch are ignored (the slice-loop kernels' bodies are straight-line
apart from the guarded prefetch loads), so read the numbers as a profile, not as the allocator's exact count.  Prints the live
count every STEP lines and the peak with its line number.
"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
NODST = ("store", "ds_write", "v_cmp", "s_", "buffer_store", "global_store", "scratch_store", "v_nop", "ds_bpermute_none")


def regs(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.append(int(m.group(1)))
        else:
            out.extend(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def main():
    path = sys.argv[1]
    lines = open(path).read().split("\n")
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(lines)
    step = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    live = set()
    prof = {}
    peak, peak_line = 0, hi
    loop = len(sys.argv) > 5 and sys.argv[5] == "loop"        # the stretch is a loop body: iterate once to get the live-out set
    for sweep in range(2 if loop else 1):
      peak, peak_line = 0, hi
      for i in range(hi, lo - 1, -1):
          ln = lines[i - 1].split(";")[0].strip()
          if not ln or ln.startswith(".") or ln.endswith(":"):
              prof[i] = len(live)
              continue
          parts = ln.split(None, 1)
          mn = parts[0]
          ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
          if ops and not mn.startswith(NODST):
              dst = regs(ops[0])
              srcs = [r for o in ops[1:] for r in regs(o)]
              if mn.startswith(("v_fmac", "v_mac")):          # destination is also a source
                  srcs += dst
              for r in dst:
                  live.discard(r)
              live.update(srcs)
          else:
              live.update(r for o in ops for r in regs(o))
          prof[i] = len(live)
          if len(live) > peak:
              peak, peak_line = len(live), i
    for i in range(lo, hi + 1, step):
        print(f"line {i:6d}: live {prof.get(i, 0):4d}   {lines[i - 1].strip()[:70]}")
    print(f"peak {peak} at line {peak_line}: {lines[peak_line - 1].strip()[:90]}")


if __name__ == "__main__":
    main()
