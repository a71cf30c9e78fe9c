#!/usr/bin/env python3
"""
Where a kernel's scratch (spill) instructions sit relative to its loops, from the ISA:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --offload-device-only -S -o unit.s upmix_amd/csrc/upx_reg_zoom512.hip
    python scripts/scratch_in_loops.py unit.s [name filter ...]

Loops are found from backward branches (label earlier in the function than the branch).  For every loop: its length
in instructions, the scratch loads / stores inside it, and whether it contains the transform loop's hallmark
instructions (s_setprio: the priority turn at the top of a transform; s_barrier count; saveexec = per-lane
branches: the signal-edge body checks every sample, the interior bodies have none around their memory operations).  A spill reload inside a
frame loop is a vmcnt(0) wait; one in the prologue / epilogue or in a short tail loop is harmless.
"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        return out if len(out) == len(names) else names
    except OSError:
        return names


def main():
    path, filters = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    names = demangle([n for _, n in starts])
    for (i, _), e, name in zip(starts, ends, names):
        if filters and not all(f in name for f in filters):
            continue
        body = lines[i:e]
        is_insn = lambda l: l.startswith("\t") and not l.strip().startswith((".", ";"))   # noqa: E731
        lab = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for k, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in lab and lab[m.group(1)] < k:
                loops.append((lab[m.group(1)], k, m.group(1)))
        scratch = [(k, l.strip().split()[0]) for k, l in enumerate(body) if "scratch_" in l and is_insn(l)]
        n_insn = sum(map(is_insn, body))
        print(f"{name}\n  {n_insn} instructions, {len(scratch)} scratch instructions "
              f"({sum(1 for _, o in scratch if 'load' in o)} loads, {sum(1 for _, o in scratch if 'store' in o)} stores)")
        for a, b, t in sorted(loops, key=lambda x: x[0] - x[1]):
            inside = [o for k, o in scratch if a <= k <= b]
            seg = body[a:b + 1]
            n = sum(map(is_insn, seg))
            if n < 40 and not inside:
                continue
            print(f"  loop {t:>10s}: {n:5d} instructions, s_barrier {sum('s_barrier' in l for l in seg):2d}, "
                  f"s_setprio {sum('s_setprio' in l for l in seg):2d}, global_store {sum('global_store' in l for l in seg):3d}, "
                  f"saveexec {sum('saveexec' in l for l in seg):3d}, "
                  f"scratch loads {sum('load' in o for o in inside)}, scratch stores {sum('store' in o for o in inside)}")
        outside = [o for k, o in scratch if not any(a <= k <= b for a, b, _ in loops)]
        print(f"  outside every loop: scratch loads {sum('load' in o for o in outside)}, stores {sum('store' in o for o in outside)}")


if __name__ == "__main__":
    main()
