#!/usr/bin/env python3
"""
Opcode histogram of a kernel's loops from the ISA (hipcc --offload-device-only -S):
    python scripts/isa_hist.py unit.s <name filter ...> [--min 500]
For every loop of at least --min instructions: the instruction count by class (packed f32, scalar f32 VALU, moves,
lane swaps, LDS, vector memory, scalar, waits).  What the interior frame-pair loop of a fused kernel spends its issue
slots on.
"""
import collections
import re
import subprocess
import sys


def classify(op):
    if op.startswith("v_pk_"):
        return "pk_mov" if "mov" in op else "pk_f32"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "v_mov"
    if op.startswith("v_permlane") or "dpp" in op or op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith(("v_cndmask", "v_cmp")):
        return "v_sel/cmp"
    if op.startswith("v_"):
        return "v_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_barrier", "s_setprio", "s_nop", "s_sleep")):
        return op
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_len = int(sys.argv[sys.argv.index("--min") + 1]) if "--min" in sys.argv else 500
    if "--min" in sys.argv:
        args = [a for a in args if a != sys.argv[sys.argv.index("--min") + 1]]
    path, filters = args[0], args[1:]
    lines = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    ends = [i for i, l in enumerate(lines) if l.startswith(".Lfunc_end")]
    try:
        names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.splitlines()
    except OSError:
        names = [n for _, n in starts]
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
        print(name)
        for a, b, t in sorted(loops, key=lambda x: x[0] - x[1]):
            seg = [l.strip().split()[0] for l in body[a:b + 1] if is_insn(l)]
            if len(seg) < min_len:
                continue
            h = collections.Counter(classify(o) for o in seg)
            top = collections.Counter(o for o in seg if classify(o) in ("v_other", "v_mov", "pk_mov")).most_common(8)
            print(f"  loop {t}: {len(seg)} instructions: " + ", ".join(f"{k} {v}" for k, v in h.most_common()))
            print("      " + ", ".join(f"{k} {v}" for k, v in top))


if __name__ == "__main__":
    main()
