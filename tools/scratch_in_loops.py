#!/usr/bin/env python3
"""Which loops of a kernel touch scratch (register spills)?  usage: scratch_in_loops.py file.s <substring of the function name>
Lists every loop (back edge) of the function whose body holds scratch_load / scratch_store instructions, innermost first."""
import re
import sys

src, want = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, ln in enumerate(lines) if re.match(r"^[A-Za-z_][\w$.]*:", ln) and want in ln and not ln.startswith(".L"))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels, insts = {}, []
for ln in body:
    t = ln.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t)
    if m:
        labels[m.group(1)] = len(insts)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    insts.append(t.split(";")[0].strip())
loops = []
for i, t in enumerate(insts):
    m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i, m.group(1)))
print(f"{want}: {len(insts)} instructions, {len(loops)} loops, scratch ops total",
      sum(1 for t in insts if t.startswith("scratch_")))
for a, b, lab in sorted(loops, key=lambda x: x[1] - x[0]):
    n_l = sum(1 for t in insts[a:b + 1] if t.startswith("scratch_load"))
    n_s = sum(1 for t in insts[a:b + 1] if t.startswith("scratch_store"))
    inner = not any(a <= a2 and b2 <= b and (a2, b2) != (a, b) for a2, b2, _ in loops)
    if n_l + n_s:
        calls = sum(1 for t in insts[a:b + 1] if t.startswith("s_swappc"))
        print(f"  loop {lab} [{a}, {b}] {b - a + 1} insts{' innermost' if inner else ''}: scratch loads {n_l} stores {n_s} calls {calls}")
