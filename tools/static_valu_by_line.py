#!/usr/bin/env python3
"""Static VALU instruction counts of ONE kernel attributed to source lines (no GPU needed): compiles a .hip unit to gfx950
assembly with -gline-tables-only and sums the v_* instructions between .loc directives.  The prologue of the objective
kernels is straight-line code, so its static count is what a wave executes once; code inside the chunk loop exists once
per loop copy (group form x full / ragged chunk) and is executed per chunk.
    python tools/static_valu_by_line.py nmrfit_amd/csrc/objective_batch.hip 'objective_batch_kernelILi0ELi4ELb1ELi0E' [bucket]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, mangled = sys.argv[1], sys.argv[2]
bucket = int(sys.argv[3]) if len(sys.argv) > 3 else 10
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "x.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-fno-fast-math",
                    "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "nmrfit_amd", "csrc"), "--cuda-device-only", "-S",
                    "-gline-tables-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and mangled in l and l.rstrip().endswith(tuple(":" + c for c in " ;")) or (l.startswith("_Z") and mangled in l and ":" in l))
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur, by = None, collections.Counter()
for l in lines[start:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if t.startswith("v_") and not t.startswith(("v_readfirstlane", "v_readlane")):
        by[cur] += 1
print("kernel %s: %d static VALU instructions" % (mangled, sum(by.values())))
per_file = collections.Counter()
for (f, ln), c in by.items():
    per_file[f] += c
print("by file:", per_file.most_common())
b = collections.Counter()
for (f, ln), c in by.items():
    b[(f, ln // bucket * bucket)] += c
for (f, ln), c in sorted(b.items(), key=lambda x: (x[0][0] or "", x[0][1])):
    if c >= 8:
        print("%-28s lines %4d-%-4d %5d" % (f, ln, ln + bucket - 1, c))
