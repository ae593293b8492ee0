#!/usr/bin/env python3
"""Per-kernel fingerprint of the gfx950 instruction stream the compiler emits for one or more .hip sources
(no GPU needed: hipcc --cuda-device-only -S).  Used to show that a restructuring of the kernel sources -- files
split, knobs removed -- left the machine code of the kernels nmrfit_amd.fit() runs untouched: the same hash
before and after means the same instructions in the same order with the same registers.

    python tools/isa_hash.py [-D...] file.hip [file2.hip ...]  > hashes.txt
    python tools/isa_hash.py --diff before.txt after.txt

A line is `<sha1 of the normalised body> <instructions> <demangled kernel name>`.  Normalisation: comments and
directives dropped, basic-block labels renumbered per kernel (their numbers depend on the function's position in
the translation unit)."""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nmrfit_amd", "csrc")


def kernels(asm):
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*; @", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[name] = cur
            cur = None
            continue
        s = line.split(";")[0].rstrip()
        if not s.strip() or s.strip().startswith("."):
            if re.match(r"^\.LBB\d+_\d+:", s.strip()):
                cur.append(re.sub(r"\.LBB\d+_", ".LBB_", s.strip()))
            continue
        cur.append(re.sub(r"\.LBB\d+_", ".LBB_", s.strip()))
    return out


def main():
    args = sys.argv[1:]
    if args and args[0] == "--diff":
        a = {l.split(None, 2)[2]: l.split()[0] for l in open(args[1]).read().splitlines() if l.strip()}
        b = {l.split(None, 2)[2]: l.split()[0] for l in open(args[2]).read().splitlines() if l.strip()}
        bad = 0
        for k in sorted(set(a) | set(b)):
            st = "same" if a.get(k) == b.get(k) else ("only-before" if k not in b else "only-after" if k not in a else "DIFFERENT")
            if st == "DIFFERENT":
                bad += 1
            print("%-12s %s" % (st, k))
        return 1 if bad else 0
    flags = [a for a in args if a.startswith("-")]
    srcs = [a for a in args if not a.startswith("-")]
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in srcs:
            out = os.path.join(tmp, "x.s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
                   "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S", src,
                   "-o", out] + flags
            subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
            ks = kernels(open(out).read())
            names = list(ks)
            dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
            for n, d in zip(names, dem):
                d = re.sub(r"\(.*", "", d.replace("nmrfit::(anonymous namespace)::", "").replace("void ", ""))
                body = "\n".join(ks[n])
                rows.append((hashlib.sha1(body.encode()).hexdigest()[:16], len(ks[n]), d))
    for r in sorted(rows, key=lambda r: r[2]):
        print("%s %6d %s" % r)
    return 0


if __name__ == "__main__":
    sys.exit(main())
