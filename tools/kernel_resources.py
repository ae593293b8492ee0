#!/usr/bin/env python3
"""Register / scratch / LDS usage of every kernel of libnmrfit_amd.so, from the compiler's own
remarks (no GPU needed): compiles each .hip of nmrfit_amd/csrc for gfx950 with
-Rpass-analysis=kernel-resource-usage and prints one line per kernel.  Spills show up as a
non-zero `scratch` column; `tools/kernel_resources.py --check` exits 1 if any objective_kernel
instantiation that nmrfit_amd.fit() can select (variants DEFAULT / FARFIELD / NOREC, with or without
the imaginary channel, plus every residual form) uses scratch memory.

    python tools/kernel_resources.py [--check] [extra hipcc flags, e.g. -DNMRFIT_POINTS=4]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nmrfit_amd", "csrc")
VARIANTS = {0: "DEFAULT", 1: "BASELINE", 2: "NOSKIP", 3: "SINGLE", 4: "QUAD", 5: "STAGED", 6: "FARFIELD", 7: "NOREC", 8: "FARFIELD32"}


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\(.*", "", o.replace("nmrfit::(anonymous namespace)::", "").replace("void ", "")) for o in out]


def resources(extra):
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("objective_default.hip", "objective_farfield.hip", "objective_farfield32.hip", "objective_norec.hip", "objective_batch.hip", "objective_batch_im.hip", "objective_batch_im2.hip", "objective_batch_im2f.hip", "objective.hip", "pso.hip", "batch.hip"):
            if not os.path.exists(os.path.join(CSRC, src)):
                continue
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
                   "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-c", os.path.join(CSRC, src),
                   "-o", os.path.join(tmp, "x.o"), "-Rpass-analysis=kernel-resource-usage"] + extra
            err = subprocess.run(cmd, capture_output=True, text=True).stderr
            cur = None
            for line in err.splitlines():
                m = re.search(r"remark: (.+?) \[-Rpass-analysis", line)
                if not m:
                    continue
                k, _, v = m.group(1).partition(": ")
                if k == "Function Name":
                    cur = {"name": v}
                    rows.append(cur)
                elif cur is not None:
                    cur[k.strip()] = v.strip()
    names = demangle([r["name"] for r in rows])
    for r, n in zip(rows, names):
        r["pretty"] = n
    return rows


def main():
    check = "--check" in sys.argv
    extra = [a for a in sys.argv[1:] if a != "--check"]
    rows = resources(extra)
    bad = []
    print("%-58s %5s %5s %8s %6s %8s %8s" % ("kernel", "VGPR", "AGPR", "scratch", "waves", "LDS", "s-spill"))
    for r in rows:
        name = r["pretty"]
        m = re.match(r"objective_kernel<(\d+), (true|false), (\d+), (\d+)>", name)
        label = name
        selectable = False
        if m:
            v, wr, fi = int(m.group(1)), m.group(2) == "true", int(m.group(3))
            label = "objective_kernel<%s,%s,fit_im=%d%s>" % (VARIANTS.get(v, v), "residual" if wr else "objective", fi,
                                                            ",8 waves" if m.group(4) == "8" else "")
            selectable = v in (0, 6, 7, 8)
        mb = re.match(r"objective_batch_kernel<(\d+), (\d+), (true|false), (\d+)>", name)
        if mb:
            label = "objective_batch_kernel<%s,%s%s>" % (VARIANTS.get(int(mb.group(1)), mb.group(1)),
                                                         "wave=particle" if mb.group(3) == "true" else "%s waves" % mb.group(2),
                                                         ",fit_im=%s" % mb.group(4) if mb.group(4) != "0" else "")
        scratch = int(r.get("ScratchSize [bytes/lane]", "0"))
        # (s-spill: scalar registers parked in VGPR lanes -- v_writelane / v_readlane, which are VALU instructions: in a
        # loop they cost issue slots like arithmetic does)
        print("%-58s %5s %5s %8d %6s %8s %8s" % (label[:58], r.get("VGPRs", "?"), r.get("AGPRs", "?"), scratch,
                                                r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"),
                                                r.get("SGPRs Spill", "?")))
        if selectable and scratch:
            bad.append(label)
    if check and bad:
        print("SPILLS in kernels fit() can select:", bad)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
