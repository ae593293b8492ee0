#!/usr/bin/env python3
"""
Static audit of the hot kernel's gfx950 ISA (no GPU needed): compiles one of the kernel translation units
(default objective_default.hip; --unit objective_batch.hip for the batched kernels) to assembly
with the flags of csrc/build.sh, takes one instantiation of objective_kernel apart into basic
blocks, and counts instructions by class in every block that sits inside a loop.

    tools/isa_audit.py [--unit objective_farfield.hip] [--kernel objective_kernelILi0ELb0ELi0E] [--min 30] [--dump BLOCK] [-D...]

The point of it: SQ_INSTS_VALU says 5.88 VALU instructions per (particle,
point, peak) unit where the algebra needs ~4.6 + the Gaussians; this shows where the rest sits
(v_mov copies, address arithmetic, waitcnts) block by block.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nmrfit_amd", "csrc")


UNIT = "objective_default.hip"
if "--unit" in sys.argv:
    i = sys.argv.index("--unit")
    UNIT = sys.argv[i + 1]
    del sys.argv[i:i + 2]


def compile_asm(extra):
    out = os.path.join(tempfile.gettempdir(), "nmrfit_objective_%d.s" % os.getpid())
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fno-fast-math",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only",
           os.path.join(CSRC, UNIT), "-o", out] + extra
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def classify(op):
    if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
        return "fma64"
    if op.startswith("v_mul_f64"):
        return "mul64"
    if op.startswith("v_add_f64"):
        return "add64"
    if op.startswith("v_rcp_f64"):
        return "rcp64"
    if re.match(r"v_(ldexp|rndne|cvt|max|min|trunc|floor|fract|frexp|div|sqrt|rsq|cmp\w*)_\w*f64", op) or "f64" in op:
        return "other64"
    if op.startswith("v_mov") or op.startswith("v_accvgpr") or op.startswith("v_pk_mov"):
        return "v_mov"
    if op.startswith("v_readfirstlane") or op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "v_lane"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    args = sys.argv[1:]
    kernel = "objective_kernelILi0ELb0ELi0E"
    min_size = 30
    dump = None
    extra = []
    i = 0
    while i < len(args):
        if args[i] == "--kernel":
            kernel = args[i + 1]
            i += 2
        elif args[i] == "--min":
            min_size = int(args[i + 1])
            i += 2
        elif args[i] == "--dump":
            dump = args[i + 1]
            i += 2
        elif args[i] == "--asm":
            asm_path = args[i + 1]
            i += 2
        else:
            extra.append(args[i])
            i += 1
    asm = compile_asm(extra)
    lines = open(asm).read().splitlines()
    start = next(n for n, l in enumerate(lines) if re.match(r"^_ZN6nmrfit.*" + kernel + r".*:\s*(;.*)?$", l))
    end = next(n for n in range(start, len(lines)) if lines[n].strip().startswith(".Lfunc_end"))
    name = lines[start].split(":")[0]
    blocks = collections.OrderedDict()
    cur = "entry"
    blocks[cur] = []
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        blocks[cur].append(t.split(";")[0].strip())
    order = list(blocks)
    index = {b: n for n, b in enumerate(order)}
    # loops: a branch to an earlier (or the same) block closes a loop over [target, here]
    in_loop = collections.defaultdict(int)
    loops = []
    for b, ins in blocks.items():
        for t in ins:
            m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
            if m and m.group(1) in index and index[m.group(1)] <= index[b]:
                loops.append((m.group(1), b))
                for k in range(index[m.group(1)], index[b] + 1):
                    in_loop[order[k]] += 1
    meta = {}
    for n in range(end, len(lines)):
        if ".name:" in lines[n] and kernel in lines[n]:
            for k in range(max(0, n - 40), min(len(lines), n + 40)):
                m = re.match(r"\s*\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s*(\d+)", lines[k])
                if m:
                    meta[m.group(1)] = int(m.group(2))
            break
    total = sum(len(v) for v in blocks.values())
    print("kernel %s" % name)
    print("  %d instructions in %d blocks; %s" % (total, len(blocks), ", ".join("%s=%d" % kv for kv in sorted(meta.items()))))
    print("  loops (head <- latch): %s" % ", ".join("%s<-%s" % l for l in loops[:40]))
    classes = ["fma64", "mul64", "add64", "rcp64", "other64", "v_mov", "v_lane", "valu_other", "lds", "vmem", "smem", "salu",
               "waitcnt", "branch", "other"]
    print("  %-12s %5s %5s | " % ("block", "depth", "n") + " ".join("%7s" % c[:7] for c in classes))
    for b, ins in blocks.items():
        if len(ins) < min_size and not (in_loop[b] and len(ins) >= 8):
            continue
        c = collections.Counter(classify(t.split()[0]) for t in ins)
        print("  %-12s %5d %5d | " % (b, in_loop[b], len(ins)) + " ".join("%7d" % c[k] for k in classes))
    if dump:
        print("\n---- %s ----" % dump)
        for t in blocks[dump]:
            print("   ", t)
    os.unlink(asm)


if __name__ == "__main__":
    main()
