#!/usr/bin/env python3
"""Reads the counter files tools/batch_instr_model.sh wrote and fits
    SQ_INSTS_VALU per particle and generation = a + chunks * (b + c * P)
(a launch of the sweep is one PART of the batch: 40 fits in two parts = 20 fits x 204 particles = 4080 waves)
Usage: python tools/batch_instr_model.py gpurun_out/prof/instr_model [particles per launch, default 4080]"""
import collections, csv, glob, os, re, sys
import numpy as np

d = sys.argv[1]
particles = int(sys.argv[2]) if len(sys.argv) > 2 else 4080
rows = []
for sub in sorted(glob.glob(os.path.join(d, "n*_p*"))):
    if not os.path.isdir(sub):
        continue
    m = re.match(r"n(\d+)_p(\d+)$", os.path.basename(sub))
    N, P = int(m.group(1)), int(m.group(2))
    fs = sorted(glob.glob(os.path.join(sub, "*", "*_counter_collection.csv")), key=os.path.getmtime)   # newest run
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if "objective_batch_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    v = np.array(agg["SQ_INSTS_VALU"])
    v = v[len(v) // 4:]          # (generation 0 and the warm-up are not different launches, but skip the first quarter)
    per = v.mean() / particles
    busy = 4 * np.mean(agg["SQ_ACTIVE_INST_VALU"]) / 1024 / (np.mean(agg["SQ_BUSY_CYCLES"]) / 32)
    rows.append((N, P, per, busy))
    print("N=%5d P=%2d: %8.1f VALU instructions per particle and generation (%d launches), VALU busy %.2f" % (N, P, per, len(v), busy))
A = np.array([[1.0, N / 512, N / 512 * P] for N, P, _, _ in rows])
y = np.array([r[2] for r in rows])
coef, res, *_ = np.linalg.lstsq(A, y, rcond=None)
print("fit: a = %.0f (the swarm step: fold, draw, move, per-peak constants, phase seeds, f, personal best), b = %.1f per chunk "
      "(phase, data, residual, block sums), c = %.1f per chunk and peak (= %.2f per unit); residuals %s"
      % (coef[0], coef[1], coef[2], coef[2] / 8, np.round(A @ coef - y, 1)))
a, b, c = coef
tot = a + 8 * b + 48 * c
print("default fit (8 chunks, 6 peaks): %.0f = %.0f + %.0f + %.0f  (%.0f %% swarm step, %.0f %% per chunk, %.0f %% peaks)"
      % (tot, a, 8 * b, 48 * c, 100 * a / tot, 800 * b / tot, 4800 * c / tot))
