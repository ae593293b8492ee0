#!/usr/bin/env python3
"""One stream against two for a device batch (NMRFIT_BATCH_STREAMS): default fits per second, stopping rule off."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for streams in ("1", "2", "4"):
    env = dict(os.environ, NMRFIT_BATCH_STREAMS=streams)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "batch_fits.py"), "2000", "16,40,100,200", "wave"], env=env,
                         capture_output=True, text=True).stdout
    print("## NMRFIT_BATCH_STREAMS=%s" % streams)
    print("".join(l + "\n" for l in out.splitlines() if l.startswith("K=")), end="", flush=True)
