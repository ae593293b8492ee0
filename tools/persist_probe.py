import os, sys, time
sys.path.insert(0, "/root/repo")
from nmrfit_amd import synth, pso
from nmrfit_amd.equations import Evaluator
for (S, N, P) in [(204, 4096, 6), (1024, 4096, 6)]:
    sp = synth.make_spectrum(N, P, seed=1)
    with Evaluator(sp["w"], sp["u"], sp["v"], sp["weights"]) as ev:
        sw = pso.DeviceSwarm(ev, sp["lower"], sp["upper"], S, seed=3, minfunc=-1.0, minstep=-1.0)
        sw.set_persistent(True)
        sw.run(200, check_every=100)
        sw.close()
