#!/usr/bin/env python3
"""Socket power and shader clock (hwmon files of the device) while a device batch of default fits runs: is the batched
kernel, like the headline kernel, paced by the part's power cap?  Prints one line per second for ~8 s."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from nmrfit_amd import _cabi, synth
from nmrfit_amd.batch import FitBatch

K = int(sys.argv[1]) if len(sys.argv) > 1 else 100
specs = [synth.make_spectrum(4096, 6, seed=100 + k % 8) for k in range(K)]
fb = FitBatch([(s["w"], s["u"], s["v"], s["weights"]) for s in specs], [s["lower"] for s in specs], [s["upper"] for s in specs],
              swarmsize=204, seeds=list(range(K)), minstep=-1.0, minfunc=-1.0)
probe = bench.PowerProbe(_cabi.device_pci_bus_id(0))
fb.run(50, 50)
done = []
def work():
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 8.0:
        fb.run(500, 500); n += 500
    done.append((n, time.perf_counter() - t0))
th = threading.Thread(target=work); th.start()
t0 = time.perf_counter()
while th.is_alive():
    time.sleep(1.0)
    probe.sample()
    if probe.samples:
        w, mhz = probe.samples[-1]
        print("t=%4.1f s  socket power %6.0f W  sclk %5.0f MHz" % (time.perf_counter() - t0, w, mhz), flush=True)
th.join()
n, dt = done[0]
s = probe.summary()
print("batch of %d default fits: %.2f us per fit and generation over %d generations; %s" % (K, dt / n / K * 1e6, n, s and {k: s[k] for k in ("socket_power_w", "power_cap_w", "sclk_mhz")}))
fb.close()
