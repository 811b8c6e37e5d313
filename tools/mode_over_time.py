#!/usr/bin/env python
"""One resident 1 M-unit batch, passes back to back for ~12 s: pass time (100 launches per sample) beside the clocks and
power the driver reports.  Does the pass switch modes by itself?   python tools/mode_over_time.py"""
import glob, os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
import bench
from svtyper_amd import hip
os.environ["SVT_PLACEMENT_TRIALS"] = "1"
b = bench.generate("c3_mixed_1m", 1_000_000, 0, bench.usable_cpus())
def sysfs(name):
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/" + name)):
        try:
            txt = open(f).read().strip().splitlines()
            cur = [l for l in txt if l.endswith("*")]
            out.append((cur or txt)[-1].strip())
        except Exception as e:
            out.append("?")
    return out
def hwmon(name):
    out = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/" + name)):
        try: out.append(open(f).read().strip())
        except Exception: out.append("?")
    return out
with hip.DeviceBatch(b, 0, 0) as d:
    d.genotype(sync=True)
    t0 = time.time()
    i = 0
    while time.time() - t0 < 12:
        ms = d.genotype_timed(100) / 100
        if i % 4 == 0:
            print("t %5.2f s  pass %.4f ms  sclk %s mclk %s power %s temp %s" % (time.time() - t0, ms, sysfs("pp_dpm_sclk"), sysfs("pp_dpm_mclk"),
                  hwmon("power1_average") or hwmon("power1_input"), hwmon("temp1_input")), flush=True)
        else:
            print("t %5.2f s  pass %.4f ms" % (time.time() - t0, ms), flush=True)
        i += 1
        if i == 150: time.sleep(1.0)      # an idle second in the middle
