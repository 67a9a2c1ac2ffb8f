"""Does the drop-in's render-into-host-memory path care which NUMA node the back buffer lives on?"""
import glob, os, subprocess, sys, time
sys.path.insert(0, ".")
for f in glob.glob("/sys/class/drm/card*/device/numa_node") + glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
    try:
        txt = open(f).read()
        if f.endswith("numa_node"):
            print(f, txt.strip())
    except OSError as e:
        print(f, e)
print("affinity:", sorted(os.sched_getaffinity(0))[:4], "...", len(os.sched_getaffinity(0)), "cpus")
try:
    print(subprocess.run(["lscpu"], capture_output=True, text=True).stdout.split("NUMA")[1:][:1])
    for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
        print(n, open(n).read().strip())
except Exception as e:
    print(e)
code = r'''
import sys, time, os
sys.path.insert(0, ".")
import numpy as np, torch
import rt_amd
from rt_amd import capi
cpus = os.environ.get("PROBE_CPUS")
if cpus:
    lo, hi = map(int, cpus.split("-")); os.sched_setaffinity(0, range(lo, hi + 1))
t = rt_amd.HipRayTracer(0)
pod = rt_amd.Scene.named("basic").set_sampling(256).describe(1920, 1080)
back = np.zeros((1080, 1920), dtype=np.uint32); back[:] = 1
def f():
    return t.render(pod, 1920, 1080, seed=1, flags=capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=back)[2]["render_ms"]
for _ in range(15): f()
ks = sorted(f() for _ in range(60))
print("cpus %-10s kernel-in-drop-in min %.3f med %.3f mean %.3f max %.3f" % (cpus, ks[0], ks[30], sum(ks) / 60, ks[-1]))
'''
nodes = sorted(glob.glob("/sys/devices/system/node/node*/cpulist"))
ranges = [open(n).read().strip().split(",")[0] for n in nodes] or [None]
for rnd in range(2):
    for r in ranges:
        env = dict(os.environ)
        if r: env["PROBE_CPUS"] = r
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(out.stdout.strip() or out.stderr[-300:], flush=True)
