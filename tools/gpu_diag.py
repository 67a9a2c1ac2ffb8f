import sys; sys.path.insert(0, '.')
import numpy as np, rt_amd
from oracle import binding as oracle
t = rt_amd.HipRayTracer(0)
print("exhaustive:", t.kat_exhaustive_math())
w,h,seed=128,72,7
scene = rt_amd.Scene.named("basic").set_sampling(4)
pod = scene.describe(w,h)
wr, wf, ws = oracle.render(pod,w,h,seed=seed)
for flags in (0,2,1):
    gr, gf, st = t.render(pod,w,h,seed=seed,flags=flags,want_rgb=True)
    bad = (gf.view(np.uint32)!=wf.view(np.uint32)).any(-1)
    print(st['kernel'], 'bad pixels', bad.sum(), 'segments', st['segments'], ws['segments'])
    for y,x in zip(*np.nonzero(bad)):
        print('  ', x,y, gf[y,x], wf[y,x], hex(gr[y,x]), hex(wr[y,x]))
