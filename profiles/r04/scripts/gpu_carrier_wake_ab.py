"""What waking the carrier's helpers costs a frame, and where: builds of librt_hip.so side by side, helpers hot (the default:
they keep looking for the next frame for 150 us) and helpers ASLEEP at every frame (RT_HIP_CARRIER_STAY_HOT_US=0), the default
frame mode against the zero-copy opt-in.

    python tools/gpu_carrier_wake_ab.py [reps] [lib.so ...]       (libs under rt_amd/lib/; one process per build and setting)

Per run: wall clock of rt_hip_render as the plug-in calls it (stats == NULL), median / mean / max over <reps> frames, and
host_issue_ms (entry of the call -> launch issued) of calls that keep stats."""
import os, subprocess, sys
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
libs = sys.argv[2:] or ["librt_hip_before.so", "librt_hip.so"]
code = f"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
import rt_amd
from rt_amd import capi
t = rt_amd.HipRayTracer(0)
pod = rt_amd.Scene.named('basic').set_sampling(256).describe(1920, 1080)
def run(flags, gap_s=0.0):
    back = np.zeros((1080, 1920), dtype=np.uint32)
    for _ in range(4):
        t.render(pod, 1920, 1080, seed=1, flags=flags, out=back, stats=False)
    wall = []
    for _ in range({reps}):
        if gap_s:
            time.sleep(gap_s)
        t0 = time.perf_counter(); t.render(pod, 1920, 1080, seed=1, flags=flags, out=back, stats=False); wall.append((time.perf_counter() - t0) * 1e3)
    issue, kernel, wall_stats = [], [], []
    for _ in range(20):
        if gap_s:
            time.sleep(gap_s)
        t0 = time.perf_counter(); st = t.render(pod, 1920, 1080, seed=1, flags=flags, out=back)[2]; wall_stats.append((time.perf_counter() - t0) * 1e3)
        issue.append(t.phases()['host_issue_ms']); kernel.append(st['render_ms'])
    wall.sort(); issue.sort(); kernel.sort(); wall_stats.sort()
    if flags & capi.RT_HIP_FLAG_PERSISTENT_FRAME:
        t.forget_frame()
    return 'wall median %.4f mean %.4f max %.4f ms; with stats: wall %.4f, kernel %.4f, host_issue %.4f ms (medians)' % (wall[len(wall)//2], sum(wall)/len(wall), wall[-1], wall_stats[10], kernel[10], issue[10])
for gap in (0.0, 0.002, 0.016):
    print('default,   frames %4.1f ms apart' % (gap * 1e3), run(0, gap))
    print('zero-copy, frames %4.1f ms apart' % (gap * 1e3), run(capi.RT_HIP_FLAG_PERSISTENT_FRAME, gap))
"""
for rnd in range(2):
    for lib in libs:
        for hot in ("", "0"):
            env = dict(os.environ, RT_HIP_LIBRARY=os.path.abspath(f"rt_amd/lib/{lib}"))
            if hot:
                env["RT_HIP_CARRIER_STAY_HOT_US"] = hot
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            print(f"--- {lib}  {'helpers asleep between frames (STAY_HOT_US=0)' if hot else 'helpers hot (default)'}", flush=True)
            print(out.stdout.strip() or out.stderr[-600:], flush=True)
