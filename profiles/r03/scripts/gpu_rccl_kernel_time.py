"""Does a context that holds an RCCL communicator render the same frame slower?  Kernel time (HIP events) and wall time of the
headline frame through: a plain context; a one-member rt_hip_create_multi context (RCCL, ncclCommInitAll); a joined rank
(rt_hip_create + rt_hip_join_ranks, world 1); and the plain context again with the communicators still alive."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch  # noqa: F401
import rt_amd
from rt_amd import capi

P = capi.RT_HIP_FLAG_PERSISTENT_FRAME
pod = rt_amd.Scene.named("basic").set_sampling(256).describe(1920, 1080)


def measure(name, tracer, frames=30):
    back = np.zeros((1080, 1920), dtype=np.uint32)
    for _ in range(50):
        tracer.render(pod, 1920, 1080, flags=P, out=back)
    walls, kernels, phases = [], [], []
    for _ in range(frames):
        t0 = time.perf_counter()
        st = tracer.render(pod, 1920, 1080, flags=P, out=back)[2]
        walls.append((time.perf_counter() - t0) * 1e3)
        kernels.append(st["render_ms"])
        phases.append(tracer.phases())
    p = {k: round(float(np.median([x[k] for x in phases])), 4) for k in ("render_ms", "gather_ms", "assemble_ms", "copy_ms", "host_issue_ms", "host_wait_ms")}
    print(f"{name:46s} kernel {np.median(kernels):.4f} ms  wall {np.median(walls):.4f} ms  {p}", flush=True)
    tracer.forget_frame()


plain = rt_amd.HipRayTracer(device=0)
measure("plain context", plain)
multi = rt_amd.HipRayTracer(devices=[0])
measure("rt_hip_create_multi, one member (RCCL)", multi)
measure("plain context, communicator alive", plain)
rank = rt_amd.HipRayTracer(device=0)
rank.join_ranks(0, 1, rt_amd.unique_id(), timeout_ms=60000)
measure("joined rank, world 1 (RCCL)", rank)
multi.close(); rank.close()
measure("plain context, communicators destroyed", plain)
peer = rt_amd.HipRayTracer(devices=[0], peer_copy=True)
measure("one member, peer-copy transport (no RCCL)", peer)
