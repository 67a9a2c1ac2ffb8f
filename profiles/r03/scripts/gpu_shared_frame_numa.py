"""Where do the pages of a back buffer live, and what does that cost the kernel that stores into it?
A private buffer (numpy) and a shared mapping (/dev/shm file, what a frame group renders into), each rendered into by a
plain single-GPU context; the NUMA node of every page before and after the first render (move_pages(2) as a query), the
kernel time, and the same with the process's CPU affinity moved to the other socket for the first touch."""
import ctypes, os, sys, time
sys.path.insert(0, ".")
import numpy as np
import rt_amd
from rt_amd import capi

libc = ctypes.CDLL(None, use_errno=True)
W, H, SPP = 1920, 1080, 256
P = capi.RT_HIP_FLAG_PERSISTENT_FRAME


def nodes_of(buf):
    page = os.sysconf("SC_PAGESIZE")
    addr = buf.ctypes.data
    first = (addr + page - 1) // page * page
    count = (addr + buf.nbytes - first) // page
    pages = (ctypes.c_void_p * count)(*[first + i * page for i in range(count)])
    status = (ctypes.c_int * count)()
    rc = libc.syscall(279, 0, ctypes.c_ulong(count), pages, None, status, 0)  # SYS_move_pages on x86-64
    if rc != 0:
        return f"move_pages failed: errno {ctypes.get_errno()}"
    vals, counts = np.unique(np.array(status[:]), return_counts=True)
    return {int(v): int(c) for v, c in zip(vals, counts)}


def gpu_node():
    import glob
    for f in glob.glob("/sys/class/drm/renderD*/device/numa_node"):
        try:
            os.close(os.open(f.replace("device/numa_node", "").rstrip("/").replace("/sys/class/drm/", "/dev/dri/"), os.O_RDWR))
            return int(open(f).read())
        except OSError:
            continue
    return None


def timed(tracer, pod, buf, label):
    before = nodes_of(buf)
    tracer.render(pod, W, H, flags=P, out=buf)
    after = nodes_of(buf)
    for _ in range(40):
        tracer.render(pod, W, H, flags=P, out=buf)
    kernels, walls = [], []
    for _ in range(30):
        t0 = time.perf_counter(); st = tracer.render(pod, W, H, flags=P, out=buf)[2]; walls.append((time.perf_counter() - t0) * 1e3); kernels.append(st["render_ms"])
    print(f"{label}: pages by node before {before} after the first render {after}; kernel {np.median(kernels):.4f} ms, wall {np.median(walls):.4f} ms", flush=True)
    tracer.forget_frame()


print("GPU hangs off NUMA node", gpu_node(), "; this process may run on CPUs", len(os.sched_getaffinity(0)), "of", os.cpu_count(), flush=True)
pod = rt_amd.Scene.named("basic").set_sampling(SPP).describe(W, H)
t = rt_amd.HipRayTracer(0)
all_cpus = os.sched_getaffinity(0)
node_cpus = {}
for n in (0, 1):
    try:
        text = open(f"/sys/devices/system/node/node{n}/cpulist").read().strip()
        cpus = set()
        for part in text.split(","):
            a, _, b = part.partition("-")
            cpus |= set(range(int(a), int(b or a) + 1))
        node_cpus[n] = cpus & all_cpus
    except OSError:
        pass
print("CPUs of this process per node:", {n: len(c) for n, c in node_cpus.items()}, flush=True)
for n, cpus in node_cpus.items():
    if not cpus:
        continue
    os.sched_setaffinity(0, cpus)
    private = np.zeros((H, W), dtype=np.uint32)  # first touch on node n
    timed(t, pod, private, f"private buffer first touched on node {n}")
    path = f"/dev/shm/rt_hip_numa_probe_{os.getpid()}_{n}"
    np.zeros((H, W), dtype=np.uint32).tofile(path)
    shared = np.memmap(path, dtype=np.uint32, mode="r+", shape=(H, W))
    timed(t, pod, shared, f"shared mapping written on node {n}")
    os.environ["RT_HIP_NUMA_MOVE"] = "0"
    shared2 = np.memmap(path, dtype=np.uint32, mode="r+", shape=(H, W))
    timed(t, pod, shared2, f"shared mapping written on node {n}, RT_HIP_NUMA_MOVE=0")
    del os.environ["RT_HIP_NUMA_MOVE"]
    del shared, shared2
    os.unlink(path)
os.sched_setaffinity(0, all_cpus)
t.close()
