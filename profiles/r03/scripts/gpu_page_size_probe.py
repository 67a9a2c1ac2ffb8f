"""Does the page size behind the back buffer matter to the kernel that stores into it?  Transparent-huge-page settings
of the box, then the headline frame into: a private buffer as numpy hands it out, the same with MADV_NOHUGEPAGE, with
MADV_HUGEPAGE (2 MB aligned), a /dev/shm mapping as it is and with MADV_HUGEPAGE, and a MAP_SHARED|MAP_ANONYMOUS mapping."""
import ctypes, mmap, os, sys, time
sys.path.insert(0, ".")
import numpy as np
import rt_amd
from rt_amd import capi

libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
W, H, SPP = 1920, 1080, 256
P = capi.RT_HIP_FLAG_PERSISTENT_FRAME
MADV_HUGEPAGE, MADV_NOHUGEPAGE = 14, 15
for name in ("enabled", "shmem_enabled", "defrag", "hpage_pmd_size"):
    try:
        print(f"/sys/kernel/mm/transparent_hugepage/{name}: {open('/sys/kernel/mm/transparent_hugepage/' + name).read().strip()}")
    except OSError as e:
        print(name, e)
print("/dev/shm mount:", [l.strip() for l in open("/proc/mounts") if " /dev/shm " in l])
print("hugetlb: nr_hugepages", open("/proc/sys/vm/nr_hugepages").read().strip(), "overcommit", open("/proc/sys/vm/nr_overcommit_hugepages").read().strip(), flush=True)


def huge_kb(addr):
    """AnonHugePages / ShmemPmdMapped of the mapping that holds addr (from /proc/self/smaps)"""
    inside, out = False, {}
    for line in open("/proc/self/smaps"):
        head = line.split()
        if "-" in head[0] and len(head) >= 5 and all(c in "0123456789abcdef-" for c in head[0]):
            a, b = (int(x, 16) for x in head[0].split("-"))
            inside = a <= addr < b
        elif inside and head[0] in ("AnonHugePages:", "ShmemPmdMapped:", "Rss:", "KernelPageSize:"):
            out[head[0].rstrip(":")] = int(head[1])
    return out


def timed(tracer, pod, buf, label):
    buf[:] = 0  # (the caller clears its back buffer: pages exist before the module sees them)
    tracer.render(pod, W, H, flags=P, out=buf)
    for _ in range(40):
        tracer.render(pod, W, H, flags=P, out=buf)
    kernels = []
    for _ in range(30):
        kernels.append(tracer.render(pod, W, H, flags=P, out=buf)[2]["render_ms"])
    print(f"{label}: kernel {np.median(kernels):.4f} ms; {huge_kb(buf.ctypes.data)}", flush=True)
    tracer.forget_frame()


def aligned_private(advice):
    size = W * H * 4
    raw = mmap.mmap(-1, size + (4 << 20), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    base = ctypes.addressof(ctypes.c_char.from_buffer(raw))
    start = (base + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    if advice is not None:
        rc = libc.madvise(start, (size + (2 << 20) - 1) // (2 << 20) * (2 << 20), advice)
        assert rc == 0, ctypes.get_errno()
    arr = np.frombuffer(raw, dtype=np.uint32, count=W * H, offset=start - base).reshape(H, W)
    return arr, raw


pod = rt_amd.Scene.named("basic").set_sampling(SPP).describe(W, H)
t = rt_amd.HipRayTracer(0)
timed(t, pod, np.empty((H, W), dtype=np.uint32), "private, as numpy hands it out")
a, keep1 = aligned_private(None); timed(t, pod, a, "private anonymous mmap, 2 MB aligned, no advice")
a, keep2 = aligned_private(MADV_HUGEPAGE); timed(t, pod, a, "private anonymous mmap, 2 MB aligned, MADV_HUGEPAGE")
a, keep3 = aligned_private(MADV_NOHUGEPAGE); timed(t, pod, a, "private anonymous mmap, 2 MB aligned, MADV_NOHUGEPAGE")
path = f"/dev/shm/rt_hip_page_probe_{os.getpid()}"
size = (W * H * 4 + (2 << 20) - 1) // (2 << 20) * (2 << 20)
with open(path, "wb") as f:
    f.truncate(size)
fd = os.open(path, os.O_RDWR)
for advice, label in ((None, "as it is"), (MADV_HUGEPAGE, "MADV_HUGEPAGE")):
    raw = mmap.mmap(fd, size, flags=mmap.MAP_SHARED)
    base = ctypes.addressof(ctypes.c_char.from_buffer(raw))
    if advice is not None:
        print("madvise rc", libc.madvise(base, size, advice), "errno", ctypes.get_errno())
    arr = np.frombuffer(raw, dtype=np.uint32, count=W * H).reshape(H, W)
    timed(t, pod, arr, f"/dev/shm mapping, {label} (base {'2 MB aligned' if base % (2 << 20) == 0 else 'not 2 MB aligned'})")
    del arr
os.close(fd); os.unlink(path)
raw = mmap.mmap(-1, size, flags=mmap.MAP_SHARED | mmap.MAP_ANONYMOUS)
arr = np.frombuffer(raw, dtype=np.uint32, count=W * H).reshape(H, W)
timed(t, pod, arr, "MAP_SHARED | MAP_ANONYMOUS")
t.close()
