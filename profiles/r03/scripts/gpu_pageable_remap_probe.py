"""Does a PAGEABLE destination survive being unmapped and mapped again at the same address?  (No RT_HIP_FLAG_PERSISTENT_FRAME:
the module registers nothing; the HIP runtime stages or pins pageable copies as it sees fit.)  For each buffer size: render
into a fresh mapping, unmap it, map the same address again, render again, and look whether the frame arrived."""
import ctypes as C, mmap, sys
sys.path.insert(0, ".")
import numpy as np, torch  # noqa: F401
import rt_amd
from oracle import binding as oracle

libc = C.CDLL(None, use_errno=True)
libc.mmap.restype = C.c_void_p
libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
libc.munmap.argtypes = [C.c_void_p, C.c_size_t]


def map_at(address, size):
    flags = mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | (0x100000 if address else 0)
    got = libc.mmap(address, size, mmap.PROT_READ | mmap.PROT_WRITE, flags, -1, 0)
    assert got not in (None, C.c_void_p(-1).value), C.get_errno()
    return got


t = rt_amd.HipRayTracer(0)
for width, height in [(256, 144), (1920, 1080), (3840, 2160)]:
    pod = rt_amd.Scene.named("basic").set_sampling(2).describe(width, height)
    want, want_rgb, _ = oracle.render(pod, width, height, seed=3)
    size = width * height * 4
    for want_rgb_too in (False, True):
        address = map_at(None, size)
        results = []
        for round_ in range(3):
            view = np.ctypeslib.as_array((C.c_uint32 * (width * height)).from_address(address)).reshape(height, width)
            view[:] = 0
            check = C.c_void_p()
            stats = rt_amd.capi.RtHipStats()
            rgb = np.zeros((height, width, 3), dtype=np.float32) if want_rgb_too else None
            rt_amd.capi.check(t._lib.rt_hip_render(t._ctx, C.byref(pod), address, width, height, 3, 0, rgb.ctypes.data if rgb is not None else None, C.byref(stats)))
            ok = bool(np.array_equal(view, want)) and (rgb is None or bool(np.array_equal(rgb.view(np.uint32), want_rgb.view(np.uint32))))
            results.append(ok)
            del view
            libc.munmap(address, size)
            again = map_at(address, size)
            assert again == address
        libc.munmap(address, size)
        print(f"{width}x{height} pageable frame{' + float mean into a fresh numpy array' if want_rgb_too else ''}: frame arrived after 0, 1, 2 re-mappings of the same address: {results}", flush=True)
# fresh numpy arrays (np.empty) as destinations, allocated and freed in a loop: the allocator recycles addresses
pod = rt_amd.Scene.named("basic").set_sampling(2).describe(1920, 1080)
want, want_rgb, _ = oracle.render(pod, 1920, 1080, seed=3)
bad = 0
addresses = set()
for i in range(40):
    rgba, rgb, _ = t.render(pod, 1920, 1080, seed=3, want_rgb=True)
    addresses.add(rgba.ctypes.data); addresses.add(rgb.ctypes.data)
    if not (np.array_equal(rgba, want) and np.array_equal(rgb.view(np.uint32), want_rgb.view(np.uint32))):
        bad += 1
    del rgba, rgb
print(f"40 renders into fresh numpy arrays (RGBA + float mean, pageable): {bad} wrong frames, {len(addresses)} distinct buffer addresses")
