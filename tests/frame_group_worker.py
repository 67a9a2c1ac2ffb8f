"""One rank of a frame group, as a process of its own (tests/test_gpu_frame_group.py starts `world` of these on the one
GPU of the box — a frame group needs no RCCL, so the ranks may share a device).

    python tests/frame_group_worker.py <rank> <world> <group name> <frame file> <W> <H> <spp> <scene> <frames> <mode> <out dir>

modes: ok | private (rank 1 renders into a private array) | mismatch (rank 1 uses another seed in frame 2) |
       dies (the last rank's process ends after frame 1 without a word) | leaves (the last rank destroys its tracer after
       frame 1) | remap (every rank maps the frame file anew before frame 2) | float (rank 0 asks for the float mean)
Prints one JSON line: {"rank", "frames_done", "error", "ms": [...], "stats": {...}, "info": {...}}.
"""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402

import rt_amd  # noqa: E402
from rt_amd import capi  # noqa: E402


def page_nodes(buf):
    """NUMA node of every whole page of `buf` (move_pages(2) as a query; negative = -errno, e.g. -2 not present)"""
    import ctypes

    libc = ctypes.CDLL(None, use_errno=True)
    page = os.sysconf("SC_PAGESIZE")
    first = (buf.ctypes.data + page - 1) // page * page
    count = (buf.ctypes.data + buf.nbytes - first) // page
    pages = (ctypes.c_void_p * count)(*[first + i * page for i in range(count)])
    status = (ctypes.c_int * count)()
    if libc.syscall(279, 0, ctypes.c_ulong(count), pages, None, status, 0) != 0:  # SYS_move_pages (x86-64)
        return None
    return list(status)


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    group_name, frame_file = sys.argv[3], sys.argv[4]
    width, height, spp = int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    scene_name, frames, mode, out_dir = sys.argv[8], int(sys.argv[9]), sys.argv[10], Path(sys.argv[11])
    result = {"rank": rank, "frames_done": 0, "error": None, "ms": [], "stats": None, "info": None, "member_stats": None}
    scene = rt_amd.Scene.named(scene_name) if not scene_name.endswith(".toml") else rt_amd.Scene.load(scene_name)
    pod = scene.set_sampling(spp).describe(width, height)
    if mode == "numa":
        os.environ["RT_HIP_NUMA_NODE"] = str(rank % 2)
    tracer = rt_amd.HipRayTracer(0)
    try:
        tracer.join_frame_group(rank, world, group_name, timeout_ms=60000)
        result["info"] = tracer.comm_info()
        shared = np.memmap(frame_file, dtype=np.uint32, mode="r+", shape=(height, width))
        frame = np.zeros((height, width), dtype=np.uint32) if (mode == "private" and rank == 1) else shared
        first_of_seed = {}
        for f in range(1, frames + 1):
            if f == 2 and mode == "remap":
                del frame, shared
                shared = np.memmap(frame_file, dtype=np.uint32, mode="r+", shape=(height, width))
                frame = shared
            if f == 2 and rank == world - 1 and mode == "dies":
                sys.stdout.write(json.dumps(result) + "\n")
                sys.stdout.flush()
                os._exit(0)  # no destructor runs: the others find its process gone (or, at the latest, their deadline)
            if f == 2 and rank == world - 1 and mode == "leaves":
                tracer.close()
                break
            seed = f + (1 if (mode == "mismatch" and rank == 1 and f == 2) else 0)
            if mode == "many":
                seed = (f - 1) % 3 + 1
            flags = {"preview": capi.RT_HIP_FLAG_PREVIEW, "sm": capi.RT_HIP_FLAG_SM_MATERIALS}.get(mode, 0)
            t0 = time.perf_counter()
            _, rgb, stats = tracer.render(pod, width, height, seed=seed, flags=flags, out=frame, want_rgb=(mode == "float" and rank == 0), stats=(mode != "many" or f <= 3))
            result["ms"].append((time.perf_counter() - t0) * 1e3)
            result["frames_done"] = f
            if stats:
                result["stats"] = stats
            if rank == 0 and mode == "many" and f > 3:
                if not np.array_equal(frame, first_of_seed[seed]):
                    result["error"] = f"frame {f} (seed {seed}) differs from the first frame of that seed in {int((np.asarray(frame) != first_of_seed[seed]).sum())} pixels"
                    break
            elif rank == 0:
                np.save(out_dir / f"frame_{f}.npy", np.asarray(frame))
                first_of_seed[seed] = np.array(frame)
        if mode == "numa" and rank == 0:
            result["page_nodes"] = page_nodes(frame)
            result["host_nodes"] = sorted(int(d[4:]) for d in os.listdir("/sys/devices/system/node") if d.startswith("node") and d[4:].isdigit())
        result["ms"] = result["ms"][-20:]
        if result["frames_done"] == frames and mode != "many":
            result["member_stats"] = [tracer.member_stats(r) for r in range(world)]
            result["phases"] = tracer.phases()
            result["devices"] = [tracer.member_device(r) for r in range(world)]
            # the lean call (stats == NULL), once more with the last seed: same frame
            tracer.render(pod, width, height, seed=frames, out=frame, stats=False)
            if rank == 0:
                np.save(out_dir / "frame_lean.npy", np.asarray(frame))
    except rt_amd.RtHipError as e:
        result["error"] = str(e)
    tracer.close()
    print(json.dumps(result))


if __name__ == "__main__":
    main()
