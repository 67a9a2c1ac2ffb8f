"""Two ranks of a frame group on one device that claim GPUs on different sockets (RT_HIP_NUMA_NODE = rank % 2), with
RT_HIP_DEBUG_FRAME=1: what move_pages says about the stripes."""
import os, subprocess, sys, uuid
import numpy as np
sys.path.insert(0, ".")
tag = uuid.uuid4().hex[:10]
frame = f"/dev/shm/rt_hip_dbg_frame_{tag}"
np.zeros((1080, 1920), dtype=np.uint32).tofile(frame)
os.makedirs("gpurun_out/dbg", exist_ok=True)
env = dict(os.environ, RT_HIP_DEBUG_FRAME="1")
procs = [subprocess.Popen([sys.executable, "tests/frame_group_worker.py", str(r), "2", f"/rt_hip_dbg_{tag}", frame, "1920", "1080", "2", "basic", "2", "numa", "gpurun_out/dbg"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(2)]
for r, p in enumerate(procs):
    out, err = p.communicate(timeout=200)
    print(f"== rank {r} rc {p.returncode}")
    print("\n".join(l for l in err.splitlines() if "rt_hip" in l))
    import json
    j = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    if r == 0:
        nodes = j["page_nodes"]
        print("error", j["error"], "host nodes", j["host_nodes"], "first 45 pages", nodes[:45], "counts", {n: nodes.count(n) for n in set(nodes)})
os.unlink(frame)
