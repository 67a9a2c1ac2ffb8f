"""The multi-GPU frame path end to end on the GPU box: two ranks (sharing this box's single GPU, gloo as the
transport because RCCL refuses two ranks on one device) each render their stripes with the HIP kernels; rank 0
gathers, assembles on the device and must hold exactly the frame the oracle renders in one piece."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, width, height, spp, seed, out_path, in_flight=1, frames=1):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rt_amd
        from rt_amd import distributed

        torch.cuda.set_device(0)
        tracers = [rt_amd.HipRayTracer(device=0) for _ in range(in_flight)]
        scene = rt_amd.Scene.named("dielectric").set_sampling(spp)
        for tracer in tracers:
            tracer.upload(scene.describe(width, height))
        frame = distributed.DistributedFrame(tracers, width, height)
        assert (frame.rank, frame.world) == (rank, world)
        outs = [frame.render(seed=seed + k) for k in range(frames)]  # enqueued back to back, like bench.py's timed loop
        torch.cuda.synchronize()
        if rank == 0:
            np.save(out_path, np.stack([out.cpu().numpy().view(np.uint32) for out in outs]))
        else:
            assert all(out is None for out in outs)
        dist.barrier()
        for tracer in tracers:
            tracer.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_render_gather_and_assemble_the_oracle_frame(tmp_path, world):
    import rt_amd
    from oracle import binding as oracle

    width, height, spp, seed = 150, 101, 20, 31
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = tmp_path / "frame.npy"
    mp.spawn(_worker, args=(world, port, width, height, spp, seed, str(out_path)), nprocs=world, join=True)
    scene = rt_amd.Scene.named("dielectric").set_sampling(spp)
    want, _, _ = oracle.render(scene.describe(width, height), width, height, seed=seed, want_rgb=False)
    assert np.array_equal(np.load(out_path)[0], want)


@pytest.mark.parametrize("world", [1, 2])
def test_two_frames_in_flight_deliver_every_frame_intact(tmp_path, world):
    """bench.py keeps two frames in flight on N > 1 (alternating streams, stripe buffers and contexts): five
    consecutive frames with different seeds must each equal the oracle's frame."""
    import rt_amd
    from oracle import binding as oracle

    width, height, spp, seed, frames = 96, 70, 8, 40, 5
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out_path = tmp_path / "frames.npy"
    mp.spawn(_worker, args=(world, port, width, height, spp, seed, str(out_path), 2, frames), nprocs=world, join=True)
    got = np.load(out_path)
    scene = rt_amd.Scene.named("dielectric").set_sampling(spp)
    for k in range(frames):
        want, _, _ = oracle.render(scene.describe(width, height), width, height, seed=seed + k, want_rgb=False)
        assert np.array_equal(got[k], want), f"frame {k}"
