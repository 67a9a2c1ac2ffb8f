"""The N>1 path rehearsed on the CPU: world_size-2 (and 3) `gloo` process groups run the product's partition,
gather and assemble code (rt_amd/distributed.py); the per-rank stripes are rendered by the ORACLE here, because
there is no GPU in this container — on the GPU the same code moves the HIP kernels' output over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, width, height, stripe, seed, out_path):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import rt_amd
        from oracle import binding as oracle
        from rt_amd import distributed

        scene = rt_amd.Scene.named("basic").set_sampling(3)
        pod = scene.describe(width, height)
        part, _, _ = oracle.render(pod, width, height, seed=seed, partition=(rank, world, stripe), want_rgb=False, threads=1)
        padded = distributed.padded_rows(height, world, stripe)
        assert padded == rt_amd.padded_local_rows(height, world, stripe)
        local = torch.zeros((padded, width), dtype=torch.int32)
        local[: part.shape[0]] = torch.from_numpy(part.view(np.int32))
        gathered = distributed.gather_stripes(local, dst=0)
        if rank == 0:
            frame = distributed.assemble(gathered, width, height, stripe)
            np.save(out_path, frame.numpy().view(np.uint32))
        else:
            assert gathered is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,width,height,stripe", [(2, 64, 37, 8), (3, 40, 50, 4)])
def test_gloo_gather_and_assemble_reproduce_the_whole_frame(tmp_path, world, width, height, stripe):
    import rt_amd
    from oracle import binding as oracle

    seed = 77
    out_path = tmp_path / "frame.npy"
    mp.spawn(_worker, args=(world, _free_port(), width, height, stripe, seed, str(out_path)), nprocs=world, join=True)
    scene = rt_amd.Scene.named("basic").set_sampling(3)
    whole, _, _ = oracle.render(scene.describe(width, height), width, height, seed=seed, want_rgb=False)
    assert np.array_equal(np.load(out_path), whole)


def test_assemble_refuses_device_buffers_without_a_tracer():
    from rt_amd import distributed

    class FakeCuda:
        is_cuda = True
        shape = (2, 8, 8)

    import rt_amd

    with pytest.raises(rt_amd.RtHipError):
        distributed.assemble(FakeCuda(), 8, 16, 8, tracer=None)
