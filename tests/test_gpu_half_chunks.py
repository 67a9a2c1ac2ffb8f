"""Half-chunk work items (render_queue<.., HALF>): launches that hold only a few 16-sample chunks per lane
of the device are cut into 8-sample items — the second half of a chunk parks its sample values and the wave's fold
continues the first half's partial sum with them, so the chunk sum is the contract's sixteen-sample sequence, bit for bit.
Which form a launch takes is the launch code's business (its size); here both are forced, on every kind of sample count
(ragged first and second halves, one chunk, many chunks), kernel and destination, and compared with the oracle."""
import numpy as np
import pytest

import rt_amd
from oracle import binding as oracle
from rt_amd import capi
from tests.conftest import PLANES_SCENE

pytestmark = pytest.mark.gpu

HALF, WHOLE = capi.RT_HIP_FLAG_FORCE_HALF_CHUNKS, capi.RT_HIP_FLAG_FORCE_WHOLE_CHUNKS


def same_bits(a, b):
    return np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


@pytest.mark.parametrize("spp", [8, 9, 12, 16, 17, 23, 24, 25, 33, 40, 64, 100, 129, 250])
def test_both_forms_give_the_oracles_frame_for_every_kind_of_sample_count(tracer, spp):
    width, height = 72, 44
    pod = rt_amd.Scene.named("basic").set_sampling(spp).describe(width, height)
    want, want_rgb, want_stats = oracle.render(pod, width, height, seed=spp, want_rgb=True)
    for flags in (HALF, WHOLE, 0):
        got, rgb, stats = tracer.render(pod, width, height, seed=spp, flags=flags, want_rgb=True)
        assert np.array_equal(got, want) and same_bits(rgb, want_rgb), (spp, flags)
        assert stats["segments"] == want_stats["segments"]
        # ... and straight into the page-locked back buffer (tiles cut for the host frame)
        frame = np.zeros((height, width), dtype=np.uint32)
        tracer.render(pod, width, height, seed=spp, flags=flags | capi.RT_HIP_FLAG_PERSISTENT_FRAME, out=frame)
        assert np.array_equal(frame, want), (spp, flags)
    tracer.forget_frame()


@pytest.mark.parametrize("scene, extra", [("dielectric", 0), ("planes", 0), ("basic", capi.RT_HIP_FLAG_FORCE_RESIDENT), ("basic", capi.RT_HIP_FLAG_FAST)])
def test_every_small_scene_kernel_has_both_forms(tracer, scene, extra):
    """7 spheres in scalar registers, the LDS-resident kernel (planes; forced), and the tolerance-bound build — whose two
    forms agree to rounding only: with -ffp-contract=fast a sample's value is fused into the running sum where the sum is
    kept in registers, and cannot be where the value is parked first."""
    width, height, spp = 100, 61, 40
    s = rt_amd.Scene.parse(PLANES_SCENE) if scene == "planes" else rt_amd.Scene.named(scene)
    pod = s.set_sampling(spp).describe(width, height)
    half, half_rgb, _ = tracer.render(pod, width, height, seed=5, flags=extra | HALF, want_rgb=True)
    whole, whole_rgb, _ = tracer.render(pod, width, height, seed=5, flags=extra | WHOLE, want_rgb=True)
    if extra & capi.RT_HIP_FLAG_FAST:
        assert np.allclose(half_rgb, whole_rgb, rtol=2e-6, atol=1e-7)
        assert (np.abs(np.asarray(half >> 8 & 0xFF, dtype=np.int64) - np.asarray(whole >> 8 & 0xFF, dtype=np.int64)) <= 1).all()
    else:
        assert np.array_equal(half, whole) and same_bits(half_rgb, whole_rgb)
        want, want_rgb, _ = oracle.render(pod, width, height, seed=5, want_rgb=True)
        assert np.array_equal(half, want) and same_bits(half_rgb, want_rgb)


def test_the_sm_table_keeps_whole_chunks_and_both_flags_at_once_are_refused(tracer):
    width, height, spp = 64, 40, 24
    pod = rt_amd.Scene.named("dielectric").set_sampling(spp).describe(width, height)
    want, _, _ = oracle.render(pod, width, height, seed=2, want_rgb=False, sm_materials=True)
    got, _, _ = tracer.render(pod, width, height, seed=2, flags=capi.RT_HIP_FLAG_SM_MATERIALS | HALF)
    assert np.array_equal(got, want)
    with pytest.raises(rt_amd.RtHipError, match="exclude each other"):
        tracer.render(pod, width, height, seed=2, flags=HALF | WHOLE)


def test_an_eighth_of_config_2_in_half_chunks_is_the_same_eighth(tracer):
    """Full size, where the form matters (1920 x 136 rows at 64 spp: two chunks per lane): the launch code picks half chunks
    for the share by itself; the rows it renders are the rows of the whole frame rendered in whole chunks."""
    import torch

    width, height, spp = 1920, 1080, 64
    pod = rt_amd.Scene.named("basic").set_sampling(spp).describe(width, height)
    whole, _, _ = tracer.render(pod, width, height, seed=1, flags=WHOLE)
    tracer.upload(pod)
    rows = rt_amd.padded_local_rows(height, 8)
    for rank in (0, 3, 7):
        share = torch.zeros((rows, width), dtype=torch.int32, device="cuda:0")
        for flags in (0, HALF):
            share.zero_()
            tracer.render_device(width, height, share.data_ptr(), seed=1, flags=flags, partition=(rank, 8, 8), stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = share.cpu().numpy().view(np.uint32)
            local = 0
            for stripe in range(rank, (height + 7) // 8, 8):
                y0 = stripe * 8
                n = min(8, height - y0)
                assert np.array_equal(got[local : local + n], whole[y0 : y0 + n]), (rank, flags, stripe)
                local += n


@pytest.mark.parametrize("spp", [2, 5, 16, 17, 40, 70])
@pytest.mark.parametrize("kernel", [capi.RT_HIP_FLAG_FORCE_STREAMED, capi.RT_HIP_FLAG_FORCE_TILED], ids=["streamed", "tiled"])
def test_big_scene_kernels_hand_out_runs_of_samples_and_add_them_up_as_the_contract_says(tracer, kernel, spp):
    """The rolling big-scene kernels with items smaller than a chunk: every sample's value travels through HBM on its own and
    the lane that brings a pixel's last item adds them up — sixteen in sample order per chunk, chunks in chunk order.  The
    launch code's own choice for a frame this small is one or two samples per item; forced: eight; and whole chunks."""
    from tests.test_gpu_parity import _sphere_field

    width, height = 40, 22
    rng = np.random.default_rng(spp)
    spheres, materials, camera = _sphere_field(rng, 1500)
    ivp = camera.describe(width, height).inverse_view_projection[:]
    pod = rt_amd.scene_from_arrays(spheres, [], materials, samples_per_pixel=spp, max_bounces=6, inverse_view_projection=ivp)
    want, want_rgb, want_stats = oracle.render(pod, width, height, seed=7)
    for flags in (0, HALF, WHOLE):
        for _ in range(2):  # (the second launch finds the pixels' arrival counters as the first one left them: zero)
            got, rgb, stats = tracer.render(pod, width, height, seed=7, flags=kernel | flags, want_rgb=True)
            assert stats["kernel"] in ("streamed", "tiled")
            assert np.array_equal(got, want) and same_bits(rgb, want_rgb), (spp, flags)
            assert stats["segments"] == want_stats["segments"]
