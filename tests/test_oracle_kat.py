"""Known-answer tests that pin the CPU oracle.

The reference ships no tests or golden vectors ("parity unpinned"), so every expectation here is
derived from the reference's SOURCE and computed independently of the oracle's code (closed forms, float64 numpy).
"""
import math
from pathlib import Path

import numpy as np
import pytest

import rt_amd
from oracle import binding as oracle
from tests.conftest import unpack

ROOT = Path(__file__).resolve().parent.parent

BLACK = 0x000000FF


# ---- colour pack: reference src/colour.hpp:101-106 (clamp to [0,1], x 255.99999f, truncate, RGBA8888) --------------
@pytest.mark.parametrize(
    "rgb,expected",
    [
        ((1.0, 0.0, 1.0), 0xFF00FFFF),
        ((0.25, 0.25, 0.25), 0x3F3F3FFF),  # 0.25 * 255.99999 = 63.99 -> 63
        ((0.5, 0.5, 0.5), 0x7F7F7FFF),
        ((2.0, -1.0, 0.999999), 0xFF00FFFF),
        ((0.0, 1.0 / 255.99999, 0.00390626), 0x000001FF),
        ((float("nan"), 0.0, 0.0), 0x000000FF),
    ],
)
def test_pack(rgb, expected):
    assert oracle.pack(*rgb) == expected


def test_pack_matches_closed_form_everywhere():
    xs = np.linspace(-0.5, 1.5, 4001, dtype=np.float32)
    for x in xs[::7]:
        want = int(np.float32(min(max(x, np.float32(0)), np.float32(1))) * np.float32(255.99999))
        assert (oracle.pack(float(x), 0.0, 0.0) >> 24) == want


# ---- sky: reference mg_ray_tracer.cpp:164 ------------------------------------------------------------------------
@pytest.mark.parametrize("dir_y", [-1.0, -0.3, 0.0, 0.123, 1.0])
def test_sky(dir_y):
    t = 0.5 * (dir_y + 1.0)
    want = np.array([1.0 + (0.5 - 1.0) * t, 1.0 + (0.7 - 1.0) * t, 1.0])
    assert np.allclose(oracle.sky(dir_y), want, rtol=0, atol=2e-7)


# ---- camera: reference src/camera.hpp:42-48,122-137 with the conventions of SURVEY.md §8c(4,5) -------------------
def analytic_primary_direction(px, py, width, height, eye_dir=(0.0, 0.0, -1.0)):
    """Direction through screen position (px, py) for a camera looking down eye_dir with +Y up, vfov pi/4."""
    f = np.array(eye_dir, dtype=np.float64)
    f /= np.linalg.norm(f)
    r = np.cross(f, [0.0, 1.0, 0.0])
    r /= np.linalg.norm(r)
    u = np.cross(r, f)
    tan_half = math.tan(math.pi / 8)
    ndc_x = 2.0 * px / width - 1.0
    ndc_y = 1.0 - 2.0 * py / height
    d = f + r * (ndc_x * tan_half * width / height) + u * (ndc_y * tan_half)
    return d / np.linalg.norm(d)


def pixel_and_jitter(p, size):
    """a frame position as (pixel, jitter numerator): p = pixel + k * 2^-24 (contract v4 hands the jitter over as its numerator)"""
    pixel = min(int(p), size - 1)
    return pixel, float((p - pixel) * 2.0**24)


def oracle_primary_ray_at(pod, width, height, px, py, **kw):
    (x, ka), (y, kb) = pixel_and_jitter(px, width), pixel_and_jitter(py, height)
    return oracle.primary_ray(pod, width, height, x, y, ka, kb, **kw)


@pytest.mark.parametrize("size", [(256, 256), (1920, 1080), (64, 36), (7, 3)])
def test_primary_rays_match_pinhole_model(size):
    width, height = size
    scene = rt_amd.Scene.named("basic")
    pod = scene.describe(width, height)
    eye = np.array([0.0, 1.0, 3.0])
    for px, py in [(width / 2, height / 2), (0.0, 0.0), (width, height), (0.5, height - 0.5), (width * 0.25, height * 0.9)]:
        o, d, form = oracle_primary_ray_at(pod, width, height, px, py, want_form=True)
        assert form == "pinhole"  # rt's camera, axis-aligned: the per-pixel base form
        want = analytic_primary_direction(px, py, width, height)
        # float32 inverse view-projection with near 0.01 / far 1000: ~1e-5 rad of direction error
        assert np.allclose(d, want, atol=5e-5), (px, py, d, want)
        # origin lies on the same eye ray, close to the eye (near plane 0.01)
        off = o.astype(np.float64) - eye
        assert 0.005 < np.linalg.norm(off) < 0.03
        assert np.allclose(off / np.linalg.norm(off), want, atol=2e-3)


def test_primary_ray_for_rotated_camera():
    scene = rt_amd.Scene.named("basic").set_camera((1.0, 2.0, 3.0), (0.3, -0.2, -1.0))
    pod = scene.describe(320, 200)
    for px, py in [(160.0, 100.0), (10.5, 20.5), (300.0, 190.0)]:
        o, d = oracle_primary_ray_at(pod, 320, 200, px, py)
        want = analytic_primary_direction(px, py, 320, 200, (0.3, -0.2, -1.0))
        assert np.allclose(d, want, atol=5e-5)
        off = o.astype(np.float64) - np.array([1.0, 2.0, 3.0])  # the near point: on the same eye ray, close to the eye
        assert 0.005 < np.linalg.norm(off) < 0.03 and np.allclose(off / np.linalg.norm(off), want, atol=2e-3)


# ---- visibility mask: max_bounces = 1 -> hit pixels are exactly black, miss pixels are the sky --------------------
def analytic_visibility(width, height, eye, spheres):
    """float64 closest-hit of the pixel-centre rays: (hit mask, |distance to silhouette| proxy, dir_y)."""
    ys, xs = np.mgrid[0:height, 0:width]
    px, py = xs + 0.5, ys + 0.5
    tan_half = math.tan(math.pi / 8)
    dx = (2.0 * px / width - 1.0) * tan_half * width / height
    dy = (1.0 - 2.0 * py / height) * tan_half
    d = np.stack([dx, dy, -np.ones_like(dx)], axis=-1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    hit = np.zeros((height, width), dtype=bool)
    margin = np.full((height, width), np.inf)
    for cx, cy, cz, r in spheres:
        e = np.array([cx, cy, cz]) - eye
        a = d @ e
        disc = r * r - (e @ e - a * a)
        hit |= (disc >= 0) & (a - np.sqrt(np.maximum(disc, 0)) >= 0.001)
        margin = np.minimum(margin, np.abs(disc) / (2 * r))  # ~ distance between ray and sphere surface
    return hit, margin, d[..., 1]


@pytest.mark.parametrize("name,eye", [("basic", (0.0, 1.0, 3.0)), ("dielectric", (0.0, 1.0, 7.0))])
def test_visibility_mask_against_analytic_silhouettes(name, eye):
    width, height = 160, 90
    scene = rt_amd.Scene.named(name).set_sampling(samples_per_pixel=1, max_bounces=1)
    pod = scene.describe(width, height)
    rgba, rgb, stats = oracle.render(pod, width, height, seed=3)
    spheres = [(pod.sphere_center_x[i], pod.sphere_center_y[i], pod.sphere_center_z[i], pod.sphere_radius[i]) for i in range(pod.n_spheres)]
    hit, margin, dir_y = analytic_visibility(width, height, np.array(eye), spheres)
    safe = margin > 2e-3  # stay clear of silhouette edges, where float32 may flip hit/miss
    assert safe.mean() > 0.9
    # hit pixels: attenuation * trace(..., 0) = attenuation * 0 -> black (mg_ray_tracer.cpp:157-158,171)
    assert np.all(rgba[hit & safe] == BLACK)
    # miss pixels: pack(sqrt(sky(dir.y))) (mg_ray_tracer.cpp:163-164,196-200)
    t = 0.5 * (dir_y + 1.0)
    sky = np.stack([1.0 - 0.5 * t, 1.0 - 0.3 * t, np.ones_like(t)], axis=-1)
    want = np.floor(np.sqrt(sky) * 255.99999).astype(np.int64)
    got = unpack(rgba)[..., :3].astype(np.int64)
    miss = ~hit & safe
    assert miss.sum() > 1000
    assert np.abs(got[miss] - want[miss]).max() <= 1  # float32 vs float64 rounding at a quantisation step
    assert (got[miss] != want[miss]).mean() < 0.01
    assert np.all(unpack(rgba)[..., 3] == 255)
    assert stats["segments"] == width * height  # exactly one closest-hit query per sample


# ---- closest hit: formulas of SURVEY.md §8c(1,2), scan order of mg_ray_tracer.cpp:36-102 -------------------------
MAT = [(0, 1, 1, 1, 1, 0.5, 0.5)]


def test_tie_break_lowest_sphere_index_wins():
    pod = rt_amd.scene_from_arrays(spheres=[(0, 0, -5, 1, 0), (0, 0, -5, 1, 0), (0, 0, -9, 1, 0)], materials=MAT)
    dist, kind, index, normal = oracle.closest_hit(pod, [(0, 0, 0)], [(0, 0, -1)])
    assert kind[0] == 1 and index[0] == 0 and dist[0] == pytest.approx(4.0, abs=1e-6)
    assert np.allclose(normal[0], (0, 0, 1), atol=1e-6)


def test_tie_break_sphere_beats_plane_at_equal_distance():
    # plane z = -4 (normal +z, d = 4) touches the near pole of the sphere: both are hit at t = 4 exactly
    pod = rt_amd.scene_from_arrays(spheres=[(0, 0, -5, 1, 0)], planes=[(0, 0, 1, 4, 0)], materials=MAT)
    dist, kind, index, _ = oracle.closest_hit(pod, [(0, 0, 0)], [(0, 0, -1)])
    assert dist[0] == 4.0 and kind[0] == 1
    # ...and the plane wins as soon as it is strictly closer
    pod = rt_amd.scene_from_arrays(spheres=[(0, 0, -5, 1, 0)], planes=[(0, 0, 1, 3.5, 0)], materials=MAT)
    dist, kind, index, normal = oracle.closest_hit(pod, [(0, 0, 0)], [(0, 0, -1)])
    assert dist[0] == 3.5 and kind[0] == 2 and np.array_equal(normal[0], (0, 0, 1))


def test_hits_behind_or_too_close_are_rejected():
    pod = rt_amd.scene_from_arrays(spheres=[(0, 0, 5, 1, 0)], planes=[(0, 1, 0, 1, 0)], materials=MAT)
    # sphere behind the ray, plane y = -1 behind an upward ray
    dist, kind, _, _ = oracle.closest_hit(pod, [(0, 0, 0)], [(0, 0.6, -0.8)])
    assert kind[0] == 0 and dist[0] < 0
    # min_hit_dist = 0.001 (mg_ray_tracer.cpp:20): origin 0.0005 above the plane, looking down -> rejected
    dist, kind, _, _ = oracle.closest_hit(pod, [(0, -0.9995, 0)], [(0, -1, 0)])
    assert kind[0] == 0
    dist, kind, _, _ = oracle.closest_hit(pod, [(0, -0.99, 0)], [(0, -1, 0)])
    assert kind[0] == 2 and dist[0] == pytest.approx(0.01, rel=1e-4)


def test_ray_from_inside_a_sphere_hits_the_far_side():
    pod = rt_amd.scene_from_arrays(spheres=[(0, 0, 0, 2, 0)], materials=MAT)
    dist, kind, _, normal = oracle.closest_hit(pod, [(0.5, 0, 0)], [(1, 0, 0)])
    assert kind[0] == 1 and dist[0] == pytest.approx(1.5, abs=1e-6)
    assert np.allclose(normal[0], (1, 0, 0), atol=1e-6)  # outward normal, not flipped toward the ray


def test_parallel_ray_misses_plane():
    pod = rt_amd.scene_from_arrays(planes=[(0, 1, 0, 0, 0)], materials=MAT)
    _, kind, _, _ = oracle.closest_hit(pod, [(0, 1, 0)], [(1, 0, 0)])
    assert kind[0] == 0


def test_closest_hit_against_float64_brute_force():
    rng = np.random.default_rng(5)
    n_s = 40
    spheres = np.column_stack([rng.uniform(-5, 5, n_s), rng.uniform(-5, 5, n_s), rng.uniform(-15, -5, n_s), rng.uniform(0.2, 1.0, n_s), np.zeros(n_s)])
    pod = rt_amd.scene_from_arrays(spheres=spheres, materials=MAT)
    origins = rng.uniform(-1, 1, (3000, 3))
    dirs = rng.normal(size=(3000, 3)) * [0.4, 0.4, 0.1] + [0, 0, -1]
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dist, kind, index, _ = oracle.closest_hit(pod, origins, dirs)
    o32, d32 = origins.astype(np.float32).astype(np.float64), dirs.astype(np.float32).astype(np.float64)
    s32 = spheres.astype(np.float32).astype(np.float64)
    e = s32[None, :, :3] - o32[:, None, :]
    a = np.einsum("nsk,nk->ns", e, d32)
    disc = s32[None, :, 3] ** 2 - (np.einsum("nsk,nsk->ns", e, e) - a * a)
    t = np.where(disc >= 0, a - np.sqrt(np.maximum(disc, 0)), np.inf)
    t[t < 0.001] = np.inf
    best = t.argmin(axis=1)
    best_t = t.min(axis=1)
    clear = np.abs(disc).min(axis=1) > 1e-3  # rays grazing a sphere may legitimately differ in float32
    hit = np.isfinite(best_t)
    assert np.array_equal(kind[clear] == 1, hit[clear])
    sel = clear & hit
    assert sel.sum() > 300
    assert np.array_equal(index[sel], best[sel])
    assert np.allclose(dist[sel], best_t[sel], rtol=1e-4)


# ---- random streams ---------------------------------------------------------------------------------------------------
def test_random_stream_is_uniform_and_deterministic():
    u = oracle.random(seed=1, pixel=12345, sample=7, n=200000)
    assert u.dtype == np.float32 and u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.003 and abs(u.var() - 1 / 12) < 0.002
    counts, _ = np.histogram(u, bins=64, range=(0, 1))
    chi2 = ((counts - len(u) / 64) ** 2 / (len(u) / 64)).sum()
    assert chi2 < 120  # 63 dof; p ~ 1e-5
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 0.01
    assert np.array_equal(u[:100], oracle.random(1, 12345, 7, 100))


def test_random_streams_of_neighbouring_pixels_and_samples_are_unrelated():
    base = oracle.random(1, 1000, 3, 4096)
    for other in (oracle.random(1, 1001, 3, 4096), oracle.random(1, 1000, 4, 4096), oracle.random(2, 1000, 3, 4096), oracle.random(1 << 32, 1000, 3, 4096)):
        assert not np.array_equal(base, other)
        assert abs(np.corrcoef(base, other)[0, 1]) < 0.06


def test_first_draws_across_pixels_are_uniform():
    firsts = np.array([oracle.random(9, p, 1, 1)[0] for p in range(20000)])
    assert abs(firsts.mean() - 0.5) < 0.01
    counts, _ = np.histogram(firsts, bins=16, range=(0, 1))
    assert ((counts - 1250) ** 2 / 1250).sum() < 50


# ---- contract v2: per-pixel keyed streams --------------------------------------------------------------------------------
M32 = np.uint64(0xFFFFFFFF)


def _hash32(x):
    """lowbias32 on uint64 arrays holding 32-bit words (an implementation independent of the oracle's C++)."""
    x = x & M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & M32
    x ^= x >> np.uint64(16)
    return x


def _frame_keys(seed):
    m = (1 << 64) - 1
    z = (seed + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    return z & 0xFFFFFFFF, z >> 32


def _stream_start(seed, pixels, samples):
    """(function key, stride, counter before the first draw) per contract v2, in numpy."""
    fa, fb = _frame_keys(seed)
    key = _hash32(pixels.astype(np.uint64) ^ np.uint64(fa))
    stride = _hash32(key ^ np.uint64(fb)) | np.uint64(1)
    counter = (stride * (samples.astype(np.uint64) << np.uint64(12))) & M32
    return key, stride, counter


# contract v4: one generator step yields three draws — the mixed word times M2 * (1, A, A^2), top 24 bits of each product
_M2 = 0x846CA68B
_LATTICE = 0xADB4A92D  # Steele & Vigna's 32-bit LCG multiplier
_STEP_MULTIPLIERS = [np.uint64(_M2), np.uint64(_M2 * _LATTICE & 0xFFFFFFFF), np.uint64(_M2 * _LATTICE * _LATTICE & 0xFFFFFFFF)]


def _draws(key, stride, counter, n):
    """first n numbers of the streams starting at (key, stride, counter), steps flattened: float32[len(key), n]"""
    out = np.empty((len(key), n), dtype=np.float32)
    c = counter.copy()
    for j in range(0, n, 3):
        c = (c + stride) & M32
        x = c ^ (c >> np.uint64(16))
        x = (x * np.uint64(0x7FEB352D) + key) & M32
        x ^= x >> np.uint64(15)
        for word, multiplier in enumerate(_STEP_MULTIPLIERS[: n - j]):
            out[:, j + word] = (((x * multiplier) & M32) >> np.uint64(8)).astype(np.float32) * np.float32(2.0**-24)
    return out


def test_stream_contract_matches_an_independent_numpy_implementation():
    rng = np.random.default_rng(3)
    pixels = rng.integers(0, 3840 * 2160, 200, dtype=np.uint32)
    samples = rng.integers(0, 1000, 200, dtype=np.uint32)
    for seed in (0, 1, 0x0123456789ABCDEF, (1 << 64) - 1):
        key, stride, counter = _stream_start(seed, pixels, samples)
        got = oracle.stream_keys(seed, pixels, samples)
        for mine, theirs in zip((key, stride, counter), got):
            assert np.array_equal(theirs, mine.astype(np.uint32))
        want = _draws(key, stride, counter, 12)
        for i in range(0, 200, 17):
            assert np.array_equal(oracle.random(seed, int(pixels[i]), int(samples[i]), 12), want[i])


def test_no_two_samples_of_a_1080p_frame_share_or_overlap_a_stream():
    """Round-1 finding: under contract v1 a 1920x1080x16 frame had 120 201 (pixel, sample) pairs
    with IDENTICAL streams and 1.6 M sample windows starting within 7 draws of another.  Under v2 a stream is the
    triple (function key, stride, counter): the function key is a bijection of the pixel index, so two pixels never
    draw through the same function, and the samples of one pixel occupy disjoint windows of its own progression."""
    width, height, spp, seed = 1920, 1080, 16, 1
    pixels = np.repeat(np.arange(width * height, dtype=np.uint32), spp)
    samples = np.tile(np.arange(spp, dtype=np.uint32), width * height)
    key, stride, counter = oracle.stream_keys(seed, pixels, samples)
    # 1. the function key is injective in the pixel index; the stride is odd
    per_pixel = key[::spp]
    assert np.all(key.reshape(-1, spp) == per_pixel[:, None])
    assert len(np.unique(per_pixel)) == width * height
    assert np.all(stride & 1 == 1)
    # 2. no two (pixel, sample) pairs start in the same state — the measurement that found 120 201 duplicates in v1
    state = (key.astype(np.uint64) << np.uint64(32)) | counter.astype(np.uint64)
    state.sort()
    assert np.all(state[1:] != state[:-1])
    # 3. shifted overlaps: a stream can only run into another one that shares its function key, i.e. its pixel; within
    #    a pixel, sample s occupies positions [4096 s + 1, 4096 (s + 1)] of the pixel's progression counter = stride * m
    #    (the stride is odd, so m -> counter is a bijection of 32-bit words: division is exact)
    inverse = np.array([pow(int(v), -1, 1 << 32) for v in stride[::spp][:5000]], dtype=np.uint64)
    position = (counter.reshape(-1, spp)[:5000].astype(np.uint64) * inverse[:, None]) & M32
    assert np.array_equal(position, np.broadcast_to(np.arange(spp, dtype=np.uint64) * np.uint64(4096), position.shape))
    # 4. runs of common counters between two pixels need equal strides.  A few pixel pairs of a frame do share a stride
    #    (32-bit hashes of 2 M pixels); their function keys must then be unrelated — at least a few bits apart
    strides = stride[::spp]
    order = np.argsort(strides, kind="stable")
    same = np.nonzero(strides[order][1:] == strides[order][:-1])[0]
    assert len(same) < 3000  # ~ (2 M)^2 / 2 / 2^31 = 1000 expected
    distance = [bin(int(per_pixel[order[i]]) ^ int(per_pixel[order[i + 1]])).count("1") for i in same]
    assert min(distance, default=32) >= 4


def test_pixels_reading_the_same_counters_get_unrelated_numbers():
    """The 2^32 counter values are shared by all pixels.  Two pixels that do walk the same counters (equal strides: about a
    thousand pairs in a 1080p frame) must still see unrelated numbers, because their functions differ."""
    n = 4096
    keys = _hash32(np.arange(2000, dtype=np.uint64) ^ np.uint64(0xDEADBEEF))  # function keys of 2000 neighbouring pixels
    stride = np.full(2000, 0x9E3779B9, dtype=np.uint64)
    draws = _draws(keys, stride, np.zeros(2000, dtype=np.uint64), n)  # every row walks the SAME counters
    assert abs(draws.mean() - 0.5) < 0.002
    a, b = draws[::2], draws[1::2]
    corr = [np.corrcoef(x, y)[0, 1] for x, y in zip(a, b)]
    assert np.abs(corr).max() < 5.5 / np.sqrt(n) and abs(np.mean(corr)) < 3.0 / np.sqrt(n * len(corr))
    # a key that merely rotated the output by a constant would pass the correlation test: the difference must be uniform
    for row in ((a - b) % 1.0)[:200]:
        counts, _ = np.histogram(row, bins=16, range=(0, 1))
        assert ((counts - n / 16) ** 2 / (n / 16)).sum() < 70  # 15 dof, p ~ 1e-8


def test_neighbouring_pixels_have_unrelated_streams():
    rng = np.random.default_rng(5)
    n = 4096
    pixels = np.arange(1000, 1000 + 64, dtype=np.uint32)
    key, stride, counter = _stream_start(1, pixels, np.full(64, 3, dtype=np.uint32))
    draws = _draws(key, stride, counter, n)
    corr = np.corrcoef(draws)
    np.fill_diagonal(corr, 0.0)
    assert np.abs(corr).max() < 5.5 / np.sqrt(n)
    # lagged: the jitter draws of one pixel against the scatter draws of the next
    for lag in (1, 2, 3):
        assert abs(np.corrcoef(draws[:-1, lag:].ravel(), draws[1:, :-lag].ravel())[0, 1]) < 4.0 / np.sqrt(63 * (n - lag))


# ---- contract v4: one generator step per random<T>() call ---------------------------------------------------------------
def test_the_draws_of_a_step_fill_the_square_and_the_cube_evenly():
    """The three draws of a step share one mixed word (they are its multiples by M2 (1, A, A^2)): each must still be uniform, and
    so must the pairs and the triple — jitter positions and unit-cube points — at the resolution a frame can see."""
    steps = oracle.random(1, 424242, 5, 3 * 120000).reshape(-1, 3).astype(np.float64)
    across = np.array([oracle.random(9, p, 2, 3) for p in range(40000)], dtype=np.float64)  # the first step of 40 000 pixels
    for points in (steps, across):
        n = len(points)
        for i, j in ((0, 1), (0, 2), (1, 2)):
            counts, _, _ = np.histogram2d(points[:, i], points[:, j], bins=16, range=((0, 1), (0, 1)))
            assert ((counts - n / 256) ** 2 / (n / 256)).sum() < 380  # 255 dof, p ~ 1e-6
            assert abs(np.corrcoef(points[:, i], points[:, j])[0, 1]) < 5.0 / np.sqrt(n)
        cells = (np.floor(points * 8).astype(int) * np.array([64, 8, 1])).sum(axis=1)
        counts = np.bincount(cells, minlength=512)
        assert ((counts - n / 512) ** 2 / (n / 512)).sum() < 680  # 511 dof, p ~ 1e-6
    # the unit vectors made of them (random_unit_vector, random.hpp:57-66) cover the positive octant like independent draws do
    unit = steps / np.linalg.norm(steps, axis=1, keepdims=True)
    independent = np.random.default_rng(7).random((len(steps), 3))
    independent /= np.linalg.norm(independent, axis=1, keepdims=True)
    assert np.allclose(unit.mean(axis=0), independent.mean(axis=0), atol=4e-3)
    assert np.allclose(np.cov(unit.T), np.cov(independent.T), atol=2e-3)


def test_the_lattice_of_a_step_is_a_good_one():
    """The triple of a step is a point of the lattice (1, A, A^2) / 2^32; tools/rng_lattice.py has the spectral test."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("rng_lattice", ROOT / "tools" / "rng_lattice.py")
    lattice = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lattice)
    assert lattice.IN_USE == _LATTICE
    two, three = lattice.figures_of_merit(_LATTICE)
    assert two > 0.95 and three > 0.9
    assert lattice.figures_of_merit(_LATTICE * _LATTICE % (1 << 32))[0] > 0.85  # the first and third draw as a pair
    assert lattice.figures_of_merit(65539)[1] < 0.1  # (the test does tell: RANDU's planes)


def test_one_step_serves_one_call():
    """random<vec2>() for the jitter, random<vec3>() per unit vector, random<float>() for sm's reflect-or-refract choice: the frame
    of a scene with a metal, a lambert and a refracting sphere consumes exactly the steps the numpy restatement
    (tests/test_independent_path_tracer.py) consumes — checked there pixel by pixel; here: the FIRST numbers of a sample's
    stream are its jitter, as numerators of 2^-24."""
    scene = rt_amd.Scene.named("basic").set_sampling(2, 1)  # one bounce: sample 1 consumes its jitter step and nothing else
    pod = scene.describe(40, 30)
    ka, kb = (float(v) * 2.0**24 for v in oracle.random(1, 17 * 40 + 11, 1, 2))
    o, d = oracle.primary_ray(pod, 40, 30, 11, 17, ka, kb)
    assert 0 <= ka < 2**24 and ka == int(ka) and 0 <= kb < 2**24 and kb == int(kb)
    # the frame's pixel (11, 17) is the mean of the centre sample and of that jittered one: both miss or hit together with the rays
    _, rgb, _ = oracle.render(pod, 40, 30, seed=1)
    centre_o, centre_d = oracle.primary_ray(pod, 40, 30, 11, 17)
    hits = [oracle.closest_hit(pod, np.array([a]), np.array([b]))[1][0] != 0 for a, b in ((centre_o, centre_d), (o, d))]
    want = np.mean([np.zeros(3) if hit else oracle.sky(float(direction[1])) for hit, direction in zip(hits, (centre_d, d))], axis=0)
    assert np.allclose(rgb[17, 11], want, rtol=1e-6)


def test_the_three_camera_forms_agree_where_they_overlap():
    """Primary rays come in three forms (oracle/cpu_ref.cpp make_frame): `pinhole` (rt's camera while axis-aligned: w constant over
    the frame), `eye` (any perspective matrix: a tilted camera's float inverse has rounding noise in its w row — every frame of an
    interactive session), `general` (no finite eye: an orthographic frustum).  Which one a matrix gets is decided from the matrix;
    where two apply they must give the same rays to rounding."""
    samples = ((0, 0, 0.0, 0.0), (160, 100, 2.0**23, 2.0**23), (319, 199, 2.0**24 - 1, 1.0), (7, 190, 123456.0, 2.0**24 - 1))
    scene = rt_amd.Scene.named("basic")
    pod = scene.describe(320, 200)
    pinhole = [oracle.primary_ray(pod, 320, 200, *sample, want_form=True) for sample in samples]
    assert all(form == "pinhole" for _, _, form in pinhole)
    # the same matrix with a w row that says "x matters" by one part in 10^12: no pinhole any more, still a perspective matrix
    m = list(pod.inverse_view_projection)
    m[12] = 1.0e-12 * m[15]
    for i, v in enumerate(m):
        pod.inverse_view_projection[i] = v
    eye = [oracle.primary_ray(pod, 320, 200, *sample, want_form=True) for sample in samples]
    assert all(form == "eye" for _, _, form in eye)
    for (o1, d1, _), (o2, d2, _) in zip(pinhole, eye):
        assert np.allclose(d1, d2, atol=2e-7) and np.allclose(o1, o2, atol=3e-7)
    # a camera that really is not axis-aligned: the eye form against screen_to_world in binary64 (camera.hpp:42-48), near and far
    tilted = rt_amd.Scene.named("basic").set_camera((1.0, 2.0, 3.0), (0.3, -0.2, -1.0)).describe(320, 200)
    big = np.array(list(tilted.inverse_view_projection), dtype=np.float64).reshape(4, 4)
    assert big[3, 0] != 0.0 or big[3, 1] != 0.0  # the rounding noise that keeps it from being a pinhole
    for x, y, ka, kb in samples:
        o, d, form = oracle.primary_ray(tilted, 320, 200, x, y, ka, kb, want_form=True)
        assert form == "eye"
        px, py = x + ka * 2.0**-24, y + kb * 2.0**-24
        ends = []
        for depth in (0.0, 1.0):
            v = big @ np.array([2.0 * px / 320 - 1.0, -2.0 * py / 200 + 1.0, depth, 1.0])
            ends.append(v[:3] / v[3])
        want = (ends[1] - ends[0]) / np.linalg.norm(ends[1] - ends[0])
        assert np.allclose(d, want, atol=2e-7), (d, want)  # (the two-division float32 form of round 4 was good to 1e-5 here)
        assert np.allclose(o, ends[0], atol=1e-6)
    # an orthographic frustum has no eye: the homogeneous form — parallel rays from different near points
    ortho = rt_amd.Scene.named("basic").describe(320, 200)
    for i, v in enumerate([2.0, 0, 0, 0, 0, 1.25, 0, 1.0, 0, 0, -10.0, 3.0, 0, 0, 0, 1.0]):
        ortho.inverse_view_projection[i] = v
    (o1, d1, f1), (o2, d2, f2) = (oracle.primary_ray(ortho, 320, 200, x, y, want_form=True) for x, y in ((10, 20), (300, 180)))
    assert f1 == "general" and f2 == "general"
    assert np.array_equal(d1, d2) and np.allclose(d1, (0, 0, -1)) and not np.allclose(o1, o2)


def test_seeds_that_differ_only_in_the_high_half_give_different_frames():
    u = [oracle.random(seed, 5, 1, 64) for seed in (7, 7 | (1 << 32), 7 | (1 << 63))]
    assert not np.array_equal(u[0], u[1]) and not np.array_equal(u[0], u[2]) and not np.array_equal(u[1], u[2])


# ---- whole-frame behaviour -----------------------------------------------------------------------------------------------
def test_sample_zero_goes_through_the_pixel_centre_and_is_seed_independent():
    # spp = 1, max_bounces = 1: no random draw is consumed at all (mg_ray_tracer.cpp:189)
    scene = rt_amd.Scene.named("basic").set_sampling(1, 1)
    pod = scene.describe(64, 36)
    a, _, _ = oracle.render(pod, 64, 36, seed=1)
    b, _, _ = oracle.render(pod, 64, 36, seed=999)
    assert np.array_equal(a, b)


def test_empty_scene_is_pure_sky():
    pod = rt_amd.scene_from_arrays(samples_per_pixel=3, max_bounces=4, inverse_view_projection=rt_amd.Scene.named("basic").describe(32, 16).inverse_view_projection[:])
    rgba, rgb, stats = oracle.render(pod, 32, 16, seed=5)
    assert stats["segments"] == 32 * 16 * 3
    assert np.all(unpack(rgba)[..., 2] == 255)  # sky blue channel is exactly 1
    assert np.all(rgb[..., 2] == 1.0)


def test_iterative_and_recursive_trace_agree_to_rounding():
    scene = rt_amd.Scene.named("dielectric").set_sampling(16)
    pod = scene.describe(96, 54)
    it_rgba, it_rgb, it_stats = oracle.render(pod, 96, 54, seed=11, trace_order=oracle.TRACE_ITERATIVE)
    re_rgba, re_rgb, re_stats = oracle.render(pod, 96, 54, seed=11, trace_order=oracle.TRACE_RECURSIVE)
    assert it_stats["segments"] == re_stats["segments"]  # same paths, only the product association differs
    assert np.allclose(it_rgb, re_rgb, rtol=2e-6, atol=1e-7)
    diff = np.abs(unpack(it_rgba).astype(int) - unpack(re_rgba).astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 0.002


def test_partition_invariance_of_the_oracle():
    scene = rt_amd.Scene.named("basic").set_sampling(3)
    width, height = 50, 37
    pod = scene.describe(width, height)
    full, full_rgb, full_stats = oracle.render(pod, width, height, seed=21)
    from rt_amd import distributed

    for world, stripe in [(2, 8), (3, 4), (8, 8), (5, 1)]:
        table = distributed.local_row_table(height, world, stripe)
        segments = 0
        for rank in range(world):
            part, part_rgb, st = oracle.render(pod, width, height, seed=21, partition=(rank, world, stripe))
            segments += st["segments"]
            rows = np.nonzero(table[:, 0] == rank)[0]
            assert np.array_equal(part, full[rows])
            assert np.array_equal(part_rgb, full_rgb[rows])
        assert segments == full_stats["segments"]


def test_counter_streams_and_mt19937_agree_statistically():
    """The reproducible counter streams must not change the statistics of the image: compare with the
    reference-faithful thread_local std::mt19937 model (src/random.cpp:9-26) at the same spp."""
    scene = rt_amd.Scene.named("basic").set_sampling(64)
    width, height = 96, 54
    pod = scene.describe(width, height)
    _, counter_rgb, counter_stats = oracle.render(pod, width, height, seed=4)
    _, mt_rgb, mt_stats = oracle.render_mt19937(pod, width, height, fixed_seed=1234, want_rgb=True)
    assert np.allclose(counter_rgb.mean(axis=(0, 1)), mt_rgb.mean(axis=(0, 1)), rtol=0.01)
    assert counter_stats["segments"] == pytest.approx(mt_stats["segments"], rel=0.01)
    # per-pixel: 8x8 block means agree within Monte-Carlo noise
    blocks = lambda img: img[: height // 6 * 6, : width // 8 * 8].reshape(height // 6, 6, width // 8, 8, 3).mean(axis=(1, 3))
    assert np.abs(blocks(counter_rgb) - blocks(mt_rgb)).max() < 0.06


def test_monte_carlo_error_falls_as_one_over_sqrt_spp():
    width, height = 48, 27
    errs = []
    ref_scene = rt_amd.Scene.named("basic").set_sampling(8000)
    _, ref, _ = oracle.render(ref_scene.describe(width, height), width, height, seed=100)
    for spp in (4, 16, 64):
        scene = rt_amd.Scene.named("basic").set_sampling(spp)
        _, rgb, _ = oracle.render(scene.describe(width, height), width, height, seed=200 + spp)
        errs.append(np.sqrt(((rgb - ref) ** 2).mean()))
    # sample 0 of every pixel is the un-jittered centre ray (mg_ray_tracer.cpp:189), so low-spp images carry a small
    # aliasing bias at silhouettes on top of the 1/sqrt(spp) noise: allow for it
    assert 1.4 < errs[0] / errs[1] < 2.6 and 1.4 < errs[1] / errs[2] < 2.6, errs


# ---- opt-in: sm_ray_tracer's dielectric scatter (reference src/renderers/sm_ray_tracer.cpp:156-219) ---------------------------
def test_dielectric_refraction_obeys_snells_law():
    n = np.array([0.0, 1.0, 0.0])
    for ior in (1.31, 1.333, 1.52):
        for theta in (0.0, 0.3, 0.8, 1.2):
            d = np.array([math.sin(theta), -math.cos(theta), 0.0])  # coming down onto the surface from outside
            out, prob = oracle.dielectric_direction(d, n, ior, u=1.0)  # u = 1 never reflects (prob <= 1... u < prob false)
            # refracted: sin(theta_t) = sin(theta_i) / ior, continuing downward
            assert out[1] < 0
            assert math.isclose(out[0] / np.linalg.norm(out), math.sin(theta) / ior, abs_tol=2e-6)
            assert math.isclose(np.linalg.norm(out), 1.0, abs_tol=2e-6)
            # Schlick: r0 + (1 - r0) (1 - cos)^5, reference sm_ray_tracer.cpp:174-179
            r0 = ((1 - ior) / (1 + ior)) ** 2
            assert math.isclose(prob, r0 + (1 - r0) * (1 - math.cos(theta)) ** 5, rel_tol=1e-5)
            # u = 0 always reflects: mirror direction
            refl, _ = oracle.dielectric_direction(d, n, ior, u=0.0)
            assert np.allclose(refl, d - 2 * (d @ n) * n, atol=1e-6)


def test_dielectric_total_internal_reflection_and_vacuum():
    n = np.array([0.0, 1.0, 0.0])
    # inside glass going up at a grazing angle: sin2_t > 1 -> reflect_prob = 1, always reflected
    d = np.array([math.sin(1.2), math.cos(1.2), 0.0])
    out, prob = oracle.dielectric_direction(d, n, 1.52, u=0.999)
    assert prob == 1.0 and np.allclose(out, d - 2 * (d @ n) * n, atol=1e-6)
    # vacuum (index 1): straight through at any angle unless the Schlick term fires
    d = np.array([0.6, -0.8, 0.0])
    out, prob = oracle.dielectric_direction(d, n, 1.0, u=0.5)
    assert np.allclose(out, d, atol=1e-6) and math.isclose(prob, (1 - 0.8) ** 5, rel_tol=1e-5)


def test_sm_material_mode_only_changes_refracting_materials():
    scene = rt_amd.Scene.named("basic").set_sampling(4)  # lambert + metal only
    pod = scene.describe(64, 36)
    a, a_rgb, _ = oracle.render(pod, 64, 36, seed=9)
    b, b_rgb, _ = oracle.render(pod, 64, 36, seed=9, sm_materials=True)
    assert np.array_equal(a, b) and np.array_equal(a_rgb, b_rgb)
    scene = rt_amd.Scene.named("dielectric").set_sampling(4)
    pod = scene.describe(64, 36)
    a, _, _ = oracle.render(pod, 64, 36, seed=9)
    b, b_rgb, _ = oracle.render(pod, 64, 36, seed=9, sm_materials=True)
    assert not np.array_equal(a, b) and np.isfinite(b_rgb).all()


# ---- contract v3: normalize()'s reciprocal square root ------------------------------------------------------------------


def _every_significand(first_bits: int) -> np.ndarray:
    """All 2^24 floats of the two binades starting at the float with bit pattern `first_bits`."""
    return (np.uint32(first_bits) + np.arange(1 << 24, dtype=np.uint32)).view(np.float32)


@pytest.mark.parametrize("first_bits", [0x3F800000, 0x21800000, 0x5C800000], ids=["1..4", "2^-60..2^-58", "2^58..2^60"])
def test_inv_sqrt_is_the_correctly_rounded_value_but_for_one_significand(first_bits):
    """The contract's inv_sqrt (one Newton step from the truncated quotient) against 1/sqrt(x) evaluated in binary64 and
    rounded to binary32, for EVERY significand at both exponent parities: equal everywhere except x = 4^k (1 - 2^-23),
    where the step's first-order value is exactly a rounding midpoint and the result is 2^-k."""
    x = _every_significand(first_bits)
    got = oracle.inv_sqrt(x)
    rounded = (1.0 / np.sqrt(x.astype(np.float64))).astype(np.float32)
    differing = np.nonzero(got.view(np.uint32) != rounded.view(np.uint32))[0]
    assert len(differing) == 1
    bits = int(x.view(np.uint32)[differing[0]])
    assert bits & 0x00FFFFFF == 0x007FFFFE  # odd biased exponent (x just below an even power of two), significand 1.11...10
    k = ((bits >> 23) + 1 - 127) // 2  # x = 4^k (1 - 2^-23)
    assert got[differing[0]] == np.float32(2.0) ** np.float32(-k)
    assert rounded[differing[0]] == np.float32(2.0) ** np.float32(-k) * (np.float32(1) + np.float32(2.0**-23))
    # and it is within 0.5000002 ulp even there
    exact = 1.0 / np.sqrt(np.float64(x[differing[0]]))
    assert abs(float(got[differing[0]]) - exact) <= 0.5000002 * float(np.spacing(got[differing[0]]))


def test_inv_sqrt_step_barely_depends_on_the_estimate_it_starts_from():
    """Why a hardware estimate can stand in for the truncated quotient — and why that has to be CHECKED on the hardware:
    the step squares the estimate's error, so from any estimate within an ulp of the truth it gives the same binary32
    result for all but a handful of the 2^24 significands (those whose exact value lies within ~2^-47 of a rounding
    midpoint), and is within one ulp there.  rt_hip_kat_exhaustive_math establishes on the device under test that
    v_rsq_f32's estimate gives the definition's result for EVERY input (tests/test_gpu_parity.py)."""
    x = _every_significand(0x3F800000)
    want = oracle.inv_sqrt(x)
    nearest = (1.0 / np.sqrt(x.astype(np.float64))).astype(np.float32)
    for offset in (-1, 0, 1):
        estimate = (nearest.view(np.int32) + np.int32(offset)).view(np.float32)
        got = oracle.inv_sqrt_step(x, estimate)
        differing = np.nonzero(got.view(np.uint32) != want.view(np.uint32))[0]
        assert len(differing) <= 8, (offset, len(differing))
        assert np.all(np.abs(got.view(np.int32)[differing].astype(np.int64) - want.view(np.int32)[differing]) == 1)


def test_inv_sqrt_outside_the_positive_normal_floats():
    with np.errstate(all="ignore"):
        x = np.array([0.0, -0.0, np.inf, -1.0, np.nan, 1e-45, 1.1754942e-38, 3.4028235e38, 1.1754944e-38], dtype=np.float32)
        got = oracle.inv_sqrt(x)
        plain = (1.0 / np.sqrt(x.astype(np.float64))).astype(np.float32)
    assert got[0] == np.inf and got[1] == -np.inf and got[2] == 0.0 and np.isnan(got[3]) and np.isnan(got[4])
    assert np.array_equal(got[5:7], plain[5:7])  # subnormal arguments: the plain quotient
    assert np.array_equal(got[7:], plain[7:])  # largest and smallest normal: the step reproduces the rounded value


def test_normalize_of_unit_vectors_stays_put():
    """The case the contract change must not disturb: normalising an (almost) unit-length direction — metal_scatter does
    it on every hit (mg_ray_tracer.cpp:133) — keeps every component within one ulp."""
    rng = np.random.default_rng(5)
    v = rng.normal(size=(200000, 3))
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    d = (v[:, 0] * v[:, 0]).astype(np.float32)
    d = (v[:, 1].astype(np.float64) * v[:, 1] + d).astype(np.float32)  # fma chain of the contract's dot()
    d = (v[:, 2].astype(np.float64) * v[:, 2] + d).astype(np.float32)
    scaled = v * oracle.inv_sqrt(d)[:, None]
    assert np.max(np.abs(scaled.view(np.int32).astype(np.int64) - v.view(np.int32).astype(np.int64))) <= 1


def test_one_plane_is_accepted_and_selected_by_one_ordered_comparison():
    """rt_amd/csrc/scan.hpp test_one_plane (scenes with ONE plane): test_planes accepts the plane's candidate when the ray crosses it
    and `!(t < 0.001)` (reference mg_ray_tracer.cpp:46-52), select() then asks `distance >= 0` of it (:29-32,96-102) — a NaN distance
    passes the first and fails the second.  The kernels ask once: crosses and `t >= 0.001`.  Checked on the whole zoo of binary32
    classes (zeros of both signs, subnormals, huge values, infinities, NaNs) crossed with itself and on a million random pairs."""
    rng = np.random.default_rng(11)
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, -1e-38, 1e-6, -1e-6, 0.001, -0.001, 1.0, -1.0, 3e38, -3e38, np.inf, -np.inf, np.nan], dtype=np.float32)
    num = np.concatenate([np.repeat(special, len(special)), rng.normal(size=1_000_000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 1_000_000).astype(np.float32)])
    den = np.concatenate([np.tile(special, len(special)), rng.normal(size=1_000_000).astype(np.float32) * np.float32(10.0) ** rng.integers(-30, 30, 1_000_000).astype(np.float32)])
    with np.errstate(all="ignore"):
        t = ((-num) * (np.float32(1.0) / den)).astype(np.float32)  # contract v4: the rounded reciprocal, then one product
    crosses = ~(np.abs(den) <= np.float32(1e-6))
    accepted = crosses & ~(t < np.float32(0.001))  # (a NaN distance is "accepted" by the negated comparison, as in the reference)
    selected = accepted & (t >= 0)
    assert np.isnan(t[accepted]).any() and selected.sum() > 100_000  # the zoo does hold the cases the two rules differ on
    assert np.array_equal(selected, crosses & (t >= np.float32(0.001)))
    # the scalar-register kernels take the reciprocal UNGUARDED (rcp_in_band): v_rcp_f32 and one residual step — the rounded
    # quotient inside the band, NaN for an infinite den (0 * inf in the residual) where the quotient is 0.  Their dens are below
    # 2^60, infinite or NaN (device_scene::planes_tame), and neither a NaN nor a zero times anything passes the comparison.
    tame = ~(np.isfinite(den) & (np.abs(den) >= np.float32(2.0**60)))
    with np.errstate(all="ignore"):
        unguarded = np.where(np.isinf(den), np.float32(np.nan), np.float32(1.0) / den).astype(np.float32)
        t_unguarded = ((-num) * unguarded).astype(np.float32)
    assert np.isinf(den[crosses & tame]).any()
    assert np.array_equal(selected[tame], (crosses & (t_unguarded >= np.float32(0.001)))[tame])
    assert np.array_equal(t[selected & tame].view(np.uint32), t_unguarded[selected & tame].view(np.uint32))
