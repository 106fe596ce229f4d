"""CPU: the oracle against the pins of tests/golden/ref_probe.json."""
import json
import os

import numpy as np

import oracle_lib as O
import scenes as S

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = json.load(open(os.path.join(HERE, "golden", "ref_probe.json")))


def test_randhash_kat():
    for k, v in PROBE["randhash"].items():
        assert O.randhash(int(k)) == v


def test_minstd_kat():
    # Thrust documentation: 10000th draw of a default-constructed minstd_rand
    assert O.minstd_nth(1, 10000) == PROBE["minstd_10000th_seed1"]
    # seed 0 is mapped to 1 (linear_congruential_engine::seed)
    assert O.minstd_nth(0, 10000) == PROBE["minstd_10000th_seed1"]
    # seed is reduced mod 2^31-1
    assert O.minstd_nth(2147483647 + 5, 3) == O.minstd_nth(5, 3)


def test_uniform_stream():
    h0 = O.randhash(0)
    u = O.uniform_stream((h0 * h0 * h0) & 0xFFFFFFFF, 2)
    want = np.array(PROBE["uniforms_seed_randhash0_cubed"], dtype=np.float32)
    assert np.array_equal(u, want)
    u = O.uniform_stream(12345, 100000)
    assert u.min() >= 0.0 and u.max() <= 1.0


def test_uniform_can_reach_one():
    # x-1 >= 2^31-64 rounds up to 2^31 in float -> u == 1.0f.  x = m-1 follows x = 1
    # backwards: find it by stepping the generator from the state whose successor is m-1.
    m = 2147483647
    inv = pow(48271, -1, m)
    prev = ((m - 1) * inv) % m
    u = O.uniform_stream(prev, 1)
    assert u[0] == np.float32(1.0)


def test_default_scene_render_statistics():
    sph = S.default_spheres()
    for r in PROBE["renders"]:
        w, h = r["width"], r["height"]
        cam = S.orbit_camera(w, h)
        p = O.params(max_depth=r["depth"])
        total = 0
        acc = np.zeros((w * h, 3), dtype=np.float64)   # the recorded sums were taken in double
        for c in r["counters"]:
            out = O.render_pass(sph, None, cam, c, p)
            total += out["bounces"]
            acc += out["rgb"]
        assert total == r["path_bounces"]
        mean = (acc / len(r["counters"])).mean(axis=0)
        assert ["%.6f" % v for v in mean] == ["%.6f" % v for v in r["mean_rgb"]]
        centre = acc[(h // 2) * w + w // 2]
        # recorded with 9 significant digits
        assert ["%.9g" % v for v in centre] == ["%.9g" % v for v in r["centre_pixel_sum"]]


def test_subset_equals_full_pass():
    sph = S.default_spheres()
    cam = S.orbit_camera(64, 64)
    p = O.params(max_depth=8)
    full = O.render_pass(sph, None, cam, 3, p, want_hits=True)
    ids = np.arange(5, 64 * 64, 7, dtype=np.int32)
    sub = O.render_pass(sph, None, cam, 3, p, pixel_ids=ids, want_hits=True)
    assert np.array_equal(full["rgb"][ids], sub["rgb"])
    assert np.array_equal(full["hits"][ids], sub["hits"])


def test_thread_count_independent():
    sph = S.default_spheres()
    cam = S.orbit_camera(96, 96)
    p = O.params(max_depth=12)
    n = O.num_threads()
    a = O.render_pass(sph, None, cam, 1, p)
    O.set_num_threads(1)
    b = O.render_pass(sph, None, cam, 1, p)
    O.set_num_threads(n)
    assert np.array_equal(a["rgb"], b["rgb"]) and a["bounces"] == b["bounces"]


def test_live_counts_monotone_and_empty_input():
    sph = S.default_spheres()
    cam = S.orbit_camera(32, 32)
    p = O.params(max_depth=30)
    out = O.render_pass(sph, None, cam, 0, p)
    live = out["live"]
    assert live[0] == 32 * 32 and np.all(np.diff(live) <= 0)
    empty = O.render_pass(sph, None, cam, 0, p, pixel_ids=np.zeros(0, np.int32))
    assert empty["bounces"] == 0 and empty["rgb"].shape == (0, 3)


def test_tonemap_rule():
    # gamma 1/2.2, *255, clamp, truncate (color.cpp:41-62); NaN -> 255 through
    # BasicMath::min's comparison order (basic_math.cpp:58-64)
    x = np.array([[0.0, 1.0, 4.0], [0.5, 0.25, np.nan], [2.0, 8.0, -1.0]], dtype=np.float32)
    out = O.tonemap_8bit(x, 2)
    c = x / np.float32(2)
    g = np.power(c.astype(np.float64), 1.0 / 2.2) * 255
    assert out[0, 0] == 0 and out[0, 2] == 255          # 4/2=2 -> clamps
    assert out[0, 1] == int(g[0, 1])
    assert out[1, 2] == 255                             # NaN
    assert out[2, 2] == 255 or out[2, 2] == 0           # pow(negative) = NaN -> 255
    assert out[2, 2] == 255


def test_oracle_reproduces_its_committed_vectors():
    """tests/golden/oracle_*.npz (tools/make_oracle_goldens.py): radiance, branch codes, primitives and live counts
    of three small scenes as the oracle produced them when they were committed -- a change detector for the oracle."""
    import golden_scenes as G
    for name in G.NAMES:
        z = np.load(os.path.join(HERE, "golden", "oracle_%s.npz" % name))
        sph, polys, cam, depth, counters, stride_mode = G.scene(name)
        for c in counters:
            r = O.render_pass(sph, polys, cam, c, O.params(max_depth=depth, row_stride_is_width=stride_mode), want_events=True)
            assert np.array_equal(r["rgb"].view(np.uint32), z["rgb_%d" % c].view(np.uint32)), (name, c)
            assert np.array_equal(r["events"], z["events_%d" % c]) and np.array_equal(r["prims"], z["prims_%d" % c])
            assert np.array_equal(r["live"], z["live_%d" % c])
