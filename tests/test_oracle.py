"""Pins the CPU oracle (oracle/oracle.cpp).

The reference has no tests or golden vectors for this path and cannot be built in this image
(OpenCV is absent), so the oracle is "parity unpinned" against a reference binary.  What is
asserted here instead:
  * the known answers SURVEY.md records from the survey session's run of the unmodified reference
    (update-table entries, zero-weight selection index, offset truncation, first-update tie);
  * properties that follow directly from the cited reference lines;
  * agreement of the oracle's two math modes (glibc vs detmath) and two running-sum orders.
"""
import numpy as np
import pytest

import _oracle as orc

rng = np.random.default_rng(7)


def free_grid(name):
    return (orc.load_map(name) * 255).astype(np.uint8)


# ---- known answers recorded in SURVEY.md ---------------------------------------------------------
def test_update_tables_known_values():
    # SURVEY.md §7 "Map update as two 256-entry LUTs": 0->1/1, 5->6/4, 128->153/102, 213->255/170, 255->255/204
    fr, oc = orc.update_tables()
    for c, f, o in [(0, 1, 1), (5, 6, 4), (128, 153, 102), (213, 255, 170), (255, 255, 204)]:
        assert (fr[c], oc[c]) == (f, o)
    assert fr.min() >= 1 and oc.min() >= 1
    assert np.all(np.diff(fr.astype(int)) >= 0) and np.all(np.diff(oc.astype(int)) >= 0)


def test_scanner_offset_truncates_through_int():
    # SURVEY.md §7 hard part 8: {0.5,0.5,0} -> d = 0 ; {0,30,0} -> 30 exactly (slam/util.h:138-146)
    assert orc.scanner_offset([0.5, 0.5, 0])[0] == 0.0
    assert orc.scanner_offset([0, 30, 0])[0] == 30.0
    assert orc.scanner_offset([3, 4.9, 0])[0] == np.sqrt(float(int(9 + 4.9 * 4.9)))
    sp = orc.sensor_position([400, 400, np.pi], [0, 30, 0])
    assert sp[0] == pytest.approx(400 + np.cos(np.pi + np.pi / 2) * 30) and sp[1] == pytest.approx(370.0)


@pytest.mark.parametrize("order", [orc.ORDER_REFERENCE, orc.ORDER_BLOCKED])
def test_all_zero_weights_select_middle(order):
    # SURVEY.md §7 hard parts 3/4: N = 9, all weights 0 -> every output equals input particle 4 = (N-1)/2
    for N in (9, 1, 2, 64, 65, 1000):
        idx, P, s = orc.resample(np.zeros(N), orc.reference_uniforms(N, tick=3), order)
        assert s == 0.0 and np.all(idx == (N - 1) // 2)


@pytest.mark.parametrize("B", [400, 440, 720])
def test_fresh_canvas_weights_underflow_to_zero(B):
    # SURVEY.md §7 hard part 4 (verified there with the unmodified reference): sigma = 5, B >= 324 -> all weights 0,
    # and resampling then returns particle (N-1)/2 for every slot.
    N = 9
    maps = np.full((N, 120, 120), 128, np.uint8)
    x = 60 + rng.uniform(-3, 3, N)
    y = 60 + rng.uniform(-3, 3, N)
    th = rng.uniform(-1, 1, N)
    ang = np.linspace(-np.pi, np.pi, B, endpoint=False)
    w, _ = orc.update_weights(x, y, th, maps, ang, np.full(B, 40.0), 5.0, 50.0, slam_mode=True)
    assert np.all(w == 0.0)
    idx, _, _ = orc.resample(w, orc.reference_uniforms(N), orc.ORDER_REFERENCE)
    assert np.all(idx == 4)


def test_first_update_on_fresh_canvas_is_a_full_tie():
    # SURVEY.md §7 hard part 4: nothing is < 128 on a fresh canvas, every beam scores 0 + 0.1 -> weight 0.1^B
    N, B = 5, 90
    maps = np.full((N, 200, 200), 128, np.uint8)
    x = 100 + rng.uniform(-10, 10, N)
    y = 100 + rng.uniform(-10, 10, N)
    th = rng.uniform(-3, 3, N)
    ang = np.linspace(-np.pi, np.pi, B, endpoint=False)
    dists = rng.uniform(20, 60, B)
    w, hxy, hcell, st = orc.update_weights(x, y, th, maps, ang, dists, 5.0, 80.0, [0, 30, 0], slam_mode=True, want_hits=True)
    assert np.all(hcell == -1) and np.all(hxy == -1)
    assert np.all(w == w[0]) and w[0] == pytest.approx(0.1 ** B, rel=1e-12)
    # second update: the occupied cells drawn to 102 (table entry of 128) now produce hits
    assert (maps == 102).any() and (maps == 153).any()
    w2, _, hcell2, _ = orc.update_weights(x, y, th, maps, ang, dists, 5.0, 80.0, [0, 30, 0], slam_mode=True, want_hits=True)
    assert (hcell2 >= 0).any() and not np.all(w2 == w2[0])


# ---- properties that follow from the cited reference lines ---------------------------------------
def test_search_rule_equality_returns_mid():
    # slam/mcl.cpp:187-191: leading zero-weight particles + cutoff 0 -> the first probed index of the zero run
    idx, P, _ = orc.resample([0, 0, 0, 1.0], [0.0, 0.5, 0.999, 0.0])
    assert list(P) == [0, 0, 0, 1] and list(idx) == [1, 3, 3, 1]
    # single non-zero weight: everything maps to it (or, at cutoff 0, to the probe inside the zero run)
    idx, _, _ = orc.resample([0, 0, 5.0, 0, 0], [0.1, 0.9, 0.5, 0.3, 0.7])
    assert np.all(idx == 2)
    # result is clamped to N-1 (j starts at N-1, slam/mcl.cpp:183)
    idx, _, _ = orc.resample([1.0, 1.0], [0.9999999, 0.0])
    assert list(idx) == [1, 0]


def test_resample_orders_agree_and_lower_bound():
    for N in (1, 63, 64, 65, 4097, 20000):
        w = rng.uniform(0, 1, N) ** 8
        w[rng.uniform(size=N) < 0.1] = 0.0
        u = orc.reference_uniforms(N, tick=N)
        i0, P0, s0 = orc.resample(w, u, orc.ORDER_REFERENCE)
        i1, P1, s1 = orc.resample(w, u, orc.ORDER_BLOCKED)
        assert np.array_equal(P0, np.cumsum(w)) or np.allclose(P0, np.cumsum(w), rtol=0, atol=0)
        assert np.allclose(P1, P0, rtol=1e-13)
        assert np.array_equal(i0, i1), "a cutoff fell between two roundings of the running sum"
        lb = np.minimum(np.searchsorted(P0, u * s0, side="left"), N - 1)
        assert np.array_equal(i0, lb)


def test_systematic_resampler_is_stratified():
    N = 1000
    w = rng.uniform(0, 1, N)
    idx, P, s = orc.resample(w, [0.37], orc.ORDER_BLOCKED, resampler=1)
    assert np.all(np.diff(idx) >= 0)
    counts = np.bincount(idx, minlength=N)
    assert np.all(np.abs(counts - w / s * N) < 1.0 + 1e-9)


def test_reference_noise_source_is_libstdcpp():
    # slam/util.cpp:27-34: re-seed per draw; minstd_rand0::seed(0) behaves as seed(1); seeds are taken mod 2^31-1
    a = orc.reference_normals(4, tick0=0, tick_stride=0)
    b = orc.reference_normals(4, tick0=1, tick_stride=0)
    c = orc.reference_normals(4, tick0=2147483647 + 1, tick_stride=0)
    assert np.array_equal(a, b) and np.array_equal(b, c) and np.all(a == a[0])
    z = orc.reference_normals(200000, tick0=12345, tick_stride=1)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    assert np.array_equal(orc.reference_normals(10, 5, 1, 2.5), 2.5 * orc.reference_normals(10, 5, 1, 1.0))
    u = orc.reference_uniforms(100000, tick=9)
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    # first draw of minstd_rand0 seeded with 1: generate_canonical<double,53> = (48271-1 + (48271^2 % m - 1) * (m-1)) / (m-1)^2
    m = 2147483647
    g1, g2 = 16807 % m, (16807 * 16807) % m
    assert orc.reference_uniforms(1, tick=1)[0] == ((g1 - 1) + (g2 - 1) * float(m - 1)) / (float(m - 1) ** 2)


def test_triangular_sampler_is_two_draws_of_one_engine():
    """slam/util.cpp:36-43: ONE re-seed, two uniform(-s, s) draws from the same minstd_rand0, the
    first scaled by sqrt(6)/2.  Checked against an integer model of libstdc++'s generate_canonical."""
    m = 2147483647

    def canonical(x):  # two engine outputs per double (range 2^31 - 2 < 2^53)
        x1 = (16807 * x) % m
        x2 = (16807 * x1) % m
        return ((x1 - 1) + (x2 - 1) * float(m - 1)) / (float(m - 1) ** 2), x2

    for tick, sd in [(1, 1.0), (12345, 2.5), (987654321, 0.125)]:
        c1, st = canonical(tick % m or 1)
        c2, _ = canonical(st)
        expect = np.sqrt(6.0) / 2 * (c1 * (sd - -sd) + -sd) + (c2 * (sd - -sd) + -sd)
        assert orc.reference_triangular(1, tick, 0, sd)[0] == expect
    t = orc.reference_triangular(200000, 99, 1, 1.0)
    assert abs(t.mean()) < 0.01 and np.abs(t).max() <= np.sqrt(6) / 2 + 1
    x = np.array([0.0, 0.5, -0.5, 3.0, -7.0])
    assert np.array_equal(orc.pdf_triangular(1.5, x), np.maximum(0.0, 1 / np.sqrt(6 * 1.5 ** 2) - np.abs(x) / 6 / 1.5 ** 2))


def test_velocity_motion_model_matches_formula():
    """slam/motion.cpp:34-56 as written (the + between the two sines included)."""
    import math

    N = 300
    x, y, th = rng.uniform(0, 100, N), rng.uniform(0, 100, N), rng.uniform(-2 * np.pi, np.pi, N)
    z = orc.reference_normals(3 * N, 5, 1)
    vel, dt, al = [1.7, 0.3], 0.25, [1e-2, 2e-2, 3e-2, 4e-2, 5e-3, 6e-3]
    nx, ny, nth = orc.predict_velocity(x, y, th, vel, dt, al, z)
    vv, ww = vel[0] * vel[0], vel[1] * vel[1]     # std::pow(v, 2)
    sv = math.sqrt(al[0] * vv + al[1] * ww)
    sw = math.sqrt(al[2] * ww + al[3] * vv)
    sg = math.sqrt(al[4] * vv + al[5] * ww)
    for p in range(N):
        v, w, g = vel[0] + z[3 * p] * sv, vel[1] + z[3 * p + 1] * sw, z[3 * p + 2] * sg
        q = v / (1 if w == 0 else w)
        assert nx[p] == x[p] - q * (math.sin(th[p]) + math.sin(th[p] + (w + 1e-6) * dt))
        assert ny[p] == y[p] + q * (math.cos(th[p]) - math.cos(th[p] + (w + 1e-6) * dt))
        a = math.fmod(th[p] + w * dt + g * dt, 2 * math.pi)
        assert nth[p] == (a - 2 * math.pi if a > math.pi else a)
    # w == 0 divides by 1 (slam/motion.cpp:48,51); zero alphas are deterministic
    ox, oy, oth = orc.predict_velocity([1.0], [2.0], [0.5], [2.0, 0.0], 0.5, [0] * 6, [9.0, 9.0, 9.0])
    assert ox[0] == 1.0 - 2.0 * (math.sin(0.5) + math.sin(0.5 + 1e-6 * 0.5)) and oth[0] == 0.5
    # deterministic math agrees with libm to the stated tolerance
    dx, dy, dth = orc.predict_velocity(x, y, th, vel, dt, al, z, math=orc.DET)
    assert np.allclose(dx, nx, rtol=1e-12) and np.allclose(dy, ny, rtol=1e-12) and np.allclose(dth, nth, rtol=1e-12, atol=1e-15)


def test_predict_matches_formula_and_normalizes_angle():
    N = 1000
    x, y, th = rng.uniform(0, 100, N), rng.uniform(0, 100, N), rng.uniform(-2 * np.pi, np.pi, N)
    z = orc.reference_normals(3 * N, 77, 1)
    odom, al = [0.3, 2.5, 3.0], [5e-4, 5e-4, 1e-2, 1e-2]
    nx, ny, nth = orc.predict(x, y, th, odom, al, z)
    s1 = np.sqrt(al[0] * odom[0] ** 2 + al[1] * odom[1] ** 2)
    st = np.sqrt(al[2] * odom[1] ** 2 + al[3] * (odom[0] ** 2 + odom[2] ** 2))
    r1 = odom[0] - z[0::3] * s1
    t = odom[1] - z[1::3] * st
    assert np.allclose(nx, x + t * np.cos(th + r1), rtol=1e-14) and np.allclose(ny, y + t * np.sin(th + r1), rtol=1e-14)
    assert nth.max() <= np.pi and nth.min() > -2 * np.pi  # slam/util.cpp:45-51: negative angles are not wrapped up
    # zero alphas: deterministic motion (benchmark/mcl.cpp:62 calls it with the default alphas)
    dx, dy, dth = orc.predict([0.0], [0.0], [0.0], [0.0, 1.0, 0.0], [0, 0, 0, 0], [1.0, -1.0, 0.5])
    assert (dx[0], dy[0], dth[0]) == (1.0, 0.0, 0.0)
    dxd, dyd, dthd = orc.predict(x, y, th, odom, al, z, math=orc.DET)
    assert np.allclose(dxd, nx, rtol=1e-14) and np.array_equal(dthd, nth) or np.allclose(dthd, nth, atol=1e-15)


def test_localization_mode_equals_fresh_clone_per_beam():
    # SURVEY.md §8a row aL: shared-map mode == unmodified measurement_model_beam on a fresh clone per beam
    g = free_grid("floor_plan")
    N, B = 6, 40
    x, y, th = rng.uniform(100, 1200, N), rng.uniform(50, 550, N), rng.uniform(-3, 3, N)
    ang = np.linspace(-np.pi, np.pi, B, endpoint=False)
    dists = rng.uniform(10, 300, B)
    off = [0, 30, 0.1]
    w, _ = orc.update_weights(x, y, th, g, ang, dists, 5.0, 500.0, off, slam_mode=False)
    d, ths = orc.scanner_offset(off)
    for p in range(N):
        sx, sy = x[p] + np.cos(th[p] + ths) * d, y[p] + np.sin(th[p] + ths) * d
        acc = 0.0
        for b in range(B):
            wb = orc.measurement_model_beam(g.copy(), sx, sy, th[p] + ang[b] + off[2], dists[b] * dists[b], 5.0, 250000.0)
            acc += np.log(wb)
        assert w[p] == pytest.approx(np.exp(acc), rel=1e-12)


def test_raycast_mapping_semantics():
    g = np.full((50, 50), 128, np.uint8)
    g[:, 30] = 0  # wall at column 30
    before = g.copy()
    # horizontal ray from (10.5, 24.5): y row = 50 - 24.5 - 1 = 24
    hit, cell, st = orc.raycast_mapping(g, 10.5, 24.5, 0.0, 15.0 ** 2, 100.0 ** 2)
    assert tuple(hit) == (30.5, 24.5) and cell == 24 * 50 + 30
    fr, oc = orc.update_tables()
    assert np.all(g[24, 11:25] == fr[128])            # r^2 < z^2: free update (distance 1..14)
    assert g[24, 25] == oc[128]                        # first cell with r^2 >= z^2: occupied update
    assert np.all(g[24, 26:30] == 128) and g[24, 30] == 0  # beyond: untouched, and the loop stops at the hit
    assert g[24, 10] == 128                            # the start cell is never visited
    assert np.array_equal(np.delete(g, 24, 0), np.delete(before, 24, 0))
    # z == zmax: no occupied update (slam/raycast.cpp:205)
    g2 = np.full((50, 50), 128, np.uint8)
    hit, cell, _ = orc.raycast_mapping(g2, 10.5, 24.5, 0.0, 20.0 ** 2, 20.0 ** 2)
    assert cell == -1 and tuple(hit) == (-1, -1)
    assert np.all(g2[24, 11:30] == fr[128]) and np.all(g2[24, 30:] == 128)
    # leaving the grid ends the ray; x in (-1, 0) still maps to column 0 (truncation, slam/util.h:30-33)
    g3 = np.full((20, 20), 128, np.uint8)
    _, _, st = orc.raycast_mapping(g3, 2.5, 10.5, np.pi, 100.0, 400.0)
    assert np.all(g3[9, 0:2] == fr[128]) and st["cells_read"] == 2 + 0 or st["cells_read"] >= 2
    # read-only flag changes nothing but the stores
    g4, g5 = before.copy(), before.copy()
    h4 = orc.raycast_mapping(g4, 10.5, 24.5, 0.3, 225.0, 1e4, write=True)
    h5 = orc.raycast_mapping(g5, 10.5, 24.5, 0.3, 225.0, 1e4, write=False)
    assert np.array_equal(h4[0], h5[0]) and h4[1] == h5[1] and np.array_equal(g5, before) and not np.array_equal(g4, before)


def test_math_modes_agree_on_cells_and_weights():
    g = free_grid("floor_plan")
    N, B = 300, 90
    x, y, th = rng.uniform(50, 1250, N), rng.uniform(30, 570, N), rng.uniform(-3, 3, N)
    ang, dists = orc.make_scan(orc.load_map("floor_plan"), [400, 300, 0.2], [0, 30, 0], B)
    w0, xy0, c0, s0 = orc.update_weights(x, y, th, g, ang, dists, 5.0, 500.0, [0, 30, 0], want_hits=True, math=orc.LIBM)
    w1, xy1, c1, s1 = orc.update_weights(x, y, th, g, ang, dists, 5.0, 500.0, [0, 30, 0], want_hits=True, math=orc.DET)
    mism = int(np.sum(c0 != c1))
    assert mism <= 1, f"{mism} of {N * B} hit cells differ between glibc and detmath"
    ok = w0 > 0
    assert np.allclose(w1[ok], w0[ok], rtol=1e-11) and np.array_equal(w0 == 0, w1 == 0)
    same = c0 == c1
    assert np.allclose(xy1[same], xy0[same], rtol=1e-12, atol=1e-9)
    assert abs(s0["cells_read"] - s1["cells_read"]) <= 4


def test_fake_lidar_and_scan_conversion():
    free = orc.load_map("map")
    pose = [400.0, 400.0, np.pi / 90]
    sensor = orc.sensor_position(pose, [0, 30, 0])
    hits = orc.fake_lidar_scan(free.astype(np.int32), sensor, 0.0, 2 * np.pi, 500.0, 90)
    ang, dist = orc.hits_to_scan(hits, sensor, 0.0, 2 * np.pi, 500.0)
    step = 2 * np.pi / 90
    assert np.allclose(ang, np.arange(90) * step - np.pi)
    got = hits[:, 0] != -1
    assert got.any() and (~got).any() or got.all()
    rows = free.shape[0]
    for (hx, hy) in hits[got]:
        assert free[int(rows - hy - 1), int(hx)] == 0  # a hit lies in a wall cell
    assert np.all(dist[~got] == 500.0) and np.all(dist[got] < 500.0)
    # raycast<int> == raycast<uchar> == raycast<double> when no limit interferes (same walk, different tests)
    for k in range(0, 90, 7):
        th = sensor[2] - np.pi + k * step
        a = orc.raycast(2, free.astype(np.int32), sensor[0], sensor[1], th, 1e12)
        b = orc.raycast(1, (free * 255).astype(np.uint8), sensor[0], sensor[1], th, 1e12)
        c = orc.raycast(0, free.astype(np.float64), sensor[0], sensor[1], th, 1e12)
        assert np.array_equal(a, b) and np.array_equal(a, c)


def test_sort_orders():
    w = np.array([0.3, 0.9, 0.3, 0.9, 0.1, 0.9])
    assert list(orc.sort_order(w, 0)) == [1, 3, 5, 0, 2, 4]  # (weight desc, slot asc)
    big = rng.integers(0, 5, 2000) / 7.0
    p0, p1 = orc.sort_order(big, 0), orc.sort_order(big, 1)   # canonical vs the reference's unstable std::sort
    assert np.array_equal(big[p0], big[p1]) and sorted(p1) == list(range(2000))


def test_average_pose():
    x, y, th = rng.uniform(0, 10, 500), rng.uniform(0, 10, 500), rng.uniform(-0.5, 0.5, 500)
    a = orc.average_pose(x, y, th)
    assert a[0] == pytest.approx(x.mean()) and a[1] == pytest.approx(y.mean())
    assert a[2] == pytest.approx(np.arctan2(np.sin(th).sum(), np.cos(th).sum()))


def test_philox_streams_are_shard_independent():
    z = orc.philox_normals(0x5EED0001, 3, 0, 1000)
    assert np.array_equal(z[3 * 400:], orc.philox_normals(0x5EED0001, 3, 400, 600))
    assert not np.array_equal(z, orc.philox_normals(0x5EED0001, 4, 0, 1000))
    assert abs(z.mean()) < 0.06 and abs(z.std() - 1) < 0.05
    u = orc.philox_uniforms(1, 0, 0, 100000)
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.01
    assert np.array_equal(u[5000:6000], orc.philox_uniforms(1, 0, 5000, 1000))
