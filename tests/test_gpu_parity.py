"""GPU parity: every call goes through the C ABI (libslam_hip.so) and is compared with the CPU oracle
on the same seeded inputs and the same injected noise streams.

Bar (BASELINE.json north_star): raycast hit cells, grid updates and resampled indices bit-exact;
weights and poses within 1e-5 relative.  Against the oracle's detmath mode (same elementary
functions as the kernels) EVERYTHING is compared bit for bit, weights and poses included; against
its glibc mode (what the reference calls) the integer results are compared exactly and the
floating-point ones with the stated tolerance.
"""
import os
import zlib

import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

REL_TOL = 1e-5  # north_star tolerance for weights and poses
ODOM = [2.5, 0.02, 0.02]            # benchmark/mcl.cpp:60
ALPHAS = [5e-4, 5e-4, 1e-2, 1e-2]   # benchmark/mcl.cpp:61
OFFSET = [0.0, 30.0, 0.0]           # benchmark/mcl.cpp:28


@pytest.fixture(scope="module")
def sa():
    import slam_amd

    slam_amd.load_library()
    return slam_amd


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bits(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    bad = np.flatnonzero(bits(a).ravel() != bits(b).ravel())
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} values differ bitwise, first at {bad[:5]}: {a.ravel()[bad[:3]]} vs {b.ravel()[bad[:3]]}"


def grid_of(name):
    return (orc.load_map(name) * 255).astype(np.uint8)


def cloud(rng, n, cx, cy, sxy=25.0, sth=0.2, th0=0.3):
    return cx + sxy * rng.standard_normal(n), cy + sxy * rng.standard_normal(n), th0 + sth * rng.standard_normal(n)


# ---- K1 motion ------------------------------------------------------------------------------------
def test_predict_injected_normals(sa):
    rng = np.random.default_rng(11)
    N = 5000
    x, y, th = rng.uniform(0, 1000, N), rng.uniform(0, 1000, N), rng.uniform(-2 * np.pi, np.pi, N)
    f = sa.Filter(N, 100, 100, noise=sa.capi.NOISE_INJECTED)
    f.set_poses(x, y, th)
    gx, gy, gth = x, y, th
    for step in range(4):
        z = orc.reference_normals(3 * N, tick0=1000 * step + 1, tick_stride=1)  # libstdc++ draws, slam/util.cpp:27-34
        f.set_noise(unit_normals=z)
        f.predict(ODOM, ALPHAS)
        ox, oy, oth = orc.predict(gx, gy, gth, ODOM, ALPHAS, z, math=orc.DET)
        lx, ly, lth = orc.predict(gx, gy, gth, ODOM, ALPHAS, z, math=orc.LIBM)
        px, py, pth = f.poses()
        assert_bits(px, ox, "x"), assert_bits(py, oy, "y"), assert_bits(pth, oth, "theta")
        assert np.allclose(px, lx, rtol=REL_TOL) and np.allclose(py, ly, rtol=REL_TOL) and np.allclose(pth, lth, rtol=REL_TOL, atol=1e-12)
        gx, gy, gth = ox, oy, oth
    f.close()


def test_predict_philox_matches_host_stream_and_shards(sa):
    N = 4096
    seed = 0xABCDEF01
    f = sa.Filter(N, 64, 64, seed=seed)
    x0, y0, th0 = f.poses()
    halves = [sa.Filter(N // 2, 64, 64, seed=seed, sort=sa.capi.SORT_NONE, global_first=k * N // 2, global_n=N) for k in range(2)]
    gx, gy, gth = x0, y0, th0
    for step in range(3):
        f.predict(ODOM, ALPHAS)
        for hf in halves:
            hf.predict(ODOM, ALPHAS)
        z = orc.philox_normals(seed, step, 0, N)
        gx, gy, gth = orc.predict(gx, gy, gth, ODOM, ALPHAS, z, math=orc.DET)
        px, py, pth = f.poses()
        assert_bits(px, gx, "x"), assert_bits(py, gy, "y"), assert_bits(pth, gth, "theta")
        hx = np.concatenate([hf.poses()[0] for hf in halves])
        assert_bits(hx, gx, "sharded x")
    for h in [f] + halves:
        h.close()


# ---- K2a raycast + likelihood, localization mode -------------------------------------------------
def test_predict_velocity_model(sa):
    """slam/motion.cpp:34-56 per particle: injected normals and the Philox stream, both against the oracle."""
    rng = np.random.default_rng(12)
    N, seed = 5000, 0x7E1
    x, y, th = rng.uniform(0, 500, N), rng.uniform(0, 500, N), rng.uniform(-2 * np.pi, np.pi, N)
    vel, dt, al = [1.7, 0.3], 0.25, [1e-2, 2e-2, 3e-2, 4e-2, 5e-3, 6e-3]
    z = rng.standard_normal(3 * N)
    f = sa.Filter(N, 600, 600, noise=sa.capi.NOISE_INJECTED)
    f.set_poses(x, y, th)
    f.set_noise(unit_normals=z)
    f.predict_velocity(vel, dt, al)
    ex = orc.predict_velocity(x, y, th, vel, dt, al, z, math=orc.DET)
    for q in range(3):
        assert_bits(f.poses()[q], ex[q], "injected normals")
    lm = orc.predict_velocity(x, y, th, vel, dt, al, z)
    for q in range(3):
        assert np.allclose(f.poses()[q], lm[q], rtol=REL_TOL, atol=1e-9)
    f.close()
    f = sa.Filter(N, 600, 600, seed=seed)
    f.set_poses(x, y, th)
    f.predict_velocity([2.0, 0.0], 0.5, [0.0] * 6)          # w == 0: divides by 1, no noise
    st = orc.predict_velocity(x, y, th, [2.0, 0.0], 0.5, [0.0] * 6, np.zeros(3 * N), math=orc.DET)
    f.predict_velocity(vel, dt, al)                          # second predict: Philox step 1
    ex = orc.predict_velocity(*st, vel, dt, al, orc.philox_normals(seed, 1, 0, N), math=orc.DET)
    for q in range(3):
        assert_bits(f.poses()[q], ex[q], "philox stream")
    with pytest.raises(RuntimeError):
        f.predict_velocity([np.nan, 0.0], 0.1, [0.0] * 6)
    f.close()


RAYCASTERS = {"fast": 0, "walk": 1, "beam_lanes": 2, "particle_lanes": 3}


@pytest.mark.parametrize("raycaster", ["beam_lanes", "particle_lanes", "walk"])
@pytest.mark.parametrize("mapname,N,B,zmax,stddev", [("floor_plan", 600, 360, 500.0, 5.0), ("map", 300, 90, 500.0, 5.0),
                                                     ("maze", 200, 65, 150.5, 2.0), ("floor_plan", 128, 1, 40.0, 5.0)])
def test_weights_localization(sa, raycaster, mapname, N, B, zmax, stddev):
    rng = np.random.default_rng(zlib.crc32(f"{mapname}-{N}-{B}".encode()))
    free = orc.load_map(mapname)
    g = (free * 255).astype(np.uint8)
    rows, cols = g.shape
    true = [cols * 0.4, rows * 0.45, 0.3]
    ang, dist = orc.make_scan(free, true, OFFSET, B, max_dist=zmax)
    x, y, th = cloud(rng, N, true[0], true[1])
    x[:8] = rng.uniform(-60, cols + 60, 8)   # some particles outside / at the border
    y[:8] = rng.uniform(-60, rows + 60, 8)
    f = sa.Filter(N, rows, cols, raycaster=RAYCASTERS[raycaster])
    f.set_map(g)
    f.set_poses(x, y, th)
    f.record_hits(True)
    f.update_weights(ang, dist, stddev, zmax, OFFSET)
    w = f.weights()
    hxy, hcell = f.hits()
    ow, oxy, ocell, ost = orc.update_weights(x, y, th, g, ang, dist, stddev, zmax, OFFSET, want_hits=True, math=orc.DET)
    assert np.array_equal(hcell, ocell), f"{np.sum(hcell != ocell)} hit cells differ"
    assert_bits(hxy, oxy, "hit positions")
    assert_bits(w, ow, "weights")
    if raycaster == "walk":
        assert f.stats() == ost
    # against glibc (the reference's own libm): integer results exact up to rare last-bit cell flips
    lw, lxy, lcell, _ = orc.update_weights(x, y, th, g, ang, dist, stddev, zmax, OFFSET, want_hits=True, math=orc.LIBM)
    flips = int(np.sum(hcell != lcell))
    assert flips <= max(1, N * B // 20000), f"{flips} of {N * B} hit cells differ from the glibc oracle"
    ok = (lw > 1e-290) & (np.all(hcell == lcell, axis=1))
    assert np.allclose(w[ok], lw[ok], rtol=REL_TOL)
    f.close()


def test_weights_of_a_scan_longer_than_any_lds_row(sa):
    """2000 beams (more than the 1536 the wave-per-particle kernel once kept as a row of doubles in LDS): its in-order sum now
    comes from a short list that is flushed whenever the next 64 beams might not fit (kwl_flush), whatever the beam count.
    Converged cloud (nearly every beam within 4 sigma: every group flushes) and a wide one (a handful of listed beams)."""
    rng = np.random.default_rng(2000)
    free = orc.load_map("floor_plan")
    g = (free * 255).astype(np.uint8)
    rows, cols = g.shape
    true = [cols * 0.4, rows * 0.45, 0.3]
    B = 2000
    ang, dist = orc.make_scan(free, true, OFFSET, B, max_dist=500.0)
    for spread, N in ((0.3, 96), (200.0, 160)):
        x, y, th = true[0] + spread * rng.standard_normal(N), true[1] + spread * rng.standard_normal(N), true[2] + 0.01 * spread * rng.standard_normal(N)
        for log_domain in (False, True):
            f = sa.Filter(N, rows, cols, raycaster=RAYCASTERS["beam_lanes"], weight_domain=sa.capi.WEIGHT_LOG if log_domain else sa.capi.WEIGHT_LINEAR)
            f.set_map(g)
            f.set_poses(x, y, th)
            f.update_weights(ang, dist, 5.0, 500.0, OFFSET)
            ow, _ = orc.update_weights(x, y, th, g, ang, dist, 5.0, 500.0, OFFSET, math=orc.DET, n_threads=8, log_domain=log_domain)
            assert_bits(f.weights(), ow, f"weights, spread {spread}, log domain {log_domain}")
            f.close()


@pytest.mark.parametrize("N", [4097, 333])     # particle lanes with a ragged last block / beam lanes
@pytest.mark.parametrize("mapname,zmax", [("floor_plan", 500.0), ("floor_plan", 3.2), ("tiny", 40.0), ("tiny", 0.6)])
def test_weights_hostile_poses(sa, N, mapname, zmax):
    """Particles far outside the grid, at int-overflowing and 2^53-sized coordinates, on cell corners and
    with huge headings; a 5x7 map; ranges below one cell.  Hit cells, positions and weights equal
    the oracle's bit for bit in every kernel shape (the order keys and the exact-jump set-up must
    cope with all of it)."""
    rng = np.random.default_rng(zlib.crc32(f"hostile-{N}-{mapname}-{zmax}".encode()))
    if mapname == "tiny":
        g = np.full((5, 7), 255, np.uint8)
        g[2, 4] = 0
    else:
        g = (orc.load_map(mapname) * 255).astype(np.uint8)
    rows, cols = g.shape
    B = 24
    ang = np.linspace(-np.pi, np.pi, B, endpoint=False)
    dist = rng.uniform(0.0, zmax, B)
    dist[::5] = zmax
    x = rng.uniform(-5, cols + 5, N)
    y = rng.uniform(-5, rows + 5, N)
    th = rng.uniform(-7, 7, N)
    k = N // 8
    x[:k] = rng.choice([-1e6, 3e9, -5e9, 1e15, 9.1e15, 2.0 ** 31, -(2.0 ** 31) - 1], k)
    y[k:2 * k] = rng.choice([-1e6, 3e9, -5e9, 1e15, 2.0 ** 52, 0.0, -0.0], k)
    x[2 * k:3 * k] = np.floor(x[2 * k:3 * k])
    y[2 * k:3 * k] = np.floor(y[2 * k:3 * k])
    # (16384: where the table-driven sin / cos hands over to the polynomial form; multiples of pi/128: the table's own arguments)
    th[3 * k:4 * k] = rng.choice([1e5, -1e5, 1e300, 0.0, np.pi / 2, -np.pi, 16383.999, -16384.0, 16384.5, 25735.9, 37 * np.pi / 128, -129 * np.pi / 128], k)
    ow, oxy, ocell, _ = orc.update_weights(x, y, th, g, ang, dist, 2.0, zmax, [0.5, 1.5, 0.1], want_hits=True, math=orc.DET)
    for shape in ("fast", "beam_lanes", "particle_lanes", "walk"):
        f = sa.Filter(N, rows, cols, raycaster=RAYCASTERS[shape])
        f.set_map(g)
        f.set_poses(x, y, th)
        f.record_hits(True)
        f.update_weights(ang, dist, 2.0, zmax, [0.5, 1.5, 0.1])
        hxy, hcell = f.hits()
        assert np.array_equal(hcell, ocell), f"{shape}: {np.sum(hcell != ocell)} hit cells differ"
        assert_bits(hxy, oxy, f"{shape}: hit positions")
        assert_bits(f.weights(), ow, f"{shape}: weights")
        f.close()


@pytest.mark.parametrize("N,B", [(1, 1), (1, 129), (2, 64), (65, 65)])
def test_smallest_filters_both_modes(sa, N, B):
    """One or two particles, one beam, beam counts around the wavefront size: full steps in both modes."""
    rng = np.random.default_rng(N * 1000 + B)
    free = orc.load_map("map")
    g = (free * 255).astype(np.uint8)[200:420, 300:560].copy()
    rows, cols = g.shape
    ang = np.linspace(-np.pi, np.pi, B, endpoint=False)
    dist = rng.uniform(5.0, 90.0, B)
    x, y, th = cloud(rng, N, cols / 2, rows / 2, sxy=3.0, sth=0.3, th0=0.2)
    seed = 99
    for mode in (sa.capi.MODE_LOCALIZATION, sa.capi.MODE_SLAM):
        slam = mode == sa.capi.MODE_SLAM
        f = sa.Filter(N, rows, cols, mode=mode, seed=seed)
        if not slam:
            f.set_map(g)
        f.set_poses(x, y, th)
        state, maps = (x, y, th), (np.full((N, rows, cols), 128, np.uint8) if slam else g)
        for step in range(3):
            f.predict(ODOM, ALPHAS)
            f.update(ang, dist, 5.0, 90.0, OFFSET)
            ox, oy, oth = orc.predict(*state, ODOM, ALPHAS, orc.philox_normals(seed, step, 0, N), math=orc.DET)
            ow, _ = orc.update_weights(ox, oy, oth, maps, ang, dist, 5.0, 90.0, OFFSET, slam_mode=slam, math=orc.DET)
            idx, _, _ = orc.resample(ow, orc.philox_uniforms(seed, step, 0, N), orc.ORDER_BLOCKED)
            src = idx[orc.sort_order(ow[idx], 0)]
            state = (ox[src], oy[src], oth[src])
            if slam:
                maps = maps[src].copy()
            assert np.array_equal(f.last_indices(), idx)
            for q in range(3):
                assert_bits(f.poses()[q], state[q], f"mode {mode}, pose column {q}, step {step}")
            assert_bits(f.weights(), ow[src], f"mode {mode}, weights, step {step}")
            if slam:
                for p in range(N):
                    assert np.array_equal(f.map_of(p), maps[p])
        f.close()


def test_weights_degenerate_underflow_and_tiny_ranges(sa):
    rng = np.random.default_rng(5)
    g = grid_of("floor_plan")
    rows, cols = g.shape
    N = 64
    x, y, th = cloud(rng, N, 500, 300)
    f = sa.Filter(N, rows, cols, raycaster=sa.capi.RAYCAST_FAST_PARTICLE_LANES)
    f.set_map(g)
    f.set_poses(x, y, th)
    # sigma = 5 and B = 720 -> exp underflows to exactly 0 for every particle (SURVEY.md §7 hard part 4)
    ang = np.linspace(-np.pi, np.pi, 720, endpoint=False)
    f.update_weights(ang, np.full(720, 100.0), 5.0, 500.0, OFFSET)
    assert np.all(f.weights() == 0.0)
    f.resample()
    assert np.all(f.last_indices() == (N - 1) // 2)
    # z == zmax, zmax < 1, zmax = 0
    f.set_poses(x, y, th)
    for zmax in (0.0, 0.6, 1.0, 2.5, 37.0):
        d = np.full(33, zmax)
        f.update_weights(ang[:33], d, 1.5, zmax, [0.5, 0.5, 0.25])
        ow, _ = orc.update_weights(x, y, th, g, ang[:33], d, 1.5, zmax, [0.5, 0.5, 0.25], math=orc.DET)
        assert_bits(f.weights(), ow, f"weights zmax={zmax}")
    f.close()


def test_particle_lane_kernel_that_sums_its_own_beams(sa):
    """SLAM_RAYCAST_FAST_PARTICLE_LANES_DIRECT (what large handles get: no slab, no summing kernel) against the
    segmented shape and the oracle, on a particle count that is no multiple of 64, in both weight domains."""
    rng = np.random.default_rng(77)
    g = grid_of("floor_plan")
    rows, cols = g.shape
    N, B = 5003, 77
    x, y, th = cloud(rng, N, 640, 300)
    x[:40] = rng.uniform(-30, cols + 30, 40)  # some from outside the grid and from inside walls
    y[:40] = rng.uniform(-30, rows + 30, 40)
    ang = np.linspace(-np.pi, np.pi, B, endpoint=False)
    dist = rng.uniform(5.0, 400.0, B)
    dist[::9] = 500.0
    got = {}
    for dom in (sa.capi.WEIGHT_LINEAR, sa.capi.WEIGHT_LOG):
        for rc in (sa.capi.RAYCAST_FAST_PARTICLE_LANES_DIRECT, sa.capi.RAYCAST_FAST_PARTICLE_LANES):
            f = sa.Filter(N, rows, cols, raycaster=rc, weight_domain=dom)
            f.set_map(g)
            f.set_poses(x, y, th)
            f.record_hits(True)
            f.update_weights(ang, dist, 5.0, 500.0, OFFSET)
            got[dom, rc] = (f.weights(), *f.hits())
            f.close()
        a, b = got[dom, sa.capi.RAYCAST_FAST_PARTICLE_LANES_DIRECT], got[dom, sa.capi.RAYCAST_FAST_PARTICLE_LANES]
        assert_bits(a[0], b[0], "weights, two shapes")
        assert_bits(a[1], b[1], "hit positions, two shapes")
        assert np.array_equal(a[2], b[2])
    sub = np.r_[0:64, rng.choice(N, 64, replace=False), N - 3:N]
    ow, oxy, ocell, _ = orc.update_weights(x[sub], y[sub], th[sub], g, ang, dist, 5.0, 500.0, OFFSET, want_hits=True, math=orc.DET)
    w, hxy, hcell = got[sa.capi.WEIGHT_LINEAR, sa.capi.RAYCAST_FAST_PARTICLE_LANES_DIRECT]
    assert_bits(w[sub], ow, "weights vs oracle")
    assert_bits(hxy[sub], oxy, "hit positions vs oracle")
    assert np.array_equal(hcell[sub], ocell)
    with pytest.raises(sa.SlamError):
        sa.Filter(16, 64, 64, mode=sa.capi.MODE_SLAM, raycaster=sa.capi.RAYCAST_FAST_PARTICLE_LANES_DIRECT)


def test_cone_planes_built_on_the_device_equal_the_host_construction(sa):
    """slam_mcl_set_map_u8 builds the eight cone-clearance planes on the device (cone_gpu.h) from the resident raw map;
    they must equal the host construction of clearance.h (itself checked against the brute-force definition on the CPU,
    tests/test_ray_host.py) byte for byte, frames and padding of the stride aside: random maps of awkward sizes (rows
    and columns around the 64-cell chunks of the row scan), densities from empty to full, and the two real maps."""
    import ctypes

    hh = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_harness", "libray_harness.so"))
    hh.hh_cone_plane_stride.restype = ctypes.c_ulong
    P = ctypes.c_void_p
    rng = np.random.default_rng(77)
    cases = [(rng.integers(1, 200), rng.integers(1, 200), d) for d in (0.0, 0.002, 0.05, 0.3, 1.0) for _ in range(3)]
    cases += [(59, 60, 0.03), (60, 124, 0.03), (61, 188, 0.01), (300, 63, 0.01), (1, 1, 0.0), (2, 700, 0.004), (700, 2, 0.004)]
    grids = [np.where(rng.random((int(r), int(c))) < d, 0, 255).astype(np.uint8) for r, c, d in cases]
    grids += [(orc.load_map("floor_plan") * 255).astype(np.uint8), (orc.load_map("maze") * 255).astype(np.uint8)]
    from slam_amd import synth

    grids.append(synth.make_grid(3000)[:2900, :])  # corridors and rooms of the synthetic benchmark grid, not square
    for g in grids:
        rows, cols = g.shape
        f = sa.Filter(8, rows, cols)
        f.set_map(g)
        got = f.clearance()
        f.close()
        stride = hh.hh_cone_plane_stride(rows, cols)
        assert got.shape == (8, stride)
        want = np.zeros(8 * stride, dtype=np.uint8)
        hh.hh_build_cone_planes(g.ctypes.data_as(P), rows, cols, want.ctypes.data_as(P))
        n = (rows + 4) * (cols + 4)
        for o in range(8):
            a, b = got[o, :n].reshape(rows + 4, cols + 4), want[o * stride:o * stride + n].reshape(rows + 4, cols + 4)
            assert np.array_equal(a, b), f"plane {o} of a {rows}x{cols} map: {np.argwhere(a != b)[:5]}"


def test_kernel_shape_and_order_follow_the_cloud_without_changing_a_bit(sa):
    """do_weights chooses per pass from the cloud's extent (k_order_keys / k_order_box leave it on the device and in
    pinned host memory): lanes over particles for a dense cloud, the wave-per-particle kernel for a sparse one (more
    than 4 cells of bounding box per particle), heading-first or place-first order.  The seeding calls refresh the
    extent, so a uniformly seeded cloud on the maze gets the second kernel from its first pass on and a Gaussian cloud
    seeded afterwards the first again.  Every pass of one cloud must give the same bits, and a subset must equal the
    oracle (which is how the two kernels are compared with each other)."""
    free = orc.load_map("maze")
    g = (free * 255).astype(np.uint8)
    rows, cols = g.shape
    N, B = 20000, 90
    ang, dist = orc.make_scan(free, [1944.5, 2188.5, 1.2], OFFSET, B)
    f = sa.Filter(N, rows, cols, seed=99, sort=sa.capi.SORT_NONE)
    f.set_map(g)
    for seeding in ("uniform", "gaussian", "uniform"):
        if seeding == "uniform":
            f.seed_uniform_free()  # 290 cells of bounding box per particle: sparse
        else:
            f.seed_gaussian([1944.5, 2188.5, 1.2], [3.0, 3.0, 0.1])  # far below one cell per particle: dense
        x, y, th = f.poses()
        ws = []
        for _ in range(4):
            f.update_weights(ang, dist, 5.0, 500.0, OFFSET)
            ws.append(f.weights())
        for k in range(1, 4):
            assert_bits(ws[k], ws[0], f"{seeding} cloud: pass {k} against pass 0")
        sub = np.arange(0, N, 997)
        ow, _ = orc.update_weights(x[sub], y[sub], th[sub], g, ang, dist, 5.0, 500.0, OFFSET, math=orc.DET)
        assert_bits(ws[0][sub], ow, f"{seeding} cloud: subset vs oracle")
    f.close()


def test_heading_buckets_sized_by_the_cloud_change_no_bit(sa, tmp_path):
    """k_order_keys sizes the heading buckets of the pose-coherent order from the cloud's extent (round 3: 10 bits for a
    converged cloud, 6 for a wide one).  A filter loop (wide cloud that converges over its steps, so that the bucket count
    changes on the way) in three child processes: buckets sized by the cloud (default), fixed at 10 bits
    (SLAM_PL_ORDER_ADAPT=0), and an explicit place-first order with 5 heading bits.  Weights of every step, selected indices and
    final poses must be the same bits in all three, and the first step's weights equal the oracle's on a subset."""
    import subprocess
    import sys

    code = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import _oracle as orc, slam_amd as sa
free = orc.load_map("floor_plan"); g = (free * 255).astype(np.uint8); rows, cols = g.shape
N, B, off = 30000, 120, [0.0, 30.0, 0.0]
f = sa.Filter(N, rows, cols, seed=4242)
f.set_map(g)
f.seed_gaussian([520.0, 280.0, 0.4], [25.0, 25.0, 0.2])
out = {}
pose = [520.0, 280.0, 0.4]
for step in range(6):
    ang, dist = orc.make_scan(free, pose, off, B)
    if step == 0:
        out["x0"], out["y0"], out["th0"] = f.poses()
        f.update_weights(ang, dist, 5.0, 500.0, off)
        out["w_first"] = f.weights()
    f.predict([2.5, 0.02, 0.02], [5e-4, 5e-4, 1e-2, 1e-2])
    f.update(ang, dist, 5.0, 500.0, off)
    out[f"w{step}"] = f.weights(); out[f"idx{step}"] = f.last_indices()
out["x"], out["y"], out["th"] = f.poses()
out["ang"], out["dist"] = orc.make_scan(free, [520.0, 280.0, 0.4], off, B)
f.close()
np.savez(sys.argv[3], **out)
"""
    root = os.path.dirname(HERE)
    res = {}
    for tag, env in (("adaptive", {}), ("fixed", {"SLAM_PL_ORDER_ADAPT": "0"}), ("place_first", {"SLAM_PL_ORDER": str(4 | (5 << 4) | (5 << 8))})):
        path = str(tmp_path / f"{tag}.npz")
        subprocess.run([sys.executable, "-c", code, root, HERE, path], check=True, env=dict(os.environ, **env), timeout=300)
        res[tag] = np.load(path)
    for k in res["fixed"].files:
        for tag in ("adaptive", "place_first"):
            a, b = res[tag][k], res["fixed"][k]
            if a.dtype.kind == "f":
                assert_bits(a, b, f"{tag} {k}")
            else:
                assert np.array_equal(a, b), f"{tag} {k}"
    r = res["adaptive"]
    g = (orc.load_map("floor_plan") * 255).astype(np.uint8)
    sub = np.arange(0, r["x0"].size, 37)
    ow, _ = orc.update_weights(r["x0"][sub], r["y0"][sub], r["th0"][sub], g, r["ang"], r["dist"], 5.0, 500.0, OFFSET, math=orc.DET, n_threads=8)
    assert_bits(r["w_first"][sub], ow, "first weighting pass vs oracle")


def test_one_chebyshev_plane_in_place_of_the_cone_planes(sa, tmp_path):
    """Maps too large for eight planes below 4 GiB get ONE plane of Chebyshev clearance (plane stride 0).  The choice is
    made when the map is set; SLAM_CONE_PLANES=0 forces it in a child process, whose weights and hits must equal the ones
    computed here with the cone planes, in both kernel shapes."""
    import subprocess
    import sys

    code = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import _oracle as orc, slam_amd as sa
rng = np.random.default_rng(303)
g = (orc.load_map("floor_plan") * 255).astype(np.uint8)
rows, cols = g.shape
N, B = 4500, 40
x, y, th = rng.uniform(-20, cols + 20, N), rng.uniform(-20, rows + 20, N), rng.uniform(-4, 4, N)
ang = np.linspace(-np.pi, np.pi, B, endpoint=False); dist = rng.uniform(3.0, 480.0, B)
out = {}
for name, rc in (("pl", sa.capi.RAYCAST_FAST_PARTICLE_LANES), ("bl", sa.capi.RAYCAST_FAST_BEAM_LANES)):
    f = sa.Filter(N, rows, cols, raycaster=rc); f.set_map(g); f.set_poses(x, y, th); f.record_hits(True)
    f.update_weights(ang, dist, 5.0, 500.0, [0.0, 30.0, 0.0])
    hxy, hc = f.hits()
    out[name + "_w"], out[name + "_xy"], out[name + "_c"] = f.weights(), hxy, hc
    f.close()
np.savez(sys.argv[3], **out)
"""
    root = os.path.dirname(HERE)
    res = {}
    for planes in ("1", "0"):
        path = str(tmp_path / f"planes{planes}.npz")
        env = dict(os.environ, SLAM_CONE_PLANES=planes)
        subprocess.run([sys.executable, "-c", code, root, HERE, path], check=True, env=env, timeout=300)
        res[planes] = np.load(path)
    for k in res["1"].files:
        a, b = res["1"][k], res["0"][k]
        if a.dtype == np.float64:
            assert_bits(a, b, k)
        else:
            assert np.array_equal(a, b), k
    assert (res["1"]["pl_c"] >= 0).mean() > 0.5


# ---- K2b SLAM mode: raycast + likelihood + occupancy update ---------------------------------------
S_SHAPES = {  # kernel shape -> (raycaster, canvas storage)
    "wave": ("RAYCAST_FAST_BEAM_LANES", "MAPS_DENSE"), "thread": ("RAYCAST_FAST_PARTICLE_LANES", "MAPS_DENSE"),
    "tiled": ("RAYCAST_FAST_BEAM_LANES", "MAPS_TILED"), "two_phase": ("RAYCAST_TWO_PHASE", "MAPS_DENSE"),
    "two_phase_tiled": ("RAYCAST_TWO_PHASE", "MAPS_TILED"), "literal": ("RAYCAST_TWO_PHASE_LITERAL", "MAPS_DENSE"),
    "literal_tiled": ("RAYCAST_TWO_PHASE_LITERAL", "MAPS_TILED"),
}


@pytest.mark.parametrize("shape", list(S_SHAPES))
@pytest.mark.parametrize("B,zmax", [(90, 500.0), (130, 61.5)])
def test_weights_slam_mode_maps_bit_exact(sa, shape, B, zmax):
    rng = np.random.default_rng(21)
    N, rows, cols = 48, 300, 400
    free = orc.load_map("map")
    ang, dist = orc.make_scan(free, [400.0, 400.0, np.pi / 90], OFFSET, B, max_dist=zmax)
    f = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, raycaster=getattr(sa.capi, S_SHAPES[shape][0]),
                  map_storage=getattr(sa.capi, S_SHAPES[shape][1]))
    x, y, th = cloud(rng, N, cols / 2, rows / 2, sxy=6.0, sth=0.3, th0=np.pi / 2)
    x[:3], y[:3] = [2.5, cols - 1.5, -20.0], [rows - 2.5, 1.5, 100.0]   # at the canvas border / outside
    x[3:6], y[3:6] = [3e9, -1e15, 150.0], [10.0, 20.0, -(2.0 ** 40)]   # far outside, beyond int range
    f.set_poses(x, y, th)
    maps = np.full((N, rows, cols), 128, np.uint8)
    f.record_hits(True)
    for it in range(3):  # repeated scans: later passes see earlier writes (hits appear from pass 2 on)
        f.update_weights(ang, dist, 5.0, zmax, OFFSET)
        ow, oxy, ocell, ost = orc.update_weights(x, y, th, maps, ang, dist, 5.0, zmax, OFFSET, slam_mode=True,
                                                 want_hits=True, math=orc.DET)
        hxy, hcell = f.hits()
        assert np.array_equal(hcell, ocell)
        assert_bits(hxy, oxy, "hit positions")
        assert_bits(f.weights(), ow, "weights")
        assert f.stats() == ost
        for p in range(N):
            assert np.array_equal(f.map_of(p), maps[p]), f"map of particle {p} differs after pass {it}"
        if it:
            assert (ocell >= 0).any()
    f.close()


# ---- K3/K4 resampling -------------------------------------------------------------------------------
def check_resample_against_the_oracle(sa, order, N):
    """slam/mcl.cpp:157-203,232-239: running sums (bits), selected indices, sort permutation and the gathered state
    of ONE resample of N weights against the oracle, in the given summation order."""
    rng = np.random.default_rng(N)
    w = rng.uniform(0, 1, N) ** 6
    w[rng.uniform(size=N) < 0.15] = 0.0
    if N > 10:
        w[:5] = 0.0  # leading zeros: the equality branch of slam/mcl.cpp:187-191 is reachable with cutoff 0
    u = orc.reference_uniforms(N, tick=N + 17)  # libstdc++ canonical uniforms
    u[N // 2] = 0.0
    f = sa.Filter(N, 16, 16, noise=sa.capi.NOISE_INJECTED,
                  prefix_order=sa.capi.PREFIX_BLOCKED if order == "blocked" else sa.capi.PREFIX_REFERENCE)
    x, y, th = rng.uniform(0, 16, N), rng.uniform(0, 16, N), rng.uniform(-1, 1, N)
    f.set_poses(x, y, th)
    f.set_weights(w)
    f.set_noise(canonical_uniforms=u)
    f.resample()
    oidx, oP, osum = orc.resample(w, u, orc.ORDER_BLOCKED if order == "blocked" else orc.ORDER_REFERENCE)
    P, s = f.prefix()
    assert_bits(P, oP, "running sums")
    assert bits([s])[0] == bits([osum])[0]
    idx = f.last_indices()
    assert np.array_equal(idx, oidx)
    # the other order selects the same particles (a draw would have to fall between two roundings)
    other, _, _ = orc.resample(w, u, orc.ORDER_REFERENCE if order == "blocked" else orc.ORDER_BLOCKED)
    if N <= 1_000_000:
        assert np.array_equal(idx, other)
    else:
        # the two running sums differ by ~1e-16 sqrt(N) relative at every one of N boundaries, so of N draws about
        # N^2 x 1e-13 fall between two roundings: dozens at 20 M.  Each order is bit-exact against ITS oracle above.
        assert np.count_nonzero(idx != other) <= N // 100_000
    # sort: (weight desc, slot asc); state = old[idx[perm]]
    perm = f.last_order()
    assert np.array_equal(perm, orc.sort_order(w[oidx], 0))
    src = oidx[perm]
    px, py, pth = f.poses()
    assert_bits(px, x[src], "x"), assert_bits(py, y[src], "y"), assert_bits(pth, th[src], "theta")
    nw = f.weights()
    assert_bits(nw, w[src], "weights")
    assert np.all(np.diff(nw) <= 0)  # slam/mcl.cpp:236-238 invariant
    ref_perm = orc.sort_order(w[oidx], 1)  # the reference's unstable std::sort: same multiset, same front
    assert np.array_equal(w[oidx][ref_perm], nw)
    f.close()
    return np.unique(oidx).size


@pytest.mark.parametrize("order", ["blocked", "reference"])
@pytest.mark.parametrize("N", [1, 2, 64, 65, 4096, 4097, 100000])
def test_resample_indices_bit_exact(sa, order, N):
    check_resample_against_the_oracle(sa, order, N)


@pytest.mark.parametrize("order,N", [("blocked", 262_145), ("reference", 262_145), ("blocked", 1_000_000), ("reference", 1_000_000),
                                     ("blocked", 20_000_000)])
def test_resample_indices_bit_exact_beyond_the_sample_sort(sa, order, N):
    """The sizes north_star promises bit-exact indices at: 4 levels of the radix-64 running sum (262 145 .. 16.7 M
    particles) and 5 (20 M: more than 2^24 slots, 25 bisection steps), the sort path taken above 262 144 particles and
    SLAM_SORT_FULL there.  The strictly sequential order (one thread) is compared up to 1 M."""
    distinct = check_resample_against_the_oracle(sa, order, N)
    assert distinct > N // 4


@pytest.mark.parametrize("N", [1000, 4096, 4097, 16384, 50001, 100000, 262144, 262145, 600000])
def test_weight_sort_of_the_sample_sort_sizes(sa, N):
    """slam/mcl.cpp:232-239 on the library's own sorts (sortk.h): one workgroup up to 4096 particles, the sample sort up to
    262 144, the radix sort beyond (no library sort is left in the step).  The order must be
    (weight descending, slot ascending) whatever the weights look like: heavy ties (few distinct weights, so that most
    buckets' splitters are decided by the slot), one surviving particle (every key equal), already sorted and reversed
    inputs, a tiny dynamic range, and one value that owns half of the array (its bucket does not fit LDS if the
    samples were unlucky; here it must simply be right)."""
    rng = np.random.default_rng(N)
    cases = {
        "generic": rng.uniform(0, 1, N) ** 8,
        "three values": rng.choice([1e-300, 0.25, 0.5], N),
        "one survivor": np.where(np.arange(N) == N // 3, 1.0, 0.0),
        "ascending": np.linspace(1e-9, 1.0, N),
        "descending": np.linspace(1.0, 1e-9, N),
        "narrow": 1.0 + rng.integers(0, 50, N) * 2.0 ** -52,
        "half equal": np.where(rng.uniform(size=N) < 0.5, 0.125, rng.uniform(0, 1, N)),
    }
    f = sa.Filter(N, 16, 16, noise=sa.capi.NOISE_INJECTED)
    x = rng.uniform(0, 16, N)
    for name, w in cases.items():
        u = orc.reference_uniforms(N, tick=len(name))
        f.set_poses(x, x, x)
        f.set_weights(w)
        f.set_noise(canonical_uniforms=u)
        f.resample()
        idx = f.last_indices()
        keys = w[idx]
        want = np.argsort(-keys, kind="stable")
        perm = f.last_order()
        assert np.array_equal(perm, want), name
        assert_bits(f.weights(), keys[want], name)
    f.close()


def test_sample_sort_equals_rocprim_and_its_path_from_memory(sa, tmp_path):
    """The same resample + sort in three child processes: sample sort (default), sample sort with every group of buckets
    ranked from memory instead of LDS (SLAM_SS_LDS_CAP=16: the path a skewed sample would take), rocPRIM
    (SLAM_SORT_IMPL=0).  All three must give the same permutation and the same sorted state."""
    import subprocess
    import sys

    code = """
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import _oracle as orc, slam_amd as sa
out = {}
for N in (1000, 5000, 20000, 100000, 300000):
    rng = np.random.default_rng(N)
    f = sa.Filter(N, 16, 16, noise=sa.capi.NOISE_INJECTED)
    for name, w in (("generic", rng.uniform(0, 1, N) ** 8), ("ties", rng.choice([1e-300, 0.25, 0.5], N))):
        x = rng.uniform(0, 16, N)
        f.set_poses(x, x, x); f.set_weights(w); f.set_noise(canonical_uniforms=orc.reference_uniforms(N, tick=7)); f.resample()
        out[f"{N}_{name}_perm"] = f.last_order(); out[f"{N}_{name}_w"] = f.weights()
    f.close()
np.savez(sys.argv[3], **out)
"""
    root = os.path.dirname(HERE)
    res = {}
    for tag, env in (("sample", {}), ("memory", {"SLAM_SS_LDS_CAP": "16"}), ("rocprim", {"SLAM_SORT_IMPL": "0"})):
        path = str(tmp_path / f"{tag}.npz")
        subprocess.run([sys.executable, "-c", code, root, HERE, path], check=True, env=dict(os.environ, **env), timeout=300)
        res[tag] = np.load(path)
    for k in res["rocprim"].files:
        for tag in ("sample", "memory"):
            if k.endswith("_w"):
                assert_bits(res[tag][k], res["rocprim"][k], f"{tag} {k}")
            else:
                assert np.array_equal(res[tag][k], res["rocprim"][k]), f"{tag} {k}"


def test_resample_all_zero_and_single_survivor(sa):
    for N in (9, 1000):
        f = sa.Filter(N, 16, 16, noise=sa.capi.NOISE_INJECTED)
        f.set_weights(np.zeros(N))
        f.set_noise(canonical_uniforms=orc.reference_uniforms(N, 3))
        f.resample()
        assert np.all(f.last_indices() == (N - 1) // 2)
        w = np.zeros(N)
        w[N // 3] = 1e-300
        f.set_weights(w)
        u = orc.reference_uniforms(N, 4)
        f.set_noise(canonical_uniforms=u)
        f.resample()
        assert np.array_equal(f.last_indices(), orc.resample(w, u, orc.ORDER_BLOCKED)[0])
        f.close()


def test_systematic_resampler(sa):
    N = 5000
    rng = np.random.default_rng(3)
    w = rng.uniform(0, 1, N)
    f = sa.Filter(N, 16, 16, noise=sa.capi.NOISE_INJECTED, resampler=sa.capi.RESAMPLE_SYSTEMATIC, sort=sa.capi.SORT_NONE)
    f.set_weights(w)
    u = np.full(N, 0.37)
    f.set_noise(canonical_uniforms=u)
    f.resample()
    assert np.array_equal(f.last_indices(), orc.resample(w, u, orc.ORDER_BLOCKED, resampler=1)[0])
    f.close()


# ---- whole steps ------------------------------------------------------------------------------------
def run_oracle_step(state, maps, g, ang, dist, stddev, zmax, z, u, slam, order):
    x, y, th = orc.predict(state[0], state[1], state[2], ODOM, ALPHAS, z, math=orc.DET)
    if slam:
        w, _ = orc.update_weights(x, y, th, maps, ang, dist, stddev, zmax, OFFSET, slam_mode=True, math=orc.DET)
    else:
        w, _ = orc.update_weights(x, y, th, g, ang, dist, stddev, zmax, OFFSET, math=orc.DET)
    idx, _, _ = orc.resample(w, u, order)
    perm = orc.sort_order(w[idx], 0)
    src = idx[perm]
    if slam:
        maps[:] = maps[src]
    return (x[src], y[src], th[src], w[src]), idx


def test_slam_loop_like_the_reference_benchmark(sa):
    """benchmark/mcl.cpp:25-64 at reduced canvas: 25 particles, 90 beams over [0, pi), the same scan re-applied."""
    N, B, rows, cols = 25, 90, 450, 800
    free = orc.load_map("floor_plan")
    sensor = orc.sensor_position([400.0, 400.0, np.pi], OFFSET)
    hits = orc.fake_lidar_scan(free.astype(np.int32), sensor, 0.0, np.pi, 500.0, B)
    ang, dist = orc.hits_to_scan(hits, sensor, 0.0, np.pi, 500.0)
    mcl = sa.MCL(N, (cols, rows), noise=sa.capi.NOISE_INJECTED)
    f = mcl.filter
    sp = mcl.starting_pose()
    assert (sp.x, sp.y, sp.theta) == (cols / 2.0, rows / 2.0, np.pi / 2)
    state = (np.full(N, sp.x), np.full(N, sp.y), np.full(N, sp.theta), np.full(N, 1.0 / N))
    assert_bits(f.weights(), state[3], "initial weights")
    maps = np.full((N, rows, cols), 128, np.uint8)
    for step in range(6):
        z = orc.reference_normals(3 * N, tick0=50 * step + 1, tick_stride=1)
        u = orc.reference_uniforms(N, tick=900 + step)
        f.set_noise(z, u)
        mcl.predict(sa.Odometry(*ODOM), ALPHAS)
        mcl.update(list(zip(ang, dist)), 5.0, 500.0, sa.Pose(*OFFSET))
        state, idx = run_oracle_step(state, maps, None, ang, dist, 5.0, 500.0, z, u, True, orc.ORDER_BLOCKED)
        assert np.array_equal(f.last_indices(), idx), f"step {step}"
        px, py, pth = f.poses()
        assert_bits(px, state[0], "x"), assert_bits(py, state[1], "y"), assert_bits(pth, state[2], "theta")
        assert_bits(f.weights(), state[3], "weights")
        for p in range(N):
            assert np.array_equal(f.map_of(p), maps[p]), f"step {step} map {p}"
    parts = mcl.get_particles()
    assert parts[0].weight == max(p.weight for p in parts) and np.array_equal(parts[0].map, maps[0])
    f.close()


# ---- tiled copy-on-write canvases: same results as the dense storage -----------
def test_tiled_canvases_match_dense_and_oracle(sa):
    """The grid-SLAM loop with canvases held as tables of 64x64 copy-on-write tiles: poses, weights,
    selection and every canvas equal the oracle's (hence the dense storage's) at every step; the pool
    holds far fewer tiles than dense canvases would, because tiles out of sensor range stay shared."""
    rng = np.random.default_rng(61)
    N, B, rows, cols, zmax, seed = 96, 60, 450, 700, 120.0, 0x711E      # canvas not a multiple of 64
    free = orc.load_map("map")
    ang, dist = orc.make_scan(free, [400.0, 400.0, np.pi / 90], OFFSET, B, max_dist=zmax)
    x, y, th = cloud(rng, N, cols / 2, rows / 2, sxy=4.0, sth=0.2, th0=np.pi / 2)
    odom = [0.1, 3.0, 0.05]
    T = ((rows + 63) // 64) * ((cols + 63) // 64)
    f = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=seed, map_storage=sa.capi.MAPS_TILED)
    f.set_poses(x, y, th)
    assert f.tile_stats() == {"pool": (3 * N * T + 1) // 2, "live": 0, "allocated_since": 0}
    assert np.all(f.map_of(5) == 128)
    state, maps = (x, y, th), np.full((N, rows, cols), 128, np.uint8)
    for step in range(5):
        f.predict(odom, ALPHAS)
        f.update(ang, dist, 5.0, zmax, OFFSET)
        ox, oy, oth = orc.predict(*state, odom, ALPHAS, orc.philox_normals(seed, step, 0, N), math=orc.DET)
        ow, _ = orc.update_weights(ox, oy, oth, maps, ang, dist, 5.0, zmax, OFFSET, slam_mode=True, math=orc.DET)
        idx, _, _ = orc.resample(ow, orc.philox_uniforms(seed, step, 0, N), orc.ORDER_BLOCKED)
        perm = orc.sort_order(ow[idx], 0)
        src = idx[perm]
        state, maps = (ox[src], oy[src], oth[src]), maps[src].copy()
        assert np.array_equal(f.last_indices(), idx), f"step {step}"
        for q in range(3):
            assert_bits(f.poses()[q], state[q], f"pose column {q}, step {step}")
        assert_bits(f.weights(), ow[src], "weights")
        for p in range(0, N, 5):
            assert np.array_equal(f.map_of(p), maps[p]), f"canvas {p}, step {step}"
        st = f.tile_stats()
        assert 0 < st["live"] < N * T / 2, st      # only the tiles within range of the sensor were ever written
    # repeated weighting without a resample: owned tiles are written in place, shared ones copied once
    before = f.tile_stats()
    for it in range(2):
        f.update_weights(ang, dist, 5.0, zmax, OFFSET)
        ow, _ = orc.update_weights(*state, maps, ang, dist, 5.0, zmax, OFFSET, slam_mode=True, math=orc.DET)
        assert_bits(f.weights(), ow, f"repeat {it}")
        if it == 0:
            first = f.tile_stats()["allocated_since"]
    assert f.tile_stats()["allocated_since"] == first > 0 and before["allocated_since"] == 0
    for p in range(0, N, 7):
        assert np.array_equal(f.map_of(p), maps[p])
    # set_maps / map_of round trip through the tiles
    custom = rng.integers(0, 256, size=(3, rows, cols), dtype=np.uint8)
    f.set_maps(custom, first=10)
    for k in range(3):
        assert np.array_equal(f.map_of(10 + k), custom[k])
    assert np.array_equal(f.map_of(9), maps[9]) and np.array_equal(f.map_of(13), maps[13])
    f.close()


def test_tiled_canvases_reuse_collected_tiles(sa):
    """A pool much smaller than the dense footprint, refilled by the collection at every resample:
    twelve steps of a moving robot, every canvas equal to the dense handle's at every step."""
    rng = np.random.default_rng(62)
    N, B, rows, cols, zmax, seed = 512, 45, 640, 640, 150.0, 0x711F
    free = orc.load_map("map")
    T = ((rows + 63) // 64) * ((cols + 63) // 64)
    dense = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=seed, map_storage=sa.capi.MAPS_DENSE)
    tiled = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=seed, map_storage=sa.capi.MAPS_TILED,
                      tile_pool_tiles=int(0.75 * N * T))
    x, y, th = cloud(rng, N, 200.0, 200.0, sxy=2.0, sth=0.1, th0=0.6)
    for f in (dense, tiled):
        f.set_poses(x, y, th)
    pose = np.array([400.0, 400.0, np.pi / 90])
    peak = 0
    for step in range(12):
        pose[:2] += 6.0 * np.array([np.cos(0.6), np.sin(0.6)])
        ang, dist = orc.make_scan(free, pose, OFFSET, B, max_dist=zmax)
        for f in (dense, tiled):
            f.predict([0.0, 6.0, 0.0], ALPHAS)
            f.update(ang, dist, 5.0, zmax, OFFSET)
        assert np.array_equal(dense.last_indices(), tiled.last_indices())
        assert_bits(tiled.weights(), dense.weights(), f"weights, step {step}")
        for p in rng.choice(N, 6, replace=False):
            assert np.array_equal(tiled.map_of(int(p)), dense.map_of(int(p))), f"canvas {p}, step {step}"
        peak = max(peak, tiled.tile_stats()["live"])
    assert 0 < peak < 0.75 * N * T
    dense.close(), tiled.close()


def test_tiled_canvases_larger_than_the_lds_table(sa):
    """Canvases of more than 3072 tiles: the tile table no longer fits the 48 KiB of LDS the walk stages
    it in and is read from memory instead.  Same weights, selection and canvases as dense storage."""
    rng = np.random.default_rng(77)
    N, B, rows, cols, zmax, seed = 6, 40, 3600, 3700, 300.0, 0x7AB1E
    free = orc.load_map("map")
    ang, dist = orc.make_scan(free, [400.0, 400.0, np.pi / 90], OFFSET, B, max_dist=zmax)
    assert ((rows + 63) // 64) * ((cols + 63) // 64) > 3072
    x, y, th = cloud(rng, N, cols / 2, rows / 2, sxy=30.0, sth=0.4, th0=np.pi / 2)
    x[0], y[0] = 10.0, rows - 5.0   # a corner of the canvas
    fs = [sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=seed, map_storage=st)
          for st in (sa.capi.MAPS_DENSE, sa.capi.MAPS_TILED)]
    for f in fs:
        f.set_poses(x, y, th)
    for step in range(3):
        for f in fs:
            f.predict([0.05, 4.0, -0.02], ALPHAS)
            f.update(ang, dist, 5.0, zmax, OFFSET)
        assert np.array_equal(fs[0].last_indices(), fs[1].last_indices())
        assert_bits(fs[1].weights(), fs[0].weights(), f"weights, step {step}")
        for p in range(N):
            assert np.array_equal(fs[1].map_of(p), fs[0].map_of(p)), f"canvas {p}, step {step}"
    assert (fs[0].map_of(0) != 128).any()
    for f in fs:
        f.close()


def test_tiled_pool_exhaustion_is_reported(sa):
    N, B, rows, cols = 32, 90, 400, 400
    free = orc.load_map("map")
    ang, dist = orc.make_scan(free, [400.0, 400.0, np.pi / 90], OFFSET, B)
    f = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, map_storage=sa.capi.MAPS_TILED, tile_pool_tiles=40)
    with pytest.raises(RuntimeError, match="tile pool exhausted"):
        f.update(ang, dist, 5.0, 500.0, OFFSET)
    f.close()


@pytest.mark.parametrize("storage", ["dense", "tiled"])
def test_cell_counters_can_be_switched_off(sa, storage):
    """slam_mcl_collect_stats(0): the two-phase kernel is compiled without its cell counters; weights, selection and
    canvases are what they are with the counters, and stats() fails loudly instead of returning stale numbers."""
    st = {"dense": sa.capi.MAPS_DENSE, "tiled": sa.capi.MAPS_TILED}[storage]
    N, B, rows, cols = 200, 90, 500, 600
    free = orc.load_map("map")
    on = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=21, map_storage=st)
    off = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=21, map_storage=st)
    off.collect_stats(False)
    pose = [300.0, 250.0, 0.3]
    for step in range(3):
        pose = [pose[0] + 2.5 * np.cos(pose[2]), pose[1] + 2.5 * np.sin(pose[2]), pose[2]]
        ang, dist = orc.make_scan(free, pose, OFFSET, B)
        for f in (on, off):
            f.predict([0.0, 2.5, 0.0], [1e-3] * 4)
            f.update(ang, dist, 5.0, 500.0, OFFSET)
        assert_bits(on.weights(), off.weights(), f"step {step}: weights")
        assert np.array_equal(on.last_indices(), off.last_indices())
        for p in (0, N // 2, N - 1):
            assert np.array_equal(on.map_of(p), off.map_of(p))
    assert on.stats()["cells_read"] > 0
    with pytest.raises(RuntimeError, match="switched off"):
        off.stats()
    off.collect_stats(True)
    off.predict([0.0, 2.5, 0.0], [1e-3] * 4), on.predict([0.0, 2.5, 0.0], [1e-3] * 4)
    off.update_weights(ang, dist, 5.0, 500.0, OFFSET), on.update_weights(ang, dist, 5.0, 500.0, OFFSET)
    assert off.stats() == on.stats()
    on.close(), off.close()


@pytest.mark.parametrize("storage", ["dense", "tiled"])
def test_device_index_arrays_are_range_checked(sa, storage):
    """local_idx_dev / pos_dev live on the device: an entry out of range must not reach memory; it is
    skipped and reported by the call (synchronous handles) as an error."""
    import torch

    from slam_amd.dist import HipShard

    st = {"dense": sa.capi.MAPS_DENSE, "tiled": sa.capi.MAPS_TILED}[storage]
    n, rows, cols = 8, 70, 90
    b = HipShard(n, rows, cols, 0, 1, 0, 5, mode=sa.capi.MODE_SLAM, map_storage=st)
    good = b.pack_maps(torch.arange(n, device="cuda"))
    assert good.shape[0] == n and int(good[:, :rows * cols].min()) == 128
    with pytest.raises(RuntimeError, match="out of range"):
        b.pack_maps(torch.tensor([0, n, 1], device="cuda"))
    with pytest.raises(RuntimeError, match="out of range"):
        b.pack_maps(torch.tensor([-1], device="cuda"))
    b.pack_maps(torch.tensor([n - 1], device="cuda"))   # the flag was cleared by the report
    if storage == "tiled":
        with pytest.raises(RuntimeError, match="out of range"):
            b.pack_tiles(torch.tensor([n + 3], device="cuda"))
    # the import side: a slot that points past the received pool
    ang, dist = np.linspace(-3, 3, 12), np.full(12, 20.0)
    b.update_weights(ang, dist, 5.0, 50.0, OFFSET)
    b.shard_select(b.packed_state().clone(), 1)
    pos = torch.zeros(n, dtype=torch.int64, device="cuda")
    pos[3] = 2
    with pytest.raises(RuntimeError, match="out of range"):
        b.clone_maps(good[:2].contiguous(), pos)
    b.f.close()


@pytest.mark.parametrize("order", ["blocked", "reference"])
def test_localization_loop(sa, order):
    rng = np.random.default_rng(99)
    N, B = 3000, 360
    free = orc.load_map("floor_plan")
    g = (free * 255).astype(np.uint8)
    rows, cols = g.shape
    true = [520.0, 280.0, 0.4]
    ang, dist = orc.make_scan(free, true, OFFSET, B)
    x, y, th = cloud(rng, N, true[0], true[1], th0=true[2])
    o = orc.ORDER_BLOCKED if order == "blocked" else orc.ORDER_REFERENCE
    f = sa.Filter(N, rows, cols, noise=sa.capi.NOISE_INJECTED,
                  prefix_order=sa.capi.PREFIX_BLOCKED if order == "blocked" else sa.capi.PREFIX_REFERENCE)
    f.set_map(g)
    f.set_poses(x, y, th)
    state = (x, y, th, None)
    for step in range(5):
        z = orc.reference_normals(3 * N, tick0=7 * step + 1, tick_stride=3)
        u = orc.reference_uniforms(N, tick=step + 1)
        f.set_noise(z, u)
        f.predict(ODOM, ALPHAS)
        f.update(ang, dist, 5.0, 500.0, OFFSET)
        state, idx = run_oracle_step(state, None, g, ang, dist, 5.0, 500.0, z, u, False, o)
        assert np.array_equal(f.last_indices(), idx), f"step {step}"
        px, py, pth = f.poses()
        assert_bits(px, state[0], "x"), assert_bits(py, state[1], "y"), assert_bits(pth, state[2], "theta")
        assert_bits(f.weights(), state[3], "weights")
    ap = f.average_pose()
    oa = orc.average_pose(*state[:3])
    assert np.allclose([ap.x, ap.y, ap.theta], oa, rtol=REL_TOL)
    f.close()


def test_philox_loop_is_deterministic_and_matches_host_streams(sa):
    rng = np.random.default_rng(4)
    N, B, seed = 2048, 90, 0x5EED0001
    free = orc.load_map("map")
    g = (free * 255).astype(np.uint8)
    true = [400.0, 400.0, 0.1]
    ang, dist = orc.make_scan(free, true, OFFSET, B)
    x, y, th = cloud(rng, N, true[0], true[1], th0=true[2])
    f = sa.Filter(N, *g.shape, seed=seed)
    f.set_map(g)
    f.set_poses(x, y, th)
    state = (x, y, th, None)
    for step in range(3):
        f.predict(ODOM, ALPHAS)
        f.update(ang, dist, 5.0, 500.0, OFFSET)
        z = orc.philox_normals(seed, step, 0, N)
        u = orc.philox_uniforms(seed, step, 0, N)
        state, idx = run_oracle_step(state, None, g, ang, dist, 5.0, 500.0, z, u, False, orc.ORDER_BLOCKED)
        assert np.array_equal(f.last_indices(), idx)
        assert_bits(f.poses()[0], state[0], "x")
    f.close()


# ---- neighbours of the path: scan synthesis on the device ------------------------------------------
def test_fake_lidar_scan_device(sa):
    from slam_amd import capi

    lib = capi.load_library()
    free = orc.load_map("map").astype(np.uint8)
    rows, cols = free.shape
    rng = np.random.default_rng(8)
    poses = np.stack([rng.uniform(50, 950, 20), rng.uniform(50, 950, 20), rng.uniform(-3, 3, 20)], 1).copy()
    hits = np.empty((20, 90, 2))
    capi.check(lib.slam_fake_lidar_scan(0, capi.ptr(free), rows, cols, capi.ptr(poses), 20, 0.0, 2 * np.pi, 500.0, 90,
                                        capi.ptr(hits)))
    for p in range(20):
        ref = orc.fake_lidar_scan(free.astype(np.int32), poses[p], 0.0, 2 * np.pi, 500.0, 90, math=orc.DET)
        assert_bits(hits[p], ref, f"scan {p}")
    g = (free * 255).astype(np.uint8)
    out = np.empty((20, 2))
    capi.check(lib.slam_raycast_u8(0, capi.ptr(g), rows, cols, capi.ptr(poses), 20, 300.0 ** 2, 0.5, capi.ptr(out)))
    for p in range(20):
        assert_bits(out[p], orc.raycast(1, g, poses[p, 0], poses[p, 1], poses[p, 2], 300.0 ** 2, 0.5, math=orc.DET), "raycast")


def test_resident_map_handle(sa):
    """slam_map_*: the grid uploaded once, then scans / raycasts / raycast_mapping on it (slam/fake_lidar.cpp:15-33,
    slam/raycast.cpp:7-223): the same results as the one-shot entry points and as the oracle."""
    free = orc.load_map("floor_plan").astype(np.uint8)
    rng = np.random.default_rng(18)
    poses = np.stack([rng.uniform(30, 1250, 12), rng.uniform(30, 560, 12), rng.uniform(-3, 3, 12)], 1).copy()
    gm = sa.GridMap(free)
    for rep in range(3):  # repeated calls on one handle, growing and shrinking batches
        n = (12, 3, 7)[rep]
        hits = gm.fake_lidar_scan(poses[:n], 0.0, 2 * np.pi, 500.0, (90, 360, 17)[rep])
        for p in range(n):
            assert_bits(hits[p], orc.fake_lidar_scan(free.astype(np.int32), poses[p], 0.0, 2 * np.pi, 500.0, hits.shape[1], math=orc.DET), "scan")
    out = gm.raycast(2, poses, 200.0 ** 2, 0.5)
    for p in range(12):
        assert_bits(out[p], orc.raycast(2, free.astype(np.int32), poses[p, 0], poses[p, 1], poses[p, 2], 200.0 ** 2, 0.5, math=orc.DET), "raycast<int>")
    g = (free * 255).astype(np.uint8)
    gm.upload(g)
    out = gm.raycast(1, poses, 300.0 ** 2, 0.25)
    for p in range(12):
        assert_bits(out[p], orc.raycast(1, g, poses[p, 0], poses[p, 1], poses[p, 2], 300.0 ** 2, 0.25, math=orc.DET), "raycast<uchar>")
    gm.close()
    # raycast_mapping on ONE canvas, rays in order (later rays meet earlier rays' walls), two step sizes
    for step in (1.0, 0.5):
        canvas = np.full((200, 260), 128, np.uint8)
        gm = sa.GridMap(canvas)
        ref = canvas.copy()
        n = 150
        rays = np.stack([130 + 5 * rng.standard_normal(n), 100 + 5 * rng.standard_normal(n), rng.uniform(-np.pi, np.pi, n)], 1).copy()
        rays[:2] = [[-5.0, 100.0, 0.1], [259.5, 199.5, 2.0]]   # from outside the canvas / from its corner
        z = rng.uniform(0, 90, n)
        z[::7] = 90.0   # z == zmax: no occupied update (slam/raycast.cpp:205)
        for rep in range(2):
            hits = gm.raycast_mapping(rays, z * z, 90.0 ** 2, step)
            for k in range(n):
                want, _, _ = orc.raycast_mapping(ref, rays[k, 0], rays[k, 1], rays[k, 2], z[k] * z[k], 90.0 ** 2, step=step, math=orc.DET)
                assert_bits(hits[k], want, f"hit of ray {k}, repetition {rep}")
            assert np.array_equal(gm.download(), ref)
        assert (hits[:, 0] != -1).sum() > 20
        with pytest.raises(sa.SlamError):
            gm.raycast_mapping(rays[:1], [91.0 ** 2], 90.0 ** 2)   # the reference asserts z <= zmax (slam/raycast.cpp:145)
        gm.close()


def test_log_domain_weights_can_be_set_and_restored(sa):
    """A SLAM_WEIGHT_LOG handle holds sums of logs (<= 0) in its weight column: set_weights takes them back, and the
    selection then runs on exp(lw - max lw) as after a weighting pass; a linear handle still refuses negatives."""
    N = 3000
    rng = np.random.default_rng(5)
    lw = -rng.uniform(0, 900, N)   # far beyond what exp() can represent in the linear domain
    u = orc.reference_uniforms(N, 11)
    f = sa.Filter(N, 16, 16, noise=sa.capi.NOISE_INJECTED, sort=sa.capi.SORT_NONE, weight_domain=sa.capi.WEIGHT_LOG)
    f.set_weights(lw)
    assert_bits(f.weights(), lw, "log weights read back")
    f.set_noise(canonical_uniforms=u)
    f.resample()
    want, _, _ = orc.resample(orc.log_domain_weights(lw, math=orc.DET), u, orc.ORDER_BLOCKED)
    assert np.array_equal(f.last_indices(), want)
    with pytest.raises(sa.SlamError):
        f.set_weights(np.array([np.nan]))
    f.close()
    lin = sa.Filter(8, 16, 16)
    with pytest.raises(sa.SlamError):
        lin.set_weights(-np.ones(8))
    lin.close()


# ---- error behaviour ------------------------------------------------------------------------------------
def test_errors_are_reported_not_swallowed(sa):
    with pytest.raises(sa.SlamError):
        sa.Filter(0, 10, 10)
    f = sa.Filter(8, 10, 10)
    with pytest.raises(sa.SlamError):
        f.update_weights([0.0], [1.0], 5.0, 10.0)          # no map yet
    f.set_map(np.full((10, 10), 255, np.uint8))
    with pytest.raises(sa.SlamError):
        f.set_poses([np.nan] * 8, [0.0] * 8, [0.0] * 8)     # SURVEY.md §7 hard part 9: finite poses only
    with pytest.raises(sa.SlamError):
        f.update_weights([0.0], [1.0], 0.0, 10.0)           # stddev must be positive
    with pytest.raises(sa.SlamError):
        f.set_map(np.zeros((5, 5), np.uint8))               # size differs
    with pytest.raises(sa.SlamError):
        f.map_of(0)                                         # localization mode holds no per-particle maps
    f.close()


# ---- BASELINE.json config 2 at full size: size-independent properties ------------------------------
def test_full_size_config2_properties(sa):
    rng = np.random.default_rng(2)
    N, B = 100_000, 360
    free = orc.load_map("floor_plan")
    g = (free * 255).astype(np.uint8)
    rows, cols = g.shape
    true = [520.0, 280.0, 0.4]
    ang, dist = orc.make_scan(free, true, OFFSET, B)
    x, y, th = cloud(rng, N, true[0], true[1], th0=true[2])
    fast = sa.Filter(N, rows, cols, seed=1)
    walk = sa.Filter(N, rows, cols, seed=1, raycaster=sa.capi.RAYCAST_WALK)
    for f in (fast, walk):
        f.set_map(g)
        f.set_poses(x, y, th)
        f.predict(ODOM, ALPHAS)
        f.update_weights(ang, dist, 5.0, 500.0, OFFSET)
    wf, ww = fast.weights(), walk.weights()
    assert_bits(wf, ww, "clearance-skipping vs cell-by-cell walk")     # two independent device algorithms agree
    sub = rng.choice(N, 4096, replace=False)                           # oracle on a sample its threads finish in a second
    px, py, pth = fast.poses()
    ow, _ = orc.update_weights(px[sub], py[sub], pth[sub], g, ang, dist, 5.0, 500.0, OFFSET, math=orc.DET, n_threads=16)
    assert_bits(wf[sub], ow, "weights vs oracle sample")
    st = walk.stats()
    assert st["cells_written"] == 0 and 10 * N * B < st["cells_read"] <= 502 * N * B
    fast.update_weights(ang, dist, 5.0, 500.0, OFFSET)                 # idempotent: read-only map
    assert_bits(fast.weights(), wf, "idempotence")
    fast.resample()
    idx = fast.last_indices()
    P, s = fast.prefix()
    oidx, oP, _ = orc.resample(wf, orc.philox_uniforms(1, 0, 0, N), orc.ORDER_BLOCKED)
    assert_bits(P, oP, "running sums")
    assert np.array_equal(idx, oidx)
    nw = fast.weights()
    assert np.all(np.diff(nw) <= 0) and np.array_equal(np.sort(wf[idx])[::-1], nw)
    fast.close(), walk.close()


# ---- sharded resampling: any number of shards gives the single-handle result ------------------------
@pytest.mark.parametrize("shards,split", [(2, False), (4, False), (2, True), (4, True)])
def test_sharded_filter_matches_single_handle(sa, shards, split):
    """split: poses and weights gathered separately (slam_mcl_shard_resample_split)."""
    import torch

    from slam_amd.dist import HipShard, ShardedFilter

    rng = np.random.default_rng(31)
    N, B, seed = 4200, 90, 0xC0FFEE   # shards of 2100 / 1050 particles: the 64-chunks of the running sum straddle them
    free = orc.load_map("floor_plan")
    g = (free * 255).astype(np.uint8)
    rows, cols = g.shape
    true = [520.0, 280.0, 0.4]
    ang, dist = orc.make_scan(free, true, OFFSET, B)
    x, y, th = cloud(rng, N, true[0], true[1], th0=true[2])
    single = sa.Filter(N, rows, cols, seed=seed, sort=sa.capi.SORT_NONE)
    single.set_map(g)
    single.set_poses(x, y, th)
    n_loc = N // shards
    backs = [HipShard(n_loc, rows, cols, r, shards, 0, seed) for r in range(shards)]
    for r, b in enumerate(backs):
        b.set_map(g)
        b.f.set_poses(x[r * n_loc:(r + 1) * n_loc], y[r * n_loc:(r + 1) * n_loc], th[r * n_loc:(r + 1) * n_loc])
    state = (x, y, th)
    for step in range(3):
        single.predict(ODOM, ALPHAS)
        single.update(ang, dist, 5.0, 500.0, OFFSET)
        for b in backs:
            b.predict(ODOM, ALPHAS)
            b.update_weights(ang, dist, 5.0, 500.0, OFFSET)
        gathered = torch.cat([b.packed_state() for b in backs])      # what all_gather_into_tensor produces
        gp = torch.cat([b.packed_state()[:3 * n_loc] for b in backs])
        gw = torch.cat([b.packed_state()[3 * n_loc:] for b in backs])
        torch.cuda.synchronize()
        for b in backs:
            if split:
                b.shard_resample_split(gp, gw, shards)
            else:
                b.shard_resample(gathered, shards)
        # oracle, unsorted order
        ox, oy, oth = orc.predict(*state, ODOM, ALPHAS, orc.philox_normals(seed, step, 0, N), math=orc.DET)
        ow, _ = orc.update_weights(ox, oy, oth, g, ang, dist, 5.0, 500.0, OFFSET, math=orc.DET)
        idx, _, _ = orc.resample(ow, orc.philox_uniforms(seed, step, 0, N), orc.ORDER_BLOCKED)
        state = (ox[idx], oy[idx], oth[idx])
        assert np.array_equal(single.last_indices(), idx)
        assert np.array_equal(np.concatenate([b.f.last_indices() for b in backs]), idx)
        for q in range(3):
            assert_bits(single.poses()[q], state[q], "single handle")
            assert_bits(np.concatenate([b.f.poses()[q] for b in backs]), state[q], f"{shards} shards")
        assert_bits(np.concatenate([b.f.weights() for b in backs]), ow[idx], "weights")
    # the driver class itself with world = 1
    sf = ShardedFilter(HipShard(N, rows, cols, 0, 1, 0, seed), world=1)
    sf.b.set_map(g)
    sf.b.f.set_poses(x, y, th)
    sf.predict(ODOM, ALPHAS)
    sf.update(ang, dist, 5.0, 500.0, OFFSET)
    ox, oy, oth = orc.predict(x, y, th, ODOM, ALPHAS, orc.philox_normals(seed, 0, 0, N), math=orc.DET)
    ow, _ = orc.update_weights(ox, oy, oth, g, ang, dist, 5.0, 500.0, OFFSET, math=orc.DET)
    idx, _, _ = orc.resample(ow, orc.philox_uniforms(seed, 0, 0, N), orc.ORDER_BLOCKED)
    assert_bits(sf.b.f.poses()[0], ox[idx], "ShardedFilter world=1")


@pytest.mark.parametrize("mode", ["localization", "slam"])
def test_sharded_read_out_average_pose_and_best_particle(sa, mode):
    """What the app reads every frame (apps/grid_slam.cpp:112-119) from a SHARDED filter: slam::average_pose over all
    particles (slam/util.cpp:66-85) and get_particles().front(), the heaviest particle with its canvas
    (slam/mcl.cpp:236-238 puts it first on one handle).  Four shards emulated on one device, their summaries combined as
    the drivers do after the 64-byte all-gather; against the oracle and against ONE sorted handle."""
    import ctypes

    from slam_amd.dist import HipShard

    rng = np.random.default_rng(77)
    shards, seed = 4, 0xBE57
    slam = mode == "slam"
    N, B = (256, 24) if slam else (4200, 45)
    n_loc = N // shards
    if slam:
        rows = cols = 96
        free = np.ones((rows, cols), np.int32)
        free[:3, :] = free[-3:, :] = free[:, :3] = free[:, -3:] = 0
        g = None
        true, off, zmax, sd = [48.0, 44.0, 0.3], [0.0, 6.0, 0.0], 40.0, 2.0
        x, y, th = cloud(rng, N, true[0], true[1], sxy=2.0, sth=0.1, th0=true[2])
    else:
        free = orc.load_map("floor_plan")
        g = (free * 255).astype(np.uint8)
        rows, cols = g.shape
        true, off, zmax, sd = [520.0, 280.0, 0.4], OFFSET, 500.0, 5.0
        x, y, th = cloud(rng, N, true[0], true[1], th0=true[2])
    ang, dist = orc.make_scan(free, true, off, B, max_dist=zmax)
    md = sa.capi.MODE_SLAM if slam else sa.capi.MODE_LOCALIZATION
    single = sa.Filter(N, rows, cols, mode=md, seed=seed)            # SLAM_SORT_FULL: the front is the best particle
    backs = [HipShard(n_loc, rows, cols, r, shards, 0, seed, mode=md) for r in range(shards)]
    for r, q in enumerate([single] + [b.f for b in backs]):
        if not slam:
            q.set_map(g)
        sl = slice(0, N) if r == 0 else slice((r - 1) * n_loc, r * n_loc)
        q.set_poses(x[sl], y[sl], th[sl])
    maps = np.full((N, rows, cols), 128, np.uint8)
    for it in range(2):                                               # second pass: the canvases differ between particles
        for q in [single] + [b.f for b in backs]:
            q.update_weights(ang, dist, sd, zmax, off)
        ow, _ = orc.update_weights(x, y, th, maps if slam else g, ang, dist, sd, zmax, off, slam_mode=slam, math=orc.DET)
    # before any resample: the weights of this pass, particles in their original slots
    q = np.concatenate([b.f.shard_summary() for b in backs])          # what the all-gather of 8 doubles per rank produces
    lib = sa.capi.load_library()
    avg = np.empty(3)
    bw, slot, rank, loc = ctypes.c_double(), ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int64()
    sa.capi.check(lib.slam_shard_summary_combine(sa.capi.ptr(q), shards, sa.capi.ptr(avg), ctypes.byref(bw), ctypes.byref(slot),
                                                 ctypes.byref(rank), ctypes.byref(loc)), lib)
    k = int(np.flatnonzero(ow == ow.max())[0])                        # heaviest, lowest slot among equals
    assert (bits([bw.value])[0], slot.value, rank.value, loc.value) == (bits([ow[k]])[0], k, k // n_loc, k % n_loc)
    want = orc.average_pose(x, y, th, math=orc.DET)
    assert np.allclose(avg, want, rtol=1e-12, atol=0)                 # floating-point sums in another order
    sp = single.average_pose()
    assert np.allclose(avg, [sp.x, sp.y, sp.theta], rtol=1e-12, atol=0)
    if slam:
        assert np.array_equal(backs[rank.value].f.map_of(loc.value), maps[k])
    # one handle after its sorted resample: its front is the heaviest SELECTED particle; the shards, resampled from the
    # same gathered state, name the same one
    import torch

    gathered = torch.cat([b.packed_state() for b in backs])
    torch.cuda.synchronize()
    single.resample()
    for b in backs:
        if slam:
            idx = b.shard_select(gathered, shards)
            want_c = torch.unique(idx)
            pool = torch.cat([backs[int(o)].pack_maps(want_c[(want_c // n_loc) == o] - o * n_loc) for o in range(shards)])
            b.clone_maps(pool, torch.searchsorted(want_c, idx))
        else:
            b.shard_resample(gathered, shards)
    q = np.concatenate([b.f.shard_summary() for b in backs])
    sa.capi.check(lib.slam_shard_summary_combine(sa.capi.ptr(q), shards, sa.capi.ptr(avg), ctypes.byref(bw), ctypes.byref(slot),
                                                 ctypes.byref(rank), ctypes.byref(loc)), lib)
    fx, fy, fth = single.poses(0, 1)
    bx, by, bth = backs[rank.value].f.poses(loc.value, 1)
    assert bits([bw.value])[0] == bits(single.weights(0, 1))[0]
    assert (bits(bx)[0], bits(by)[0], bits(bth)[0]) == (bits(fx)[0], bits(fy)[0], bits(fth)[0])
    assert slot.value == int(single.last_order()[0])                  # the slot the sort moved to the front
    if slam:
        assert np.array_equal(backs[rank.value].f.map_of(loc.value), single.map_of(0))
    sp = single.average_pose()
    assert np.allclose(avg, [sp.x, sp.y, sp.theta], rtol=1e-12, atol=0)
    single.close()
    for b in backs:
        b.f.close()


def test_sharded_systematic_resampling(sa):
    """Low-variance (systematic) resampling over shards: one draw for the whole filter, slot k of rank r
    takes the cutoff (global k + u) * sum / N.  Four shards = one handle = the oracle."""
    import torch

    from slam_amd.dist import HipShard

    rng = np.random.default_rng(12)
    shards, N, B, seed = 4, 4200, 45, 0xABCD
    n_loc = N // shards
    free = orc.load_map("floor_plan")
    g = (free * 255).astype(np.uint8)
    rows, cols = g.shape
    true = [520.0, 280.0, 0.4]
    ang, dist = orc.make_scan(free, true, OFFSET, B)
    x, y, th = cloud(rng, N, true[0], true[1], th0=true[2])
    kw = dict(resampler=sa.capi.RESAMPLE_SYSTEMATIC)
    single = sa.Filter(N, rows, cols, seed=seed, sort=sa.capi.SORT_NONE, **kw)
    single.set_map(g)
    single.set_poses(x, y, th)
    backs = [HipShard(n_loc, rows, cols, r, shards, 0, seed, **kw) for r in range(shards)]
    for r, b in enumerate(backs):
        b.set_map(g)
        b.f.set_poses(x[r * n_loc:(r + 1) * n_loc], y[r * n_loc:(r + 1) * n_loc], th[r * n_loc:(r + 1) * n_loc])
    state = (x, y, th)
    for step in range(3):
        single.predict(ODOM, ALPHAS)
        single.update(ang, dist, 5.0, 500.0, OFFSET)
        for b in backs:
            b.predict(ODOM, ALPHAS)
            b.update_weights(ang, dist, 5.0, 500.0, OFFSET)
        gathered = torch.cat([b.packed_state() for b in backs])
        torch.cuda.synchronize()
        for b in backs:
            b.shard_resample(gathered, shards)
        ox, oy, oth = orc.predict(*state, ODOM, ALPHAS, orc.philox_normals(seed, step, 0, N), math=orc.DET)
        ow, _ = orc.update_weights(ox, oy, oth, g, ang, dist, 5.0, 500.0, OFFSET, math=orc.DET)
        idx, _, _ = orc.resample(ow, orc.philox_uniforms(seed, step, 0, N), orc.ORDER_BLOCKED, resampler=1)
        state = (ox[idx], oy[idx], oth[idx])
        assert np.all(np.diff(idx) >= 0)            # systematic: sources in order
        assert np.array_equal(single.last_indices(), idx)
        assert np.array_equal(np.concatenate([b.f.last_indices() for b in backs]), idx)
        for q in range(3):
            assert_bits(np.concatenate([b.f.poses()[q] for b in backs]), state[q], f"pose column {q}, step {step}")
    single.close()
    for b in backs:
        b.f.close()


def test_eight_shards_of_the_bench_size(sa):
    """The 8-GPU bench configuration (8 shards x 100 000 particles, floor plan), its shards emulated on
    one device: same selection, poses and weights as ONE handle of 800 000 particles."""
    import torch

    from slam_amd.dist import HipShard

    rng = np.random.default_rng(8)
    shards, n_loc, B, seed = 8, 100_000, 24, 0x5EED0001
    N = shards * n_loc
    free = orc.load_map("floor_plan")
    g = (free * 255).astype(np.uint8)
    rows, cols = g.shape
    true = [520.0, 280.0, 0.4]
    ang, dist = orc.make_scan(free, true, OFFSET, B)
    x, y, th = cloud(rng, N, true[0], true[1], sxy=25.0, sth=0.2, th0=true[2])
    single = sa.Filter(N, rows, cols, seed=seed, sort=sa.capi.SORT_NONE)
    single.set_map(g)
    single.set_poses(x, y, th)
    backs = [HipShard(n_loc, rows, cols, r, shards, 0, seed) for r in range(shards)]
    for r, b in enumerate(backs):
        b.set_map(g)
        b.f.set_poses(x[r * n_loc:(r + 1) * n_loc], y[r * n_loc:(r + 1) * n_loc], th[r * n_loc:(r + 1) * n_loc])
    for step in range(2):
        single.predict(ODOM, ALPHAS)
        single.update(ang, dist, 5.0, 500.0, OFFSET)
        for b in backs:
            b.predict(ODOM, ALPHAS)
            b.update_weights(ang, dist, 5.0, 500.0, OFFSET)
        gathered = torch.cat([b.packed_state() for b in backs])
        torch.cuda.synchronize()
        for b in backs:
            b.shard_resample(gathered, shards)
        assert np.array_equal(np.concatenate([b.f.last_indices() for b in backs]), single.last_indices())
        for q in range(3):
            assert_bits(np.concatenate([b.f.poses()[q] for b in backs]), single.poses()[q], f"pose column {q}, step {step}")
        assert_bits(np.concatenate([b.f.weights() for b in backs]), single.weights(), "weights")
    assert len(np.unique(single.last_indices())) > 1000
    single.close()
    for b in backs:
        b.f.close()


def test_eight_shards_of_config_3_global_size_against_the_oracle(sa):
    """BASELINE configs[2]'s resampling at its GLOBAL size: 8 shards x 1.25 M particles (emulated on one device), every
    shard rebuilding the running sum of all 10 M weights and selecting its own slots (slam/mcl.cpp:157-203).  The
    concatenated indices and the gathered state against the ORACLE's, not against another handle."""
    import torch

    from slam_amd.dist import HipShard

    shards, n_loc, seed = 8, 1_250_000, 0x5EED0003
    N = shards * n_loc
    rng = np.random.default_rng(83)
    w = rng.uniform(0, 1, N) ** 6
    w[rng.uniform(size=N) < 0.15] = 0.0
    x, y, th = rng.uniform(0, 16, N), rng.uniform(0, 16, N), rng.uniform(-1, 1, N)
    backs = [HipShard(n_loc, 16, 16, r, shards, 0, seed) for r in range(shards)]
    for r, b in enumerate(backs):
        sl = slice(r * n_loc, (r + 1) * n_loc)
        b.f.set_poses(x[sl], y[sl], th[sl])
        b.f.set_weights(w[sl])
    gathered = torch.cat([b.packed_state() for b in backs])      # what all_gather_into_tensor produces
    torch.cuda.synchronize()
    for b in backs:
        b.shard_resample(gathered, shards)
    oidx, oP, osum = orc.resample(w, orc.philox_uniforms(seed, 0, 0, N), orc.ORDER_BLOCKED)
    for r, b in enumerate(backs):
        sl = slice(r * n_loc, (r + 1) * n_loc)
        assert np.array_equal(b.f.last_indices(), oidx[sl]), f"indices of shard {r}"
        P, total = b.f.prefix()
        assert_bits(P, oP, f"running sums of shard {r}")
        assert bits([total])[0] == bits([osum])[0]
        px, py, pth = b.f.poses()
        assert_bits(px, x[oidx[sl]], "x"), assert_bits(py, y[oidx[sl]], "y"), assert_bits(pth, th[oidx[sl]], "theta")
        assert_bits(b.f.weights(), w[oidx[sl]], "weights")
    assert np.unique(oidx).size > N // 4
    for b in backs:
        b.f.close()


@pytest.mark.parametrize("shards,storage", [(2, "dense"), (4, "dense"), (2, "tiled"), (4, "tiled"), (2, "tiles"), (4, "tiles")])
def test_sharded_slam_mode_matches_single_handle(sa, shards, storage):
    """SLAM-mode shards: select, canvas exchange (emulated here on one device: what the two small
    all-to-alls and the canvas all-to-all of slam_amd.dist deliver), clone.  Poses, weights and every
    canvas equal the single handle's and the oracle's.  "tiled": tiled handles, canvases travel dense;
    "tiles": tiled handles, tables + distinct tiles travel (slam_mcl_pack_tiles / shard_import_tiles)."""
    import torch

    from slam_amd.dist import HipShard, ShardedFilter

    rng = np.random.default_rng(41)
    N, B, seed, rows, cols, zmax = 256, 60, 0xBEEF, 150, 200, 120.0
    free = orc.load_map("map")
    ang, dist = orc.make_scan(free, [400.0, 400.0, np.pi / 90], OFFSET, B, max_dist=zmax)
    x, y, th = cloud(rng, N, cols / 2, rows / 2, sxy=4.0, sth=0.2, th0=np.pi / 2)
    odom = [0.0, 1.5, 0.0]
    single = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=seed, sort=sa.capi.SORT_NONE)
    single.set_poses(x, y, th)
    n_loc = N // shards
    st = {"dense": sa.capi.MAPS_DENSE, "tiled": sa.capi.MAPS_TILED, "tiles": sa.capi.MAPS_TILED}[storage]
    backs = [HipShard(n_loc, rows, cols, r, shards, 0, seed, mode=sa.capi.MODE_SLAM, map_storage=st) for r in range(shards)]
    for r, b in enumerate(backs):
        b.f.set_poses(x[r * n_loc:(r + 1) * n_loc], y[r * n_loc:(r + 1) * n_loc], th[r * n_loc:(r + 1) * n_loc])
    state, maps = (x, y, th), np.full((N, rows, cols), 128, np.uint8)
    for step in range(3):
        single.predict(odom, ALPHAS)
        single.update(ang, dist, 5.0, zmax, OFFSET)
        for b in backs:
            b.predict(odom, ALPHAS)
            b.update_weights(ang, dist, 5.0, zmax, OFFSET)
        gathered = torch.cat([b.packed_state() for b in backs])
        torch.cuda.synchronize()
        idxs = [b.shard_select(gathered, shards) for b in backs]
        with pytest.raises(RuntimeError):  # canvases pending: weighting must refuse
            backs[0].update_weights(ang, dist, 5.0, zmax, OFFSET)
        pools, wants = [], []
        for r, b in enumerate(backs):   # requester r
            want = torch.unique(idxs[r])
            owner = torch.div(want, n_loc, rounding_mode="floor")
            wants.append(want)
            if storage == "tiles":
                tabs, tiles, base = [], [], 0
                for o in range(shards):
                    t, p = backs[o].pack_tiles(want[owner == o] - o * n_loc)
                    tabs.append(torch.where(t > 0, t + base, t))
                    tiles.append(p)
                    base += p.shape[0]
                pools.append((torch.cat(tabs), torch.cat(tiles)))
                continue
            parts = [backs[o].pack_maps((want[owner == o] - o * n_loc)) for o in range(shards)]
            pools.append(torch.cat(parts))
        for r, b in enumerate(backs):
            if storage == "tiles":
                b.import_tiles(*pools[r], torch.searchsorted(wants[r], idxs[r]))
            else:
                b.clone_maps(pools[r], torch.searchsorted(wants[r], idxs[r]))
        ox, oy, oth = orc.predict(*state, odom, ALPHAS, orc.philox_normals(seed, step, 0, N), math=orc.DET)
        ow, _ = orc.update_weights(ox, oy, oth, maps, ang, dist, 5.0, zmax, OFFSET, slam_mode=True, math=orc.DET)
        idx, _, _ = orc.resample(ow, orc.philox_uniforms(seed, step, 0, N), orc.ORDER_BLOCKED)
        state, maps = (ox[idx], oy[idx], oth[idx]), maps[idx].copy()
        assert np.array_equal(single.last_indices(), idx)
        assert np.array_equal(np.concatenate([b.f.last_indices() for b in backs]), idx)
        for q in range(3):
            assert_bits(np.concatenate([b.f.poses()[q] for b in backs]), state[q], f"{shards} shards")
        assert_bits(np.concatenate([b.f.weights() for b in backs]), ow[idx], "weights")
        for p in range(0, N, 7):
            assert np.array_equal(single.map_of(p), maps[p]), f"single handle, canvas {p}, step {step}"
            assert np.array_equal(backs[p // n_loc].f.map_of(p % n_loc), maps[p]), f"shard canvas {p}, step {step}"
    assert len(np.unique(maps.reshape(N, -1), axis=0)) > 1
    # the driver class itself with world = 1 (pack + clone through the same entry points)
    sf = ShardedFilter(HipShard(N, rows, cols, 0, 1, 0, seed, mode=sa.capi.MODE_SLAM, map_storage=st), world=1)
    sf.b.f.set_poses(x, y, th)
    m1 = np.full((N, rows, cols), 128, np.uint8)
    sf.predict(odom, ALPHAS)
    sf.update(ang, dist, 5.0, zmax, OFFSET)
    ox, oy, oth = orc.predict(x, y, th, odom, ALPHAS, orc.philox_normals(seed, 0, 0, N), math=orc.DET)
    ow, _ = orc.update_weights(ox, oy, oth, m1, ang, dist, 5.0, zmax, OFFSET, slam_mode=True, math=orc.DET)
    idx, _, _ = orc.resample(ow, orc.philox_uniforms(seed, 0, 0, N), orc.ORDER_BLOCKED)
    assert_bits(sf.b.f.poses()[0], ox[idx], "ShardedFilter world=1, SLAM mode")
    for p in range(0, N, 17):
        assert np.array_equal(sf.b.f.map_of(p), m1[idx[p]])


# ---- log-domain weights (opt-in extension) ----------------------------------------
@pytest.mark.parametrize("raycaster,order", [("beam_lanes", "blocked"), ("particle_lanes", "blocked"), ("walk", "reference")])
def test_log_domain_weights_and_selection(sa, raycaster, order):
    """B = 720: the reference's exp(sum of logs) is 0 for almost every particle; the
    log-domain mode keeps the sum and selects on exp(lw - max).  Both against the oracle, bit for bit."""
    rng = np.random.default_rng(51)
    N, B, seed = 3000, 720, 0x106
    free = orc.load_map("floor_plan")
    g = (free * 255).astype(np.uint8)
    rows, cols = g.shape
    true = [520.0, 280.0, 0.4]
    ang, dist = orc.make_scan(free, true, OFFSET, B)
    x, y, th = cloud(rng, N, true[0], true[1], sxy=0.2, sth=0.001, th0=true[2])   # tight: many particles keep some mass
    po = sa.capi.PREFIX_BLOCKED if order == "blocked" else sa.capi.PREFIX_REFERENCE
    oo = orc.ORDER_BLOCKED if order == "blocked" else orc.ORDER_REFERENCE
    f = sa.Filter(N, rows, cols, seed=seed, raycaster=RAYCASTERS[raycaster], prefix_order=po, sort=sa.capi.SORT_NONE,
                  weight_domain=sa.capi.WEIGHT_LOG)
    lin = sa.Filter(N, rows, cols, seed=seed, raycaster=RAYCASTERS[raycaster], sort=sa.capi.SORT_NONE)
    for q in (f, lin):
        q.set_map(g)
        q.set_poses(x, y, th)
        q.update_weights(ang, dist, 5.0, 500.0, OFFSET)
    olw, _ = orc.update_weights(x, y, th, g, ang, dist, 5.0, 500.0, OFFSET, math=orc.DET, log_domain=True)
    lw = f.weights()
    assert_bits(lw, olw, "sum of log likelihoods")
    assert np.all(np.isfinite(lw)) and lw.max() < 0
    assert (lin.weights() == 0).mean() > 0.5                    # what the linear domain makes of this scan
    f.resample()
    wl = orc.log_domain_weights(olw, orc.DET)
    assert wl.max() == 1.0
    idx, P, _ = orc.resample(wl, orc.philox_uniforms(seed, 0, 0, N), oo)
    assert_bits(f.prefix()[0], P, "running sum of exp(lw - max)")
    assert np.array_equal(f.last_indices(), idx)
    assert_bits(f.weights(), olw[idx], "weights after the gather stay in the log domain")
    assert np.unique(idx).size > 1
    f.close(), lin.close()


def test_log_domain_slam_mode_and_sort(sa):
    rng = np.random.default_rng(52)
    N, B, rows, cols, zmax, seed = 64, 500, 160, 200, 90.0, 0x107
    free = orc.load_map("map")
    ang, dist = orc.make_scan(free, [400.0, 400.0, np.pi / 90], OFFSET, B, max_dist=zmax)
    x, y, th = cloud(rng, N, cols / 2, rows / 2, sxy=3.0, sth=0.2, th0=np.pi / 2)
    f = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=seed, weight_domain=sa.capi.WEIGHT_LOG)
    f.set_poses(x, y, th)
    maps = np.full((N, rows, cols), 128, np.uint8)
    for it in range(2):
        f.update_weights(ang, dist, 5.0, zmax, OFFSET)
        olw, _ = orc.update_weights(x, y, th, maps, ang, dist, 5.0, zmax, OFFSET, slam_mode=True, math=orc.DET, log_domain=True)
        assert_bits(f.weights(), olw, f"pass {it}")
    f.resample()
    idx, _, _ = orc.resample(orc.log_domain_weights(olw, orc.DET), orc.philox_uniforms(seed, 0, 0, N), orc.ORDER_BLOCKED)
    assert np.array_equal(f.last_indices(), idx)
    perm = f.last_order()
    sel = olw[idx]
    assert np.array_equal(perm, orc.sort_order(sel, 0))        # (log-weight desc, slot asc): same order as the weights
    assert_bits(f.weights(), sel[perm], "sorted log weights")
    for p in range(0, N, 5):
        assert np.array_equal(f.map_of(p), maps[idx[perm[p]]])
    f.close()


# ---- per-GPU shard sizes of BASELINE configs[2] (10M x 1080 over 8 GPUs) and [4] (100M x 720 over 8) ---
@pytest.mark.parametrize("size,N,B,zmax,uniform", [(8192, 1_250_000, 1080, 500.0, False), (16384, 12_500_000, 720, 4000.0, True)])
def test_full_shard_sizes_of_configs_3_and_5(sa, size, N, B, zmax, uniform):
    """One GPU's share of the two 8-GPU configurations, through properties the size does not change:
    a random subset of the particles recomputed by the cell-by-cell kernel gives the same bits, the
    pass is idempotent, the running sum ends at the sum and the selection stays in range.  Log-domain
    weights: at these beam counts the linear weights are all zero."""
    import time

    from slam_amd import synth

    g = synth.make_grid(size)
    free01 = (g >= 128).astype(np.uint8)
    rng = np.random.default_rng(size + 1)
    true = synth.open_pose(g)
    lib = sa.capi.load_library()
    sensor = orc.sensor_position(true, OFFSET, orc.DET)
    hits = np.empty((B, 2))
    sa.capi.check(lib.slam_fake_lidar_scan(0, sa.capi.ptr(free01), size, size, sa.capi.ptr(np.ascontiguousarray(sensor[None])), 1,
                                           0.0, 2 * np.pi, zmax, B, sa.capi.ptr(hits)))
    ang, dist = orc.hits_to_scan(hits, sensor, 0.0, 2 * np.pi, zmax)
    seed = 0x5EED1000 + size
    f = sa.Filter(N, size, size, seed=seed, sort=sa.capi.SORT_NONE, weight_domain=sa.capi.WEIGHT_LOG)
    f.set_map(g)
    if uniform:
        f.seed_uniform_free()
    else:
        f.seed_gaussian(true, [25.0, 25.0, 0.2])
    f.predict(ODOM, ALPHAS)
    t0 = time.perf_counter()
    f.update_weights(ang, dist, 0.5, zmax, OFFSET)
    dt = time.perf_counter() - t0
    print(f"\n{N} x {B} on {size}^2: weighting pass {dt * 1e3:.1f} ms = {N * B / dt / 1e9:.1f} G particle-beams/s")
    x, y, th = f.poses()
    lw = f.weights()
    assert np.all(np.isfinite(lw)) and lw.max() < 0
    sub = np.sort(rng.choice(N, 4096, replace=False))
    walk = sa.Filter(sub.size, size, size, raycaster=sa.capi.RAYCAST_WALK, sort=sa.capi.SORT_NONE,
                     weight_domain=sa.capi.WEIGHT_LOG)
    walk.set_map(g)
    walk.set_poses(x[sub], y[sub], th[sub])
    walk.update_weights(ang, dist, 0.5, zmax, OFFSET)
    assert_bits(lw[sub], walk.weights(), "subset recomputed cell by cell")
    walk.close()
    osub = rng.permutation(sub)[:2048]   # the threaded oracle walks these ~2 M rays cell by cell in a few seconds
    olw, _ = orc.update_weights(x[osub], y[osub], th[osub], g, ang, dist, 0.5, zmax, OFFSET, math=orc.DET, log_domain=True,
                                n_threads=16)
    assert_bits(lw[osub], olw, "subset vs oracle")
    # (both passes know the cloud's extent: the uniform cloud over 16384^2 runs on the wave-per-particle kernel in place
    # order, the Gaussian one on lanes over particles with the heading first)
    f.synchronize()
    t0 = time.perf_counter()
    f.update_weights(ang, dist, 0.5, zmax, OFFSET)
    f.synchronize()
    print(f"second pass {(time.perf_counter() - t0) * 1e3:.1f} ms")
    assert_bits(f.weights(), lw, "idempotence")
    f.resample()
    P, total = f.prefix()
    # (the blocked order is monotone up to the rounding at chunk seams)
    assert np.all(np.diff(P) >= -1e-12 * total) and P[-1] == total and 1.0 <= total <= N
    idx = f.last_indices()
    assert idx.min() >= 0 and idx.max() < N
    assert_bits(f.weights(), lw[idx], "gathered weights")
    # slam/mcl.cpp:157-203 at the full shard size against the oracle: running sums bit for bit and every selected index
    # (4 levels of the radix-64 hierarchy; at 12.5 M more than 2^23 slots, 24 bisection steps)
    oidx, oP, osum = orc.resample(orc.log_domain_weights(lw, orc.DET), orc.philox_uniforms(seed, 0, 0, N), orc.ORDER_BLOCKED)
    assert_bits(P, oP, "running sums vs oracle")
    assert bits([total])[0] == bits([osum])[0]
    assert np.array_equal(idx, oidx), "selected indices vs oracle"
    f.close()


def test_full_shard_size_of_config_4_grid_slam(sa):
    """One GPU's share of BASELINE configs[3] (full grid_slam loop, 1M particles over 8 GPUs): 125 000 particles
    with maze-sized 2354 x 2464 canvases as copy-on-write tiles, 90 beams, several filter steps (the app's
    odometry and alphas, apps/grid_slam.cpp:48-50,132).  Size-independent checks per step: a random subset of
    particles (poses AND canvases fetched from the big filter) recomputed by the one-thread-per-particle
    literal kernel on dense canvases gives the same weights, hit cells and canvases bit for bit; a smaller
    subset by the CPU oracle too; the pool never runs dry; the selection stays in range and the gathered
    state is the selected state."""
    import time

    N, B, rows, cols = 125_000, 90, 2354, 2464
    free = orc.load_map("maze")
    assert free.shape == (rows, cols)
    rng = np.random.default_rng(4)
    odom, alphas = [0.0, 2.5, 0.0], [1e-3] * 4
    true = [1944.5, 2188.5, np.pi / 2]  # a place of the maze with 40 cells of clearance and a mean range of ~140 cells
    gm = sa.GridMap(free.astype(np.uint8))
    pool = 16_000_000  # 64 GB: every particle copies the ~40 tiles its scan writes in every pass (clones share tiles again after each resample)
    f = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=0x5EED4004, sort=sa.capi.SORT_NONE, map_storage=sa.capi.MAPS_TILED,
                  tile_pool_tiles=pool)
    S, SO = 64, 64   # per step: 64 particles recomputed by the literal kernel AND by the CPU oracle, canvases included
    small = sa.Filter(S, rows, cols, mode=sa.capi.MODE_SLAM, sort=sa.capi.SORT_NONE, map_storage=sa.capi.MAPS_DENSE,
                      raycaster=sa.capi.RAYCAST_FAST_PARTICLE_LANES)
    small.record_hits(True)
    for step in range(4):
        true = [true[0] + 2.5 * np.cos(true[2]), true[1] + 2.5 * np.sin(true[2]), true[2]]
        sensor = orc.sensor_position(true, OFFSET, orc.DET)
        ang, dist = orc.hits_to_scan(gm.fake_lidar_scan(sensor, 0.0, 2 * np.pi, 500.0, B)[0], sensor, 0.0, 2 * np.pi, 500.0)
        f.predict(odom, alphas)
        x, y, th = f.poses()
        sub = np.sort(rng.choice(N, S, replace=False))
        before = np.stack([f.map_of(int(p)) for p in sub])
        t0 = time.perf_counter()
        f.update_weights(ang, dist, 5.0, 500.0, OFFSET)
        dt = time.perf_counter() - t0
        w = f.weights()
        st, ts = f.stats(), f.tile_stats()
        print(f"\nstep {step}: weighting pass of {N} x {B} on {rows}x{cols} tiles {dt * 1e3:.1f} ms; per ray {st['cells_read'] / (N * B):.1f} cells read, "
              f"{st['cells_written'] / (N * B):.1f} written; tiles live {ts['live']} + copied {ts['allocated_since']} of {ts['pool']}")
        assert ts["live"] + ts["allocated_since"] < pool
        small.set_poses(x[sub], y[sub], th[sub])
        small.set_maps(before)
        small.update_weights(ang, dist, 5.0, 500.0, OFFSET)
        assert_bits(w[sub], small.weights(), f"step {step}: subset recomputed by the literal kernel on dense canvases")
        for k, p in enumerate(sub):
            assert np.array_equal(f.map_of(int(p)), small.map_of(k)), f"step {step}: canvas of particle {p}"
        omaps = before[:SO].copy()
        ow, oxy, ocell, _ = orc.update_weights(x[sub[:SO]], y[sub[:SO]], th[sub[:SO]], omaps, ang, dist, 5.0, 500.0, OFFSET,
                                               slam_mode=True, want_hits=True, math=orc.DET, n_threads=16)
        assert_bits(w[sub[:SO]], ow, f"step {step}: subset vs oracle")
        assert np.array_equal(small.hits()[1][:SO], ocell)
        for k in range(SO):   # slam/raycast.cpp:194-214: the canvases the oracle drew = the tiles of the big filter
            assert np.array_equal(small.map_of(k), omaps[k]), f"step {step}: canvas of particle {sub[k]} vs oracle"
        if step:
            assert (ocell >= 0).any()  # from the second scan on the rays meet the walls the first one drew
        f.resample()
        idx = f.last_indices()
        assert idx.min() >= 0 and idx.max() < N
        nx, ny, nth = f.poses()
        assert_bits(nx, x[idx], "gathered x"), assert_bits(ny, y[idx], "gathered y"), assert_bits(nth, th[idx], "gathered theta")
        assert_bits(f.weights(), w[idx], "gathered weights")
        k = int(rng.integers(0, S))
        hit = np.flatnonzero(idx == sub[k])
        if hit.size:  # a selected particle carries its source's canvas (slam/mcl.cpp:147-155)
            assert np.array_equal(f.map_of(int(hit[0])), small.map_of(k))
    small.close(), f.close(), gm.close()


def test_tiled_canvases_beyond_262144_particles(sa):
    """SLAM_MAPS_AUTO on canvases too large to hold dense (300 000 x 2354 x 2464 x 2 = 3.5 TB) picks tiles, and the
    two-phase kernel runs on them at that particle count (round 1 fell back to dense storage from 262 144 on): a
    random subset recomputed by the literal one-thread-per-particle kernel on dense canvases gives the same weights
    and canvases bit for bit over three steps; the pool never runs dry."""
    N, B, rows, cols = 300_000, 24, 2354, 2464
    free = orc.load_map("maze")
    rng = np.random.default_rng(11)
    gm = sa.GridMap(free.astype(np.uint8))
    pool = 8_000_000  # 32 GB
    f = sa.Filter(N, rows, cols, mode=sa.capi.MODE_SLAM, seed=0xB16, sort=sa.capi.SORT_NONE, tile_pool_tiles=pool)
    assert f.tile_stats()["pool"] == pool  # AUTO chose tiles
    S = 32
    small = sa.Filter(S, rows, cols, mode=sa.capi.MODE_SLAM, sort=sa.capi.SORT_NONE, map_storage=sa.capi.MAPS_DENSE,
                      raycaster=sa.capi.RAYCAST_FAST_PARTICLE_LANES)
    true = [1944.5, 2188.5, np.pi / 2]
    for step in range(3):
        true = [true[0] + 2.5 * np.cos(true[2]), true[1] + 2.5 * np.sin(true[2]), true[2]]
        sensor = orc.sensor_position(true, OFFSET, orc.DET)
        ang, dist = orc.hits_to_scan(gm.fake_lidar_scan(sensor, 0.0, 2 * np.pi, 60.0, B)[0], sensor, 0.0, 2 * np.pi, 60.0)
        f.predict([0.0, 2.5, 0.0], [1e-3] * 4)
        x, y, th = f.poses()
        sub = np.sort(rng.choice(N, S, replace=False))
        before = np.stack([f.map_of(int(p)) for p in sub])
        f.update_weights(ang, dist, 5.0, 60.0, OFFSET)
        w = f.weights()
        ts = f.tile_stats()
        assert ts["live"] + ts["allocated_since"] < pool
        small.set_poses(x[sub], y[sub], th[sub])
        small.set_maps(before)
        small.update_weights(ang, dist, 5.0, 60.0, OFFSET)
        assert_bits(w[sub], small.weights(), f"step {step}: subset recomputed on dense canvases")
        for k, p in enumerate(sub):
            assert np.array_equal(f.map_of(int(p)), small.map_of(k)), f"step {step}: canvas of particle {p}"
        f.resample()
        idx = f.last_indices()
        assert idx.min() >= 0 and idx.max() < N
    small.close(), f.close(), gm.close()


# ---- BASELINE configs[2] and [4] at reduced particle counts: synthetic grids, long ranges, seeding ----
def test_seeding_matches_host_replay(sa):
    from slam_amd import synth

    g = synth.make_grid(1024)
    N, seed = 5000, 0xFACE
    f = sa.Filter(N, *g.shape, seed=seed)
    f.set_map(g)
    f.seed_uniform_free()
    x, y, th = f.poses()
    ox, oy, oth = orc.seed_uniform_free(seed, 0, N, g)
    assert_bits(x, ox, "x"), assert_bits(y, oy, "y"), assert_bits(th, oth, "theta")
    i, j = (g.shape[0] - y - 1).astype(int), x.astype(int)
    assert np.all(g[i, j] >= 128) and th.min() > -np.pi - 1e-12 and th.max() <= np.pi
    assert np.unique(i * g.shape[1] + j).size > 0.98 * N       # spread over the free space
    f.seed_gaussian([500.0, 400.0, 0.3], [25.0, 25.0, 0.2])
    gx, gy, gth = orc.seed_gaussian(seed, 0, N, [500.0, 400.0, 0.3], [25.0, 25.0, 0.2])
    x, y, th = f.poses()
    assert_bits(x, gx, "x"), assert_bits(y, gy, "y"), assert_bits(th, gth, "theta")
    assert_bits(f.weights(), np.full(N, 1.0 / N), "weights")
    f.close()


@pytest.mark.parametrize("size,N,B,zmax,stddev,uniform", [(8192, 1536, 1080, 500.0, 0.5, False), (16384, 1024, 720, 4000.0, 0.5, True)])
def test_large_synthetic_grids(sa, size, N, B, zmax, stddev, uniform):
    from slam_amd import synth

    g = synth.make_grid(size)
    free01 = (g >= 128).astype(np.uint8)
    rng = np.random.default_rng(size)
    free_cells = np.argwhere(g[size // 4:size // 4 + 64, size // 4:size // 4 + 2000] >= 128)
    ci, cj = free_cells[len(free_cells) // 2]
    true = [float(cj + size // 4) + 0.5, float(size - 1 - (ci + size // 4)) + 0.5, 0.7]
    lib = sa.capi.load_library()
    sensor = orc.sensor_position(true, OFFSET, orc.DET)
    hits = np.empty((B, 2))
    sa.capi.check(lib.slam_fake_lidar_scan(0, sa.capi.ptr(free01), size, size, sa.capi.ptr(np.ascontiguousarray(sensor[None])), 1,
                                           0.0, 2 * np.pi, zmax, B, sa.capi.ptr(hits)))
    ang, dist = orc.hits_to_scan(hits, sensor, 0.0, 2 * np.pi, zmax)
    seed = 0x5EED0000 + size
    fast = sa.Filter(N, size, size, seed=seed, raycaster=sa.capi.RAYCAST_FAST_PARTICLE_LANES)
    beam = sa.Filter(N, size, size, seed=seed, raycaster=sa.capi.RAYCAST_FAST_BEAM_LANES)
    for f in (fast, beam):
        f.set_map(g)
        if uniform:
            f.seed_uniform_free()
        else:
            f.seed_gaussian(true, [25.0, 25.0, 0.2])
        f.predict(ODOM, ALPHAS)
        f.record_hits(True)
        f.update_weights(ang, dist, stddev, zmax, OFFSET)
    assert_bits(fast.weights(), beam.weights(), "two kernel shapes")
    x, y, th = fast.poses()
    sub = rng.choice(N, 48, replace=False)
    ow, oxy, ocell, _ = orc.update_weights(x[sub], y[sub], th[sub], g, ang, dist, stddev, zmax, OFFSET, want_hits=True, math=orc.DET)
    hxy, hcell = fast.hits()
    assert np.array_equal(hcell[sub], ocell) and (ocell >= 0).any()
    assert_bits(hxy[sub], oxy, "hit positions")
    assert_bits(fast.weights()[sub], ow, "weights")
    fast.resample()
    w = fast.prefix()  # running sum and selection stay consistent at this size
    idx = fast.last_indices()
    assert idx.min() >= 0 and idx.max() < N
    for f in (fast, beam):
        f.close()
