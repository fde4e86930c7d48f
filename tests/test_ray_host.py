"""CPU checks of the device ray functions (slam_amd/csrc/ray.h compiled for the host by
tests/host_harness): the exact-jump position update and the clearance-skipping hit search must give
bit-identical results to plain repeated addition / the literal cell-by-cell walk."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import _oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
P = ctypes.c_void_p


@pytest.fixture(scope="module")
def hh():
    d = os.path.join(HERE, "host_harness")
    subprocess.run(["make", "-C", d, "-s"], check=True)
    lib = ctypes.CDLL(os.path.join(d, "libray_harness.so"))
    lib.hh_rays_compare.restype = ctypes.c_long
    lib.hh_advance_check.restype = ctypes.c_long
    return lib


def ptr(a):
    return a.ctypes.data_as(P)


def advance_bad(hh, x0, y0, dx, dy, total, jumps):
    j = np.ascontiguousarray(jumps, dtype=np.int32)
    js = ctypes.c_long(0)
    bad = hh.hh_advance_check(ctypes.c_double(x0), ctypes.c_double(y0), ctypes.c_double(dx), ctypes.c_double(dy),
                              total, ptr(j), len(j), ctypes.byref(js))
    return bad, js.value


def test_exact_jumps_equal_repeated_addition(hh):
    rng = np.random.default_rng(1)
    jumped = 0
    for t in range(6000):
        th = rng.uniform(-np.pi, np.pi)
        scale = 10.0 ** rng.uniform(-1, 4.3)
        x0, y0 = rng.uniform(-1, 1) * scale, rng.uniform(-1, 1) * scale
        if t % 3 == 0:  # small binades: ties (dx an odd multiple of ulp/2) are frequent there
            x0, y0 = rng.uniform(0.9, 4.1), rng.uniform(0.4, 2.2)
        dx, dy = np.cos(th), np.sin(th)
        if t % 5 == 0:  # few significant bits -> exact ties
            dx = np.round(dx * 2.0 ** rng.integers(3, 54)) / 2.0 ** 53 * 2.0 ** rng.integers(0, 3)
        bad, js = advance_bad(hh, x0, y0, dx, dy, 600, rng.integers(1, 40, size=16))
        assert bad == 0, (x0, y0, dx, dy)
        jumped += js
    for t in range(2000):  # forced ties on both axes
        dx = (2 * int(rng.integers(2 ** 50, 2 ** 52)) + 1) * 2.0 ** -53
        dy = -(2 * int(rng.integers(2 ** 50, 2 ** 52)) + 1) * 2.0 ** -53
        bad, _ = advance_bad(hh, rng.uniform(1, 2), rng.uniform(2, 4), dx, dy, 300, rng.integers(1, 20, size=8))
        assert bad == 0
    tie_jumped = 0
    for t in range(3000):  # ties in the binades a 4096-cell map uses: one real add, then constant increments again
        ex, ey = int(rng.integers(10, 13)), int(rng.integers(10, 13))
        dx = np.ldexp(2 * int(rng.integers(2 ** (51 - ex), 2 ** (52 - ex))) + 1, ex - 53)  # odd multiple of ulp(2^ex)/2, < 1
        dy = -np.ldexp(2 * int(rng.integers(2 ** (51 - ey), 2 ** (52 - ey))) + 1, ey - 53)
        x0, y0 = rng.uniform(2.0 ** ex, 2.0 ** (ex + 1) - 400), rng.uniform(2.0 ** ey + 400, 2.0 ** (ey + 1))
        bad, js = advance_bad(hh, x0, y0, dx, dy, 300, rng.integers(1, 40, size=8))
        assert bad == 0, (x0, y0, dx, dy)
        tie_jumped += js
    assert tie_jumped > 0.9 * 3000 * 300, "ties must not fall back to sample-by-sample adds"
    for dx, dy in [(0.0, 1.0), (1.0, 0.0), (6.123233995736766e-17, 1.0), (-1.0, 1e-300), (2.0 ** -60, -1.0)]:
        bad, _ = advance_bad(hh, 500.25, 0.75, dx, dy, 800, [1, 7, 64, 255, 3])
        assert bad == 0
    assert jumped > 0.5 * 6000 * 600, "the fma fast path should carry most samples"


def compare(hh, grid, sx, sy, th, zmax):
    rows, cols = grid.shape
    E = np.empty_like(grid)
    hh.hh_build_clearance(ptr(grid), rows, cols, ptr(E))
    n = sx.size
    fb, rf, rw = ctypes.c_long(), ctypes.c_ulonglong(), ctypes.c_ulonglong()
    hits = np.empty((n, 2))
    bad = hh.hh_rays_compare(ptr(grid), ptr(E), rows, cols, ctypes.c_long(n), ptr(sx), ptr(sy), ptr(th),
                             ctypes.c_double(zmax), ctypes.byref(fb), ctypes.byref(rf), ctypes.byref(rw), ptr(hits))
    return bad, fb.value, rf.value / n, rw.value / n, hits, E


@pytest.mark.parametrize("name", ["floor_plan", "map", "maze"])
def test_fast_hit_search_equals_literal_walk(hh, name):
    rng = np.random.default_rng(2)
    g = (orc.load_map(name) * 255).astype(np.uint8)
    rows, cols = g.shape
    for n, zmax, spread in [(60000, 500.0, 0), (20000, 500.0, 60), (20000, 37.25, 0), (10000, 0.7, 0), (10000, 1.0, 5),
                            (5000, 4000.0, 0)]:
        sx, sy = rng.uniform(-spread, cols + spread, n), rng.uniform(-spread, rows + spread, n)
        th = rng.uniform(-7, 7, n)
        bad, first, rfast, rwalk, hits, _ = compare(hh, g, sx, sy, th, zmax)
        assert bad == 0, f"{name} zmax={zmax}: {bad} rays differ, first {first}: {sx[first]!r} {sy[first]!r} {th[first]!r}"
        if zmax >= 500:
            assert rfast < 0.1 * rwalk  # the point of the clearance grid
    # grid-aligned starts and axis/diagonal directions (cell edges, exact ties)
    n = 40000
    sx, sy = rng.integers(0, cols, n).astype(float), rng.integers(0, rows, n).astype(float)
    th = rng.integers(0, 8, n) * (np.pi / 4)
    assert compare(hh, g, sx, sy, th, 500.0)[0] == 0
    # far outside the grid, including the coordinate range where only the literal loop is used
    sx = np.array([-1e7, 3e6, 2e9, -5e9, 1e15, 100.0, cols + 0.5])
    sy = np.array([10.0, -4e6, 2e9, 7.0, 3.0, -2e6, rows - 0.5])
    th = np.array([0.0, 1.0, 2.0, 3.0, -1.0, 1.5707963267948966, 3.0])
    assert compare(hh, g, sx, sy, th, 500.0)[0] == 0


@pytest.mark.parametrize("name", ["floor_plan", "maze"])
def test_sample_limit_of_the_hit_search_keeps_the_beam_likelihood(hh, name):
    """ray_fast_L0's k_lim: a search that stops beyond z + 4 sigma must give the same log-likelihood, bit for bit,
    as the search to max_dist (slam/util.cpp:14-19 clamps the density to 0 beyond 4 sigma, slam/raycast.cpp:233-241)."""
    rng = np.random.default_rng(21)
    hh.hh_limit_compare.restype = ctypes.c_long
    g = (orc.load_map(name) * 255).astype(np.uint8)
    rows, cols = g.shape
    total_stopped = 0
    for n, zmax, sigma, spread in [(40000, 500.0, 5.0, 0), (20000, 500.0, 0.5, 30), (10000, 4000.0, 0.5, 0), (10000, 60.0, 20.0, 0),
                                   (5000, 3.0, 0.1, 0)]:
        sx, sy = rng.uniform(-spread, cols + spread, n), rng.uniform(-spread, rows + spread, n)
        th = rng.uniform(-7, 7, n)
        # ranges of every kind: short, near the limit, exactly max_dist, beyond it; some on whole cells
        z = rng.uniform(0, zmax * 1.05, n)
        z[::7] = zmax
        z[1::7] = np.floor(z[1::7])
        z[2::7] = np.maximum(zmax - 4 * sigma + rng.uniform(-3, 3, z[2::7].size), 0)
        fb, st, rl, rf = ctypes.c_long(), ctypes.c_long(), ctypes.c_ulonglong(), ctypes.c_ulonglong()
        bad = hh.hh_limit_compare(ptr(g), rows, cols, ctypes.c_long(n), ptr(sx), ptr(sy), ptr(th), ptr(z), ctypes.c_double(zmax),
                                  ctypes.c_double(sigma), ctypes.byref(fb), ctypes.byref(st), ctypes.byref(rl), ctypes.byref(rf))
        assert bad == 0, (name, zmax, sigma, fb.value, sx[fb.value], sy[fb.value], th[fb.value], z[fb.value])
        assert rl.value <= rf.value
        total_stopped += st.value
    assert total_stopped > 1000  # the limit does end searches


def test_beam_likelihood_in_two_steps_equals_the_one_step_form(hh):
    """slam_beam_clamped decides "beyond 4 sigma" from squared distances (no square root) and may only say so when
    slam_beam_logw (slam/raycast.cpp:233-241, slam/util.cpp:9-19) returns the clamped value; everything else is evaluated by
    the same statements from the squared distance.  Rays placed ON the clamp (to the last bits), far from it, misses, limit stops."""
    rng = np.random.default_rng(44)
    hh.hh_logw_two_step.restype = ctypes.c_long
    total_deferred = 0
    for sigma, zmax in [(5.0, 500.0), (0.5, 4000.0), (0.01, 30.0), (40.0, 100.0), (1e-3, 2.0)]:
        n = 300000
        sx, sy = rng.uniform(0, 2000, n), rng.uniform(0, 2000, n)
        z = rng.uniform(0, zmax, n)
        z[::9] = zmax
        z[1::9] = np.minimum(rng.uniform(0, 8 * sigma, z[1::9].size), zmax)   # lo = z - clamp near or below 0
        # hit distance: anywhere; near z; within a few ulp of z +- 4 sigma
        d = rng.uniform(0, 1.2 * zmax, n)
        near = rng.uniform(size=n) < 0.5
        d[near] = np.abs(z[near] + rng.uniform(-6, 6, near.sum()) * sigma)
        edge = rng.uniform(size=n) < 0.25
        d[edge] = np.abs(z[edge] + rng.choice([-4.0, 4.0], edge.sum()) * sigma * (1 + rng.integers(-40, 41, edge.sum()) * 2.0 ** -50))
        phi = rng.uniform(-np.pi, np.pi, n)
        hx, hy = sx + d * np.cos(phi), sy + d * np.sin(phi)
        hx[2::13] = -1.0   # misses
        hx[3::17] = -2.0   # searches that ended at their limit
        hy[hx < 0] = hx[hx < 0]
        for fast_div in (0, 1):
            dfr = ctypes.c_long()
            bad = hh.hh_logw_two_step(ctypes.c_long(n), ptr(hx), ptr(hy), ptr(sx), ptr(sy), ptr(z), ctypes.c_double(zmax), ctypes.c_double(sigma),
                                      fast_div, ctypes.byref(dfr))
            assert bad == 0, (sigma, zmax, fast_div, bad)
            total_deferred += dfr.value
    assert 200000 < total_deferred < 2200000   # both kinds of ray occur


def test_quotient_from_the_reciprocal_is_the_correctly_rounded_quotient(hh):
    """slam_beam_logw forms xarg / stddev (slam/util.cpp:11) from RN(1 / stddev) with two correction steps in the
    particle-lane kernel; the result must be the division's, bit for bit: random operands in the range the kernel
    uses (|xarg| <= 4 stddev), operands one ulp around exact quotients, denominators with all-ones significands
    (where the reciprocal is hardest) and the extremes of the accepted stddev range."""
    rng = np.random.default_rng(31)
    hh.hh_fast_div_check.restype = ctypes.c_long
    n = 2_000_000
    b = np.concatenate([rng.uniform(0.01, 50.0, n // 2), 10.0 ** rng.uniform(-100, 100, n // 4),
                        np.nextafter(2.0 ** rng.integers(-300, 300, n // 4).astype(float), 0.0)])
    a = rng.uniform(-4, 4, n) * b
    q = np.round(rng.uniform(-4, 4, n // 4) * 2.0 ** 30) / 2.0 ** 30   # short quotients: a = q b is nearly exact
    a[:n // 4] = np.nextafter(q * b[:n // 4], rng.choice([-np.inf, np.inf], n // 4))
    a[n // 4:n // 4 + 1000] = 0.0
    a[n // 4 + 1000:n // 4 + 2000] = rng.choice([-1.0, 1.0], 1000) * 2.2e-16
    assert hh.hh_fast_div_check(ctypes.c_long(n), ptr(a), ptr(b)) == 0


def test_rays_from_beyond_two_to_the_twenty_are_walked_to_the_end(hh):
    """A grid side of 2^20 cells (the largest create() accepts): a ray that starts just outside it and heads inward
    lies in the coordinate range that takes the literal loop only; it must be walked as far as the reference walks
    it (slam/raycast.cpp:163-220 has no sample limit), not cut after a few samples."""
    rows, cols = 6, 1 << 20
    g = np.full((rows, cols), 255, np.uint8)
    g[:, cols - 300] = 0   # a wall 300 cells inside
    sx = np.array([cols + 0.5, cols + 0.25, cols + 0.5, cols - 0.5])
    sy = np.array([2.5, 3.5, 2.5, 1.5])
    th = np.array([np.pi, np.pi, 0.0, np.pi])
    bad, first, _, _, hits, _ = compare(hh, g, sx, sy, th, 500.0)
    assert bad == 0, first
    assert abs(hits[0, 0] - (cols - 299.5)) < 1.0 and hits[2, 0] == -1.0 and hits[3, 0] > 0


def test_fast_hit_search_matches_the_oracle_walk(hh):
    """The harness' literal walk is the same source as the kernels'; tie it to the oracle as well."""
    rng = np.random.default_rng(3)
    g = (orc.load_map("floor_plan") * 255).astype(np.uint8)
    rows, cols = g.shape
    n = 3000
    sx, sy, th = rng.uniform(0, cols, n), rng.uniform(0, rows, n), rng.uniform(-4, 4, n)
    bad, _, _, _, hits, _ = compare(hh, g, sx, sy, th, 500.0)
    assert bad == 0
    for r in range(n):
        h, cell, _ = orc.raycast_mapping(g, sx[r], sy[r], th[r], 250000.0, 250000.0, write=False, math=orc.DET)
        assert h[0] == hits[r, 0] and h[1] == hits[r, 1]


def test_clearance_grid_brute_force(hh):
    rng = np.random.default_rng(4)
    for shape, dens in [((60, 70), 0.03), ((33, 17), 0.2), ((5, 300), 0.0), ((1, 1), 0.0)]:
        m = ((rng.uniform(size=shape) >= dens) * 255).astype(np.uint8)
        E = np.empty_like(m)
        hh.hh_build_clearance(ptr(m), shape[0], shape[1], ptr(E))
        occ = np.argwhere(m < 128)
        for i in range(shape[0]):
            for j in range(shape[1]):
                if m[i, j] < 128:
                    exp = 0
                else:
                    d = min(i + 1, shape[0] - i, j + 1, shape[1] - j)
                    if len(occ):
                        d = min(d, int(np.max(np.abs(occ - [i, j]), axis=1).min()))
                    exp = min(d, 255)
                assert E[i, j] == exp


def test_ray_functions_under_address_and_ub_sanitizers(hh):
    """tests/host_harness/ray_san: the fast hit search and the exact jumps built with -fsanitize=address,undefined.
    The padded clearance grid is an exactly sized heap buffer there, so a lookup outside its two-cell
    frame (a GPU memory fault in the kernel) or an out-of-range double->int conversion aborts the run."""
    exe = os.path.join(HERE, "host_harness", "ray_san")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "mismatches 0 advance_mismatches 0" in out.stdout


def test_cone_planes_equal_their_definition(hh):
    """The 8 cone-clearance planes (slam_amd/csrc/clearance.h) against a brute-force evaluation of their
    definition: smallest a such that a cell (a, b), 0 <= b <= a + 2, of the octant's cone is occupied or
    outside the grid; stored as 0 for occupied cells and the frame, else max(1, min(255, a))."""
    rng = np.random.default_rng(11)
    hh.hh_cone_plane_stride.restype = ctypes.c_ulong
    for rows, cols, density in [(23, 31, 0.03), (30, 17, 0.10), (9, 40, 0.0)]:
        g = np.where(rng.random((rows, cols)) < density, 0, 255).astype(np.uint8)
        stride = hh.hh_cone_plane_stride(rows, cols)
        assert stride % 256 == 0 and stride >= (rows + 4) * (cols + 4)
        planes = np.full(8 * stride, 7, dtype=np.uint8)
        hh.hh_build_cone_planes(ptr(g), rows, cols, ptr(planes))
        occ = lambda i, j: not (0 <= i < rows and 0 <= j < cols) or g[i, j] < 128
        for o in range(8):
            sx, sy = (-1 if o & 1 else 1), (-1 if o & 2 else 1)
            (mi, mj), (ni, nj) = ((0, sx), (-sy, 0)) if not o & 4 else ((-sy, 0), (0, sx))  # image rows go down as y goes up
            plane = planes[o * stride:o * stride + (rows + 4) * (cols + 4)].reshape(rows + 4, cols + 4)
            assert not plane[:2].any() and not plane[-2:].any() and not plane[:, :2].any() and not plane[:, -2:].any()
            for i in range(rows):
                for j in range(cols):
                    if occ(i, j):
                        want = 0
                    else:
                        a = 0
                        while not any(occ(i + a * mi + b * ni, j + a * mj + b * nj) for b in range(a + 3)):
                            a += 1
                        want = max(1, min(255, a))
                    assert plane[i + 2, j + 2] == want, (rows, cols, o, i, j)
