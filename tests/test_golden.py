"""Committed fixtures (tests/golden/oracle_vectors.npz, made by tests/golden/make_vectors.py).
CPU part: the oracle still reproduces them.  GPU part: the HIP path reproduces them through the C ABI
without consulting the oracle at run time."""
import os
import zlib

import numpy as np
import pytest

import _oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
ODOM, ALPHAS, OFFSET = [2.5, 0.02, 0.02], [5e-4, 5e-4, 1e-2, 1e-2], [0.0, 30.0, 0.0]


@pytest.fixture(scope="module")
def V():
    return np.load(os.path.join(HERE, "golden", "oracle_vectors.npz"))


def same_bits(a, b):
    a, b = np.ascontiguousarray(a, np.float64), np.ascontiguousarray(b, np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


# ---- CPU: the oracle against its frozen outputs ------------------------------------------------------
def test_oracle_tables_offsets_motion(V):
    fr, oc = orc.update_tables()
    assert np.array_equal(fr, V["g7_free"]) and np.array_equal(oc, V["g7_occ"])
    assert same_bits(np.array([orc.scanner_offset(o) for o in V["g8_off"]])[:, 0], V["g8_dtheta"][:, 0])
    assert np.allclose(np.array([orc.scanner_offset(o) for o in V["g8_off"]]), V["g8_dtheta"], rtol=1e-15)
    x, y, th = V["g1_in"]
    assert same_bits(np.stack(orc.predict(x, y, th, ODOM, ALPHAS, V["g1_z"], orc.DET)), V["g1_det_out"])
    assert np.allclose(np.stack(orc.predict(x, y, th, ODOM, ALPHAS, V["g1_z"], orc.LIBM)), V["g1_libm_out"], rtol=1e-13, atol=1e-13)
    assert np.allclose(V["g1_det_out"], V["g1_libm_out"], rtol=1e-12, atol=1e-12)
    assert same_bits(orc.reference_normals(3 * 256, 1, 1), V["g1_z"])  # libstdc++ <random> did not change


def test_oracle_single_rays(V):
    g = V["g2_canvas"].copy()
    zmax = float(V["g2_zmax"])
    for r, (px, py, t, zz) in enumerate(V["g2_rays"]):
        h, c, st = orc.raycast_mapping(g, px, py, t, min(zz, zmax) ** 2, zmax ** 2, math=orc.DET)
        assert same_bits(h, V["g2_det_hits"][r]) and c == V["g2_det_cells"][r]
        assert [st["cells_read"], st["cells_written"], st["steps"]] == list(V["g2_det_stats"][r])
    assert np.array_equal(g, V["g2_det_canvas_after"])


def test_oracle_scans_updates_resampling(V):
    for name in ("map", "floor_plan", "maze"):
        free = orc.load_map(name)
        hits = orc.fake_lidar_scan(free.astype(np.int32), V[f"g6_{name}_sensor"], 0.0, 2 * np.pi, 500.0, 90, orc.DET)
        assert same_bits(hits, V[f"g6_{name}_det_hits"])
    x, y, th = V["g3_in"]
    maps = np.full((16, 200, 200), 128, np.uint8)
    for p in range(2):
        w, _ = orc.update_weights(x, y, th, maps, V["scan_ang"], V["scan_dist"], 5.0, 500.0, OFFSET, slam_mode=True, math=orc.DET)
        assert same_bits(w, V[f"g3_det_w{p}"]) and np.array_equal([crc(m) for m in maps], V[f"g3_det_crc{p}"])
    gf = (orc.load_map("floor_plan") * 255).astype(np.uint8)
    x, y, th = V["gl_in"]
    w, hxy, hc, st = orc.update_weights(x, y, th, gf, V["gl_ang"], V["gl_dist"], 5.0, 500.0, OFFSET, want_hits=True, math=orc.DET)
    assert same_bits(w, V["gl_det_w"]) and same_bits(hxy, V["gl_det_hit_xy"]) and np.array_equal(hc, V["gl_det_hit_cell"])
    wl, _, hcl, _ = orc.update_weights(x, y, th, gf, V["gl_ang"], V["gl_dist"], 5.0, 500.0, OFFSET, want_hits=True, math=orc.LIBM)
    assert np.sum(hcl != V["gl_libm_hit_cell"]) <= 1 and np.sum(hcl != V["gl_det_hit_cell"]) <= 1
    ok = V["gl_libm_w"] > 1e-290
    assert np.allclose(wl[ok], V["gl_libm_w"][ok], rtol=1e-9)
    for name in ("generic", "zeros", "single", "repeats", "one", "big"):
        w, u = V[f"g4_{name}_w"], V[f"g4_{name}_u"]
        for order, tag in ((orc.ORDER_REFERENCE, "ref"), (orc.ORDER_BLOCKED, "blk")):
            idx, P, _ = orc.resample(w, u, order)
            assert np.array_equal(idx, V[f"g4_{name}_{tag}_idx"]) and same_bits(P, V[f"g4_{name}_{tag}_prefix"])
        assert np.array_equal(V[f"g4_{name}_ref_idx"], V[f"g4_{name}_blk_idx"])
        assert same_bits(orc.reference_uniforms(w.size, tick=w.size + 5), u)


# ---- GPU: the HIP path against the same fixtures -----------------------------------------------------
@pytest.mark.gpu
def test_gpu_reproduces_fixtures(V):
    import slam_amd as sa
    from slam_amd import capi

    # motion (G1)
    x, y, th = V["g1_in"]
    f = sa.Filter(256, 64, 64, noise=capi.NOISE_INJECTED)
    f.set_poses(x, y, th)
    f.set_noise(unit_normals=V["g1_z"])
    f.predict(ODOM, ALPHAS)
    assert same_bits(np.stack(f.poses()), V["g1_det_out"])
    assert np.allclose(np.stack(f.poses()), V["g1_libm_out"], rtol=1e-5, atol=1e-9)
    f.close()
    # localization weights + hit cells (GL), every kernel shape
    gf = (orc.load_map("floor_plan") * 255).astype(np.uint8)
    x, y, th = V["gl_in"]
    for rc in (capi.RAYCAST_FAST_BEAM_LANES, capi.RAYCAST_FAST_PARTICLE_LANES, capi.RAYCAST_FAST_PARTICLE_LANES_DIRECT, capi.RAYCAST_WALK):
        f = sa.Filter(96, *gf.shape, raycaster=rc)
        f.set_map(gf)
        f.set_poses(x, y, th)
        f.record_hits(True)
        f.update_weights(V["gl_ang"], V["gl_dist"], 5.0, 500.0, OFFSET)
        hxy, hc = f.hits()
        assert same_bits(f.weights(), V["gl_det_w"]) and same_bits(hxy, V["gl_det_hit_xy"]) and np.array_equal(hc, V["gl_det_hit_cell"])
        assert np.sum(hc != V["gl_libm_hit_cell"]) <= 1
        ok = V["gl_libm_w"] > 1e-290
        assert np.allclose(f.weights()[ok], V["gl_libm_w"][ok], rtol=1e-5)
        if rc == capi.RAYCAST_WALK:
            st = f.stats()
            assert [st["cells_read"], st["cells_written"], st["steps"]] == list(V["gl_det_stats"])
        f.close()
    # SLAM-mode update (G3): weights and every particle's canvas
    x, y, th = V["g3_in"]
    f = sa.Filter(16, 200, 200, mode=capi.MODE_SLAM)
    f.set_poses(x, y, th)
    for p in range(2):
        f.update_weights(V["scan_ang"], V["scan_dist"], 5.0, 500.0, OFFSET)
        assert same_bits(f.weights(), V[f"g3_det_w{p}"])
        assert np.array_equal([crc(f.map_of(k)) for k in range(16)], V[f"g3_det_crc{p}"])
    f.close()
    # single rays on an evolving canvas (G2): one particle, one beam per update
    g = V["g2_canvas"]
    zmax = float(V["g2_zmax"])
    f = sa.Filter(1, *g.shape, mode=capi.MODE_SLAM)
    f.set_maps(g[None])
    f.record_hits(True)
    for r, (px, py, t, zz) in enumerate(V["g2_rays"]):
        f.set_poses([px], [py], [t])
        f.update_weights([0.0], [min(zz, zmax)], 5.0, zmax, [0.0, 0.0, 0.0])
        hxy, hc = f.hits()
        assert same_bits(hxy[0, 0], V["g2_det_hits"][r]) and hc[0, 0] == V["g2_det_cells"][r], r
        st = f.stats()
        assert [st["cells_read"], st["cells_written"], st["steps"]] == list(V["g2_det_stats"][r]), r
    assert np.array_equal(f.map_of(0), V["g2_det_canvas_after"])
    f.close()
    # resampling (G4)
    for name in ("generic", "zeros", "single", "repeats", "one", "big"):
        w, u = V[f"g4_{name}_w"], V[f"g4_{name}_u"]
        for po, tag in ((capi.PREFIX_REFERENCE, "ref"), (capi.PREFIX_BLOCKED, "blk")):
            f = sa.Filter(w.size, 8, 8, noise=capi.NOISE_INJECTED, prefix_order=po)
            f.set_weights(w)
            f.set_noise(canonical_uniforms=u)
            f.resample()
            assert np.array_equal(f.last_indices(), V[f"g4_{name}_{tag}_idx"]), (name, tag)
            assert same_bits(f.prefix()[0], V[f"g4_{name}_{tag}_prefix"])
            if tag == "blk":
                assert np.array_equal(f.last_order(), V[f"g4_{name}_perm"])
            f.close()
    # end-to-end SLAM loop (G5)
    N, rows, cols = 25, 180, 240
    mcl = sa.MCL(N, (cols, rows), noise=capi.NOISE_INJECTED)
    f = mcl.filter
    for step in range(5):
        f.set_noise(orc.reference_normals(3 * N, tick0=50 * step + 1, tick_stride=1), orc.reference_uniforms(N, tick=900 + step))
        mcl.predict(sa.Odometry(*ODOM), ALPHAS)
        mcl.update(list(zip(V["g5_ang"], V["g5_dist"])), 5.0, 500.0, sa.Pose(*OFFSET))
        assert np.array_equal(f.last_indices(), V[f"g5_det_idx{step}"])
        assert same_bits(np.stack(list(f.poses()) + [f.weights()]), V[f"g5_det_state{step}"])
        assert np.array_equal([crc(f.map_of(k)) for k in range(N)], V[f"g5_det_crc{step}"])
    f.close()
    # FakeLidar on the device (G6)
    lib = capi.load_library()
    for name in ("map", "floor_plan", "maze"):
        free = orc.load_map(name).astype(np.uint8)
        pose = np.ascontiguousarray(V[f"g6_{name}_sensor"][None])
        hits = np.empty((90, 2))
        capi.check(lib.slam_fake_lidar_scan(0, capi.ptr(free), free.shape[0], free.shape[1], capi.ptr(pose), 1, 0.0, 2 * np.pi,
                                            500.0, 90, capi.ptr(hits)))
        assert same_bits(hits, V[f"g6_{name}_det_hits"])
