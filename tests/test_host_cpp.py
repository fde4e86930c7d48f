"""The C++ drop-in shim (slam_amd/host): slam::MCL / FakeLidar / motion with the reference's signatures,
driven by the reference benchmark's call sequence (tests/host_cpp/mcl_like_benchmark.cpp mirrors
benchmark/mcl.cpp:14-64)."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(HERE, "host_cpp", "mcl_like_benchmark")


def test_shim_builds_against_the_c_abi_only():
    subprocess.run(["make", "-C", os.path.join(HERE, "host_cpp"), "-s"], check=True)
    assert os.path.exists(EXE)
    src = open(os.path.join(HERE, "..", "slam_amd", "host", "src", "slam_host.cpp")).read()
    assert "hip_runtime" not in src and "slam_hip.h" in src  # host C++ reaches HIP only through the C ABI


def test_shim_headers_coexist_with_the_headers_of_a_tree_behind_them():
    """inside the reference's tree the shim's pose.h / util.h hand over to the tree's own
    (SLAM_REFERENCE_POSE_H / SLAM_REFERENCE_UTIL_H), so one translation unit can use the planners' types
    (Coordinate, Box, helpers only the tree's util.h has) together with slam::MCL, and slam_host.cpp leaves
    the util functions to the tree's util.cpp.  tests/host_cpp/tree_like_reference/ stands for that tree
    (headers written for this test: the reference's own need OpenCV, absent here)."""
    root = os.path.join(HERE, "..")
    tree = os.path.join(HERE, "host_cpp", "tree_like_reference", "slam")
    flags = ["-std=gnu++17", "-Wall", "-Werror", "-c", "-I", os.path.join(root, "slam_amd", "host", "include", "slam"),
             "-I", os.path.join(root, "include"), "-I", tree,
             f'-DSLAM_REFERENCE_POSE_H="{os.path.join(tree, "pose.h")}"', f'-DSLAM_REFERENCE_UTIL_H="{os.path.join(tree, "util.h")}"']
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        for cxx in ("g++", "/opt/rocm/lib/llvm/bin/clang++"):
            subprocess.run([cxx] + flags + [os.path.join(tree, "..", "tu.cpp"), "-o", os.path.join(d, "tu.o")], check=True)
            obj = os.path.join(d, "host.o")
            subprocess.run([cxx] + flags + [os.path.join(root, "slam_amd", "host", "src", "slam_host.cpp"), "-o", obj], check=True)
            syms = subprocess.run(["nm", "-C", obj], check=True, capture_output=True, text=True).stdout
            assert " T slam::MCL::update(" in syms and " T slam::raycast_mapping(" in syms and " T slam::measurement_model_beam(" in syms
            assert " T slam::normalize_angle(" not in syms and " T slam::average_pose(" not in syms  # the tree's util.cpp has them
        # on its own the shim defines them itself
        obj = os.path.join(d, "alone.o")
        subprocess.run(["g++", "-std=gnu++17", "-Wall", "-c", "-I", os.path.join(root, "slam_amd", "host", "include", "slam"),
                        "-I", os.path.join(root, "include"), os.path.join(root, "slam_amd", "host", "src", "slam_host.cpp"), "-o", obj],
                       check=True)
        syms = subprocess.run(["nm", "-C", obj], check=True, capture_output=True, text=True).stdout
        assert " T slam::normalize_angle(" in syms and " T slam::average_pose(" in syms


@pytest.mark.gpu
@pytest.mark.parametrize("storage", ["dense", "tiled"])
def test_cpp_shim_matches_python_path(tmp_path, storage):
    import slam_amd as sa
    from slam_amd import capi

    free = orc.load_map("floor_plan").astype(np.int32)
    raw = tmp_path / "map.raw"
    free.tofile(raw)
    iters, seed = 4, 0x1234ABCD
    env = dict(os.environ, SLAM_SEED=str(seed), SLAM_MAPS=storage)   # the shim's storage must not change any result
    out = subprocess.run([EXE, str(raw), str(free.shape[0]), str(free.shape[1]), str(iters)], check=True, env=env,
                         capture_output=True, text=True, timeout=300).stdout.splitlines()
    parts = np.array([[int(t, 16) for t in ln.split()[1:]] for ln in out if ln.startswith("P ")], dtype=np.uint64)
    avg = np.array([float(t) for t in next(ln for ln in out if ln.startswith("AVG")).split()[1:]])
    # the same sequence through the Python mirror
    off = sa.Pose(0, 30, 0)
    sensor = sa.MCL.sensor_position(sa.Pose(400, 400, np.pi), off)
    lib = capi.load_library()
    hits = np.empty((90, 2))
    m01 = free.astype(np.uint8)
    pose = np.array([[sensor.x, sensor.y, sensor.theta]])
    capi.check(lib.slam_fake_lidar_scan(0, capi.ptr(m01), m01.shape[0], m01.shape[1], capi.ptr(pose), 1, 0.0, np.pi, 500.0, 90,
                                        capi.ptr(hits)))
    step = np.pi / 90
    scan = [(i * step - np.pi / 2,
             float(np.sqrt((hits[i, 1] - sensor.y) ** 2 + (hits[i, 0] - sensor.x) ** 2)) if hits[i, 0] != -1 else 500.0)
            for i in range(90)]
    mcl = sa.MCL(25, (1600, 900), seed=seed)
    for _ in range(iters):
        mcl.predict(sa.Odometry(2.5, 0.02, 0.02), [0.0005, 0.0005, 0.01, 0.01])
        mcl.update(scan, 5.0, 500.0, off)
    f = mcl.filter
    x, y, th = f.poses()
    ref = np.stack([x, y, th, f.weights()], 1).view(np.uint64)
    assert parts.shape == (25, 4) and np.array_equal(parts, ref)
    ap = f.average_pose()
    assert np.allclose(avg, [ap.x, ap.y, ap.theta], rtol=1e-12)
    best = f.map_of(0)
    rows, cols, mh = next(ln for ln in out if ln.startswith("MAP")).split()[1:]
    assert (int(rows), int(cols)) == best.shape
    hh = 0  # the same multiplicative hash of the best particle's canvas as the C++ program prints
    for v in best.ravel().tolist():
        hh = (hh * 1099511628211 + v) & 0xFFFFFFFFFFFFFFFF
    assert int(mh, 16) == hh
    f.close()
    # free functions of the shim against the oracle's restatements
    vel = [float(t) for t in next(ln for ln in out if ln.startswith("VEL")).split()[1:]]
    ex = orc.predict_velocity([10.0], [20.0], [0.3], [2.0, 0.5], 0.25, [0.0] * 6, [0.0, 0.0, 0.0])
    assert vel == [ex[0][0], ex[1][0], ex[2][0]]
    pdf = [float(t) for t in next(ln for ln in out if ln.startswith("PDF")).split()[1:]]
    assert abs(pdf[0] - 0.18762017345846896) < 1e-15 and pdf[1] == 0.0 and pdf[2] == orc.pdf_triangular(1.5, [0.5])[0]
    assert next(ln for ln in out if ln.startswith("TRI")).split()[1] == "1"
    # raycast_mapping / measurement_model_beam of the shim (slam/raycast.h:19-28) against the oracle: same call sequence
    got = [float(t) for t in next(ln for ln in out if ln.startswith("RCM ")).split()[1:]]
    canvas = np.full((120, 160), 128, np.uint8)
    want, want_libm = [], []
    canvas_libm = canvas.copy()
    for rnd in range(2):
        for k in range(12):
            th, z = 0.3 + 0.5 * k, 20.0 + 3.0 * k
            for cv_, acc, math in ((canvas, want, orc.DET), (canvas_libm, want_libm, orc.LIBM)):
                if k % 3 == 2:
                    acc.append(orc.measurement_model_beam(cv_, 60.5, 40.25, th, z * z, 2.0, 70.0 * 70.0, math=math))
                else:
                    hit, _, _ = orc.raycast_mapping(cv_, 60.5, 40.25, th, 70.0 * 70.0 if k % 4 == 1 else z * z, 70.0 * 70.0, math=math)
                    acc.extend(hit.tolist())
    # hit positions are bit-exact against the oracle's detmath mode; the likelihoods (host libm in the shim) to 1e-12
    assert len(got) == len(want) and np.allclose(got, want, rtol=1e-12, atol=0)
    assert np.allclose(got, want_libm, rtol=1e-5)
    hh = 0
    for v in canvas.ravel().tolist():
        hh = (hh * 1099511628211 + v) & 0xFFFFFFFFFFFFFFFF
    assert int(next(ln for ln in out if ln.startswith("RCMMAP")).split()[1], 16) == hh
    assert (canvas != 128).sum() > 300 and np.array_equal(canvas, canvas_libm)


@pytest.mark.gpu
@pytest.mark.parametrize("storage", ["dense", "tiled"])
def test_cpp_shim_under_host_sanitizers(tmp_path, storage):
    """The same call sequence against a build of the library whose host code is instrumented with
    AddressSanitizer and UndefinedBehaviorSanitizer (`make -C tests/host_cpp san`, done by build()):
    no report, and the same particles as the plain build."""
    san = os.path.join(HERE, "host_cpp", "san", "mcl_san")
    if not os.path.exists(san):
        pytest.skip("sanitized build absent (make -C tests/host_cpp san)")
    free = orc.load_map("floor_plan").astype(np.int32)
    raw = tmp_path / "map.raw"
    free.tofile(raw)
    env = dict(os.environ, SLAM_SEED="4660", SLAM_MAPS=storage,
               ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1")
    args = [str(raw), str(free.shape[0]), str(free.shape[1]), "4"]
    got = subprocess.run([san] + args, env=env, capture_output=True, text=True, timeout=300)
    assert got.returncode == 0, got.stderr[-2000:]
    assert "ERROR" not in got.stderr and "runtime error" not in got.stderr, got.stderr[-2000:]
    plain = subprocess.run([EXE] + args, env=env, capture_output=True, text=True, timeout=300, check=True)
    pick = lambda out: [ln for ln in out.splitlines() if ln.startswith(("P ", "AVG", "MAP"))]
    assert len(pick(got.stdout)) == 27 and pick(got.stdout) == pick(plain.stdout)


@pytest.mark.gpu
def test_detmath_on_the_device_equals_the_host_bit_for_bit():
    """tests/host_cpp/detmath_devcheck.hip: dm_exp, dm_log, dm_sincos (table driven since round 3) and dm_sqrt (the compiler's expansion on the device) evaluated by a kernel and by
    the host on 4 M arguments (likelihood range, the whole exponent range, near 1): the same bits, which is what lets the CPU
    oracle stand in for the kernels' arithmetic."""
    exe = os.path.join(HERE, "host_cpp", "detmath_devcheck")
    if not os.path.exists(exe):
        pytest.skip("detmath_devcheck absent (make -C tests/host_cpp detmath_devcheck)")
    got = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert got.returncode == 0 and "exp differs 0, log differs 0, sincos differs 0, sqrt differs 0" in got.stdout, got.stdout + got.stderr


@pytest.mark.gpu
def test_abi_tour_under_host_sanitizers():
    """tests/host_cpp/abi_tour.hip: every entry point of include/slam_hip.h called from C++ (both modes,
    every kernel shape and option, two shards emulated on one device in all three exchange forms, the
    documented error returns), against the ASan + UBSan host build of the library.  It checks sharded
    against unsharded results itself; here: exit 0, its last line, no sanitizer report."""
    exe = os.path.join(HERE, "host_cpp", "san", "abi_tour")
    if not os.path.exists(exe):
        pytest.skip("sanitized build absent (make -C tests/host_cpp san)")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0", UBSAN_OPTIONS="print_stacktrace=1")
    got = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    assert got.returncode == 0, (got.stdout + got.stderr)[-2000:]
    assert got.stdout.strip().endswith("abi_tour ok")
    assert "ERROR" not in got.stderr and "runtime error" not in got.stderr, got.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["single", "split"])
def test_cpp_rccl_shard_driver_matches_one_handle(tmp_path, exchange):
    """slam::ShardedMCL (slam_amd/host/src/slam_shard_rccl.cpp): the sharded step driven from C++ with RCCL
    called directly.  The test box has one GPU, so the communicator has one rank (the all-gather still goes
    through RCCL); the particles must equal those of one plain handle, bit for bit."""
    import slam_amd as sa

    exe = os.path.join(HERE, "host_cpp", "shard_rccl")
    if not os.path.exists(exe):
        pytest.skip("shard_rccl absent (make -C tests/host_cpp shard_rccl)")
    g = (orc.load_map("floor_plan") * 255).astype(np.uint8)
    raw = tmp_path / "map.u8"
    g.tofile(raw)
    n, steps = 4096, 3
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    args = [exe, str(raw), str(g.shape[0]), str(g.shape[1]), str(n), str(steps)] + (["split"] if exchange == "split" else [])
    got = subprocess.run(args, env=env, capture_output=True, text=True, timeout=300)
    assert got.returncode == 0, (got.stdout[-500:] + got.stderr[-2000:])
    parts = np.array([[int(t, 16) for t in ln.split()[1:]] for ln in got.stdout.splitlines() if ln.startswith("P ")], dtype=np.uint64)
    f = sa.Filter(n, *g.shape, seed=0x5EED0001, sort=sa.capi.SORT_NONE)
    f.set_map(g)
    f.seed_gaussian([520.0, 280.0, 0.4], [25.0, 25.0, 0.2])
    ang = np.arange(90) * (2 * np.pi / 90) - np.pi
    dist = 40.0 + 3.0 * (np.arange(90) % 17)
    for _ in range(steps):
        f.predict([2.5, 0.02, 0.02], [0.0005, 0.0005, 0.01, 0.01])
        f.update(ang, dist, 5.0, 500.0, [0.0, 30.0, 0.0])
    x, y, th = f.poses()
    ref = np.stack([x, y, th, f.weights()], 1).view(np.uint64)
    f.close()
    assert parts.shape == (n, 4) and np.array_equal(parts, ref)
    assert np.unique(parts[:, 0]).size > 10  # a real cloud, not one collapsed particle


def canvas_hash(m):
    """hash_bytes of tests/host_cpp/shard_rccl.cpp."""
    v = m.reshape(-1).astype(np.uint64)
    with np.errstate(over="ignore"):
        return int(((v + np.uint64(1)) * (np.arange(v.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345))).sum(dtype=np.uint64))


@pytest.mark.gpu
def test_cpp_rccl_shard_driver_read_out(tmp_path):
    """slam::ShardedMCL::average_pose() / best(): slam::average_pose and get_particles().front() across ranks
    (slam/util.cpp:66-85, apps/grid_slam.cpp:112-119), here through a one-rank communicator: against the oracle on the
    particles the program printed."""
    exe = os.path.join(HERE, "host_cpp", "shard_rccl")
    if not os.path.exists(exe):
        pytest.skip("shard_rccl absent (make -C tests/host_cpp shard_rccl)")
    g = (orc.load_map("floor_plan") * 255).astype(np.uint8)
    raw = tmp_path / "map.u8"
    g.tofile(raw)
    n = 3000
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    got = subprocess.run([exe, str(raw), str(g.shape[0]), str(g.shape[1]), str(n), "2"], env=env, capture_output=True, text=True, timeout=300)
    assert got.returncode == 0, (got.stdout[-500:] + got.stderr[-2000:])
    lines = got.stdout.splitlines()
    parts = np.array([[int(t, 16) for t in ln.split()[1:]] for ln in lines if ln.startswith("P ")], dtype=np.uint64).view(np.float64)
    avg = [float(t) for t in next(ln for ln in lines if ln.startswith("AVG")).split()[1:]]
    best = next(ln for ln in lines if ln.startswith("BEST ")).split()[1:]
    assert np.allclose(avg, orc.average_pose(parts[:, 0], parts[:, 1], parts[:, 2], math=orc.DET), rtol=1e-12, atol=0)
    k = int(np.flatnonzero(parts[:, 3] == parts[:, 3].max())[0])
    assert (float(best[0]), int(best[1]), int(best[2])) == (parts[k, 3], k, 0)
    assert [float(t) for t in best[3:]] == parts[k, :3].tolist()


@pytest.mark.gpu
def test_cpp_rccl_shard_driver_slam_mode_matches_one_tiled_handle():
    """SLAM mode of slam::ShardedMCL (BASELINE configs[3] from the reference's own language): select, request counts and
    indices, tables + distinct tiles through grouped ncclSend / ncclRecv, import (slam/mcl.cpp:147-155,205-227 across
    ranks).  One-rank communicator (the sends and receives still go through RCCL) against a plain tiled handle: particles,
    EVERY canvas, the best particle with its canvas and the mean pose."""
    import slam_amd as sa

    exe = os.path.join(HERE, "host_cpp", "shard_rccl")
    if not os.path.exists(exe):
        pytest.skip("shard_rccl absent (make -C tests/host_cpp shard_rccl)")
    n, rows, cols, steps = 600, 200, 264, 4
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    got = subprocess.run([exe, "slam", str(rows), str(cols), str(n), str(steps)], env=env, capture_output=True, text=True, timeout=300)
    assert got.returncode == 0, (got.stdout[-500:] + got.stderr[-2000:])
    lines = got.stdout.splitlines()
    parts = np.array([[int(t, 16) for t in ln.split()[1:]] for ln in lines if ln.startswith("P ")], dtype=np.uint64)
    hashes = [int(ln.split()[2], 16) for ln in lines if ln.startswith("M ")]
    f = sa.Filter(n, rows, cols, mode=sa.capi.MODE_SLAM, seed=0x5EED0004, sort=sa.capi.SORT_NONE, map_storage=sa.capi.MAPS_TILED)
    for k in range(steps):
        b = np.arange(24)
        ang, dist = b * (2 * np.pi / 24) - np.pi, 12.0 + 2.5 * ((b + 3 * k) % 11)
        f.predict([0.0, 1.5, 0.01], [0.001] * 4)
        f.update(ang, dist, 2.0, 40.0, [0.0, 6.0, 0.0])
    x, y, th = f.poses()
    w = f.weights()
    ref = np.stack([x, y, th, w], 1).view(np.uint64)
    assert parts.shape == (n, 4) and np.array_equal(parts, ref)
    maps = [f.map_of(k) for k in range(n)]
    assert hashes == [canvas_hash(m) for m in maps]
    assert len(set(hashes)) > 5 and np.unique(parts[:, 0]).size > 5     # canvases and particles do differ
    k = int(np.flatnonzero(w == w.max())[0])
    best = next(ln for ln in lines if ln.startswith("BEST ")).split()[1:]
    assert (float(best[0]), int(best[1]), int(best[2])) == (w[k], k, 0) and [float(t) for t in best[3:]] == [x[k], y[k], th[k]]
    assert int(next(ln for ln in lines if ln.startswith("BESTMAP")).split()[1], 16) == canvas_hash(maps[k])
    avg = [float(t) for t in next(ln for ln in lines if ln.startswith("AVG")).split()[1:]]
    assert np.allclose(avg, orc.average_pose(x, y, th, math=orc.DET), rtol=1e-12, atol=0)
    f.close()
