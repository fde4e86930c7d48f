"""slam_resample_select_static — the one entry point that allocates nothing (include/slam_hip.h).

It has no counterpart in the reference; it is the selection half of slam_mcl_resample
(slam/mcl.cpp:157-203) on caller-owned device buffers, with the running-sum levels in a static device
array of the library.  Checked here, through the C ABI, against the existing entry point
(slam_mcl_resample + slam_mcl_get_last_indices) on the same weights and uniforms: indices are integer
work, so the bar is bit-exact.  Two calls in a row, the second with a SMALLER input (the static array
keeps the first call's contents), then the Philox form and the systematic rule, then the size limit.
Collected last on purpose (file name): the round's end runs pytest -x.
"""
import ctypes

import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu


def handle_indices(sa, w, u, resampler, seed=1, injected=True):
    n = len(w)
    f = sa.Filter(n, 16, 16, noise=sa.capi.NOISE_INJECTED if injected else sa.capi.NOISE_PHILOX, resampler=resampler,
                  sort=sa.capi.SORT_NONE, seed=seed)
    f.set_weights(w)
    if injected:
        f.set_noise(canonical_uniforms=u)
    f.resample()
    idx = f.last_indices()
    f.close()
    return idx


def static_indices(sa, torch, w, u, resampler, seed=0, step=0, stream=None):
    lib = sa.capi.load_library()
    n = len(w)
    wd = torch.from_numpy(np.ascontiguousarray(w)).cuda()
    ud = torch.from_numpy(np.ascontiguousarray(u)).cuda() if u is not None else None
    out = torch.full((n + 8,), -7, dtype=torch.int32, device="cuda")  # 8 guard entries behind the output
    torch.cuda.synchronize()
    st = stream.cuda_stream if stream is not None else None
    # "allocates nothing" is measured at the DRIVER: hipMemGetInfo's free bytes see a hipMalloc / hipFree made inside
    # libslam_hip.so, which torch's caching-allocator statistics cannot (slam_static_init ran before: the first call
    # on a device may load the code object that holds the static array)
    free_before = torch.cuda.mem_get_info()[0]
    before = torch.cuda.memory_allocated()
    rc = lib.slam_resample_select_static(0, ctypes.c_void_p(st), ctypes.c_void_p(wd.data_ptr()), n, resampler,
                                         ctypes.c_void_p(ud.data_ptr()) if ud is not None else None, seed, step,
                                         ctypes.c_void_p(out.data_ptr()))
    sa.capi.check(rc)
    assert torch.cuda.memory_allocated() == before
    assert torch.cuda.mem_get_info()[0] == free_before, "device memory was allocated or freed inside the call"
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "device memory was allocated or freed by the call's kernels"
    got = out.cpu().numpy()
    assert np.all(got[n:] == -7), "wrote past idx_dev[n]"
    return got[:n]


def test_select_static_matches_resample_entry_point():
    import torch

    import slam_amd as sa

    lib = sa.capi.load_library()
    sa.capi.check(lib.slam_static_init(0))
    rng = np.random.default_rng(2)
    # first call: large (three levels of the hierarchy); second call: smaller (two levels) — it must not
    # see anything the first call left in the static array; third: a single chunk
    for n in (100_000, 3_000, 50, 262_144, 1):
        w = rng.uniform(0, 1, n) ** 6
        w[rng.uniform(size=n) < 0.15] = 0.0
        if n > 10:
            w[:5] = 0.0
        u = orc.reference_uniforms(n, tick=n + 5)
        u[n // 2] = 0.0
        want = handle_indices(sa, w, u, sa.capi.RESAMPLE_MULTINOMIAL)
        got = static_indices(sa, torch, w, u, sa.capi.RESAMPLE_MULTINOMIAL)
        assert np.array_equal(got, want), f"n={n}: {np.count_nonzero(got != want)} indices differ"
        assert np.array_equal(got, orc.resample(w, u, orc.ORDER_BLOCKED)[0])  # and the oracle agrees
    # all-zero weights: every draw selects (n-1)/2 (SURVEY.md hard part 3)
    assert np.all(static_indices(sa, torch, np.zeros(999), orc.reference_uniforms(999, 3), 0) == 499)
    # systematic rule, and a non-default stream
    n = 5000
    w = rng.uniform(0, 1, n)
    u = np.array([0.3721])
    want = handle_indices(sa, w, np.concatenate([u, np.zeros(n - 1)]), sa.capi.RESAMPLE_SYSTEMATIC)
    s = torch.cuda.Stream()
    got = static_indices(sa, torch, w, u, sa.capi.RESAMPLE_SYSTEMATIC, stream=s)
    assert np.array_equal(got, want)
    # Philox form: the stream a handle with the same seed draws at its first resample (step 0)
    n = 20_000
    w = rng.uniform(0, 1, n) ** 3
    want = handle_indices(sa, w, None, sa.capi.RESAMPLE_MULTINOMIAL, seed=0xC0FFEE, injected=False)
    got = static_indices(sa, torch, w, None, sa.capi.RESAMPLE_MULTINOMIAL, seed=0xC0FFEE, step=0)
    assert np.array_equal(got, want)


def test_select_static_rejects_sizes_beyond_the_static_array():
    import torch

    import slam_amd as sa

    lib = sa.capi.load_library()
    n = sa.capi.SELECT_STATIC_MAX + 1
    wd = torch.ones(n, dtype=torch.float64, device="cuda")
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    rc = lib.slam_resample_select_static(0, None, ctypes.c_void_p(wd.data_ptr()), n, 0, ctypes.c_void_p(wd.data_ptr()), 0, 0,
                                         ctypes.c_void_p(out.data_ptr()))
    assert rc == -1 and b"SLAM_SELECT_STATIC_MAX" in lib.slam_last_error()
    assert lib.slam_resample_select_static(0, None, ctypes.c_void_p(wd.data_ptr()), 0, 0, None, 0, 0, ctypes.c_void_p(out.data_ptr())) == -1
    assert lib.slam_resample_select_static(0, None, None, 5, 0, None, 0, 0, ctypes.c_void_p(out.data_ptr())) == -1


def test_select_static_under_hip_graph_capture():
    """The caller the entry point exists for: the call recorded into a HIP graph (hipStreamBeginCapture through
    torch.cuda.graph) and replayed.  An allocation, a free or a synchronisation inside the call would fail the capture;
    the replayed indices must equal the eager ones, also after the weights behind the captured pointers changed."""
    import torch

    import slam_amd as sa

    lib = sa.capi.load_library()
    sa.capi.check(lib.slam_static_init(0))
    rng = np.random.default_rng(9)
    n = 70_000
    w = rng.uniform(0, 1, n) ** 6
    u = orc.reference_uniforms(n, tick=11)
    wd = torch.from_numpy(w).cuda()
    ud = torch.from_numpy(u).cuda()
    out = torch.full((n,), -7, dtype=torch.int32, device="cuda")

    def call(stream):
        sa.capi.check(lib.slam_resample_select_static(0, ctypes.c_void_p(stream.cuda_stream), ctypes.c_void_p(wd.data_ptr()), n, 0,
                                                      ctypes.c_void_p(ud.data_ptr()), 0, 0, ctypes.c_void_p(out.data_ptr())))

    side = torch.cuda.Stream()
    call(side)                                   # eager, on the stream the capture will use
    side.synchronize()
    eager = out.cpu().numpy().copy()
    assert np.array_equal(eager, orc.resample(w, u, orc.ORDER_BLOCKED)[0])
    out.fill_(-7)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):   # capture_error_mode "global": any prohibited call made meanwhile fails
        call(torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert np.all(out.cpu().numpy() == -7), "a captured call must not run"
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), eager)
    # new weights behind the same pointers, replayed: the graph holds no stale scratch contents
    w2 = rng.uniform(0, 1, n) ** 2
    wd.copy_(torch.from_numpy(w2))
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), orc.resample(w2, u, orc.ORDER_BLOCKED)[0])
