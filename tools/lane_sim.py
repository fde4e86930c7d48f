"""CPU design study for the wave-per-particle localization kernel on a wide cloud (BASELINE configs[4] share): clearance
lookups per ray of the real search (tests/host_harness), then the number of lock-step lookup iterations a wavefront pays
under different assignments of beams to lanes.  Counting only."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc
from slam_amd import synth
hh = ctypes.CDLL(os.path.join(ROOT, "tests", "host_harness", "libray_harness.so"))
hh.hh_cone_plane_stride.restype = ctypes.c_ulong
size = int(os.environ.get("SIZE", 16384)); B = int(os.environ.get("B", 720)); zmax = float(os.environ.get("ZMAX", 4000)); NP = int(os.environ.get("NP", 1500))
g = synth.make_grid(size)
stride = hh.hh_cone_plane_stride(size, size)
planes = np.empty(8 * stride, np.uint8)
hh.hh_build_cone_planes(g.ctypes.data_as(ctypes.c_void_p), size, size, planes.ctypes.data_as(ctypes.c_void_p))
true = synth.open_pose(g)
off = [0.0, 30.0, 0.0]
sensor = orc.sensor_position(true, off, orc.DET)
free01 = (g >= 128).astype(np.int32)
hits = orc.fake_lidar_scan(free01, sensor, 0.0, 2 * np.pi, zmax, B, orc.DET)
ang, dist = orc.hits_to_scan(hits, sensor, 0.0, 2 * np.pi, zmax)
x, y, th = orc.seed_uniform_free(1, 0, NP, g)
so = orc.scanner_offset(off)
reads = np.empty((NP, B), np.int32)
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
hh.hh_pass_reads(P(planes), size, size, ctypes.c_long(NP), P(x), P(y), P(th), B, P(ang), P(dist), ctypes.c_double(zmax), ctypes.c_double(0.5),
                 ctypes.c_double(so[0]), ctypes.c_double(so[1]), ctypes.c_double(off[2]), P(reads))
np.save("/tmp/lane_sim_reads.npy", reads)
trips = np.maximum(reads - 1, 0)   # the first read is the start cell's (before the loop)
print("scan: mean range %.0f median %.0f; lookups per ray mean %.2f, p50 %d p90 %d p99 %d max %d" % (dist.mean(), np.median(dist), reads.mean(), *np.percentile(reads, [50, 90, 99]), reads.max()))

def lockstep(order):
    """beams dealt to groups of 64 in `order`; a group costs its longest ray"""
    t = trips[:, order]
    pad = (-B) % 64
    t = np.pad(t, ((0, 0), (0, pad)))
    return t.reshape(NP, -1, 64).max(axis=2).sum(axis=1).mean()

ideal = trips.sum(axis=1).mean() / 64
print("ideal (perfect packing): %.1f iterations per particle" % ideal)
print("natural order, groups of 64: %.1f" % lockstep(np.arange(B)))
print("beams sorted by range:      %.1f" % lockstep(np.argsort(dist, kind="stable")))
# persistent lanes with batched refill: a lane that finishes parks; when >= T lanes are parked (or nothing runs) the parked
# lanes take the next beams (one heavy block execution); counts lookup iterations and heavy-block executions
def persistent(T, order=None):
    it_tot = heavy_tot = 0
    for p in range(min(NP, 300)):
        t = trips[p] if order is None else trips[p][order]
        nxt = 0; cur = np.zeros(64, np.int64); act = np.zeros(64, bool); its = heavy = 0
        while True:
            parked = ~act
            if nxt < B and (parked.sum() >= T or not act.any()):
                k = min(int(parked.sum()), B - nxt)
                idx = np.flatnonzero(parked)[:k]
                cur[idx] = t[nxt:nxt + k]; act[idx] = True; nxt += k; heavy += 1
                done = act & (cur == 0); act[done] = False
                continue
            if not act.any():
                break
            its += 1
            cur[act] -= 1
            act &= cur > 0
        it_tot += its; heavy_tot += heavy
    n = min(NP, 300)
    return it_tot / n, heavy_tot / n
for T in (64, 48, 32, 16, 8):
    a, b = persistent(T)
    c, d = persistent(T, np.argsort(dist, kind="stable"))
    print("persistent lanes, refill at %2d parked: %.1f lookup iterations, %.1f heavy blocks | sorted by range: %.1f, %.1f" % (T, a, b, c, d))
desc = np.argsort(-dist, kind="stable")
for T in (48, 32, 16):
    c, d = persistent(T, desc)
    print("persistent lanes, refill at %2d parked, beams by range DESCENDING: %.1f lookup iterations, %.1f heavy blocks" % (T, c, d))
print("lock-step, descending: %.1f" % lockstep(desc))
