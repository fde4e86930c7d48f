"""CPU design study: lookups per ray of the clearance search with (a) the Chebyshev clearance the kernels use and
(b) a per-octant cone clearance (distance, along the octant's major axis, to the nearest occupied cell inside the
octant's cone dilated by one cell).  Positions are x0 + k*dx (counting only)."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
d = os.path.join(ROOT, "tests", "host_harness")
subprocess.run(["make", "-C", d, "-s"], check=True)
hh = ctypes.CDLL(os.path.join(d, "libray_harness.so"))
name = os.environ.get("MAP", "floor_plan")
g = (bench.load_free_grid(name) * 255).astype(np.uint8)
rows, cols = g.shape
E = np.empty_like(g)
hh.hh_build_clearance(g.ctypes.data_as(ctypes.c_void_p), rows, cols, E.ctypes.data_as(ctypes.c_void_p))
PAD = 520
occ = np.ones((rows + 2 * PAD, cols + 2 * PAD), bool)   # outside = occupied
occ[PAD:PAD + rows, PAD:PAD + cols] = g < 128
Ep = np.zeros(occ.shape, np.uint8); Ep[PAD:PAD + rows, PAD:PAD + cols] = E

def cone_plane(occ, major, minor):
    """major, minor: unit steps (di, dj) of the octant's major axis and of the side the cone opens to.
    C(c) = min a >= 0 such that a cell (c + a*major + b*minor), 0 <= b <= a + 1, is occupied; also b = -1?  no: the
    ray's minor coordinate moves monotonically away from the start, never to the other side."""
    R, Cc = occ.shape
    # up[c] = distance (in minor steps, >= 0) to the nearest occupied cell at or beyond c in the minor direction
    up = np.full(occ.shape, 10000, np.int32)
    # scan along minor direction
    if minor[0] != 0:
        rng_i = range(R - 1, -1, -1) if minor[0] > 0 else range(R)
        prev = np.full(Cc, 10000, np.int32)
        for i in rng_i:
            prev = np.where(occ[i], 0, np.minimum(prev + 1, 10000))
            up[i] = prev
    else:
        rng_j = range(Cc - 1, -1, -1) if minor[1] > 0 else range(Cc)
        prev = np.full(R, 10000, np.int32)
        for j in rng_j:
            prev = np.where(occ[:, j], 0, np.minimum(prev + 1, 10000))
            up[:, j] = prev
    out = np.full(occ.shape, 255, np.int32)
    found = np.zeros(occ.shape, bool)
    for a in range(0, 255):
        sh = np.roll(up, (-a * major[0], -a * major[1]), axis=(0, 1))   # up at c + a*major  (wrap lands in the occupied frame)
        hit = (~found) & (sh <= a + int(os.environ.get("WIDE", "1")))
        out[hit] = a
        found |= hit
    out = np.where(occ, 0, np.maximum(out, 1))   # 0 = occupied cell itself; a free cell can always take one step
    return out.astype(np.uint8)

# octants by (sign dx, sign dy, |dx| >= |dy|); image rows go DOWN when y goes up
planes = {}
for sx_ in (1, -1):
    for sy_ in (1, -1):
        for xmajor in (True, False):
            major = (0, sx_) if xmajor else (-sy_, 0)
            minor = (-sy_, 0) if xmajor else (0, sx_)
            planes[(sx_, sy_, xmajor)] = cone_plane(occ, major, minor)
print("planes built", flush=True)
rng = np.random.default_rng(0)
true = np.array([520.0, 280.0, 0.4])
NP, B, ZMAX = 64, 360, 500
SP = float(os.environ.get('SP', '0.3')); ST = float(os.environ.get('ST', '0.002'))
px = true[0] + SP * rng.standard_normal(NP); py = true[1] + SP * rng.standard_normal(NP); pth = true[2] + ST * rng.standard_normal(NP)

def cell(x, y):
    return int(rows - y - 1) + PAD, int(x) + PAD

def trace(sx, sy, th, cone):
    dx, dy = np.cos(th), np.sin(th)
    sc = 0.9999 / max(abs(dx), abs(dy))
    P = planes[(1 if dx >= 0 else -1, 1 if dy >= 0 else -1, abs(dx) >= abs(dy))] if cone else Ep
    k = 0
    i, j = cell(sx, sy)
    e = int(P[i, j]); reads = 1
    while e and k < ZMAX:
        k += 1 + int((e - 1) * sc)
        i, j = cell(sx + k * dx, sy + k * dy)
        e = int(P[i, j]); reads += 1
    return reads, (k if e == 0 else -1)

for cone in (False, True):
    tot = np.zeros((NP, B)); ks = np.zeros((NP, B))
    for p in range(NP):
        sx = px[p] + 30 * np.cos(pth[p] + np.pi / 2); sy = py[p] + 30 * np.sin(pth[p] + np.pi / 2)
        for b in range(B):
            tot[p, b], ks[p, b] = trace(sx, sy, pth[p] + b * 2 * np.pi / B - np.pi, cone)
    if not cone: k0 = ks
    print(f"cone={cone}: mean lookups/ray {tot.mean():.2f}; wavefront (max over the 64 particles of a beam) {tot.max(axis=0).mean():.2f}; "
          f"same hit sample as Chebyshev: {np.mean(ks == k0):.4f}")

# ---- the O(n) recurrence the library uses: C(c) = min(H(c), H(c + minor), 1 + C(c + major + minor)) ----
def cone_plane_dp(occ, major, minor):
    R, Cc = occ.shape
    BIG = 255
    H = np.zeros(occ.shape, np.int32)
    if major[0] == 0:   # along columns
        js = range(Cc - 1, -1, -1) if major[1] > 0 else range(Cc)
        prev = np.zeros(R, np.int32)   # outside = occupied
        for j in js:
            prev = np.where(occ[:, j], 0, np.minimum(prev + 1, BIG))
            H[:, j] = prev
    else:
        is_ = range(R - 1, -1, -1) if major[0] > 0 else range(R)
        prev = np.zeros(Cc, np.int32)
        for i in is_:
            prev = np.where(occ[i], 0, np.minimum(prev + 1, BIG))
            H[i] = prev
    def shifted(A, di, dj):   # A at c + (di, dj), 0 outside
        out = np.zeros_like(A)
        src_i = slice(max(di, 0), R + min(di, 0)); dst_i = slice(max(-di, 0), R + min(-di, 0))
        src_j = slice(max(dj, 0), Cc + min(dj, 0)); dst_j = slice(max(-dj, 0), Cc + min(-dj, 0))
        out[dst_i, dst_j] = A[src_i, src_j]
        return out
    Hm = shifted(H, minor[0], minor[1])
    base = np.minimum(H, Hm)
    C = np.zeros(occ.shape, np.int32)
    di, dj = major[0] + minor[0], major[1] + minor[1]
    rows_order = range(R - 1, -1, -1) if di > 0 else range(R)
    for i in rows_order:
        ii = i + di
        if 0 <= ii < R:
            nxt = np.zeros(Cc, np.int32)
            if dj > 0: nxt[:-1] = C[ii, 1:]
            else: nxt[1:] = C[ii, :-1]
        else:
            nxt = np.zeros(Cc, np.int32)
        C[i] = np.minimum(base[i], np.minimum(nxt + 1, BIG))
    return np.where(occ, 0, np.maximum(C, 1)).astype(np.uint8)

bad = 0
for key in planes:
    sx_, sy_, xmajor = key
    major = (0, sx_) if xmajor else (-sy_, 0)
    minor = (-sy_, 0) if xmajor else (0, sx_)
    dp = cone_plane_dp(occ, major, minor)
    # compare away from the wrap-around of np.roll in the brute-force version
    inner = (slice(260, -260), slice(260, -260))
    diff = (dp[inner] != planes[key][inner]).sum()
    bad += diff
    print(key, "cells that differ from the brute-force definition:", int(diff))
print("DP == brute force" if bad == 0 else "MISMATCH")
