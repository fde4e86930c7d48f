"""CPU design study for the wide-cloud / long-ray regime: lookups per ray of the clearance search with the
8 octant cones the kernels use (0 <= b <= a + 2) against 32 direction sectors (each octant split by slope into
[k/4, (k+1)/4], cone floor(a k / 4) - 1 <= b <= ceil(a (k + 1) / 4) + 2), on a crop of the synthetic 16384^2 grid.
One octant (dx >= dy >= 0) stands for all.  Brute-force planes (prefix sums along the minor axis), counting only.
python tools/sector_sim.py [crop] [rays]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from slam_amd import synth

CROP = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
NRAYS = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
g = synth.make_grid(16384)[4096:4096 + CROP, 4096:4096 + CROP]
occ = g < 128
occ[0, :] = occ[-1, :] = occ[:, 0] = occ[:, -1] = True
R = C = CROP
# world frame of the octant: major = +x = +column, minor = +y = -row (image rows go down)
S = np.zeros((R + 1, C), np.int32)          # S[i] = occupied cells in rows [i, R) of each column (suffix sum downwards)
S[:R] = np.cumsum(occ[::-1], axis=0)[::-1]

def plane(lo, hi):
    """C(c) = min a >= 0 with an occupied cell (row - b, col + a), lo(a) <= b <= hi(a); outside the crop = occupied."""
    out = np.full((R, C), 255, np.int32)
    found = np.zeros((R, C), bool)
    rows = np.arange(R)[:, None]
    for a in range(255):
        l, h = lo(a), hi(a)
        # rows row - h .. row - l at column col + a
        top = np.clip(rows - h, 0, R)          # first row of the window
        bot = np.clip(rows - l + 1, 0, R)      # one past its last row
        cnt = np.zeros((R, C), np.int32)
        cols_ok = C - a
        if cols_ok <= 0:
            break
        Sa = S[:, a:]                           # column col + a
        win = np.take_along_axis(Sa, np.broadcast_to(top, (R, cols_ok)), axis=0) - np.take_along_axis(Sa, np.broadcast_to(bot, (R, cols_ok)), axis=0)
        # windows that poke out of the crop at the top count as occupied
        poke = (rows - h < 0)
        hit_here = (win > 0) | np.broadcast_to(poke, (R, cols_ok))
        cnt[:, :cols_ok] = hit_here
        cnt[:, cols_ok:] = 1                    # beyond the right edge: occupied
        hit = (~found) & (cnt > 0)
        out[hit] = a
        found |= hit
    return np.where(occ, 0, np.maximum(out, 1)).astype(np.uint8)

t0 = time.time()
wide = plane(lambda a: 0, lambda a: a + 2)
narrow = [plane(lambda a, k=k: (a * k) // 4 - 1, lambda a, k=k: -((-a * (k + 1)) // 4) + 2) for k in range(4)]
print(f"planes built in {time.time() - t0:.0f} s", flush=True)

rng = np.random.default_rng(5)
free = np.argwhere(~occ[64:-64, 64:-64]) + 64
res = {"wide": [], "narrow": []}
for _ in range(NRAYS):
    i0, j0 = free[rng.integers(len(free))]
    k = int(rng.integers(4))
    slope = (k + rng.random()) / 4.0
    th = np.arctan(slope)
    dx, dy = np.cos(th), np.sin(th)
    for name, pl in (("wide", wide), ("narrow", narrow[k])):
        x, y = j0 + rng.random() * 0 + 0.5, (R - 1 - i0) + 0.5
        n, looks, length = 0, 0, 0
        while n < 3000:
            i, j = int(R - 1 - np.floor(y)), int(np.floor(x))
            if not (0 <= i < R and 0 <= j < C):
                break
            e = int(pl[i, j])
            looks += 1
            if e == 0:
                break
            m = 1 + int((e - 1) * 0.9999 / dx)
            x += m * dx; y += m * dy; n += m
        res[name].append((looks, n))
for name, v in res.items():
    v = np.array(v)
    print(f"{name:7s}: {v[:, 0].mean():.1f} lookups per ray, mean length {v[:, 1].mean():.0f} samples, {v[:, 1].mean() / v[:, 0].mean():.1f} samples per lookup")
