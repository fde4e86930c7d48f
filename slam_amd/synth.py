"""Seeded procedural occupancy grids for the large configurations (BASELINE configs[2]
8192^2 and configs[4] 16384^2): border walls plus axis-aligned rooms and corridors with door gaps
until about 10 % of the cells are occupied (the reference's assets measure 3.7-9.9 %).
uint8, 255 = free, 0 = wall."""
import numpy as np

SEEDS = {8192: 0x51A30003, 16384: 0x51A30005}


def make_grid(size, seed=None, target_occupied=0.10, wall=3):
    seed = SEEDS.get(size, 0x51A30000 + size) if seed is None else seed
    rng = np.random.default_rng(seed)
    g = np.full((size, size), 255, np.uint8)
    g[:wall, :] = g[-wall:, :] = 0
    g[:, :wall] = g[:, -wall:] = 0
    occupied = int((g == 0).sum())
    goal = int(target_occupied * size * size)
    lo, hi = max(24, size // 64), max(48, size // 10)
    while occupied < goal:
        h, w = (int(v) for v in rng.integers(lo, hi, 2))
        i, j = int(rng.integers(0, size - h)), int(rng.integers(0, size - w))
        before = int((g[i:i + h, j:j + w] == 0).sum())
        t = wall
        if rng.uniform() < 0.35:  # solid block (pillar / unexplored mass)
            hh, ww = min(h, lo * 2), min(w, lo * 2)
            g[i:i + hh, j:j + ww] = 0
            after = int((g[i:i + h, j:j + w] == 0).sum())
        else:  # room outline with one door per side
            g[i:i + t, j:j + w] = 0
            g[i + h - t:i + h, j:j + w] = 0
            g[i:i + h, j:j + t] = 0
            g[i:i + h, j + w - t:j + w] = 0
            door = max(8, min(h, w) // 6)
            di, dj = int(rng.integers(t, max(t + 1, h - door - t))), int(rng.integers(t, max(t + 1, w - door - t)))
            g[i:i + t, j + dj:j + dj + door] = 255
            g[i + h - t:i + h, j + dj:j + dj + door] = 255
            g[i + di:i + di + door, j:j + t] = 255
            g[i + di:i + di + door, j + w - t:j + w] = 255
            after = int((g[i:i + h, j:j + w] == 0).sum())
        occupied += after - before
    g[:wall, :] = g[-wall:, :] = 0
    g[:, :wall] = g[:, -wall:] = 0
    return g


def open_pose(g, margin=48, theta=0.7):
    """A pose [x, y, theta] (world frame: x = column, y = rows - 1 - row) in the upper left quarter of the grid whose
    surroundings are free for `margin` cells in every direction, so that a sensor mounted up to that far from the pose
    (bench.py's offset is 30) still stands in free space and the scan taken there has real ranges.  (Round 2's first
    choice, the middle free cell of a strip, put the 16384^2 sensor into a wall: every beam measured half a cell.)"""
    size = g.shape[0]
    free = (g >= 128)
    step = 2 * margin + 1
    i0 = size // 4
    for i in range(i0, size - step, step):
        for j in range(i0, size - step, step):
            if free[i:i + step, j:j + step].all():
                ci, cj = i + margin, j + margin
                return [float(cj) + 0.5, float(size - 1 - ci) + 0.5, theta]
    raise ValueError("no open place found")
