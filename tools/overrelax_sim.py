"""CPU estimate of lookups per ray for the clearance search with over-relaxed jumps (design study).
Positions are x0 + k*dx here (counting only; the kernel's exact-jump machinery is not needed for that)."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
d = os.path.join(ROOT, "tests", "host_harness")
subprocess.run(["make", "-C", d, "-s"], check=True)
hh = ctypes.CDLL(os.path.join(d, "libray_harness.so"))
g = (bench.load_free_grid("floor_plan") * 255).astype(np.uint8)
rows, cols = g.shape
E = np.empty_like(g)
hh.hh_build_clearance(g.ctypes.data_as(ctypes.c_void_p), rows, cols, E.ctypes.data_as(ctypes.c_void_p))
Ep = np.zeros((rows + 600, cols + 600), np.uint8)   # wide zero frame for the study
Ep[300:300 + rows, 300:300 + cols] = E
rng = np.random.default_rng(0)
true = np.array([float(sys.argv[1]) if len(sys.argv) > 1 else 520.0, float(sys.argv[2]) if len(sys.argv) > 2 else 280.0, 0.4])
NP, B, ZMAX = 60, 360, 500
SP=float(os.environ.get('SP','3')); ST=float(os.environ.get('ST','0.03'))
px = true[0] + SP * rng.uniform(-1,1,NP); py = true[1] + SP * rng.uniform(-1,1,NP); pth = true[2] + ST * rng.uniform(-1,1,NP)

def cell(x, y):
    return int(rows - y - 1) + 300, int(x) + 300

def trace(sx, sy, th, omega):
    dx, dy = np.cos(th), np.sin(th)
    sc = 0.9999 / max(abs(dx), abs(dy))
    k = 0
    i, j = cell(sx, sy)
    e = int(Ep[i, j]); reads = 1
    if e == 0:
        return reads
    relax = omega
    while k < ZMAX:
        m0 = 1 + int((e - 1) * sc)
        x = int(relax * (m0 - 1)) if relax > 0 else 0
        kk = k + m0 + x
        i, j = cell(sx + kk * dx, sy + kk * dy)
        e1 = int(Ep[i, j]); reads += 1
        if x > 0:
            ok = e1 > 0 and x <= int((e1 - 1) * sc)
            if not ok:
                kk = k + m0
                i, j = cell(sx + kk * dx, sy + kk * dy)
                e1 = int(Ep[i, j]); reads += 1
                relax = 0.0          # stay literal for the rest of the ray (or: for a few steps)
        k = kk
        e = e1
        if e == 0:
            break
    return reads

for omega in (0.0,):
    tot = []
    for p in range(NP):
        sx = px[p] + 30 * np.cos(pth[p] + np.pi / 2); sy = py[p] + 30 * np.sin(pth[p] + np.pi / 2)
        tot.append([trace(sx, sy, pth[p] + b * 2 * np.pi / B, omega) for b in range(B)])
    tot = np.array(tot)
    print(f"omega_extra={omega}: mean lookups/ray {tot.mean():.2f}  max-over-particles mean {tot.max(axis=0).mean():.2f}  p90 {np.percentile(tot, 90):.0f}")
