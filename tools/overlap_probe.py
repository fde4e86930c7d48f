"""Do a torch side stream and the stream the library runs on execute concurrently on this box?
Run under `rocprofv3 --kernel-trace --output-format csv`; tools/overlap_probe.py --analyze <csv> reads the trace."""
import os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    rays = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_rays_pl" in r["Kernel_Name"]]
    probes = [(int(r["Start_Timestamp"]), r.get("Queue_Id"), r["Kernel_Name"][:50]) for r in rows if "sin" in r["Kernel_Name"].lower()]
    inside = sum(any(a <= s < b for a, b in rays) for s, _, _ in probes)
    print(f"{len(probes)} side-stream probes, {inside} started while k_rays_pl was running; queues of probes:",
          sorted({q for _, q, _ in probes}), "queues of k_rays_pl:", sorted({r.get('Queue_Id') for r in rows if 'k_rays_pl' in r['Kernel_Name']}))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
import slam_amd as sa
free = bench.load_free_grid("floor_plan"); grid = (free * 255).astype(np.uint8)
use_null = os.environ.get("PROBE_MAIN", "null") == "null"
main = torch.cuda.current_stream() if use_null else torch.cuda.Stream()
side = torch.cuda.Stream()
with torch.cuda.stream(main):
    f = sa.Filter(100_000, *grid.shape, seed=1, stream=main.cuda_stream, async_launch=True, sort=sa.capi.SORT_NONE)
    f.set_map(grid); f.seed_gaussian(bench.TRUE_POSE, [25.0, 25.0, 0.2])
    scans = bench.synth_scans(sa, free, 0, 360, bench.true_trajectory(12))
    x = torch.ones(1 << 16, device="cuda"); y = torch.empty_like(x)
    torch.cuda.synchronize()
    for k in range(12):
        f.predict(bench.ODOM, bench.ALPHAS)
        f.update_weights(*scans[k], bench.STDDEV, bench.MAX_DIST, bench.OFFSET)
        with torch.cuda.stream(side):
            torch.sin(x, out=y)           # independent of the main stream: overlaps k_rays_pl if the queues differ
        f.resample()
    torch.cuda.synchronize()
print("probe run done")
