"""Generates tests/golden/maps.npz from the reference's map images (run in the dev container only).

The three PNGs under /root/reference/assets are the reference's ground-truth maps.  Its callers
turn them into a binary grid with `cv::imread(GRAYSCALE)` + `cv::threshold(128 -> {0,1})`
(apps/grid_slam.cpp:79-86, benchmark/mcl.cpp:17-19): gray > 128 -> 1 (free), else 0 (wall).
OpenCV is not available here, so PIL does the grayscale conversion (ITU-R 601 luma like OpenCV's;
for these black/white drawings the two agree except possibly on anti-aliased edge pixels).
Only derived data is stored: one bit per cell, row-major, np.packbits order.
"""
import sys

import numpy as np
from PIL import Image

ASSETS = "/root/reference/assets"


def main(out="tests/golden/maps.npz"):
    arrays = {}
    for name in ("map", "floor_plan", "maze"):
        img = Image.open(f"{ASSETS}/{name}.png").convert("L")
        gray = np.asarray(img, dtype=np.uint8)
        free = (gray > 128).astype(np.uint8)
        arrays[f"{name}_bits"] = np.packbits(free.reshape(-1))
        arrays[f"{name}_shape"] = np.array(free.shape, dtype=np.int32)
        print(name, free.shape, "free fraction %.4f" % free.mean())
    np.savez_compressed(out, **arrays)


if __name__ == "__main__":
    main(*sys.argv[1:])
