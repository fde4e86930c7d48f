"""Builds libslam_hip.so (gfx950) in-tree with hipcc.  Run: python -m slam_amd.build"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libslam_hip.so")

# -ffp-contract=off: host and device must evaluate detmath.h / ray.h without fused contraction so
# that results are bit-identical to the strict-IEEE CPU oracle (explicit fma() calls remain).
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fPIC",
    "-shared",
    "-fvisibility=hidden",
    "-Wall",
    "-Wno-unused-function",
    "-Wno-unused-value",
    "-Wno-unused-result",
]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force=False, verbose=False, extra=()):
    srcs = [os.path.join(CSRC, "slam_hip.hip")]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [
        os.path.join(HERE, "..", "include", "slam_hip.h")
    ]
    if not force and os.path.exists(LIB):
        newest = max(os.path.getmtime(d) for d in deps)
        if os.path.getmtime(LIB) >= newest:
            return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc()] + FLAGS + list(extra) + ["-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra=[a for a in sys.argv[1:] if a.startswith("-R") or a.startswith("-save")])
    print(LIB)
