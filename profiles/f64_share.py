"""Static share of 64-bit VALU instructions in the hot kernels (gfx950 disassembly of `hipcc -save-temps`), written to
profiles/valu_f64_share.json with the fingerprint of the sources.  An f64 VALU instruction occupies the SIMD for 4.6-4.9
cycles on gfx950 where a 32-bit one takes 3.2-3.5 (profiles/ubench_r03.txt), so bench.py weighs the wavefront instruction counts of
the SQ counters with this share (static: every instruction of the kernel body counted once, not by how often it runs).
Usage: python profiles/f64_share.py   (needs hipcc; ~1 minute)"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"k_rays_pl": "k_rays_plILi5ELb0ELb0EE", "k_sum_lw": "k_sum_lw", "k_weights_L<0>": "k_weights_LILi0ELi6EE",
           "k_weights_S2<tiled>": "k_weights_S2ILb1ELb0ELb0EE", "k_weights_S2F": "k_weights_S2FE"}


def count(lines, key):
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_0-9]+:", l) and key in l)
    v = v64 = s = 0
    for l in lines[start + 1:]:
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            break
        op = t.split(" ")[0].split("\t")[0]
        if op.startswith("v_"):
            v += 1
            if "f64" in op or "_b64" in op or "u64" in op or "i64" in op:
                v64 += 1
        elif op.startswith("s_") and not op.startswith("s_waitcnt"):
            s += 1
    return {"valu": v, "valu_64bit": v64, "salu": s, "f64_share": v64 / max(v, 1)}


def main():
    with tempfile.TemporaryDirectory() as d:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-fvisibility=hidden",
               "-Wno-unused-result", "-save-temps", "-o", os.path.join(d, "lib.so"), os.path.join(ROOT, "slam_amd", "csrc", "slam_hip.hip")]
        subprocess.run(cmd, check=True, cwd=d, capture_output=True)
        asm = next(f for f in os.listdir(d) if f.endswith(".s") and "gfx950" in f)
        lines = open(os.path.join(d, asm)).read().split("\n")
    out = {"kernels": {k: count(lines, key) for k, key in KERNELS.items()}}
    h = hashlib.sha256()
    src = os.path.join(ROOT, "slam_amd", "csrc")
    for name in sorted(os.listdir(src)):
        h.update(open(os.path.join(src, name), "rb").read())
    out["source_fingerprint"] = h.hexdigest()[:16]
    out["note"] = "static counts over the kernel body (gfx950 ISA of hipcc -O3 -save-temps); 64-bit = f64 and 64-bit integer VALU opcodes"
    json.dump(out, open(os.path.join(ROOT, "profiles", "valu_f64_share.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
