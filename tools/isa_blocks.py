"""Per-basic-block instruction counts of one kernel in a gfx950 .s file (hipcc -save-temps).
Usage: python tools/isa_blocks.py <file.s> <mangled-kernel-name-substring> [--dump]
Prints, per block label: VALU / SALU / VMEM / LDS / branch counts and the f64 VALU share, so that the
per-phase cost of a kernel is read off the code instead of guessed from ablation timings."""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^[A-Za-z_0-9]+:", l) and key in l)
    blocks = []
    cur = {"name": "entry", "v": 0, "v64": 0, "s": 0, "m": 0, "l": 0, "br": 0, "w": 0, "first": start}
    for i in range(start + 1, len(lines)):
        l = lines[i]
        m = re.match(r"^(\.LBB[0-9_]+):", l)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "v": 0, "v64": 0, "s": 0, "m": 0, "l": 0, "br": 0, "w": 0, "first": i}
            continue
        t = l.strip()
        if t.startswith("; %bb."):
            continue
        op = t.split(" ")[0].split("\t")[0]
        if op.startswith("v_"):
            cur["v"] += 1
            if "f64" in op or "_b64" in op or "u64" in op or "i64" in op:
                cur["v64"] += 1
        elif op.startswith("s_waitcnt"):
            cur["w"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"):
            cur["br"] += 1
            cur["s"] += 1
        elif op.startswith("s_"):
            cur["s"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["m"] += 1
        elif op.startswith("ds_"):
            cur["l"] += 1
        if op == "s_endpgm":
            # keep going: more blocks may follow an early exit
            pass
        if t.startswith(".Lfunc_end"):
            break
    blocks.append(cur)
    tot = {k: sum(b[k] for b in blocks) for k in ("v", "v64", "s", "m", "l", "br")}
    print("%-14s %6s %6s %6s %5s %5s %4s" % ("block", "VALU", "v64", "SALU", "VMEM", "LDS", "br"))
    for b in blocks:
        print("%-14s %6d %6d %6d %5d %5d %4d   line %d" % (b["name"], b["v"], b["v64"], b["s"], b["m"], b["l"], b["br"], b["first"] - start))
    print("%-14s %6d %6d %6d %5d %5d %4d" % ("total", tot["v"], tot["v64"], tot["s"], tot["m"], tot["l"], tot["br"]))
    if dump:
        end = blocks[-1]["first"] + 400
        for i in range(start, min(end, len(lines))):
            print("%5d %s" % (i - start, lines[i]))
            if lines[i].strip().startswith(".Lfunc_end"):
                break


if __name__ == "__main__":
    main()
