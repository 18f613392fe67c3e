"""Static instruction mix per basic block of one kernel in a hipcc -save-temps .s file (SALU / VALU / LDS / VMEM / lane ops /
scratch), with the loop nesting guessed from backward branches -- the tool behind the SALU / scratch attribution in DESIGN.md.
usage: isa_blocks.py file.s kernel-substring [top]"""
import re
import sys


def cat(op):
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_load", "s_buffer", "s_store")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    return "other"


def parse(path, needle):
    lines = open(path).read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l.split(":")[0] and ":" in l][0]
    blocks = []
    cur = {"name": "entry", "ins": [], "line": start}
    blocks.append(cur)
    for i in range(start + 1, len(lines)):
        s = lines[i].strip()
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            cur = {"name": m.group(1), "ins": [], "line": i}
            blocks.append(cur)
            continue
        if not s or s.startswith((";", ".")):
            continue
        cur["ins"].append(s)
    return blocks


if __name__ == "__main__":
    blocks = parse(sys.argv[1], sys.argv[2])
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    index = {b["name"]: k for k, b in enumerate(blocks)}
    depth = [0] * len(blocks)
    loops = []
    for k, b in enumerate(blocks):
        for ins in b["ins"]:
            op = ins.split()[0]
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = ins.split()[-1]
                if tgt in index and index[tgt] <= k:
                    loops.append((index[tgt], k))
    for lo, hi in loops:
        for k in range(lo, hi + 1):
            depth[k] += 1
    tot = {}
    for k, b in enumerate(blocks):
        c = {}
        for ins in b["ins"]:
            key = cat(ins.split()[0])
            c[key] = c.get(key, 0) + 1
            tot[key] = tot.get(key, 0) + 1
        b["c"] = c
        b["depth"] = depth[k]
    print(len(blocks), "blocks;", "loops:", len(loops), "totals:", tot)
    for b in sorted(blocks, key=lambda b: -len(b["ins"]))[:top]:
        print(b["name"], "line", b["line"], "depth", b["depth"], "n", len(b["ins"]), b["c"])
