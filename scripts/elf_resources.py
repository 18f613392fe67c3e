"""Kernel descriptors of the shipped library, read from the gfx950 code objects themselves (the `.hip_fatbin` bundles -> ELF notes):
registers, spills and scratch per kernel as the hardware sees them -- NOT rocprofv3's `VGPR_Count` column, which is an allocation
granule count.  No GPU needed.
usage: elf_resources.py [lib.so] [substring ...]          -> one JSON line per matching kernel (demangled name)
       elf_resources.py --summary [lib.so]                 -> kernel count, kernels with scratch, library size, worst offenders
As a module: `kernels(path)` -> list of dicts; `find(path, *substrings)` -> the kernels whose demangled name holds every substring."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
DEFAULT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "solverforge_amd", "libsolverforge_amd.so")
FIELDS = {".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr", ".private_segment_fixed_size": "scratch", ".sgpr_spill_count": "sgpr_spills",
          ".vgpr_spill_count": "vgpr_spills", ".group_segment_fixed_size": "lds_static", ".max_flat_workgroup_size": "max_wg"}


def _code_objects(path):
    data = open(path, "rb").read()
    pos = 0
    while True:
        b = data.find(MAGIC, pos)
        if b < 0:
            return
        n = struct.unpack_from("<Q", data, b + 24)[0]
        p = b + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple and size:
                yield data[b + off:b + off + size]
        pos = b + 24


def _demangle(names):
    if not names:
        return {}
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def kernels(path=DEFAULT):
    res = []
    for co in _code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True).stdout
        cur = None
        for line in txt.split("\n"):
            if line.startswith("  - ."):  # a new entry of amdhsa.kernels (two-space indent; argument entries sit deeper)
                cur = {}
                res.append(cur)
                line = "    " + line[4:]
            if cur is None:
                continue
            m = re.match(r"^    (\.[a-z_]+):\s+(.*)$", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == ".name":
                cur["mangled"] = v.strip("'\"")
            elif k in FIELDS:
                try:
                    cur[FIELDS[k]] = int(v)
                except ValueError:
                    pass
    res = [r for r in res if "mangled" in r and "vgpr" in r]
    dm = _demangle([r["mangled"] for r in res])
    for r in res:
        r["kernel"] = dm.get(r["mangled"], r["mangled"])
        del r["mangled"]
    return res


def find(path, *subs):
    return [k for k in kernels(path) if all(s in k["kernel"] for s in subs)]


def summary(path=DEFAULT):
    ks = kernels(path)
    sc = [k for k in ks if k.get("scratch", 0) > 0]
    worst = sorted(sc, key=lambda k: -k["scratch"])[:8]
    return {"library": os.path.basename(path), "library_bytes": os.path.getsize(path), "kernels": len(ks), "kernels_with_scratch": len(sc),
            "kernels_scratch_over_700B": sum(1 for k in ks if k.get("scratch", 0) > 700),
            "largest_scratch": [{"kernel": k["kernel"][:140], "scratch": k["scratch"], "vgpr": k["vgpr"], "sgpr_spills": k.get("sgpr_spills", 0),
                                 "vgpr_spills": k.get("vgpr_spills", 0)} for k in worst]}


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--summary":
        print(json.dumps(summary(a[1] if len(a) > 1 else DEFAULT), indent=1))
        sys.exit(0)
    path = a[0] if a and a[0].endswith(".so") else DEFAULT
    subs = [x for x in a if not x.endswith(".so")]
    for k in find(path, *subs):
        print(json.dumps(k))
