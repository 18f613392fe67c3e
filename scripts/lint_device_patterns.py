"""Grep-level lint for the code shapes DESIGN 8.15 lists as miscompiled by ROCm 7.2 hipcc on gfx950:
 1. a struct returned BY VALUE from a `__noinline__` device function (8.15 item 2: came back with stale fields through `?:`);
 2. a call of such a function inside a conditional expression `c ? f<A>(..) : f<B>(..)`.
A function annotated `lint: small-pod-return` in the comment above it (<= 16 bytes, returned in registers) passes check 1 and is still
subject to check 2.  Prints one line per finding; exit code 1 when there is any (tests/test_abi.py runs it on csrc/)."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCALARS = {"void", "bool", "int", "uint32_t", "int32_t", "uint64_t", "int64_t", "uint16_t", "uint8_t", "float", "double", "unsigned", "size_t"}


def findings(paths):
    out = []
    noinline_struct = set()
    sig = re.compile(r"__noinline__[^;{()]*?\b([A-Za-z_][A-Za-z0-9_:<>]*)\s+([A-Za-z_][A-Za-z0-9_]*)\s*\(")
    for p in paths:
        text = open(p).read()
        for m in sig.finditer(text):
            ret, name = m.group(1), m.group(2)
            base = ret.split("<")[0].split("::")[-1]
            if base in SCALARS or ret.endswith("*"):
                continue
            line = text.count("\n", 0, m.start()) + 1
            noinline_struct.add(name)
            if "lint: small-pod-return" in text[max(0, m.start() - 600):m.start()]:  # annotated: <= 16 bytes, register-returned
                continue
            out.append(f"{os.path.relpath(p, ROOT)}:{line}: __noinline__ device function `{name}` returns `{ret}` by value")
    if noinline_struct:
        tern = re.compile(r"\?\s*(" + "|".join(map(re.escape, noinline_struct)) + r")\s*[<(]")
        for p in paths:
            for i, l in enumerate(open(p).read().split("\n"), 1):
                if tern.search(l):
                    out.append(f"{os.path.relpath(p, ROOT)}:{i}: struct-returning __noinline__ call inside a conditional expression")
    return out


if __name__ == "__main__":
    src = sorted(glob.glob(os.path.join(ROOT, "solverforge_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "solverforge_amd", "csrc", "*.hip"))
                 + glob.glob(os.path.join(ROOT, "solverforge_amd", "csrc", "*.inc")))
    f = findings(src)
    for l in f:
        print(l)
    sys.exit(1 if f else 0)
