#!/bin/bash
# Guard against the gfx950 code-generation pitfalls of DESIGN 8.15: the differential fuzz (GPU vs CPU oracle) on the shipped -O3
# library AND on a -O1 build of the same sources, same seeds.  A divergence of either build from the oracle is a finding; a case
# that fails under one optimisation level only points at the compiler (or at latent undefined behaviour) rather than at the algorithm.
#   here (no GPU):   bash scripts/opt_level_guard.sh build          -> build/libsf_O1.so
#   on the GPU box:  bash scripts/opt_level_guard.sh run <seconds per leg> [first seed]   -> gpurun_out/opt_guard_*.json
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
case "${1:-run}" in
build)
  make -C "$R/solverforge_amd/csrc" -j"$(nproc)" -s EXTRA="-O1" OUT="$R/build/libsf_O1.so" OBJ=/tmp/sf_O1_obj ;;
run)
  secs=${2:-120}; seed=${3:-1}; O=$R/gpurun_out; mkdir -p "$O"; cd "$R"
  for lib in O3 O1; do
    [ $lib = O1 ] && export SF_AMD_LIB=$R/build/libsf_O1.so || unset SF_AMD_LIB
    python scripts/fuzz_parity.py "$secs" "$seed" > "$O/opt_guard_parity_$lib.json" 2> "$O/opt_guard_parity_$lib.err"
    python scripts/fuzz_construction.py "$secs" "$seed" > "$O/opt_guard_construction_$lib.json" 2> "$O/opt_guard_construction_$lib.err"
  done
  python - <<PY
import json
for leg in ("parity", "construction"):
    for lib in ("O3", "O1"):
        try:
            d = json.loads(open("$O/opt_guard_%s_%s.json" % (leg, lib)).read().strip().splitlines()[-1])
            print(leg, lib, "cases", d.get("cases"), "failures", d.get("failures"))
        except Exception as e:
            print(leg, lib, "no result:", e)
PY
  ;;
esac
