#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_final; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/gpu_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $O/gpu_tests.txt
timeout 400 python scripts/fuzz_parity.py 240 21000 > $O/fuzz_parity.json 2> $O/fuzz_parity.err; tail -c 400 $O/fuzz_parity.json; echo
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value", d["value"]/1e9, "traffic", r.get("traffic"), "frac", r.get("frac"), "failed", (r.get("kernel_resources") or {}).get("failed_passes"))
e=d["extra"]; print("m2", e["best_score_at_60s"]["gpu"], e["best_score_at_60s"]["gpu_moves_per_s_rank0"], "c5", e["side_configs"]["cvrp5000_nearby2"].get("moves_per_s_rank0"), "match", e.get("replica0_matches_cpu_oracle"))
PY
