#!/bin/bash
# round 5, session 15: the whole GPU suite, the differential fuzz, the default bench line on the current library
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r05_s15; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tee $O/gpu_tests.txt
timeout 400 python scripts/fuzz_parity.py 240 5000 > $O/fuzz_parity.json 2> $O/fuzz_parity.err; tail -c 600 $O/fuzz_parity.json
SF_FUZZ_MODEL=cvrp timeout 300 python scripts/fuzz_parity.py 150 9000 > $O/fuzz_parity_cvrp.json 2> $O/fuzz_parity_cvrp.err; tail -c 400 $O/fuzz_parity_cvrp.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("value", d["value"]/1e9, "traffic", r.get("traffic"), "hbm_frac", r.get("hbm_frac"), "per_cand", r.get("per_candidate"), "full", r.get("per_candidate_full_launch"))
print("failed", (r.get("kernel_resources") or {}).get("failed_passes"))
print("side", json.dumps(d["extra"].get("side_configs"))[:900])
print("m2", d["extra"]["best_score_at_60s"]["gpu"], d["extra"]["best_score_at_60s"]["gpu_moves_per_s_rank0"], "match", d["extra"].get("replica0_matches_cpu_oracle"))
PY
