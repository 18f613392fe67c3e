#!/bin/bash
# round 6, tenth GPU call: precedence stages with by-reference parameter blocks + a private copy inside each stage; scalar engine without the SCarve stack copy.
# A/B of the precedence rates over three libraries (wbase = by value, g1 = by reference, current), C2 / C4 rates old vs new, parity of the touched families
R=$GRAFT_REPO_ROOT; cd $R; O=$R/gpurun_out/r06_r10; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_precedence.py tests/test_gpu_precedence_leaf.py tests/test_gpu_scalar.py tests/test_gpu_mixed.py tests/test_gpu_timed_states.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee $O/tests.txt
for lib in build/libsf_wbase.so build/libsf_g1.so solverforge_amd/libsolverforge_amd.so; do
  export SF_AMD_LIB=$R/$lib
  echo "$lib four-leaf 50x20: $(timeout 300 python scripts/precedence_bench.py 50 20 2048 5 2 list_change,list_swap,sublist_change,list_reverse 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e6,2),'M', d['replica0_matches_oracle'], d['kernel_ms_per_launch'])")" | tee -a $O/prec_ab.txt
  echo "$lib nine-leaf 50x20: $(timeout 300 python scripts/prec_policy_launches.py 50 20 2048 10 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['moves_per_s']/1e6,2),'M', d['ms_per_launch'])")" | tee -a $O/prec_ab.txt
  echo "$lib c4 makespan: $(timeout 300 python scripts/c4_makespan_rate.py 256 5 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e3,1),'K', d['kernel_ms_per_launch'])")" | tee -a $O/prec_ab.txt
done
for lib in build/libsf_wbase.so solverforge_amd/libsolverforge_amd.so; do
  export SF_AMD_LIB=$R/$lib
  for pol in la sa; do echo "$lib graph $pol: $(timeout 300 python scripts/graph_bench.py 3072 100 10 $pol 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e9,3),'G', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/scalar_ab.txt; done
  echo "$lib jobshop: $(timeout 300 python scripts/jobshop_bench.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['gpu_moves_per_s']/1e9,3),'G', d['kernel_ms_per_launch'], d.get('replica0_matches_indexed_cpu'))")" | tee -a $O/scalar_ab.txt
done
unset SF_AMD_LIB
timeout 300 python scripts/m2_probe.py 12288 default6 250 8 2>&1 | tail -1 | tee -a $O/m2_late.jsonl
timeout 300 python scripts/m2_probe.py 6144 default 250 8 2>&1 | tail -1 | tee -a $O/m2_late.jsonl
