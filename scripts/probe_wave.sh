#!/bin/bash
# Diagnostics of the wave kernel on the GPU box: phase shares (SF_PHASE_PROFILE build), perf-only probe variants
# (wrong results on purpose), and the vector-memory pipeline counters.
R=$GRAFT_REPO_ROOT; cd $R
echo "== phase shares"; SF_AMD_LIB=$R/build/libsf_phase.so python scripts/phase_probe.py 4096 2>&1 | tail -4
B="python bench.py --no-pmc --solve-seconds 0 --steps 20 --warmup 5 --no-cpu-baseline"
for lib in solverforge_amd/libsolverforge_amd.so build/libsf_probe_NO_DEMAND.so build/libsf_probe_HALF_LEGS.so; do
  SF_AMD_LIB=$R/$lib $B | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value']/1e9, d['roofline']['avg_launch_ms'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name" | grep -E "TA_|TCP_|TD_" | awk '{print $3}' | tr '\n' ' ' | head -c 6000; echo
for grp in "TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  timeout 120 rocprofv3 --pmc $grp -f csv -d /tmp/pp -o b -- python $R/bench.py --steps 6 --warmup 5 --pmc-child > /tmp/o.log 2> /tmp/e.log; echo "pass [$grp] rc=$?"
  python - <<'PY'
import csv,glob,collections
for f in glob.glob('/tmp/pp/**/*counter_collection.csv',recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_list_search_wave' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print({k:(len(v),sum(v[5:])/max(len(v[5:]),1)) for k,v in agg.items()})
PY
  rm -rf /tmp/pp
done
