cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "FETCH_SIZE|RDREQ|WRITE_SIZE|WRREQ" | head -40
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "FETCH_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  s=$(date +%s)
  timeout 90 rocprofv3 --pmc $grp -f csv -d /tmp/pp -o b -- python $R/bench.py --steps 4 --warmup 1 --pmc-child > /tmp/o.log 2> /tmp/e.log
  rc=$?
  e=$(date +%s)
  echo "pass [$grp] rc=$rc secs=$((e - s))"
  tail -3 /tmp/e.log
  python - <<'PY'
import csv,glob,collections
for f in glob.glob('/tmp/pp/**/*counter_collection.csv',recursive=True):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_list_search_wave' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print({k:(len(v),sum(v)/len(v)) for k,v in agg.items()})
PY
  rm -rf /tmp/pp
done
