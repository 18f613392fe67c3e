# u16 matrix copy for the trial gathers (ListModel::mat16): A/B at CVRP-1000 (bench default), CVRP-5000 and the six-leaf generic union
B="python bench.py --solve-seconds 0 --no-cpu-baseline --no-pmc"
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'G moves/s', round(d['value']/1e9,3), 'ms/launch', round(d['roofline']['avg_launch_ms'],2))"; }
$B --steps 10 --warmup 3 | pr "C3 mat16"
SF_AMD_NO_MAT16=1 $B --steps 10 --warmup 3 --replicas 4096 | pr "C3 u32 (4 waves, wide)"
$B --customers 5000 --vehicles 500 --replicas 1280 --steps 6 --warmup 2 --ls-steps 100 | pr "C5 mat16"
U=nearby_change,nearby_swap,sublist_change,sublist_swap,list_reverse,kopt
echo "6leaf mat16: $(python scripts/union_probe.py 3072 100 3 $U 2>&1 | tail -1 | cut -c100-330)"
echo "6leaf u32:   $(SF_AMD_NO_MAT16=1 python scripts/union_probe.py 3072 100 3 $U 2>&1 | tail -1 | cut -c100-330)"
