# CVRP-5000 / 500 (BASELINE config 5) on the wave engine: COMPACT slice (5 replicas per CU) vs the wide slice (3 per CU)
B="python bench.py --customers 5000 --vehicles 500 --steps 6 --warmup 2 --ls-steps 100 --solve-seconds 0 --no-cpu-baseline --no-pmc"
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'G moves/s', round(d['value']/1e9,3), 'ms/launch', round(d['roofline']['avg_launch_ms'],2), d['config']['replicas_per_gpu'])"; }
SF_AMD_NO_COMPACT=1 $B --replicas 768 | pr "wide-768"
SF_AMD_NO_COMPACT=1 $B --replicas 1280 | pr "wide-1280"
$B --replicas 1280 | pr "compact-1280"
$B --replicas 2560 | pr "compact-2560"
