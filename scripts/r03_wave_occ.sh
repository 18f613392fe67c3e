# headline kernel (CVRP-1000, 2-leaf nearby union): waves per SIMD 4 (wide slice) vs 5 / 6 (COMPACT slice, VGPR budget 96 / 80 with spills)
B="python bench.py --steps 10 --warmup 3 --solve-seconds 0 --no-cpu-baseline --no-pmc"
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'G moves/s', round(d['value']/1e9,3), 'ms/launch', round(d['roofline']['avg_launch_ms'],2), d['config']['replicas_per_gpu'])"; }
$B --replicas 4096 | pr "wpe4-4096"
D=$PWD/solverforge_amd/csrc/_diag
SF_AMD_LIB=$D/libsf_wpe5.so $B --replicas 5120 | pr "wpe5-5120"
SF_AMD_LIB=$D/libsf_wpe6.so $B --replicas 6144 | pr "wpe6-6144"
SF_AMD_LIB=$D/libsf_wpe6.so $B --replicas 5120 | pr "wpe6-5120"
