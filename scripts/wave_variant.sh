#!/bin/bash
# Kernel-variant experiments on the wave engine: compiles ONLY the wave units (sf_tu_list_wave.hip, L = 2 and 4) with extra flags and links
# them with the other objects of the in-tree build into build/libsf_<name>.so (about 30 s instead of a full 4 min build).  The in-tree
# library must be current (make) -- the other objects come from csrc/_obj.   usage: wave_variant.sh <name> "<extra hipcc flags>"
set -e
name=$1; extra=${2:-}
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/solverforge_amd/csrc; T=/tmp/sf_wv_$name; mkdir -p $T $R/build
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I/opt/rocm/include $extra"
(cd $C && hipcc $FL -DSF_TU_L=2 -c sf_tu_list_wave.hip -o $T/list_wave_2.o -Rpass-analysis=kernel-resource-usage 2> $T/res2.txt || { tail -20 $T/res2.txt; exit 1; }
 hipcc $FL -DSF_TU_L=4 -c sf_tu_list_wave.hip -o $T/list_wave_4.o)
objs=$(ls $C/_obj/*.o | grep -v list_wave_)
hipcc --offload-arch=gfx950 -shared -fPIC $objs $T/list_wave_2.o $T/list_wave_4.o -o $R/build/libsf_$name.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
grep -A12 "Function Name: _ZN2sf18k_list_search_waveILi2ELb0ELi2ELb1ELi5E" $T/res2.txt | grep -E "SGPRs:|VGPRs:|Spill|ScratchSize|Occupancy" | tr '\n' ' '; echo
