#!/bin/bash
# Kernel-variant experiments beyond the wave engine: recompiles the named translation units of csrc/Makefile with extra flags and links them with
# the other objects of the in-tree build into build/libsf_<name>.so.   usage: tu_variant.sh <name> "<extra flags>" <unit> [<unit> ...]
#   units: api | list_wave_<L> | scalar_<L>_<VTB> | mixed_<L>_<VTB>_<RUIN>_<PREC> | list_block        (names of csrc/_obj/*.o)
set -e
name=$1; extra=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/solverforge_amd/csrc; T=/tmp/sf_tuv_$name; mkdir -p $T $R/build
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I/opt/rocm/include $extra"
skip=""
for u in "$@"; do
  case $u in
    api) src=sf_api.hip; defs="" ;;
    list_block) src=sf_tu_list_block.hip; defs="" ;;
    list_wave_*) src=sf_tu_list_wave.hip; defs="-DSF_TU_L=${u#list_wave_}" ;;
    scalar_*) IFS=_ read -r _ l v <<< "$u"; src=sf_tu_scalar.hip; defs="-DSF_TU_L=$l -DSF_TU_VTB=$v" ;;
    mixed_*) IFS=_ read -r _ l v r p <<< "$u"; src=sf_tu_mixed.hip; defs="-DSF_TU_L=$l -DSF_TU_VTB=$v -DSF_TU_RUIN=$r -DSF_TU_PREC=$p" ;;
    *) echo "unknown unit $u"; exit 1 ;;
  esac
  (cd $C && hipcc $FL $defs -c $src -o $T/$u.o -Rpass-analysis=kernel-resource-usage 2> $T/$u.res) || { grep -E "error" $T/$u.res | head; exit 1; }
  skip="$skip -e /$u.o"
done
OBJ_DIR=${OBJ_DIR:-$C/_obj}
objs=$(ls $OBJ_DIR/*.o | grep -v $skip)
hipcc --offload-arch=gfx950 -shared -fPIC $objs $T/*.o -o $R/build/libsf_$name.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
for u in "$@"; do grep -E "Function Name|VGPRs:|ScratchSize|Occupancy|VGPRs Spill" $T/$u.res | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - - - | grep -E "search_wave" | cut -c1-60,140-260; done
