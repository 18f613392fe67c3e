#!/bin/bash
# A variant library for A/B runs on the GPU box: ONE unit recompiled with extra flags, linked with the stock objects of csrc/_obj.
# (SF_CSRC=<dir>: compile the unit from another copy of csrc/, e.g. the previous commit's, for the baseline of an A/B)
# usage: build_variant.sh <name> <unit: list_wave_2 | mixed_2_2_1_0 | ...> "<extra flags>"      -> build/libsf_<name>.so   (SF_AMD_LIB=... selects it)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=${SF_CSRC:-$R/solverforge_amd/csrc}; O=$R/solverforge_amd/csrc/_obj; name=$1; unit=$2; extra=$3
mkdir -p $R/build/var_$name
case $unit in
  list_wave_*) src=sf_tu_list_wave.hip; defs="-DSF_TU_L=${unit#list_wave_}";;
  mixed_*) IFS=_ read -r _ l v ru pr <<< "$unit"; src=sf_tu_mixed.hip; defs="-DSF_TU_L=$l -DSF_TU_VTB=$v -DSF_TU_RUIN=$ru -DSF_TU_PREC=$pr";;
  scalar_*) IFS=_ read -r _ l v ir <<< "$unit"; src=sf_tu_scalar.hip; defs="-DSF_TU_L=$l -DSF_TU_VTB=$v -DSF_TU_IR=$ir";;
  api) src=sf_api.hip; defs="";;
  *) echo "unknown unit $unit"; exit 1;;
esac
(cd $C && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I/opt/rocm/include $defs $extra -c $src -o $R/build/var_$name/$unit.o)
objs=$(ls $O/*.o | grep -v "/$unit.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/build/var_$name/$unit.o -o $R/build/libsf_$name.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo built $R/build/libsf_$name.so
