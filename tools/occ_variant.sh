#!/bin/bash
# Builds libtbrm.so variants whose occlusion kernel is compiled for more waves per SIMD (fewer registers), into tools/tmp/exp/:
#   tools/occ_variant.sh 7 8   ->  libtbrm_occ7.so libtbrm_occ8.so   (selected at run time by TBRM_LIB_PATH)
set -e
cd "$(dirname "$0")/.."
python -c "from tbraymarcherplugin_amd import build as tb; tb.build(verbose=False)"
CS=tbraymarcherplugin_amd/csrc
OUT=tools/tmp/exp
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-function"
for u in tbrm_api tbrm_light_passes tbrm_host_math; do hipcc $FLAGS -c -x hip $CS/$u.cpp -o $OUT/$u.o & done
hipcc $FLAGS -c -x hip $CS/tbrm_kernels.hip -o $OUT/tbrm_kernels.o &
hipcc $FLAGS -c -x hip $CS/tbrm_light_sweep.hip -o $OUT/sweep.o &
hipcc $FLAGS -DTBRM_CHAIN_LFMT=0 -c -x hip $CS/tbrm_light_chain.hip -o $OUT/chain_u8.o &
hipcc $FLAGS -DTBRM_CHAIN_LFMT=2 -c -x hip $CS/tbrm_light_chain.hip -o $OUT/chain_f32.o &
for e in "$@"; do hipcc $FLAGS ${OCC_DEFS:--DTBRM_OCC_WAVES_PER_EU=$e} -DTBRM_OCC_TAG=$e -c -x hip $CS/tbrm_light_kernels.hip -o $OUT/lk_$e.o & done
wait
for e in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden $OUT/tbrm_api.o $OUT/tbrm_light_passes.o $OUT/tbrm_host_math.o $OUT/tbrm_kernels.o $OUT/lk_$e.o $OUT/chain_u8.o $OUT/chain_f32.o $OUT/sweep.o -o $OUT/libtbrm_occ$e.so
done
rm -f $OUT/*.o
ls -la $OUT
