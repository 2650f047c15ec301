#!/bin/bash
# Builds one libtbrm.so per ablation of the sweep kernel's slice loop (TBRM_SWEEP_EXP, tbrm_light_sweep.hip) into
# tools/tmp/exp/, to be timed on the GPU with sweep_debug = 1 (tiles do not wait): what a slice costs without this or that.
set -e
cd "$(dirname "$0")/.."
python -c "from tbraymarcherplugin_amd import build as tb; tb.build(verbose=False)"
CS=tbraymarcherplugin_amd/csrc
OUT=tools/tmp/exp
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable"
for u in tbrm_api tbrm_light_passes tbrm_host_math; do hipcc $FLAGS -c -x hip $CS/$u.cpp -o $OUT/$u.o & done
hipcc $FLAGS -c -x hip $CS/tbrm_kernels.hip -o $OUT/tbrm_kernels.o &
hipcc $FLAGS -c -x hip $CS/tbrm_light_kernels.hip -o $OUT/tbrm_light_kernels.o &
hipcc $FLAGS -DTBRM_CHAIN_LFMT=0 -c -x hip $CS/tbrm_light_chain.hip -o $OUT/chain_u8.o &
hipcc $FLAGS -DTBRM_CHAIN_LFMT=2 -c -x hip $CS/tbrm_light_chain.hip -o $OUT/chain_f32.o &
for e in "$@"; do hipcc $FLAGS -DTBRM_SWEEP_EXP=$e -c -x hip $CS/tbrm_light_sweep.hip -o $OUT/sweep_$e.o & done
wait
for e in "$@"; do
  hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden $OUT/tbrm_api.o $OUT/tbrm_light_passes.o $OUT/tbrm_host_math.o $OUT/tbrm_kernels.o $OUT/tbrm_light_kernels.o $OUT/chain_u8.o $OUT/chain_f32.o $OUT/sweep_$e.o -o $OUT/libtbrm_exp$e.so
done
rm -f $OUT/*.o
ls -la $OUT
