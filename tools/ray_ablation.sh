#!/bin/bash
# What a sample of the lit march costs: one libtbrm.so per TBRM_RAY_EXP value (parts of the sample compiled out: WRONG frames)
# into tools/tmp/exp/, then (on a GPU box) the raymarch-only bench per variant.  tools/ray_ablation.sh build 1 2 4 ... | run 1 2 4 ...
set -e
cd "$(dirname "$0")/.."
CS=tbraymarcherplugin_amd/csrc
OUT=tools/tmp/exp
mode=$1; shift
if [ "$mode" = build ]; then
  python -c "from tbraymarcherplugin_amd import build as tb; tb.build(verbose=False)"
  mkdir -p $OUT
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-unused-variable -Wno-unused-but-set-variable"
  for u in tbrm_api tbrm_light_passes tbrm_host_math; do hipcc $FLAGS -c -x hip $CS/$u.cpp -o $OUT/$u.o & done
  hipcc $FLAGS -c -x hip $CS/tbrm_light_kernels.hip -o $OUT/lk.o &
  hipcc $FLAGS -c -x hip $CS/tbrm_light_sweep.hip -o $OUT/sweep.o &
  hipcc $FLAGS -DTBRM_CHAIN_LFMT=0 -c -x hip $CS/tbrm_light_chain.hip -o $OUT/chain_u8.o &
  hipcc $FLAGS -DTBRM_CHAIN_LFMT=2 -c -x hip $CS/tbrm_light_chain.hip -o $OUT/chain_f32.o &
  for e in "$@"; do hipcc $FLAGS -DTBRM_RAY_EXP=$e -c -x hip $CS/tbrm_kernels.hip -o $OUT/kern_$e.o & done
  wait
  for e in "$@"; do
    hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden $OUT/tbrm_api.o $OUT/tbrm_light_passes.o $OUT/tbrm_host_math.o $OUT/kern_$e.o $OUT/lk.o $OUT/chain_u8.o $OUT/chain_f32.o $OUT/sweep.o -o $OUT/libtbrm_ray$e.so
  done
  rm -f $OUT/*.o
else
  for e in 0 "$@"; do
    lib=$OUT/libtbrm_ray$e.so; [ $e = 0 ] && lib=tbraymarcherplugin_amd/lib/libtbrm.so
    TBRM_LIB_PATH=$PWD/$lib python bench.py --raymarch-only --no-cpu-baseline --timed-only 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TBRM_RAY_EXP $e: frame', d['gpu_ms']['raymarch'], 'ms (wall per step', d['ms_per_step'], ')')"
  done
fi
