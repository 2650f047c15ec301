#!/bin/bash
# What the occlusion of pass i+1 gains from running beside the sweep of pass i: cached fused Changes at bench size, per
# variant the mean GPU time of a Change and the kernel timeline of the last ones (rocprofv3 kernel trace).
#   tools/overlap_probe.sh "name:lib:ENV=val ENV2=val" ...
export TMPDIR=/tmp
[ $# -eq 0 ] && set -- "base::"
for v in "$@"; do
  name=${v%%:*}; rest=${v#*:}; lib=${rest%%:*}; envs=${rest#*:}
  echo "== $name"
  ( [ -n "$lib" ] && export TBRM_LIB_PATH=$PWD/$lib; [ -n "$envs" ] && export $envs
    for L in 0 1 2 3; do LIGHT=$L STEPS=8 python tools/change_sequence.py 2>/dev/null | grep cached | awk '{s+=$5; n++} END {printf "light '$L': mean cached Change %.3f ms over %d\n", s/n, n}'; done
    rm -rf /tmp/prof_ov; rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_ov -o p -- env LIGHT=1 STEPS=4 python tools/change_sequence.py > /dev/null 2>&1
    find /tmp/prof_ov -name "*kernel_trace.csv" -exec cp {} /tmp/ov_kt.csv \;
    python tools/trace_timeline.py /tmp/ov_kt.csv 16 | cut -c1-120 | grep -v "copyBuffer\|k_occ_\|fillBuffer" )
done
