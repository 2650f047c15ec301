#!/bin/bash
# Round measurement protocol (run on the GPU box through gpurun): parity tests, the default bench line, the rocprofv3
# kernel-trace summary of the same command, the PMC traffic passes, the raymarch-only bench and the operator timings.
# Output: gpurun_out/$1/ ; copy what is to be judged into profiles/.
set -u
OUT=gpurun_out/${1:-round}
mkdir -p "$OUT"
export TMPDIR=/tmp
python bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o p -- python bench.py --no-cpu-baseline --timed-only > "$OUT/bench_n1_under_rocprof.json" 2> /dev/null
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_bench_n1.csv" \;
find /tmp/prof_stats -name "*kernel_trace.csv" -exec cp {} /tmp/kt.csv \;
python tools/trace_kernels.py /tmp/kt.csv > "$OUT/kernel_durations_bench_n1.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -o p -- python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o p -- python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 4 > /dev/null 2>&1
find /tmp/prof_fetch -name "*counter_collection.csv" -exec cp {} /tmp/pmc_fetch.csv \;
find /tmp/prof_write -name "*counter_collection.csv" -exec cp {} /tmp/pmc_write.csv \;
python tools/pmc_traffic.py /tmp/pmc_fetch.csv /tmp/pmc_write.csv "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.txt" 2>&1
# instruction-issue view (tools/issue_roofline.py): two more counter passes of the same command
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d /tmp/prof_issue1 -o p -- python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 4 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d /tmp/prof_issue2 -o p -- python bench.py --no-cpu-baseline --timed-only --steps 20 --warmup 4 > /dev/null 2>&1
find /tmp/prof_issue1 -name "*counter_collection.csv" -exec cp {} /tmp/pmc_issue1.csv \;
find /tmp/prof_issue2 -name "*counter_collection.csv" -exec cp {} /tmp/pmc_issue2.csv \;
python tools/issue_roofline.py "$OUT/issue.json" /tmp/pmc_issue1.csv /tmp/pmc_issue2.csv > "$OUT/issue.txt" 2>&1
python bench.py --raymarch-only --no-cpu-baseline > "$OUT/bench_n1_raymarch_only.json" 2> /dev/null
(echo "== tools/prof_light.py, light_cache_mb=0 (every call samples the volume)"; TBRM_LIGHT_CACHE_MB=0 python tools/prof_light.py 2>&1 | grep -v amdgpu.ids
 echo "== tools/prof_light.py, defaults (repeated calls propagate from kept factors)"; python tools/prof_light.py 2>&1 | grep -v amdgpu.ids) > "$OUT/operators.txt"
(echo "== tools/sweep_time.py: sweep / chain, cache on / off"; VARIANTS="light_sweep=0,light_cache_mb=0;light_cache_mb=0;light_cache_mb=-1" python tools/sweep_time.py 2>&1 | grep -v amdgpu.ids
 echo "== tools/change_sequence.py (one light turned 5 degrees per call)"; python tools/change_sequence.py 2>&1 | grep -v amdgpu.ids
 echo "== the same, light_cache_mb=0"; TBRM_LIGHT_CACHE_MB=0 python tools/change_sequence.py 2>&1 | grep -v amdgpu.ids | head -6
 echo "== tools/sweep_stamps.py (timeline of one sweep launch)"; python tools/sweep_stamps.py 2>&1 | grep -v amdgpu.ids | tee /tmp/stamps32.txt
 echo "== tools/stamps_summary.py of the above (32 x 32 tiles)"; python tools/stamps_summary.py < /tmp/stamps32.txt
 echo "== tools/host_enqueue_time.py"; python tools/host_enqueue_time.py 2>&1 | grep -v amdgpu.ids) >> "$OUT/operators.txt"
(echo "== tools/chain_ab.py: chained sweeps (sweep_chain = 4) against one launch per pass, 512^3"; python tools/chain_ab.py 2>&1 | grep -v amdgpu.ids
 for n in 256 128; do echo "== the same at $n^3"; N=$n python tools/chain_ab.py 2>&1 | grep -v amdgpu.ids; done
 echo "== tools/chain_stamps.py: timeline of chained launches, 512^3"; python tools/chain_stamps.py 2>&1 | grep "chain stamps\|^--"
 echo "== the same at 256^3"; N=256 python tools/chain_stamps.py 2>&1 | grep "chain stamps\|^--") > "$OUT/sweep_chain.txt"
for c in 1 2 4 5; do python bench.py --config $c --no-cpu-baseline --timed-only 2> /dev/null | tail -1 > "$OUT/bench_n1_config$c.json"; done
TBRM_BENCH_ONE_GPU_DRY_RUN=1 python bench.py --gpus 2 --steps 5 --warmup 2 2> /dev/null | tail -1 > "$OUT/bench_dry_run_2_ranks_on_one_gpu.json"
# the parity suite last (a slow box must not cost the measurements above their place in the call's time limit)
(timeout 1700 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error|FAILED|ERROR|Timeout" | tail -8) > "$OUT/tests.txt"
cat "$OUT/tests.txt" "$OUT/operators.txt"
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for f in ("bench_n1.json", "bench_n1_under_rocprof.json", "bench_n1_raymarch_only.json", "bench_n1_config1.json", "bench_n1_config2.json",
          "bench_n1_config4.json", "bench_n1_config5.json", "bench_dry_run_2_ranks_on_one_gpu.json"):
    try:
        d = json.load(open(f"{out}/{f}"))
        print(f, d["value"], d["ms_per_step"], d["gpu_ms"], d["roofline"]["achieved"], d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("value"),
              d.get("full_size_parity"), d.get("gathered_frame_equals_single_gpu_render"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
