export TMPDIR=/tmp
for lib in "" tools/tmp/exp/libtbrm_occ91.so; do
  rm -rf /tmp/prof_pm
  ( [ -n "$lib" ] && export TBRM_LIB_PATH=$PWD/$lib; VARIANTS="light_cache_mb=0" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d /tmp/prof_pm -o p -- python tools/sweep_time.py > /dev/null 2>&1 )
  f=$(find /tmp/prof_pm -name "*counter_collection.csv" | head -1)
  echo "== ${lib:-product}"
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "k_light_occlusion<1, 0, 0>" in n:
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVES": cnt[n] += 1
for n in acc:
    c = cnt[n] or 1
    print("  ", n[:50], {k: round(v / c / 1e6, 2) for k, v in acc[n].items()}, "launches", c)
PY
done
