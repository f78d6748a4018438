# Waves per workgroup against time per frame: bash tools/try_group_waves.sh "<configs>"
set -e -o pipefail
for cfg in ${1:-c2}; do
  for g in 4 2 1; do
    echo "== $cfg RT_GROUP_WAVES=$g"
    RT_GROUP_WAVES=$g timeout -k 10 150 python bench.py --no-cpu-baseline --no-live-counters --steps 20 --warmup 5 --config $cfg | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*'
  done
done
