# One-shot frames with and without the probe launch: bash tools/try_noprobe.sh "<configs>"
set -e -o pipefail
for cfg in ${1:-c1 c2 c4 c5}; do
  for np in 0 1; do
    echo "== $cfg RT_NO_PROBE=$np"
    if [ $np = 1 ]; then export RT_NO_PROBE=1; else unset RT_NO_PROBE; fi
    for rep in 1 2; do
      timeout -k 10 150 python bench.py --no-cpu-baseline --no-live-counters --settle-ms 0 --steps 2 --warmup 1 --config $cfg | grep -o '"cold_first_frame_ms": [0-9.]*'
    done
  done
done
