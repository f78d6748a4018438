# Frames in flight against time per frame, headline configuration: bash tools/try_streams.sh
set -e -o pipefail
for s in 1 2 3 4; do
  for k in 20; do
    echo "== streams $s steps $k"
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-live-counters --steps $k --warmup 5 --settle-ms ${SETTLE:-250} --streams $s --config c2 | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*'
  done
done
