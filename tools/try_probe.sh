# Samples of the one-shot frame's probe launch against the cold frame's time: bash tools/try_probe.sh
set -e -o pipefail
for p in 8 4 2 1; do
  for mesh in "" bunny.off; do
    echo "== RT_PROBE_SPP=$p mesh=${mesh:-Bunny-LowPoly.off}"
    RT_PROBE_SPP=$p timeout -k 10 150 python bench.py --no-cpu-baseline --no-live-counters --settle-ms 0 --steps 3 --warmup 1 --config c3 ${mesh:+--mesh $mesh} | grep -o '"cold_first_frame_ms": [0-9.]*'
  done
  echo "== RT_PROBE_SPP=$p c2"
  RT_PROBE_SPP=$p timeout -k 10 150 python bench.py --no-cpu-baseline --no-live-counters --settle-ms 0 --steps 3 --warmup 1 --config c2 | grep -o '"cold_first_frame_ms": [0-9.]*'
done
