# Do the camera rays' lists pay at few samples per pixel?  ms per step with them (RT_LISTS_MIN_SPP=1) and without (default 32).
run() { echo "== $1 | $2"; env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-counters --config $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'pshare', d['primary_steps_share'])"; }
for rep in 1 2; do
for c in "c3 --spp 8 --steps 40" "c3 --spp 16 --steps 30" "c3 --mesh bunny.off --spp 8 --steps 40" "c3 --mesh bunny.off --spp 16 --steps 30" "c2 --spp 8 --steps 200" "c2 --spp 16 --steps 200" "c5 --spp 8 --steps 50" "c5 --spp 16 --steps 50" "c5e --spp 8 --steps 10"; do
  for e in "RT_X=0" "RT_LISTS_MIN_SPP=1"; do run "$e" "$c"; done
done; done
