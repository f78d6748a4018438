# The slicing parameters re-swept with primary steps in place: steady ms per step of the two mesh scenes.
run() { echo "== $1 | $2"; env $1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-counters --config $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' ms/step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'cold', d['cold_first_frame_ms'])"; }
for c in "c3 --steps 5 --warmup 2" "c3 --mesh bunny.off --steps 5 --warmup 2"; do
  for e in "RT_X=0" "RT_SLICE_MAX=32 RT_SLICE_MIN_SPP=8" "RT_SLICE_MIN_SPP=8" "RT_SLICE_SHARE=0.125" "RT_SLICE_SHARE=0.5" "RT_SLICE_SHARE=0.18 RT_SLICE_EARLY=0.35" "RT_SLICE_EARLY=0.25" "RT_X=0"; do run "$e" "$c"; done
done
