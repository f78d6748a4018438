# A/B of the camera rays' candidate lists (primary steps) inside one GPU call: bash tools/ab_lists.sh
for c in "c2 --steps 20" "c4 --steps 3 --warmup 1" "c3 --steps 5 --warmup 2" "c3 --mesh bunny.off --steps 5 --warmup 2" "c1 --steps 50"; do
for v in "RT_NO_TILE_LISTS=1" "RT_NO_TILE_LISTS=0"; do
  echo "== $c $v"; env $v timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-counters --config $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' kernel_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'], 'lane_use', d['lane_use'], 'pshare', d['primary_steps_share'], 'frac', d['roofline']['frac'], 'cold', d['cold_first_frame_ms'])"
done; done
