# A/B of prebuilt library variants (variants/*.so) on every configuration, inside one GPU call: bash tools/try_variants5.sh variant...
cp ray_tracing_amd/librt_mi355x.so /tmp/librt_orig.so
for v in "$@"; do
  cp variants/$v ray_tracing_amd/librt_mi355x.so
  for c in "c2 --steps 20" "c3 --steps 5 --warmup 2" "c3 --mesh bunny.off --steps 5 --warmup 2" "c5 --steps 10 --warmup 2" "c5e --steps 3 --warmup 1"; do
    echo "== $v $c"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-counters --config $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' kernel_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'], 'lane_use', d['lane_use'], 'pshare', d['primary_steps_share'], 'cold', d['cold_first_frame_ms'])"
  done
done
cp /tmp/librt_orig.so ray_tracing_amd/librt_mi355x.so
