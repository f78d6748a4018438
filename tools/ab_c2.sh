cp ray_tracing_amd/librt_mi355x.so /tmp/librt_orig.so
for rep in 1 2; do for v in call5.so cur.so nocount.so; do
  cp variants/$v ray_tracing_amd/librt_mi355x.so
  echo "== $v"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-counters --config c2 --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' kernel_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'], 'unsettled', d['ms_per_step_unsettled'], 'pshare', d['primary_steps_share'], 'cold', d['cold_first_frame_ms'])"
done; done
cp /tmp/librt_orig.so ray_tracing_amd/librt_mi355x.so
