# A/B of prebuilt library variants (variants/*.so), each measured twice, on the given configurations:
#   bash tools/ab_variants.sh "c2 --steps 20;c5 --steps 10 --warmup 2" variant...
IFS=';' read -ra CONFIGS <<< "$1"; shift
cp ray_tracing_amd/librt_mi355x.so /tmp/librt_orig.so
for rep in 1 2; do for c in "${CONFIGS[@]}"; do for v in "$@"; do
  cp variants/$v ray_tracing_amd/librt_mi355x.so
  echo "== $v $c"; timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-counters --config $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' kernel_ms', d['roofline']['kernel_ms'], 'ms/step', d['ms_per_step'], 'unsettled', d['ms_per_step_unsettled'], 'pshare', d['primary_steps_share'], 'cold', d['cold_first_frame_ms'])"
done; done; done
cp /tmp/librt_orig.so ray_tracing_amd/librt_mi355x.so
