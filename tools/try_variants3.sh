# A/B of prebuilt library variants on the mesh scene with both meshes and on the enclosed sphere scene:
#   bash tools/try_variants3.sh variant...
set -e
cp ray_tracing_amd/librt_mi355x.so /tmp/librt_orig.so
for v in "$@"; do
  cp variants/$v ray_tracing_amd/librt_mi355x.so
  for mesh in "" bunny.off; do
    echo "== $v ${mesh:-Bunny-LowPoly.off}"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-counters --steps 8 --warmup 2 --config c3 ${mesh:+--mesh $mesh} | grep -o '"kernel_ms": [0-9.]*'
  done
  echo "== $v c5e"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-counters --steps 3 --warmup 1 --config c5e | grep -o '"kernel_ms": [0-9.]*'
done
cp /tmp/librt_orig.so ray_tracing_amd/librt_mi355x.so
