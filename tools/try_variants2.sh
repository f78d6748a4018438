# A/B of prebuilt library variants over several configurations inside one GPU call:
#   bash tools/try_variants2.sh "<config> ..." variant...
set -e
cfgs=$1; shift
cp ray_tracing_amd/librt_mi355x.so /tmp/librt_orig.so
for v in "$@"; do
  cp variants/$v ray_tracing_amd/librt_mi355x.so
  for cfg in $cfgs; do
    echo "== $v $cfg"; timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-counters --steps 8 --warmup 2 --config $cfg | grep -o '"kernel_ms": [0-9.]*'
  done
done
cp /tmp/librt_orig.so ray_tracing_amd/librt_mi355x.so
