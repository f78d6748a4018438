# A/B of prebuilt library variants (variants/*.so, built on the CPU box) inside one GPU call:
#   bash tools/try_variants.sh <config> variant...
# Each variant is copied over the package's library, benched, and the original put back.
set -e
cfg=$1; shift
cp ray_tracing_amd/librt_mi355x.so /tmp/librt_orig.so
for v in "$@"; do
  cp variants/$v ray_tracing_amd/librt_mi355x.so
  echo "== $v"; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 12 --config $cfg | grep -o '"kernel_ms": [0-9.]*'
done
cp /tmp/librt_orig.so ray_tracing_amd/librt_mi355x.so
