# A/B of prebuilt library variants (variants/*.so) x workgroup sizes inside one GPU call:
#   bash tools/try_groups.sh "<bench arguments>" "<group sizes>" variant...
set -e
args=$1; groups=$2; shift; shift
cp ray_tracing_amd/librt_mi355x.so /tmp/librt_orig.so
for v in "$@"; do
  cp variants/$v ray_tracing_amd/librt_mi355x.so
  for g in $groups; do
    echo "== $v RT_GROUP_WAVES=$g"; RT_GROUP_WAVES=$g timeout -k 10 200 python bench.py --no-cpu-baseline --steps 12 $args | grep -o '"kernel_ms": [0-9.]*'
  done
done
cp /tmp/librt_orig.so ray_tracing_amd/librt_mi355x.so
