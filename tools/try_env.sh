# A/B of prebuilt library variants (variants/*.so) x environment settings inside one GPU call:
#   bash tools/try_env.sh "<bench arguments>" "<VAR=value ...;VAR=value ...>" variant...     (settings separated by ';')
set -e
args=$1; IFS=';' read -ra envs <<< "$2"; shift; shift
cp ray_tracing_amd/librt_mi355x.so /tmp/librt_orig.so
for v in "$@"; do
  cp variants/$v ray_tracing_amd/librt_mi355x.so
  for e in "${envs[@]}"; do
    echo "== $v $e"; env $e timeout -k 10 200 python bench.py --no-cpu-baseline --steps 12 $args | grep -o '"kernel_ms": [0-9.]*'
  done
done
cp /tmp/librt_orig.so ray_tracing_amd/librt_mi355x.so
