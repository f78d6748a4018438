# Frame times of large meshes under builds with another limit on the tree's top level (variants/top*.so).
cp ray_tracing_amd/librt_mi355x.so /tmp/librt_orig.so
for v in top32 top64 top128; do
  cp variants/$v.so ray_tracing_amd/librt_mi355x.so
  for spp in 8 32; do echo "== $v, $spp spp"; BIG_SPP=$spp timeout -k 10 500 python tools/big_mesh.py "$@" 2>/dev/null | grep "at most 8"; done
done
cp /tmp/librt_orig.so ray_tracing_amd/librt_mi355x.so
