# A/B of the tile-order policies inside one GPU call: bash tools/ab.sh [bench args]
for rep in 1 2; do
for v in "RT_NO_TILE_ORDER=1" "RT_X=1" "RT_TILE_LPT=1"; do
  echo "== $v"; env $v timeout -k 10 100 python bench.py --no-cpu-baseline --steps 30 "$@" | grep -o '"kernel_ms": [0-9.]*'
done; done
