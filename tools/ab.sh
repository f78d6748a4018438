# A/B of the launch-plan policies inside one GPU call: bash tools/ab.sh [configs...]
for c in "${@:-c2}"; do
for v in "RT_NO_TILE_ORDER=1" "RT_NO_SLICES=1" "RT_X=1"; do
  echo "== $c $v"; env $v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --config $c | grep -o '"kernel_ms": [0-9.]*'
done; done
