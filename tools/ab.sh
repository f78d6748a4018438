# A/B of environment switches inside one GPU call: bash tools/ab.sh [configs...]
for c in "${@:-c2}"; do
for v in "RT_GROUP_WAVES=4" "RT_GROUP_WAVES=1"; do
  echo "== $c $v"; env $v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --config $c | grep -o '"kernel_ms": [0-9.]*'
done; done
