# Kernel time of the headline configurations inside one GPU call: bash tools/bench_quick.sh
set -e -o pipefail
for c in c2 c3 c5; do
  echo "== $c"; timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --config $c | grep -o '"kernel_ms": [0-9.]*'
done
