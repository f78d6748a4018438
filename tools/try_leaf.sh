# Leaf size of the triangle tree, per mesh: bash tools/try_leaf.sh   (kernel time of the mesh scene with both meshes)
set -e -o pipefail
for leaf in 0 2 4; do
  for mesh in "" bunny.off; do
    echo "== RT_LEAF=$leaf mesh=${mesh:-Bunny-LowPoly.off}"
    if [ "$leaf" = 0 ]; then unset RT_LEAF; else export RT_LEAF=$leaf; fi
    timeout -k 10 150 python bench.py --no-cpu-baseline --no-live-counters --steps 10 --config c3 ${mesh:+--mesh $mesh} | grep -o '"ms_per_step": [0-9.]*\|"kernel_ms": [0-9.]*\|"cold_first_frame_ms": [0-9.]*'
  done
done
