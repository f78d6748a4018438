# Where leaves of two stop paying: the first N faces of bunny.off as the mesh of C3, leaves of two against four.
#   bash tools/try_leaf_border.sh
set -e -o pipefail
python - <<'PY'
import os
src = os.path.join("ray_tracing_amd", "data", "meshes", "bunny.off")
lines = [l for l in open(src).read().splitlines() if l.strip() and not l.startswith("#")]
head = lines[0]; nv, nf = (int(x) for x in lines[1].split()[:2])
verts, faces = lines[2:2 + nv], lines[2 + nv:2 + nv + nf]
for n in (600, 1000, 1500, 2200, 3200):
    with open(f"/tmp/bunny_{n}.off", "w") as f:
        f.write(head + "\n" + f"{nv} {n} 0\n" + "\n".join(verts) + "\n" + "\n".join(faces[:n]) + "\n")
PY
for n in 600 1000 1500 2200 3200; do
  for leaf in 2 4; do
    echo "== $n triangles RT_LEAF=$leaf"
    RT_LEAF=$leaf timeout -k 10 150 python bench.py --no-cpu-baseline --no-live-counters --steps 8 --warmup 2 --config c3 --mesh /tmp/bunny_$n.off | grep -o '"kernel_ms": [0-9.]*'
  done
done
