# Bench lines of every configuration (one GPU): bash tools/bench_all.sh <outdir>
out=${1:-gpurun_out/bench_all}; mkdir -p "$out"
timeout -k 10 200 python bench.py --config c1 --steps 50 > "$out/c1.json" &&
timeout -k 10 200 python bench.py --config c2 > "$out/c2.json" &&
timeout -k 10 300 python bench.py --config c3 --steps 5 --warmup 2 > "$out/c3.json" &&
timeout -k 10 300 python bench.py --config c4 --steps 3 --warmup 1 > "$out/c4.json" &&
timeout -k 10 300 python bench.py --config c5 --steps 10 --warmup 2 > "$out/c5.json" &&
timeout -k 10 300 python bench.py --config c3 --mesh bunny.off --steps 5 --warmup 2 > "$out/c3_bunny.json" &&
timeout -k 10 400 python bench.py --config c5e --steps 3 --warmup 1 > "$out/c5e.json" &&
for f in c1 c2 c3 c3_bunny c4 c5 c5e; do cat "$out/$f.json"; done
