# Cold first frame of the mesh scenes with the probe planned from the candidate lists (RT_PROBE_PRIOR = weight of a leaf; 0: off)
for rep in 1 2; do for v in 0 3 1 8; do for c in "c3 --steps 3 --warmup 1" "c3 --mesh bunny.off --steps 3 --warmup 1"; do
echo "== prior $v $c"; RT_PROBE_PRIOR=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-counters --config $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' cold', d['cold_first_frame_ms'], 'ms/step', d['ms_per_step'])"
done; done; done
