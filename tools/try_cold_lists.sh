for rep in 1 2; do for v in 32 8; do for c in "c3 --steps 3 --warmup 1" "c3 --mesh bunny.off --steps 3 --warmup 1"; do
echo "== min_spp $v $c"; RT_LISTS_MIN_SPP=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-counters --config $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' cold', d['cold_first_frame_ms'], 'ms/step', d['ms_per_step'])"
done; done; done
