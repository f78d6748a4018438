# Cold first frame against steady state for environment settings: bash tools/try_cold.sh "<bench arguments>" "<VAR=value;...>"
args=$1; IFS=';' read -ra envs <<< "$2"
for e in "${envs[@]}"; do
  echo "== $e"; env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-live-counters --steps 6 $args | grep -o '"kernel_ms": [0-9.]*\|"cold_first_frame_ms": [0-9.]*' | paste - -
done
