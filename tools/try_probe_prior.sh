# Duration of the probe launch of a cold frame under RT_PROBE_PRIOR = 0, 3, 20, 100 (rocprofv3 kernel trace).
#   bash tools/try_probe_prior.sh "<bench arguments>" <tag>
cd /tmp && export TMPDIR=/tmp
for w in 0 3 20 100; do
  export RT_PROBE_PRIOR=$w
  timeout -s KILL 200 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/probe_prior_$2_$w -- python3 $GRAFT_REPO_ROOT/bench.py $1 --steps 2 --warmup 1 --no-cpu-baseline --no-live-counters --streams 1 --settle-ms 0 > $GRAFT_REPO_ROOT/gpurun_out/probe_prior_$2_$w.log 2>&1 || exit 1
done
