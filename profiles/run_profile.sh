#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's dominant kernel on the GPU box.
# usage: profiles/run_profile.sh <tag>      (writes gpurun_out/prof_<tag>/...)
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
EXTRA=${BENCH_EXTRA:-}
# --streams 1: one frame in flight, so that a launch's duration and counters are its own; the counter passes skip the
# frames that bring the device to its sustained clock (counters do not depend on it), the timed trace run keeps them
BENCH="python3 $ROOT/bench.py --steps ${PMC_STEPS:-5} --warmup 1 --no-cpu-baseline --settle-ms 0 --streams 1 $EXTRA"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ROOT/bench.py --steps ${TRACE_STEPS:-10} --warmup 2 --no-cpu-baseline --streams 1 $EXTRA > "$OUT/trace.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc1" -- $BENCH > "$OUT/pmc1.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d "$OUT/pmc2" -- $BENCH > "$OUT/pmc2.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc3" -- $BENCH > "$OUT/pmc3.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE GRBM_COUNT --output-format csv -d "$OUT/pmc4" -- $BENCH > "$OUT/pmc4.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS --output-format csv -d "$OUT/pmc5" -- $BENCH > "$OUT/pmc5.log" 2>&1
ls "$OUT"
