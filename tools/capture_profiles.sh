#!/bin/bash
# Captures the rocprofv3 evidence behind bench.py's roofline numbers on the GPU box (run from the repository root):
#   1. kernel trace of the full bench configuration            -> gpurun_out/<tag>_kernel_trace_summary.txt
#   1b. the other four BASELINE configurations (base B=32, medium B=15, large-v3-turbo B=32, tiny.en B=1): a bench line
#      (roofline + path_frac, no CPU baseline) and a kernel trace each -> gpurun_out/<tag>_<config>_{bench_line.json,kernel_trace_summary.txt}
#   2. PMC passes, one counter block per run (never mixed with a trace): FETCH_SIZE, WRITE_SIZE, SQ busy / wait counters
#      (a TCC_HIT / TCC_MISS pass over this program produced no output for three minutes on this pool and is left out)
#                                                               -> gpurun_out/<tag>_pmc_summary.txt
# Usage: tools/capture_profiles.sh <tag> [all|main|configs|pmc]   (copy the summaries you want kept into profiles/)
set -o pipefail
TAG=${1:-r03}
PART=${2:-all}
want() { [ "$PART" = all ] || [ "$PART" = "$1" ]; }
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
BENCH="python3 bench.py --inflight 1 --steps 1 --solo-steps 1 --warmup 1 --no-cpu-baseline"
if want main; then
# one batch in flight: kernel durations as the roofline figures use them (1 warm-up + 1 + 1 timed passes)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 bench.py --inflight 1 --steps 1 --solo-steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_trace.log 2>&1 || exit 1
python3 tools/prof_summary.py $OUT/${TAG}_trace 3 > $OUT/${TAG}_bench_small_b128_kernel_trace_summary.txt || exit 1
rm -rf $OUT/${TAG}_trace
# the default command (three batches in flight): durations include the time kernels of different batches share the GPU
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace3 -o t -- python3 bench.py --steps 6 --solo-steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_trace3.log 2>&1 || exit 1
python3 tools/prof_summary.py $OUT/${TAG}_trace3 10 > $OUT/${TAG}_bench_small_b128_inflight3_kernel_trace_summary.txt || exit 1
rm -rf $OUT/${TAG}_trace3
fi
if want configs; then
# the other BASELINE configurations: bench line first (un-profiled timings), then a kernel trace of one batch in flight
for cfg in "base 32" "medium 15" "large-v3-turbo 32" "tiny.en 1"; do
  set -- $cfg; model=$1; batch=$2; name=${model}_b${batch}
  timeout -k 10 300 python3 bench.py --model $model --batch $batch --steps 12 --solo-steps 3 --no-cpu-baseline > $OUT/${TAG}_${name}_bench_line.json 2> $OUT/${TAG}_${name}_bench.err || { echo "bench $name failed"; tail -3 $OUT/${TAG}_${name}_bench.err; exit 1; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace_$name -o t -- python3 bench.py --model $model --batch $batch --inflight 1 --steps 1 --solo-steps 1 --warmup 1 --no-cpu-baseline > $OUT/${TAG}_trace_$name.log 2>&1 || { echo "trace $name failed"; exit 1; }
  python3 tools/prof_summary.py $OUT/${TAG}_trace_$name 3 > $OUT/${TAG}_${name}_kernel_trace_summary.txt || exit 1
  rm -rf $OUT/${TAG}_trace_$name
  echo "config $name done"
done
fi
want pmc || exit 0
{
  echo "# rocprofv3 --pmc passes (one counter block per run) over: $BENCH   (3 passes of the path in each run, one batch in flight)"
  echo "# whisper-small shapes, 128 windows, bf16, 170-token prompt, 128 generated tokens.  FETCH_SIZE / WRITE_SIZE are in KiB;"
  echo "# on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes for wide coalesced reads, so read bytes = 2 * FETCH_SIZE * 1024."
} > $OUT/${TAG}_pmc_summary.txt
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 280 rocprofv3 --pmc $pass --output-format csv -d $OUT/${TAG}_pmc_$name -o p -- $BENCH > $OUT/${TAG}_pmc_$name.log 2>&1 || { echo "pmc pass $name failed"; exit 1; }
  { echo; echo "## --pmc $pass"; python3 tools/pmc_summary.py $OUT/${TAG}_pmc_$name; } >> $OUT/${TAG}_pmc_summary.txt
  rm -rf $OUT/${TAG}_pmc_$name
  echo "pass $name done"
done
