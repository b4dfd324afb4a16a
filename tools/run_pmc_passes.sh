#!/bin/bash
# The profiling runs of record (on the GPU box):  tools/run_pmc_passes.sh gpurun_out/<dir>
#   1. rocprofv3 --kernel-trace --stats over the default bench command  -> kernel stats CSV
#   2. the PMC bench command without a profiler (algorithmic bytes + launch times for tools/summarize_pmc_tree.py)
#   3. four counter passes over that command, one counter set each (FETCH_SIZE; WRITE_SIZE; read requests by size; write requests),
#      restricted to the real steps (rocprofv3 --selected-regions + bench.py's roctxProfilerResume / Pause under XQ_BENCH_ROCTX=1)
# A heartbeat line goes to <dir>/heartbeat.log every minute (counter passes print nothing for minutes).
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
( while true; do sleep 50; date >> "$OUT/heartbeat.log"; done ) &
HB=$!
trap "kill $HB 2>/dev/null" EXIT
PMCARGS="--steps 4 --warmup 2 --prewarm ${XQ_PMC_PREWARM:-400} --no-stagger --complete-games 0 --cpu-seconds 0"
export XQ_BENCH_ROCTX=1          # bench.py brackets the real steps; --selected-regions keeps the prewarm out of the counter passes
[ -n "$XQ_SKIP_STATS" ] || rocprofv3 --kernel-trace --stats -f csv -d "$OUT/prof" -- python3 bench.py --no-peaked --complete-games 0 --cpu-seconds 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof.err"
python3 bench.py $PMCARGS > "$OUT/bench_pmc_unprofiled.json" 2> "$OUT/bench_pmc_unprofiled.err" && \
timeout -k 10 200 rocprofv3 --selected-regions --pmc FETCH_SIZE --kernel-trace -f csv -d "$OUT/pmc_fetch" -- python3 bench.py $PMCARGS > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" && \
timeout -k 10 200 rocprofv3 --selected-regions --pmc WRITE_SIZE --kernel-trace -f csv -d "$OUT/pmc_write" -- python3 bench.py $PMCARGS > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" && \
timeout -k 10 200 rocprofv3 --selected-regions --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace -f csv -d "$OUT/pmc_rdreq" -- python3 bench.py $PMCARGS > "$OUT/pmc_rdreq.json" 2> "$OUT/pmc_rdreq.err" && \
timeout -k 10 200 rocprofv3 --selected-regions --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -f csv -d "$OUT/pmc_wrreq" -- python3 bench.py $PMCARGS > "$OUT/pmc_wrreq.json" 2> "$OUT/pmc_wrreq.err"
du -sh "$OUT"; ls "$OUT"
