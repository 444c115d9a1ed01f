#!/bin/bash
# Collect the round's rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command           -> profiles/<tag>_bench_kernel_stats.csv
#   2. separate --pmc passes for FETCH_SIZE, WRITE_SIZE and the MFMA-pipe counters -> profiles/<tag>_pmc_summary.json (carries the source hash)
#   3. the bench line itself (no profiler)                           -> profiles/<tag>_bench.json
#   4. device timeline of one N=4096 update under the resident chain -> profiles/<tag>_chain_timeline.log (the traced build of the same
#      sources, -DBOSS_CHAIN_TRACE, is compiled here into /tmp: nothing but the shipped library travels in the tree)
# usage: bash tools/collect_profiles.sh r03_v1
set -e
tag=${1:-r03}
out=/tmp/prof_$tag          # raw profiler output stays on the box (hundreds of MB); only the summaries travel back
mkdir -p $out profiles gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_stats.json 2> $out/stats.err
cp $(find $out/stats -name "s_kernel_stats.csv" | head -1) profiles/${tag}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_f -o f -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_f.json 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w -o w -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_w.json 2> $out/w.err
# third pass: matrix-pipe utilisation (SQ block: 8 slots) — busy cycles of the MFMA pipe, busy cycles of the CUs, fp64 MFMA operations
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_m -o m -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --with-config5 > $out/bench_m.json 2> $out/m.err || true
python3 tools/pmc_summary.py $(find $out/pmc_f -name "f_counter_collection.csv" | head -1) $(find $out/pmc_w -name "w_counter_collection.csv" | head -1) profiles/${tag}_pmc_summary.json $(find $out/pmc_m -name "m_counter_collection.csv" | head -1)
python3 bench.py --steps 50 --warmup 5 > profiles/${tag}_bench.json 2> $out/bench.err
export BOSS_TRACE_LIB=/tmp/libbosship_t3_$tag.so
python3 tools/chain_trace3.py --build > $out/trace_build.log 2>&1 || true
if [ -f $BOSS_TRACE_LIB ]; then
    python3 tools/chain_trace3.py 4096 > profiles/${tag}_chain_timeline.log 2> $out/trace.err || true
    python3 tools/rider_timeline.py > profiles/${tag}_rider_timeline.log 2> $out/rider_trace.err || true
    M=2048 python3 tools/rider_timeline.py > profiles/${tag}_rider_timeline_M2048.log 2>> $out/rider_trace.err || true
fi
echo "profiles written for $tag"
python3 tools/rider_probe.py > profiles/${tag}_rider_probe.log 2> $out/rider_probe.err || true
python3 tools/rider_time.py > profiles/${tag}_rider_time.log 2> $out/rider_time.err || true
python3 tools/stall_hunt.py > profiles/${tag}_update_latency_2000.log 2> $out/stall.err || true
# 300 back-to-back updates under the kernel trace: chain-kernel durations and main-stream gaps inside every update
REP=300 rocprofv3 --kernel-trace --output-format csv -d $out/upd300 -o u -- python3 tools/stall_hunt.py > $out/upd300.log 2> $out/upd300.err || true
python3 tools/trace_gaps.py $(find $out/upd300 -name "u_kernel_trace.csv" | head -1) > profiles/${tag}_update_trace_gaps.log 2>> $out/upd300.err || true
mkdir -p gpurun_out/profiles_$tag && cp profiles/${tag}_* gpurun_out/profiles_$tag/ && cp $out/*.err gpurun_out/profiles_$tag/ 2>/dev/null || true
