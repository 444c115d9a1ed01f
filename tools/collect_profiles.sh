#!/bin/bash
# Collect the round's rocprofv3 evidence for bench.py (run on the GPU box from the repo root):
#   1. kernel trace + stats of the default bench command           -> profiles/<tag>_bench_kernel_stats.csv
#   2. separate --pmc passes for FETCH_SIZE, WRITE_SIZE and the MFMA-pipe counters -> profiles/<tag>_pmc_summary.json (carries the source hash)
#   3. the bench line itself (no profiler)                           -> profiles/<tag>_bench.json
#   4. device timeline of one N=4096 update under the resident chain -> profiles/<tag>_chain_timeline.log (needs tools/libbosship_t3.so:
#      python tools/chain_trace3.py --build on the CPU box first)
# usage: bash tools/collect_profiles.sh r03_v1
set -e
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p $out profiles
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_stats.json 2> $out/stats.err
cp $(find $out/stats -name "s_kernel_stats.csv" | head -1) profiles/${tag}_bench_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_f -o f -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_f.json 2> $out/f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_w -o w -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $out/bench_w.json 2> $out/w.err
# third pass: matrix-pipe utilisation (SQ block: 8 slots) — busy cycles of the MFMA pipe, busy cycles of the CUs, fp64 MFMA operations
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_m -o m -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --with-config5 > $out/bench_m.json 2> $out/m.err || true
python3 tools/pmc_summary.py $(find $out/pmc_f -name "f_counter_collection.csv" | head -1) $(find $out/pmc_w -name "w_counter_collection.csv" | head -1) profiles/${tag}_pmc_summary.json $(find $out/pmc_m -name "m_counter_collection.csv" | head -1)
python3 bench.py --steps 50 --warmup 5 > profiles/${tag}_bench.json 2> $out/bench.err
if [ -f tools/libbosship_t3.so ]; then python3 tools/chain_trace3.py 4096 > profiles/${tag}_chain_timeline.log 2> $out/trace.err || true; fi
cp profiles/${tag}_pmc_summary.json profiles/${tag}_bench_kernel_stats.csv profiles/${tag}_bench.json gpurun_out/
[ -f profiles/${tag}_chain_timeline.log ] && cp profiles/${tag}_chain_timeline.log gpurun_out/
echo "profiles written for $tag"
