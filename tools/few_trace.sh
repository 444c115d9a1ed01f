#!/bin/bash
# kernel trace of the first-call few-candidates path (run on the GPU box from the repo root): tools/few_trace.sh M tag
M=${1:-1024}; tag=${2:-few}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp_$tag -o few -- python3 /root/repo/tools/few_probe.py $M > /root/repo/gpurun_out/${tag}.log 2>&1
cp /tmp/fp_$tag/*kernel_stats.csv /root/repo/gpurun_out/${tag}_stats.csv
cp /tmp/fp_$tag/*kernel_trace.csv /root/repo/gpurun_out/${tag}_trace.csv
