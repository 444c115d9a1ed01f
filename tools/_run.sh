set -e
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout -k 10 300 python tools/small_latency.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sl -o sl -- python3 tools/small_latency.py > gpurun_out/sl_prof.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof_sl/**/sl_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('winv','linv_col','kstar_args','append_')): print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2),'us min',round(float(r['MinNs'])/1e3,2))
PY
