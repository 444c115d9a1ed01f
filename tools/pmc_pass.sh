#!/bin/bash
# one rocprofv3 counter pass over a short bench run (GPU box, repo root): tools/pmc_pass.sh TAG "COUNTER1 COUNTER2 ..."
tag=$1; ctrs=$2
out=gpurun_out/pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $out/bench.json 2> $out/err.log
python3 - "$out" <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][-50:]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:14]:
    print(k, {c: f"{v / cnt[(k, c)]:.4g}" for c, v in d.items()}, "launches", max(cnt[(k, c)] for c in d))
PY
