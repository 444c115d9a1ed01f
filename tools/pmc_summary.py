"""Summarise rocprofv3 --pmc counter_collection CSVs (separate FETCH_SIZE / WRITE_SIZE passes) into
profiles/<round>_pmc_summary.json: per kernel, the average raw FETCH_SIZE and WRITE_SIZE (KB) per launch.
bench.py applies the gfx950 x2 correction to FETCH_SIZE (MI355X_MICROARCH.md, HBM section)."""
import csv, json, os, sys, collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry

def load(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            a = acc[name]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return acc

fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
out = {}
for k, (v, n) in fetch.items():
    if not k.startswith("boss::") and "boss::" not in k:
        continue
    w = write.get(k, [0.0, 1])
    out[k] = {"fetch_kb_raw": v / n, "write_kb": w[0] / max(w[1], 1), "launches": n}
out["source_hash"] = entry.source_hash()      # bench.py only accepts a summary collected on the sources it runs
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "predict" in k or k == "source_hash"}, indent=1))
