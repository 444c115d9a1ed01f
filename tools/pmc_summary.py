"""Summarise rocprofv3 --pmc counter_collection CSVs (separate FETCH_SIZE / WRITE_SIZE passes, optionally a third pass with the
matrix-pipe counters) into profiles/<round>_pmc_summary.json: per kernel, the average raw FETCH_SIZE and WRITE_SIZE (KB) per launch
and, from the third pass, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F64 (× 512 = FLOP), GRBM_GUI_ACTIVE per launch.
bench.py applies the gfx950 x2 correction to FETCH_SIZE (MI355X_MICROARCH.md, HBM section).
usage: pmc_summary.py FETCH.csv WRITE.csv OUT.json [MFMA.csv]"""
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
if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
    ctr = {c: load(sys.argv[4], c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F64", "GRBM_GUI_ACTIVE")}
    for k in set().union(*[set(v) for v in ctr.values()]):
        if "boss::" not in k:
            continue
        e = out.setdefault(k, {})
        vals = {c: (ctr[c][k][0] / max(ctr[c][k][1], 1)) if k in ctr[c] else None for c in ctr}
        e["mfma_busy_cycles"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"]
        e["busy_cu_cycles"] = vals["SQ_BUSY_CU_CYCLES"]
        e["mfma_mops_f64"] = vals["SQ_INSTS_VALU_MFMA_MOPS_F64"]
        e["gui_active_cycles"] = vals["GRBM_GUI_ACTIVE"]
        e["mfma_launches"] = ctr["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, [0, 0])[1]
        if vals["SQ_VALU_MFMA_BUSY_CYCLES"] and vals["SQ_BUSY_CU_CYCLES"]:
            # matrix-pipe busy cycles (counted per SIMD, four per CU) over the cycles the kernel's CUs were busy
            e["mfma_busy_over_busy_cu"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / vals["SQ_BUSY_CU_CYCLES"]
        if vals["SQ_VALU_MFMA_BUSY_CYCLES"] and vals["GRBM_GUI_ACTIVE"]:
            e["mfma_util_chip"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] * 256 * 4)   # MfmaUtil: over every SIMD of the chip for the kernel's duration
out["source_hash"] = entry.source_hash()      # bench.py only accepts a summary collected on the sources it runs
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if "predict_kernel<" in k or "potrf_colupd" in k or k == "source_hash"}, indent=1))
