"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected in separate runs of the same command) per kernel.
usage: python tools/pmc_summary.py TAG FETCH_counter_collection.csv WRITE_counter_collection.csv [OUT.json]
Per kernel name: launches, the largest grid, and for the dispatch with the largest grid the raw counter values in KiB
(rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB).  Corrections (MI355X_MICROARCH.md, HBM section) are applied by the reader
(bench.py: FETCH_SIZE x2 for wide coalesced reads on gfx950, WRITE_SIZE as is)."""
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.split("(")[0]
    return name.replace("arkbp::", "")


def load(path):
    out = {}
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        d = out.setdefault(k, {"launches": 0, "largest_grid": 0, "value": 0.0, "vgprs": int(r["VGPR_Count"]), "scratch_bytes": int(r["Scratch_Size"]), "sum": 0.0})
        d["launches"] += 1
        d["sum"] += float(r["Counter_Value"])
        g = int(r["Grid_Size"])
        if g > d["largest_grid"]:
            d["largest_grid"], d["value"] = g, float(r["Counter_Value"])
    return out


def main():
    tag, fpath, wpath = sys.argv[1:4]
    f, w = load(fpath), load(wpath)
    res = {}
    for k in sorted(set(f) | set(w)):
        a, b = f.get(k), w.get(k)
        src = a or b
        res["%s/%s" % (tag, k)] = {
            "launches": src["launches"], "largest_grid": src["largest_grid"], "vgprs": src["vgprs"], "scratch_bytes": src["scratch_bytes"],
            "fetch_KiB_largest": a["value"] if a else None, "write_KiB_largest": b["value"] if b else None,
            "fetch_KiB_all_launches": a["sum"] if a else None, "write_KiB_all_launches": b["sum"] if b else None,
        }
    out = json.dumps(res, indent=1)
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(out + "\n")
    else:
        print(out)


if __name__ == "__main__":
    main()
