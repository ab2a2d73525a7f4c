"""Fold rocprofv3 --pmc counter CSVs (one pass per counter) into per-kernel HBM traffic per launch.

Usage: python tools/pmc_summary.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass> <out.json>

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  On gfx950 FETCH_SIZE counts 64 B per 128-B request of wide
coalesced reads (MI355X_MICROARCH.md, HBM section): `fetch_x2_MB` doubles it as that guide prescribes; WRITE_SIZE is exact
for streaming stores.  `raw_MB` keeps the uncorrected sum.
"""
import csv, glob, json, re, sys
from collections import defaultdict


def per_kernel(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void sba::", "").replace("sba::", "")
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fd, wd, out = sys.argv[1:4]
    F, W = per_kernel(fd, "FETCH_SIZE"), per_kernel(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(F) | set(W)):
        f, w = F.get(k, 0.0), W.get(k, 0.0)
        res[k] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "raw_MB": (f + w) / 1024.0, "fetch_x2_MB": (2 * f + w) / 1024.0}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res.items():
        print(f"{k[:50]:50s} fetch {v['FETCH_SIZE_KB']:10.1f} KB  write {v['WRITE_SIZE_KB']:10.1f} KB  traffic(fetch x2) {v['fetch_x2_MB']:8.2f} MB")


if __name__ == "__main__":
    main()
