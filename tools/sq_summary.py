"""Fold rocprofv3 --pmc SQ-counter passes (several passes, a few counters each) into per-kernel per-launch means.

Usage: python tools/sq_summary.py <dir holding one sub-directory per pass> <out.json> "<note>"
"""
import csv, glob, json, re, sys
from collections import defaultdict


def main():
    root, out, note = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = defaultdict(float)
        names = {}
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            per_dispatch[key] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sba11::", "")
        for (d, c), v in per_dispatch.items():
            acc[names[d]][c].append(v)
    res = {k: {c: sum(v) / len(v) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
    json.dump({"note": note, "kernels": res}, open(out, "w"), indent=1)
    for k, cs in res.items():
        if "schur" in k:
            print(k)
            for c, v in cs.items():
                print(f"   {c:32s} {v:16.0f}")


if __name__ == "__main__":
    main()
