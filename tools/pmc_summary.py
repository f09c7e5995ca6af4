"""Per-kernel totals of rocprofv3 --pmc runs.

usage: python tools/pmc_summary.py OUT.json DIR [DIR ...]
Every DIR is the output directory of one `rocprofv3 --pmc ... -d DIR --output-format csv` run; all
*_counter_collection.csv files below it are read.  Result: {kernel: {"dispatches": n, counter: sum,
counter + "_max": largest single dispatch}}.
"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    res = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = {}
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    v = float(row["Counter_Value"])
                    e = res.setdefault(k, {})
                    e[c] = e.get(c, 0.0) + v
                    e[c + "_max"] = max(e.get(c + "_max", 0.0), v)
                    seen.setdefault((k, c), set()).add(row["Dispatch_Id"])
            for (k, c), ids in seen.items():
                res[k]["dispatches"] = max(res[k].get("dispatches", 0), len(ids))
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
