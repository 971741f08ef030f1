"""Fold rocprofv3 counter_collection CSVs: per kernel (name prefix filter) the per-dispatch average of every counter.

    python tools/pmc_fold.py PREFIX file1.csv [file2.csv ...]
"""
import csv, re, sys, json
from collections import defaultdict
prefix = sys.argv[1]
out = defaultdict(lambda: defaultdict(list))
for path in sys.argv[2:]:
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = re.sub(r"^void ", "", row["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            if not name.startswith(prefix):
                continue
            out[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}
print(json.dumps(res, indent=1))
