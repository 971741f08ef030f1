#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes into profiles/<tag>_pmc_traffic.json and profiles/traffic.json.

    python tools/pmc_traffic.py FETCH.csv WRITE.csv TAG [workload] [WxH]

FETCH.csv / WRITE.csv are the *_counter_collection.csv files of
`rocprofv3 --pmc FETCH_SIZE --kernel-trace ...` and `--pmc WRITE_SIZE ...` runs of
`bench.py --steps 3 --warmup 1 --cpu-seconds 0` (separate passes, as
MI355X_MICROARCH.md prescribes).  Per kernel the per-dispatch averages are
combined as hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (the guide's gfx950
correction: FETCH_SIZE counts 128-B requests as 64 B).
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def averages(path, counter):
    tot = defaultdict(float)
    cnt = defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            name = re.sub(r"^void ", "", row["Kernel_Name"])
            name = re.sub(r"\(.*$", "", name)
            tot[name] += float(row["Counter_Value"])
            cnt[name] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main():
    fetch_csv, write_csv, tag = sys.argv[1:4]
    workload = sys.argv[4] if len(sys.argv) > 4 else "crash"
    res = sys.argv[5] if len(sys.argv) > 5 else "1920x1080"
    fetch, nf = averages(fetch_csv, "FETCH_SIZE")
    write, _ = averages(write_csv, "WRITE_SIZE")
    rows = []
    for k in fetch:
        if not k.startswith("k_"):
            continue
        fk, wk = fetch[k], write.get(k, 0.0)
        rows.append({"kernel": k, "dispatches": nf[k],
                     "FETCH_SIZE_KB_avg": round(fk, 1), "WRITE_SIZE_KB_avg": round(wk, 1),
                     "hbm_bytes_corrected": int((2.0 * fk + wk) * 1024.0)})
    rows.sort(key=lambda r: -r["hbm_bytes_corrected"])
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes "
                   "(bench.py --steps 3 --warmup 1 --cpu-seconds 0), per-dispatch averages; "
                   "hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md "
                   "(gfx950 FETCH_SIZE counts 128-B requests at 64 B; calibrated there for wide "
                   "coalesced reads, uncalibrated for the 16-B gathers of the tracers). Kernels "
                   "launched once per grid (k_count_*, k_fill, ...) average over the three grids.",
           "build": tag, "kernels": rows}
    with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    by = {r["kernel"].split("<")[0]: r["hbm_bytes_corrected"] for r in rows}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    tr = json.load(open(tpath)) if os.path.exists(tpath) else {}
    for stage, kern in (("shadow_cull", "k_shadow_cull"), ("trace_shadow", "k_trace_shadow"),
                        ("trace_primary", "k_trace_primary"), ("trace_dda", "k_trace_dda")):
        if kern in by:
            tr[f"{workload}:{stage}:{res}:scale1"] = by[kern]
    with open(tpath, "w") as f:
        json.dump(tr, f, indent=1)
    for r in rows[:12]:
        print(f"{r['kernel']:32s} {r['hbm_bytes_corrected'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
