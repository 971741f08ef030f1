#!/usr/bin/env python3
"""Fold one round's rocprofv3 passes (tools/profile_round.sh) into profiles/:

    python tools/pmc_traffic.py gpurun_out/prof_TAG TAG [workload] [WxH]

  profiles/TAG_kernel_stats.csv     the --stats summary (per kernel: calls, total, average, min, max)
  profiles/TAG_pmc_traffic.json     HBM bytes per kernel: (2*FETCH_SIZE + WRITE_SIZE) * 1024 per MI355X_MICROARCH.md
                                    (FETCH_SIZE / WRITE_SIZE count KiB; gfx950 FETCH_SIZE counts 128-B requests at 64 B)
  profiles/TAG_sq_counters.json     per kernel and launch: waves, VALU instructions, issue / wait shares
  profiles/TAG_bench.json           the bench line of the un-profiled run
  profiles/traffic.json             what bench.py reads: per-launch and per-frame bytes + the hash of the kernel
                                    sources the profile was taken on
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def counters(path):
    """{kernel: {counter: [values per dispatch]}}"""
    out = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            out[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


def main():
    d, tag = sys.argv[1], sys.argv[2]
    workload = sys.argv[3] if len(sys.argv) > 3 else "crash"
    res = sys.argv[4] if len(sys.argv) > 4 else "1920x1080"
    prof = os.path.join(ROOT, "profiles")
    one = lambda pat: sorted(glob.glob(os.path.join(d, pat)))[0]
    shutil.copy(one("stats/*kernel_stats.csv"), os.path.join(prof, tag + "_kernel_stats.csv"))
    shutil.copy(os.path.join(d, "bench.json"), os.path.join(prof, tag + "_bench.json"))
    shutil.copy(os.path.join(d, "bench_under_rocprof.json"), os.path.join(prof, tag + "_bench_under_rocprof.json"))
    fetch = counters(one("fetch/*counter_collection.csv"))
    write = counters(one("write/*counter_collection.csv"))
    timed = [k for k in fetch if k.startswith("k_trace_primary<") and not k.endswith(", true>")]
    frames = sum(len(fetch[k]["FETCH_SIZE"]) for k in timed) or 1
    rows, per_launch, per_frame, most = [], {}, {}, {}
    for k in fetch:
        if not k.startswith("k_"):
            continue
        f, w = fetch[k]["FETCH_SIZE"], write.get(k, {}).get("WRITE_SIZE", [0.0])
        favg, wavg = sum(f) / len(f), sum(w) / max(len(w), 1)
        b = int((2.0 * favg + wavg) * 1024.0)
        base = k.split("<")[0]
        rows.append({"kernel": k, "dispatches": len(f), "dispatches_per_frame": round(len(f) / frames, 2),
                     "FETCH_SIZE_KB_avg": round(favg, 1), "WRITE_SIZE_KB_avg": round(wavg, 1), "hbm_bytes_corrected": b})
        if ("<true, " in k and base in ("k_trace_dda_ray", "k_trace_dda_walk")) or \
                (k.endswith(", true>") and base == "k_trace_primary"):
            continue  # the counting variants (k_trace_dda_*<COUNT, REC>, k_trace_primary<REC, COUNT>) run once, outside the timed frames
        # (several instantiations of a kernel may run: the one launched most is the timed configuration's -- the bounce's
        # lean form with frames in flight, its split-walk form in the one-frame leg)
        if len(f) > most.get(base, 0):
            most[base] = len(f)
            per_launch[base] = b
        per_frame[base] = per_frame.get(base, 0) + int((2.0 * sum(f) + sum(w)) * 1024.0 / frames)
    per_launch["k_trace_dda"] = per_launch.get("k_trace_dda_walk", per_launch.get("k_trace_dda_ray", 0))
    rows.sort(key=lambda r: -r["hbm_bytes_corrected"] * r["dispatches"])
    import bench

    h = bench.kernel_source_hash()
    out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --steps 3 --warmup 1 "
                   "--cpu-seconds 0 --repeats 0: %d frames), per-dispatch averages; hbm_bytes_corrected = (2*FETCH_SIZE + "
                   "WRITE_SIZE)*1024 per MI355X_MICROARCH.md (calibrated there for wide coalesced reads, uncalibrated "
                   "for the 16-B gathers of the tracers)" % frames,
           "build": tag, "kernel_source_hash": h, "frames": frames, "frame_bytes": sum(per_frame.values()),
           "kernels": rows}
    json.dump(out, open(os.path.join(prof, tag + "_pmc_traffic.json"), "w"), indent=1)
    # SQ counters
    sq = counters(one("sq/*counter_collection.csv"))
    for extra in ("sq2", "l2"):
        g = glob.glob(os.path.join(d, extra + "/*counter_collection.csv"))
        if g:
            for k, c in counters(g[0]).items():
                sq[k].update(c)
    sqrows = {}
    for k, c in sq.items():
        if not k.startswith("k_"):
            continue
        a = {n: sum(v) / len(v) for n, v in c.items()}
        wc = a.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        a["share_issuing_any"] = round(a.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)
        a["share_waiting_waitcnt_or_barrier"] = round(a.get("SQ_WAIT_ANY", 0.0) / wc, 3)
        a["share_issue_stalled"] = round(a.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
        if a.get("TCC_HIT_sum") is not None:
            tot = a["TCC_HIT_sum"] + a.get("TCC_MISS_sum", 0.0)
            a["l2_hit_rate"] = round(a["TCC_HIT_sum"] / tot, 3) if tot else None
        sqrows[k] = {n: (round(v, 3 if (n.startswith('share_') or n == 'l2_hit_rate') else 1) if isinstance(v, float) else v)
                     for n, v in a.items()}
    json.dump({"note": "rocprofv3 --pmc, per-launch averages over the frames of bench.py --steps 3 --warmup 1; "
                       "SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles summed over the waves; "
                       "issuing + waiting + issue-stalled ~ 1", "build": tag, "kernel_source_hash": h,
               "kernels": sqrows}, open(os.path.join(prof, tag + "_sq_counters.json"), "w"), indent=1)
    tpath = os.path.join(prof, "traffic.json")
    tr = {}
    if os.path.exists(tpath):
        try:
            old = json.load(open(tpath))
            tr = {k: v for k, v in old.items() if isinstance(v, dict)}
        except Exception:
            tr = {}
    tr["%s:%s:scale1" % (workload, res)] = {"profile": "profiles/%s_pmc_traffic.json" % tag, "kernel_source_hash": h,
                                             "per_launch_bytes": per_launch, "per_frame_bytes": per_frame,
                                             "frame_bytes": sum(per_frame.values())}
    json.dump(tr, open(tpath, "w"), indent=1)
    print("frames", frames, "frame bytes %.2f GB" % (sum(per_frame.values()) / 1e9))
    for r in rows[:14]:
        print("%-34s x%5.2f/frame %10.1f MB" % (r["kernel"], r["dispatches_per_frame"], r["hbm_bytes_corrected"] / 1e6))


if __name__ == "__main__":
    main()
