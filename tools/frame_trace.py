#!/usr/bin/env python3
"""Print one frame of a rocprofv3 --kernel-trace CSV of bench.py as a timeline (kernel, workgroups, µs, gap before).

    rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0
    python tools/frame_trace.py out/t_kernel_trace.csv [name filter]
"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_count_persp")]
st, en = idx[-3], idx[-2]
prev, busy = None, 0.0
for r in rows[st:en]:
    n = r["Kernel_Name"].replace("void ", "").split("(")[0][:34]
    if "rocprim" in n:
        n = "rocprim"
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    prev = e
    busy += (e - s) / 1e3
    if flt in n:
        print(f"{n:36s} wg {int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']):6d} {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  vgpr {r['VGPR_Count']} lds {r['LDS_Block_Size']}")
t0, t1 = int(rows[st]["Start_Timestamp"]), int(rows[en]["Start_Timestamp"])
print(f"frame {(t1 - t0) / 1e3:.1f} us, kernels {busy:.1f} us, idle {(t1 - t0) / 1e3 - busy:.1f} us, launches {en - st}")
