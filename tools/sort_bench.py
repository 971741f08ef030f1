#!/usr/bin/env python3
"""Per-pass rate of the pair sort (csrc/ugrt_sort.hip) at the sizes of a frame's seven sorts.

    python tools/sort_bench.py [--out gpurun_out/sort_bench.json]

A pass moves 16 B per pair (key + value in, key + value out); the histogram kernel reads the keys once
more per sort (4 B per pair).  GB/s = pairs * (16 * passes + 4) / time; the keys mimic the builds' (cell
ids in fill order, i.e. long runs of a few neighbouring digits) and the ray sort's (random cells).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--items", type=int, default=0, help='option "sort_items" (pairs per thread of a pass: 8 or 16)')
    ap.add_argument("--quick", action="store_true", help="16- and 32-bit random keys only")
    args = ap.parse_args()
    import torch
    import ugrt

    ctx = ugrt.Context(256, 256, device=0)
    if args.items:
        ctx.set_option("sort_items", args.items)
    rows = []
    for n in (1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 1 << 22, 1 << 23):
        for bits in ((16, 32) if args.quick else (8, 16, 24, 32)):
            for shape in (("random",) if args.quick else ("random", "runs")):
                g = torch.Generator(device="cuda").manual_seed(n ^ bits)
                hi = (1 << bits) - 1
                if shape == "random":
                    k = torch.randint(0, hi + 1, (n,), generator=g, device="cuda", dtype=torch.int64)
                else:  # neighbouring cells, as k_fill emits them
                    k = (torch.arange(n, device="cuda", dtype=torch.int64) // 37 * 2654435761) & hi
                    k = (k + torch.randint(0, 4, (n,), generator=g, device="cuda", dtype=torch.int64)) & hi
                k = k.to(torch.uint32) if hasattr(torch, "uint32") else k.to(torch.int32)
                k = k.view(torch.int32)
                v = torch.arange(n, device="cuda", dtype=torch.int32)
                ko, vo = torch.empty_like(k), torch.empty_like(v)
                res = {}
                for lib in (False, True):
                    for _ in range(3):
                        ctx.sort_pairs(k, ko, v, vo, bits, library=lib)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(args.reps):
                        ctx.sort_pairs(k, ko, v, vo, bits, library=lib)
                    torch.cuda.synchronize()
                    res[lib] = (time.perf_counter() - t0) / args.reps * 1e6
                passes = (bits + 7) // 8
                byts = n * (16 * passes + 4)
                row = {
                    "pairs": n, "key_bits": bits, "keys": shape, "passes": passes, "us": round(res[False], 2),
                    "us_per_pass": round(res[False] / passes, 2), "GB_per_s": round(byts / res[False] / 1e3, 1),
                    "us_rocprim": round(res[True], 2),
                }
                rows.append(row)
                print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"rows": rows, "note": __doc__.strip().splitlines()[0]}, f, indent=1)


if __name__ == "__main__":
    main()
