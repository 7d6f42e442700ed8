#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE collected
separately, as MI355X_MICROARCH.md §HBM / §rocprofv3 PMC slots prescribe).

gfx950 corrections applied exactly as the guide states them: both counters are in KiB; FETCH_SIZE reports
half of the bytes of wide (16 B/lane) coalesced streaming reads, so it is doubled; WRITE_SIZE is exact for
16-B-per-lane streaming stores.

    python tools/pmc_traffic.py gemm_f32_kernel <fetch counter_collection.csv> <write counter_collection.csv> out.json [--exclude SUBSTR ...] [--workload 30sx32|10minx4|...]

--exclude drops dispatches whose (demangled) kernel name contains SUBSTR: the projection / conv GEMM bucket of bench.py is the
gemm_f16x3_dma_kernel template WITHOUT its relative-position-table instantiation ("<0, false, 2, 2, 2,": 128x128 tiles, K = 64) and
its positional-conv instantiation ("<4, false, 8, 1,"), which have buckets of their own.
"""
import csv
import json
import sys


EXCLUDE = []


def total(path, kernel, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter and not any(x in r["Kernel_Name"] for x in EXCLUDE):
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


def main():
    kernel, fpath, wpath, out = sys.argv[1:5]
    rest = sys.argv[5:]
    workload = "30sx32"  # the key bench.py looks the file up by: <clip seconds>sx<batch> or <minutes>minx<batch>
    while rest:
        assert rest[0] in ("--exclude", "--workload") and len(rest) >= 2, rest
        if rest[0] == "--exclude":
            EXCLUDE.append(rest[1])
        else:
            workload = rest[1]
        rest = rest[2:]
    f, nf = total(fpath, kernel, "FETCH_SIZE")
    w, nw = total(wpath, kernel, "WRITE_SIZE")
    assert nf == nw and nf > 0, (nf, nw)
    res = {"kernel": kernel, "workload": workload, "excluded_instantiations": EXCLUDE, "launches": nf, "fetch_kib_raw_per_launch": f / nf, "write_kib_per_launch": w / nw,
           "hbm_bytes_per_launch": (2.0 * f / nf + w / nw) * 1024.0,
           "correction": "FETCH_SIZE x2 (gfx950 wide coalesced reads are tallied at half), WRITE_SIZE x1, both KiB",
           "source": [fpath, wpath]}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
