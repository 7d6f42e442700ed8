#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch)."""
import collections, csv, glob, sys

def load(pattern):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            key = (name, r["Grid_Size"])
            out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            out[key]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return out

if __name__ == "__main__":
    for pat in sys.argv[1:]:
        d = load(pat)
        print("##", pat)
        for key, cs in sorted(d.items(), key=lambda kv: -sum(kv[1]["_dur_ns"])):
            n = len(cs["_dur_ns"])
            dur = sum(cs["_dur_ns"]) / n
            if dur < 2e5:
                continue
            line = f"{key[0][-28:]:28s} grid={key[1]:>9s} n={n:3d} dur_ms={dur/1e6:8.3f}"
            for c, v in cs.items():
                if c != "_dur_ns":
                    line += f" {c}={sum(v)/len(v):.4g}"
            print(line)
