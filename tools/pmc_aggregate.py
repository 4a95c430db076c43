#!/usr/bin/env python3
"""Turns rocprofv3 --pmc counter CSVs into profiles/pmc_traffic.json (what bench.py reports as roofline.traffic).

    python tools/pmc_aggregate.py --section config2_f32 --fetch <..._counter_collection.csv> --write <..._counter_collection.csv> \
        [--section ... --fetch ... --write ... for another workload] --out profiles/pmc_traffic.json --note "..." --digest <sha>

One section per workload (bench.py asks for "config2_f32", "config3_f16", "config3_f32": the same kernel has other launch
sizes in another workload).  --digest = the source digest of the library the passes ran on (lib/libmi355_nnunet.so.digest on
the GPU box); bench.py compares it with the running library's and reports `traffic_stale`.

Per kernel: the mean over its launches of FETCH_SIZE and WRITE_SIZE (KiB), and
hbm_bytes_per_launch = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.  FETCH_SIZE is doubled because gfx950 tallies
128-B requests as 64 B (MI355X_MICROARCH.md, HBM / rocprofv3 section); it counts MALL hits as well, so it is an
upper bound on DRAM reads.  Kernel names are reduced to the form the library's own profiler uses
("void mi355::name<...>(args)" -> "name<...>").
"""
import argparse
import csv
import json
import re
from collections import defaultdict


def short(name: str) -> str:
    m = re.match(r"^_ZN5mi355(\d+)", name)  # the demangler gives up on _Float16 parameters: take the bare name
    if m:
        n = int(m.group(1))
        base = name[m.end():m.end() + n]
        tm = re.match(r"^I(DF16_|f)E", name[m.end() + n:])
        return base + ({"DF16_": "<f16>", "f": "<f32>"}[tm.group(1)] if tm else "")
    name = re.sub(r"^void\s+", "", name)
    name = re.sub(r"^mi355::", "", name)
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut].strip()


def read(path, counter):
    acc = defaultdict(lambda: [0, 0.0])
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != counter:
                continue
            a = acc[short(row["Kernel_Name"])]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--section", action="append", required=True)
    ap.add_argument("--digest", default=None)
    ap.add_argument("--fetch", action="append", required=True)
    ap.add_argument("--write", action="append", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    out = {"_method": args.note, "_source_digest": args.digest}
    n = 0
    for sec, fpath, wpath in zip(args.section, args.fetch, args.write):
        f, w = read(fpath, "FETCH_SIZE"), read(wpath, "WRITE_SIZE")
        tab = out.setdefault(sec, {})
        for k in f:
            if k.startswith("__amd") or k not in w:
                continue
            fk, wk = f[k][1] / f[k][0], w[k][1] / w[k][0]
            tab[k] = {"launches": f[k][0], "FETCH_SIZE_KiB_per_launch": round(fk, 1), "WRITE_SIZE_KiB_per_launch": round(wk, 1),
                      "hbm_bytes_per_launch": int(2 * fk * 1024 + wk * 1024)}
            n += 1
    with open(args.out, "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"{args.out}: {n} kernel entries in {len(args.section)} sections")


if __name__ == "__main__":
    main()
