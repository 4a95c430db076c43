#!/usr/bin/env python3
"""Average shader clock per kernel, from a rocprofv3 --pmc pass that holds GRBM_GUI_ACTIVE (the SQ passes of
tools/collect_profiles.sh do): GRBM_GUI_ACTIVE counts the cycles the graphics engine was active, summed over the 8 XCDs,
and the counter CSV carries each dispatch's start / end timestamps in ns, so

    clock [GHz] = sum(GRBM_GUI_ACTIVE) / 8 / sum(end - start)

over the launches of a kernel.  This is the clock the chip actually granted while THAT kernel ran (DVFS), the number that
turns "fraction of the nominal 2.5 PFLOP/s (at 2.4 GHz)" into "fraction of what the clock allowed".  Launches shorter than
50 us are left out (timestamp granularity).

    python tools/kernel_clocks.py config3_f16=profiles/r04_pmc_sq_f16.csv config2_f32=profiles/r04_pmc_sq_f32.csv \
        [--json profiles/kernel_clocks.json] [--digest <sha>]

--json writes {section: {kernel: {"shader_clock_ghz": .., "launches": .., "avg_ms": .., "source": csv}}} for bench.py's roofline
block (sections = workloads, as profiles/pmc_traffic.json) and the digest of the library the passes ran on.
"""
import collections
import csv
import json
import sys


def short(name):
    name = name.split("(")[0].replace("void ", "").replace("mi355::", "")
    if name.startswith("_ZN5mi355"):  # (the demangler gives up on _Float16 parameters)
        digits = ""
        rest = name[len("_ZN5mi355"):]
        while rest and rest[0].isdigit():
            digits, rest = digits + rest[0], rest[1:]
        name = rest[:int(digits)] + "<...>"
    return name


def main():
    args = sys.argv[1:]
    out_json = None
    if "--json" in args:
        i = args.index("--json")
        out_json = args[i + 1]
        del args[i:i + 2]
    digest = None
    if "--digest" in args:
        i = args.index("--digest")
        digest = args[i + 1]
        del args[i:i + 2]
    table = {}
    for spec in args:
        section, path = spec.split("=", 1) if "=" in spec else ("", spec)
        sec = table.setdefault(section, {}) if section else table
        acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
        for r in csv.DictReader(open(path, newline="")):
            if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
                continue
            ns = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            if ns < 50e3:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += ns; a[2] += 1
        print(path)
        print(f"  {'kernel':62s} launches   avg ms   shader clock")
        for k, (cyc, ns, n) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            print(f"  {k[:62]:62s} {n:8d} {ns / n / 1e6:8.3f}   {cyc / 8 / ns:5.2f} GHz")
            sec[k] = dict(shader_clock_ghz=round(cyc / 8 / ns, 3), launches=n, avg_ms=round(ns / n / 1e6, 4), source=path)
    if out_json:
        table["_method"] = ("rocprofv3 --pmc ... GRBM_GUI_ACTIVE passes of `python3 bench.py --steps 1 --warmup 0 --no-secondary --no-cpu-baseline` "
                            "(config 2 f32 / config 3 f16, tools/collect_profiles.sh): sum(GRBM_GUI_ACTIVE) / 8 XCDs / sum(dispatch end - start), "
                            "launches >= 50 us, tools/kernel_clocks.py")
        table["_source_digest"] = digest
        json.dump(table, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
