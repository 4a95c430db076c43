#!/usr/bin/env python3
"""Per-kernel summary of an SQ counter pass (rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU
SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY ... GRBM_GUI_ACTIVE): VALU instructions per MFMA and the busy fraction of
the matrix pipe (MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)).  The table in DESIGN.md section 5.

    python tools/sq_summary.py profiles/r01_pmc_sq_f32.csv profiles/r01_pmc_sq_f16.csv
"""
import collections
import csv
import sys


def main():
    for path in sys.argv[1:]:
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(path, newline="")):
            k = r["Kernel_Name"]
            if "mi355::" not in k:
                continue
            acc[k.split("(")[0].replace("void ", "").replace("mi355::", "")][r["Counter_Name"]] += float(r["Counter_Value"])
        print(path)
        for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
            mf = v.get("SQ_INSTS_MFMA", 0)
            if mf == 0:
                continue
            busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (v.get("GRBM_GUI_ACTIVE", 1) / 8 * 1024)
            print(f"  {k:48s} VALU/MFMA {v.get('SQ_INSTS_VALU', 0) / mf:5.2f}   matrix pipe busy {busy:5.3f}")


if __name__ == "__main__":
    main()
