#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the package's HIP sources (hipcc -Rpass-analysis=kernel-resource-usage).
A kernel with a non-zero ScratchSize spills: on this path spill reloads after a chunk of streaming traffic miss every
cache level (round 2: 10k cycles per tile in conv3_f32_wino2_kernel), so the table is worth a look after every edit."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def main():
    files = sys.argv[1:] or ["conv3d", "conv3d_f16", "conv_stem", "tconv", "elementwise", "extras"]
    rows = []
    for f in files:
        res = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-c", os.path.join(PKG, f + ".hip"),
                              "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        cur = None
        for line in res.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = dict(name=m.group(1), file=f)
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
    names = demangle([r["name"] for r in rows])
    print(f"{'kernel':84s} {'VGPR':>5s} {'AGPR':>5s} {'scratch':>8s} {'spill':>6s} {'occ':>4s}")
    for r in rows:
        n = names[r["name"]].replace("mi355::", "").split("(")[0]
        print(f"{n[:84]:84s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('ScratchSize', 0):8d} {r.get('VGPRs Spill', 0):6d} {r.get('Occupancy', 0):4d}")


if __name__ == "__main__":
    main()
