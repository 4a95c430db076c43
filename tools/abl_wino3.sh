#!/bin/bash
# Ablation builds of tools/wino3_probe (MI355_W3_ABL bits: 1 no epilogue, 4 no brick DMA, 8 no weight loads, 16 no input transform, 32 no stores)
set -e
cd "$(dirname "$0")/.."
PKG=automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DMI355_W3_ABL=$a -I$PKG/csrc tools/wino3_probe.hip -o tools/wino3_probe_abl$a &
done
wait
