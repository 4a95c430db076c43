#!/bin/bash
# Builds tools/wino3_probe (plain) and tools/wino3_probe_stamps (-DMI355_W3_STAMPS); extra -D flags in $1
set -e
cd "$(dirname "$0")/.."
PKG=automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $1 -I$PKG/csrc tools/wino3_probe.hip -o tools/wino3_probe &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DMI355_W3_STAMPS $1 -I$PKG/csrc tools/wino3_probe.hip -o tools/wino3_probe_stamps &
wait
