#!/bin/bash
# Builds tools/wino2_probe (plain) and tools/wino2_probe_stamps (-DMI355_W2_STAMPS) from the package's conv3d.hip.
# Extra -D flags for ablations go in $1 (stamps build) / $2 (plain build).
set -e
cd "$(dirname "$0")/.."
PKG=automated-brain-mri-analysis-and-report-generation-with-retrieval-augmented-clinical-assistance_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DMI355_W2_STAMPS $1 -I$PKG/csrc tools/wino2_probe.hip -o tools/wino2_probe_stamps
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $2 -I$PKG/csrc tools/wino2_probe.hip -o tools/wino2_probe
# tools/h16_probe (+ _stamps, -DMI355_H16_STAMPS) from conv3d_f16.hip; extra -D flags in $3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -DMI355_H16_STAMPS $3 -I$PKG/csrc tools/h16_probe.hip -o tools/h16_probe_stamps
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -w $3 -I$PKG/csrc tools/h16_probe.hip -o tools/h16_probe
