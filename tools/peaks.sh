#!/bin/bash
# Measured denominators: dense fp32 / fp64 MFMA issue rate and HBM streaming rates on the box.
#   bash tools/peaks.sh > gpurun_out/peaks.txt   (then `python3 tools/peaks_json.py gpurun_out/peaks.txt profiles/<round>/peaks.json`)
set -e
cd "$(dirname "$0")/.."
H="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3"
$H tools/mfma_probe_f32.hip -o /tmp/probe_f32 && /tmp/probe_f32
$H tools/mfma_probe_f64.hip -o /tmp/probe_f64 && /tmp/probe_f64
$H tools/hbm_probe.hip -o /tmp/hbm_probe && /tmp/hbm_probe
