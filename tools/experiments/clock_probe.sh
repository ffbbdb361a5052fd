#!/bin/bash
# Builds the clock probe (a library of its own under gpurun_out/) and runs tools/probes/clock_probe.py.  GPU box.
R="$(cd "$(dirname "$0")/../.." && pwd)"
mkdir -p "$R/gpurun_out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -fPIC -shared "$R/tools/probes/clock_probe.hip" -o "$R/gpurun_out/libclockprobe.so" || exit 1
python3 "$R/tools/probes/clock_probe.py"
