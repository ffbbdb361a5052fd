#!/bin/bash
# builds and runs the micro-probes on the GPU box
cd "$(dirname "$0")" || exit 1
for p in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value $p.hip -o /tmp/$p && /tmp/$p
done
