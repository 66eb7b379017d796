#!/bin/bash
# builds tools/_bin/ff_fused_probe (gfx950); the two-launch baseline comes from streamingt2v_amd/libsvdhip.so (rpath relative to the binary)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -Wno-unused-result -Wno-inline-asm tools/ff_fused_probe.hip -o tools/_bin/ff_fused_probe \
    -Lstreamingt2v_amd -lsvdhip -Wl,-rpath,'$ORIGIN/../../streamingt2v_amd'
