#!/bin/bash
# Builds the hardware-probe helpers of tools/ (not part of the product): tools/native/lib<name>.so from <name>.hip, gfx950.
cd "$(dirname "$0")"
for f in l2_rate mfma_calib tile_stream graph_launch_mt request_rate; do
  [ -f lib$f.so ] && [ lib$f.so -nt $f.hip ] && continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -shared -fPIC $f.hip -o lib$f.so || exit 1
done
