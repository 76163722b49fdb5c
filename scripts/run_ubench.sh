#!/bin/bash
# Builds and runs the VALU issue-cost microbenchmarks; output is committed under profiles/<round>/ubench_valu.txt
set -e
cd "$(dirname "$0")/ubench"
for f in valu_rate valu_rate2 mfma_coissue; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-value $f.hip -o $f
  echo "== $f"; ./$f
done
