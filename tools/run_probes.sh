#!/bin/bash
# builds and runs the round-3 micro-probes on the GPU box; outputs under gpurun_out/
set -e
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -w tools/fp16_split_probe.hip -o gpurun_out/fp16_split_probe
gpurun_out/fp16_split_probe > gpurun_out/fp16_split_probe.txt
for k in 1 4 3; do
  hipcc -O3 --offload-arch=gfx950 -w -DVKIND=$k tools/coexec_probe32.hip -o gpurun_out/coexec_probe32_v$k
  timeout -k 10 120 gpurun_out/coexec_probe32_v$k > gpurun_out/coexec_probe32_v$k.txt
done
rm -f gpurun_out/fp16_split_probe gpurun_out/coexec_probe32_v?
