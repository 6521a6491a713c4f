#!/bin/bash
# round 3: instruction-order probe of a forward-shaped k step (tools/stream_order_probe.hip), NV = 30 / 45 vector instructions per k step
set -e
mkdir -p gpurun_out
: > gpurun_out/stream_order_probe.txt
for nv in 30 45 60; do
  hipcc -O3 --offload-arch=gfx950 -w -DNV=$nv tools/stream_order_probe.hip -o gpurun_out/sop_$nv
  timeout -k 10 120 gpurun_out/sop_$nv >> gpurun_out/stream_order_probe.txt
  rm -f gpurun_out/sop_$nv
done
cat gpurun_out/stream_order_probe.txt
