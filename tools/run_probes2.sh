#!/bin/bash
# round 3: issue cost of the fp16-split building blocks (v_fma_mix_f32 vs v_cvt_f32_f16 + v_sub_f32) -- builds and runs on the GPU box
set -e
mkdir -p gpurun_out
hipcc -O3 --offload-arch=gfx950 -w tools/probes/fma_mix_check.hip -o gpurun_out/fma_mix_check && gpurun_out/fma_mix_check > gpurun_out/fma_mix_probe.txt
for k in 1 12 13 14 15 16 3; do
  hipcc -O3 --offload-arch=gfx950 -w -DVKIND=$k tools/coexec_probe.hip -o gpurun_out/cp_$k
  timeout -k 10 60 gpurun_out/cp_$k | grep -E "VKIND|64 VALU|interleaved" >> gpurun_out/fma_mix_probe.txt
done
rm -f gpurun_out/cp_* gpurun_out/fma_mix_check
