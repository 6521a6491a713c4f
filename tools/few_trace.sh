#!/bin/bash
# tools/few_trace.sh RUNS KERNELS_PER_STEP_A KERNELS_PER_STEP_B : rocprofv3 kernel trace of the few-runs regime with ORL_FUSE_SMALL=0 (A) and =1 (B)
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/few_trace
mkdir -p $O
B="python3 bench.py --steps 40 --warmup 5 --no-sides --no-cpu-baseline --profile-steps 0 --min-reps 1 --min-seconds 0 --engines-per-gpu 1 --runs-per-gpu $1"
for v in 0 1; do
  export ORL_FUSE_SMALL=$v
  rm -rf $O/t$v
  rocprofv3 --kernel-trace --output-format csv -d $O/t$v -o r -- $B > $O/t$v.log 2>&1
  f=$(find $O/t$v -name "*kernel_trace.csv" | head -1)
  n=$2; [ $v = 1 ] && n=$3
  echo "== ORL_FUSE_SMALL=$v ($1 runs)"; python3 tools/trace_summary.py $f $n 20
done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
