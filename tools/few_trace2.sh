#!/bin/bash
# tools/few_trace2.sh RUNS KERNELS_PER_STEP [label=ENV=VAL ...]: rocprofv3 kernel trace of the few-runs regime, one pass per "label=ENV=VAL[,ENV=VAL]" spec
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/few_trace
mkdir -p $O
R=$1; N=$2; shift 2
for spec in "$@"; do
  label=${spec%%=*}; envs=${spec#*=}
  B="python3 bench.py --steps 40 --warmup 5 --no-sides --no-cpu-baseline --profile-steps 0 --min-reps 1 --min-seconds 0 --engines-per-gpu 1 --runs-per-gpu $R"
  ( for kv in ${envs//,/ }; do export "$kv"; done
    rm -rf $O/$label
    rocprofv3 --kernel-trace --output-format csv -d $O/$label -o r -- $B > $O/$label.log 2>&1 )
  f=$(find $O/$label -name "*kernel_trace.csv" | head -1)
  echo "== $label ($envs; $R runs)"; python3 tools/trace_summary.py $f $N 20
done
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
