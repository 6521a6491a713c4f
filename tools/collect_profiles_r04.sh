#!/bin/bash
# Round-4 profile collection on the GPU box (outputs under gpurun_out/prof_r04/; summaries are made from them by tools/summarise_profiles_r04.py).
# rocprofv3 always wraps `python3 <script>` directly (no env / shell hop); --pmc passes are separate runs with --kernel-trace only.
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r04
mkdir -p $O
B="python3 bench.py --steps 20 --warmup 5 --no-sides --no-cpu-baseline --profile-steps 0 --min-reps 1 --min-seconds 0"
what=${1:-all}      # all | cql | few | algos | p2
if [ "$what" = all ] || [ "$what" = cql ]; then
  echo "== default 2x96: kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/default_stats -o r -- $B > $O/default_stats.log 2>&1
  echo "== 1x128: kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/1x128_stats -o r -- $B --engines-per-gpu 1 --runs-per-gpu 128 > $O/1x128_stats.log 2>&1
  export ORL_WS_ONE_ROUND=1      # one engine x 96 runs in the decomposition the two-engine default uses
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== 1x96 one round: pmc $c"; rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/1x96_$c -o r -- $B --engines-per-gpu 1 --runs-per-gpu 96 > $O/1x96_$c.log 2>&1
  done
  unset ORL_WS_ONE_ROUND
  echo "== 1x128: SQ counters"
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d $O/1x128_sq1 -o r -- $B --engines-per-gpu 1 --runs-per-gpu 128 > $O/1x128_sq1.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/1x128_sq2 -o r -- $B --engines-per-gpu 1 --runs-per-gpu 128 > $O/1x128_sq2.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/1x128_sq3 -o r -- $B --engines-per-gpu 1 --runs-per-gpu 128 > $O/1x128_sq3.log 2>&1 || echo "(SQ_VALU_MFMA_COEXEC_CYCLES pass failed)"
  python3 bench.py --steps 20 --warmup 5 --no-sides --no-cpu-baseline --engines-per-gpu 1 --runs-per-gpu 128 --profile-steps 20 --profile-dump $O/tags_1x128.txt > $O/bench_1x128.json 2> $O/bench_1x128.err
  python3 bench.py --steps 20 --warmup 5 --no-sides --no-cpu-baseline --engines-per-gpu 1 --runs-per-gpu 128 --precision 0 --profile-steps 10 --profile-dump $O/tags_fp32_1x128.txt > $O/bench_fp32_1x128.json 2> $O/bench_fp32_1x128.err
fi
if [ "$what" = all ] || [ "$what" = few ]; then
  for r in 1 8; do
    echo "== few runs: $r"; rocprofv3 --kernel-trace --output-format csv -d $O/few_$r -o r -- $B --engines-per-gpu 1 --runs-per-gpu $r --steps 40 > $O/few_$r.log 2>&1
  done
  python3 tools/few_runs_ab.py "ORL_FUSE_SMALL=0" "ORL_FUSE_SMALL=1" --runs 1 2 4 8 16 > $O/few_runs_ab.txt 2>/dev/null
  python3 tools/few_runs_ab.py "ORL_FUSE_SMALL=0" "ORL_FUSE_SMALL=1" --runs 1 8 --precision 0 > $O/few_runs_ab_fp32.txt 2>/dev/null
fi
if [ "$what" = all ] || [ "$what" = algos ]; then
  for a in iql td3bc edac cql_h3; do
    echo "== $a: tags + kernel stats + pmc"
    python3 tools/algo_run.py $a 128 1 30 --tags $O/tags_$a.txt > $O/run_$a.log 2>&1
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${a}_stats -o r -- python3 tools/algo_run.py $a 128 1 30 > $O/${a}_stats.log 2>&1
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/${a}_$c -o r -- python3 tools/algo_run.py $a 128 1 30 > $O/${a}_$c.log 2>&1
    done
  done
fi
if [ "$what" = all ] || [ "$what" = p2 ]; then
  P2="$B --engines-per-gpu 1 --runs-per-gpu 128 --precision 2"
  echo "== precision 2, 1x128: kernel stats"; rocprofv3 --kernel-trace --stats --output-format csv -d $O/p2_stats -o r -- $P2 > $O/p2_stats.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== precision 2, 1x128: pmc $c"; rocprofv3 --kernel-trace --output-format csv --pmc $c -d $O/p2_$c -o r -- $P2 > $O/p2_$c.log 2>&1
  done
  python3 bench.py --steps 20 --warmup 5 --no-sides --no-cpu-baseline --engines-per-gpu 1 --runs-per-gpu 128 --precision 2 --profile-steps 20 --profile-dump $O/tags_p2_1x128.txt > $O/bench_p2_1x128.json 2> $O/bench_p2_1x128.err
  python3 tools/few_runs_ab.py "-" --runs 1 8 16 32 128 --precision 0 --reps 2 > $O/few_runs_p0.txt 2>/dev/null
  python3 tools/few_runs_ab.py "-" --runs 1 8 16 32 128 --precision 2 --reps 2 > $O/few_runs_p2.txt 2>/dev/null
fi
# keep what the summaries need (the merge back is capped at 64 MiB): stats csv, counter csv, kernel trace csv of the few-runs passes
find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
du -sh $O
