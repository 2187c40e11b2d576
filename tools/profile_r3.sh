#!/bin/bash
# Round-3 evidence: rocprofv3 kernel stats of bench.py (TGT headline, then C3 and C4 alone) and the PMC passes.
# Run on the GPU box from the repo root: bash tools/profile_r3.sh [what...] ; results under gpurun_out/r3prof/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r3prof; mkdir -p $O
WHAT=${*:-stats pmc_TGT pmc_C4 pmc_C3}
for w in $WHAT; do
  case $w in
  stats)
    for cfg in TGT C3 C4; do
      timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- python3 bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof_$cfg.json 2> $O/bench_under_rocprof_$cfg.err || { echo "stats $cfg failed"; tail -3 $O/bench_under_rocprof_$cfg.err; exit 1; }
      f=$(find $O/stats_$cfg -name "*kernel_stats.csv" | head -1)
      { head -1 "$f"; grep ppoxp "$f"; } > $O/rocprof_kernel_stats_$cfg.csv
      echo "== $cfg"; cut -d, -f1-4 $O/rocprof_kernel_stats_$cfg.csv | cut -c1-150
    done ;;
  pmc_TGT) bash tools/pmc_passes.sh TGT TGT_r3 > $O/pmc_TGT.log 2>&1 || { echo "pmc TGT failed"; tail -5 $O/pmc_TGT.log; exit 1; }; cp gpurun_out/pmc_TGT_r3_summary.txt $O/ ;;
  pmc_C4) bash tools/pmc_passes.sh C4 C4_r3 257 > $O/pmc_C4.log 2>&1 || { echo "pmc C4 failed"; tail -5 $O/pmc_C4.log; exit 1; }; cp gpurun_out/pmc_C4_r3_summary.txt $O/ ;;
  pmc_C3) bash tools/pmc_passes.sh C3 C3_r3 > $O/pmc_C3.log 2>&1 || { echo "pmc C3 failed"; tail -25 $O/pmc_C3.log; tail -20 gpurun_out/pmc_C3_r3_1.log; exit 1; }; cp gpurun_out/pmc_C3_r3_summary.txt $O/ ;;
  esac
done
find gpurun_out -name "*.csv" -size +2M -delete
echo done
