#!/bin/bash
# rocprofv3 kernel stats of bench.py for the configs given (default TGT C3 C4): the first half of tools/profile_r2.sh.
# Run on the GPU box from the repo root; results under gpurun_out/r2prof/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2prof; mkdir -p $O
for cfg in ${@:-TGT C3 C4}; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$cfg -- python3 bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $O/bench_under_rocprof_$cfg.json 2> $O/bench_under_rocprof_$cfg.err || { echo "stats $cfg failed"; tail -3 $O/bench_under_rocprof_$cfg.err; exit 1; }
  f=$(find $O/stats_$cfg -name "*kernel_stats.csv" | head -1)
  { head -1 "$f"; grep ppoxp "$f"; } > $O/rocprof_kernel_stats_$cfg.csv
  echo "== $cfg"; cut -d, -f1-4 $O/rocprof_kernel_stats_$cfg.csv | cut -c1-150
done
find gpurun_out -name "*.csv" -size +2M -delete
echo done
