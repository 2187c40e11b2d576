#!/bin/bash
# Separate rocprofv3 --pmc passes (<= 4 TCC counters each) over tools/prof_step.py; summaries to gpurun_out/.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-TGT}
TAG=${2:-$CFG}      # output name; kernel variants are chosen with PPO_XP_TUNABLES=name=value,...
ALGO=${3:-0}
TOVR=${4:-}      # optional: time steps instead of the config's T
i=0
for pmc in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "FETCH_SIZE TCC_EA0_RDREQ_DRAM_sum" \
           "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  # counters only for this library's kernels (ppoxp::...): the thousands of torch / insert dispatches that fill the buffer are
  # not instrumented (profiles/r3/README.md: what killed the full-size C3 passes in round 2)
  timeout -s KILL 300 rocprofv3 --pmc $pmc --kernel-include-regex "ppoxp" --output-format csv -d gpurun_out/pmc_${TAG}_$i -- python3 tools/prof_step.py $CFG 3 $ALGO $TOVR > gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_${TAG}_$i.log; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG}_1 gpurun_out/pmc_${TAG}_2 gpurun_out/pmc_${TAG}_3 gpurun_out/pmc_${TAG}_4 > gpurun_out/pmc_${TAG}_summary.txt
cat gpurun_out/pmc_${TAG}_summary.txt
# keep the merge small
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
