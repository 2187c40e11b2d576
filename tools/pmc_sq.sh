#!/bin/bash
# One rocprofv3 --pmc pass of SQ counters (<= 8) over tools/prof_step.py: where a kernel's wave cycles go.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CFG=${1:-TGT}
TAG=${2:-${CFG}_sq}
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d gpurun_out/pmc_${TAG} -- python3 tools/prof_step.py $CFG 3 0 > gpurun_out/pmc_${TAG}.log 2>&1 || { echo "pass failed"; tail -3 gpurun_out/pmc_${TAG}.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_${TAG} > gpurun_out/pmc_${TAG}_summary.txt
cat gpurun_out/pmc_${TAG}_summary.txt
find gpurun_out -name "*counter_collection.csv" -size +2M -delete
