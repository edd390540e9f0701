#!/bin/bash
# Instruction-mix / wait counters for the hot kernels:  tools/pmc_kernel.sh <tag> [bench args]
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d /tmp/pmc_$i -- python3 $R/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pmc_$i.log 2>&1
done
python3 $R/tools/pmc_summary.py /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3 > $R/gpurun_out/pmc_$TAG/summary.txt 2>&1
tail -3 /tmp/pmc_1.log >> $R/gpurun_out/pmc_$TAG/summary.txt
