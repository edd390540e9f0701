#!/bin/bash
# Instruction-mix / wait counters for the hot kernels:  tools/pmc_kernel.sh <tag> [bench args]
# Only this library's kernels are instrumented (--kernel-include-regex 'cmb::'): the harness' torch kernels on
# tensors of more than 2^31 elements crash under --pmc (see tools/profile_round.sh).  Every rocprofv3 pass is
# checked; the program itself follows `--` (no env / bash -c hop: the profiler has initialised the GPU by then).
set -u
TAG=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CMB_SERIAL_SUBBATCHES=1
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-include-regex 'cmb::' --output-format csv -d /tmp/pmc_$i -- python3 $R/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-streaming --no-rlc > $OUT/pass_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then
    echo "rocprofv3 pass $i ($SET) failed with exit code $rc; see $OUT/pass_$i.log" | tee $OUT/summary.txt
    tail -20 $OUT/pass_$i.log
    exit $rc
  fi
done
python3 $R/tools/pmc_summary.py /tmp/pmc_1 /tmp/pmc_2 /tmp/pmc_3 > $OUT/summary.txt 2>&1 || { echo "pmc_summary.py failed"; exit 1; }
tail -3 $OUT/pass_1.log >> $OUT/summary.txt
