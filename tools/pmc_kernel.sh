#!/bin/bash
# Counter passes over tools/pairbench.py for the blend kernels (run on the GPU box through gpurun):
#   tools/pmc_kernel.sh <tag> [pairbench flags]   -> gpurun_out/pmc_<tag>/{sq1,sq2}/...counter_collection.csv
# --pmc only with --kernel-trace (gpurun refuses the other trace domains beside counters); the program itself after `--`.
set -e
tag=${1:-x}; shift || true
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --kernel-trace --output-format csv -d $out/$name "$@" -- python3 $GRAFT_REPO_ROOT/tools/pairbench.py --rounds 1 --reps 3 $EXTRA > $out/$name.log 2>&1; }
run sq1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA
run occ --pmc VALUBusy OccupancyPercent MemUnitStalled
echo done > $out/done.txt
