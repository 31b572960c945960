#!/bin/bash
# Counter passes for profiles/ (run on the GPU box through gpurun; writes under gpurun_out/prof_$1).
# Separate passes, --pmc only with --kernel-trace (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not
# fit one pass; gpurun refuses --pmc combined with the trace domains).
set -e
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
run() { name=$1; shift; rocprofv3 --kernel-trace --output-format csv -d $out/$name "$@" -- python3 tools/kprobe.py > $out/$name.log 2>&1; }
run sq1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run occ --pmc VALUBusy OccupancyPercent MemUnitStalled
run atom --pmc TCC_EA0_ATOMIC_sum TCP_UTCL1_TRANSLATION_MISS_sum
run coexec --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-config5 > $out/stats.log 2>&1
find $out -name "*.csv" | head -40
