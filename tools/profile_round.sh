#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats + separate PMC passes of the default
# bench command; raw outputs land in gpurun_out/prof_$1, summaries are copied to profiles/ by hand.
set -e
tag=${1:-r01}
export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
out=gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -- $B > $out/pmc_l2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_lds -- $B > $out/pmc_lds.log 2>&1
ls $out/*/*/ | head -30
