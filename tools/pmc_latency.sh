#!/bin/bash
# Mean latency of vector-memory and LDS instructions of the dominant kernels: SQ_INST_LEVEL_* / SQ_INSTS_* (separate pass).
set -e
export TMPDIR=/tmp
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
out=gpurun_out/prof_lat${1:-}
mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_lat -- $B > $out/pmc_lat.log 2>&1
python3 tools/pmc_summary.py $out/pmc_lat --json $out/pmc_lat.json > $out/pmc_lat.txt
