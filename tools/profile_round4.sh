#!/bin/bash
# Round-4 profile run on the GPU box (through gpurun):  bash tools/profile_round4.sh r04
#   * default bench command (config 3): kernel-trace stats + the separate PMC passes (as tools/profile_round.sh)
#   * kernel-trace stats of the other measured paths: forward + backward at config 3, the Poisson NSF step,
#     configs[4] as one rank of eight holds it (fp64, L = 4), configs[1]; the panel kernel (csrc/gemmp.hip) at M = 256
#     (where the library takes it) and at configs[1] (on request)
# Under rocprofv3 the program itself follows `--` (python3 ...): no env / bash -c hop.
set -e
tag=${1:-r04}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1
echo "trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -- $B > $out/pmc_l2.log 2>&1
echo "traffic passes done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_lds -- $B > $out/pmc_lds.log 2>&1
echo "pmc passes done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_bwd -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --with-backward > $out/trace_bwd.log 2>&1
echo "backward trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_poisson -- python3 tools/poisson_step.py 3 > $out/trace_poisson.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_poisson20 -- python3 tools/poisson_step.py 20 > $out/trace_poisson20.log 2>&1
echo "poisson traces done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_cfg5 -- python3 bench.py --config 5 --L 4 --steps 2 --warmup 1 --no-cpu-baseline > $out/trace_cfg5.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_cfg2 -- python3 bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > $out/trace_cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_m256 -- python3 bench.py --N 200000 --M 256 --L 32 --steps 5 --warmup 2 --no-cpu-baseline > $out/trace_m256.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_cfg2panel -- python3 bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline --panel-products > $out/trace_cfg2panel.log 2>&1   # (= trace_cfg2 since the panel kernel became the default there)
echo "config traces done"
ls $out
