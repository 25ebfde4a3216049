#!/bin/bash
# Round-5 profile run on the GPU box (through gpurun):  bash tools/profile_round5.sh r05
#   1. PMC passes (FETCH_SIZE / WRITE_SIZE / L2 / SQ, one pass each) of the timed loop of the default bench command
#      (--no-extra-legs: nothing but the timed evaluations), published at once as profiles/<tag>/traffic_stage1.json so that
#   2. the kernel-trace run of the same command prints a JSON line whose roofline.traffic is the figure just measured;
#   3. kernel-trace stats of the paths behind the other quoted numbers: the default bench line WITH its training legs and the
#      configs[1] record, configs[1] alone, the minibatch step, the Poisson step (E = 3, 20), VNNGP;
#   4. the text outputs: the NSF_benchmarks table, small steps, backward forms against fp64, Poisson floors
#      (tools/bench_suite.sh > profiles/<tag>/bench_suite.txt is a gpurun call of its own: it ends with the whole GPU test suite).
# Under rocprofv3 the program itself follows `--` (python3 ...): no env / bash -c hop.
set -e
tag=${1:-r05}
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
pub=profiles/$tag
mkdir -p $out $pub
B="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $B > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $B > $out/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2 -- $B > $out/pmc_l2.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq -- $B > $out/pmc_sq.log 2>&1
echo "pmc passes done"
python3 tools/publish_profile5.py traffic $out $pub
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $B > $out/trace.log 2>&1
echo "trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_train -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/trace_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_cfg2 -- python3 bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline > $out/trace_cfg2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_minibatch -- python3 tools/minibatch_step.py > $out/trace_minibatch.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_poisson -- python3 tools/poisson_step.py 3 > $out/trace_poisson.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_poisson20 -- python3 tools/poisson_step.py 20 > $out/trace_poisson20.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_vnngp -- python3 tools/vnngp_step.py > $out/trace_vnngp.log 2>&1
echo "traces done"
# PMC passes over the default bench line WITH its training legs (forward + backward at config 3 and configs[1]): how busy the
# matrix pipes are under the backward pass's kernels
T="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_sq_train -- $T > $out/pmc_sq_train.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_l2_train -- $T > $out/pmc_l2_train.log 2>&1
python3 tools/pmc_summary.py $out/pmc_sq_train $out/pmc_l2_train --json $pub/pmc_train.json > /dev/null
echo "train pmc done"
python3 tools/publish_profile5.py rest $out $pub
python3 tools/nsf_benchmark_step.py 300 > $pub/nsf_benchmark_steps.txt 2>&1 || true
python3 tools/small_step.py > $pub/small_step.txt 2>/dev/null || true
python3 tools/backward_forms_error.py > $pub/backward_forms_vs_fp64.txt 2>/dev/null || true
(python3 tools/poisson_step.py 3; python3 tools/poisson_step.py 20) > $pub/poisson_step.txt 2>/dev/null || true
cp -r $pub gpurun_out/published_$tag
ls $pub
