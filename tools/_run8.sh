mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3/poisson_trace -- python3 tools/poisson_step.py > gpurun_out/r3/poisson_trace.log 2>&1
f=$(ls gpurun_out/r3/poisson_trace/*/*kernel_stats.csv | head -1); head -12 $f | cut -c1-200
