mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03 > gpurun_out/r3/profile_round.log 2>&1; tail -5 gpurun_out/r3/profile_round.log
python3 tools/publish_profile.py gpurun_out/prof_r03 gpurun_out/r3/published > gpurun_out/r3/publish.log 2>&1; cat gpurun_out/r3/publish.log
python3 tools/fp32_gap.py > gpurun_out/r3/fp32_gap.txt 2>&1; cat gpurun_out/r3/fp32_gap.txt
bash tools/peaks.sh > gpurun_out/r3/peaks.txt 2>&1; tail -12 gpurun_out/r3/peaks.txt
