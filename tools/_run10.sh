mkdir -p gpurun_out/r3 && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3/bench_w8.json 2>/dev/null
python3 -c "
import json; r=json.loads(open('gpurun_out/r3/bench_w8.json').read().strip().splitlines()[-1]); print(r['ms_per_step'], r['roofline']['achieved'], r['kernels']['stage2_LuT_Wt']['achieved_TFLOPs'])"
bash tools/profile_round.sh r03b > gpurun_out/r3/profile_round_b.log 2>&1
python3 tools/publish_profile.py gpurun_out/prof_r03b gpurun_out/r3/published_b 2>&1 | tail -6
