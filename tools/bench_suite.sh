#!/bin/bash
# Every secondary number quoted in DESIGN.md / README.md, in one run on the GPU box:
#   bash tools/bench_suite.sh > gpurun_out/bench_suite.txt 2>&1   (then copied to profiles/<round>/)
set -e
one() { python3 - "$1" <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k = r["kernels"]
clk = r.get("clocks") or {}
fac = ("factor (Cholesky + inverse, one launch) %.3f ms = %.1f TF fp64" % (k["factor"]["ms_per_eval"], k["factor"]["achieved_TFLOPs"])
       if "factor" in k else "potrf %.2f ms (trailing %.1f TF) | trtri %.2f ms" % (
           k["potrf_ms_per_eval"], k["potrf_trailing"]["achieved_TFLOPs"], k["trtri_ms_per_eval"]))
panel = r["roofline"]["kernel"].startswith("panel_kernel")
print("  value %.4f %s | %.2f ms/step | %s %.1f TF (frac %.3f; sclk %.0f MHz: frac at that clock %.3f) | %s %.1f TF | kfill %.0f GB/s | %s%s" % (
    r["value"], r["unit"], r["ms_per_step"], "both products (one launch)" if panel else "stage1", r["roofline"]["achieved"], r["roofline"]["frac"],
    clk.get("sclk_MHz_mean", 0.0), r["roofline"].get("frac_at_sampled_clock", 0.0), "(stage2: same launch)" if panel else "stage2",
    k["stage2_LuT_Wt"]["achieved_TFLOPs"],
    k["kuf_fill"]["achieved_GBps"] or 0.0, fac,
    (" | fwd+bwd %s ms" % {a: round(b, 1) for a, b in r["forward_backward_ms"].items()}) if "forward_backward_ms" in r else ""))
PY
}
echo "== config 3 (default): N=200k M=2048 L=32 Matern-3/2 fp32, with forward+backward"
python3 bench.py --no-cpu-baseline > /tmp/b.log 2>/dev/null; one /tmp/b.log
echo "== config 2: N=50k M=512 L=8 RBF fp32, with forward+backward"
python3 bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline > /tmp/b.log 2>/dev/null; one /tmp/b.log
echo "== config 2 on the fill + tile kernels (GPZ_SVGP_PRODUCTS=tiles; the line above is the panel kernel: fill and both products in one launch, kfill 0)"
GPZ_SVGP_PRODUCTS=tiles python3 bench.py --config 2 --steps 20 --warmup 3 --no-cpu-baseline --no-extra-legs > /tmp/b.log 2>/dev/null; one /tmp/b.log
echo "== N=200k M=256 L=32 Matern-3/2 fp32: the panel kernel (the library's choice for M <= 512); then the tile kernels (GPZ_SVGP_PRODUCTS=tiles)"
python3 bench.py --N 200000 --M 256 --L 32 --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs > /tmp/b.log 2>/dev/null; one /tmp/b.log
GPZ_SVGP_PRODUCTS=tiles python3 bench.py --N 200000 --M 256 --L 32 --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs > /tmp/b.log 2>/dev/null; one /tmp/b.log
echo "== config 5: MGGP 4 groups, N=200k M=2048 fp64, L=32 on one GPU"
python3 bench.py --config 5 --no-cpu-baseline --no-extra-legs > /tmp/b.log 2>/dev/null; one /tmp/b.log
echo "== config 5 as sharded over 8 GPUs: 4 latents per GPU"
python3 bench.py --config 5 --L 4 --no-cpu-baseline --no-extra-legs > /tmp/b.log 2>/dev/null; one /tmp/b.log
echo "== config 4 on one GPU: L=256"
python3 bench.py --L 256 --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > /tmp/b.log 2>/dev/null; one /tmp/b.log
echo "== minibatch training step (N_b=7000, M=3000, L=20, fp32): tools/minibatch_step.py"
python3 tools/minibatch_step.py 2>/dev/null | grep step
echo "== Poisson NSF minibatch step: tools/poisson_step.py (E = 3, then E = 20 in one call)"
python3 tools/poisson_step.py 2>/dev/null | tail -4
python3 tools/poisson_step.py 20 2>/dev/null | tail -1
echo "== VNNGP: tools/vnngp_step.py"
python3 tools/vnngp_step.py 2>/dev/null | grep VNNGP
echo "== batched Cholesky alone, L=32 M=2048 fp64: tools/potrf_only.py"
python3 tools/potrf_only.py 32 2048 2>/dev/null | tail -2
echo "== batched Cholesky alone, L=8 M=512 fp64 (config 2's factor): tools/potrf_only.py"
python3 tools/potrf_only.py 8 512 2>/dev/null | tail -2
echo "== the same two through the launch-per-step chain of rounds 1-3 (GPZ_FACTOR_PATH=launches)"
GPZ_FACTOR_PATH=launches python3 tools/potrf_only.py 32 2048 2>/dev/null | tail -2
GPZ_FACTOR_PATH=launches python3 tools/potrf_only.py 8 512 2>/dev/null | tail -2
echo "== one-launch factorisation, timeline of a traced launch (tools/coop_trace.py: Cholesky alone, then with the inverse)"
python3 tools/coop_trace.py 32 2048 2>/dev/null | grep -v "    D("
python3 tools/coop_trace.py 32 2048 1 2>/dev/null | grep -v "    D("
python3 tools/coop_trace.py 8 512 1 2>/dev/null | grep -v "    D("
python3 tools/coop_trace.py 20 3072 1 2>/dev/null | grep -v "    D("
python3 tools/coop_trace.py 4 2048 1 2>/dev/null | grep -v "    D("
echo "== the same with tiles (j,j-1) and (j,j) as two tasks, as in round 4 (GPZ_COOP_UNFUSED=1): L=8 M=512, L=4 M=2048, L=32 M=2048"
GPZ_COOP_UNFUSED=1 python3 tools/coop_trace.py 8 512 1 2>/dev/null | grep "host-timed\|span"
GPZ_COOP_UNFUSED=1 python3 tools/coop_trace.py 4 2048 1 2>/dev/null | grep "host-timed\|span"
GPZ_COOP_UNFUSED=1 python3 tools/coop_trace.py 32 2048 1 2>/dev/null | grep "host-timed\|span"
echo "== config 3 with the launch-per-step factor path, for comparison"
GPZ_FACTOR_PATH=launches python3 bench.py --no-cpu-baseline --no-extra-legs --steps 5 --warmup 2 > /tmp/b.log 2>/dev/null; one /tmp/b.log
echo "== minibatch step with the launch-per-step factor path"
GPZ_FACTOR_PATH=launches python3 tools/minibatch_step.py 2>/dev/null | grep step
echo "== python bench.py --gpus 2 as typed (two ranks on this one GPU, gloo rendezvous: a rehearsal, not a scaling number)"
GPZ_DIST_BACKEND=gloo python3 bench.py --gpus 2 --N 40000 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
print('  n_gpus %d ranks %d backend %s devices %d | value %.3f %s | %.2f ms/step' % (r['n_gpus'], r['ranks'], r['backend'], r['devices'], r['value'], r['unit'], r['ms_per_step']))"
echo "== all-parameter training step at the minibatch shape (Z, sigma, lengthscale, mu, Lu trainable): tools/svgp_allparam_step.py"
python3 tools/svgp_allparam_step.py 2>/dev/null | grep step
