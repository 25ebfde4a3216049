#!/usr/bin/env python3
"""The forward pass as a captured HIP graph: eager launches vs graph replay at the notebooks' small sizes.

    python3 tools/graph_replay.py

gpz_svgp_forward is a pure sequence of asynchronous launches on the caller's stream (no host synchronisation, no
allocation, no library-owned stream), so torch.cuda.graph() can capture it; check_info=False leaves the factorisation's
`info` on the device (read it after the replay).  Replays see in-place updates of the captured input tensors.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import ops  # noqa: E402
from gpzoo_amd.configs import spec_for_config  # noqa: E402
from gpzoo_amd.synthetic import make_config  # noqa: E402

dev = torch.device("cuda", 0)


def one(cfg, **kw):
    c = make_config(cfg, **kw)
    g = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in c.items()}
    spec, extra = spec_for_config(g, dev)

    def fwd():
        return ops.svgp_forward(spec, g["X"], g["Z"], g["mu"], g["Lu_raw"], c["jitter"], c["whitened"], y=g["y"],
                                noise_sd=c["noise_sd"], want_Lu=False, check_info=False, **extra)

    for _ in range(3):
        ref = fwd()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        fwd()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / n * 1e3
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fwd()                                   # warm-up on the capture stream (workspace of that stream's pool)
        with torch.cuda.graph(graph, stream=side):
            out = fwd()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    same = abs(float(out["elbo"]) - float(ref["elbo"])) <= 1e-12 * abs(float(ref["elbo"]))
    t0 = time.perf_counter()
    for _ in range(n):
        graph.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / n * 1e3
    # replays follow in-place edits of the captured inputs
    g["mu"].mul_(0.5)
    graph.replay()
    torch.cuda.synchronize()
    e_graph = float(out["elbo"])
    e_eager = float(fwd()["elbo"])
    follows = abs(e_graph - e_eager) <= 1e-12 * abs(e_eager)
    N, M = c["X"].shape[0], c["Z"].shape[0]
    print("config %d N=%6d M=%4d L=%2d | eager %.3f ms | graph replay %.3f ms | same ELBO %s | follows input edits %s" % (
        cfg, N, M, c["mu"].shape[0] if c["mu"].dim() > 1 else 1, eager, rep, same, follows), flush=True)


one(1)
one(2, N=2000, M=300, L=8)
one(2, N=10000, M=500, L=2)
one(2)
one(3, N=7000, M=1024, L=8)
