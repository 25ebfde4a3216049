"""Seeded synthetic inputs for BASELINE.json's five configurations.

Shapes and hyper-parameters follow SURVEY.md §8(d).  Everything is drawn on the
CPU with explicit ``torch.Generator`` seeds so the build container, the GPU box
and every rank of a multi-GPU job see identical numbers; per-latent quantities
are seeded per latent so a rank can draw just its own shard.

Returned dict (CPU tensors, float64 unless ``dtype`` says otherwise):
  kind, whitened, dtype, X (N,d), y (L,N) or (N,), Z (M,d), sigma, lengthscale,
  mu, Lu_raw, jitter, noise_sd, and for config 5 gX, gZ, group_diff, n_groups.
"""
from __future__ import annotations

import math

import torch

CONFIGS = {
    1: dict(N=1000, M=64, L=0, d=1, kind="rbf", whitened=False, dtype=torch.float64, jitter=1e-3),
    2: dict(N=50_000, M=512, L=8, d=2, kind="nsf_rbf", whitened=True, dtype=torch.float32, jitter=1e-1),
    3: dict(N=200_000, M=2048, L=32, d=2, kind="matern32", whitened=True, dtype=torch.float32, jitter=1e-1),
    4: dict(N=200_000, M=2048, L=256, d=2, kind="matern32", whitened=True, dtype=torch.float32, jitter=1e-1),
    5: dict(N=200_000, M=2048, L=32, d=2, kind="mggp_nsf_rbf", whitened=True, dtype=torch.float64, jitter=1e-2,
            n_groups=4),
}


def _gen(seed: int) -> torch.Generator:
    return torch.Generator().manual_seed(int(seed))


def _softplus(v: float) -> float:
    return math.log1p(math.exp(v))


def make_config(cfg: int, *, N: int | None = None, M: int | None = None, L: int | None = None,
                latents: range | None = None, dtype: torch.dtype | None = None) -> dict:
    """Draw configuration ``cfg`` (1..5), optionally scaled down via N/M/L.

    ``latents`` restricts the per-latent arrays (sigma, lengthscale, mu, Lu_raw,
    y, group_diff) to a contiguous latent range -- the block a rank owns.
    """
    c = dict(CONFIGS[cfg])
    N = c["N"] if N is None else N
    M = c["M"] if M is None else M
    L = c["L"] if L is None else L
    dt = c["dtype"] if dtype is None else dtype
    out = dict(cfg=cfg, kind=c["kind"], whitened=c["whitened"], dtype=dt, jitter=c["jitter"], N=N, M=M, L=L)

    if cfg == 1:
        X = (torch.rand(N, 1, generator=_gen(1001), dtype=torch.float64) - 0.5) * 10.0
        y = 2.0 * torch.sin(2.0 * X[:, 0]) + 0.1 * torch.randn(N, generator=_gen(4001), dtype=torch.float64)
        Z = torch.linspace(-5.0, 5.0, M, dtype=torch.float64)[:, None]
        g = _gen(3001)
        mu = 0.3 * torch.randn(M, generator=g, dtype=torch.float64)
        Lu = 0.1 * torch.randn(M, M, generator=g, dtype=torch.float64)
        out.update(X=X, y=y, Z=Z, mu=mu, Lu_raw=Lu, sigma=torch.tensor(1.0, dtype=torch.float64),
                   lengthscale=torch.tensor(1.0, dtype=torch.float64), noise_sd=_softplus(0.1))
        return _cast(out, dt)

    X = (torch.rand(N, 2, generator=_gen(1000 + cfg), dtype=torch.float64) - 0.5) * 200.0
    if cfg == 5:
        G = c["n_groups"]
        per = N // G
        gX = torch.arange(G).repeat_interleave(per)
        gX = torch.cat([gX, torch.full((N - per * G,), G - 1)])
        mper = M // G
        rows = []
        for g in range(G):
            idx = torch.nonzero(gX == g)[:, 0]
            pick = torch.randperm(idx.numel(), generator=_gen(2000 + cfg + 17 * g))[:mper if g < G - 1 else M - mper * (G - 1)]
            rows.append(idx[pick])
        zrows = torch.cat(rows)
        out.update(gX=gX, gZ=gX[zrows].clone(), n_groups=G)
    else:
        zrows = torch.randperm(N, generator=_gen(2000 + cfg))[:M]
    Z = X[zrows].clone()

    lat = range(L) if latents is None else latents
    if cfg == 2:
        ell_all = torch.linspace(3.0, 12.0, L, dtype=torch.float64)
    elif cfg == 5:
        ell_all = torch.full((L,), 8.0, dtype=torch.float64)
    else:
        ell_all = torch.linspace(5.0, 20.0, L, dtype=torch.float64)
    ell = ell_all[lat.start:lat.stop]
    Ll = len(lat)
    mu = torch.empty(Ll, M, dtype=torch.float64)
    Lu = torch.empty(Ll, M, M, dtype=torch.float32 if dt == torch.float32 else torch.float64)
    y = torch.empty(Ll, N, dtype=torch.float64)
    for i, l in enumerate(lat):
        g = _gen((3000 + cfg) * 100_003 + l)
        mu[i] = torch.randn(M, generator=g, dtype=torch.float64)
        blk = 0.01 * torch.randn(M, M, generator=g, dtype=Lu.dtype)
        Lu[i] = blk.tril(-1)
        Lu[i].diagonal().fill_(1.0)
        gy = _gen((4000 + cfg) * 100_003 + l)
        y[i] = (torch.sin(X[:, 0] / ell[i]) + torch.cos(X[:, 1] / ell[i])
                + 0.1 * torch.randn(N, generator=gy, dtype=torch.float64))
    out.update(X=X, y=y, Z=Z, mu=mu, Lu_raw=Lu, sigma=torch.ones(Ll, dtype=torch.float64),
               lengthscale=ell, noise_sd=0.5, latents=lat)
    if cfg == 5:
        out["group_diff"] = torch.full((Ll,), 0.7, dtype=torch.float64)
    return _cast(out, dt)


def _cast(out: dict, dt: torch.dtype) -> dict:
    for k, v in out.items():
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            out[k] = v.to(dt)
    return out


def shard_latents(L: int, world: int, rank: int) -> range:
    """Contiguous block of latents owned by ``rank``: the first
    ``L % world`` ranks get one extra latent."""
    base, rem = divmod(L, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))
