"""Seeded synthetic inputs of the golden cases.  No reference import: the generator script (make_golden.py, build
container only) and the tests (which regenerate the inputs of the larger fixtures instead of storing them) share it."""
import torch


def gen(seed, *shape, dist="randn"):
    g = torch.Generator().manual_seed(seed)
    f = torch.randn if dist == "randn" else torch.rand
    return f(*shape, generator=g, dtype=torch.float64)


def make_inputs(seed, N, M, d, L, n_groups=0, span=10.0):
    X = (gen(seed, N, d, dist="rand") - 0.5) * 2 * span
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(seed + 1))
    Z = X[perm[:M]].clone() + 0.05 * gen(seed + 2, M, d)
    shape_mu = (L, M) if L else (M,)
    shape_Lu = (L, M, M) if L else (M, M)
    mu = 0.5 * gen(seed + 3, *shape_mu)
    Lu = 0.05 * gen(seed + 4, *shape_Lu)
    Lu = Lu + torch.diag_embed(-0.3 + 0.1 * gen(seed + 5, *shape_mu)) - torch.diag_embed(torch.diagonal(Lu, dim1=-2, dim2=-1))
    shape_y = (L, N) if L else (N,)
    y = torch.sin(X[:, 0] / 3.0).expand(shape_y) + 0.1 * gen(seed + 6, *shape_y)
    out = dict(X=X, Z=Z, mu=mu, Lu_raw=Lu, y=y)
    if n_groups:
        g = torch.Generator().manual_seed(seed + 7)
        out["gX"] = torch.randint(0, n_groups, (N,), generator=g)
        out["gZ"] = torch.randint(0, n_groups, (M,), generator=g)
    return out
