import os, sys, torch, torch.nn as nn
sys.path.insert(0, os.getcwd())
from gpzoo.gp import SVGP, WSVGP
from gpzoo.kernels import NSF_RBF
from gpzoo.likelihoods import GaussianLikelihood
from gpzoo.utilities import train_batched
torch.manual_seed(0)
N, M, L = 20000, 600, 6
X = (torch.rand(N, 2) * 100).cuda(); y = torch.randn(L, N).cuda()
for cls in (WSVGP, SVGP):
    gp = cls(NSF_RBF(sigma=1.0, lengthscale=5.0, L=L), dim=2, M=M, jitter=1e-2)
    gp.Z = nn.Parameter(X[torch.randperm(N)[:M]].clone().cpu()); gp.mu = nn.Parameter(torch.zeros(L, M)); gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
    model = GaussianLikelihood(gp, noise=0.5).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    mem = []
    for rep in range(4):
        losses = train_batched(model, opt, X, y, torch.device("cuda"), steps=50, E=2, batch_size=3000)
        torch.cuda.synchronize()
        mem.append((torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20))
        assert all(l == l for l in losses), "NaN loss"
    print(cls.__name__, "loss", round(losses[0], 1), "->", round(losses[-1], 1), "MiB allocated/reserved after each 50 steps:", mem)
    assert mem[-1][0] <= mem[1][0] + 1

# ---- Poisson factor models and VNNGP
from torch import distributions
from gpzoo.gp import VNNGP
from gpzoo.likelihoods import NSF2, Hybrid_NSF
from gpzoo.utilities import train_hybrid_batched
D = 300
yc = torch.poisson(2.0 * torch.rand(D, N)).cuda()
for name in ("NSF2", "Hybrid_NSF", "VNNGP"):
    torch.manual_seed(1)
    if name == "VNNGP":
        gp = VNNGP(NSF_RBF(sigma=1.0, lengthscale=5.0, L=L), dim=2, M=M, K=6, jitter=1e-2)
    else:
        gp = SVGP(NSF_RBF(sigma=1.0, lengthscale=5.0, L=L), dim=2, M=M, jitter=1e-2)
    gp.Z = nn.Parameter(X[torch.randperm(N)[:M]].clone().cpu()); gp.mu = nn.Parameter(torch.zeros(L, M)); gp.Lu = nn.Parameter(0.01 * torch.randn(L, M, M))
    mem = []
    if name == "VNNGP":
        gp = gp.cuda(); opt = torch.optim.Adam(gp.parameters(), lr=1e-2)
        for rep in range(4):
            for _ in range(25):
                opt.zero_grad()
                qF, qU, pU = gp(X[:8000])
                loss = -(distributions.Normal(qF.mean, 0.5).log_prob(y[:, :8000]).sum() - distributions.kl_divergence(qU, pU).sum())
                loss.backward(); opt.step()
            torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated() >> 20)
        last = float(loss.detach())
    else:
        model = (NSF2(gp, yc, L=L) if name == "NSF2" else Hybrid_NSF(gp, yc, L=L, non_spatial_factors=2)).cuda()
        opt = torch.optim.Adam(model.parameters(), lr=1e-2)
        loop = train_batched if name == "NSF2" else train_hybrid_batched
        for rep in range(4):
            losses = loop(model, opt, X, yc, torch.device("cuda"), steps=25, E=2, batch_size=3000)
            torch.cuda.synchronize(); mem.append(torch.cuda.memory_allocated() >> 20)
        last = losses[-1]
    print(name, "last loss", round(last, 1), "MiB allocated after each 25 steps:", mem)
    assert last == last and mem[-1] <= mem[1] + 1
