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
