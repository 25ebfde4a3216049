"""Numeric helpers importable under the reference's names (gpzoo/utilities.py).

On the fused HIP path none of these is called: jitter, the whitened KL and the
svgp_forward moments are produced inside ``gpz_svgp_forward``.  They exist so
notebook code that calls them directly on its own tensors keeps working; they
are element-wise / tiny operations expressed with torch on whatever device the
caller's tensors live on.
"""
from __future__ import annotations

import torch


def _torch_sqrt(x, eps=1e-12):
    """sqrt(x + eps): avoids the NaN gradient of sqrt at 0 (reference utilities.py:450-456)."""
    return (x + eps).sqrt()


def _embed_distance_matrix(distance_matrix):
    """Classical-MDS embedding of a (G,G) group-distance matrix (reference
    utilities.py:459-469).  G is a handful of tissue groups: stays in torch, runs once."""
    G = len(distance_matrix)
    centre = torch.eye(G) - torch.full((G, G), 1.0 / G)
    gram = -0.5 * (centre @ (distance_matrix ** 2) @ centre)
    evals, evecs = torch.linalg.eigh(gram)
    evals = evals.clamp(min=0)
    return evecs @ torch.diag(_torch_sqrt(evals, 1e-6))


def _squared_dist(X, Z):
    """Pairwise squared distances, clamped at zero (reference utilities.py:399-405)."""
    r2 = (X ** 2).sum(1, keepdim=True) - 2 * X.matmul(Z.t()) + (Z ** 2).sum(1, keepdim=True).t()
    return r2.clamp(min=0)


def add_jitter(K, jitter=1e-3):
    """IN-PLACE diagonal jitter on (M,M) or (L,M,M); returns the same tensor, and
    None for other ranks like the reference (utilities.py:407-418)."""
    if K.dim() in (2, 3):
        K.diagonal(dim1=-2, dim2=-1).add_(jitter)
        return K
    return None


def reshape_param(param):
    return param.view(-1, param.shape[-2], param.shape[-1])


def whitened_KL(mz, Lz):
    """KL(N(mz, Lz Lz^T) || N(0, I)) for ONE GP: mz (M,), Lz (M,M) -- the reference's
    2-D-only contract (utilities.py:27-36).  Use ``whitened_KL_batched`` for (L,M,M)."""
    M = len(mz)
    return 0.5 * (-2 * torch.log(torch.diagonal(Lz)).sum() + (Lz ** 2).sum() + (mz ** 2).sum() - M)


def whitened_KL_batched(mz, Lz):
    """Per-latent whitened KL for mz (..., M), Lz (..., M, M)."""
    M = mz.shape[-1]
    logdiag = torch.log(torch.diagonal(Lz, dim1=-2, dim2=-1)).sum(-1)
    return 0.5 * (-2 * logdiag + (Lz ** 2).sum((-2, -1)) + (mz ** 2).sum(-1) - M)


def svgp_forward(Kxx, Kzz, W, inducing_mean, inducing_cov):
    """mean = W mu (L,N,1); cov = Kxx + sum((W (S - Kzz)) * W, -1) (L,N) for caller-supplied
    matrices (reference utilities.py:382-397).  SVGP.forward does not call this: the fused HIP
    path evaluates the same moments from Linv without forming W or S."""
    mean = W @ inducing_mean.unsqueeze(-1)
    cov = Kxx + ((W @ (inducing_cov - Kzz)) * W).sum(-1)
    return mean, cov


def _kl_u(qU, pU):
    """sum KL(qU || pU); the whitened closed form when the GP returns pU = None (the reference's loops
    call kl_divergence(qU, pU) and therefore only run un-whitened priors)."""
    from torch import distributions
    if pU is None:
        kl = getattr(qU, "_gpz_kl", None)       # the fused pass's own per-latent KL (differentiable), when present
        return kl.sum() if kl is not None else whitened_KL_batched(qU.mean, qU.scale_tril).sum()
    return torch.sum(distributions.kl_divergence(qU, pU))


def _elbo_terms(model, X, y, E, **kwargs):
    """-ELBO of one step exactly as the reference forms it (utilities.py:476-484):
    ``pY.log_prob(y).mean(axis=0).sum()`` -- axis 0 is the sample axis for the Monte-Carlo likelihoods --
    minus KL(qU || pU)."""
    pY, _, qU, pU = model(X=X, E=E, **kwargs)
    return -(pY.log_prob(y).mean(dim=0).sum() - _kl_u(qU, pU))


def _clamp_loadings(model, names=("W", "W2")):
    """keep raw loadings non-negative after an update (utilities.py:522-523, 551-552, 628)"""
    for n in names:
        w = getattr(model, n, None)
        if isinstance(w, torch.Tensor):
            w.data = torch.clamp(w.data, min=0.0)


def _spots(X, batch_size):
    return torch.multinomial(torch.ones(X.shape[0], device=X.device), num_samples=batch_size, replacement=False)


class GraphedStep:
    """One optimisation step -- ``loss = loss_fn(); loss.backward(); optimizer.step()`` -- captured ONCE as a HIP graph
    and replayed: at the notebooks' sizes (N ~ 1e3, M ~ 1e2) an eager step is a hundred launches of a few microseconds
    each and the host cannot issue them as fast as the GPU retires them; a replay is one host call.

    What a capture needs, and how it is met: no host synchronisation inside the step (the factorisation's ``info`` word
    and the distributions' argument checks are registered as in ``ops.deferred_info()`` and read after the replay:
    ``check()``); static shapes and the same tensors every step (parameters are updated in place by the optimiser, the
    graph re-reads them); an optimiser whose step counter lives on the device (``capturable=True`` is switched on here
    for torch's Adam-family optimisers -- construct the optimiser, then this object, before any eager step of your own);
    no cross-call factor cache (a captured step factors Kzz at every replay, like the reference).  ``warmup`` eager steps
    run first on the capture stream (they are real steps: ``first_losses`` holds their losses)."""

    def __init__(self, loss_fn, optimizer, warmup: int = 3):
        from . import ops
        params = [p for g in optimizer.param_groups for p in g["params"]]
        dev = params[0].device
        for g in optimizer.param_groups:
            if "capturable" in g:
                g["capturable"] = True
        for st in optimizer.state.values():
            if torch.is_tensor(st.get("step")) and st["step"].device != dev:
                st["step"] = st["step"].to(dev)
        self.optimizer, self.pending = optimizer, ops._DeferredInfo()
        self.first_losses = []
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):        # allocator, library workspaces and optimiser state of this stream
                optimizer.zero_grad(set_to_none=True)
                with ops.deferred_info():
                    loss = loss_fn()
                    loss.backward()
                optimizer.step()
                self.first_losses.append(loss.detach().clone())
            optimizer.zero_grad(set_to_none=True)
            self.graph = torch.cuda.CUDAGraph()
            stack = ops._deferred.__dict__.setdefault("stack", [])
            stack.append(self.pending)
            try:
                with torch.cuda.graph(self.graph, stream=side):
                    loss = loss_fn()
                    loss.backward()
                    optimizer.step()
                    self.loss = loss.detach()
                    bad = self.pending.any_bad()
                    self.bad = bad if bad is not None else torch.zeros((), dtype=torch.bool, device=dev)
                    # what one device-to-host copy per step brings back: the loss and "something needs a look"
                    self.status = torch.stack([self.loss.double(), self.bad.double()])
            finally:
                stack.pop()
        cur.wait_stream(side)

    def __call__(self) -> torch.Tensor:
        """Replay the step; returns the (static) loss tensor of the replayed step -- clone it to keep it."""
        self.graph.replay()
        return self.loss

    def check(self) -> float:
        """One sync: the loss of the last replay as a float; raises what the eager step would have raised
        (torch.linalg.LinAlgError for a Kzz that is not positive-definite, torch's ValueError for invalid distribution
        arguments) -- AFTER that step's parameter update, which is the one difference from the eager loop."""
        loss, bad = self.status.tolist()
        if bad:
            self.pending.check(keep=True)
        return loss


def train(model, optimizer, X, y, device=None, steps=200, E=20, fused=True, sync_losses=True, graph=False, **kwargs):
    """Full-batch optimisation loop with the reference's signature (utilities.py:471-493).  The
    forward and the gradients run on the fused HIP path; the optimiser step is torch's.
    ``fused``: Poisson factor models evaluate ``pY.log_prob(y).mean(0).sum()`` through ``model.expected_loglik``
    (gpz_poisson_nsf: the (E,D,N) rate is never materialised), as the mini-batch loops below do.
    Returns the list of losses: floats, one host sync per step as in the reference (``losses.append(loss.item())``,
    utilities.py:487), or with ``sync_losses=False`` 0-d device tensors converted once at the end -- the same numbers
    without stalling the launch queue every step, which is most of a step at the notebooks' small sizes.
    ``graph``: the step is captured once as a HIP graph and replayed (``GraphedStep``: same objective, same updates;
    the optimiser is switched to ``capturable``; errors surface after the failing step's update instead of before)."""
    from .ops import deferred_info
    losses = []
    if graph and steps > 0:
        def loss_fn():
            if fused and hasattr(model, "expected_loglik"):
                ll, _, qU, pU = model.expected_loglik(X, y, E=E, **kwargs)
                return -(ll - _kl_u(qU, pU))
            return _elbo_terms(model, X, y, E, **kwargs)
        warm = min(3, steps)
        step = GraphedStep(loss_fn, optimizer, warmup=warm)
        dev_losses = list(step.first_losses[:steps])
        bad = None
        for _ in range(steps - warm):
            loss = step()
            if sync_losses:
                dev_losses.append(step.check())
            else:
                dev_losses.append(loss.clone())
                bad = step.bad.clone() if bad is None else bad | step.bad
        if bad is not None and bool(bad):
            step.pending.check(keep=True)
        tens = [v for v in dev_losses if torch.is_tensor(v)]
        vals = iter(torch.stack([t.double() for t in tens]).tolist()) if tens else iter(())
        return [next(vals) if torch.is_tensor(v) else v for v in dev_losses]
    for _ in range(steps):
        optimizer.zero_grad()
        with deferred_info():      # Kzz's `info` is read once, behind the backward pass's launches; a failure raises here
            if fused and hasattr(model, "expected_loglik"):
                ll, _, qU, pU = model.expected_loglik(X, y, E=E, **kwargs)     # (the hybrids' 6-tuples fail here as in the reference)
                loss = -(ll - _kl_u(qU, pU))
            else:
                loss = _elbo_terms(model, X, y, E, **kwargs)
            loss.backward()
        optimizer.step()
        losses.append(loss.item() if sync_losses else loss.detach())
    if not sync_losses and losses:
        losses = torch.stack(losses).tolist()
    return losses


def train_batched(model, optimizer, X, y, device=None, steps=200, E=20, batch_size=1000, fused=True, **kwargs):
    """Mini-batch driver of the Poisson factor models (reference utilities.py:600-632): a fresh subset of
    ``batch_size`` spots per step through ``model.forward_batched``, ``pY.log_prob(y[:, idx]).mean(0).sum()``
    minus KL(qU || pU), ``model.W`` clamped at zero after the update.  ``fused`` evaluates the same
    expected log-likelihood through ``model.expected_loglik`` (gpz_poisson_nsf: the (E,D,N_b) rate is never
    materialised).  The index draw stays on the device (the reference samples on the host every step).
    Models without ``forward_batched`` (plain GP likelihoods) get ``model(X[idx])`` on the sampled spots."""
    from .ops import deferred_info
    losses = []
    for _ in range(steps):
        idx = _spots(X, min(batch_size, X.shape[0]))
        optimizer.zero_grad()
        with deferred_info():
            if not hasattr(model, "forward_batched"):
                kw = dict(kwargs)
                if "groupsX" in kw:
                    kw["groupsX"] = kw["groupsX"][idx]
                loss = _elbo_terms(model, X[idx], y[..., idx], E, **kw)
            else:
                if fused and hasattr(model, "expected_loglik"):
                    ll, _, qU, pU = model.expected_loglik(X, y[:, idx], idx=idx, E=E, **kwargs)
                else:
                    pY, _, qU, pU = model.forward_batched(X=X, idx=idx, E=E, **kwargs)
                    ll = pY.log_prob(y[:, idx]).mean(dim=0).sum()
                loss = -(ll - _kl_u(qU, pU))
            loss.backward()
        optimizer.step()
        _clamp_loadings(model, ("W",))
        losses.append(loss.item())
    return losses


def train_hybrid(model, optimizer, X, y, device=None, steps=200, E=20, fused=True, **kwargs):
    """Full-batch driver of the hybrid (spatial + non-spatial) models, reference utilities.py:530-558."""
    from torch import distributions
    from .ops import deferred_info
    losses = []
    for _ in range(steps):
        optimizer.zero_grad()
        with deferred_info():
            if fused and hasattr(model, "expected_loglik"):
                ll, _, qU, pU, qF, pF = model.expected_loglik(X, y, E=E, **kwargs)
            else:
                pY, _, qU, pU, qF, pF = model(X=X, E=E, **kwargs)
                ll = pY.log_prob(y).mean(dim=0).sum()
            loss = -(ll - _kl_u(qU, pU) - torch.sum(distributions.kl_divergence(qF, pF)))
            loss.backward()
        optimizer.step()
        _clamp_loadings(model)
        losses.append(loss.item())
    return losses


def train_hybrid_batched(model, optimizer, X, y, device=None, steps=200, E=20, batch_size=1000, fused=True, **kwargs):
    """Mini-batch driver of the hybrid models, reference utilities.py:497-527: the log-likelihood is
    ``y log(rate) - rate`` (no log y! term), both KL terms are subtracted, W / W2 are clamped at zero."""
    from torch import distributions
    from .ops import deferred_info
    losses = []
    for _ in range(steps):
        idx = _spots(X, batch_size)
        optimizer.zero_grad()
        with deferred_info():
            if fused and hasattr(model, "expected_loglik"):
                ll, _, qU, pU, qF, pF = model.expected_loglik(X, y[:, idx], idx=idx, E=E, with_lgamma=False, **kwargs)
            else:
                pY, _, qU, pU, qF, pF = model.forward_batched(X=X, idx=idx, E=E, **kwargs)
                ll = (y[:, idx] * torch.log(pY.rate) - pY.rate).mean(dim=0).sum()
            loss = -(ll - _kl_u(qU, pU) - torch.sum(distributions.kl_divergence(qF, pF)))
            loss.backward()
        optimizer.step()
        _clamp_loadings(model)
        losses.append(loss.item())
    return losses


def train_closure_batched(model, optimizer, X, groupsX, y, device=None, steps=200, E=20, batch_size=1000):
    """Closure-style mini-batch driver for optimisers that re-evaluate the loss (LBFGS), multi-group models;
    reference utilities.py:561-597."""
    losses = []

    from .ops import deferred_info

    def closure(idx):
        optimizer.zero_grad()
        with deferred_info():
            pY, _, qU, pU = model.forward_batched(X, groupsX, idx, E=E)
            loss = -(pY.log_prob(y[:, idx]).mean(dim=0).sum() - _kl_u(qU, pU))
            loss.backward()
        losses.append(loss.item())
        return loss

    for _ in range(steps):
        idx = _spots(X, batch_size)
        optimizer.step(lambda: closure(idx))
    return losses


# Host-side data preparation of the reference's utilities module (AnnData conversion, scanpy size factors,
# sklearn NMF initialisation, plotting, ...) is outside the accelerated path and not rebuilt here.  The
# names resolve so that ``from gpzoo.utilities import train_hybrid, anndata_to_train_val`` -- the notebooks'
# import lines -- keep working; calling one says where it lives.
_NOT_REBUILT = ("build_group_distances", "init_softplus", "smooth_spatial_factors", "rescale_spatial_coords",
                "anndata_to_train_val", "scanpy_sizefactors", "dims_autocorr", "lnormal_approx_dirichlet",
                "regularized_nmf", "shrink_factors", "shrink_loadings", "plot_factors")


def __getattr__(name):
    if name in _NOT_REBUILT:
        def _missing(*args, **kwargs):
            raise NotImplementedError(
                f"gpzoo.utilities.{name} is host-side data preparation outside the MI355X hot path and is not "
                f"rebuilt in gpzoo_amd; use the reference package's function of the same name for it")
        _missing.__name__ = name
        return _missing
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
