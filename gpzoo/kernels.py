"""gpzoo.kernels -> gpzoo_amd.kernels (MI355X HIP implementation behind the reference API)."""
from gpzoo_amd.kernels import *  # noqa: F401,F403
from gpzoo_amd import kernels as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
