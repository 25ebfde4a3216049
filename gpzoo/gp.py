"""gpzoo.gp -> gpzoo_amd.gp (MI355X HIP implementation behind the reference API)."""
from gpzoo_amd.gp import *  # noqa: F401,F403
from gpzoo_amd import gp as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
