"""gpzoo.utilities -> gpzoo_amd.utilities (MI355X HIP implementation behind the reference API)."""
from gpzoo_amd.utilities import *  # noqa: F401,F403
from gpzoo_amd import utilities as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})


def __getattr__(name):   # names that resolve lazily in the implementation module (see its _NOT_REBUILT)
    return getattr(_impl, name)
