"""MI355X-native Neural-SDE integration engine (drop-in for the ``torchsde.sdeint`` hot path of
yongkyung-oh/Stable-Neural-SDEs).  See DESIGN.md / INTEGRATION.md."""
import sys as _sys

from . import _lib, build, controldiffeq, engine, fields, modules, sharding, torchcde, torchsde, train  # noqa: F401
from .modules import (Diffusion_model, IstsNeuralSDE, NeuralSDE, NeuralSDE_forecasting,  # noqa: F401
                      make_sde_model, prepare_sde_solver_kwargs)
from .torchsde import sdeint  # noqa: F401

__version__ = '0.1.0'


def install(force=False):
    """Register this package's ``torchsde`` / ``torchcde`` / ``controldiffeq`` under those module names, so
    the reference's own ``import torchsde`` / ``import torchcde`` / ``import controldiffeq`` (e.g.
    models_sde/neuralsde.py:14-21) resolve to the MI355X engine.  Existing real packages are kept unless
    ``force``."""
    for name, mod in (('torchsde', torchsde), ('torchcde', torchcde), ('controldiffeq', controldiffeq)):
        if force or name not in _sys.modules:
            _sys.modules[name] = mod
    return True
