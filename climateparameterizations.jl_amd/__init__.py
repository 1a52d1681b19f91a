"""colnde — MI355X-native NDE column-model hot path.

`import colnde` (alias module at the repo root) loads this package.  Host code is pure Python over the
C ABI of libcolnde.so; the HIP extension is mandatory (no CPU fallback)."""
from .config import (NDEConfig, ZeroMeanUnitVarianceScaling, WIND_MIXING, FREE_CONVECTION,
                     CONVECTIVE_ADJUSTMENT_NDE)
from . import flux_compat, synthetic
from ._lib import ColndeError, build as build_extension, LIB_PATH
from .nde import ColumnNDE, min_substeps, rkc_stages
from . import distributed, wind_mixing, free_convection
