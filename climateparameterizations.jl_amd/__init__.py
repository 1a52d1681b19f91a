"""colnde — MI355X-native NDE column-model hot path."""
from .config import (NDEConfig, ZeroMeanUnitVarianceScaling, WIND_MIXING, FREE_CONVECTION,
                     CONVECTIVE_ADJUSTMENT_NDE)
from . import flux_compat, synthetic
