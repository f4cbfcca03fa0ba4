"""deqsci_amd - MI355X-native DEQ-SCI reconstruction hot path.

The HIP library (lib/libdeqsci_hip.so, C ABI in include/deqsci_hip.h) is loaded lazily by the
first operator call; importing the package never needs a GPU."""
from .operators import A_torch_, At_torch_, initial_point, phi_sum, LinearOperator, SCIOperator  # noqa: F401
from .solvers import (EquilibriumProxGradSCI, andersonexp, forward_iteration, DEQFixedPoint,  # noqa: F401
                      EquilibriumADMMSCI, admmexp, DEQFixedPointADMM, initial_point_admm)
from .engine import DEQSCIEngine, sigma_schedule  # noqa: F401
from .networks import FFDNet, DnCNN  # noqa: F401

__version__ = "0.1.0"
