"""adaptive-proximal-algorithms_amd — MI355X-native adaptive proximal-gradient hot path.

Python host mirror of AdaProx.jl's solver entry points over the C-ABI in
``include/adaprox_hip.h`` (hand-written gfx950 HIP kernels in ``csrc/``).  The directory
name is not a Python identifier; import it with

    import importlib; apx = importlib.import_module("adaptive-proximal-algorithms_amd")

or through the alias module ``adaprox_amd`` at the repository root.
"""
from ._lib import ApxError, ApxInvalid, DomainError, LIB_PATH, load  # noqa: F401
from .api import (  # noqa: F401
    Context, DeviceMatrix, default_context, set_default_context, as_device_matrix,
    Counting, without_counting, is_counting_enabled,
    Zero, IndZero, NormL1, NormL2, IndBox, Translate, ConvexConjugate, convex_conjugate, prox,
    LinearLeastSquares, LogisticLoss, Quadratic, eval_with_pullback, eval_with_gradient,
    FixedStepsize, MalitskyMishchenkoRule, OurRule, OurRulePlus,
    adaptive_primal_dual, adaptive_proxgrad, fixed_proxgrad, condat_vu, backtracking_proxgrad,
    adaptive_linesearch_primal_dual,
    Solver, last_incident,
)

from .baselines import backtracking_nesterov, fixed_nesterov, agraal, malitsky_pock  # noqa: F401,E402

Record = -1  # `const Record = Logging.LogLevel(-1)` (src/AdaProx.jl:7): records go to the `record=` callable
