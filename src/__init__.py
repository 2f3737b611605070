"""Drop-in ``src`` package (same module names as the reference's ``src/``).

The modules themselves live in ``language-diffusion-model_amd/src/`` next to the HIP
sources they drive; this stub only points the package path there so that
``from src.loss import variational_lower_bound_loss`` etc. resolve from the
repository root exactly as they do in the reference checkout.
"""

import os as _os

_impl = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "language-diffusion-model_amd", "src")
if not _os.path.isdir(_impl):  # pragma: no cover
    raise ImportError(f"implementation package not found at {_impl}")
__path__ = [_impl]
