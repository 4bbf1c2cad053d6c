"""robo-vln_amd: MI355X-native per-step policy forward of robo-vln's HCM agent.

Only the hot path lives here: the HIP kernels + C-ABI
library under `csrc/`, and the Python host-side mirror of the reference's
model-call interface (`policy.py`).  Import through `hcm_pkg.load()`.
"""
from .config import HCMConfig  # noqa: F401
