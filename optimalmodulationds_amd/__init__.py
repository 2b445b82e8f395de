"""MI355X-native MPPI rollout + DS-modulation hot path (drop-in for the reference's
``python_scripts/ds_mppi/functions`` MPPI / policy / cost / LinDS classes and
``mlp_learn/sdf/robot_sdf.RobotSdfCollisionNet``).

The arithmetic runs in hand-written gfx950 HIP kernels behind the C-ABI declared in
``include/omds.h`` (``optimalmodulationds_amd/csrc/libomds_hip.so``); this package is the thin
Python host side.  Importing the package does not load the library; constructing ``MPPI``
does, and fails loudly when the library or a GPU is missing (there is no CPU fallback).
"""
__all__ = ["MPPI", "TensorPolicyMPPI", "RobotSdfCollisionNet", "LinDS", "SEDS", "Cost", "scenes"]

from . import scenes  # noqa: E402,F401  (pure numpy; safe without the HIP library)


def __getattr__(name):
    if name == "MPPI":
        from .mppi import MPPI
        return MPPI
    if name == "TensorPolicyMPPI":
        from .policy import TensorPolicyMPPI
        return TensorPolicyMPPI
    if name == "RobotSdfCollisionNet":
        from .robot_sdf import RobotSdfCollisionNet
        return RobotSdfCollisionNet
    if name == "LinDS":
        from .lin_ds import LinDS
        return LinDS
    if name == "SEDS":
        from .seds import SEDS
        return SEDS
    if name == "Cost":
        from .cost import Cost
        return Cost
    raise AttributeError(name)
