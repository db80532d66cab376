"""olympic_hip: MI355X-native hot path of the olympics-mujoco locomotion environments.

Nothing here imports the CPU oracle; the HIP library is loaded lazily by olympic_hip._ffi and
its absence is an error, not a fallback."""
from . import _abi, specs  # noqa: F401

__all__ = ["_abi", "specs"]
