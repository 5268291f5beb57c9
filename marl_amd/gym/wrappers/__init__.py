from ..core import TimeLimit
from .assembly_wrapper import AssemblySwarmWrapper

__all__ = ["TimeLimit", "AssemblySwarmWrapper"]
