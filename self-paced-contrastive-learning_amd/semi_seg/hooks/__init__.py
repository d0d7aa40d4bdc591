from .infonce import INFONCEHook, SelfPacedINFONCEHook, PScheduler, get_n_point_coordinate  # noqa: F401
from .creator import create_infonce_hooks, create_sp_infonce_hooks, feature_until_from_hooks  # noqa: F401


def create_discrete_mi_consistency_hook(*args, **kwargs):
    """named by the reference's ``hook_creator.py:1`` import line; a comparison baseline outside the HIP hot path"""
    raise NotImplementedError("create_discrete_mi_consistency_hook: comparison baseline outside the HIP hot path")
