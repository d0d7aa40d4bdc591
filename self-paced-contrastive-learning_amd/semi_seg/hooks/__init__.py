from .infonce import INFONCEHook, SelfPacedINFONCEHook, PScheduler  # noqa: F401
from .creator import create_infonce_hooks, create_sp_infonce_hooks, feature_until_from_hooks  # noqa: F401
