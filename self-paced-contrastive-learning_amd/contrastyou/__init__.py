"""Mirror package of the reference's ``contrastyou`` (hot-path modules only)."""
import os


def success(save_dir: str):
    """``contrastyou.success`` (contrastyou/__init__.py:37-49): mark a finished run with an empty ``.success`` file"""
    os.makedirs(save_dir, exist_ok=True)
    with open(os.path.join(save_dir, ".success"), "w"):
        pass
