"""run pytest with the f32 convolutions on the exact-f32 MFMA (spcl_conv_set_f32_split(0)) in this process:
    python tools/diag/pytest_f32_exact.py <pytest args>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import pytest
import tests.conftest  # noqa: F401  (puts the package on the path)
import spcl_amd  # noqa: F401
from spcl_amd import native
native.call("spcl_conv_set_f32_split", 0)
sys.exit(pytest.main(sys.argv[1:]))
