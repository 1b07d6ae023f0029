"""Selects the bench library for the variant sweeps unless TFX_LIB already names one (must happen before textflux_amd._lib is imported)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
os.environ.setdefault("TFX_LIB", os.path.join(REPO, "textflux_amd", "libtextflux_hip_bench.so"))

from tests.conftest import pytest_collection_modifyitems, pytest_configure   # noqa: E402,F401
