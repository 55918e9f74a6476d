"""USMain.py:10"""
from ...beamform import Probe, build_probe  # noqa: F401
