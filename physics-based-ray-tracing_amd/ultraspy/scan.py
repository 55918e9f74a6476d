"""USMain.py:9"""
from ..beamform import GridScan  # noqa: F401
