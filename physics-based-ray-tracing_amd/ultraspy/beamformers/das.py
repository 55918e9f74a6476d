"""USMain.py:8"""
from ...beamform import DelayAndSum  # noqa: F401
