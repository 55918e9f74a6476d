"""Import-compatible alias of the reference module `CustomBSDF.py`:
    from pbrt_amd.CustomBSDF import UltraBSDF        # reference: from CustomBSDF import UltraBSDF  (USMain.py:14-24)
The implementation lives in plugins.py."""
from .plugins import UltraBSDF  # noqa: F401
