"""Import-compatible alias of the reference module `CustomEmmitter.py`:
    from pbrt_amd.CustomEmmitter import CustomEmitter        # reference: from CustomEmmitter import CustomEmitter  (USMain.py:14-24)
The implementation lives in plugins.py."""
from .plugins import CustomEmitter  # noqa: F401
