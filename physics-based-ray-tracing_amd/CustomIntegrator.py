"""Import-compatible alias of the reference module `CustomIntegrator.py`:
    from pbrt_amd.CustomIntegrator import UltraIntegrator        # reference: from CustomIntegrator import UltraIntegrator  (USMain.py:14-24)
The implementation lives in plugins.py."""
from .plugins import UltraIntegrator  # noqa: F401
