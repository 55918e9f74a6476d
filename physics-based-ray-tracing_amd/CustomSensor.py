"""Import-compatible alias of the reference module `CustomSensor.py`:
    from pbrt_amd.CustomSensor import UltraSensor     # USMain.py:17 (class survives only as bytecode)
    from pbrt_amd.CustomSensor import CustomSensor    # CustomSensor.py:7 (source class, put_data)
The implementation lives in plugins.py."""
from .plugins import CustomSensor, UltraSensor  # noqa: F401
