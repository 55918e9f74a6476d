"""Import-compatible stand-in for the three `ultraspy` entry points the reference driver uses (USMain.py:8-10):
    from pbrt_amd.ultraspy.beamformers.das import DelayAndSum
    from pbrt_amd.ultraspy.scan import GridScan
    from pbrt_amd.ultraspy.probes.factory import build_probe
`ultraspy` itself is third-party and absent; the implementation (this build's own definition, GPU only) lives in
beamform.py."""
