"""see pbrt_amd.ultraspy"""
