"""A/B of prebuilt library variants on one box (run on the GPU box): python tools/ab_variants.py name=lib.so[,ENV=VAL,...] ...
Each variant renders the ring scene (1024^2 x 64 spp) in a child process; prints kernel ms and a hash of the film."""
import hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import hashlib, os, sys
sys.path.insert(0, %r)
import pbrt_amd as mi
scene, res, spp = os.environ.get("AB_SCENE", "tests/scenes/testring.xml"), int(os.environ.get("AB_RES", "1024")), int(os.environ.get("AB_SPP", "64"))
if os.environ.get("AB_MODE") == "us":   # BASELINE config 3: the Sphere_Box phantom, 5 x 64 rays x 838 912 paths
    import numpy as np
    kw = dict(kv.split("=") for kv in os.environ.get("AB_US_KW", "").split(";") if kv)
    us = mi.load_file(os.path.join(%r, os.environ.get("AB_US_SCENE", "tests/scenes/us_sphere_box.xml")), **kw)
    ui = us.integrator()
    best = 1e9
    for i in range(int(os.environ.get("AB_REPS", "3"))):
        buf = ui._acquire(us, ui.quirks, paths_per_ray=int(os.environ.get("AB_PPR", "838912")), seed=0); st = mi.default_context().stats(); best = min(best, st["kernel_ms"])
    print(f"{best:8.2f} ms  {st['samples']/best/1e3:7.0f} Msamples/s  |buf| {float(np.abs(buf).sum()):.9g} nonzero {int((buf != 0).sum())}", flush=True)
    sys.exit(0)
sc = mi.load_file(os.path.join(%r, scene), res=res, spp=spp)
best = 1e9
for i in range(int(os.environ.get("AB_REPS", "3"))):
    img = mi.render(sc, seed=0); st = mi.default_context().stats(); best = min(best, st["kernel_ms"])
print(f"{best:8.2f} ms  {res*res*spp/best/1e3:7.0f} Msamples/s  film {hashlib.sha256(img.tobytes()).hexdigest()[:12]}", flush=True)
''' % (ROOT, ROOT, ROOT)
for rnd in range(int(os.environ.get("AB_ROUNDS", "1"))):
    for spec in sys.argv[1:]:
        name, rest = spec.split("=", 1)
        parts = rest.split(",")
        env = dict(os.environ, PBRT_HIP_LIB=os.path.join(ROOT, parts[0]))
        for kv in parts[1:]:
            k, v = kv.split("=", 1)
            env[k] = v
        try:
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=120)
            line = [l for l in out.stdout.splitlines() if "Msamples" in l]
            print(f"{name:28s} {line[-1] if line else 'FAILED: ' + (out.stderr.strip().splitlines() or ['?'])[-1][:200]}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"{name:28s} TIMEOUT", flush=True)
            sys.exit(1)  # a hung kernel: no further GPU step in this call
