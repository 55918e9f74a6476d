"""bench.py's command line, the parts that need no GPU."""
import os
import subprocess
import sys

from conftest import ROOT


def test_more_ranks_than_gpus_is_refused_before_anything_is_spawned():
    """`--gpus N` on a node with fewer GPUs must not start N ranks that fight over the devices there are: non-zero exit and a
    message that names the rehearsal switch (the build container shows no GPU at all)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 64" in r.stderr and "--rehearse-on-one-gpu" in r.stderr and not r.stdout.strip()


def test_the_launcher_parent_never_loads_a_gpu_runtime():
    """launch_ranks counts the GPUs in sysfs (or from HIP_VISIBLE_DEVICES): neither torch nor a HIP library may be mapped
    into the parent, whose children are forked from it."""
    code = (
        "import sys, types; sys.path.insert(0, %r); import bench\n"
        "args = types.SimpleNamespace(gpus=64, rehearse_on_one_gpu=False)\n"
        "rc = bench.launch_ranks(args, ['--gpus', '64'])\n"
        "maps = open('/proc/self/maps').read()\n"
        "assert rc == 2, rc\n"
        "assert 'torch' not in sys.modules and 'libamdhip64' not in maps and 'libhsa-runtime' not in maps\n"
        "n = bench.visible_gpu_count(); assert isinstance(n, int) and 0 <= n <= 64\n"
        "print('ok', n)\n") % ROOT
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "LD_PRELOAD"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stderr[-2000:]
    env["HIP_VISIBLE_DEVICES"] = ""          # an empty list hides every GPU
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok 0", r.stderr[-2000:]


def test_tool_scripts_compile():
    """the measurement helpers under tools/ are run by hand on the GPU box: at least keep them parseable"""
    import glob
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(glob.glob(os.path.join(root, "tools", "*.py")))
    assert len(scripts) >= 20
    for f in scripts:
        compile(open(f).read(), f, "exec")      # (syntax only: nothing is imported, nothing is written)
