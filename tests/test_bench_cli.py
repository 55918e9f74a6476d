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
