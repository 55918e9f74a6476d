"""Build-container check of the compiled device code (no GPU): the traversal stack of the BVH kernels lives in two address
spaces -- LDS rows and a private overflow array (csrc/device_scene.h BvhStack / BvhOvf) -- and the two must never be reached
through ONE instruction.  A FLAT access picks its aperture per lane at run time, and FLAT accesses to LDS are not ordered
against the ds_* instructions around them: round 3's first BVH4 stack cycled on the device because the compiler had merged an
LDS push and a scratch push into one flat_store through a selected pointer.  The pointers now carry their address space in the
type (LDS_AS / PRIV_AS), so the rows can only become ds_* and the overflow only scratch_*; this test compiles both libraries to
assembly and holds the compiler to it: no flat_* memory instruction in any kernel, ds_* and scratch_* both present where the
stack is."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "physics-based-ray-tracing_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-S", "--cuda-device-only"]
FLAT = re.compile(r"^\s+flat_(load|store|atomic)\w*\s")
KERNEL = re.compile(r"^(_Z\w+):")


def kernels_of(asm_text):
    """{mangled name: [instruction lines]} of every kernel (.amdhsa_kernel entries) of an assembly file"""
    names = set(re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", asm_text, flags=re.M))
    out, cur = {}, None
    for line in asm_text.splitlines():
        m = KERNEL.match(line)
        if m:
            cur = m.group(1) if m.group(1) in names else None
            if cur:
                out[cur] = []
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        if line.startswith("\t") and not line.lstrip().startswith((".", ";")):
            out[cur].append(line)
    return out


@pytest.fixture(scope="module", params=["product", "diag"])
def asm(request, tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not on PATH")
    out = str(tmp_path_factory.mktemp("asm") / f"pbrt_{request.param}.s")
    extra = ["-DPBRT_DIAG"] if request.param == "diag" else []
    subprocess.run(["hipcc", *FLAGS, *extra, "-o", out, "pbrt_api.hip"], cwd=CSRC, check=True, capture_output=True, timeout=600)
    return request.param, kernels_of(open(out).read())


def test_no_flat_memory_instruction_in_any_kernel(asm):
    build, kernels = asm
    assert len(kernels) > 40, (build, len(kernels))
    bad = {k: [l.strip() for l in lines if FLAT.match(l)] for k, lines in kernels.items()}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, f"{build}: FLAT memory instructions in {sorted(bad)[:5]}: {list(bad.values())[0][:3]}"


def test_the_stack_of_the_bvh_kernels_is_ds_rows_plus_scratch_overflow(asm):
    build, kernels = asm
    # k_trace<FIRST, ACCEL, CURVED> (ACCEL 1 = tree in global memory, 2 = in LDS), the ultrasound bounce on BVH scenes and the
    # closest-hit / any-hit leaf operators on a tree in global memory
    walkers = [k for k in kernels if re.match(r"_Z7k_traceILb[01]ELi[12]ELb[01]E", k) or re.match(r"_Z11k_us_bounceILb[01]ELi[12]E", k)
               or re.match(r"_Z(15k_ray_intersect|10k_ray_test)ILi1E", k)]
    # (the fused ultrasound bounce on BVH scenes lives in the diagnostic build only: the product runs k_trace + k_us_shade)
    assert len(walkers) >= 8 + 2 + (4 if build == "diag" else 0), walkers
    for k in walkers:
        text = "\n".join(kernels[k])
        assert re.search(r"^\s+ds_write_b32\s", text, flags=re.M) and re.search(r"^\s+ds_read_b32\s", text, flags=re.M), k
        assert re.search(r"^\s+scratch_store_dword\s", text, flags=re.M) and re.search(r"^\s+scratch_load_dword\s", text, flags=re.M), k
    # the packet walk keeps its stack in the lanes of one register: no scratch at all
    for k in [k for k in kernels if k.startswith("_Z15k_trace_primary")]:
        assert not re.search(r"^\s+scratch_", "\n".join(kernels[k]), flags=re.M), k
