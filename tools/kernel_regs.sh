#!/bin/bash
# register / spill / LDS figures of the kernels matching a pattern, from a scratch compile (no GPU needed):
#   tools/kernel_regs.sh <pattern> [extra hipcc flags]     -> _ab/regs/ holds the .s for reading
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/_ab/regs; mkdir -p $OUT; cd $OUT
PAT=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC "$@" -c --save-temps -o x.o $ROOT/physics-based-ray-tracing_amd/csrc/pbrt_api.hip 2>&1 | grep -E "error" -A5
python3 - "$PAT" <<'PY'
import re, sys
txt = open("pbrt_api-hip-amdgcn-amd-amdhsa-gfx950.s").read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if not re.search(sys.argv[1], name): continue
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    print(f"{name:60s} vgpr {g('vgpr_count'):3d} spill {g('vgpr_spill_count'):3d}  sgpr {g('sgpr_count'):3d} spill {g('sgpr_spill_count'):3d}  scratch {g('private_segment_fixed_size'):4d} B  lds {g('group_segment_fixed_size'):6d} B")
PY
