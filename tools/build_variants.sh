#!/bin/bash
# build library variants here (no GPU needed): tools/build_variants.sh "name:-Dflags" ...   -> _ab/libpbrt_<name>.so
ROOT=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $ROOT/_ab; cd $ROOT/physics-based-ray-tracing_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared $flags -o $ROOT/_ab/libpbrt_$name.so pbrt_api.hip 2>/dev/null || echo "build failed: $name" ) &
  while [ $(jobs -r | wc -l) -ge 6 ]; do sleep 1; done
done
wait; ls -la $ROOT/_ab/*.so | awk '{print $5, $9}'
