#!/bin/bash
# build + run the packed-fp32 hazard reproducer (two objects: the compiler free to use v_pk_*_f32 / with -packed-fp32-ops); usage:
#   tools/micro/pk_hazard.sh [launches]      (on the GPU box; build only: PK_BUILD_ONLY=1)
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math"
hipcc $F -DPK_VARIANT=1 -c pk_hazard.hip -o /tmp/pk_hazard_pk.o
hipcc $F -DPK_VARIANT=0 -Xclang -target-feature -Xclang -packed-fp32-ops -c pk_hazard.hip -o /tmp/pk_hazard_sc.o 2> >(grep -v "is not a recognized feature" >&2)
hipcc --offload-arch=gfx950 /tmp/pk_hazard_pk.o /tmp/pk_hazard_sc.o -o pk_hazard
[ -n "$PK_BUILD_ONLY" ] || ./pk_hazard "${1:-10000}"
