#!/bin/bash
# variant of the library that differs in ONE source: tools/ab/build1.sh name source(.hip, without suffix) [extra flags]; the other objects
# come from gpim_amd/build (python -m gpim_amd._build first)
name=$1; src=$2; shift; shift
R=$(cd $(dirname $0)/../.. && pwd); O=/tmp/ab1_$name; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c $R/gpim_amd/csrc/$src.hip -o $O/$src.o 2>/dev/null || exit 1
objs=""
for s in gemm gemm32 cholstep cholstep32 distops engine smalln vfe kron select predict api; do
  if [ $s = $src ]; then objs="$objs $O/$s.o"; else objs="$objs $R/gpim_amd/build/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/lib_$name.so $objs && echo built lib_$name.so
