#!/bin/bash
# build a variant of the library into tools/ab/lib_<name>.so:  tools/ab/build.sh name [extra hipcc flags, e.g. -DSTEP_W=3]
name=$1; shift
R=$(cd $(dirname $0)/../.. && pwd); O=/tmp/ab_$name; mkdir -p $O
for s in gemm gemm32 cholstep cholstep32 distops engine smalln vfe kron select predict api; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c $R/gpim_amd/csrc/$s.hip -o $O/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/lib_$name.so $O/*.o && echo built lib_$name.so
