cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export GPIMHIP_NO_EARLY_TRTRI=1
echo "== NT 4-wave"
GPIMHIP_NT_4WAVE=1 PROF_STAGES=1 python $R/tests/tools/prof_fit.py 16384 3 0 Matern52 2>&1 | grep -E "stage potrf"
cd $R/gpim_amd
for W in 8 6; do
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DOUTER_W=$W -DLOOKAHEAD_MIN_PANELS_DEFAULT=$((48/W)) -c csrc/api.hip -o build/api.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgpimhip.so build/gemm.o build/potf2.o build/engine.o build/smalln.o build/vfe.o build/api.o
echo "== OUTER_W=$W"
for n in 16384 8192 6400 4096; do PROF_STAGES=1 python $R/tests/tools/prof_fit.py $n 3 0 Matern52 2>&1 | grep -E "stage potrf"; done
done
