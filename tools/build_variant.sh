#!/bin/bash
# bash tools/build_variant.sh <name> [-DKNOB=val ...] : rebuild the N=1024 (and optionally other) units with
# experiment knobs and link pseudo-3d-interpolation_amd/libp3d_hip_<name>.so (build container, no GPU needed)
set -e
name=$1; shift
CS=/root/repo/pseudo-3d-interpolation_amd/csrc
make -C $CS -j8 >/dev/null
mkdir -p $CS/build_$name
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -munsafe-fp-atomics -I/root/repo/include -I$CS -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -DP3D_N=1024 -c $CS/p3d_inst.hip -o $CS/build_$name/inst_1024.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|Occupancy|LDS Size" | paste - - - - | grep -E "error|Li1024ELi1E|Li1024ELi[0-9]+ELi0E|pipe_kernelILi1024ELb1ELi0ELb0" \
  | sed -e 's/\[-Rpass[^]]*\]//g' -e 's/[^ ]*p3d_kernels.hpp:[0-9]*:1: remark://g' -e 's/Function Name: //' || true
/opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/p3d_api.hip -o $CS/build_$name/api.o
objs="$CS/build/generic.o $CS/build/wavelet.o $CS/build/shearlet.o $CS/build/flex.o $CS/build/chirp.o $CS/build/smooth.o $CS/build/resident.o $CS/build/select.o $CS/build/f64.o"; for n in 2 4 8 16 32 64 128 256 512 2048 4096; do objs="$objs $CS/build/inst_$n.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/pseudo-3d-interpolation_amd/libp3d_hip_$name.so $objs $CS/build_$name/inst_1024.o $CS/build_$name/api.o
echo "built libp3d_hip_$name.so"
