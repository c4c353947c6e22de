#!/bin/bash
# bash tools/build_variant_n.sh <name> <N> [-DKNOB=val ...] : like build_variant.sh for the units of line length N
set -e
name=$1; n=$2; shift; shift
CS=/root/repo/pseudo-3d-interpolation_amd/csrc
make -C $CS -j8 >/dev/null
mkdir -p $CS/build_$name
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -munsafe-fp-atomics -I/root/repo/include -I$CS -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc $FLAGS "$@" -DP3D_N=$n -c $CS/p3d_inst.hip -o $CS/build_$name/inst_$n.o
objs="$CS/build/api.o $CS/build/generic.o $CS/build/wavelet.o $CS/build/shearlet.o $CS/build/flex.o $CS/build/smooth.o $CS/build/resident.o $CS/build/select.o"
for m in 2 4 8 16 32 64 128 256 512 1024 2048 4096; do [ $m = $n ] || objs="$objs $CS/build/inst_$m.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/pseudo-3d-interpolation_amd/libp3d_hip_$name.so $objs $CS/build_$name/inst_$n.o
echo "built libp3d_hip_$name.so"
