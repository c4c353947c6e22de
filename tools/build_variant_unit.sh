#!/bin/bash
# bash tools/build_variant_unit.sh <name> <unit> [-DKNOB=val ...] : rebuild ONE translation unit (wavelet, shearlet, flex, chirp, f64, ...)
# with experiment knobs and link pseudo-3d-interpolation_amd/libp3d_hip_<name>.so from it and the default objects (build container)
set -e
name=$1; unit=$2; shift 2
CS=/root/repo/pseudo-3d-interpolation_amd/csrc
make -C $CS -j8 >/dev/null
mkdir -p $CS/build_$name
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -munsafe-fp-atomics -I/root/repo/include -I$CS -Wno-unused-value -Wno-unused-result"
case $unit in flex|chirp|f64) FLAGS="$FLAGS -ffp-contract=fast";; esac
/opt/rocm/bin/hipcc $FLAGS "$@" -c $CS/p3d_$unit.hip -o $CS/build_$name/$unit.o -Rpass-analysis=kernel-resource-usage 2>&1 \
  | grep -E "error|Function Name|VGPRs:|ScratchSize|Occupancy|LDS Size" | paste - - - - - | grep -E "error|${SHOW:-.}" \
  | sed -e 's/\[-Rpass[^]]*\]//g' -e 's/[^ ]*\.h[i]*p[p]*:[0-9]*:[0-9]*: remark://g' -e 's/Function Name: //' || true
objs=""; for o in $CS/build/*.o; do b=$(basename $o .o); [ "$b" = "$unit" ] || objs="$objs $o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o /root/repo/pseudo-3d-interpolation_amd/libp3d_hip_$name.so $objs $CS/build_$name/$unit.o
echo "built libp3d_hip_$name.so"
