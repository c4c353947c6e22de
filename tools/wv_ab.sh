for lib in ${LIBS:-libp3d_hip.so libp3d_hip_w512.so libp3d_hip.so libp3d_hip_w512.so}; do
  echo -n "$lib: "; P3D_LIB_PATH=$PWD/pseudo-3d-interpolation_amd/$lib timeout -k 10 200 python3 tools/wavelet_bench.py 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_iteration'], d['iterations_per_s'], d['rel_l2_vs_oracle'])"
done
