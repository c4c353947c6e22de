# chunk size (MiB) x workers of the host-buffer entry point: bash tools/e2e_sweep.sh   (CFGS="128 4;64 8" KS="20 100")
IFS=';' read -ra CF <<< "${CFGS:-128 4;128 6;128 8;64 6;64 8;256 4}"
for K in ${KS:-20 100}; do
for cfg in "${CF[@]}"; do
  set -- $cfg
  echo "=== K=$K chunk=$1 MiB workers=$2"
  P3D_CHUNK_MIB=$1 P3D_CHUNK_WORKERS=$2 timeout -k 10 120 python tools/e2e_timeline.py $K 2>&1 | grep "^call"
done; done
