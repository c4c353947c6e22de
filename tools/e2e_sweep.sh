for K in 20 100; do
for cfg in "128 4" "256 4" "256 3" "512 4" "512 3" "512 2" "1024 2"; do
  set -- $cfg
  echo "=== K=$K chunk=$1 MiB workers=$2"
  P3D_CHUNK_MIB=$1 P3D_CHUNK_WORKERS=$2 timeout -k 10 120 python tools/e2e_timeline.py $K 2>&1 | grep "^call"
done; done
