python -m pytest tests/test_gpu_parity.py -q -m gpu -k "flexible or real_cubes or fft2 or time2freq or early_exit or slower" 2>&1 | tail -3
bash tools/shape_sweep.sh "1001 999" "1009 1013" "1101 1451" "520 530" "999 999" "500 509" "2039 1999" "74 62" "300 331" "1000 1000"
