for shp in "1009 1013" "1101 1451" "520 530"; do set -- $shp
python3 tools/pocs_driver.py --nil $1 --nxl $2 --nslices 128 --niter 20 2>&1 | grep colpass
python3 tools/pocs_driver.py --nil $1 --nxl $2 --nslices 128 --niter 20 --real 2>&1 | grep colpass
done
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "flexible or real_cubes or fft2" 2>&1 | tail -3
