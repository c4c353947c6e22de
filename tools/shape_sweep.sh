#!/bin/bash
# Run on the GPU box: iterations/s and points/s of the FFT path for a list of "nil nxl" shapes (128 slices, 20 iterations)
for shp in "$@"; do
  set -- $shp
  timeout -k 10 300 python bench.py --nil $1 --nxl $2 --nslices 128 --steps 20 --warmup 2 --no-cpu-baseline --no-profile --no-dense --repeats 3 2>/dev/null | tail -1 \
   | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 x $2:', round(d['value'],1), 'it/s on 128 slices =', round(d['slice_iterations_per_s']*$1*$2/1e9,2), 'Gpt/s')" || exit 1
done
