echo "default (2 per CU, half table in LDS):"; python tools/f64_bench.py 2>/dev/null | tail -1
echo "2 per CU, table global:"; P3D_F64_NO_HALF_TABLE=1 python tools/f64_bench.py 2>/dev/null | tail -1
echo "one per CU (table in LDS), tiles 2/2:"; P3D_F64_ONE_PER_CU=1 python tools/f64_bench.py 2>/dev/null | tail -1
echo "one per CU, tiles 4/4:"; P3D_F64_ONE_PER_CU=1 P3D_F64_COL_TILE=4 P3D_F64_ROW_TILE=4 python tools/f64_bench.py 2>/dev/null | tail -1
echo "tiles 1/1 (3 per CU):"; P3D_F64_COL_TILE=1 P3D_F64_ROW_TILE=1 python tools/f64_bench.py 2>/dev/null | tail -1
echo "2048 x 512:"; NIL=2048 NXL=512 python tools/f64_bench.py 2>/dev/null | tail -1
echo "4096 x 300:"; NIL=4096 NXL=300 NS=16 python tools/f64_bench.py 2>/dev/null | tail -1
echo "5000 x 64 (unfused fallback):"; NIL=5000 NXL=64 NS=8 K=5 python tools/f64_bench.py 2>/dev/null | tail -1
echo "complex128 1000x1000 soft:"; NIL=1000 NXL=1000 OP=soft DTYPE=complex128 python tools/f64_bench.py 2>/dev/null | tail -1
