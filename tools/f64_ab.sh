echo "default:"; python tools/f64_bench.py 2>/dev/null | tail -1
echo "table global:"; P3D_F64_TW_GLOBAL=1 python tools/f64_bench.py 2>/dev/null | tail -1
echo "one per CU, tiles 4/4:"; P3D_F64_ONE_PER_CU=1 P3D_F64_COL_TILE=4 P3D_F64_ROW_TILE=4 python tools/f64_bench.py 2>/dev/null | tail -1
echo "complex128 1000x1000 soft:"; NIL=1000 NXL=1000 OP=soft DTYPE=complex128 python tools/f64_bench.py 2>/dev/null | tail -1
