# bench.py with a forced GEMM variant (argv[1] = nj), everything else identical
import sys, runpy
nj = int(sys.argv[1]); sys.argv = ["bench.py", "--no-cpu-baseline"]
sys.path.insert(0, ".")
import medvill_amd.hip_ops as ops
ops.set_gemm_variant(0, nj)
runpy.run_path("bench.py", run_name="__main__")
