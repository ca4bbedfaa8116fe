"""Run one GEMM shape repeatedly (for rocprofv3 --pmc):  python tools/gemm_one.py kind M N K [tile]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
os.environ["ETP_GEMM_TILE"] = sys.argv[5] if len(sys.argv) > 5 else ""
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_bench
print(kind, M, N, K, os.environ["ETP_GEMM_TILE"], gemm_bench.run(kind, M, N, K, iters=20))
