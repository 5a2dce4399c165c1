"""bench.py against another build of the library: ABL_LIB=/path/to/lib.so python profiles/microbench/bench_lib.py <bench flags>
(A/B of kernel changes on ONE box in ONE gpurun call: boxes differ by +-1 %, see profiles/microbench/ab_lib.sh)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'centernet-uda_amd'))
import hip_runtime as hr
if os.environ.get('ABL_LIB'):
    hr.LIB_PATH = os.environ['ABL_LIB']
import bench
bench.main()
