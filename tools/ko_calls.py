"""timing only: run bench.py with some C-ABI calls left out (wrong results).  python ko_calls.py name1,name2 [bench args]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
skip = set(sys.argv[1].split(',')) - {''}
sys.argv = ['bench.py'] + sys.argv[2:]
from grl_amd import engine, train_engine
orig = engine._call
def call(name, *a):
    if name in skip:
        return 0
    return orig(name, *a)
engine._call = call
train_engine._call = call
import bench
bench.main()
