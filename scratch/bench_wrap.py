import os, sys, traceback, atexit
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
atexit.register(lambda: print("atexit reached", file=sys.stderr, flush=True))
import bench
try:
    bench.main(sys.argv[1:])
    print("main returned", file=sys.stderr, flush=True)
except BaseException as e:
    print("EXC", type(e).__name__, e, file=sys.stderr, flush=True)
    traceback.print_exc()
