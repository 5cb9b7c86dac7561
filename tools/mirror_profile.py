"""cProfile of tools/mirror_bench.py's timed generate() call: where the host time of the drop-in step goes (top functions by own and cumulative time).
usage: mirror_profile.py [steps]"""
import cProfile
import io
import os
import pstats
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "400"
sys.argv = [os.path.join(ROOT, "tools", "mirror_bench.py"), steps]
pr = cProfile.Profile()
pr.enable()
runpy.run_path(sys.argv[0], run_name="__main__")
pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(28)
    print(s.getvalue()[:6000])
