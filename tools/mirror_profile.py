"""cProfile of the TIMED generate() call of tools/mirror_bench.py (MIRROR_PROFILE=1): where the host time of the drop-in step goes.
usage: mirror_profile.py [steps]   (the profile goes to stderr, the bench line to stdout; times are inflated by the profiler: read the proportions)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "400"
r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "mirror_bench.py"), steps], env=dict(os.environ, MIRROR_PROFILE="1"), capture_output=True, text=True)
print(r.stdout[-600:])
print(r.stderr[-14000:])
