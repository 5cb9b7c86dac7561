#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
for m in lumina_static anole_static llamagen_static lumina; do timeout -k 10 300 python tools/draft_bench.py $m 1200 30 > $O/$m.json 2> $O/$m.err || tail -5 $O/$m.err; cat $O/$m.json; done
LANTERN_NO_PLAN=1 timeout -k 10 300 python - <<'PY' > $O/lumina_static_python_loop.json 2> $O/py.err || tail -5 $O/py.err
import runpy, sys
from lantern_amd.drafters import cnets
cnets.Model.use_depth_plan = False
sys.argv = ["draft_bench.py", "lumina_static", "1200", "30"]
runpy.run_path("tools/draft_bench.py", run_name="__main__")
PY
cat $O/lumina_static_python_loop.json
