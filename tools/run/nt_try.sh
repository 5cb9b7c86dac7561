#!/bin/bash
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["model"], round(d["us_per_cycle_wall"],1))'
for thr in 30 80 95 120 200 80; do
  echo -n "nt_min_mb=$thr: "; LANTERN_SK_NT_MIN_MB=$thr python tools/draft_bench.py lumina_static 1200 30 2>/dev/null | python -c "$P"
done
for m in lumina anole anole_static llamagen llamagen_static; do
  for thr in -1 80; do echo -n "nt_min_mb=$thr: "; LANTERN_SK_NT_MIN_MB=$thr python tools/draft_bench.py $m 1200 30 2>/dev/null | python -c "$P"; done
done
