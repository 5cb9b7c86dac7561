#!/bin/bash
for N in 4 10 40; do
for cfg in "1 1" "2 1" "4 1" "8 1" "16 1" "0 2"; do set -- $cfg; LANTERN_TA_SPLITS=$1 LANTERN_TA_MIN_TILES=$2 python tools/probe/ta_draft_shape.py $N 1210 2>/dev/null | tail -1; done; done
