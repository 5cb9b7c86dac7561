"""Per-stream-group timeline of the timed loop from a rocprofv3 --kernel-trace CSV: for every queue, the sequence prepare -> walk -> commit with each
kernel's duration and the gap in front of it, for the last steps of the run (the timed region + event pass).  usage: timeline.py <dir with *kernel_trace.csv> [n_last_steps]"""
import csv
import glob
import statistics as st
import sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
kn = lambda r: r.get("Kernel_Name") or r.get("kernel_name")
role = lambda n: "prep" if "prep_rows_kernel" in n else ("walk" if "epw_kernel" in n else ("commit" if "update_inputs" in n else None))
byq = {}
for r in rows:
    ro = role(kn(r))
    if ro is None:
        continue
    q = r.get("Queue_Id") or r.get("queue_id")
    byq.setdefault(q, []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), ro))
out = {}
for q, ev in byq.items():
    ev.sort()
    if len(ev) < 3 * 40:
        continue
    ev = ev[-3 * n_last - 3 * 130:-3 * 125] if False else ev          # keep everything; select below
    # steady state of the timed loop: three launches per step on this queue; take a window near the middle of the recorded run
    mid = len(ev) // 2
    win = ev[max(0, mid - 3 * n_last // 2): mid + 3 * n_last // 2]
    dur, gap, per = {}, {}, []
    last_end, last_prep_start = None, None
    for s, e, ro in win:
        dur.setdefault(ro, []).append((e - s) / 1e3)
        if last_end is not None:
            gap.setdefault(ro, []).append((s - last_end) / 1e3)
        if ro == "prep":
            if last_prep_start is not None:
                per.append((s - last_prep_start) / 1e3)
            last_prep_start = s
        last_end = e
    out[q] = {"launches": len(ev), "dur_us": {k: round(st.median(v), 1) for k, v in dur.items()}, "gap_before_us": {k: round(st.median(v), 1) for k, v in gap.items()},
              "step_period_us": round(st.median(per), 1) if per else None}
for q, v in out.items():
    print("queue", q, v)
# overlap: at the start of each commit, how many other commits are running
allc = sorted((s, e) for q, ev in byq.items() for s, e, ro in ev if ro == "commit")
if allc:
    mid = len(allc) // 2
    ov = []
    for s, e in allc[mid - 100: mid + 100]:
        ov.append(sum(1 for s2, e2 in allc[max(0, mid - 300): mid + 300] if s2 < e and e2 > s) - 1)
    print("commits overlapping a commit (median / mean):", st.median(ov), round(st.mean(ov), 2))
