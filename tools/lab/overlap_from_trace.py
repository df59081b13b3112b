#!/usr/bin/env python3
"""How much do consecutive launches of one kernel overlap? Reads a rocprofv3 --kernel-trace CSV (…_kernel_trace.csv).
usage: overlap_from_trace.py <kernel_trace.csv> [name-substring]
Prints, per queue id, the launch count and, over the launches sorted by start, the distribution of
  gap  = start[i+1] - end[i]   (negative: launch i+1 began before launch i ended = overlap)
  dur  = end - start, pitch = start[i+1] - start[i]."""
import csv, statistics, sys
rows = []
sub = sys.argv[2] if len(sys.argv) > 2 else "ed_mfcc2_kernel"
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
print("%d launches of *%s*" % (len(rows), sub))
byq = {}
for s, e, q, st in rows:
    byq.setdefault((q, st), 0)
    byq[(q, st)] += 1
for k, v in sorted(byq.items()):
    print("  queue %s stream %s: %d launches" % (k[0], k[1], v))

def pct(x, p):
    x = sorted(x)
    return x[min(len(x) - 1, int(p / 100.0 * len(x)))]

def report(tag, sel):
    if len(sel) < 3:
        return
    gaps = [sel[i + 1][0] - sel[i][1] for i in range(len(sel) - 1)]
    durs = [e - s for s, e, _, _ in sel]
    pitch = [sel[i + 1][0] - sel[i][0] for i in range(len(sel) - 1)]
    # drop the pauses between timed blocks (host synchronisations): gaps above 20 us are not launch-to-launch gaps
    g2 = [g for g in gaps if g < 20000]
    p2 = [p for p, g in zip(pitch, gaps) if g < 20000]
    print("%s: n=%d  dur us p5/p50/p95 %.2f/%.2f/%.2f   gap us p5/p50/p95 %.2f/%.2f/%.2f   pitch us p50 %.2f   overlapping pairs %.1f %%" % (
        tag, len(sel), pct(durs, 5) / 1e3, pct(durs, 50) / 1e3, pct(durs, 95) / 1e3, pct(g2, 5) / 1e3, pct(g2, 50) / 1e3, pct(g2, 95) / 1e3,
        statistics.median(p2) / 1e3, 100.0 * sum(1 for g in g2 if g < 0) / max(1, len(g2))))

# segments of the run in which the queue pattern is constant: split where the set of queues in a sliding window changes
report("all launches", rows)
seg, cur = [], [rows[0]]
for a_, b_ in zip(rows, rows[1:]):
    if b_[0] - a_[1] > 200000:      # > 200 us pause = a new timed block
        seg.append(cur); cur = []
    cur.append(b_)
seg.append(cur)
for i, s_ in enumerate(seg):
    qs = sorted(set(x[2] for x in s_))
    if len(s_) >= 20:
        print("block %d: %d launches in %.1f us -> %.2f us per launch; per queue median duration: %s" % (
            i, len(s_), (max(x[1] for x in s_) - s_[0][0]) / 1e3, (max(x[1] for x in s_) - s_[0][0]) / 1e3 / len(s_),
            ", ".join("q%s %.1f us" % (q_, statistics.median([x[1] - x[0] for x in s_ if x[2] == q_]) / 1e3) for q_ in qs)))
        mid = s_[len(s_) // 2: len(s_) // 2 + 8]
        print("   mid-block timeline (us from the first of them): " + "  ".join("q%s[%.1f..%.1f]" % (x[2], (x[0] - mid[0][0]) / 1e3, (x[1] - mid[0][0]) / 1e3) for x in mid))
        report("block %d (%d launches, queues %s)" % (i, len(s_), ",".join(qs)), s_)
