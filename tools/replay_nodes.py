"""Dev / profiles: the node list of ONE replayed step out of a `rocprofv3 --kernel-trace --output-format csv` trace of bench.py.

    python tools/replay_nodes.py <kernel_trace.csv> [--out profiles/r05_replay_nodes.txt] [--json profiles/r05_replay_nodes.json]

The optimizer launch (k_gradboost) ends every step: the kernels between two consecutive ones are one replay of the captured step (plus the optimizer
launch itself); the step with the median span is reported.  Reports the node count split into own kernels (k_* / frost_*), aten element-wise kernels, and runtime copy / fill kernels
(__amd_rocclr_*), the busy time, the gaps between consecutive kernels on the timeline, and one line per node.
"""
import argparse
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:90]


def kind_of(name):
    if name.startswith("__amd_rocclr") or "rocclr" in name:
        return "copy_fill"
    if re.match(r"^(void )?(k_|frost_)", name):
        return "own"
    return "aten"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--out")
    ap.add_argument("--json")
    ap.add_argument("--marker", default="k_gradboost")
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", ""), r.get("Queue_Id", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    if len(marks) < 2:
        sys.exit(f"fewer than two {a.marker} launches in the trace")
    # the step with the MEDIAN span among the replays (the last one can overlap the profiler's own teardown, the first ones are eager / capture: round 6 found a
    # 1.3 ms hole in the last step of one trace that no other step had -- tools/step_gaps.py lists every step)
    spans = sorted((max(r[1] for r in rows[marks[i] + 1: marks[i + 1] + 1]) - rows[marks[i] + 1][0], i) for i in range(len(marks) - 1))
    pick = spans[len(spans) // 2][1]
    lo, hi = marks[pick] + 1, marks[pick + 1] + 1
    step = rows[lo:hi]
    t0 = step[0][0]
    boundary_us = (t0 - rows[marks[pick]][1]) / 1e3        # end of the previous step's optimizer launch -> this step's first node
    cnt = collections.Counter(kind_of(r[2]) for r in step)
    busy = sum(e - s for s, e, *_ in step)
    # timeline gaps: time in which NO kernel of the step is running
    gap, cur_end, ngap_big = 0, step[0][1], 0
    for s, e, *_ in step[1:]:
        if s > cur_end:
            gap += s - cur_end
            ngap_big += (s - cur_end) > 3000
        cur_end = max(cur_end, e)
    span = cur_end - t0
    by = collections.defaultdict(lambda: [0, 0])
    for s, e, n, *_ in step:
        k = short(n)
        by[k][0] += 1
        by[k][1] += e - s
    doc = dict(nodes_total=len(step), nodes_own=cnt["own"], nodes_aten=cnt["aten"], nodes_copy_fill=cnt["copy_fill"], span_us=span / 1e3, busy_sum_us=busy / 1e3,
               idle_us=gap / 1e3, gaps_over_3us=ngap_big, step_boundary_idle_us=boundary_us,
               non_own={k: v[0] for k, v in by.items() if kind_of(k) != "own"})
    lines = [f"# one replayed step: {len(step)} nodes = {cnt['own']} own + {cnt['aten']} aten + {cnt['copy_fill']} runtime copy/fill; span {span / 1e3:.1f} us, "
             f"sum of kernel durations {busy / 1e3:.1f} us, idle (no kernel running) {gap / 1e3:.1f} us, gaps > 3 us: {ngap_big}; "
             f"idle between the previous step's optimizer launch and this step's first node: {boundary_us:.1f} us"]
    lines.append("# by kernel (count, total us):")
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"#   {v[0]:4d} {v[1] / 1e3:9.1f}  {k}")
    lines.append("# timeline: start_us  dur_us  gap_before_us  stream  kernel")
    prev_end = t0
    for s, e, n, st, q in step:
        lines.append(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:7.1f}  {st or q:>3}  {short(n)}")
        prev_end = max(prev_end, e)
    txt = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(txt)
    else:
        sys.stdout.write(txt)
    if a.json:
        json.dump(doc, open(a.json, "w"), indent=1)
    print(json.dumps(doc), file=sys.stderr)


if __name__ == "__main__":
    main()
