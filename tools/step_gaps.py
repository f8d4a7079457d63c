"""Dev: per replayed step of a `rocprofv3 --kernel-trace --output-format csv` trace of bench.py, the largest timeline gaps (time in which no kernel runs) and the
kernels on either side of each -- to tell a gap of the captured graph (same place every step) from a tracing artefact (random places).

    python tools/step_gaps.py <kernel_trace.csv> [--top 4]
"""
import argparse
import csv
import re


def short(name):
    return re.sub(r"^void ", "", re.sub(r"\(.*$", "", name))[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--top", type=int, default=4)
    ap.add_argument("--marker", default="k_gradboost")
    a = ap.parse_args()
    rows = []
    with open(a.trace, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if a.marker in r[2]]
    for si in range(len(marks) - 1):
        step = rows[marks[si] + 1: marks[si + 1] + 1]
        t0, cur_end, prev, gaps, idle = step[0][0], step[0][1], step[0], [], 0
        for r in step[1:]:
            if r[0] > cur_end:
                gaps.append((r[0] - cur_end, (cur_end - t0) / 1e3, short(prev[2]), short(r[2]), r[3]))
                idle += r[0] - cur_end
            if r[1] >= cur_end:
                cur_end, prev = r[1], r
        gaps.sort(reverse=True)
        print(f"step {si}: {len(step)} launches, span {(cur_end - t0) / 1e3:.0f} us, idle {idle / 1e3:.0f} us; largest gaps: " +
              "; ".join(f"{g[0] / 1e3:.0f} us at {g[1]:.0f} ({g[2]} -> {g[3]}, queue {g[4]})" for g in gaps[: a.top]))


if __name__ == "__main__":
    main()
