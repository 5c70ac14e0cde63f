"""Timeline of the LAST training step in a rocprofv3 kernel trace (kernel_trace.csv): per kernel the start offset from the
step's first launch, duration, queue and name — which launches overlap, where a stream idles, what the tail is.
   python tools/timeline.py <kernel_trace.csv> [--tail-ms 2.5]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.split("(")[0]
    return re.sub(r"<.*", "", n) + (re.search(r"<([^>]*)>", n).group(0)[:28] if "<" in n else "")


def main():
    path = sys.argv[1]
    tail_ms = float(sys.argv[sys.argv.index("--tail-ms") + 1]) if "--tail-ms" in sys.argv else 2.5
    rows = list(csv.DictReader(open(path)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"])) for r in rows))
    adam = [i for i, e in enumerate(ev) if e[3].startswith("adamw_kernel")]
    if len(adam) < 2:
        print("fewer than two AdamW launches in the trace"); return
    a, b = adam[-2], adam[-1]
    step = ev[a + 1:b + 1]
    # the step starts at the first launch after the previous AdamW's transposes
    t0, t1 = step[0][0], step[-1][1]
    print(f"last step: {len(step)} launches, {(t1 - t0) / 1e6:.3f} ms from the first launch after AdamW to the end of the next AdamW")
    queues = sorted({e[2] for e in step})
    busy = {q: sum(e[1] - e[0] for e in step if e[2] == q) / 1e6 for q in queues}
    print("busy ms per queue:", {q: round(v, 3) for q, v in busy.items()})
    cut = t1 - tail_ms * 1e6
    print(f"--- launches ending in the last {tail_ms} ms (start_us, dur_us, queue, kernel):")
    for s, e, q, n in step:
        if e >= cut:
            print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  q{q}  {n}")
    # gaps on the main queue (the one with the most busy time)
    mainq = max(busy, key=busy.get)
    m = [e for e in step if e[2] == mainq]
    gaps = sorted(((m[i + 1][0] - m[i][1]) / 1e3, m[i][3], m[i + 1][3]) for i in range(len(m) - 1))
    print("--- largest gaps on the main queue (us, after, before):")
    for g in gaps[-8:]:
        print(f"{g[0]:8.1f}  {g[1]} -> {g[2]}")
    print(f"sum of main-queue gaps: {sum(g[0] for g in gaps) / 1e3:.3f} ms over {len(gaps)} boundaries")
    # the step by kernel name and queue: launches, total and mean duration
    import collections
    by = collections.defaultdict(lambda: [0, 0.0])
    for s, e, q, n in step:
        by[(q, n)][0] += 1
        by[(q, n)][1] += (e - s) / 1e3
    print("--- the step by kernel (queue, launches, total us, mean us):")
    for (q, n), (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print(f"  q{q} {c:4d} {t:9.1f} {t / c:8.1f}  {n}")


if __name__ == "__main__":
    main()
