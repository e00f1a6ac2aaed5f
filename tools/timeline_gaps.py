"""Where a step's GPU time goes BETWEEN kernels: reads the chrome trace `bench.py --timeline PATH` writes (5 steps in the timed
region's launch mode) and prints, per step, the device-busy time, the idle time and every idle gap above a threshold with the
kernels on either side -- graph-launch boundaries, eager collectives and event waits show up as gaps, not as kernels.

    python tools/timeline_gaps.py trace.json [min_gap_us=3.0]
"""
import json
import sys


def short(name, n=58):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def main():
    path = sys.argv[1]
    min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    with open(path) as f:
        tr = json.load(f)
    ev = tr["traceEvents"] if isinstance(tr, dict) else tr
    dev = [e for e in ev if e.get("ph") == "X" and e.get("cat") in ("kernel", "gpu_memcpy", "gpu_memset")]
    dev.sort(key=lambda e: e["ts"])
    if not dev:
        print("no device activity in the trace")
        return
    # steps end with the optimizer's kernel
    ends = [i for i, e in enumerate(dev) if "adam" in e["name"].lower()]
    last_of_step = []
    for i in ends:                      # several Adam launches per step: keep the last of each run of launches
        if last_of_step and i - last_of_step[-1] <= 2:
            last_of_step[-1] = i
        else:
            last_of_step.append(i)
    bounds = [0] + [i + 1 for i in last_of_step]
    steps = [dev[a:b] for a, b in zip(bounds[:-1], bounds[1:]) if b > a]
    print("%d device activities, %d steps (split at the optimizer's kernel)" % (len(dev), len(steps)))
    rows = []
    for si, st in enumerate(steps):
        t0 = st[0]["ts"]
        busy_end = t0
        busy = 0.0
        gaps = []
        prev = None
        for e in st:
            s, d = e["ts"], e.get("dur", 0)
            if s > busy_end:
                if prev is not None and s - busy_end >= min_gap:
                    gaps.append((s - busy_end, busy_end - t0, prev["name"], e["name"]))
                busy += d
                busy_end = s + d
            else:                      # overlaps what is already running (another stream)
                if s + d > busy_end:
                    busy += s + d - busy_end
                    busy_end = s + d
            prev = e
        span = busy_end - t0
        rows.append((span, busy, span - busy, len(st), gaps))
    for si, (span, busy, idle, n, gaps) in enumerate(rows):
        print("step %d: %d activities, span %.1f us, busy %.1f us, idle %.1f us (%d gaps >= %.1f us: %.1f us)" % (
            si, n, span, busy, idle, len(gaps), min_gap, sum(g[0] for g in gaps)))
    # between steps
    for a, b in zip(steps[:-1], steps[1:]):
        ea = max(e["ts"] + e.get("dur", 0) for e in a)
        print("   step boundary: %.1f us idle before %s" % (b[0]["ts"] - ea, short(b[0]["name"])))
    if rows:
        span, busy, idle, n, gaps = rows[len(rows) // 2]
        print("\ngaps of the middle step (us idle, at us into the step, after -> before):")
        for g, at, a, b in gaps:
            print("  %6.1f  @%7.1f  %s  ->  %s" % (g, at, short(a), short(b)))
    # host side: graph launches and collectives
    host = {}
    for e in ev:
        if e.get("ph") == "X" and e.get("cat") in ("cuda_runtime", "cuda_driver"):
            host.setdefault(e["name"], []).append(e.get("dur", 0))
    print("\nhost runtime calls (count, mean us, total us):")
    for k, v in sorted(host.items(), key=lambda kv: -sum(kv[1]))[:12]:
        print("  %-40s %5d %8.1f %9.1f" % (k, len(v), sum(v) / len(v), sum(v)))
    streams = {}
    for e in dev:
        streams.setdefault(e.get("args", {}).get("stream", e.get("tid")), 0)
        streams[e.get("args", {}).get("stream", e.get("tid"))] += 1
    print("\ndevice activities per stream: %s" % streams)


if __name__ == "__main__":
    main()
