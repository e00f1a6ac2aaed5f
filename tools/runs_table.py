"""Markdown table of bench lines side by side:  python tools/runs_table.py profiles/r05_bench_n1*.json
(one row per file; the columns are the ones profiles/rNN_runs.md quotes)."""
import json
import sys


def g(d, *path, fmt="%.3f"):
    for p in path:
        if not isinstance(d, dict) or p not in d:
            return "–"
        d = d[p]
    return fmt % d if isinstance(d, (int, float)) else str(d)


def main(files):
    cols = ["file", "library", "headline ms (M evals/s)", "roofline.frac (IW1 bwd µs)", "iw1_fwd_frac", "k1_frac_1M / _4M (sustained clock)", "k1_frac_1M_cold / _4M_cold (after 1 s idle)",
            "c3_dp_step_n1 ms (vs_headline / same-process ratio / +µs)", "hbm_resident_frac", "k3_fwd_frac", "c3_refresh (vs headline)", "c3_eager / _torch_linear", "c3_reference_example / _graphed",
            "c5 / c5_eager / c5_reference_example / _graphed", "c2", "iwae_default / bnn_default", "c3_forward_only",
            "CPU 16 thr / fwd+bwd only / 1 thr (k evals/s)"]
    print("| " + " | ".join(cols) + " |")
    print("|" + "---|" * len(cols))
    for f in files:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r, e, c = d.get("roofline", {}), d.get("extra_configs", {}), d.get("cpu_baseline") or {}
        ms = lambda n: g(e, n, "ms_per_step")
        k = lambda v: "–" if not isinstance(v, (int, float)) else "%.1f" % (v / 1e3)
        row = [f.split("/")[-1], g(d, "library", "sha256")[:8] + "…",
               "%.4f (%.2f)" % (d["ms_per_step"], d["value"] / 1e6),
               "%s (%s)" % (g(r, "frac"), g(r, "avg_launch_us", fmt="%.2f")), g(r, "iw1_fwd_frac"),
               "%s / %s" % (g(r, "k1_frac_1M"), g(r, "k1_frac_4M")), "%s / %s" % (g(r, "k1_frac_1M_cold"), g(r, "k1_frac_4M_cold")),
               "%s (%s / %s / %s)" % (ms("c3_dp_step_n1"), g(e, "c3_dp_step_n1", "vs_headline", fmt="%.4f"),
                                      g(e, "c3_dp_step_n1", "same_process_ratio", fmt="%.4f"), g(e, "c3_dp_step_n1", "extra_us_per_step", fmt="%.1f")),
               g(r, "hbm_resident_frac"), g(r, "k3_fwd_frac"),
               "%s (%s)" % (ms("c3_refresh"), g(e, "c3_refresh", "vs_headline", fmt="%.4f")),
               "%s / %s" % (ms("c3_eager"), ms("c3_eager_torch_linear")),
               "%s / %s" % (ms("c3_reference_example"), ms("c3_reference_example_graphed")),
               " / ".join(ms(n) for n in ("c5", "c5_eager", "c5_reference_example", "c5_reference_example_graphed")),
               ms("c2"), "%s / %s" % (ms("iwae_default"), ms("bnn_default")), ms("c3_forward_only"),
               "%s / %s / %s" % (k(c.get("value")), k(c.get("fwd_bwd_only_value")), k(c.get("one_thread_value")))]
        print("| " + " | ".join(row) + " |")


if __name__ == "__main__":
    main(sys.argv[1:])
