#!/usr/bin/env python
"""HBM traffic per launch of the hot-path kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
in SEPARATE runs, as MI355X_MICROARCH.md prescribes) of the bench command, written as profiles/rNN_pmc_traffic.json --
the file bench.py quotes (with its provenance) under roofline.traffic.

  python tools/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE --out profiles/r02_pmc_traffic.json

Units: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts a 128-byte read request as 64 bytes,
so wide coalesced read streams are doubled (the guide's correction; calibrated here on the Bernoulli kernels whose
algorithmic read volume is known exactly)."""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "zhusuan-pytorch_amd"))

B, K, D, X = 256, 50, 40, 784
N = B * K
ALGO = {
    "zs_bernoulli_logprob_f32": 4 * N * X + 4 * B * X + 4 * N,
    "zs_bernoulli_logprob_bwd_f32": 8 * N * X + 4 * B * X + 4 * N,
    "zs_bernoulli_logits_logprob_f32": 4 * N * X + 4 * B * X + 4 * N,
    "zs_bernoulli_logits_logprob_bwd_f32": 8 * N * X + 4 * B * X + 4 * N,
    "zs_normal_sample_logprob_pair_f32": 2 * (4 * N * D + 4 * N) + 8 * B * D,      # both draws of the latent in one launch (the bench step)
    "zs_normal_logprob_f32": 4 * N * D + 4 * N + 8 * B * D,
    "zs_normal_logprob_bwd_ksum_f32": 4 * N * D + 4 * N + 16 * B * D,
    "zs_iw_objective_f32": 20 * N + 4 * B + 4,
    "zs_adam_step_f32": 28 * 1346864,
    # IW1 (round 4): the generator side of the objective in one launch each way
    "zs_bernoulli_iw_objective_f32": 4 * N * X + 4 * B * X + 4 * N * D + 8 * B * D + 4 * N + 16 * N + 8 * B + 4,
    "zs_bernoulli_iw_objective_bwd_f32": (8 * N * X + 4 * B * X + 4 * N) + (4 * N * D + 4 * N + 16 * B * D),
}
# (the kernels shared by the location-scale families live in namespace zs: template argument 0 = Normal)
FRAGS = [("k_iw1_persist", "zs_bernoulli_iw_objective_f32"), ("k_iw1_block", "zs_bernoulli_iw_objective_f32"),
                 ("k_iw1_bwd", "zs_bernoulli_iw_objective_bwd_f32"),
         ("k_bern_logprob_bwd", "zs_bernoulli%s_logprob_bwd_f32"), ("k_bern_logprob", "zs_bernoulli%s_logprob_f32"),
         ("k_sample_tile<0", "zs_normal_sample_logprob_pair_f32"), ("k_logprob_bwd_ksum<0", "zs_normal_logprob_bwd_ksum_f32"),
         ("k_logprob_krep<0", "zs_normal_logprob_f32"), ("k_adam_step<float", "zs_adam_step_f32"),
         ("k_normal_sample_bwd", None), ("k_normal_sample", None),
         ("k_normal_logprob_bwd_ksum", "zs_normal_logprob_bwd_ksum_f32"), ("k_normal_logprob_bwd", None),
         ("k_normal_logprob", "zs_normal_logprob_f32"), ("k_iw_", "zs_iw_objective_f32")]


def entry_of(name):
    for frag, entry in FRAGS:
        i = name.find(frag)
        if i >= 0:
            if entry and "%s" in entry:
                j = name.find("<", i)
                return entry % ("_logits" if (j >= 0 and name[j + 1:j + 5] == "true") else "")
            return entry
    return None


def mean_counter(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                e = entry_of(row["Kernel_Name"])
                if e:
                    a = acc.setdefault(e, {"kernel": row["Kernel_Name"].split("(")[0][-60:], "v": []})
                    a["v"].append(float(row["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--out", required=True)
    ap.add_argument("--fused-logits", action="store_true")
    ap.add_argument("--command", default="python3 bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-extras")
    a = ap.parse_args()
    from zhusuan import _hip
    fetch, write = mean_counter(a.fetch_dir, "FETCH_SIZE"), mean_counter(a.write_dir, "WRITE_SIZE")
    kernels = {}
    for e in sorted(set(fetch) & set(write)):
        if e not in ALGO:
            continue
        f_kib = sum(fetch[e]["v"]) / len(fetch[e]["v"])
        w_kib = sum(write[e]["v"]) / len(write[e]["v"])
        hbm = (2.0 * f_kib + w_kib) * 1024.0
        kernels[e] = {"kernel": fetch[e]["kernel"], "FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib, "launches": len(fetch[e]["v"]),
                      "hbm_bytes_corrected": hbm, "algorithmic_bytes": ALGO[e], "traffic_over_algorithmic": hbm / ALGO[e]}
    doc = {"tool": "rocprofv3 --kernel-trace --pmc <C> (one counter per pass), " + a.command,
           "units": "FETCH_SIZE / WRITE_SIZE are reported in KiB; gfx950 FETCH_SIZE counts 128-B read requests at 64 B, so wide "
                    "coalesced reads are doubled (MI355X_MICROARCH.md, HBM section)",
           "workload": "IWAE-MNIST VIMCO batch=256 K=50 (config 3), per launch",
           "abi_version": _hip.ABI_VERSION, "fused_logits": bool(a.fused_logits), "launch_mode": "eager", "kernels": kernels}
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    for e, k in kernels.items():
        print("%-40s %10.0f B  = %.3f x algorithmic" % (e, k["hbm_bytes_corrected"], k["traffic_over_algorithmic"]))


if __name__ == "__main__":
    main()
