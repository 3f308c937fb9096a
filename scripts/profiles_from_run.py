"""Turn the rocprofv3 output of `scripts/profile_round.sh` into the tables bench.py reads back from profiles/.

    python scripts/profiles_from_run.py gpurun_out/<tag> <round>      # e.g. r05

reads   <dir>/<workload>/trace/**/*kernel_stats.csv        (rocprofv3 --kernel-trace --stats)
        <dir>/<workload>/pmc/**/*counter_collection.csv     (rocprofv3 --kernel-trace --pmc <4 fabric counters>)
        <dir>/<workload>/bench.json                          (the bench line printed by the traced run)
writes  <dir>/<round>_kernel_times.json    {workload: {C-ABI entry point: {avg_ms, calls, kernels}}}
        <dir>/<round>_pmc_traffic.json     {workload: {C-ABI entry point: {traffic_bytes, read_bytes, write_bytes}}}
        <dir>/<round>_<workload>_kernel_trace.md

A C-ABI entry point's time = the total time of every kernel it launches / the launch count of its main kernel —
the quantity the HIP events around the C-ABI call in bench.py measure.  Bytes per launch (gfx950, MI355X_MICROARCH.md
"HBM"): RDREQ x 128 B (RDREQ_32B x 32 B where non-zero) + WRREQ_64B x 64 B + other WRREQ x 32 B.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

# C-ABI entry point -> regexes of the kernels it launches (first = the main kernel, whose launch count counts the calls)
CABI = {
    "lr_fm_rows_adam_f32": [r"lr::fm_rows_adam_kernel"],
    "lr_fm_field_stats_f32": [r"lr::fm_field_stats_kernel"],
    "lr_deepfm_l1_fwd_sb_f32": [r"lr::l1_fwd_sb_kernel", r"lr::l1_sb_combine_kernel"],
    "lr_deepfm_l1_wgrad_sb_f32": [r"lr::l1_wgrad_sb_kernel"],
    "lr_deepfm_l1_dgrad_sb_f32": [r"lr::l1_dgrad_sb_kernel"],
    "lr_deepfm_l1_sb_pack": [r"lr::l1_sb_pack_kernel"],
    "lr_deepfm_l1_sb_gz_pack": [r"lr::l1_sb_gz_pack_kernel"],
    "lr_deepfm_l1_fwd_f32": [r"lr::l1_fwd\d*_kernel"],
    "lr_deepfm_l1_wgrad_f32": [r"lr::l1_wgrad_kernel"],
    "lr_deepfm_l1_dgrad_f32": [r"lr::l1_dgrad\d*_kernel"],
    "lr_mlp_tail3_f32": [r"lr::mlp_tail3_kernel"],
    "lr_adam_dense_rows_f32": [r"lr::adam_rows_kernel", r"lr::mark_slots_kernel", r"lr::clear_slots_kernel"],
    "lr_adam_dense_rows_dc_f32": [r"lr::adam_rows_kernel", r"lr::mark_slots_kernel", r"lr::clear_slots_kernel"],
    "lr_din_attn_pool_fwd_f32": [r"lr::din_fwd\w*_kernel"],
    "lr_din_attn_pool_bwd_parts_f32": [r"lr::din_bwd_data\w*_kernel", r"lr::din_bwd_param\w*_kernel", r"lr::din_bwd_kernel",
                                       r"lr::din_reduce\w*_kernel"],
    "lr_embed_scatter_adam_f32": [r"lr::seg_vec_kernel", r"lr::seg_scalar_kernel", r"lr::seg_long_\w+_kernel", r"lr::seg_adam_lin_kernel"],
    # the default arithmetic (six-term split-bf16 products) under the entry point's name, the f32 fma chain beside it
    "lr_softmax_ce_fwd_f32": [r"lr::softmax_ce_sb_kernel<\d+, 0,", r"lr::sce_merge_kernel"],
    "lr_softmax_ce_bwd_cols_f32": [r"lr::softmax_ce_sb_kernel<\d+, 1,"],
    "lr_softmax_ce_fwd_f32@f32_chain": [r"lr::softmax_ce_kernel<\d+, 0,"],
    "lr_softmax_ce_bwd_cols_f32@f32_chain": [r"lr::softmax_ce_kernel<\d+, 1,"],
    "lr_spmm_csr_bucketed_f32": [r"lr::spmm_bucketed_kernel<\d+, false, false>", r"lr::spmm_finish_kernel<\d+, false>", r"lr::spmm_vec_kernel"],
    "lr_spmm_csr_masked_f32": [r"lr::spmm_bucketed_kernel<\d+, true, false>"],
    "lr_spmm_csr_adam_f32": [r"lr::spmm_bucketed_kernel<\d+, false, true>", r"lr::spmm_finish_kernel<\d+, true>"],
    "lr_score_topk_f32": [r"lr::score_topk_kernel<\d+, \d+, 0, 1, false>", r"lr::topk_merge_keys_kernel<256>"],
    "lr_score_topk_sb_f32": [r"lr::score_topk_kernel<\d+, \d+, 1, 1, false>", r"lr::topk_merge_keys_kernel<256>"],
    # the filtered form: the one-term pass (pre-pass + main), its merge, the f32 rescoring, the compaction of the uncertified users, the masked exact
    # passes over them (return at once when every user is certified) and the row scatter (rocprofv3 prints the non-template kernels
    # without their namespace)
    "lr_score_topk_filter_f32": [r"lr::score_topk_kernel<\d+, 4, 2, 2, false>",      # (the 1,024-user form: two user tiles per wave; the bench's 1- and 32-user passes run the one-tile forms) r"lr::topk_merge_keys_kernel<512>", r"topk_rescore_kernel",
                                 r"lr::score_topk_kernel<\d+, \d+, \d+, 1, true>", r"topk_compact_failed_kernel", r"topk_scatter_rows_kernel"],
    "lr_pair_mlp_f32": [r"lr::pair_mlp_kernel"],
    "lr_pair_mlp_sb_f32": [r"lr::pair_mlp_sb_kernel"],
}
# launches of the main kernel per C-ABI call where it is not one (score_topk at >= 2^20 items: strided threshold
# pre-pass + main pass, csrc/score_topk.hip)
PER_CALL = {"lr_score_topk_f32": 2, "lr_score_topk_sb_f32": 2, "lr_score_topk_filter_f32": 2}


def find(d, pat):
    f = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return f[0] if f else None


def kernel_stats(path):
    """[(name, calls, total_ns, avg_ns, min_ns, max_ns, pct)] from rocprofv3's kernel_stats.csv."""
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]),
                     float(r["MinNs"]), float(r["MaxNs"]), float(r["Percentage"])))
    return sorted(rows, key=lambda r: -r[2])


def cabi_times(rows):
    out = {}
    for cabi, pats in CABI.items():
        main = [r for r in rows if re.search(pats[0], r[0])]
        if not main:
            continue
        calls = sum(r[1] for r in main) / PER_CALL.get(cabi, 1)
        hit = [r for r in rows if any(re.search(p, r[0]) for p in pats)]
        total = sum(r[2] for r in hit)
        out[cabi] = {"avg_ms": round(total / calls / 1e6, 5), "calls": calls,
                     "kernels": {r[0][:110]: {"calls": r[1], "avg_us": round(r[3] / 1e3, 2)} for r in hit}}
    return out


def counters_by_kernel(path):
    """{kernel: {counter: [n, mean]}} from a counter_collection.csv (kept beside it as pmc_kernels.json: the csv of a full-size
    pass is tens of MB and is deleted on the GPU box)."""
    side = os.path.join(os.path.dirname(path), "pmc_kernels.json")
    if path.endswith(".json"):
        return json.load(open(path))
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {n: [len(v), sum(v) / len(v)] for n, v in c.items()} for k, c in agg.items()}
    json.dump(out, open(side, "w"))
    return out


def traffic(path):
    per_kernel = {}
    for k, c in counters_by_kernel(path).items():
        m = {n: v[1] for n, v in c.items()}
        n = max(v[0] for v in c.values())
        rd, rd32 = m.get("TCC_EA0_RDREQ_sum", 0.0), m.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        wr, wr64 = m.get("TCC_EA0_WRREQ_sum", 0.0), m.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        per_kernel[k] = (n, (rd - rd32) * 128 + rd32 * 32, wr64 * 64 + (wr - wr64) * 32)
    out = {}
    for cabi, pats in CABI.items():
        main = [(k, v) for k, v in per_kernel.items() if re.search(pats[0], k)]
        if not main:
            continue
        calls = sum(v[0] for _, v in main) / PER_CALL.get(cabi, 1)
        hit = [(k, v) for k, v in per_kernel.items() if any(re.search(p, k) for p in pats)]
        rd = sum(v[0] * v[1] for _, v in hit) / calls
        wr = sum(v[0] * v[2] for _, v in hit) / calls
        out[cabi] = {"traffic_bytes": int(rd + wr), "read_bytes": int(rd), "write_bytes": int(wr), "launches_sampled": calls}
    return out


def trace_md(rnd, w, rows, bench, cmd):
    lines = [f"# {rnd} — rocprofv3 --kernel-trace --stats, workload `{w}` (MI355X, this round's tree)", "", f"`{cmd}`", ""]
    if bench:
        rf = bench.get("roofline", {})
        lines += [f"bench line of the SAME traced run: `ms_per_step` {bench.get('ms_per_step', bench.get('ms_per_pass'))}, dominant kernel "
                  f"`{rf.get('kernel')}` {rf.get('mean_launch_ms')} ms by HIP events (frac {rf.get('frac')})", ""]
    lines += ["| kernel | calls | total ms | mean us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for n, c, tot, avg, mn, mx, pct in rows[:45]:
        nm = n if len(n) <= 100 else n[:97] + "..."
        lines.append(f"| `{nm}` | {c} | {tot/1e6:.3f} | {avg/1e3:.1f} | {mn/1e3:.1f} | {mx/1e3:.1f} | {pct:.2f} |")
    return "\n".join(lines) + "\n"


def main():
    d, rnd = sys.argv[1], sys.argv[2]
    times, traf = {}, {}
    for w in sorted(os.listdir(d)):
        wd = os.path.join(d, w)
        if not os.path.isdir(wd):
            continue
        bench = None
        if os.path.exists(os.path.join(wd, "bench.json")):
            try:
                bench = json.loads(open(os.path.join(wd, "bench.json")).read().strip().splitlines()[-1])
            except (ValueError, IndexError):
                bench = None
        ks = find(os.path.join(wd, "trace"), "*kernel_stats.csv")
        if ks:
            rows = kernel_stats(ks)
            times[w] = cabi_times(rows)
            if w == "dense_adam":       # (the run also holds the default line's kernels: they are the `deepfm` entry's business)
                times[w] = {k: v for k, v in times[w].items() if k.startswith("lr_adam_dense_rows")}
            cmd = open(os.path.join(wd, "cmd.txt")).read().strip() if os.path.exists(os.path.join(wd, "cmd.txt")) else ""
            with open(os.path.join(d, f"{rnd}_{w}_kernel_trace.md"), "w") as fh:
                fh.write(trace_md(rnd, w, rows, bench, cmd))
            if bench:
                with open(os.path.join(d, f"{rnd}_{w}_bench_traced_run.json"), "w") as fh:
                    json.dump(bench, fh, indent=1)
        cc = find(os.path.join(wd, "pmc"), "*counter_collection.csv") or find(os.path.join(wd, "pmc"), "pmc_kernels.json")
        if cc:
            traf[w] = traffic(cc)
            if w == "dense_adam":
                traf[w] = {k: v for k, v in traf[w].items() if k.startswith("lr_adam_dense_rows")}
    for k_ in ("lr_score_topk_f32", "lr_score_topk_sb_f32", "lr_score_topk_filter_f32"):  # the recommend leg rides in the default (deepfm) command
        if k_ in times.get("deepfm", {}):
            times.setdefault("recommend_100m", {})[k_] = times["deepfm"].pop(k_)
    # (the exact arithmetics share topk_merge_keys_kernel<256>, and the masked exact pass of the filtered form uses it too: with all
    #  in one run each exact entry point's mean holds the merge launches of the others — ~0.1 ms next to 140 / 210 ms)
    meta = {"_comment": f"{rnd}: per C-ABI entry point, mean duration per call from `rocprofv3 --kernel-trace --stats` of the bench "
                        "command of each workload on this round's tree (scripts/profile_round.sh); bench.py prints it beside the "
                        "live HIP-event mean as roofline.profiles_avg_ms / frac_from_profiles"}
    json.dump({**meta, **times}, open(os.path.join(d, f"{rnd}_kernel_times.json"), "w"), indent=1)
    meta = {"_comment": f"{rnd}: fabric bytes per launch (TCC_EA0 request counters, separate PMC pass, eager launches) of each workload's "
                        "kernels on this round's tree; RDREQ x 128 B (32 B for RDREQ_32B) + WRREQ_64B x 64 B + other WRREQ x 32 B; "
                        "Infinity-Cache hits included"}
    json.dump({**meta, **traf}, open(os.path.join(d, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
    for w in times:
        print(w, {k: v["avg_ms"] for k, v in times[w].items()})
    for w in traf:
        print(w, {k: round(v["traffic_bytes"] / 1e9, 3) for k, v in traf[w].items()})


if __name__ == "__main__":
    main()
