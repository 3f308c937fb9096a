"""cfg 5's graph (bench_workloads.distinct_interactions, seed 42): where do the nonzeros' COLUMNS fall?
(1) by id — the synthetic ids are Zipf ranks, id 0 is the most popular endpoint;  (2) by degree rank — what a degree-sorted
permutation would make of them.  For cache sizes c (rows): the share of nonzeros whose column is among the c hottest rows of its
side = the best hit rate a cache of c rows can have on the gathered operand (pinned hot set), and the fabric bytes per product that
leaves: col/val stream + Y once + the misses x 256 B."""
import json
import sys

import torch

sys.path.insert(0, ".")
import bench_workloads as bw  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(bw.LG_CFG)
nu, ni, E, K = cfg["n_users"], cfg["n_items"], cfg["n_edges"], cfg["embed_size"]
g = torch.Generator(device=dev).manual_seed(42)
eu, ei = bw.distinct_interactions(E, nu, ni, g, dev)
out = {"n_users": nu, "n_items": ni, "nnz": 2 * E, "K": K}
sizes = [256, 1024, 4096, 16384, 65536, 262144, 1 << 20, 4 << 20]
for side, ids, n in (("item columns (user rows)", ei, ni), ("user columns (item rows)", eu, nu)):
    deg = torch.bincount(ids.long(), minlength=n)
    by_id = torch.cumsum(deg, 0).double() / E
    by_deg = torch.cumsum(torch.sort(deg, descending=True).values, 0).double() / E
    out[side] = {"share_of_nonzeros_in_first_c_ids": {str(c): round(float(by_id[min(c, n) - 1]), 4) for c in sizes},
                 "share_of_nonzeros_in_c_highest_degree_rows": {str(c): round(float(by_deg[min(c, n) - 1]), 4) for c in sizes},
                 "max_degree": int(deg.max()), "rows_with_degree_0": int((deg == 0).sum()),
                 "degree_quantiles": {q: int(torch.quantile(deg.float()[:: max(n // 1_000_000, 1)], float(q))) for q in ("0.5", "0.9", "0.99")}}
# fabric bytes of one product if a cache pinned the c hottest rows of each side (both sides alternate: rows of one side read the other)
rowbytes = K * 4
stream = 2 * E * 8 + (nu + ni) * rowbytes + (nu + ni + 1) * 8
ideal = {}
for c in sizes:
    hit = 0.5 * (out["item columns (user rows)"]["share_of_nonzeros_in_c_highest_degree_rows"][str(c)]
                 + out["user columns (item rows)"]["share_of_nonzeros_in_c_highest_degree_rows"][str(c)])
    ideal[str(c)] = {"hit_share": round(hit, 4), "bytes_per_product_GB": round((stream + (1 - hit) * 2 * E * rowbytes) / 1e9, 1)}
out["pinned_hot_set_model"] = {"stream_bytes_GB": round(stream / 1e9, 2), "by_cache_rows": ideal,
                               "note": "cache of c rows per side holding exactly the c highest-degree rows; L2 = 4 MB per XCD = 16384 rows of 256 B, "
                                       "Infinity Cache 256 MB = 1 M rows"}
# (3) the row-partitioned net (nets/graph_nets.py:ShardedLightGCNNet): how many DISTINCT columns does one rank's row slice reference?
# = the rows a "needed rows only" exchange would have to fetch, against the all-gather's (W - 1) / W of every row
n = nu + ni
need = {}
for W in (2, 4, 8):
    per = -(-n // W)
    shares = []
    for r in (0, W // 2, W - 1):                     # first (hub users), middle (first item rows), last slice
        lo, hi = r * per, min(n, (r + 1) * per)
        # rows [lo, hi) of the symmetric matrix: user rows reference item columns, item rows user columns
        cols = []
        if lo < nu:
            m = (eu >= lo) & (eu < min(hi, nu))
            cols.append(torch.unique(ei[m]).numel())
        if hi > nu:
            m = (ei >= max(lo, nu) - nu) & (ei < hi - nu)
            cols.append(torch.unique(eu[m]).numel())
        remote_rows_all_gather = n - (hi - lo)
        shares.append({"rank": r, "distinct_columns": int(sum(cols)), "share_of_all_rows": round(sum(cols) / n, 4),
                       "all_gather_receives_rows": int(remote_rows_all_gather)})
    need[str(W)] = shares
out["needed_rows_per_rank_slice"] = need
print(json.dumps(out, indent=1))
