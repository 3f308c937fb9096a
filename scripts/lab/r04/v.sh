#!/bin/bash
# r04v: score_topk with the loose lockstep of an item range's workgroups: parity tests, time at 100 M items, fabric traffic (PMC)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
out=gpurun_out/r04v
mkdir -p "$out"
timeout 900 python -m pytest tests/test_score_topk_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -3 "$out/pytest.log"
timeout 600 python scripts/score_topk_traffic.py 2>&1 | tail -3
bash scripts/pmc_cmd.sh r04topk "python scripts/score_topk_traffic.py --once" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" > "$out/pmc.log" 2>&1
grep -E "score_topk|merge" "$out/pmc.log" | cut -c1-400
