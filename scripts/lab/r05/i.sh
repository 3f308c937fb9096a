#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
timeout 120 python scripts/lab/r05/tail_marks.py 2>&1 | tail -40
