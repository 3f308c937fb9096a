#!/bin/bash
# Pin the TensorFlow half of the oracle (FM / DeepFM / DIN / TwoTower numerics are "parity unpinned" until the
# reference's own TF graphs have produced golden vectors).  ONE command on any box with Python 3.8-3.11, network access for pip
# and a checkout of the reference:
#
#     bash scripts/pin_tf_half.sh /path/to/LibRecommender
#
# It (1) makes a throw-away venv with the TensorFlow range the reference pins (requirements.txt:5, tfops/version.py:1-4),
# (2) runs the reference's graphs for one deterministic train step per model (oracle/make_tf_golden.py) -> tests/golden/tf_*.npz,
# (3) writes tests/golden/TF_MANIFEST.json (sha256 + TensorFlow version of every fixture), (4) runs the consumer test, which from
# then on PINS oracle/models_torch.py against TensorFlow instead of skipping.  Commit what step (5) prints.
set -euo pipefail
REF=${1:-${LIBRECO_REFERENCE:-}}
[ -n "$REF" ] && [ -d "$REF/libreco" ] || { echo "usage: $0 /path/to/LibRecommender   (a checkout holding libreco/)"; exit 2; }
ROOT=$(cd "$(dirname "$0")/.." && pwd)
VENV=${PIN_TF_VENV:-$ROOT/build/tf_venv}
PY=${PYTHON:-python3}
if [ ! -x "$VENV/bin/python" ]; then
  "$PY" -m venv "$VENV"
  "$VENV/bin/pip" install --upgrade pip
  # the reference's own pins (requirements.txt:1-12); torch (CPU) is what the consumer test runs the oracle on
  "$VENV/bin/pip" install "tensorflow==2.12.*" "numpy>=1.19.5,<2" "scipy>=1.2.1,<1.13" "pandas>=1.0" "scikit-learn>=0.20" \
      "gensim>=4.0" "tqdm" "torch>=1.10" "pytest"
fi
cd "$ROOT"
export LIBRECO_REFERENCE="$REF" PYTHONDONTWRITEBYTECODE=1 TF_CPP_MIN_LOG_LEVEL=2 TF_DETERMINISTIC_OPS=1 CUDA_VISIBLE_DEVICES=""
echo "== (2) one deterministic train step per model through the reference's TensorFlow graphs"
"$VENV/bin/python" -m oracle.make_tf_golden "${@:2}"
echo "== (3) manifest"
"$VENV/bin/python" -m oracle.tf_manifest --write
echo "== (4) the oracle restatement against the fixtures"
"$VENV/bin/python" -m pytest tests/test_tf_golden_cpu.py -q -rs
echo "== (5) commit these:"
echo "    git add tests/golden/tf_*.npz tests/golden/TF_MANIFEST.json && git commit -m 'tests/golden: TensorFlow-made fixtures (parity pinned for the TF half)'"
