"""Checksum manifest of the TensorFlow-made fixtures (`tests/golden/tf_*.npz`).  TEST INFRASTRUCTURE ONLY.

`python -m oracle.tf_manifest --write` (called by scripts/pin_tf_half.sh after `python -m oracle.make_tf_golden`) records the
sha256, size and TensorFlow version of every fixture in `tests/golden/TF_MANIFEST.json`; `tests/test_tf_golden_cpu.py` verifies
the committed files against it, so that a reviewer sees in ONE diff which bytes pin the TF half of the oracle and which
TensorFlow made them.  With no fixture present the manifest says `"status": "unpinned"` and names the files the generator writes."""
from __future__ import annotations

import hashlib
import json
import sys
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent.parent / "tests" / "golden"
MANIFEST = GOLDEN / "TF_MANIFEST.json"
MODELS = ("FM", "DeepFM", "DIN", "TwoTower")
EXPECTED = [f"tf_{m.lower()}.npz" for m in MODELS] + [f"tf_{m.lower()}_tf_variables.npz" for m in MODELS]


def sha256(path: Path) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def scan() -> dict:
    files = {}
    for p in sorted(GOLDEN.glob("tf_*.npz")):
        ent = {"sha256": sha256(p), "bytes": p.stat().st_size}
        try:
            with np.load(p, allow_pickle=False) as z:
                if "meta" in z.files:
                    meta = json.loads(str(z["meta"]))
                    ent["tensorflow"] = meta.get("tf_version")
                    ent["model"] = meta.get("model")
        except Exception:  # noqa: BLE001  (a *_tf_variables.npz holds arrays only)
            pass
        files[p.name] = ent
    return {"status": "pinned" if any(f"tf_{m.lower()}.npz" in files for m in MODELS) else "unpinned",
            "made_by": "scripts/pin_tf_half.sh (python -m oracle.make_tf_golden; tensorflow>=1.15,<2.16, reference requirements.txt:5)",
            "expected_files": EXPECTED, "files": files}


def main() -> int:
    cur = scan()
    if "--write" in sys.argv:
        MANIFEST.write_text(json.dumps(cur, indent=1, sort_keys=True) + "\n")
        print(f"wrote {MANIFEST} ({cur['status']}, {len(cur['files'])} file(s))")
    for name, ent in cur["files"].items():
        print(f"{ent['sha256']}  {name}  ({ent['bytes']} bytes, TensorFlow {ent.get('tensorflow')})")
    if not cur["files"]:
        print("no tests/golden/tf_*.npz present: the TF half of the oracle is UNPINNED")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
