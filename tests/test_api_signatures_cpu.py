"""Drop-in seam: every constructor / method / function signature the reference declares for the models on and
next to the hot path, its data layer and its scoring functions (`tests/golden/api_signatures.json`, written by
`oracle/make_golden.py:gen_api_signatures` from the imported reference) is accepted by this package: the same parameter
names in the same positional order with the same defaults.  The package may add keyword parameters BEHIND the
reference's (device-side options) and may forward a tail through `*args, **kwargs`."""
import importlib
import inspect
import json
from pathlib import Path

import pytest

GOLDEN = json.loads((Path(__file__).parent / "golden" / "api_signatures.json").read_text())

# Known, documented gaps: the reference method exists, this package has none.
KNOWN_MISSING = set()          # round 4: `YouTubeRetrieval.rebuild_model` exists (tests/test_retrain_gpu.py)


def resolve(key):
    parts = key.split(".")
    obj = importlib.import_module("librecommender_amd." + parts[0])
    for p in parts[1:]:
        obj = getattr(obj, p)
    return obj


def accepts(ref_sig, fn):
    """None if `fn` can be called the way the reference's signature allows, else a description of the mismatch."""
    mine = list(inspect.signature(fn).parameters.values())
    names = [p.name for p in mine]
    var_pos = any(p.kind is p.VAR_POSITIONAL for p in mine)
    var_kw = any(p.kind is p.VAR_KEYWORD for p in mine)
    named = [p for p in mine if p.kind in (p.POSITIONAL_OR_KEYWORD, p.KEYWORD_ONLY)]
    pos = 0
    for name, kind, default in ref_sig:
        if kind in ("VAR_POSITIONAL", "VAR_KEYWORD"):
            continue                       # the reference's own catch-alls: nothing a caller can rely on
        if name in names:
            p = mine[names.index(name)]
            if kind == "POSITIONAL_OR_KEYWORD" and p.kind is p.POSITIONAL_OR_KEYWORD:
                if pos >= len(named) or named[pos].name != name:
                    return f"`{name}` is not at positional slot {pos} (package order: {[q.name for q in named]})"
            got = None if p.default is inspect.Parameter.empty else repr(p.default)
            if got != default:
                return f"default of `{name}`: reference {default}, package {got}"
            pos += 1
        elif not (var_kw and (var_pos or kind == "KEYWORD_ONLY")):
            return f"`{name}` is not accepted"
        else:
            pos += 1                       # forwarded through *args / **kwargs
    return None


@pytest.mark.parametrize("key", sorted(GOLDEN))
def test_reference_signature_is_accepted(key):
    if key in KNOWN_MISSING:
        with pytest.raises(AttributeError):
            resolve(key)
        pytest.skip("documented gap")
    fn = resolve(key)
    problem = accepts(GOLDEN[key], fn)
    assert problem is None, f"{key}: {problem}"
