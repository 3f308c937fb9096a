"""Stand-in for `ujson` (absent here) so the reference's libserving serializers can run in the
build container when generating fixtures.  TEST INFRASTRUCTURE."""
import json


def dump(obj, fp, ensure_ascii=True, **_):
    json.dump(obj, fp, ensure_ascii=ensure_ascii)


def dumps(obj, ensure_ascii=True, **_):
    return json.dumps(obj, ensure_ascii=ensure_ascii)


load, loads = json.load, json.loads
