"""Mid-scale end-to-end goldens (tests/golden/midscale.json, made by tests/golden/make_golden_midscale.py from complete --t 1 runs
of the compiled reference): the inputs regenerate from the counter-based generator, the expected outputs are digests."""
import hashlib
import importlib.util
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_golden_midscale", os.path.join(HERE, "golden", "make_golden_midscale.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


def cases() -> dict:
    p = os.path.join(HERE, "golden", "midscale.json")
    return json.load(open(p))["cases"] if os.path.exists(p) else {}


def digest_bytes(b: bytes) -> dict:
    return {"sha256": hashlib.sha256(b).hexdigest(), "bytes": len(b), "lines": b.count(b"\n")}


def digest_file(path: str) -> dict:
    return gen.digest(path)
