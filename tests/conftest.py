import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("av1-base_amd", "oracle", "tools"):
    sys.path.insert(0, os.path.join(ROOT, p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _watchdog():
    """A test that stops making progress (a deadlocked worker thread, a kernel that never drains) ends the run with every
    thread's stack on stderr after 5 minutes instead of hanging the box until the runner's own limit."""
    import faulthandler
    faulthandler.dump_traceback_later(300, exit=True)
    yield
    faulthandler.cancel_dump_traceback_later()


@pytest.fixture(scope="session")
def oracle():
    import av1o
    av1o.build()
    return av1o


@pytest.fixture(scope="session")
def golden_cases():
    import json
    gdir = os.path.join(ROOT, "tests", "golden")
    out = []
    for name in json.load(open(os.path.join(gdir, "index.json"))):
        meta = json.load(open(os.path.join(gdir, name + ".json")))
        meta["obu"] = open(os.path.join(gdir, name + ".obu"), "rb").read()
        out.append(meta)
    return out


@pytest.fixture(scope="session")
def golden_sequences():
    """inter-coded sequences: concatenated temporal units + per-frame sizes and dav1d hashes"""
    import json
    gdir = os.path.join(ROOT, "tests", "golden")
    out = []
    for name in json.load(open(os.path.join(gdir, "index_seq.json"))):
        meta = json.load(open(os.path.join(gdir, name + ".json")))
        meta["obu"] = open(os.path.join(gdir, name + ".obu"), "rb").read()
        out.append(meta)
    return out


@pytest.fixture(scope="session")
def av1mi():
    """the product's Python host mirror over the C ABI (loads av1-base_amd/libav1mi.so)"""
    import av1mi as m
    return m


@pytest.fixture(scope="session")
def ctx(av1mi):
    c = av1mi.Context(0)
    yield c
    c.close()
