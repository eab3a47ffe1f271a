import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def fixture_tsv(golden, tmp_path_factory):
    """taxons.tsv of the reference's 6-taxon fixture (src/fixtures.rs:4-21)."""
    p = tmp_path_factory.mktemp("tax") / "fixture.tsv"
    with open(p, "wb") as f:
        for tid, name, rank, parent, valid in golden["fixture_taxonomy"]["rows"]:
            f.write(("%d\t%s\t%s\t%d\t" % (tid, name, rank, parent)).encode() + (b"\x01" if valid else b"\x00") + b"\n")
    return str(p)


@pytest.fixture(scope="session")
def large_tsv(golden, tmp_path_factory):
    """15-node tree of src/rmq/lca.rs:131-153."""
    p = tmp_path_factory.mktemp("tax") / "large.tsv"
    with open(p, "wb") as f:
        for tid, parent in golden["large_taxonomy"]["rows"]:
            f.write(("%d\t\tno rank\t%d\t" % (tid, parent)).encode() + b"\x01\n")
    return str(p)
