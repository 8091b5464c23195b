import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json

    with open(os.path.join(GOLDEN, "expected.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def csv_tables():
    from naive_query_engine_amd import read_csv

    return {name: read_csv(os.path.join(GOLDEN, f"{name}.csv")) for name in ("test_data", "employee", "rank", "department")}
