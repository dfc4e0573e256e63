import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """VITK_TEST_ORDER=reverse / a number: run the collected tests backwards / shuffled with that seed (tests must not
    depend on what ran before them: cached scratch buffers, debug variants, the weight cache)."""
    order = os.environ.get("VITK_TEST_ORDER", "")
    if order == "reverse":
        items.reverse()
    elif order:
        import random
        random.Random(int(order)).shuffle(items)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
