import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: heavy CPU case, skipped unless BAGEL_SLOW=1")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("BAGEL_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="heavy CPU case; set BAGEL_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)
