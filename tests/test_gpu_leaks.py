"""Device memory is returned: many Sampler lifecycles over every model / proposal family, histories,
device reductions, tempering and adaptation leave the free device memory where it was
(tools/leak_check.py)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_no_device_memory_leak_over_sampler_lifecycles(ra):
    spec = importlib.util.spec_from_file_location(
        "leak_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "leak_check.py"))
    lc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lc)
    assert lc.main(4) == 0
