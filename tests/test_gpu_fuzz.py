"""A fixed slice of the randomised differential test (tools/fuzz_vs_oracle.py): random
model x proposal x size x run-schedule configurations on the device against the CPU oracle.
The tool itself has been run over several thousand configurations."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_random_configurations_vs_oracle():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_vs_oracle.py")
    spec = importlib.util.spec_from_file_location("fuzz_vs_oracle", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    from riemann_amd._lib import kernel_policy
    g, gp = np.random.default_rng(2026), np.random.default_rng(3003)
    seen = set()
    for it in range(140):
        with kernel_policy(*fz.random_policy(gp)):     # half on the library's own kernel choice, half on A/B switches
            desc = fz.one(g, it)
        seen.add((desc.split()[0], desc.split("|")[0].split()[-1]))
    assert {k for k, _ in seen} == {"gauss", "logreg", "mix", "cp", "bigd", "cprj"}    # every model family came up
