import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size GPT-2 on CPU (tens of seconds)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(autouse=True, scope="session")
def _memoize_synthetic_gpt2():
    """Dozens of tests rebuild the same seeded GPT-2-small state dict (124 M normal draws, ~2 s each; the forced-geometry
    child processes do it again): generate each (seed, geometry) once per process and hand out CLONES, so a test that edits
    its weights in place cannot leak into another."""
    from collections import OrderedDict
    from capdec_amd import synth
    orig = synth.hot_gpt2_state_dict
    cache = {}

    def cached(seed=42, dims=synth.GPT2_SMALL, prefix="gpt."):
        key = (seed, dims.n_layer, dims.n_head, dims.n_embd, dims.vocab, dims.n_pos, prefix)
        if key not in cache:
            cache[key] = orig(seed, dims, prefix)
        clones, out = {}, OrderedDict()
        for k, v in cache[key].items():              # (the tied lm_head stays ONE tensor)
            if id(v) not in clones:
                clones[id(v)] = v.clone()
            out[k] = clones[id(v)]
        return out

    synth.hot_gpt2_state_dict = cached
    yield
    synth.hot_gpt2_state_dict = orig
