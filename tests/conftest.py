import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def renderer_factory():
    """Factory of HIP renderers; fails loudly (no fallback) when the library or GPU is missing."""
    from nori_amd.render import Renderer

    made = []

    def make(scene, **kw):
        r = Renderer(0).upload(scene, **kw)
        made.append(r)
        return r

    yield make
    for r in made:
        r.close()
