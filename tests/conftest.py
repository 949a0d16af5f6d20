import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle, build
    build(ref=True)
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The unmodified reference library (present in the authoring container and, as a prebuilt
    .so under oracle/_ref/, on the GPU box)."""
    from oracle.oracle import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libhuffman_ref.so not built")
    return Reference()
