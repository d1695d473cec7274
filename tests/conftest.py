import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from zl_testlib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from zl_testlib import Ref, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref/libzstd_ref.so not built (make -C oracle ref needs /root/reference)")
    return Ref()


@pytest.fixture(scope="session")
def golden():
    from zl_testlib import load_golden
    return load_golden()


@pytest.fixture(scope="session")
def golden_large():
    from zl_testlib import load_golden_large
    return load_golden_large()
