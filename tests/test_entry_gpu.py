"""The driver's round-end check, kept from rotting: __graft_entry__.smoke() is one small invocation of the hot path on
cuda:0 against the oracle (it broke silently once, when the host entries' default layout changed)."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_smoke_entry_runs():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as g
    g.smoke()
