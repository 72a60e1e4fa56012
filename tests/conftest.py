import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the native libraries exist (builds them if a toolchain is here)."""
    import __graft_entry__ as g
    g.build(quiet=True)
    return True


@pytest.fixture(scope="session")
def xmls(tmp_path_factory):
    """Problem-definition XML files, generated for this session (tests/golden/gen_xml_fixtures.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_xml_fixtures
    return gen_xml_fixtures.write_all(str(tmp_path_factory.mktemp("etol_xml")))
