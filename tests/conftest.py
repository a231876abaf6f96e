import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the product library, the CLI and the oracle are build artefacts (git-ignored): build them if
    # this checkout has not been built yet (hipcc cross-compiles gfx950 without a GPU)
    import subprocess

    need = [os.path.join(ROOT, "famseq_amd", "lib", "libfamseq_hip.so"), os.path.join(ROOT, "bin", "FamSeq"),
            os.path.join(ROOT, "oracle", "liboracle_bn.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-s", "-j", "8", "-C", ROOT, "all"])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"])


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
