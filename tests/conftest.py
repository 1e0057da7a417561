import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


ORACLE_THREADS = 32


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference mounted (build container only)")
    # The fp32 oracle is many small torch ops: on the GPU box's 128 / 256 host cores the default thread count makes them SLOWER by one
    # to two orders of magnitude (measured r5, tools/oracle_threads_probe.py: 30 layers x 3 steps at width 256 take 43.8 s on 128
    # threads and 0.39 s on 16; one 5B-width layer at L = 2912 13.4 s against 3.9 s on 32).  Results do not depend on the count to the
    # tolerance of any test here (fp32 summation order inside a GEMM).
    try:
        import torch
        if torch.get_num_threads() > ORACLE_THREADS:
            torch.set_num_threads(ORACLE_THREADS)
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    from oracle import ref_import
    have_ref = ref_import.reference_available()
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    for item in items:
        if "needs_reference" in item.keywords and not have_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not mounted here"))
        if "gpu" in item.keywords and not have_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU in this container"))


@pytest.fixture(scope="session")
def golden():
    from safetensors.torch import load_file

    def load(name):
        return load_file(os.path.join(GOLDEN, name + ".safetensors"))
    return load
