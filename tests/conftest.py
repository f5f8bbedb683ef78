import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The stream -> hardware-queue mapping bench.py runs under (it sets the same before the HIP runtime starts; subprocess
# workers inherit it): the suite exercises the schedules with the queue setting the benchmark is measured with.
# SSA_TEST_HW_QUEUES=<n> runs the suite under another setting (the HIP default is 4).
os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("SSA_TEST_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES", "8"))

# SSA_POISON=nan|big: every uninitialised device buffer the host layer allocates comes back filled (tools/poison.py):
# a kernel that reads memory this run never wrote then shows, instead of finding the previous run's values there.
if os.environ.get("SSA_POISON"):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import poison

    poison.install()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


# Collection order of the GPU suite (the driver runs it with -x: whatever fails first hides everything behind it).
# Golden / oracle comparisons first, cheapest first; the full-size property tests (bit-identity, residuals at
# BASELINE sizes, minutes of GPU time) last:
#   0  kernels against goldens / numpy                       tests/test_kernels_gpu.py
#   1  solve() against the reference fixtures (config 1 = disk_K26 is here) and the API built on it   tests/test_solve_gpu.py
#   2  oracle comparisons at headline sizes (configs 3/H, config 4's 64-field sweep, the 4-film stack)
#   3  full-size properties (configs 2, 3, H, 5)
_FULL_SIZE_PROPERTIES = ("test_full_size_london_system", "test_full_size_float32_and_lu_routes_agree_with_float64",
                         "test_system_assemble_sampled_rows_at_full_size",
                         "test_cold_factorizations_bit_identical_beside_other_work")


# BASELINE config 1 (the 2 107-vertex disk) and config 4 (64 applied fields) are cheap: they go first of all
_FIRST = ("test_single_film_vs_reference_fixture", "test_solve_sweep_64_fields_vs_oracle")


def _gpu_rank(item) -> int:
    path = item.nodeid.split("::")[0]
    if item.name.split("[")[0] in _FIRST:
        return -2 + _FIRST.index(item.name.split("[")[0])
    if path.endswith("test_kernels_gpu.py"):
        return 0
    if path.endswith("test_solve_gpu.py"):
        return 1
    if path.endswith("test_headline_gpu.py"):
        return 3 if item.name.split("[")[0] in _FULL_SIZE_PROPERTIES else 2
    return 0


def pytest_collection_modifyitems(config, items):
    items.sort(key=_gpu_rank)   # (stable: the order inside a group is the files' own)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
        return cache[name]

    return load
