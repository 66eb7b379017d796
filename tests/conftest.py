import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# The slowest bf16 DUPLICATES of end-to-end GPU tests (same code paths as their fp16 twins, asserted at 8x the fp16 tolerance; bf16 is selectable, not the
# default -- it stays covered by every kernel / block test, the full-size wrapper and decoder tests and the product-size kernel tests): on demand only
# (SVD_TEST_BF16_SLOW=1), to keep the -m gpu suite well inside the driver's time limit (round-4 review: 708 s of 1 200 s).
_BF16_ON_DEMAND = ("test_gpu_fullsize_parity.py::test_shipped_architecture_small_latent_vs_reference[bf16]",
                   "test_gpu_ar_parity.py::test_initial_chunk_vs_oracle[bf16]", "test_gpu_ar_parity.py::test_autoregressive_chunks_vs_oracle[bf16]",
                   "test_gpu_ar_parity.py::test_video_vs_reference_fp32_within_the_reference_autocast_envelope[bf16", "test_gpu_ar_parity.py::test_video_vs_reference_fp32_envelope_other_input_seeds[bf16",
                   "test_gpu_parity.py::test_config1_end_to_end_vs_oracle[bf16]", "test_gpu_parity.py::test_native_conditioner_and_front_end_single_chunk[bf16]")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("SVD_TEST_BF16_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow bf16 duplicate of an fp16 end-to-end test: SVD_TEST_BF16_SLOW=1 runs it")
    for it in items:
        if "bf16" in it.nodeid and any(k in it.nodeid for k in _BF16_ON_DEMAND):
            it.add_marker(skip)
