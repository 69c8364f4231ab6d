import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "expects_chain_timeout: the test provokes a persistent-kernel timeout on purpose")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _library_is_there():
    """A fresh checkout has no libinpaintnet_hip.so (built artefacts are not tracked) and the host-side tests that spawn ranks load it
    in child processes (tests/test_dp_gloo.py sorts in front of the test that used to build it): build it once per session when it is
    MISSING -- hipcc cross-compiles without a GPU; an existing library is left alone (the driver's build() check owns staleness)."""
    from inpaintnet_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build(verbose=False)
    yield


def poison_allocator_pool(total_mb=1536):
    """Fill the caching allocator's free blocks with NaN: what a later `torch.empty` hands out is then poisoned, and a kernel that
    reads a buffer (or a workspace region) before anything wrote it shows up as NaN instead of passing on the zeros a fresh
    process happens to get.  Blocks of many sizes, so both of the allocator's pools and every split of a large block are covered."""
    import torch
    blocks = []
    nan = float("nan")
    for mb, count in ((64, max(1, total_mb // 128)), (16, max(1, total_mb // 64)), (1, 256)):
        for _ in range(count):
            blocks.append(torch.full((mb << 18,), nan, dtype=torch.float32, device="cuda"))
    for kb, count in ((256, 256), (16, 1024), (1, 2048)):
        for _ in range(count):
            blocks.append(torch.full((kb << 8,), nan, dtype=torch.float32, device="cuda"))
    torch.cuda.synchronize()
    del blocks


@pytest.fixture(autouse=True)
def _poisoned_pool(request):
    """INET_TEST_POISON=1: every GPU test starts with a NaN-filled allocator pool (tools/README: the uninitialised-read sweep)."""
    if os.environ.get("INET_TEST_POISON") == "1" and request.node.get_closest_marker("gpu"):
        poison_allocator_pool()
    yield


@pytest.fixture(autouse=True)
def _fresh_dropout_stream(request):
    """Every GPU test starts from the package's initial dropout stream (seed, call counter): a test that seeds the stream
    (set_dropout_seed) or draws masks must not change which masks a later test sees.  (Found the hard way: behind
    test_decoder_multinomial_sampling the AnticipationRNN step drew a mask under which one pre-activation of linear_1 lands within
    round-off of the ReLU kink -- product and oracle take different branches there and one row of one gradient tensor moves by
    4 % -- so the suite passed in file order and failed when test_gpu_inference.py ran first.)"""
    if request.node.get_closest_marker("gpu"):
        from inpaintnet_amd import measure_vae as MV
        MV._DropState.seed = 0x5eed
        MV._mask_counter[0] = 0
    yield


@pytest.fixture(autouse=True)
def _clean_chain_status(request):
    """A persistent kernel that gave up (bounded spin) leaves a sticky status word: every later test that asserts `chain_status() ==
    0` would fail with it, and the one log line that matters -- WHICH test left it -- would drown (one of ~20 runs of the suite in
    round 5 ended with 24 failures behind a single event that was never identified).  Every GPU test starts from a clean word; a
    test that leaves it dirty is named on stdout."""
    gpu = request.node.get_closest_marker("gpu") is not None
    if gpu:
        from inpaintnet_amd import ops
        ops.chain_status(reset=True)
    yield
    if gpu:
        from inpaintnet_amd import ops
        left = ops.chain_status()
        if left:
            ops.chain_status(reset=True)
            if request.node.get_closest_marker("expects_chain_timeout") is None:
                # the fallbacks recompute correct results behind a timeout, so a test that does not look at the word itself would
                # pass over a lost hand-off: the culprit fails HERE, with what the library's recorder saw (ops.slow_waits)
                seen = ops.slow_waits(reset=True) if hasattr(ops, "slow_waits") else None
                pytest.fail(f"chain status {left} left behind by {request.node.nodeid}; recorder: {seen}", pytrace=False)
