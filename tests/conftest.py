import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _release_device_objects(request):
    """GPU tests build engines with multi-GB workspaces, hipGraphs and CU-masked streams; objects a test leaves to the cyclic garbage collector
    (pipelines that own their strategies' engines) would otherwise be finalised -- device synchronize + sr_engine_destroy -- at an arbitrary
    allocation inside a LATER test's launch sequence.  Finalise them here, between tests, with the device idle."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import gc

        import torch
        gc.collect()
        if torch.cuda.is_available():
            torch.cuda.synchronize()


@pytest.fixture(autouse=True)
def _library_switches_follow_the_environment():
    """libsocior.so reads its SR_* switches once (and at every engine creation), not per call (round 5).  A test that flips one with
    ``switch(monkeypatch, name, value)`` has the library re-read it; this fixture is set up before ``monkeypatch`` and therefore torn down after
    it has restored the environment, so the next test starts from the shipped defaults again."""
    yield
    from socioreasoner_amd import lib
    if lib._lib is not None:
        lib.reload_switches()

