import json
import os
import subprocess
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


class RankLauncher:
    """front end of tests/rank_launcher.py (a helper process started before this one touches the GPU)"""

    def __init__(self):
        self.proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rank_launcher.py")], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                     text=True, bufsize=1)

    def run(self, argv, world, env=None, timeout=600):
        self.proc.stdin.write(json.dumps({"argv": list(argv), "world": world, "env": env or {}, "timeout": timeout}) + "\n")
        self.proc.stdin.flush()
        return json.loads(self.proc.stdout.readline())["rc"]

    def close(self):
        try:
            self.proc.stdin.close()
            self.proc.wait(timeout=30)
        except Exception:
            self.proc.kill()


_launcher = None


def pytest_sessionstart(session):
    """where GPUs are visible, start the rank launcher before any test initialises the device in this process (counting devices does not)"""
    global _launcher
    try:
        import torch
        if torch.cuda.device_count() > 0:
            _launcher = RankLauncher()
    except Exception:
        _launcher = None


def pytest_sessionfinish(session, exitstatus):
    if _launcher is not None:
        _launcher.close()


@pytest.fixture(scope="session")
def rank_launcher():
    if _launcher is None:
        pytest.skip("no GPU visible: the rank launcher was not started")
    return _launcher
