import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_REPORT_LINES = []


@pytest.fixture
def report(request):
    """Numeric evidence of a parity test, three ways: appended to gpurun_out/parity_report.txt (merged back by gpurun),
    attached to the test as a user property (JUnit XML / any reporter: `record_property`), and printed in the terminal
    summary of the run, so that whoever runs `pytest -m gpu` -- not only the builder -- sees the numbers."""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    name = request.node.name

    def write(line):
        with open(os.path.join(d, "parity_report.txt"), "a") as f:
            f.write(line + "\n")
        request.node.user_properties.append(("parity", line))
        _REPORT_LINES.append((name, line))
    return write


def pytest_terminal_summary(terminalreporter):
    if not _REPORT_LINES:
        return
    terminalreporter.section("parity report (full lines: gpurun_out/parity_report.txt)")
    for name, line in _REPORT_LINES:
        terminalreporter.write_line(f"[{name}] {line[:400]}")
