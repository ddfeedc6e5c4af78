import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def w2e_opt():
    """Setter for the library's process-wide options (w2e_set_option); everything it touched is reset afterwards."""
    from where2edit_amd import _lib
    touched = set()
    env_default = {"conv_precision": os.environ.get("W2E_CONV_PRECISION", "f32"), "deterministic": os.environ.get("W2E_DETERMINISTIC", "0")}

    def set_(name, value):
        touched.add(name)
        _lib.set_option(name, value)

    yield set_
    for name in touched:
        _lib.set_option(name, env_default.get(name, ""))


def pytest_terminal_summary(terminalreporter):
    """The measured end-to-end gradient errors of this run (helpers.assert_grad_close), so that the tolerance it uses can be
    read against what was actually observed (DESIGN.md section 2 quotes this table)."""
    try:
        from helpers import GRAD_ERRORS
    except ImportError:
        return
    if GRAD_ERRORS:
        terminalreporter.write_sep("-", "end-to-end gradient parity: max-norm relative error, cosine, tolerance held to")
        for what, e, cos, *tol in GRAD_ERRORS:
            held = f"(<= {tol[0]:.0e})" if tol and tol[0] is not None else "          "
            terminalreporter.write_line(f"{e:10.3e}  {cos:.7f}  {held}  {what}")
