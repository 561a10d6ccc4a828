import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A per-test wall-clock cap (pytest-timeout, where installed): a persistent kernel that waits for a hand-off that never comes must end the
    run with a stack dump, not hold the GPU box until somebody else's limit kills it.  The slowest test takes ~20 s."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for it in items:
        if it.get_closest_marker("timeout") is None:
            it.add_marker(pytest.mark.timeout(900, method="thread"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ---------------------------------------------------------------------------------------------------------
# Process state harness.  The C ABI keeps no state between calls, but the option table behind
# ctcn_set_option and the module-level state of ctc_pytorch_amd.ops / .parallel are process-wide: a test that leaks a switch, a precision
# or a learnt batch-chunk mark changes what every later test of the session computes.  Every GPU test therefore
#   * starts from the library defaults for the part that tests legitimately move without restoring it (matmul precision) and from the
#     learnt state its predecessor found (shapes marked for batch chunks, dropout stream position are put back after it);
#   * is held to leaving the configuration part exactly as it found it (options, flags, hooks, batch split): a leak fails the test
#     that made it, with the difference, after the harness has put the state back so that the leak does not travel;
#   * gets the snapshot taken at its start and at its end attached to its report when it fails (a trajectory mismatch then shows the
#     state it ran in).
# CTCN_STATE_LOG=<file>: one JSON line per test (id, snapshot at start, what it moved) -- the record a session's divergence is read from.
# ---------------------------------------------------------------------------------------------------------
def _config_part(snap):
    return {k: v for k, v in snap.items() if k not in ("fallback_shapes", "drop_counter", "precision")}


def _diff(a, b, prefix=""):
    out = []
    for k in sorted(set(a) | set(b)):
        va, vb = a.get(k), b.get(k)
        if isinstance(va, dict) and isinstance(vb, dict):
            out += _diff(va, vb, prefix + k + ".")
        elif va != vb:
            out.append("%s%s: %r -> %r" % (prefix, k, va, vb))
    return out


@pytest.fixture(autouse=True)
def process_state_left_as_found(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import json
    from ctc_pytorch_amd import ops
    ops.set_precision(ops.DEFAULT_PRECISION)
    before = ops.state_snapshot()
    request.node._ctcn_state = {"start": before}
    yield
    after = ops.state_snapshot()
    request.node._ctcn_state["end"] = after
    leaks = _diff(_config_part(before), _config_part(after))
    moved = _diff({k: before[k] for k in ("fallback_shapes", "drop_counter", "precision")},
                  {k: after[k] for k in ("fallback_shapes", "drop_counter", "precision")})
    ops.restore_state(before)
    log = os.environ.get("CTCN_STATE_LOG")
    if log:
        with open(log, "a") as f:
            f.write(json.dumps({"test": request.node.nodeid, "start": before, "moved": moved, "leaked": leaks}) + "\n")
    assert not leaks, "test left process-wide state changed (put back by the harness): " + "; ".join(leaks)


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    st = getattr(item, "_ctcn_state", None)
    if st and rep.failed:
        import json
        if "end" not in st:
            try:
                from ctc_pytorch_amd import ops
                st["end"] = ops.state_snapshot()
            except Exception as e:          # noqa: BLE001 -- the report must not mask the failure it annotates
                st["end"] = {"error": repr(e)}
        rep.sections.append(("process state at the start of the test", json.dumps(st["start"], sort_keys=True)))
        rep.sections.append(("process state moved during the test", "; ".join(_diff(st["start"], st["end"])) or "nothing"))


def pytest_sessionfinish(session, exitstatus):
    """CTCN_AFTER_SUITE=<script.py>: run a script IN this process once the suite is over (tools/after_suite_ab.py: the projection-order A/B in the
    long-lived process the cfg4 divergence needed).  Development aid; unset in every normal run."""
    path = os.environ.get("CTCN_AFTER_SUITE")
    if path:
        import runpy
        try:
            runpy.run_path(path, run_name="__main__")
        except SystemExit:
            pass
