import faulthandler
import os
import signal
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# per-test wall-clock limit: a stuck native call must cost one test, not the whole suite
GPU_TEST_TIMEOUT_S = int(os.environ.get("PS_AMD_TEST_TIMEOUT", "180"))

# collection order of the -m gpu suite: the oracle-parity tests of the hot path first, the
# threads-on-one-GPU exchange tests last
_ORDER = ["test_gpu_operators", "test_gpu_parity", "test_gpu_configs", "test_gpu_layer_ops", "test_gpu_sumorder", "test_gpu_fieldsort", "test_gpu_schedule",
          "test_gpu_auc", "test_gpu_ckpt", "test_gpu_ingest", "test_gpu_ps_server", "test_gpu_router", "test_gpu_multirank", "test_gpu_multiproc", "test_gpu_rccl_wire", "test_gpu_rehearse_n8"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    faulthandler.enable()
    if os.environ.get("PS_TUNE"):       # measurement / debugging: ps_tune_set knobs for the whole session, "knob=value,knob=value"
        from ps_amd import native as N
        for kv_ in os.environ["PS_TUNE"].split(","):
            if "=" in kv_:
                N.lib().ps_tune_set(kv_.split("=")[0].encode(), int(kv_.split("=")[1]))


def pytest_collection_modifyitems(config, items):
    def rank(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return _ORDER.index(mod) if mod in _ORDER else len(_ORDER) // 2
    items.sort(key=rank)          # stable: order inside a module is kept


class TestTimeout(Exception):
    pass


# Modules that set ISOLATE_IN_SUBPROCESS = True (the N-rank-threads-on-one-GPU tests) get every test
# run in its own python process: N host threads sharing one HIP context is a test stand-in, not a
# deployment shape (the product is one process per GPU), and a wedged rank thread must not take the
# suite down.  A child that exceeds its limit is killed with its whole process group and re-run ONCE;
# the retry is reported loudly (stderr + gpurun_out/isolated_retries.log), never silently.
ISOLATED_CHILD_TIMEOUT_S = int(os.environ.get("PS_AMD_CHILD_TIMEOUT", "75"))


def _run_isolated(item):
    import subprocess
    root = str(item.config.rootpath)
    env = dict(os.environ, PS_AMD_TEST_CHILD="1")
    cmd = [sys.executable, "-m", "pytest", item.nodeid, "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"]
    for attempt in (1, 2):
        proc = subprocess.Popen(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=ISOLATED_CHILD_TIMEOUT_S)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)      # exactly the group this test started
            except ProcessLookupError:
                pass
            out, _ = proc.communicate()
            msg = "%s: child exceeded %d s (attempt %d) and was killed" % (item.nodeid, ISOLATED_CHILD_TIMEOUT_S, attempt)
            sys.stderr.write("\n[isolated] " + msg + "\n" + out.decode(errors="replace")[-4000:] + "\n")
            try:
                os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
                with open(os.path.join(root, "gpurun_out", "isolated_retries.log"), "a") as f:
                    f.write(msg + "\n")
            except OSError:
                pass
            if attempt == 2:
                pytest.fail(msg, pytrace=False)
            continue
        if proc.returncode != 0:
            pytest.fail("%s failed in its child process (rc %d):\n%s" % (item.nodeid, proc.returncode, out.decode(errors="replace")[-6000:]),
                        pytrace=False)
        return


@pytest.hookimpl(tryfirst=True)
def pytest_pyfunc_call(pyfuncitem):
    if getattr(pyfuncitem.module, "ISOLATE_IN_SUBPROCESS", False) and os.environ.get("PS_AMD_TEST_CHILD") != "1":
        _run_isolated(pyfuncitem)
        return True
    return None


@pytest.fixture(autouse=True)
def _per_test_timeout(request):
    """SIGALRM watchdog (main thread): dumps every thread's Python stack, then fails the test."""
    if request.node.get_closest_marker("gpu") is None or not hasattr(signal, "SIGALRM"):
        yield
        return

    limit = int(getattr(request.module, "TEST_TIMEOUT_S", GPU_TEST_TIMEOUT_S))     # (a module may ask for more: 8 full-size rank processes)

    def on_alarm(signum, frame):
        faulthandler.dump_traceback(all_threads=True)
        raise TestTimeout("test exceeded %d s" % limit)

    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(limit)
    try:
        yield
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
