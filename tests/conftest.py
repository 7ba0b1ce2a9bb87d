import os as _os


def _cpu_cap():
    n = _os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    # the GPU boxes show 256 cpus behind a 16-CPU cgroup quota: BLAS pools sized after the visible count spin, the cgroup
    # is throttled and the whole process (GPU-driving thread included) freezes for tens of ms at a time
    _os.environ.setdefault(_v, str(_cpu_cap()))
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
