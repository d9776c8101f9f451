"""Several threads of the program reading x->value right after lis_solve, on a real MI355X (tests/c/pages_threads.c gpu-solve).

The solution comes home from HBM on the first touch; the copy is written through the library's alias mapping of the vector's pages while
the program's own mapping still has no access (lis_amd/csrc/host/lis_pages.c), so every thread either waits for the complete vector or
faults -- none reads a stale page.  Three solves in a row (the same threads fault at the same addresses again: the handler must stay
installed), host writes from all threads in between, then the same under LIS_AMD_COHERENCE=eager: the solution's bits must agree."""
import subprocess

import pytest

import lis_amd
from test_host_cpu import build_pages_driver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("threads,N", [(8, 64), (3, 40)])
def test_openmp_readers_after_lis_solve(tmp_path, threads, N):
    assert lis_amd.gpu_available(), "no HIP device: the product path has no CPU fallback"
    exe = build_pages_driver(tmp_path)
    out = subprocess.run([exe, "gpu-solve", str(threads), str(N)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "ok gpu-solve" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])
