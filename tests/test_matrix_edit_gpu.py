"""A plain Lis program that edits A->value[] between two solves (tests/c/matrix_edit.c: no lis_amd_* call in it), linked to liblis_amd.so and -- the same source,
the same header -- to the reference library (oracle/_ref/liblis_ref.so), both run here.  The reference adopts the caller's arrays (src/matrix/lis_matrix_csr.c:98-103)
and reads them live on every product (src/matvec/lis_matvec_csr.c:97-109); liblis_amd multiplies an HBM copy built once, so a host write must be SEEN:
  * arrays from lis_matrix_malloc_csr (lis_matrix_csr.c:170) live on pages of the library's own, read-only while the copy lives: the write faults once, the copy is rebuilt;
  * arrays the program malloc'ed cannot be watched: lis_amd_matrix_host_modified is the contract, LIS_AMD_MATRIX_CHECK=1 the debugging aid that finds the stale copy."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_SO = os.path.join(ROOT, "oracle", "_ref", "liblis_ref.so")


@pytest.fixture(scope="module")
def exes(tmp_path_factory):
    d = tmp_path_factory.mktemp("matrix_edit")
    src, inc = os.path.join(HERE, "c", "matrix_edit.c"), os.path.join(ROOT, "include")
    amd = str(d / "matrix_edit_amd")
    libdir = os.path.join(ROOT, "lis_amd", "lib")
    subprocess.run(["gcc", "-O1", "-Wall", "-I" + inc, src, "-o", amd, "-L" + libdir, "-llis_amd", "-lm", "-Wl,-rpath," + libdir], check=True)
    ref = None
    if os.path.exists(REF_SO):             # the same source against the reference library (include/lis.h is its header, byte for byte in layout: test_host_cpu.py)
        ref = str(d / "matrix_edit_ref")
        refdir = os.path.dirname(REF_SO)
        subprocess.run(["gcc", "-O1", "-Wall", "-I" + inc, src, "-o", ref, "-L" + refdir, "-llis_ref", "-lm", "-fopenmp", "-Wl,-rpath," + refdir], check=True)
    return amd, ref


def run(exe, *args, env_extra=None):
    env = dict(os.environ, OMP_NUM_THREADS="1", **(env_extra or {}))
    p = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, (p.returncode, p.stdout[-500:], p.stderr[-800:])
    return p.stdout, p.stderr


def parse(out):
    y = {m.group(1): m.group(2) for m in re.finditer(r"^(\w+) (y_sum .*)$", out, flags=re.M)}
    s = {m.group(1): (int(m.group(2)), int(m.group(3)), float(m.group(4))) for m in re.finditer(r"^(\w+) iter (\d+) resid_ok (\d) xnorm (\S+)$", out, flags=re.M)}
    return y, s


def same(got, want):
    (gy, gs), (wy, ws) = got, want
    assert gy == wy, (gy, wy)                                   # hexfloat sums and samples of y = A w: bit for bit
    assert set(gs) == set(ws) == {"first", "second"}
    for k in gs:
        assert gs[k][0] == ws[k][0] and gs[k][1] == ws[k][1] == 1, (k, gs[k], ws[k])      # CG iteration counts (reproducible across reduction orders)
        assert abs(gs[k][2] - ws[k][2]) <= 1e-9 * ws[k][2]


@pytest.mark.parametrize("N", [20, 48])
def test_host_writes_to_library_allocated_arrays_are_seen(exes, N):
    amd, ref = exes
    if ref is None:
        pytest.skip("oracle/_ref not in this snapshot")
    got, err = run(amd, "lis", N)
    want, _ = run(ref, "lis", N)
    same(parse(got), parse(want))
    y = parse(got)[0]
    assert y["first"] != y["second"] != y["third"]              # the edits do change the product
    assert "MATRIX_CHECK" not in err


def test_caller_malloced_arrays_need_the_call_or_the_check(exes):
    amd, ref = exes
    if ref is None:
        pytest.skip("oracle/_ref not in this snapshot")
    want = parse(run(ref, "malloc", 20)[0])
    out, err = run(amd, "malloc", 20, env_extra={"LIS_AMD_MATRIX_CHECK": "1"})
    same(parse(out), want)
    assert err.count("LIS_AMD_MATRIX_CHECK") == 2               # the copy was found stale twice (after the diagonal edit, after the single write), each time rebuilt
    # without the check nothing can tell: the HBM copy built for the first product keeps answering (include/lis_amd.h: lis_amd_matrix_host_modified is the contract)
    stale = parse(run(amd, "malloc", 20)[0])
    assert stale[0]["first"] == want[0]["first"]
    assert stale[0]["second"].split()[1] == stale[0]["first"].split()[1] != want[0]["second"].split()[1]


def test_eager_coherence_keeps_the_contract_of_the_explicit_call(exes):
    """LIS_AMD_COHERENCE=eager never protects pages: library-allocated arrays behave like the caller's own there, and the check finds the stale copy"""
    amd, ref = exes
    if ref is None:
        pytest.skip("oracle/_ref not in this snapshot")
    want = parse(run(ref, "lis", 20)[0])
    out, err = run(amd, "lis", 20, env_extra={"LIS_AMD_COHERENCE": "eager", "LIS_AMD_MATRIX_CHECK": "1"})
    same(parse(out), want)
    assert err.count("LIS_AMD_MATRIX_CHECK") == 2
