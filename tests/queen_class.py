"""The Queen_4147 stand-in of BASELINE config 4 (tests/golden/gen_queen_class.c): build the generator with gcc, write the
symmetric Matrix Market file, name the cases the committed fixture (tests/golden/queen_class_golden.json) holds.
Test infrastructure only."""
import os
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "golden", "gen_queen_class.c")
# name -> (G, band): "mini" runs everywhere in a second (CPU reader parity), "full" is Queen's scale (4.1 M rows, 2.9e8 non-zeros)
CASES = {"mini": (12, 64), "full": (111, 1024)}


def scratch_dir():
    """/dev/shm when it has room for the 3.3 GB file (RAM-backed: the reader's time is then parsing, not the disk), else $TMPDIR"""
    want = 5 << 30
    for d in ("/dev/shm", tempfile.gettempdir()):
        try:
            st = os.statvfs(d)
            if st.f_bavail * st.f_frsize > want and os.access(d, os.W_OK):
                return d
        except OSError:
            pass
    return tempfile.gettempdir()


def generate(case, out_dir=None):
    """-> (path of the .mtx, rows, stored entries); the file is rewritten every time (deterministic)"""
    G, band = CASES[case]
    out_dir = out_dir or (scratch_dir() if case == "full" else tempfile.gettempdir())
    exe = os.path.join(tempfile.gettempdir(), f"gen_queen_class_{os.getpid()}")
    subprocess.run(["gcc", "-O2", "-o", exe, SRC], check=True)
    path = os.path.join(out_dir, f"queen_class_{case}_{os.getpid()}.mtx")
    try:
        out = subprocess.run([exe, str(G), str(band), path], check=True, capture_output=True, text=True).stdout.split()
    finally:
        os.unlink(exe)
    return path, int(out[0]), int(out[1])
