#!/usr/bin/env python
"""bench.py -- the hot path's headline metric on N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps 50] [--warmup 5]        (N > 1: launches its own N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W                (the same job under an outside launcher)

Workload (BASELINE.json): the 3-D 7-point Poisson matrix on a 512^3 grid in CSR (test/test3.c entry order, f64 values, i32 indices), generated
directly in HBM.  A "step" is one y = A*x through lis_matvec() of liblis_amd.so in the REFERENCE LAYOUT mode (lis_amd_set_reference_layout): the kernel
streams the reference's own arrays -- 4 B index[] + 8 B value[] per non-zero, ptr[], y, x gathered: the loop of src/matvec/lis_matvec_csr.c:97-109 on
SURVEY 8d's 12 nnz + 20 n + 4 bytes -- on a NON-TRIVIAL x (x_i = frac(i * golden ratio) - 0.5).  That is the line's `value`, `ms_per_step` and `roofline`
(`frac` <= 1 by construction: exactly those bytes over the kernel's HIP-event time over 8 TB/s; `traffic` = this run's own PMC counters).
Beside it, each labelled with what it applies to:
  x_equals_one           the same kernel on the reference's x = 1 (test/spmvtest3.c), with its closed-form norm check
  krylov                 CG+Jacobi / BiCGSTAB / BiCG / GMRES(30) iterations per second in the same mode: iter / itime of lis_solver_get_timeex (lis_solver.c:902-908)
  structured_fast_path   what lis_matvec runs for THIS matrix by default -- the plan found a constant-coefficient box stencil and marches it without reading the
                         matrix arrays at all; NOT a CSR-roofline figure (its own bytes, its own `applies_to`), with its Krylov rates and the values-streamed form
  configs                one driver-timed leg per BASELINE.json config: config1 spmvtest1 (n = 10000, all six formats, the 2-norm check), config2 256^3 CG+Jacobi
                         to convergence, config3 512^3 BiCGSTAB (this job's ranks), config4 the Queen_4147 stand-in (or LIS_AMD_BENCH_MTX=/path/file.mtx),
                         config5 CSR / ELL / DIA at 256^3 and 512^3 through their native kernels on SURVEY 8d's bytes and through the default forms
  cpu_baseline           the reference's own OpenMP path (oracle/_ref = Lis 2.1.11) on this box's host cores
With N > 1 the matrix is row-block partitioned (LIS_GET_ISIE, whole grid planes) and every step does the halo exchange over RCCL.  Default WEAK scaling: every
GPU keeps the 512^3 rows the metric is quoted on (global grid 512 x 512 x 512 N); --scaling strong splits the one 512^3 grid.  value = 2 nnz K / t whole job.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BASELINE_CONFIGS = {           # BASELINE.json `configs`, verbatim: every leg under `configs` carries its name
    "config1": "test/spmvtest1: 1D 3-pt Poisson n=10000 CSR, reference CPU OpenMP path (plumbing, no GPU)",
    "config2": "3D 7-pt Poisson n=256^3 CSR, CG+Jacobi, 1 MI355X",
    "config3": "3D 7-pt Poisson n=512^3 CSR, BiCGSTAB, row-block across 8×MI355X (RCCL halo+allreduce)",
    "config4": "SuiteSparse Queen_4147 (irregular CSR), GMRES(30), 1 MI355X — merge-path load-balance",
    "config5": "3D 7-pt Poisson n=256^3 in ELL and DIA formats, CG, 1 MI355X — format sweep vs CSR",
}
SOLVERS = (("cg_jacobi", "-i cg -p jacobi"), ("bicgstab_none", "-i bicgstab -p none"),
           ("bicg_none", "-i bicg -p none"),            # Lis's default solver: needs A^T (built in HBM on first use)
           ("gmres30_none", "-i gmres -restart 30 -p none"))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=512, help="cubic grid edge (BASELINE: 512)")
    ap.add_argument("--solver-iters", type=int, default=200,
                    help="Krylov iterations timed for the it/s figures at the headline size (maxiter of the timed solve; none of the solvers "
                         "converges earlier at 512^3: CG needs 1504)")
    ap.add_argument("--preroll", type=int, default=200,
                    help="untimed clock-ramp steps before the W warm-up steps (reported in the line): a fresh box's first process measured "
                         "7 %% slower for its first ~half second of kernels; same count on every rank")
    ap.add_argument("--comm", choices=["rccl", "callbacks"], default="rccl",
                    help="callbacks: collectives through torch.distributed/gloo host callbacks -- bring-up of the N>1 "
                         "path with several ranks on ONE GPU (RCCL refuses that); never a measurement")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = 512^3 rows per GPU (grid 512 x 512 x 512 N), strong = the one 512^3 grid split over the ranks")
    ap.add_argument("--planes", type=int, default=0,
                    help="the slowest grid dimension when it is not --grid (whole job): `--planes 64` on one GPU is the slab ONE rank of the 8-rank strong-scaling job "
                         "owns (512 x 512 x 64 rows) -- the per-rank kernel times tools/scale_predict.py builds the predicted curve from.  Not the headline workload: "
                         "`config.workload` says so")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed profiles/ figure instead of two rocprofv3 --pmc passes of this run (about 15 s)")
    ap.add_argument("--no-solvers", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the 27-point stencil leg and the config legs (rank 0, N = 1 only)")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config legs (config1, 2, 4, 5; rank 0, N = 1 only)")
    ap.add_argument("--launch-check", action="store_true",
                    help="bring-up of the launch path without a GPU: form the process group, gather the ranks, print one JSON line, exit")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks here, one per GPU, under torch.distributed.run on
    127.0.0.1 with a free port -- the same command line the driver's launcher would run.  Rank 0's JSON line is this process's stdout; the exit code is the job's."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), LIS_AMD_BENCH_SELF_LAUNCHED="1")
    print(f"bench.py: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    # stdout carries ONE line, the JSON: the library (lis_input's "matrix size = ...", as the reference prints it), the reference build of the CPU baseline and child
    # processes write to file descriptor 1 too -- from here on that is stderr, and the line goes out through the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world

    def emit(obj):
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(obj) + "\n").encode())          # the ONE line of stdout (everything else this process printed went to stderr)

    # torch first: liblis_amd.so then binds the HIP / RCCL runtime torch already loaded (one runtime per process),
    # so torch.cuda.synchronize() and the library see the same device context.  torch is plumbing only.
    torch = dist = None
    try:
        import torch
        import torch.distributed as dist
    except Exception:                                    # single-GPU runs do not need it
        if world > 1:
            raise
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)      # control plane only; data plane is RCCL
    if args.launch_check:
        ranks = [None] * world
        if world > 1:
            dist.all_gather_object(ranks, (rank, local_rank))
            dist.destroy_process_group()
        else:
            ranks = [(rank, local_rank)]
        if rank == 0:
            emit({"launch_check": True, "world": world, "ranks": [list(r) for r in ranks], "self_launched": os.environ.get("LIS_AMD_BENCH_SELF_LAUNCHED") == "1"})
        return
    import numpy as np
    import lis_amd
    from lis_amd import _capi as capi, check
    lib = lis_amd.load()
    dll = lib.dll
    if world == 1:
        os.environ.setdefault("LIS_AMD_DEVICE", str(local_rank))
    assert lib.initialize([]) == 0
    comm_used = args.comm
    if world > 1 and args.comm == "rccl":
        # RCCL communicator: rank 0's unique id travels over the gloo control plane.  If ANY rank fails to join,
        # every rank drops to the callback backend so that the job still reports (marked as such: `degraded`, exit code 3).
        uid = [None]
        ok = int(torch.cuda.is_available() and local_rank < torch.cuda.device_count())   # a rank without its own GPU
        flag = torch.tensor([ok], dtype=torch.int32)                                      # must not leave the others
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                                       # waiting inside ncclCommInitRank
        ok = int(flag[0])
        if rank == 0 and ok:
            buf = (C.c_char * 128)()
            if dll.lis_amd_comm_get_unique_id(buf) == 0:
                uid[0] = bytes(buf)
        dist.broadcast_object_list(uid, src=0)
        if ok and uid[0] is not None:
            ok = int(dll.lis_amd_comm_init_rccl(uid[0], rank, world, local_rank) == 0)
        else:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            if rank == 0:
                print("bench.py: RCCL communicator could not be formed on every rank -- falling back to host callbacks",
                      file=sys.stderr, flush=True)
            dll.lis_amd_comm_finalize()
            comm_used = "callbacks"
    if world > 1 and comm_used == "callbacks":
        from lis_amd._hostcomm import make_callbacks
        local_rank = local_rank % max(1, torch.cuda.device_count())
        os.environ["LIS_AMD_DEVICE"] = str(local_rank)
        cb = make_callbacks(world)
        assert dll.lis_amd_comm_init_callbacks(C.byref(cb), rank, world) == 0
    if torch is not None and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)
    dll.lis_amd_set_residency(1)                          # objects live in HBM; nothing crosses PCIe in the timed region
    dll.lis_amd_stream.restype = C.c_void_p
    for fn in ("lis_amd_matrix_index_codes", "lis_amd_matrix_row_patterns", "lis_amd_matrix_pattern_records", "lis_amd_matrix_value_records",
               "lis_amd_matrix_dominant_pattern", "lis_amd_matrix_marching", "lis_amd_matrix_strip_rows", "lis_amd_matrix_device_type"):
        getattr(dll, fn).argtypes = [capi.PM]
    dll.lis_amd_matrix_poisson3d.argtypes = [capi.PM, C.c_int, C.c_int, C.c_int, C.c_int]
    dll.lis_amd_vector_poisson3d_rhs.argtypes = [capi.PV, C.c_int, C.c_int, C.c_int]

    def sync():
        assert dll.lis_amd_synchronize() == 0
        if torch is not None and torch.cuda.is_available():
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    def reference_layout(on):
        assert dll.lis_amd_set_reference_layout(1 if on else 0) == 0

    N = args.grid
    L = N * world if args.scaling == "weak" else N        # grid planes (slowest dimension): whole planes per rank
    if args.planes > 0:
        L = args.planes
    slab = args.planes > 0 and L != N
    n_global = L * N * N
    if L % world:
        sys.exit(f"grid edge {N} is not divisible by {world} ranks (whole planes per rank)")
    if n_global >= 2 ** 31:
        sys.exit(f"{L} x {N} x {N} rows do not fit LIS_INT (32 bit, as the reference's default build)")

    def poisson(l, m, n, sorted_=0):
        M = capi.PM()
        assert lib.lis_matrix_create(capi.LIS_COMM_WORLD, C.byref(M)) == 0
        assert lib.lis_matrix_set_size(M, 0, l * m * n) == 0
        assert dll.lis_amd_matrix_poisson3d(M, l, m, n, sorted_) == 0    # generated in HBM + the plan
        return M

    def vec(M):
        v = capi.PV()
        assert lib.lis_vector_duplicate(C.cast(M, C.c_void_p), C.byref(v)) == 0
        return v

    def golden_x(v, n_rows):
        """x_i = frac(i * golden ratio) - 0.5 on this rank's rows: every mantissa bit toggles (the FP64 multipliers then draw their real power and the clocks follow)"""
        lo = C.c_int(); hi = C.c_int()
        assert lib.lis_vector_get_range(v, C.byref(lo), C.byref(hi)) == 0
        chunk = 1 << 24
        for s0 in range(0, n_rows, chunk):
            cnt = min(chunk, n_rows - s0)
            part = np.modf(np.arange(lo.value + s0, lo.value + s0 + cnt, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
            assert lib.lis_vector_set_values2(capi.LIS_INS_VALUE, lo.value + s0, cnt, part.ctypes.data_as(capi.P_DBL), v) == 0

    dll.lis_amd_synchronize()
    t_setup = time.perf_counter()
    A = poisson(L, N, N)                                  # the DEFAULT plan (row split, index codes, row patterns, value records): what a Lis program gets
    dll.lis_amd_synchronize()
    setup_ms = (time.perf_counter() - t_setup) * 1e3
    n_local, nnz_local = A.contents.n, A.contents.nnz
    # the generator alone, into scratch arrays: what is left of setup_ms is the plan -- a one-off per matrix, paid at assemble / convert time
    gen_ms = None
    try:
        from lis_amd import DeviceArray as _DA
        s_ptr, s_idx, s_val = _DA(n_local + 5, np.int32), _DA(nnz_local + 4, np.int32), _DA(nnz_local + 2, np.float64)
        lo_r, hi_r = A.contents.is_, A.contents.ie
        check(lib.liship_poisson3d_csr(L, N, N, lo_r, hi_r, 0, s_ptr.ptr, s_idx.ptr, s_val.ptr, None))     # (first call: code load)
        dll.lis_amd_synchronize()
        t_gen = time.perf_counter()
        check(lib.liship_poisson3d_csr(L, N, N, lo_r, hi_r, 0, s_ptr.ptr, s_idx.ptr, s_val.ptr, None))
        dll.lis_amd_synchronize()
        gen_ms = (time.perf_counter() - t_gen) * 1e3
        s_ptr.free(); s_idx.free(); s_val.free()
    except Exception:
        gen_ms = None
    nnz_global = 7 * n_global - 2 * (N * N + 2 * L * N)

    x1, xg, y, yc, b = vec(A), vec(A), vec(A), vec(A), vec(A)
    assert lib.lis_vector_set_all(1.0, x1) == 0
    golden_x(xg, n_local)

    stream = dll.lis_amd_stream()
    timer = C.c_void_p()
    check(lib.liship_timer_create(C.byref(timer)))
    ev_ms = C.c_float()
    nrm = C.c_double()

    def timed_products(M, xv, yv, steps, collective=True):
        """exactly `steps` products between barrier + device sync on both sides: (seconds, MAX over ranks; HIP-event ms per launch on the library's stream)"""
        sync()
        if collective:
            barrier()
        t0 = time.perf_counter()
        check(lib.liship_timer_start(timer, stream))
        for _ in range(steps):
            assert lib.lis_matvec(M, xv, yv) == 0
        check(lib.liship_timer_stop(timer, stream))
        sync()
        if collective:
            barrier()
        el = time.perf_counter() - t0
        check(lib.liship_timer_elapsed_ms(timer, C.byref(ev_ms)))
        if world > 1 and collective:
            tt = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt[0])
        return el, ev_ms.value / steps

    def leg(M, xv, yv, steps, nnz, collective=True):
        for _ in range(max(args.warmup, 5)):
            assert lib.lis_matvec(M, xv, yv) == 0
        el, k_ms = timed_products(M, xv, yv, steps, collective)
        return {"value": round(2.0 * nnz * steps / el / 1e9, 2), "unit": "GFLOP/s", "ms_per_step": round(el / steps * 1e3, 4), "kernel_ms": round(k_ms, 4)}

    # result check outside the timed region: ||A*1||_2^2 = 6(N-2)^2 + 48(N-2) + 72 exactly on the cube (spmvtest3)
    # (a row sums to the number of neighbours it lacks: 8 corners 3, the edges 2, the faces 1)
    expect = (72.0 + 4.0 * (4 * (L - 2) + 8 * (N - 2)) + 2.0 * (N - 2) ** 2 + 4.0 * (L - 2) * (N - 2)) ** 0.5

    def check_a_times_one(what):
        assert lib.lis_vector_nrm2(y, C.byref(nrm)) == 0
        if abs(nrm.value - expect) > 1e-12 * expect:
            sys.exit(f"rank {rank}: {what}: ||A*1||_2 = {nrm.value!r}, expected {expect!r}")

    def same_bits(u, v, what):
        """||u - v||_2 == 0 exactly (u is overwritten): two forms of the product must agree to the last bit"""
        assert lib.lis_vector_axpy(-1.0, v, u) == 0 and lib.lis_vector_nrm2(u, C.byref(nrm)) == 0
        if nrm.value != 0.0:
            sys.exit(f"rank {rank}: {what}: the two forms of the product differ: ||dy||_2 = {nrm.value!r}")

    # ---- which form of the matrix the DEFAULT plan keeps (found on the device at upload)
    alg_bytes = 12 * nnz_local + 20 * n_local + 4        # 12 B per non-zero + 20 B per row (+ the last ptr entry)
    coded = int(dll.lis_amd_matrix_index_codes(A))
    patterns = int(dll.lis_amd_matrix_row_patterns(A))     # > 0: one byte per ROW (pattern) instead of 1 B per non-zero + 4 B per row
    records = int(dll.lis_amd_matrix_pattern_records(A))   # 1: patterns of <= 7 offsets kept as 32 B records (gathers ahead of the value slice)
    values = int(dll.lis_amd_matrix_value_records(A))      # 1: the rows of a pattern share their values too (constant coefficients)
    dominant = int(dll.lis_amd_matrix_dominant_pattern(A))     # 1: one pattern carries most rows and its x gathers are issued with the pattern bytes (round 3)
    marching = int(dll.lis_amd_matrix_marching(A))             # round 4: 1 = the z-marching form of that product (each x loaded once per plane tile), 2 = its box form (no pattern bytes read)
    if world > 1:                                          # (one answer for the whole job: the legs below are collective)
        vv = torch.tensor([values, marching], dtype=torch.int32)
        dist.all_reduce(vv, op=dist.ReduceOp.MIN)
        values, marching = int(vv[0]), int(vv[1])

    REF = "reference_layout"      # kernel_name / live_traffic / roofline_of: the product on the reference's own arrays (4 B indices, 8 B values)

    def kernel_name(v):
        if v == REF:
            return "spmv_csr_rowgather_kernel"
        pair = n_local * 8 > (256 << 20)                  # round-2 kernels: x beyond the Infinity Cache takes the two-rows-per-lane form
        if patterns and records and v:
            if dominant and marching:
                return "spmv_csr_valuerec_march_kernel"
            return "spmv_csr_valuerec_dom_kernel" if dominant else "spmv_csr_valuerec_pair_kernel" if pair else "spmv_csr_valuerec_kernel"
        return ("spmv_csr_pattern7_kernel" if patterns and records else
                "spmv_csr_pattern_kernel" if patterns else "spmv_csr_coded_kernel" if coded else "spmv_csr_rowgather_kernel")

    def pmc_traffic(name):
        """HBM-side bytes per launch from the PMC passes of the same products (rocprofv3 --pmc cannot run inside the process it profiles;
        separate passes as the microarchitecture guide prescribes: tools/prof.sh + tools/traffic_json.py), committed under profiles/.
        Counted at the L2 <-> fabric boundary by request size, so re-reads the 256 MB Infinity Cache answers are included: an UPPER bound
        of the HBM bytes; the stored bytes are the lower one."""
        if N != 512 or world != 1 or slab:
            return None, None
        for tf in ("r06_spmv512_traffic%s.json", "r05_spmv512_traffic%s.json", "r04_spmv512_traffic%s.json", "r03_spmv512_traffic%s.json"):
            tf = os.path.join(ROOT, "profiles", tf % name)
            if os.path.exists(tf):
                tj = json.load(open(tf))
                detail = {k: tj[k] for k in ("source", "kernel", "level", "avg_kernel_ns", "read_bytes_per_launch", "write_bytes_per_launch",
                                             "x_bytes_per_launch", "x_refetch_factor", "fabric_rate_GBs", "hbm_bytes_bounds") if k in tj}
                detail["file"] = os.path.relpath(tf, ROOT)
                return tj.get("fabric_bytes_per_launch", tj.get("hbm_traffic_bytes_per_launch")), detail
        return None, None

    live_cache = {}

    def live_traffic(v):
        """The same bytes measured IN THIS RUN: rocprofv3 --kernel-trace --pmc over a child process that sets up the same matrix and runs 12 products
        (tools/traffic_child.py), one pass for the read requests by size, one for the writes -- separate passes with kernel-trace only, as the
        microarchitecture guide prescribes.  None when rocprofv3 is not there or a pass fails (the committed figure then serves, and `traffic_detail` says so)."""
        if v in live_cache:
            return live_cache[v]
        live_cache[v] = None
        if N != 512 or world != 1 or slab or args.no_live_traffic:
            return None
        import csv
        import glob
        import shutil
        import subprocess
        import tempfile
        prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
        if not os.path.exists(prof):
            return None
        want = kernel_name(v)
        per, t0 = {}, time.perf_counter()
        tmp = tempfile.mkdtemp(prefix="lis_amd_traffic_", dir="/tmp")
        try:
            for tag, ctrs in (("rd", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]),
                              ("wr", ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"])):
                out_dir = os.path.join(tmp, tag)
                cmd = [prof, "--kernel-trace", "--output-format", "csv", "--pmc", *ctrs, "-d", out_dir, "-o", "pmc", "--",
                       sys.executable, os.path.join(ROOT, "tools", "traffic_child.py"), str(N), str(2 if v == REF else int(bool(v))), "12"]
                r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180)
                if r.returncode != 0:
                    return None
                acc = {}
                for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        if want in row.get("Kernel_Name", "") and "_dot" not in row.get("Kernel_Name", ""):
                            a = acc.setdefault(row["Counter_Name"], [0.0, 0])
                            a[0] += float(row["Counter_Value"]); a[1] += 1
                for k, (tot, cnt) in acc.items():
                    per[k] = tot / cnt
            need = ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum")
            if any(k not in per for k in need):
                return None
            r32, r64, r128 = per["TCC_EA0_RDREQ_32B_sum"], per["TCC_EA0_RDREQ_64B_sum"], per["TCC_EA0_RDREQ_128B_sum"]
            rd = 32 * r32 + 64 * r64 + 128 * r128 + 64 * max(0.0, per["TCC_EA0_RDREQ_sum"] - r32 - r64 - r128)
            wr = 64 * per["TCC_EA0_WRREQ_64B_sum"] + 32 * (per["TCC_EA0_WRREQ_sum"] - per["TCC_EA0_WRREQ_64B_sum"])
            live_cache[v] = {"read_bytes_per_launch": int(rd), "write_bytes_per_launch": int(wr), "fabric_bytes_per_launch": int(rd + wr), "kernel": want,
                             "source": "measured in this run: rocprofv3 --kernel-trace --pmc (two passes) over tools/traffic_child.py, 12 products of the same matrix in the same mode",
                             "level": "L2 <-> fabric requests by size (Infinity Cache hits included: an upper bound of the HBM bytes)",
                             "seconds": round(time.perf_counter() - t0, 1)}
            return live_cache[v]
        except Exception:
            return None
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    def roofline_of(v, k_ms, traffic_name, applies_to, live=True):
        """`achieved` / `frac`: the bytes THIS kernel is asked to move over its own HIP-event time -- for the reference layout exactly SURVEY 8d's 12 nnz + 20 n + 4;
        for a derived form its stored streams + y + the compulsory x.  A physical rate, <= the peak by construction.  `traffic`: the PMC bytes (upper bound, see
        pmc_traffic).  `contract_*`: SURVEY 8d's count over the same time -- identical to `frac` for the reference layout; for a derived form NOT a rate of that
        kernel (it reads fewer bytes) and it may exceed 1."""
        if v == REF:
            moved = alg_bytes
        else:
            moved = spmv_stored_bytes(n_local, nnz_local, coded, patterns, v) - (n_local if (v and marching == 2) else 0)      # (the box form reads no pattern byte: x and y alone)
        traffic, detail = pmc_traffic(traffic_name)
        lt = live_traffic(v) if (rank == 0 and live) else None
        if lt:                                              # this run's own counters take precedence over the committed figure (kept beside them)
            detail = dict(lt, committed=detail)
            traffic = lt["fabric_bytes_per_launch"]
        sec = k_ms * 1e-3
        r = {"bound": "hbm", "achieved": round(moved / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(moved / sec / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_detail": detail,
             "kernel": kernel_name(v), "kernel_ms": round(k_ms, 4), "bytes_per_launch": moved, "per_gpu": True,
             "bytes_are": ("SURVEY 8d's algorithmic bytes: 12 B per non-zero (8 B value[] + 4 B index[]) + 20 B per row (4 B ptr[], 8 B y, 8 B compulsory x) + 4" if v == REF else
                           "stored matrix streams + y + compulsory x of the timed kernel (lower bound of its HBM bytes; `traffic` is the counters' upper bound)"),
             "contract_bytes_per_launch": alg_bytes, "contract_achieved": round(alg_bytes / sec / 1e9, 1),
             "contract_frac": round(alg_bytes / sec / 1e9 / HBM_PEAK_GBS, 4),
             "index_codes": 0 if v == REF else coded, "row_patterns": 0 if v == REF else patterns,
             "value_records": 0 if v == REF else v, "marching": marching if (v and v != REF) else 0, "applies_to": applies_to}
        if traffic:
            r["traffic_over_bytes"] = round(traffic / moved, 3)
        return r

    def solve_rates(M, rhs, sol, keys, iters, n_rows, nnz_rows, form):
        """Krylov iterations/s on M (b = A*1, x0 = 0), whole job: iter / itime as the reference splits it (lis_solver.c:902-908).
        form: (coded, patterns, values) of the products the loops run, for the bytes one iteration is asked to stream."""
        out = {}
        for key, opts in SOLVERS:
            if key not in keys:
                continue
            S = capi.PS()
            assert lib.lis_solver_create(C.byref(S)) == 0
            # warm-up pass: work-vector pool, A^T, first launches -- the one-off costs the reference pays in its own setup
            assert lib.lis_solver_set_option(f"{opts} -tol 1e-12 -maxiter 20".encode(), S) == 0
            assert lib.lis_solve(M, rhs, sol, S) == 0
            assert lib.lis_solver_set_option(f"-maxiter {iters}".encode(), S) == 0
            sync(); barrier()
            t1 = time.perf_counter()
            assert lib.lis_solve(M, rhs, sol, S) == 0
            sync(); barrier()
            wall = time.perf_counter() - t1
            tm = [C.c_double() for _ in range(5)]           # time, itime, ptime, p_c_time, p_i_time
            assert lib.lis_solver_get_timeex(S, *[C.byref(t) for t in tm]) == 0
            itime = tm[1].value
            it = min(S.contents.iter, iters)
            if world > 1:
                tt = torch.tensor([itime, wall], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                itime, wall = float(tt[0]), float(tt[1])
            uniform = int(dll.lis_amd_last_solve_uniform_jacobi())      # CG + Jacobi on a constant diagonal: 1/diag rides as a scalar
            loop_b, contract_b = krylov_bytes(key, it, n_rows, nnz_rows, form[0], form[1], form[2], uniform)
            sec_per_iter = itime / max(1, it)
            out[key] = {
                "iters_per_sec": round(it / itime, 2), "iters_timed": it, "itime_s": round(itime, 6),
                "lis_solve_wall_s": round(wall, 4), "rel_residual_after": S.contents.resid, "status": S.contents.retcode,
                # per GPU: bytes one iteration's passes are asked to stream (fused loops: DESIGN.md 6)
                # and the bytes of the reference's unfused operator sequence over the same time
                "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "per_gpu": True,
                             "loop_bytes_per_iter": loop_b, "achieved": round(loop_b / sec_per_iter / 1e9, 1),
                             "frac": round(loop_b / sec_per_iter / 1e9 / HBM_PEAK_GBS, 4),
                             "contract_bytes_per_iter": contract_b,
                             "contract_frac": round(contract_b / sec_per_iter / 1e9 / HBM_PEAK_GBS, 4)}}
            lib.lis_solver_destroy(S)
        return out

    # =====================================================================================================================================
    # HEADLINE: the reference layout on a non-trivial x.  Untimed clock ramp (power state of an idle box), the contract's W warm-up steps,
    # then exactly K steps between barrier + device sync on both sides.
    reference_layout(True)
    for _ in range(args.preroll):
        assert lib.lis_matvec(A, xg, y) == 0
    sync()
    for _ in range(args.warmup):
        assert lib.lis_matvec(A, xg, y) == 0
    dt, kernel_ms = timed_products(A, xg, y, args.steps)
    ms_per_step = dt / args.steps * 1e3
    gflops = 2.0 * nnz_global * args.steps / dt / 1e9
    assert lib.lis_vector_copy(y, yc) == 0                # kept: the default form below must reproduce it bit for bit
    roofline = roofline_of(REF, kernel_ms, "_contract_form",
                           "ANY CSR matrix with short rows: nothing about the matrix is assumed or precomputed beyond the row split (12 B per non-zero + 20 B per row streamed)")
    roofline["xcd_strip_rows"] = int(dll.lis_amd_matrix_strip_rows(A))     # rows per plane the XCD strips are cut from (0: natural block order)
    x_one = leg(A, x1, y, args.steps, nnz_global)
    check_a_times_one("reference layout, x = 1")
    x_one["x"] = "x = 1 (test/spmvtest3.c): the same kernel, the same bytes; the multipliers see one mantissa pattern and the clocks run a few per cent higher"
    x_one["frac"] = round(alg_bytes / (x_one["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    x_one["norm_check"] = "||A*1||_2 equals its closed form to 1e-12"

    # ---- N > 1: what the exchange costs by itself, and the product with the overlap switched off (A/B), in the headline's mode
    multi = None
    if world > 1:
        dll.lis_amd_halo_exchange.argtypes = [capi.PM, capi.PV]

        def timed(fn, reps):
            sync(); barrier()
            t = time.perf_counter()
            for _ in range(reps):
                assert fn() == 0
            sync(); barrier()
            el = time.perf_counter() - t
            tt = torch.tensor([el], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return float(tt[0]) / reps * 1e3
        halo_ms = timed(lambda: dll.lis_amd_halo_exchange(A, xg), args.steps)
        dll.lis_amd_set_overlap(0)
        serial_ms = timed(lambda: lib.lis_matvec(A, xg, y), args.steps)
        dll.lis_amd_set_overlap(1)
        nb = [int(rank > 0), int(rank < world - 1)]           # (whole planes per rank: the ranks before and behind)
        dll.lis_amd_comm_halo_communicator.restype = C.c_int
        multi = {"halo_ms_per_step": round(halo_ms, 4), "ms_per_step_no_overlap": round(serial_ms, 4),
                 "ms_per_step_overlap": round(ms_per_step, 4),
                 # what one product moves over xGMI per rank: one N x N plane of doubles out and one in per neighbour (interior ranks: two neighbours)
                 "halo_bytes_per_neighbour": 8 * N * N, "halo_bytes_per_interior_rank_per_step": 2 * 2 * 8 * N * N if world > 2 else 2 * 8 * N * N,
                 "neighbours_of_rank0": sum(nb), "halo_communicator": bool(dll.lis_amd_comm_halo_communicator()) if comm_used == "rccl" else None,
                 "folds_per_iteration": {"cg_jacobi": 2, "bicgstab_none": 4, "bicg_none": 2, "gmres30_none": "i + 1 at inner step i"}}

    keys = [k for k, _ in SOLVERS]
    krylov = {}
    if not args.no_solvers:
        assert dll.lis_amd_vector_poisson3d_rhs(b, L, N, N) == 0
    reference_layout(False)

    # =====================================================================================================================================
    # BESIDE IT: what lis_matvec runs for this matrix by default (the plan's derived form), same matrix object, same vectors
    CONSTANT = ("constant-coefficient matrices only: the 27 row patterns of this stencil carry their VALUES (checked bit for bit at plan time), one pattern "
                "byte per row is the only matrix stream" + ("; the grid is a box (checked row by row at plan time), the z-marching kernel reads x once per plane tile "
                "and no pattern byte: x and y alone are streamed (DESIGN.md 4).  Applies to no matrix with varying coefficients -- NOT a CSR-roofline figure" if marching == 2 else "; latency-bound, not byte-bound (DESIGN.md 4)"))
    GENERAL = "any matrix on these sparsity patterns, whatever its coefficients: 8 B per non-zero + one pattern byte per row streamed"
    fast = leg(A, xg, y, args.steps, nnz_global)
    same_bits(y, yc, "default form vs reference layout")
    fast["x"] = "x_i = frac(i * 0.618...) - 0.5 (as the headline); y equals the headline's y bit for bit (checked)"
    fast["roofline"] = roofline_of(values, fast["kernel_ms"], "", CONSTANT if values else GENERAL, live=False)
    fast["kernel"] = fast["roofline"]["kernel"]
    fast["applies_to"] = fast["roofline"]["applies_to"]
    fast["x_equals_one"] = leg(A, x1, y, args.steps, nnz_global)
    check_a_times_one("default form, x = 1")
    fast["x_equals_one"]["frac"] = round(fast["roofline"]["bytes_per_launch"] / (fast["x_equals_one"]["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if values:                                              # the same product with the value records switched off (the general kernel: 8 B per non-zero + 17 B per row)
        check(lib.liship_spmv_csr_set_row_values(0))
        try:
            st = leg(A, xg, y, args.steps, nnz_global)
            same_bits(y, yc, "values-streamed form vs reference layout")
            st["roofline"] = roofline_of(0, st["kernel_ms"], "_values_streamed", GENERAL, live=False)
            st["kernel"] = st["roofline"]["kernel"]
            fast["values_streamed"] = st
        finally:
            check(lib.liship_spmv_csr_set_row_values(1))
    if not args.no_solvers:
        # (the default form first: the transposed copy BiCG needs is built once, with the default plan's column codes; the reference layout mode then switches
        #  the codes of BOTH copies off at run time, exactly as it does for A itself)
        fast["krylov"] = solve_rates(A, b, y, keys, args.solver_iters, n_local, nnz_local, (coded, patterns, values))
        reference_layout(True)
        krylov = solve_rates(A, b, y, keys, args.solver_iters, n_local, nnz_local, (0, 0, 0))
        reference_layout(False)

    configs = {}
    if not args.no_solvers:
        k3 = krylov.get("bicgstab_none", {})
        configs["config3"] = {"name": BASELINE_CONFIGS["config3"], "n_gpus": world, "grid": f"{L} x {N} x {N}",
                              "parallelism": f"row-block x{world}" + (" + RCCL halo + rank-order folds" if world > 1 and comm_used == "rccl" else ""),
                              "reference_layout": {"iters_per_sec": k3.get("iters_per_sec"), "frac": k3.get("roofline", {}).get("frac"), "iters_timed": k3.get("iters_timed")},
                              "default_form": {"iters_per_sec": fast.get("krylov", {}).get("bicgstab_none", {}).get("iters_per_sec"),
                                               "frac": fast.get("krylov", {}).get("bicgstab_none", {}).get("roofline", {}).get("frac")},
                              "note": "BiCGSTAB -p none on this job's ranks (`krylov.bicgstab_none` / `structured_fast_path.krylov.bicgstab_none` hold the full entries); "
                                      "the 8-GPU figure is this line at --gpus 8"}

    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        extras = stencil27_leg(lib, np, C, stream)
        if not args.no_configs:
            ctx = {"lib": lib, "dll": dll, "np": np, "capi": capi, "check": check, "poisson": poisson, "vec": vec, "golden_x": golden_x, "leg": leg,
                   "reference_layout": reference_layout, "solve_rates": solve_rates, "nrm": nrm, "steps": args.steps, "grid": N, "headline_matrix": A}
            configs["config1"] = config1_leg(ctx)
            configs["config2"] = config2_leg(ctx)
            configs["config4"] = config4_leg(lib, np, C)
            configs["config5"] = config5_leg(ctx, sizes=(256, N) if N >= 256 else (N,))

    yard = None
    if rank == 0 and world == 1 and not args.no_extras:
        yard = box_yardstick(lib, np, C, check, stream, timer, n_local)
        if yard and yard.get("rate_GBs"):
            roofline["frac_of_box_yardstick"] = round(roofline["achieved"] / yard["rate_GBs"], 4)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(np, N)

    if rank == 0:
        out = {
            "metric": "SpMV GFLOP/s, CSR, 3-D 7-pt Poisson n=512^3" if (N == 512 and not slab) else f"SpMV GFLOP/s, CSR, 3-D 7-pt Poisson n={L}x{N}x{N}" if slab else f"SpMV GFLOP/s, CSR, 3-D 7-pt Poisson n={N}^3",
            "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"3-D 7-point Poisson {N}^3" if L == N else f"3-D 7-point Poisson {N}^3 per GPU (grid {N} x {N} x {L})") +
                                   ", CSR f64/i32 (test3.c entry order), y=A*x via lis_matvec streaming the reference's own value[] / index[] / ptr[] arrays (12 B per non-zero + 20 B per row), "
                                   "x_i = frac(i * 0.618...) - 0.5",
                       "n": n_global, "nnz": nnz_global, "mode": "lis_amd_set_reference_layout(1)",
                       "parallelism": f"row-block x{world}" + ((" + RCCL halo" if comm_used == "rccl" else " + gloo-callback halo (bring-up, not a measurement)") if world > 1 else "")},
            "roofline": roofline,
            "x_equals_one": x_one,
            "krylov": krylov,
            "structured_fast_path": fast,
            "configs": configs,
            "reading_guide": ("`value` / `ms_per_step` / `roofline`: lis_matvec on the 512^3 CSR matrix in the reference layout mode -- spmv_csr_rowgather_kernel streams the reference's own "
                              "index[] / value[] / ptr[] arrays, SURVEY 8d's 12 B per non-zero + 20 B per row, on a non-trivial x; `frac` is exactly those bytes / the kernel's HIP-event time / "
                              "8 TB/s, <= 1 by construction, and `traffic` is this run's own PMC count of the same kernel.  `krylov`: the solver loops in the same mode.  "
                              "`structured_fast_path`: what the library does for THIS matrix by default (a constant-coefficient box stencil: matrix-free marching) -- the same bits, "
                              "priced on its own bytes, not a CSR-roofline claim.  `configs`: one leg per BASELINE.json config.  Every `frac` in the line is bytes-the-kernel-is-asked-to-move / "
                              "its measured time / 8 TB/s (0 < frac <= 1, asserted); `contract_frac` prices SURVEY 8d's layout over the same time and exceeds 1 exactly where a derived form reads fewer bytes."),
            "setup": {"generate_and_plan_ms": round(setup_ms, 1), "generate_ms": None if gen_ms is None else round(gen_ms, 1),
                      "plan_build_ms": None if gen_ms is None else round(max(setup_ms - gen_ms, 0.0), 1),
                      "plan_build_in_products": None if gen_ms is None else round(max(setup_ms - gen_ms, 0.0) / max(ms_per_step, 1e-9), 1),
                      "note": "one-off per matrix, outside the timed region (the reference pays its own at lis_matrix_assemble / _convert): the matrix generated in HBM, then "
                              "the DEFAULT plan -- merge-path row split, one-byte column codes, row patterns, value records, the dominant pattern (DESIGN.md 4); "
                              "plan_build_in_products = how many timed (reference-layout) products it costs"},
            "preroll": args.preroll,          # untimed clock-ramp launches before the W warm-up steps (a cold process: +7 %)
            "degraded": bool(world > 1 and comm_used != "rccl"),   # True: the RCCL communicator could not be formed, NOT a measurement
            "rccl_ranks": world if (world > 1 and comm_used == "rccl" and int(dll.lis_amd_comm_kind()) == 1) else (0 if world > 1 else None),
            "self_launched": os.environ.get("LIS_AMD_BENCH_SELF_LAUNCHED") == "1",
            "multi_gpu": multi,
            "box_yardstick": yard,            # what a pure streaming kernel with the product's read : write ratio reaches on THIS box (no gather, no index): the practical ceiling behind `roofline.frac`
            "stencil27": extras,              # beside the headline: the 27-point stencil (spmvtest3b / HPCG) through the round-3 kernels; not part of `value`
            "cpu_baseline": cpu,
        }
        # the HEADLINE's fraction is a hard rule (SURVEY 8d's bytes over the timed kernel: above 1 the kernel is not doing the work); a side leg whose working set sits in
        # the 256 MB Infinity Cache (a slab run, a small --grid) can legitimately beat the HBM peak: reported in the line, not fatal to it
        assert_fracs_physical({"roofline": out["roofline"]}, shared_gpu=out["degraded"])
        assert alg_bytes / (ms_per_step * 1e-3) <= HBM_PEAK_GBS * 1e9, "SURVEY 8d bytes / ms_per_step exceeds the HBM peak: the timed kernel is not doing the work"
        bad = []
        collect_unphysical_fracs(out, "line", bad, shared_gpu=out["degraded"])
        out["fracs_outside_0_1"] = bad        # [] in every run at the headline size
        if bad:
            print(f"bench.py: fractions outside (0, 1] in side legs (cache-resident working sets?): {bad}", file=sys.stderr, flush=True)
        emit(out)
    degraded = world > 1 and comm_used != "rccl" and args.comm == "rccl"
    if world > 1:
        dll.lis_amd_comm_finalize()
        dist.destroy_process_group()
    if degraded:
        sys.exit(3)                   # the line above says "degraded": true; a driver that only looks at the exit code sees it too


# ============================================================================================================================ per-config legs
def config1_leg(ctx):
    """BASELINE config 1 = test/spmvtest1.c: the 1-D 3-point matrix (2 on the diagonal, -1 beside it, test/spmvtest1.c:139-146) of n = 10000 rows, y = A*1 in every
    storage format the driver walks that this library serves, `iter` products each, the 2-norm it prints (sqrt 2: 1 at both ends, 0 inside) and MFLOPS = 2 nnz iter / t
    (:225).  A plumbing case: 30 kB of matrix, latency-bound -- no roofline fraction."""
    lib, np, capi = ctx["lib"], ctx["np"], ctx["capi"]
    try:
        import lisdrv
        n, iters = 10000, 100
        idx = np.stack([np.arange(n) - 1, np.arange(n), np.arange(n) + 1], axis=1).ravel()
        val = np.tile(np.array([-1.0, 2.0, -1.0]), n)
        keep = (idx >= 0) & (idx < n)
        idx, val = idx[keep].astype(np.int32), val[keep]
        ptr = np.concatenate([[0], np.cumsum(keep.reshape(n, 3).sum(axis=1))]).astype(np.int32)
        A0 = lisdrv.make_csr(lib, ptr, idx, val)
        nnz = int(ptr[-1])
        out = {"name": BASELINE_CONFIGS["config1"], "n": n, "nnz": nnz, "iter": iters, "formats": {}}
        nrm = C.c_double()
        for fmt in ("csr", "csc", "dia", "ell", "jad", "bsr"):
            M = A0 if fmt == "csr" else lisdrv.convert(lib, A0, fmt)
            x, y = ctx["vec"](M), ctx["vec"](M)
            assert lib.lis_vector_set_all(1.0, x) == 0
            for _ in range(5):
                assert lib.lis_matvec(M, x, y) == 0
            ctx["dll"].lis_amd_synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                assert lib.lis_matvec(M, x, y) == 0
            ctx["dll"].lis_amd_synchronize()
            el = time.perf_counter() - t0
            assert lib.lis_vector_nrm2(y, C.byref(nrm)) == 0
            ok = abs(nrm.value - 2.0 ** 0.5) <= 1e-14
            out["formats"][fmt.upper()] = {"two_norm": nrm.value, "two_norm_is_sqrt2": ok, "mflops": round(2.0 * nnz * iters * 1e-6 / el, 1), "us_per_product": round(el / iters * 1e6, 2)}
            if not ok:
                out["error"] = f"{fmt}: 2-norm {nrm.value!r} != sqrt(2)"
            lib.lis_vector_destroy(x); lib.lis_vector_destroy(y)
            if M is not A0:
                lib.lis_matrix_destroy(M)
        lib.lis_matrix_destroy(A0)
        out["note"] = "spmvtest1's own report: `2-norm = 1.414214e+00` for every format; products are queued, the clock stops behind one synchronize"
        return out
    except Exception as exc:
        return {"name": BASELINE_CONFIGS["config1"], "error": f"{type(exc).__name__}: {exc}"}


def fmt_bytes(fmt, n, nnz, width):
    """the algorithmic bytes of one product in the format's own (reference) layout"""
    if fmt == "csr":
        return 12 * nnz + 20 * n + 4
    if fmt == "ell":
        return 12 * width * n + 16 * n
    if fmt == "dia":
        return 8 * width * n + 16 * n
    raise KeyError(fmt)


def format_legs(ctx, G, fmts, cg_iters, reference):
    """CSR / ELL / DIA of the G^3 7-point matrix (spmvtest3's sorted rows: DIA adds a row's terms by ascending offset, so all three give the same bits), each through
    lis_matvec on a non-trivial x and through CG + Jacobi.  reference=True: the reference layout mode -- native kernels on the formats' own arrays, `frac` on SURVEY 8d's
    bytes (CSR 12 nnz + 20 n, ELL 100 n, DIA 72 n for this stencil).  reference=False: the default forms (constant coefficients: all formats collapse onto the row form
    and its marching kernel -- reported as times and contract_frac, which exceeds 1 there because the arrays are not read)."""
    lib, dll, capi, np = ctx["lib"], ctx["dll"], ctx["capi"], ctx["np"]
    import lisdrv
    n, nnz = G ** 3, 7 * G ** 3 - 6 * G * G
    ctx["reference_layout"](reference)
    out = {}
    try:
        As = ctx["poisson"](G, G, G, 1)
        x, y, yref, b, sol = (ctx["vec"](As) for _ in range(5))
        ctx["golden_x"](x, n)
        assert dll.lis_amd_vector_poisson3d_rhs(b, G, G, G) == 0
        nrm = ctx["nrm"]
        for fmt in fmts:
            M = As if fmt == "csr" else lisdrv.convert(lib, As, fmt)
            width = {"csr": 0, "ell": M.contents.maxnzr, "dia": M.contents.nnd}[fmt]
            e = ctx["leg"](M, x, y, ctx["steps"], nnz, collective=False)
            if fmt == "csr":
                assert lib.lis_vector_copy(y, yref) == 0
            else:
                assert lib.lis_vector_axpy(-1.0, yref, y) == 0 and lib.lis_vector_nrm2(y, C.byref(nrm)) == 0
                e["same_bits_as_csr"] = nrm.value == 0.0
            B = fmt_bytes(fmt, n, nnz, width)
            sec = e["kernel_ms"] * 1e-3
            dtype = int(dll.lis_amd_matrix_device_type(M))
            native = dtype == {"csr": capi.LIS_MATRIX_CSR, "ell": capi.LIS_MATRIX_ELL, "dia": capi.LIS_MATRIX_DIA}[fmt]
            e.update({"contract_bytes_per_launch": B, "width": width, "device_layout": {capi.LIS_MATRIX_CSR: "CSR rows", capi.LIS_MATRIX_ELL: "ELL", capi.LIS_MATRIX_DIA: "DIA"}.get(dtype, dtype),
                      "value_records": int(dll.lis_amd_matrix_value_records(M)), "marching": int(dll.lis_amd_matrix_marching(M)) if dtype == capi.LIS_MATRIX_CSR else 0})
            if reference:
                assert native and e["value_records"] == 0, "the reference layout mode must keep the format's own arrays"
                e["kernel"] = {"csr": "spmv_csr_rowgather_kernel", "ell": "spmv_ell_kernel", "dia": "spmv_dia_kernel"}[fmt]
                e["frac"] = round(B / sec / 1e9 / HBM_PEAK_GBS, 4)
                form = (0, 0, 0)
            else:
                e["contract_frac"] = round(B / sec / 1e9 / HBM_PEAK_GBS, 4)
            # CG + Jacobi: to convergence when cg_iters is None (the reference's count is the check), else cg_iters timed iterations
            S = capi.PS()
            assert lib.lis_solver_create(C.byref(S)) == 0
            assert lib.lis_solver_set_option(b"-i cg -p jacobi -tol 1e-12 -maxiter 20", S) == 0 and lib.lis_solve(M, b, sol, S) == 0      # warm-up
            assert lib.lis_solver_set_option(f"-maxiter {cg_iters or 3000}".encode(), S) == 0
            assert lib.lis_solve(M, b, sol, S) == 0
            it, itime = S.contents.iter, S.contents.itime
            if cg_iters:
                it = min(it, cg_iters)                    # (a solve that stops at maxiter reports maxiter + 1)
            cg = {"iters_per_sec": round(it / itime, 2) if itime > 0 else None, "iter": it, "itime_s": round(itime, 6), "status": S.contents.retcode, "rel_residual": S.contents.resid,
                  "to_convergence": cg_iters is None}
            if reference and itime > 0:
                uniform = int(dll.lis_amd_last_solve_uniform_jacobi())
                loop_b = B - 4 * (fmt == "csr") + (CG_VECTOR_BYTES_PER_ROW - (16 if uniform else 0)) * n
                cg["loop_bytes_per_iter"] = loop_b
                cg["frac"] = round(loop_b / (itime / max(1, it)) / 1e9 / HBM_PEAK_GBS, 4)
                cg["contract_bytes_per_iter"] = B + 136 * n
            e["cg_jacobi"] = cg
            lib.lis_solver_destroy(S)
            out[fmt.upper()] = e
            if M is not As:
                lib.lis_matrix_destroy(M)
        for v in (x, y, yref, b, sol):
            lib.lis_vector_destroy(v)
        lib.lis_matrix_destroy(As)
    finally:
        ctx["reference_layout"](False)
    return out


def config2_leg(ctx):
    """BASELINE config 2: 256^3 CSR, CG + Jacobi to convergence (tol 1e-12: 764 iterations, the reference's count in every format and at every thread count,
    tests/golden/known_answers.json), in the reference layout mode and in the default form."""
    try:
        G = 256 if ctx["grid"] >= 256 else ctx["grid"]
        out = {"name": BASELINE_CONFIGS["config2"], "grid": f"{G}^3", "n": G ** 3, "nnz": 7 * G ** 3 - 6 * G * G,
               "reference_iterations": 764 if G == 256 else None}
        out["reference_layout"] = format_legs(ctx, G, ("csr",), None, True)["CSR"]
        out["default_form"] = format_legs(ctx, G, ("csr",), None, False)["CSR"]
        for k in ("reference_layout", "default_form"):
            cg = out[k]["cg_jacobi"]
            if G == 256 and (cg["iter"] != 764 or cg["status"] != 0):
                out["error"] = f"{k}: CG + Jacobi took {cg['iter']} iterations (status {cg['status']}), the reference takes 764"
        return out
    except Exception as exc:
        return {"name": BASELINE_CONFIGS["config2"], "error": f"{type(exc).__name__}: {exc}"}


def config5_leg(ctx, sizes):
    """BASELINE config 5: the format sweep -- CSR / ELL / DIA at 256^3 (the config's size; x + y = 268 MB sit at the edge of the 256 MB Infinity Cache) and at the
    headline's 512^3 (nothing fits), native kernels on SURVEY 8d's bytes and the default forms; CG + Jacobi each (256^3: to convergence = 764 iterations; 512^3: 100 timed)."""
    out = {"name": BASELINE_CONFIGS["config5"]}
    for G in sizes:
        try:
            full = G <= 256
            out[f"{G}^3"] = {"n": G ** 3, "nnz": 7 * G ** 3 - 6 * G * G,
                             "native_kernels_reference_layout": format_legs(ctx, G, ("csr", "ell", "dia"), None if full else 100, True),
                             "default_forms": format_legs(ctx, G, ("csr", "ell", "dia"), None if full else 100, False)}
            if G == 256:
                for mode in ("native_kernels_reference_layout", "default_forms"):
                    for fmt, e in out["256^3"][mode].items():
                        if e["cg_jacobi"]["iter"] != 764:
                            out["error"] = f"{mode} {fmt}: {e['cg_jacobi']['iter']} iterations, the reference takes 764"
        except Exception as exc:
            out[f"{G}^3"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def stencil27_leg(lib, np, C, stream, G=256, launches=30):
    """Beside the headline matrix: the 27-point stencil at G^3 through the same plan API the library uses, with constant coefficients (26 / -1: the matrix of the
    reference's spmvtest3b and of HPCG -- wide value records, x staged per wavefront, the dominant pattern in scalar registers) and with every row's values its
    own (8 B per non-zero streamed, four lanes per row).  G = 256: 16.8 M rows, x + y + pattern bytes = 285 MB -- beyond the 256 MB Infinity Cache (round 3 ran
    160^3 = 70 MB, a cache-resident working set and not a roofline figure).  HIP-event ms per launch; never fatal: None when anything goes wrong."""
    try:
        from lis_amd import DeviceArray as DA, check
        n = G ** 3
        nnz = (3 * G - 2) ** 3
        d = np.array([-1, 0, 1], np.int32)
        offs = ((d[:, None, None] * G + d[None, :, None]) * G + d[None, None, :]).reshape(27)
        const = np.where(offs == 0, 26.0, -1.0)
        dptr, didx = DA(n + 1, np.int32), DA(nnz, np.int32)
        dval_c, dval_v = DA(nnz, np.float64), DA(nnz, np.float64)
        rng = np.random.default_rng(7)
        ptr_base, planes = 0, max(1, (1 << 20) // (G * G))            # ~1 M rows per piece: the host never holds more than that of the matrix
        check(lib.liship_memset(dptr.ptr, 0, 4, None))
        for z0 in range(0, G, planes):
            z1 = min(G, z0 + planes)
            z, y, x = np.meshgrid(np.arange(z0, z1, dtype=np.int32), np.arange(G, dtype=np.int32), np.arange(G, dtype=np.int32), indexing="ij")
            z, y, x = z.ravel(), y.ravel(), x.ravel()
            rows = len(z)
            inz, iny, inx = ((c[:, None] + d >= 0) & (c[:, None] + d < G) for c in (z, y, x))
            mask = (inz[:, :, None, None] & iny[:, None, :, None] & inx[:, None, None, :]).reshape(rows, 27)
            cols = (np.arange(z0 * G * G, z1 * G * G, dtype=np.int32))[:, None] + offs[None, :]
            idx = np.ascontiguousarray(cols[mask])                        # row by row, ascending columns
            pp = (ptr_base + np.cumsum(mask.sum(axis=1))).astype(np.int32)
            vc = np.ascontiguousarray(np.broadcast_to(const, (rows, 27))[mask])
            vv = vc * rng.uniform(0.5, 1.5, len(vc))
            check(lib.liship_memcpy_h2d(dptr.ptr + 4 * (z0 * G * G + 1), pp.ctypes.data, pp.nbytes, None))
            check(lib.liship_memcpy_h2d(didx.ptr + 4 * ptr_base, idx.ctypes.data, idx.nbytes, None))
            check(lib.liship_memcpy_h2d(dval_c.ptr + 8 * ptr_base, vc.ctypes.data, vc.nbytes, None))
            check(lib.liship_memcpy_h2d(dval_v.ptr + 8 * ptr_base, vv.ctypes.data, vv.nbytes, None))
            check(lib.liship_device_synchronize())
            ptr_base += len(idx)
        assert ptr_base == nnz
        xh = np.modf(np.arange(n, dtype=np.float64) * 0.6180339887498949)[0] - 0.5
        xv, yv = DA.from_host(xh, np.float64), DA(n, np.float64)
        del xh
        timer, ev = C.c_void_p(), C.c_float()
        check(lib.liship_timer_create(C.byref(timer)))
        out = {"grid": f"{G}^3", "n": n, "nnz": int(nnz), "launches": launches, "x": "x_i = frac(i * 0.618...) - 0.5"}
        for key, dval in (("constant_coefficients", dval_c), ("varying_coefficients", dval_v)):
            plan = C.c_void_p()
            check(lib.liship_csr_plan_create(C.byref(plan), n, dptr.ptr, stream))
            check(lib.liship_csr_plan_encode_indices(plan, dptr.ptr, didx.ptr, stream))
            check(lib.liship_csr_plan_encode_row_patterns(plan, dptr.ptr, stream))
            check(lib.liship_csr_plan_encode_row_values(plan, dptr.ptr, dval.ptr, stream))
            for _ in range(10):
                check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, xv.ptr, yv.ptr, stream))
            check(lib.liship_timer_start(timer, stream))
            for _ in range(launches):
                check(lib.liship_spmv_csr_f64(plan, dptr.ptr, didx.ptr, dval.ptr, xv.ptr, yv.ptr, stream))
            check(lib.liship_timer_stop(timer, stream))
            check(lib.liship_stream_synchronize(stream))
            check(lib.liship_timer_elapsed_ms(timer, C.byref(ev)))
            ms = ev.value / launches
            wide = int(lib.liship_csr_plan_wide_dominant(plan))
            box27 = int(lib.liship_csr_plan_box27(plan))           # round 5: the grid is a box and the product walks its planes (x and y alone: 16 B per row)
            stored = 16 * n if box27 else 17 * n if wide else 8 * nnz + 17 * n          # (one pattern byte,) y, the compulsory x per row (+ the streamed values)
            out[key] = {"kernel_ms": round(ms, 4), "gflops": round(2e-6 * nnz / ms, 1), "stored_bytes_per_launch": int(stored),
                        "working_set_note": "x (8 B per row) alone still fits the 256 MB Infinity Cache at this size; the bytes of one launch do not" if 8 * n <= (256 << 20) < stored else None,
                        "frac": round(stored / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "kernel": "spmv_csr_box27_march_kernel" if box27 else "spmv_csr_valuerecw_staged_kernel" if wide else ("spmv_csr_pattern_team_staged_kernel" if int(lib.liship_csr_plan_team_form(plan)) == 2 else "other")}
            check(lib.liship_csr_plan_destroy(plan))
        return out
    except Exception as exc:                                          # an extra: its failure must not cost the line
        return {"error": f"{type(exc).__name__}: {exc}"}


def config4_leg(lib, np, C, reps=50):
    """BASELINE config 4 (SuiteSparse Queen_4147, irregular CSR with long rows, GMRES(30)) through lis_input, lis_matvec and lis_solve, as a Lis program would run it.
    The file: LIS_AMD_BENCH_MTX=/path/to/file.mtx when somebody has one (Queen_4147.mtx itself, or any Matrix Market file), else the STAND-IN -- Queen_4147 cannot be
    fetched here: tests/golden/gen_queen_class.c writes a 3-dof mesh of its size (4.1 M rows, 2.9e8 non-zeros, node numbers scrambled inside runs of 1024) as a symmetric
    coordinate file on this box.  Reported: the reader, upload + plan, the product in the caller's numbering (contract_frac on 12 B per non-zero + 20 B per row; sha256 of
    y for a check against any other implementation), GMRES(30) / BiCGSTAB / CG + Jacobi / BiCG with iterations, it/s and time to solution.  The renumbered form
    (liship_csr_plan_reorder) is LAZY since round 6 -- none of these first solves triggers it; its cost (`reorder_s`: the numbering found on the device, P A P^T and its plan built in HBM), its gain and the break-even
    iteration count are measured beside them by asking for it explicitly.  Host-clock ms per product over `reps` calls behind one synchronize; never fatal: an error string instead."""
    path, made = os.environ.get("LIS_AMD_BENCH_MTX"), False
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        from lis_amd import _capi as capi
        dll = lib.dll
        out = {"name": BASELINE_CONFIGS["config4"]}
        if path:
            out.update({"stand_in": False, "matrix": f"LIS_AMD_BENCH_MTX={path}, through lis_input"})
        else:
            import queen_class
            t0 = time.time(); path, rows, stored = queen_class.generate("full"); made = True
            out.update({"stand_in": True, "generate_s": round(time.time() - t0, 2),
                        "matrix": "STAND-IN for Queen_4147 (not fetchable here; LIS_AMD_BENCH_MTX=/path/file.mtx runs a real file): tests/golden/gen_queen_class.c 111 1024 -- 3 unknowns per "
                                  "node of a 111^3 grid, 27-node connectivity, scrambled numbering -- a symmetric .mtx through lis_input"})
        A, b, x0 = capi.PM(), capi.PV(), capi.PV()
        assert lib.lis_matrix_create(0, C.byref(A)) == 0 and lib.lis_vector_create(0, C.byref(b)) == 0 and lib.lis_vector_create(0, C.byref(x0)) == 0
        dll.lis_amd_synchronize()
        t0 = time.time(); assert lib.lis_input(A, b, x0, path.encode()) == 0; dll.lis_amd_synchronize(); t_read = time.time() - t0
        if made:
            os.unlink(path)
        path = None
        dll.lis_amd_matrix_local_columns.argtypes = [capi.PM]
        dll.lis_amd_matrix_reordered.argtypes = [capi.PM]; dll.lis_amd_matrix_reordered.restype = C.c_longlong
        dll.lis_amd_matrix_upload.argtypes = [capi.PM]
        dll.lis_amd_set_reorder_after.argtypes = [C.c_longlong]
        t0 = time.time(); assert dll.lis_amd_matrix_upload(A) == 0; dll.lis_amd_synchronize(); t_plan = time.time() - t0   # (zero when lis_input already uploaded and planned: resident mode does)
        tot, asm = C.c_double(), C.c_double()
        assert dll.lis_amd_last_input_times(C.byref(tot), C.byref(asm)) == 0
        t_reader, t_plan = max(0.0, t_read - asm.value), t_plan + asm.value      # lis_input's assemble step IS the upload + plan in resident mode
        out.update({"lis_input_s": round(t_read, 2), "reader_s": round(t_reader, 2), "upload_and_plan_s": round(t_plan, 3),
                    "lis_input_note": "lis_input_s = reader_s (file read, parse, symmetric expansion, rows placed: host) + upload_and_plan_s (the HBM copy and its plan, inside lis_input's assemble step)"})
        irregular_measure(lib, np, C, A, out, reps)
        for v in (b, x0):
            lib.lis_vector_destroy(v)
        lib.lis_matrix_destroy(A)
        if not os.environ.get("LIS_AMD_BENCH_MTX"):
            out["unstructured_mesh_class"] = mesh_leg(lib, np, C, reps)
        return out
    except Exception as exc:                                          # an extra: its failure must not cost the line
        return {"name": BASELINE_CONFIGS["config4"], "error": f"{type(exc).__name__}: {exc}"}
    finally:
        if made and path and os.path.exists(path):
            os.unlink(path)


def mesh_leg(lib, np, C, reps=50, nodes=4000000):
    """Beside the Queen-class stand-in (long rows, 3 unknowns per node): the irregular class none of the plan's special forms catch -- an unstructured 3-D mesh, ONE unknown per
    node, ragged rows of 7 .. 32 entries (mean 18), varying coefficients, numbered along a coarse Morton curve and at random inside a cell (tests/orc.py unstructured_mesh;
    reference goldens at 60 000 nodes in tests/test_configs_gpu.py).  Round 6: short rows try block-local columns, and a plan whose lists fail is renumbered on the device."""
    try:
        import lisdrv
        import orc
        t0 = time.time()
        ptr, idx, val = orc.unstructured_mesh(nodes)
        out = {"matrix": f"tests/orc.py unstructured_mesh({nodes}): k-nearest-neighbour graph of random points in the unit cube, symmetrised, 4096 Morton cells, random order inside a cell",
               "generate_s": round(time.time() - t0, 1)}
        A = lisdrv.make_csr(lib, ptr, idx, val)
        irregular_measure(lib, np, C, A, out, reps, rhs_of_ones=False)
        lib.lis_matrix_destroy(A)
        return out
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"}


def irregular_measure(lib, np, C, A, out, reps, rhs_of_ones=True):
    """the product in the caller's numbering, four solvers with time to solution, the renumbered form asked for explicitly (config4_leg, mesh_leg)"""
    import hashlib
    from lis_amd import _capi as capi
    dll = lib.dll
    dll.lis_amd_matrix_local_columns.argtypes = [capi.PM]
    dll.lis_amd_matrix_reordered.argtypes = [capi.PM]; dll.lis_amd_matrix_reordered.restype = C.c_longlong
    dll.lis_amd_set_reorder_after.argtypes = [C.c_longlong]
    if True:
        n, nnz = A.contents.n, A.contents.nnz
        listed = int(dll.lis_amd_matrix_local_columns(A))
        vx, vy, ones = capi.PV(), capi.PV(), capi.PV()
        for v in (vx, vy, ones):
            assert lib.lis_vector_duplicate(A, C.byref(v)) == 0
        xs = np.cos(np.arange(n) * 0.01) + 1.25
        assert lib.lis_vector_set_values(0, n, np.arange(n, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int)), xs.ctypes.data_as(capi.P_DBL), vx) == 0
        assert lib.lis_vector_set_all(1.0, ones) == 0

        def timed():
            for _ in range(10):
                assert lib.lis_matvec(A, vx, vy) == 0
            dll.lis_amd_synchronize()
            t0 = time.time()
            for _ in range(reps):
                assert lib.lis_matvec(A, vx, vy) == 0
            dll.lis_amd_synchronize()
            return (time.time() - t0) / reps * 1e3
        ms = timed()
        yh = np.empty(n)
        assert lib.lis_vector_get_values(vy, 0, n, yh.ctypes.data_as(capi.P_DBL)) == 0
        out.update({"n": n, "nnz": nnz,
                    "block_local_columns_listed": listed, "kernel": "spmv_csr_local_kernel" if listed else "spmv_csr_rowgather_kernel / spmv_csr_products_kernel (the plan's choice)",
                    "spmv_ms": round(ms, 4), "spmv_gflops": round(2.0 * nnz / ms / 1e6, 1),
                    "contract_bytes": 12 * nnz + 20 * n, "contract_frac": round((12.0 * nnz + 20.0 * n) / (ms * 1e-3) / 8e12, 4),
                    "x": "x_i = cos(0.01 i) + 1.25", "y_sha256": hashlib.sha256(yh.tobytes()).hexdigest(),
                    "numbering": "the caller's (the renumbered form is lazy: lis_amd_set_reorder_after, default 4096 products)"})
        rhs = capi.PV()
        assert lib.lis_vector_duplicate(A, C.byref(rhs)) == 0 and lib.lis_matvec(A, ones if rhs_of_ones else vx, rhs) == 0          # b = A*1 (test/test1.c:138-139); the mesh: b = A*x_s
        out["rhs"] = "b = A*1 (test/test1.c:138-139)" if rhs_of_ones else "b = A*x_s, x_s the product's x (with b = A*1 the Jacobi-preconditioned residual of this diagonally weighted matrix is a multiple of the solution: one iteration)"
        SOLVES = ("-i gmres -restart 30 -p none", "-i bicgstab -p none", "-i cg -p jacobi", "-i bicg -p none")       # (BiCG: Lis's default solver, lis_solver.c:242)

        def solve(opts):
            S = capi.PS()
            assert lib.lis_solver_create(C.byref(S)) == 0 and lib.lis_solver_set_option((opts + " -tol 1e-12 -maxiter 2000 -print none").encode(), S) == 0
            assert lib.lis_vector_set_all(0.0, vy) == 0
            dll.lis_amd_synchronize()
            t0 = time.time()
            assert lib.lis_solve(A, rhs, vy, S) == 0
            dll.lis_amd_synchronize()
            wall = time.time() - t0
            r = {"iter": S.contents.iter, "status": S.contents.retcode, "rel_residual": S.contents.resid,
                 "iters_per_sec": round(S.contents.iter / S.contents.itime, 1) if S.contents.itime > 0 else None,
                 "time_to_solution_s": round(wall, 4), "renumbered": int(dll.lis_amd_last_solve_renumbered())}
            lib.lis_solver_destroy(S)
            return r
        out["solves"] = {}
        for opts in SOLVES:
            first = solve(opts)             # the FIRST solve of its kind pays the one-off costs (work-vector pool; BiCG: the transposed copy built in HBM): that is a program's time to solution
            again = solve(opts)
            again["first_solve_time_to_solution_s"] = first["time_to_solution_s"]
            again["first_solve_iters_per_sec"] = first["iters_per_sec"]
            out["solves"][opts] = again
        # ---- the renumbered form, asked for explicitly: what it costs, what it gains, when it pays
        dll.lis_amd_set_reorder_after(1)
        try:
            trig = solve("-i cg -p jacobi")                           # this solve builds P A P^T first (reorder_s = what that added to its time to solution) ...
            reordered = int(dll.lis_amd_matrix_reordered(A))
            ren = {"listed_after_reordering": reordered, "built": bool(reordered)}
            if reordered:
                base = out["solves"]["-i cg -p jacobi"]
                ren["reorder_s"] = round(max(0.0, trig["time_to_solution_s"] - base["time_to_solution_s"]), 3)
                ren["solves"] = {opts: solve(opts) for opts in SOLVES[:3]}     # ... and these iterate on it
                gain = 1.0 / base["iters_per_sec"] - 1.0 / ren["solves"]["-i cg -p jacobi"]["iters_per_sec"]
                ren["cg_jacobi_seconds_saved_per_iteration"] = round(gain, 7)
                ren["break_even_iterations"] = int(ren["reorder_s"] / gain) if gain > 0 else None
                ren["policy"] = ("lazy: the first lis_solve that finds 4096 products served by the plan builds it (lis_amd_set_reorder_after / LIS_AMD_REORDER_AFTER; 0 = at plan time); "
                                 "single products always keep the caller's numbering")
            out["renumbered_form"] = ren
        finally:
            dll.lis_amd_set_reorder_after(4096)
        for v in (vx, vy, ones, rhs):
            lib.lis_vector_destroy(v)


def assert_fracs_physical(node, path="line", shared_gpu=False):
    """every `frac` in the line is a fraction of the HBM peak on bytes a kernel moves over its measured time: 0 < frac <= 1
    (shared_gpu: a bring-up run of several ranks on ONE GPU, flagged degraded -- the other process's time slices fall inside the event intervals of a 48^3 kernel
    and a fraction may round to 0.0000)"""
    if isinstance(node, dict):
        for k, v in node.items():
            if k == "frac":
                assert v is not None and (0.0 < v or (shared_gpu and v == 0.0)) and v <= 1.0, f"{path}.frac = {v!r} is not a physical fraction of the roofline"
            else:
                assert_fracs_physical(v, f"{path}.{k}", shared_gpu)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            assert_fracs_physical(v, f"{path}[{i}]", shared_gpu)


def collect_unphysical_fracs(node, path, bad, shared_gpu=False):
    if isinstance(node, dict):
        for k, v in node.items():
            if k == "frac":
                if not (v is not None and (0.0 < v or (shared_gpu and v == 0.0)) and v <= 1.0):
                    bad.append([path, v])
            else:
                collect_unphysical_fracs(v, f"{path}.{k}", bad, shared_gpu)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            collect_unphysical_fracs(v, f"{path}[{i}]", bad, shared_gpu)


def spmv_stored_bytes(n, nnz, coded, patterns, values=0):
    """bytes one product is asked to stream: values + column information + row starts + y + the compulsory x"""
    if patterns and values:
        return (1 + 8 + 8) * n + 4                         # value records: one pattern byte per row, y, the compulsory x
    if patterns:
        return 8 * nnz + (1 + 8 + 8) * n + 4               # one pattern byte per row (row starts: a scan of the pattern lengths)
    return (9 if coded else 12) * nnz + 20 * n + 4


def krylov_bytes(key, iters, n, nnz, coded, patterns=0, values=0, uniform_jacobi=0):
    """Bytes per iteration and per GPU: (what the passes of the fused device loops are asked to stream, what the
    reference's unfused operator sequence moves by SURVEY 8d's count).  n / nnz are the local rows / non-zeros.
    Product: S = (9 coded | 12) B per non-zero + 20 B per row (ptr, y, compulsory x), or 8 B per non-zero + 17 B per row with row
    patterns, 17 B per row alone with value records; contract B = 12 nnz + 20 n.
    Vector passes: every array a pass reads or writes counts 8 B per row once."""
    S = spmv_stored_bytes(n, nnz, coded, patterns, values) - 4
    B = 12 * nnz + 20 * n
    if key == "cg_jacobi":
        # p = dinv.*r + beta p (+ the deferred x += alpha p: r dinv p x | p x) ; q = A p with <p,q> (w = p: no extra stream) ;
        # r -= alpha q, ||r||^2, <r, dinv.*r> (q r dinv | r)
        # (a constant diagonal: dinv is one double, neither pass reads the array: 16 B per row fewer)
        return S + (CG_VECTOR_BYTES_PER_ROW - (16 if uniform_jacobi else 0)) * n, B + 136 * n
    if key == "bicgstab_none":
        # p-update 32 ; v = A p + <rtld,v> (+8: rtld) ; s = r - alpha v, ||s|| 24 ; t = A s + <t,s>,<t,t> ;
        # x += alpha p + omega s, r = s - omega t, ||r||, <rtld,r> in one pass 56 (t s rtld p x | x r)
        return 2 * S + 120 * n, 2 * B + 248 * n
    if key == "bicg_none":
        # two direction updates 24 + 24 ; q = A p + <p~,q> (+8: p~) ; q~ = A^T p~ ; x,r update 48 ; r~ update + rho 32
        # reference: 2 products, 2 copies (psolve none), 2 dots, 2 xpays, 3 axpys, nrm2 (lis_solver_bicg.c:176-262) = 192 n
        return 2 * S + 136 * n, 2 * B + 192 * n
    if key == "gmres30_none":
        m, loop, contract, done = 30, 0, 0, 0
        while done < iters:
            steps = min(m, iters - done)
            loop += 24 * n                                  # ||v0||, v0 /= ||v0||
            for i in range(1, steps + 1):
                # w = A v_{i-1} + <w,v0> (+8), i-1 chained Gram-Schmidt steps (vprev w vnext | w), the last one with ||w||^2, w /= ||w||
                loop += S + 8 * n + 32 * n * (i - 1) + 24 * n + 16 * n
                contract += B + 40 * n * i + 40 * n         # B_spmv + 40 n i + 40 n
            loop += 8 * n * (steps + 1) + 24 * n + 8 * n * (steps + 2)     # z = sum y_j v_j ; x += z ; v0 += sum g_j v_j
            contract += 16 * n + 24 * n * (steps - 1) + 24 * n + 24 * n * (steps + 1)
            done += steps
        return loop // max(1, iters), contract // max(1, iters)
    raise KeyError(key)


CG_VECTOR_BYTES_PER_ROW = 80      # 48 (x += alpha_prev p, p = dinv.*r + beta p: r dinv p x | p x) + 32 (r -= alpha q with both sums: q r dinv | r)


def box_yardstick(lib, np, C, check, stream, timer, n):
    """The box's streaming yardstick (liship_stream_yardstick): every workgroup sums 13 consecutive 4 KB tiles of one read stream into one 4 KB tile of the write stream --
    the CSR product's own read : write ratio on SURVEY 8d's bytes (13.0 : 1 at 7 entries per row), 14 n * 8 B per launch, nontemporal both ways, no gather and no index.  MI355X boxes of this
    pool differ by several per cent in what they stream; this says what THIS one does, so that `roofline.frac` (priced on the 8 TB/s spec peak) can
    be read against the practical ceiling measured in the same process."""
    from lis_amd import DeviceArray as DA
    try:
        n = int(n) & ~511
        out = {"kernel": "stream_sum_kernel<13>", "reads": 13, "writes": 1, "n": n, "bytes_per_launch": 14 * 8 * n}
        src, dst = DA(13 * n, np.float64), DA(n, np.float64)
        check(lib.liship_memset(src.ptr, 0, src.nbytes, None))
        ms = C.c_float()
        for reads in (13, 8, 1):
            for _ in range(5):
                check(lib.liship_stream_yardstick(reads, n, src.ptr, dst.ptr, 0, stream))
            check(lib.liship_timer_start(timer, stream))
            for _ in range(20):
                check(lib.liship_stream_yardstick(reads, n, src.ptr, dst.ptr, 0, stream))
            check(lib.liship_timer_stop(timer, stream))
            check(lib.liship_device_synchronize())
            check(lib.liship_timer_elapsed_ms(timer, C.byref(ms)))
            rate = (reads + 1) * 8 * n / (ms.value / 20 * 1e-3) / 1e9
            if reads == 13:
                out.update({"kernel_ms": round(ms.value / 20, 4), "rate_GBs": round(rate, 1), "frac_of_spec_peak": round(rate / HBM_PEAK_GBS, 4)})
            elif reads == 8:
                out["rate_8_to_1_GBs"] = round(rate, 1)      # the DIA product's ratio (7 diagonals + x : y)
            else:
                out["copy_1_to_1_GBs"] = round(rate, 1)
        out["note"] = ("a workgroup per 4 KB tile, 16 B per lane, nontemporal loads and stores (persistent grid-stride workgroups measured 5-12 % slower).  Every byte of this "
                       "kernel comes from HBM; the products' x gathers are partly served by the 256 MB Infinity Cache, which is how a product can exceed this rate on SURVEY 8d's bytes")
        src.free(); dst.free()
        return out
    except Exception as e:                                 # a yardstick must never sink the bench line
        return {"rate_GBs": None, "error": str(e)}


def usable_cores(why=None):
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota.  `why` (a dict) receives what limited the count."""
    try:
        cores = len(os.sched_getaffinity(0))
        src = f"affinity mask of {cores} of the machine's {os.cpu_count()} logical CPUs"
    except Exception:
        cores = os.cpu_count() or 1
        src = f"os.cpu_count() = {cores}"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            q = int(int(quota) / int(period))
            if q < cores:
                src = f"cgroup CPU quota {quota}/{period} = {q} cores (affinity mask: {cores}, machine: {os.cpu_count()} logical CPUs)"
            cores = max(1, min(cores, q))
    except Exception:
        pass
    if why is not None:
        why["cores_limited_by"] = src
    return cores


def host_memory_available():
    """bytes this process may still allocate on the host: MemAvailable capped by what the cgroup leaves"""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
                break
    except OSError:
        pass
    try:
        mx = open("/sys/fs/cgroup/memory.max").read().strip()
        cur = int(open("/sys/fs/cgroup/memory.current").read().strip())
        if mx != "max":
            left = int(mx) - cur
            avail = left if avail is None else min(avail, left)
    except (OSError, ValueError):
        pass
    return avail


def cpu_baseline(np, N=512):
    """The reference's OpenMP CSR SpMV (oracle/_ref = Lis 2.1.11 compiled from its own sources) on this box's host cores, on the FULL workload when the host
    has the memory for it (512^3: 11.3 GB of CSR arrays generated straight into the reference's own lis_matrix_malloc_csr arrays, 3 vectors of 1 GB): 30
    products and 10 iterations of its CG + Jacobi (iter / itime, as for the GPU) -- about 15 s of CPU work.  Otherwise (and when that fails) the 256^3 sample
    of rounds 1-3: 1/8 of the rows, 600 products, 40 iterations.  Falls back to the oracle's scalar C port when oracle/_ref is not in the snapshot."""
    import ctypes as C
    import lisdrv
    import orc
    from lis_amd import _capi as capi
    why = {}
    cores = usable_cores(why)
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass

    def reference_run(Nc, reps, cg_iters):
        ref = lisdrv.open_lib(orc.REF_SO, threads=cores)
        n = Nc ** 3
        nnz = 7 * n - 6 * Nc * Nc
        t_gen = time.perf_counter()
        A = capi.PM()
        assert ref.lis_matrix_create(capi.LIS_COMM_WORLD, C.byref(A)) == 0
        assert ref.lis_matrix_set_size(A, n, 0) == 0
        p, i, v = capi.P_INT(), capi.P_INT(), capi.P_DBL()
        assert ref.lis_matrix_malloc_csr(n, nnz, C.byref(p), C.byref(i), C.byref(v)) == 0
        # the oracle's generator (test/test3.c:114-127 restated) writes into the reference's own arrays: no second copy of the matrix on the host
        got = orc.lib().orc_gen_poisson3d(Nc, Nc, Nc, 0, n, 0, np.ctypeslib.as_array(p, shape=(n + 1,)), np.ctypeslib.as_array(i, shape=(nnz,)),
                                          np.ctypeslib.as_array(v, shape=(nnz,)))
        assert got == nnz
        assert ref.lis_matrix_set_csr(nnz, p, i, v, A) == 0 and ref.lis_matrix_assemble(A) == 0
        vx, vy, vb = (capi.PV() for _ in range(3))
        for w in (vx, vy, vb):
            assert ref.lis_vector_duplicate(C.cast(A, C.c_void_p), C.byref(w)) == 0
        assert ref.lis_vector_set_all(1.0, vx) == 0
        gen_s = time.perf_counter() - t_gen
        ref.lis_matvec(A, vx, vb)                         # thread-team start-up and first touch excluded; b = A*1 on the way
        t0 = time.perf_counter()
        for _ in range(reps):
            ref.lis_matvec(A, vx, vy)
        el = time.perf_counter() - t0
        cg = None
        try:                                              # the reference's own CG + Jacobi, iterations per second by its own clock
            S = capi.PS()
            assert ref.lis_solver_create(C.byref(S)) == 0
            assert ref.lis_solver_set_option(f"-i cg -p jacobi -tol 1e-30 -maxiter {cg_iters}".encode(), S) == 0
            assert ref.lis_vector_set_all(0.0, vy) == 0
            ref.lis_solve(A, vb, vy, S)
            it = min(S.contents.iter, cg_iters)
            cg = round(it / S.contents.itime, 2) if S.contents.itime > 0 else None
            ref.lis_solver_destroy(S)
        except Exception:
            cg = None
        for w in (vx, vy, vb):
            ref.lis_vector_destroy(w)
        ref.lis_matrix_destroy(A)
        return nnz, el, cg, gen_s

    try:
        if os.path.exists(orc.REF_SO):
            attempts = []
            need = 16 * (7 * N ** 3) + 40 * N ** 3                  # CSR arrays + vectors + slack
            avail = host_memory_available()
            if avail is not None and avail > 1.3 * need:
                attempts.append((N, 30, 10))
            attempts.append((256, 600, 40))
            last = None
            for Nc, reps, cg_iters in attempts:
                try:
                    t_all = time.perf_counter()
                    nnz, el, cg, gen_s = reference_run(Nc, reps, cg_iters)
                    full = Nc == N
                    return {"value": round(2.0 * nnz * reps / el / 1e9, 3), "unit": "GFLOP/s", "cores": cores, "kind": "reference", "cpu_model": model,
                            "cg_jacobi_iters_per_sec": cg, "full_workload": full, "seconds": round(time.perf_counter() - t_all, 1),
                            "omp_threads": cores, "cores_limited_by": why.get("cores_limited_by"),
                            "sample": (f"the FULL workload: {reps} CSR SpMV on the {Nc}^3 stencil matrix ({el / reps * 1e3:.1f} ms each), lis_matvec of Lis 2.1.11 with OpenMP on {cores} "
                                       f"threads; cg_jacobi_iters_per_sec: {cg_iters} iterations of its lis_solve on the same matrix; matrix built on the host in {gen_s:.1f} s"
                                       if full else
                                       f"{reps} CSR SpMV on the {Nc}^3 stencil matrix (1/8 of the workload's rows: the host lacks the memory for 512^3, or the full run failed), "
                                       f"lis_matvec of Lis 2.1.11 with OpenMP; cg_jacobi_iters_per_sec: {cg_iters} iterations of its lis_solve on the same matrix")}
                except (MemoryError, AssertionError) as e:
                    last = e
            raise RuntimeError(f"reference run failed: {last}")
        Nc, reps = 256, 600
        ptr, idx, val = orc.poisson3d(Nc, Nc, Nc)
        x = np.ones(Nc ** 3)
        orc.spmv_csr(ptr, idx, val, x)
        t0 = time.perf_counter()
        for _ in range(reps):
            orc.spmv_csr(ptr, idx, val, x)
        el = time.perf_counter() - t0
        return {"value": round(2.0 * len(idx) * reps / el / 1e9, 3), "unit": "GFLOP/s", "cores": 1, "kind": "port", "cpu_model": model, "cg_jacobi_iters_per_sec": None,
                "full_workload": False, "sample": f"{reps} CSR SpMV on the {Nc}^3 stencil matrix, scalar C port (oracle/lis_oracle.c)"}
    except Exception as e:                                 # the baseline must never sink the bench line
        return {"value": None, "unit": "GFLOP/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}


if __name__ == "__main__":
    main()
