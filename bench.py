#!/usr/bin/env python3
"""bench.py -- RAS iterations/sec of the MI355X hot path on the 3-D Poisson problem.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without a launcher around it (no WORLD_SIZE in the
environment) starts N rank processes itself -- before anything touches the GPU -- and relays
rank 0's JSON line; under `python -m torch.distributed.run ... bench.py --gpus N` the ranks are
already there and each one runs the same code.  One rank per GPU, halos over RCCL (backend
"nccl"); when fewer GPUs than ranks are visible (rehearsal on a 1-GPU box) the ranks share the
GPU over gloo and the line says so.

A "step" is one outer iteration of SchwarzBase::run (schwarz_base.cpp:387-452):
halo exchange -> boundary update -> convergence check (local residual SpMV +
all-gather of the norms) -> local solve -> restricted write-back, at the
fixed-work operating point of SURVEY 8(d)(i): CG + scalar Jacobi, exactly
`--inner` CG iterations per local solve (local_tol = 0).

N = 1 : BASELINE.json configs[1], 3-D Poisson 256^3, one subdomain on one GPU.
N > 1 : weak scaling with the same 16.8 M rows per GPU: grid 512 x 512 x 64N,
        z-slab partition (regular), overlap 2, RCCL halo exchange; N = 8 is
        BASELINE.json configs[2] (512^3, 8 subdomains).
value = N * (outer iterations / s): subdomain-iterations per second, whole job.

Roofline fields.  `roofline.achieved` prices the dominant launch on the bytes the LAUNCHED matrix
format needs (matrix bytes of the coding the upload chose + every vector of the launch once), so
`frac` <= 1 is an HBM fraction; the same launch priced on uncoded CSR bytes is kept
as `csr_equivalent_x` = how many times faster than an ideal plain-CSR launch it runs.
`roofline_csr_plain` is the plain-CSR SpMV on the same matrix (north_star's ">= 60 %" figure) and
`csr_plain_loop` the whole step with every coding switched off (general-matrix path).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))

try:
    BASELINE_METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
except Exception:
    BASELINE_METRIC = "RAS iterations/sec + time-to-residual(1e-6), 3D Poisson, 1/2/4/8 subdomains"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--inner", type=int, default=10, help="CG iterations per local solve")
    ap.add_argument("--size", type=int, default=256, help="1-GPU grid edge (256 = configs[1])")
    ap.add_argument("--slab", default="512,512,64",
                    help="N>1: per-GPU slab nx,ny,nz (grid = nx x ny x nz*N; default = configs[2] at N=8)")
    ap.add_argument("--spmv-variant", type=int, default=0)
    ap.add_argument("--strong", default=None, metavar="NX,NY,NZ",
                    help="strong scaling: this whole grid (e.g. 512,512,512 = configs[2]) on every N, z-slabs")
    ap.add_argument("--overlapped", action="store_true",
                    help="one-sided overlapped exchange with decentralised convergence (configs[4] flavour)")
    ap.add_argument("--mixed-halo", action="store_true", help="fp32 halo wire format (use_mixed_precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ttr", action="store_true", help="skip the time-to-residual runs")
    ap.add_argument("--no-plain-loop", action="store_true", help="skip the plain-CSR leg of the whole step")
    ap.add_argument("--ttr-subdomains", default="1,2,4,8",
                    help="N=1: time-to-residual(1e-6) of the same grid cut into this many z-slabs, all held by "
                         "this GPU (the iterations-vs-subdomains half of the metric)")
    ap.add_argument("--ttr-budget-s", type=float, default=40.0,
                    help="wall-clock cap of each time-to-residual run (reported as not converged past it)")
    ap.add_argument("--cpu-iters", type=int, default=30,
                    help="outer iterations of the CPU sample (about 10-30 s of CPU work)")
    ap.add_argument("--cpu-subdomains", default="2,4,8",
                    help="N=1: CPU samples of the same grid cut into this many z-slabs as well (SURVEY 8d: P in 1,2,4,8)")
    ap.add_argument("--no-mirror", action="store_true",
                    help="skip the C++ mirror leg (the reference's unchanged bench_ras on libschwz.so)")
    ap.add_argument("--mirror-iters", type=int, default=200)
    ap.add_argument("--no-shapes", action="store_true",
                    help="N=1: plain-CSR SpMV on this grid only, not on the per-GPU slabs of configs[2] / configs[4]")
    ap.add_argument("--csr-shapes", default="512,512,64;1024,1024,128",
                    help="N=1: further per-GPU shapes of the plain-CSR SpMV figure (configs[2] / configs[4] slabs)")
    ap.add_argument("--strong-grid", default="512,512,512",
                    help="grid of the strong-scaling leg every N runs after its weak-scaling leg ('' = skip)")
    ap.add_argument("--strong-steps", type=int, default=10)
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- self launch

def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(n, argv):
    """`python bench.py --gpus N` with no launcher around it: N rank processes of this same
    script, started as CHILDREN (never exec) before this process has imported torch or touched
    the GPU.  Rank 0's stdout is the JSON line and is relayed; the exit code is the worst one.
    Every child is watched while the job runs: when one exits with an error the others are ended
    (they would wait for it in the rendezvous until the process-group timeout), and the whole job
    has a wall-clock cap (SCHWZ_BENCH_LAUNCH_TIMEOUT seconds, default 1500)."""
    import tempfile
    port = free_port()
    procs = []
    cap = float(os.environ.get("SCHWZ_BENCH_LAUNCH_TIMEOUT", "1500"))
    with tempfile.TemporaryFile() as out0:
        for r in range(n):
            env = dict(os.environ)
            env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        deadline = time.time() + cap
        rc = 0
        while True:
            codes = [p.poll() for p in procs]
            failed = [c for c in codes if c not in (None, 0)]
            if failed or all(c is not None for c in codes) or time.time() > deadline:
                if failed:
                    rc = failed[0]
                elif any(c is None for c in codes):
                    rc = 124  # the wall-clock cap
                break
            time.sleep(0.2)
        if rc:
            # rank 0 has been given a few seconds to finish its line; then the exact children started above go
            t_end = time.time() + 5.0
            for p in procs:
                while p.poll() is None and time.time() < t_end:
                    time.sleep(0.1)
                if p.poll() is None:
                    p.kill()
                p.wait()
            sys.stderr.write("bench.py: a rank ended with code %d; the other ranks were stopped\n" % rc)
        out0.seek(0)
        sys.stdout.write(out0.read().decode())
        sys.stdout.flush()
    return rc


def launch_probe(a):
    """SCHWZ_BENCH_LAUNCH_PROBE=1 (tests/test_dist_gloo.py): every rank joins a gloo group under the
    environment self_launch (or an outer launcher) gave it, the ranks are all-gathered and rank 0
    prints them -- the rendezvous of bench.py without the GPU work."""
    if os.environ.get("SCHWZ_BENCH_PROBE_FAIL_RANK") == os.environ.get("RANK"):
        return 3  # tests: this rank dies before the rendezvous, the others would wait for it
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    mine = torch.tensor([dist.get_rank(), int(os.environ["LOCAL_RANK"])], dtype=torch.int64)
    out = [torch.zeros(2, dtype=torch.int64) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    if dist.get_rank() == 0:
        print(json.dumps({"launch_probe": True, "n_gpus": a.gpus, "world": dist.get_world_size(),
                          "ranks": [[int(t[0]), int(t[1])] for t in out]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


# ----------------------------------------------------------------------------- helpers

def make_solver(schwz, comm, shape, inner, tol, max_iters, local_tol, variant, quiet=True, overlapped=False,
                mixed=False):
    s = schwz.Settings(laplacian_dim=3, laplacian_shape=shape, overlap=2,
                       partition=schwz.PARTITION_REGULAR, spmv_variant=variant, use_mixed_precision=mixed)
    s.convergence_settings.enable_global_check = True
    if overlapped:
        # BASELINE configs[4] flavour: halos posted on a side stream and consumed one iteration late,
        # decentralised stop agreement, no collective in the loop
        s.comm_settings.enable_onesided = True
        s.comm_settings.enable_overlap = True
        s.convergence_settings.enable_decentralized_leader_election = True
    m = schwz.Metadata(tolerance=tol, max_iters=max_iters, local_precond="block-jacobi",
                       precond_max_block_size=1, local_solver_tolerance=local_tol,
                       local_max_iters=inner, num_subdomains=comm.size)
    solver = schwz.SolverRAS(s, m, comm=comm, quiet=quiet)
    solver.initialize()
    return solver, m


def sd_csr(sd, schwz):
    """schwz_csr handle of the subdomain's local matrix."""
    import ctypes
    h = ctypes.c_void_p()
    schwz.capi.check(schwz.capi.lib.schwz_ras_local_csr(sd.h, ctypes.byref(h)))
    return h


def host_cores():
    """CPU share of this process: cgroup quota if set, else the affinity mask; a 1-GPU box
    shares its host, so never more than 16 threads (the box's CPU share for one GPU)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("SCHWZ_CPU_THREADS", "16"))))


def cpu_baseline(shape, inner, iters, subdomains=()):
    """The oracle (CPU restatement, kind 'port') on a bounded sample of the same
    workload: the same grid and settings, `iters` outer iterations; then the same grid cut into
    P z-slabs (P in {1, 2, 4, 8}), the P subdomains taken in turn by all `cores`
    OpenMP threads (the throughput of P ranks x cores/P threads on these cores), a third of the
    iterations each."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    cores = host_cores()
    rp, col, val = O.laplacian3d(*shape)
    N = len(rp) - 1

    def sample(P, k):
        st = O.make_settings(max_iters=k, tol=1e-30, precond=O.PRECOND_JACOBI, local_tol=0.0,
                             local_max_iters=inner, num_threads=cores)
        r = O.ras_run(rp, col, val, np.ones(N), P, O.first_rows_regular(N, P), st, history=False)
        return r["iter_count"], r["elapsed_s"]

    k1, t1 = sample(1, iters)
    out = dict(value=k1 / t1, unit="subdomain-iter/s", cores=cores, kind="port",
               sample="%d outer iterations of the same %dx%dx%d workload (1 subdomain, %d CG "
                      "iterations each), OpenMP oracle, %.1f s" % (k1, shape[0], shape[1], shape[2], inner, t1))
    by_p = {"1": {"outer_iter_per_s": k1 / t1, "subdomain_iter_per_s": k1 / t1, "outer_iters": k1, "seconds": t1}}
    for P in subdomains:
        if P <= 1 or shape[2] // P < 4:
            continue
        k, t = sample(P, max(3, iters // 3))
        by_p[str(P)] = {"outer_iter_per_s": k / t, "subdomain_iter_per_s": P * k / t, "outer_iters": k, "seconds": t}
    out["by_subdomains"] = by_p
    out["by_subdomains_note"] = ("same grid, P z-slabs (overlap 2), the restated reference loop with the P "
                                 "subdomains taken in turn by all %d threads" % cores)
    return out


def mirror_bench_ras(size, inner, iters):
    """The reference's own driver (benchmarking/bench_ras.cpp compiled UNCHANGED against
    schwarz-lib_amd/host/include, linked with libschwz.so -- north_star's boundary) on the same workload,
    as a child process under mpiexec.  Iterations/s from the loop time the mirror prints, the five timing
    ids from the CSV the driver's own writer produces (bench_base.hpp:219-273)."""
    import re
    import shutil
    import tempfile
    binary = os.path.join(ROOT, "schwarz-lib_amd", "build", "bench_ras")
    mpiexec = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
    if not os.path.exists(binary) or not os.path.exists(mpiexec):
        return {"skipped": "bench_ras binary or mpiexec missing (built where the reference checkout exists)"}
    with tempfile.TemporaryDirectory() as tmp:
        cmd = [mpiexec, "-n", "1", binary, "--executor=hip", "--matrix_filename=poisson3d:%d" % size,
               "--enable_global_check", "--set_tol=1e-30", "--num_iters=%d" % iters, "--local_precond=block-jacobi",
               "--precond_max_block_size=1", "--local_max_iters=%d" % inner, "--local_tol=0",
               "--timings_file=%s" % os.path.join(tmp, "t")]
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=tmp)
        except Exception as exc:
            return {"error": str(exc)}
        wall = time.perf_counter() - t0
        m = re.search(r"outer loop: (\d+) iterations in ([0-9.eE+-]+) s, cg flavour (\d+)", p.stdout)
        if p.returncode != 0 or not m:
            return {"error": (p.stdout + p.stderr)[-400:], "returncode": p.returncode}
        k, t, flav = int(m.group(1)), float(m.group(2)), int(m.group(3))
        timings = {}
        try:
            for l in open(os.path.join(tmp, "t_00.csv")).read().splitlines()[1:]:
                f = l.split(",")
                timings[f[0]] = {"total_s": float(f[1]), "avg_s": float(f[2]), "min_s": float(f[3]),
                                 "med_s": float(f[4]), "max_s": float(f[5])}
        except Exception:
            pass
        med = sum(v["med_s"] for v in timings.values()) if timings else None
        return {"value": k / t, "unit": "subdomain-iter/s", "ms_per_step": 1e3 * t / k, "outer_iters": k,
                "loop_seconds": t, "wall_seconds_with_setup": wall, "cg_flavour": flav,
                "ms_per_step_median": 1e3 * med if med else None, "timings": timings,
                "command": " ".join(["mpiexec -n 1 bench_ras"] + cmd[4:-1] + ["--timings_file=t"]),
                "note": "reference benchmarking/bench_ras.cpp, unchanged, on libschwz.so (C++ mirror over the C ABI); "
                        "no warm-up: the first iterations (first-use allocations) are inside the loop time"}


def kernel_source_hash():
    """Hash of the kernel sources of this tree: profiles/traffic.json carries the hash of the tree
    its counters were taken on, and a PMC figure is only quoted for the very same kernels."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "schwarz-lib_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")):  # device code (host_setup.cpp is host-only)
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(shape, tags):
    """HBM bytes per launch from profiles/traffic.json ((2 FETCH_SIZE + WRITE_SIZE) KiB of separate
    --pmc passes, tools/profile_bench.sh) for the kernels named by `tags`; None unless the entry was
    taken on this very grid shape AND on these very kernel sources."""
    out = {t: None for t in tags}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath))
        ent = tj.get("%dx%dx%d" % shape, {})
        if ent.get("kernel_source_hash") != kernel_source_hash():
            return out, "stale" if ent else "absent"
        for t in tags:
            out[t] = ent.get(t, {}).get("hbm_bytes_per_launch")
        return out, "profiles/traffic.json (kernel sources match)"
    except Exception:
        return out, "absent"


def timed_steps(solver, comm, torch, warmup, steps):
    solver.begin_run()
    for _ in range(warmup):
        solver.step()
    torch.cuda.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.step()
    torch.cuda.synchronize()
    comm.barrier()
    return time.perf_counter() - t0


def time_to_residual(schwz, torch, comm, shape, variant, budget_s):
    """time-to-residual(1e-6) at the authors' inexact setting (local_tol 0.1, <= 70 inner
    iterations; run_script:35-38), capped at 3000 outer iterations and `budget_s` seconds."""
    s2, m2 = make_solver(schwz, comm, shape, 70, 1e-6, 3000, 0.1, variant)
    s2.begin_run()
    torch.cuda.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    done = False
    while m2.iter_count < m2.max_iters:
        if s2.step():
            done = True
            break
        if comm.size == 1 or type(comm).__name__ == "InProcessComm":
            if time.perf_counter() - t0 > budget_s:
                break
    torch.cuda.synchronize()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    out = s2.finish_run(elapsed, gather_solution=False)
    res = dict(subdomains=m2.num_subdomains, seconds=elapsed, outer_iters=out["iter_count"],
               converged=bool(out["converged"] and done),
               true_relative_residual=out["residual_norm"] / out["rhs_norm"])
    del s2
    torch.cuda.empty_cache()
    return res


# ----------------------------------------------------------------------------- main

def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("SCHWZ_BENCH_NO_SPAWN") != "1":
        sys.exit(self_launch(a.gpus, sys.argv[1:]))
    if os.environ.get("SCHWZ_BENCH_LAUNCH_PROBE") == "1":
        sys.exit(launch_probe(a))
    # stdout carries the one JSON line and nothing else: libraries that print to the C-level stdout
    # (RCCL writes its version banner there when the first communicator comes up, on every rank) are
    # pointed at stderr until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    mirror = None
    if a.gpus == 1 and not a.no_mirror and os.environ.get("WORLD_SIZE", "1") == "1":
        # a child process, finished before this process touches the GPU
        mirror = mirror_bench_ras(a.size, a.inner, a.mirror_iters)
    import torch
    import schwz_amd as schwz
    import ctypes
    N = a.gpus
    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = None
    # SCHWZ_BENCH_FORCE_DIST=1: take the torch.distributed branch with a single rank as well
    # (brings the nccl process group, the gloo side group and the slab workload up on a 1-GPU box)
    if N > 1 or world > 1 or os.environ.get("SCHWZ_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        ndev = max(torch.cuda.device_count(), 1)
        # One GPU per rank over RCCL.  With fewer visible GPUs than ranks the ranks share them and the
        # halos are staged through host memory over gloo: a rehearsal of the N > 1 code path on a
        # smaller box, named in config.exchange_backend.  SCHWZ_DIST_BACKEND overrides.
        # (source/schwarz_base.cpp:102-109: device = node-local rank)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        backend = os.environ.get("SCHWZ_DIST_BACKEND") or ("nccl" if ndev >= local_world else "gloo")
        local_rank = local_rank % ndev  # a launcher that masks the devices per rank leaves index 0 only
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        comm = schwz.TorchDistComm(device=torch.device("cuda", local_rank))
        comm.device_index = local_rank
        assert comm.size == N, "--gpus must equal the launched world size"
        if a.strong:
            shape = tuple(int(t) for t in a.strong.split(","))
            workload = "3D Poisson %dx%dx%d (fixed), %d z-slab subdomains, overlap 2" % (shape + (N,))
        else:
            sx, sy, sz = [int(t) for t in a.slab.split(",")]
            shape = (sx, sy, sz * N)
            workload = "3D Poisson %dx%dx%d, %d z-slab subdomains (%.1fM rows/GPU), overlap 2" % (
                sx, sy, sz * N, N, sx * sy * sz / 1e6)
    else:
        comm = schwz.InProcessComm(1)
        if a.strong:
            shape = tuple(int(t) for t in a.strong.split(","))
            workload = "3D Poisson %dx%dx%d (fixed), 1 subdomain on 1 MI355X" % shape
        else:
            shape = (a.size, a.size, a.size)
            workload = "3D Poisson %d^3, 1 subdomain on 1 MI355X (BASELINE configs[1])" % a.size
    rank = comm.rank
    lib, check = schwz.capi.lib, schwz.capi.check

    # the GPU context of this process (first HIP call) is not problem setup: created before the setup clock starts
    t_ctx = time.perf_counter()
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
    gpu_context_s = time.perf_counter() - t_ctx
    t_setup = time.perf_counter()
    solver, m = make_solver(schwz, comm, shape, a.inner, 1e-30, a.warmup + 2 * a.steps + 4, 0.0,
                            a.spmv_variant, overlapped=a.overlapped, mixed=a.mixed_halo)
    # first-use allocations belong to setup, whatever --warmup says: the search-direction ring of the deferred
    # x update (14 vectors, allocated by the first solve), the side stream and the RCCL channels of the first
    # exchange -- one untimed step
    solver.begin_run()
    solver.step()
    torch.cuda.synchronize()
    comm.barrier()
    setup_s = time.perf_counter() - t_setup
    sd = solver.subdomains[comm.local_ranks[0]]
    early_exchange = (not a.overlapped) and solver._early_exchange_ok()
    # timed region: W warmup steps, then exactly K steps, no instrumentation
    elapsed = timed_steps(solver, comm, torch, a.warmup, a.steps)
    # roofline leg: the same K steps again with a HIP-event pair around every launch of the
    # dominant kernel on its launch stream.  Kept out of the timed region above because each
    # event record costs ~5 us of GPU idle (measured with rocprofv3 --kernel-trace), which
    # would tax the headline value; the kernel durations themselves are unaffected.
    check(lib.schwz_profile_begin(2 * a.steps * a.inner + 8))
    t1 = time.perf_counter()
    for _ in range(a.steps):
        solver.step()
    torch.cuda.synchronize()
    comm.barrier()
    elapsed_instrumented = time.perf_counter() - t1
    tot_ms, launches = ctypes.c_double(0.0), ctypes.c_int64(0)
    check(lib.schwz_profile_end(ctypes.byref(tot_ms), ctypes.byref(launches)))
    if N > 1:
        elapsed = max(comm.allgather_scalars({rank: elapsed}))
        elapsed_instrumented = max(comm.allgather_scalars({rank: elapsed_instrumented}))
    upd_ms, upd_launches = ctypes.c_double(0.0), ctypes.c_int64(0)
    check(lib.schwz_profile_kind(1, ctypes.byref(upd_ms), ctypes.byref(upd_launches)))
    iters_per_s = a.steps / elapsed
    n_rows = sd.local_size_x
    csr_spmv_bytes = sd.algorithmic_bytes(0)  # 12 nnz + 4 (n + 1) + 16 n, uncoded CSR
    hist = m.post_process_data["global_residual_vector_out"]
    csr_h = sd_csr(sd, schwz)
    fmt = int(lib.schwz_csr_format(csr_h)) if a.spmv_variant == 0 else 0
    mat_bytes = int(lib.schwz_csr_matrix_bytes(csr_h, a.spmv_variant))  # matrix bytes of the launched format
    coded = fmt != 0
    qfree = upd_launches.value > 0
    csr_spmv_equiv = csr_spmv_bytes
    spmv_avg_ms = tot_ms.value / max(launches.value, 1)
    fmt_name = {3: "row-pair coded", 2: "row-pattern coded", 1: "dictionary coded", 0: "plain CSR"}[fmt]
    spmv_name = {3: "spmv_pair_kernel<kSpmvDot> (q = A p, fused p.q; row-pair pattern coded CSR)",
                 2: "spmv_pattern_kernel<kSpmvDot> (q = A p, fused p.q; row-pattern coded CSR tiles)",
                 1: "spmv_dict_kernel<kSpmvDot> (q = A p, fused p.q; dictionary-coded CSR tiles)",
                 0: "spmv_stream_kernel<kSpmvDot> (q = A p, fused p.q; plain CSR, straight-line pipeline)"}[fmt]
    spmv_tag = {3: "spmv_pair_kernel<1,", 2: "spmv_pattern_kernel<1,", 1: "spmv_dict_kernel<1>",
                0: "spmv_stream_kernel<1,"}[fmt]
    # bytes of the SpMV launch in its own format: the matrix once, x once, y once
    spmv_fmt_bytes = mat_bytes + 16 * n_rows
    if qfree:
        # q-free CG iteration (row-pair coded matrix): the launch that recomputes (A p)_i row by row
        # while it updates r (and x) is the longest one.  Bytes of the launched format: the pair codes,
        # p once (gathered), r read + written, [x read + written], [1/diag when it is a full vector].
        # CSR-equivalent bytes, SURVEY 8(d): B_spmv + the x and r updates (2 x 24n) + Jacobi (24n) +
        # the r.z dot (16n) + the norm (8n), minus the 24n of a deferred x update.
        deferred = n_rows > (1 << 21) and os.environ.get("SCHWZ_CG_DEFERX", "1")[:1] != "0"
        diag_vec = 8 * n_rows if int(lib.schwz_ras_jacobi_form(sd.h)) == 1 else 0
        kernel_name = ("spmv_pair_kernel<kSpmvCgUpdate> (row-pair coded A: q_i = (A p)_i recomputed, "
                       + ("" if deferred else "x += alpha p, ") + "r -= alpha q, z = D^-1 r, partial r.z and r.r"
                       + ("; x += alpha p deferred)" if deferred else ")"))
        dom_tag = "spmv_pair_kernel<6,"
        dom_fmt_bytes = mat_bytes + 24 * n_rows + (0 if deferred else 16 * n_rows) + diag_vec
        dom_csr_bytes = csr_spmv_bytes + (72 if deferred else 96) * n_rows
        avg_ms = upd_ms.value / upd_launches.value
        dom_launches = upd_launches.value
        sym = bool(lib.schwz_csr_symmetric(csr_h)) and os.environ.get("SCHWZ_CG_SYM", "1")[:1] != "0"
        spmv_name = ("spmv_pair_kernel<kSpmvDotSym> (partial sums of p.(A p) from the upper triangle of the "
                     "symmetric row-pair coded matrix, nothing stored)") if sym else \
            "spmv_pair_kernel<kSpmvDotOnly> (partial sums of p.(A p), nothing stored; row-pair coded CSR)"
        spmv_tag = "spmv_pair_kernel<7," if sym else "spmv_pair_kernel<5,"
        spmv_fmt_bytes = mat_bytes + 8 * n_rows  # the pair codes and p once; nothing is stored
        flav = int(lib.schwz_ras_cg_flavour(sd.h))
        if flav & 64 and upd_launches.value >= a.steps > 0:
            # virtual first direction: the first update launch of every solve builds its windows from r0, which it
            # reads anyway -- 8 n bytes less for one launch in (update launches per solve); priced as the average
            dom_fmt_bytes -= int(8 * n_rows * a.steps / upd_launches.value)
            kernel_name += " [first launch of a solve: windows from r0, 16n]"
        if flav & 8:
            kernel_name = kernel_name.replace("spmv_pair_kernel<kSpmvCgUpdate>", "spmv_pair_sweep_kernel (z-sweep walk of kSpmvCgUpdate)")
            dom_tag = "spmv_pair_sweep_kernel<"
        if flav & 3 == 2:
            # two launches per iteration: the direction update is fused into the next iteration's p.(A p)
            # launch (r and p read once each, p' written: 24 n + the pair codes)
            spmv_name = ("spmv_pair_dirdot_sweep_kernel (z-sweep walk)" if flav & 16 else "spmv_pair_kernel<kSpmvDirDotSym>") + \
                " (p' = z + beta p and the partial sums of p'.(A p') from the upper triangle of the symmetric " \
                "row-pair coded matrix in one launch)"
            spmv_tag = "spmv_pair_dirdot_sweep_kernel<" if flav & 16 else "spmv_pair_kernel<8,"
            spmv_fmt_bytes = mat_bytes + 24 * n_rows
            csr_spmv_equiv = csr_spmv_bytes + 24 * n_rows
            if flav & 128 and launches.value >= a.steps > 0:
                # the last direction of a fixed-work solve is never stored: one fused launch per solve writes nothing
                spmv_fmt_bytes -= int(8 * n_rows * a.steps / launches.value)
                spmv_name += " [last launch of a solve: no store, 16n]"
    else:
        kernel_name, dom_tag, avg_ms, dom_launches = spmv_name, spmv_tag, spmv_avg_ms, launches.value
        dom_fmt_bytes, dom_csr_bytes = spmv_fmt_bytes, csr_spmv_bytes
    achieved = dom_fmt_bytes / (avg_ms * 1e-3) / 1e9 if dom_launches else 0.0
    spmv_achieved = spmv_fmt_bytes / (spmv_avg_ms * 1e-3) / 1e9 if launches.value else 0.0
    traffic, traffic_src = pmc_traffic(shape if N == 1 else tuple(int(t) for t in a.slab.split(",")),
                                       (dom_tag, spmv_tag))

    def roof(name, fmt_bytes, csr_bytes, ach, ms, n_launch, tag):
        tr = traffic[tag]
        return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS,
                # HBM bytes per launch the PMC counters saw (None unless taken on these kernel sources)
                "traffic": tr, "traffic_source": traffic_src,
                "traffic_over_algorithmic": (tr / fmt_bytes) if tr else None,
                "algorithmic_bytes_per_launch": fmt_bytes,
                "algorithmic_bytes_note": "bytes the launched format needs: %s matrix (%d B) + every vector "
                                          "of the launch once" % (fmt_name, mat_bytes),
                # the same launch priced on uncoded CSR bytes: not a roofline fraction
                "csr_equivalent_bytes_per_launch": csr_bytes,
                "csr_equivalent_x": (csr_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms else None,
                "launches": n_launch, "avg_launch_ms": ms}

    line = {
        # BASELINE.json's metric string; `value` is its first half (outer RAS iterations per second,
        # counted per subdomain and summed over the GPUs), the time-to-residual half is reported in
        # the time_to_residual_* fields of the same line
        "metric": BASELINE_METRIC,
        "metric_note": "value = RAS outer iterations/s x subdomains (whole-job aggregate); "
                       "time-to-residual(1e-6) in time_to_residual_1e-6_s",
        # weak scaling: subdomain-iterations/s of the whole job; strong scaling (fixed grid): the job's
        # outer iterations/s
        "value": iters_per_s if a.strong else N * iters_per_s,
        "unit": "outer-iter/s" if a.strong else "subdomain-iter/s",
        "n_gpus": N,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True,
        "scaling": "strong" if a.strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (7-point Dirichlet Laplacian generated in place, rhs = 1, x0 = 0)",
        "config": {"workload": workload, "inner_cg_iters": a.inner, "precond": "jacobi",
                   "local_tol": 0.0, "overlap": 2, "partition": "regular",
                   "rows_per_gpu": sd.local_size_x, "nnz_per_gpu": sd.nnz_local,
                   "spmv_variant": a.spmv_variant, "matrix_format": fmt_name,
                   "cg_launches_per_iteration": (2 if int(lib.schwz_ras_cg_flavour(sd.h)) & 3 == 2 else 3),
                   "cg_flavour": int(lib.schwz_ras_cg_flavour(sd.h)),
                   "exchange": ("one-sided overlapped, decentralised stop" if a.overlapped
                                else "two-sided, all-gathered residual norms") +
                               (", fp32 halos" if a.mixed_halo else "") +
                               (", posted on a side stream beside the tail of the local solve" if early_exchange else ""),
                   "exchange_backend": (None if backend is None else
                                        ("nccl (RCCL, one GPU per rank)" if backend == "nccl" else
                                         backend + " (ranks share %d GPU(s), halos staged through host: "
                                                   "rehearsal, not a scaling measurement)" %
                                         torch.cuda.device_count()))},
        "ras_iters_per_s": iters_per_s,
        "setup_s": setup_s,
        "setup_note": "problem generation, index sets, matrix codings, upload, solver objects and one untimed step "
                      "(first-use allocations); the process's GPU context (%.2f s) is created before" % gpu_context_s,
        "residual_reduction_in_timed_steps": (sum(h[-1] for h in hist) / sum(h[0] for h in hist))
        if hist and hist[0] else None,
        "roofline": dict(roof(kernel_name, dom_fmt_bytes, dom_csr_bytes, achieved, avg_ms, dom_launches, dom_tag),
                         ms_per_step_instrumented=1e3 * elapsed_instrumented / a.steps),
        "roofline_spmv": roof(spmv_name, spmv_fmt_bytes, csr_spmv_equiv, spmv_achieved, spmv_avg_ms,
                              launches.value, spmv_tag),
    }
    stream = torch.cuda.current_stream().cuda_stream
    # the plain-CSR kernel (variant 6), timed on its own: the figure the north_star's ">= 60 % of the HBM
    # roofline on the local CSR SpMV" refers to -- on this run's matrix and (N = 1) on the slabs a GPU holds
    # in configs[2] (512 x 512 x 64) and configs[4] (1024 x 1024 x 128), keyed by shape
    def csr_plain_entry(handle, n, alg_bytes, shape_key):
        xs_t = torch.randn(n, dtype=torch.float64, device="cuda")
        keep = torch.empty(n, dtype=torch.float64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            check(lib.schwz_csr_spmv(handle, 1.0, xs_t.data_ptr(), 0.0, keep.data_ptr(), 6, stream))
        e0.record()
        for _ in range(20):
            check(lib.schwz_csr_spmv(handle, 1.0, xs_t.data_ptr(), 0.0, keep.data_ptr(), 6, stream))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        tr, tr_src = pmc_traffic(tuple(int(t) for t in shape_key.split("x")), ("spmv_stream_kernel<0,",))
        tr = tr["spmv_stream_kernel<0,"]
        del xs_t, keep
        return {"achieved": alg_bytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": ms,
                "algorithmic_bytes_per_launch": alg_bytes, "rows": n,
                "traffic": tr, "traffic_source": tr_src, "traffic_over_algorithmic": (tr / alg_bytes) if tr else None}

    if rank == 0 and (coded or a.spmv_variant == 0):
        plain = {"kernel": "spmv_stream_kernel<kSpmvPlain> (plain CSR as stored, y = A x; spmv_stream.hip)",
                 "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "note": "north_star: >= 0.60 of the HBM roofline on the local CSR SpMV; bytes = SURVEY 8(d) B_spmv = "
                         "12 nnz + 4 (rows + 1) + 16 rows; one entry per per-GPU shape", "shapes": {}}
        own_key = ("%dx%dx%d" % shape) if N == 1 else a.slab.replace(",", "x")
        plain["shapes"][own_key] = csr_plain_entry(csr_h, sd.local_size_x, csr_spmv_bytes, own_key)
        line["roofline_csr_plain"] = plain
    # the box's own HBM ceilings (STREAM-style kernels of the library, 2 GiB), quoted beside the
    # 8 TB/s spec the roofline fractions are priced against
    if rank == 0:
        try:
            big = torch.empty(1 << 28, dtype=torch.float64, device="cuda")
            dst = torch.empty_like(big)
            meas = {}
            for mode, name, factor in ((1, "read", 1), (2, "copy", 2), (0, "copy_grid_stride", 2)):
                for _ in range(2):
                    check(lib.schwz_stream_probe(big.numel(), mode, big.data_ptr(), dst.data_ptr(), stream))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    check(lib.schwz_stream_probe(big.numel(), mode, big.data_ptr(), dst.data_ptr(), stream))
                e1.record()
                torch.cuda.synchronize()
                meas[name] = factor * big.numel() * 8 / (e0.elapsed_time(e1) / 10) / 1e6
            line["hbm_measured"] = {"read": meas["read"], "copy": meas["copy"],
                                    "copy_grid_stride": meas["copy_grid_stride"], "unit": "GB/s",
                                    "note": "STREAM-style 16-byte kernels over 2 GiB on this GPU: read-only; copy with one "
                                            "element per thread and a grid that covers the buffer once (the form the "
                                            "guide's 6.29 TB/s is quoted for); the same copy as a grid-stride loop of "
                                            "2048 persistent workgroups (what rounds 1-2 quoted as the copy ceiling)"}
            del big, dst
        except Exception as exc:  # never let the side measurement break the bench line
            line["hbm_measured"] = {"error": str(exc)}
    del solver, sd
    torch.cuda.empty_cache()
    if rank == 0 and N == 1 and not a.no_shapes and "roofline_csr_plain" in line:
        # the other per-GPU shapes: the matrix of one z-slab as plain CSR (codings off for these uploads)
        saved = {k: os.environ.get(k) for k in ("SCHWZ_SPMV_PAIR", "SCHWZ_SPMV_PATTERN", "SCHWZ_SPMV_DICT")}
        os.environ.update({k: "0" for k in saved})
        try:
            for spec in [t for t in a.csr_shapes.split(";") if t.strip()]:
                shp = tuple(int(t) for t in spec.split(","))
                key = "%dx%dx%d" % shp
                if key in line["roofline_csr_plain"]["shapes"]:
                    continue
                t_s = time.perf_counter()
                prob = schwz.Problem.laplacian(3, *shp)
                sdx = schwz.Subdomain(prob, 1, 0, 2, schwz.partition_regular(prob.N, 1))
                rp_, col_, val_ = sdx.local_matrix()
                Ax = schwz.Csr(rp_, col_, val_)
                nrows, nnz = len(rp_) - 1, int(rp_[-1])
                del rp_, col_, val_
                ent = csr_plain_entry(Ax.h, nrows, 12 * nnz + 4 * (nrows + 1) + 16 * nrows, key)
                ent["setup_s"] = time.perf_counter() - t_s
                line["roofline_csr_plain"]["shapes"][key] = ent
                Ax.close()
                del Ax, sdx, prob
                torch.cuda.empty_cache()
        except Exception as exc:  # a side measurement: report, never break the line
            line["roofline_csr_plain"]["shapes_error"] = str(exc)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if "roofline_csr_plain" in line:
        fr = [e["frac"] for e in line["roofline_csr_plain"]["shapes"].values()]
        line["roofline_csr_plain"]["min_frac"] = min(fr)
    # strong-scaling leg (north_star: ">= 0.85 strong-scaling efficiency 1 -> 8 GPUs"): the SAME fixed grid on
    # every N, z-slabs, same operating point; the ratio of this field between N = 8 and N = 1 is the efficiency x 8
    if a.strong_grid and not a.strong:
        try:
            sg = tuple(int(t) for t in a.strong_grid.split(","))
            t_s = time.perf_counter()
            ss, _ = make_solver(schwz, comm, sg, a.inner, 1e-30, a.warmup + a.strong_steps + 4, 0.0, a.spmv_variant,
                                overlapped=a.overlapped, mixed=a.mixed_halo)
            ss.begin_run()
            ss.step()
            torch.cuda.synchronize()
            comm.barrier()
            s_setup = time.perf_counter() - t_s
            el = timed_steps(ss, comm, torch, a.warmup, a.strong_steps)
            if N > 1:
                el = max(comm.allgather_scalars({rank: el}))
            line["strong"] = {"grid": "%dx%dx%d" % sg, "subdomains": N, "outer_iter_per_s": a.strong_steps / el,
                              "ms_per_step": 1e3 * el / a.strong_steps, "steps": a.strong_steps, "warmup": a.warmup,
                              "setup_s": s_setup,
                              "note": "fixed grid on every N (z-slabs, overlap 2): outer_iter_per_s(N) / outer_iter_per_s(1) / N "
                                      "is the strong-scaling efficiency"}
            del ss
            torch.cuda.empty_cache()
        except Exception as exc:
            line["strong"] = {"error": str(exc)}
    # the whole step with every matrix coding off (spmv_variant 6: plain CSR, stored-q CG): what a
    # matrix that does not pattern-code (FEM, < 90 % coverage) runs at.  Same workload, same K steps.
    if N == 1 and not a.no_plain_loop and a.spmv_variant == 0 and coded:
        s6, _ = make_solver(schwz, comm, shape, a.inner, 1e-30, a.warmup + a.steps + 2, 0.0, 6)
        el6 = timed_steps(s6, comm, torch, a.warmup, a.steps)
        line["csr_plain_loop"] = {"value": a.steps / el6, "unit": "subdomain-iter/s",
                                  "ms_per_step": 1e3 * el6 / a.steps, "spmv_variant": 6,
                                  "note": "same workload and steps with the lossless matrix codings switched off: "
                                          "plain CSR SpMV kernel (spmv_stream_kernel), CG with a stored q (the general-matrix path)"}
        del s6
        torch.cuda.empty_cache()
    # time-to-residual(1e-6) at the authors' inexact setting (SURVEY 8d ii)
    if not a.no_ttr:
        r1 = time_to_residual(schwz, torch, comm, shape, a.spmv_variant, 1e9 if N > 1 else 4 * a.ttr_budget_s)
        line["time_to_residual_1e-6_s"] = r1["seconds"]
        line["time_to_residual_iters"] = r1["outer_iters"]
        line["time_to_residual_converged"] = r1["converged"]
        line["true_relative_residual"] = r1["true_relative_residual"]
        if N == 1:
            # the iterations-vs-subdomains half of the metric on ONE GPU: the same grid cut into P z-slabs
            # (overlap 2), all P subdomains held by this process, halo messages = device copies
            by_p = {"1": r1}
            for P in [int(t) for t in a.ttr_subdomains.split(",") if t.strip()]:
                if P <= 1 or shape[2] // P < 4:
                    continue
                by_p[str(P)] = time_to_residual(schwz, torch, schwz.InProcessComm(P), shape, a.spmv_variant,
                                                a.ttr_budget_s)
            line["time_to_residual_by_subdomains"] = by_p
    if rank == 0 and N == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(shape, a.inner, a.cpu_iters,
                                            [int(t) for t in a.cpu_subdomains.split(",") if t.strip()])
    if mirror is not None:
        if "value" in mirror:
            mirror["vs_python_host"] = mirror["value"] / line["value"]
            mirror["same_cg_flavour_as_python_host"] = mirror["cg_flavour"] == line["config"]["cg_flavour"]
        line["mirror_bench_ras"] = mirror
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        comm.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
