#!/usr/bin/env python3
"""bench.py -- RAS iterations/sec of the MI355X hot path on the 3-D Poisson problem.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

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
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "schwarz-lib_amd"))

try:
    BASELINE_METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
except Exception:
    BASELINE_METRIC = "RAS iterations/sec + time-to-residual(1e-6), 3D Poisson, 1/2/4/8 subdomains"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
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
    ap.add_argument("--no-ttr", action="store_true", help="skip the time-to-residual run")
    ap.add_argument("--cpu-iters", type=int, default=30,
                    help="outer iterations of the CPU sample (about 10-30 s of CPU work)")
    return ap.parse_args()


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
    """schwz_csr handle of the subdomain's local matrix (for the plain-CSR roofline probe)."""
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


def cpu_baseline(shape, inner, iters):
    """The oracle (CPU restatement, kind 'port') on a bounded sample of the same
    workload: the same grid and settings, `iters` outer iterations."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import oracle as O
    cores = host_cores()
    rp, col, val = O.laplacian3d(*shape)
    N = len(rp) - 1
    st = O.make_settings(max_iters=iters, tol=1e-30, precond=O.PRECOND_JACOBI, local_tol=0.0,
                         local_max_iters=inner, num_threads=cores)
    r = O.ras_run(rp, col, val, np.ones(N), 1, O.first_rows_regular(N, 1), st, history=False)
    return dict(value=r["iter_count"] / r["elapsed_s"], unit="subdomain-iter/s", cores=cores,
                kind="port",
                sample="%d outer iterations of the same %dx%dx%d workload (1 subdomain, %d CG "
                       "iterations each), OpenMP oracle, %.1f s" %
                       (r["iter_count"], shape[0], shape[1], shape[2], inner, r["elapsed_s"]))


def main():
    a = parse()
    # stdout carries the one JSON line and nothing else: libraries that print to the C-level stdout
    # (RCCL writes its version banner there when the first communicator comes up, on every rank) are
    # pointed at stderr until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import schwz_amd as schwz
    N = a.gpus
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SCHWZ_BENCH_FORCE_DIST=1: take the torch.distributed branch with a single rank as well
    # (brings the nccl process group, the gloo side group and the slab workload up on a 1-GPU box)
    if N > 1 or world > 1 or os.environ.get("SCHWZ_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # SCHWZ_DIST_BACKEND=gloo: several ranks may share one GPU (halos staged through host);
        # used to rehearse the N>1 path on a 1-GPU box.  The product backend is nccl (= RCCL).
        backend = os.environ.get("SCHWZ_DIST_BACKEND", "nccl")
        if backend == "nccl":
            # one GPU per rank; a launcher that masks the devices per rank leaves each rank with
            # a single visible device (index 0)
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
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

    t_setup = time.perf_counter()
    solver, m = make_solver(schwz, comm, shape, a.inner, 1e-30, a.warmup + 2 * a.steps + 2, 0.0,
                            a.spmv_variant, overlapped=a.overlapped, mixed=a.mixed_halo)
    setup_s = time.perf_counter() - t_setup
    sd = solver.subdomains[comm.local_ranks[0]]
    solver.begin_run()
    for _ in range(a.warmup):
        solver.step()
    torch.cuda.synchronize()
    comm.barrier()
    # timed region: exactly K steps, no instrumentation
    t0 = time.perf_counter()
    for _ in range(a.steps):
        solver.step()
    torch.cuda.synchronize()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    # roofline leg: the same K steps again with a HIP-event pair around every launch of the
    # dominant kernel on its launch stream.  Kept out of the timed region above because each
    # event record costs ~5 us of GPU idle (measured with rocprofv3 --kernel-trace), which
    # would tax the headline value; the kernel durations themselves are unaffected.
    import ctypes
    schwz.capi.check(schwz.capi.lib.schwz_profile_begin(2 * a.steps * a.inner + 8))
    t1 = time.perf_counter()
    for _ in range(a.steps):
        solver.step()
    torch.cuda.synchronize()
    comm.barrier()
    elapsed_instrumented = time.perf_counter() - t1
    tot_ms, launches = ctypes.c_double(0.0), ctypes.c_int64(0)
    schwz.capi.check(schwz.capi.lib.schwz_profile_end(ctypes.byref(tot_ms), ctypes.byref(launches)))
    if N > 1:
        elapsed = max(comm.allgather_scalars({rank: elapsed}))
        elapsed_instrumented = max(comm.allgather_scalars({rank: elapsed_instrumented}))
    upd_ms, upd_launches = ctypes.c_double(0.0), ctypes.c_int64(0)
    schwz.capi.check(schwz.capi.lib.schwz_profile_kind(1, ctypes.byref(upd_ms), ctypes.byref(upd_launches)))
    iters_per_s = a.steps / elapsed
    alg_spmv = sd.algorithmic_bytes(0)
    n_rows = sd.local_size_x
    hist = m.post_process_data["global_residual_vector_out"]
    fmt = int(schwz.capi.lib.schwz_csr_format(sd_csr(sd, schwz))) if a.spmv_variant == 0 else 0
    dict_coded = fmt != 0
    qfree = upd_launches.value > 0
    spmv_avg_ms = tot_ms.value / max(launches.value, 1)
    spmv_name = {3: "spmv_pair_kernel<kSpmvDot> (q = A p, fused p.q; row-pair pattern coded CSR)",
                 2: "spmv_pattern_kernel<kSpmvDot> (q = A p, fused p.q; row-pattern coded CSR tiles)",
                 1: "spmv_dict_kernel<kSpmvDot> (q = A p, fused p.q; dictionary-coded CSR tiles)",
                 0: "spmv_tiled2_kernel<kSpmvDot> (q = A p, fused p.q; plain CSR)"}[fmt]
    spmv_tag = {3: "spmv_pair_kernel<1,", 2: "spmv_pattern_kernel<1,", 1: "spmv_dict_kernel<1>",
                0: "spmv_tiled2_kernel<1>"}[fmt]
    if qfree:
        # q-free CG iteration (row-pair coded matrix): the launch that recomputes (A p)_i row by row
        # while it updates x and r is the longest one.  Algorithmic bytes, SURVEY 8(d): the SpMV
        # (B_spmv) + the x and r updates (2 x 24n) + Jacobi (24n) + the r.z dot (16n) + the norm (8n).
        kernel_name = ("spmv_pair_kernel<kSpmvCgUpdate> (row-pair coded A: q_i = (A p)_i recomputed, "
                       "x += alpha p, r -= alpha q, z = D^-1 r, partial r.z and r.r)")
        dom_tag = "spmv_pair_kernel<6,"
        alg_dom = alg_spmv + 96 * n_rows
        if n_rows > (1 << 21) and os.environ.get("SCHWZ_CG_DEFERX", "1")[:1] != "0":
            # large systems: x += alpha p is not part of this launch (deferred, applied to up to 16
            # search directions at once by cg_flush_x_kernel): 24 n fewer algorithmic bytes
            kernel_name = ("spmv_pair_kernel<kSpmvCgUpdate> (row-pair coded A: q_i = (A p)_i recomputed, "
                           "r -= alpha q, z = D^-1 r, partial r.z and r.r; x += alpha p deferred)")
            alg_dom = alg_spmv + 72 * n_rows
        avg_ms = upd_ms.value / upd_launches.value
        dom_launches = upd_launches.value
        spmv_name = "spmv_pair_kernel<kSpmvDotOnly> (partial sums of p.(A p), nothing stored; row-pair coded CSR)"
        spmv_tag = "spmv_pair_kernel<5,"
        if (int(schwz.capi.lib.schwz_csr_symmetric(sd_csr(sd, schwz))) and
                os.environ.get("SCHWZ_CG_SYM", "1")[:1] != "0"):
            # the upload found the matrix symmetric: p.(A p) from the upper triangle
            spmv_name = ("spmv_pair_kernel<kSpmvDotSym> (partial sums of p.(A p) from the upper triangle of the "
                         "symmetric row-pair coded matrix, nothing stored)")
            spmv_tag = "spmv_pair_kernel<7,"
    else:
        kernel_name, dom_tag, alg_dom, avg_ms, dom_launches = spmv_name, spmv_tag, alg_spmv, spmv_avg_ms, launches.value
    achieved = alg_dom / (avg_ms * 1e-3) / 1e9 if dom_launches else 0.0
    traffic = None
    spmv_traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            ent = tj.get("%dx%dx%d" % shape, {})
            # only quote a PMC figure taken for the very kernel this run is using
            traffic = ent.get(dom_tag, {}).get("hbm_bytes_per_launch")
            spmv_traffic = ent.get(spmv_tag, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # the plain-CSR kernel on the same matrix (variant 6), timed on its own: the figure the
    # north_star's ">= 60 % of the HBM roofline on the local CSR SpMV" refers to
    csr_plain = None
    if rank == 0 and dict_coded:
        xs, _ = sd.vector(2)
        ys, _ = sd.vector(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stream = torch.cuda.current_stream().cuda_stream
        keep = torch.empty(sd.local_size_x, dtype=torch.float64, device="cuda")
        for _ in range(3):
            schwz.capi.check(schwz.capi.lib.schwz_csr_spmv(sd_csr(sd, schwz), 1.0, xs, 0.0, keep.data_ptr(), 6, stream))
        e0.record()
        for _ in range(20):
            schwz.capi.check(schwz.capi.lib.schwz_csr_spmv(sd_csr(sd, schwz), 1.0, xs, 0.0, keep.data_ptr(), 6, stream))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        csr_plain = {"kernel": "spmv_tiled2_kernel<kSpmvPlain> (plain CSR, y = A x)", "bound": "hbm",
                     "achieved": alg_spmv / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": alg_spmv / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_ms": ms,
                     "algorithmic_bytes_per_launch": alg_spmv}
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
                   "spmv_variant": a.spmv_variant,
                   "exchange": ("one-sided overlapped, decentralised stop" if a.overlapped
                                else "two-sided, all-gathered residual norms") +
                               (", fp32 halos" if a.mixed_halo else "")},
        "ras_iters_per_s": iters_per_s,
        "setup_s": setup_s,
        "residual_reduction_in_timed_steps": (sum(h[-1] for h in hist) / sum(h[0] for h in hist))
        if hist and hist[0] else None,
        "roofline": {"kernel": kernel_name, "bound": "hbm",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     # bytes the counters saw, over the same launch time: the HBM utilisation proper
                     # (the lossless matrix coding is why it is far below `frac`)
                     "traffic_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and avg_ms else None,
                     "algorithmic_bytes_per_launch": alg_dom, "launches": dom_launches,
                     "avg_launch_ms": avg_ms,
                     "ms_per_step_instrumented": 1e3 * elapsed_instrumented / a.steps},
        "roofline_spmv": {"kernel": spmv_name, "bound": "hbm",
                          "achieved": alg_spmv / (spmv_avg_ms * 1e-3) / 1e9 if launches.value else 0.0,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": (alg_spmv / (spmv_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if launches.value else 0.0,
                          "traffic": spmv_traffic,
                          "traffic_frac": (spmv_traffic / (spmv_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                          if spmv_traffic and spmv_avg_ms else None,
                          "algorithmic_bytes_per_launch": alg_spmv,
                          "launches": launches.value, "avg_launch_ms": spmv_avg_ms},
        "roofline_csr_plain": csr_plain,
    }
    # the box's own HBM ceilings (STREAM-style kernels of the library, 2 GiB), quoted beside the
    # 8 TB/s spec the roofline fractions are priced against
    if rank == 0:
        try:
            big = torch.empty(1 << 28, dtype=torch.float64, device="cuda")
            dst = torch.empty_like(big)
            stream = torch.cuda.current_stream().cuda_stream
            meas = {}
            for mode, name, factor in ((1, "read", 1), (0, "copy", 2)):
                for _ in range(2):
                    schwz.capi.check(schwz.capi.lib.schwz_stream_probe(big.numel(), mode, big.data_ptr(),
                                                                       dst.data_ptr(), stream))
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    schwz.capi.check(schwz.capi.lib.schwz_stream_probe(big.numel(), mode, big.data_ptr(),
                                                                       dst.data_ptr(), stream))
                e1.record()
                torch.cuda.synchronize()
                meas[name] = factor * big.numel() * 8 / (e0.elapsed_time(e1) / 10) / 1e6
            line["hbm_measured"] = {"read": meas["read"], "copy": meas["copy"], "unit": "GB/s",
                                    "note": "STREAM-style double2 kernels over 2 GiB on this GPU"}
            del big, dst
        except Exception as exc:  # never let the side measurement break the bench line
            line["hbm_measured"] = {"error": str(exc)}
    # time-to-residual(1e-6) at the authors' inexact setting (SURVEY 8d ii)
    if not a.no_ttr:
        del solver
        torch.cuda.empty_cache()
        s2, m2 = make_solver(schwz, comm, shape, 70, 1e-6, 3000, 0.1, a.spmv_variant)
        out = s2.run(gather_solution=False)
        line["time_to_residual_1e-6_s"] = out["elapsed"]
        line["time_to_residual_iters"] = out["iter_count"]
        line["time_to_residual_converged"] = out["converged"]
        line["true_relative_residual"] = out["residual_norm"] / out["rhs_norm"]
        del s2
    if rank == 0 and N == 1 and not a.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(shape, a.inner, a.cpu_iters)
    sys.stdout.flush()
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        comm.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
