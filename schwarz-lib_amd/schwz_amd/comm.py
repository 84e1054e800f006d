"""Communication layer of the RAS iteration.

Replaces the MPI calls of the reference: the index handshake
(restricted_schwarz.cpp:400-472), the per-iteration halo Isend/Irecv
(:884-943) or Put/Get (comm_helpers.hpp:122-150), the residual Allgather
(solve.cpp:890-891) and the final solution gather (solve.cpp:1067-1068).

Two implementations with the same interface:
  * InProcessComm  -- all P subdomains live in this process (one GPU); halo
    "messages" are device-to-device copies.  Used by the 1-GPU parity tests.
  * TorchDistComm  -- one subdomain per process / GPU over torch.distributed;
    backend "nccl" is RCCL over xGMI on MI355X, "gloo" is used by the CPU tests.
"""
import os

import numpy as np


class InProcessComm:
    def __init__(self, num_subdomains):
        self.size = int(num_subdomains)
        self.rank = 0
        self.local_ranks = list(range(self.size))
        self.is_root = True
        self._side = None

    def supports_early_exchange(self):
        return True

    def handshake(self, get_lists):
        """get_lists: {me: [(p, ids), ...]} -> put lists {me: [(q, ids), ...]}
        with q ascending: q's get list for me becomes my put list for q."""
        put = {me: [] for me in self.local_ranks}
        for q in sorted(get_lists):
            for p, ids in get_lists[q]:
                put[p].append((q, ids))
        for me in put:
            put[me].sort(key=lambda t: t[0])
        return put

    def exchange(self, sends, recvs):
        """sends/recvs: {(src, dst): 1-D tensor}.  All sources were packed
        before this call; copies are stream-ordered on the current stream."""
        for key, dst_buf in recvs.items():
            dst_buf.copy_(sends[key])

    def start_exchange(self, sends, recvs, overlap=False):
        self.exchange(sends, recvs)
        return None

    def finish_exchange(self, handle):
        if handle is not None:  # an early exchange: the compute stream meets the side stream again
            import torch
            torch.cuda.current_stream().wait_event(handle)

    def start_exchange_early(self, sends, recvs, pack):
        """The exchange that belongs to the START of the next iteration, posted beside the tail of the
        running local solves: `pack(stream)` makes the side stream wait for the solvers' boundary events
        and fills the send buffers from the solves' results; the copies follow on the same stream."""
        import torch
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(priority=-1)  # high priority: the small pack / copy kernels go first
        with torch.cuda.stream(self._side):
            pack(self._side.cuda_stream)
            self.exchange(sends, recvs)
            ev = torch.cuda.Event()
            ev.record(self._side)
        return ev

    def start_flags(self, values, neighbours):
        """values: {me: (mask, stop_at)}; neighbours: {me: (out_ranks, in_ranks)}.
        Delivered by finish_flags as {me: [(mask, stop_at) of every in-neighbour]}."""
        return {me: [values[q] for q in neighbours[me][1]] for me in values}

    def finish_flags(self, handle):
        return handle

    def allgather_scalars(self, values):
        """values: {me: float} for the local subdomains -> list of P floats."""
        return [float(values[p]) for p in range(self.size)]

    def gather_vectors(self, pieces):
        """pieces: {me: np.ndarray} -> concatenation in rank order on root."""
        return np.concatenate([pieces[p] for p in range(self.size)])

    def barrier(self):
        pass


class TorchDistComm:
    """One subdomain per rank.

    Halo values travel on the default process group: with the `nccl` backend that is
    ncclSend/ncclRecv (RCCL over xGMI) on device buffers, grouped per iteration.  Everything
    that is host data anyway -- the index handshake, the P residual norms per iteration, the
    final solution pieces -- goes through a gloo group, so those small collectives never
    serialise against the GPU stream the local solve is running on.  When the default backend
    itself is gloo (CPU tests, or several ranks sharing one GPU) device halo buffers are staged
    through host memory (the reference's `stage_through_host`)."""

    def __init__(self, device=None, group=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("TorchDistComm needs an initialised torch.distributed group")
        self._torch = torch
        self._dist = dist
        self.group = group
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.local_ranks = [self.rank]
        self.is_root = self.rank == 0
        self.device = device if device is not None else torch.device("cpu")
        # the GPU this rank's subdomain lives on (SolverRAS binds its backend to it)
        self.device_index = self.device.index if self.device.type == "cuda" and self.device.index is not None else 0
        self.backend = dist.get_backend(group)
        self.host_group = group if self.backend == "gloo" else dist.new_group(backend="gloo")
        self.stage_through_host = self.backend == "gloo" and self.device.type != "cpu"
        self._stage = {}
        self._side = None
        # the P residual norms per iteration as an RCCL all-gather on device buffers (SCHWZ_NORM_ALLGATHER=host:
        # over the gloo group as in rounds 1-2).  A communicator of its own: its collectives are issued on a side
        # stream while halo sends / receives of the same iteration may be in flight on the default group.
        self.device_norms = (self.backend == "nccl" and self.device.type == "cuda"
                             and os.environ.get("SCHWZ_NORM_ALLGATHER", "device") != "host")
        self.norm_group = dist.new_group(backend="nccl") if self.device_norms else None
        self._norm = None

    def handshake(self, get_lists):
        """Counts and ids travel as int64 host tensors (tags 1 and 2 of the reference become
        one all_gather of counts + point-to-point id lists)."""
        torch, dist = self._torch, self._dist
        mine = get_lists[self.rank]
        counts = torch.zeros(self.size, dtype=torch.int64)
        for p, ids in mine:
            counts[p] = len(ids)
        all_counts = [torch.zeros(self.size, dtype=torch.int64) for _ in range(self.size)]
        dist.all_gather(all_counts, counts, group=self.host_group)
        ops, keep = [], []
        for p, ids in mine:
            t = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64))
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, p, group=self.host_group))
        incoming = []
        for q in range(self.size):
            c = int(all_counts[q][self.rank])
            if q != self.rank and c > 0:
                buf = torch.empty(c, dtype=torch.int64)
                incoming.append((q, buf))
                ops.append(dist.P2POp(dist.irecv, buf, q, group=self.host_group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return {self.rank: [(q, buf.numpy()) for q, buf in incoming]}

    def _host_copy(self, key, buf):
        t = self._stage.get(key)
        if t is None or t.numel() != buf.numel():
            t = self._torch.empty(buf.numel(), dtype=buf.dtype).pin_memory()
            self._stage[key] = t
        return t

    def start_flags(self, values, neighbours):
        """Two int64 per neighbour over the host group: (mask of subdomains known to have
        converged locally, agreed stop iteration).  The decentralised flooding of
        conv_tools.hpp:213-275 on matched point-to-point messages."""
        torch, dist = self._torch, self._dist
        mask, stop = values[self.rank]
        outs, ins = neighbours[self.rank]
        ops, bufs = [], []
        payload = torch.tensor([int(mask), int(stop)], dtype=torch.int64)
        for q in outs:
            ops.append(dist.P2POp(dist.isend, payload, q, group=self.host_group))
        for q in ins:
            b = torch.zeros(2, dtype=torch.int64)
            bufs.append(b)
            ops.append(dist.P2POp(dist.irecv, b, q, group=self.host_group))
        works = dist.batch_isend_irecv(ops) if ops else []
        return works, bufs, payload

    def finish_flags(self, handle):
        works, bufs, _ = handle
        for w in works:
            w.wait()
        return {self.rank: [(int(b[0]), int(b[1])) for b in bufs]}

    def start_exchange(self, sends, recvs, overlap=False):
        """Grouped send/recv of the packed halo buffers ({(src, dst): 1-D tensor}).  With
        `overlap` (nccl) the transfers are issued on a side stream that first waits for the
        pack kernel; the compute stream only meets them again in finish_exchange."""
        dist = self._dist
        ops, post = [], []
        if self.stage_through_host:
            for (src, dst), buf in sorted(sends.items()):
                h = self._host_copy(("s", src, dst), buf)
                h.copy_(buf)  # synchronous device->host copy on the current stream
                ops.append(dist.P2POp(dist.isend, h, dst, group=self.group))
            for (src, dst), buf in sorted(recvs.items()):
                h = self._host_copy(("r", src, dst), buf)
                ops.append(dist.P2POp(dist.irecv, h, src, group=self.group))
                post.append((buf, h))
        else:
            for (src, dst), buf in sorted(sends.items()):
                ops.append(dist.P2POp(dist.isend, buf, dst, group=self.group))
            for (src, dst), buf in sorted(recvs.items()):
                ops.append(dist.P2POp(dist.irecv, buf, src, group=self.group))
        if overlap and ops and not self.stage_through_host and self.device.type != "cpu":
            torch = self._torch
            if self._side is None:
                self._side = torch.cuda.Stream(device=self.device, priority=-1)
            self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side):
                works = dist.batch_isend_irecv(ops)
        else:
            works = dist.batch_isend_irecv(ops) if ops else []
        return works, post

    def finish_exchange(self, handle):
        works, post = handle
        for w in works:
            w.wait()
        for buf, h in post:
            buf.copy_(h, non_blocking=True)

    def supports_early_exchange(self):
        """Device-buffer send / recv on a side stream (nccl); the host-staged path has nothing to overlap."""
        return not self.stage_through_host and self.device.type != "cpu"

    def start_exchange_early(self, sends, recvs, pack):
        """The exchange that belongs to the START of the next iteration, posted beside the tail of the
        running local solve: `pack(stream)` makes the side stream wait for the solver's boundary event and
        fills the send buffer from the solve's result; the grouped ncclSend / ncclRecv follow on that
        stream.  finish_exchange() makes the compute stream wait for them."""
        torch, dist = self._torch, self._dist
        if self._side is None:
            # high priority: the pack kernel and the send / recv kernels are small and everybody waits for them
            self._side = torch.cuda.Stream(device=self.device, priority=-1)
        ops = []
        for (src, dst), buf in sorted(sends.items()):
            ops.append(dist.P2POp(dist.isend, buf, dst, group=self.group))
        for (src, dst), buf in sorted(recvs.items()):
            ops.append(dist.P2POp(dist.irecv, buf, src, group=self.group))
        with torch.cuda.stream(self._side):
            pack(self._side.cuda_stream)
            works = dist.batch_isend_irecv(ops) if ops else []
        return works, []

    def exchange(self, sends, recvs):
        self.finish_exchange(self.start_exchange(sends, recvs))

    def allgather_scalars(self, values):
        torch, dist = self._torch, self._dist
        mine = torch.tensor([float(values[self.rank])], dtype=torch.float64)
        out = [torch.zeros(1, dtype=torch.float64) for _ in range(self.size)]
        dist.all_gather(out, mine, group=self.host_group)
        return [float(t[0]) for t in out]

    def start_allgather_norm_sq(self, fill):
        """Device-side all-gather of the squared residual norms (solve.cpp:890-891).  `fill(ptr, raw_stream)`
        enqueues, on that stream, the write of this rank's value to the device double at `ptr` as soon as it is
        final (Subdomain.norm_sq_to_device); the collective and the copy of the P values to pinned host memory follow
        on the same side stream -- beside the local solve the compute stream is already running."""
        torch, dist = self._torch, self._dist
        if self._norm is None:
            self._norm = dict(stream=torch.cuda.Stream(device=self.device, priority=-1),
                              mine=torch.zeros(1, dtype=torch.float64, device=self.device),
                              all=torch.zeros(self.size, dtype=torch.float64, device=self.device),
                              host=torch.zeros(self.size, dtype=torch.float64).pin_memory(),
                              event=torch.cuda.Event())
        nb = self._norm
        with torch.cuda.stream(nb["stream"]):
            fill(nb["mine"].data_ptr(), nb["stream"].cuda_stream)
            dist.all_gather_into_tensor(nb["all"], nb["mine"], group=self.norm_group)
            nb["host"].copy_(nb["all"], non_blocking=True)
            nb["event"].record()
        return nb

    def finish_allgather_norm_sq(self, handle):
        handle["event"].synchronize()
        return [float(v) for v in handle["host"]]

    def gather_vectors(self, pieces):
        dist = self._dist
        objs = [None] * self.size if self.is_root else None
        dist.gather_object(pieces[self.rank], objs, dst=0, group=self.host_group)
        if self.is_root:
            return np.concatenate(objs)
        return None

    def barrier(self):
        self._dist.barrier(group=self.host_group)

    def broadcast_object(self, obj, src=0):
        """A host object from rank `src` to every rank (setup only)."""
        box = [obj if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src, group=self.host_group)
        return box[0]

    def scatter_objects(self, objs, src=0):
        """objs[r] (given on rank `src`) to rank r (setup only): the distributed ingest's per-rank rows."""
        out = [None]
        self._dist.scatter_object_list(out, objs if self.rank == src else None, src=src, group=self.host_group)
        return out[0]


class WindowComm(TorchDistComm):
    """TorchDistComm plus what the free-running one-sided mode needs (communicate.cpp's windows,
    conv_tools.hpp's window_convergence / window_residual_vector), for the rank processes of ONE node:

      * `share(obj)` -- every rank's object to every rank (setup only): the IPC handles and buffer
        offsets of the halo windows, which live in device memory and are mapped by the neighbours
        (core.DeviceWindow / PeerWindow), so that a halo "put" is a pack kernel storing over xGMI and
        a "get" an unpack kernel loading over it -- no matched receive, no collective in the loop;
      * `host_windows()` -- the convergence and residual windows as numpy arrays on a POSIX
        shared-memory segment that every rank maps: a remote MPI_Put / MPI_Accumulate becomes a store /
        an atomic update on the target rank's row.
    """

    node_windows = True

    def __init__(self, device=None, group=None):
        super().__init__(device=device, group=group)
        self._shm = None

    def share(self, obj):
        out = [None] * self.size
        self._dist.all_gather_object(out, obj, group=self.host_group)
        return out

    def host_windows(self):
        """{'tree': int32 [P][4], 'flags': int32 [P][P], 'count': int32 [P], 'resid': float64 [P][P]},
        row r = rank r's window.  Created by rank 0, zeroed (resid: max double, conv_tools.hpp:69)."""
        from multiprocessing import shared_memory
        P = self.size
        nint = P * 4 + P * P + P
        nint += nint % 2
        nbytes = 4 * nint + 8 * P * P
        name = [None]
        if self.is_root:
            self._shm = shared_memory.SharedMemory(create=True, size=nbytes)
            self._shm.buf[:nbytes] = bytes(nbytes)
            name[0] = self._shm.name
        self._dist.broadcast_object_list(name, src=0, group=self.host_group)
        if not self.is_root:
            self._shm = shared_memory.SharedMemory(name=name[0])
            try:  # the creator unlinks; keep the resource tracker of this process out of it
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self._shm._name, "shared_memory")
            except Exception:
                pass
        ints = np.ndarray((nint,), dtype=np.int32, buffer=self._shm.buf, offset=0)
        resid = np.ndarray((P, P), dtype=np.float64, buffer=self._shm.buf, offset=4 * nint)
        if self.is_root:
            resid[:] = np.finfo(np.float64).max
        self.barrier()
        return dict(tree=ints[:4 * P].reshape(P, 4), flags=ints[4 * P:4 * P + P * P].reshape(P, P),
                    count=ints[4 * P + P * P:4 * P + P * P + P], resid=resid)

    def close_windows(self):
        if self._shm is not None:
            self.barrier()
            shm, self._shm = self._shm, None
            try:
                shm.close()
            except BufferError:
                pass  # numpy views still alive: the mapping goes with the process
            if self.is_root:
                try:
                    shm.unlink()
                except FileNotFoundError:
                    pass
