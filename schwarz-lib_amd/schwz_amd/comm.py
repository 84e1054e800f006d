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
import numpy as np


class InProcessComm:
    def __init__(self, num_subdomains):
        self.size = int(num_subdomains)
        self.rank = 0
        self.local_ranks = list(range(self.size))
        self.is_root = True

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

    def start_exchange(self, sends, recvs):
        self.exchange(sends, recvs)
        return None

    def finish_exchange(self, handle):
        pass

    def allgather_scalars(self, values):
        """values: {me: float} for the local subdomains -> list of P floats."""
        return [float(values[p]) for p in range(self.size)]

    def gather_vectors(self, pieces):
        """pieces: {me: np.ndarray} -> concatenation in rank order on root."""
        return np.concatenate([pieces[p] for p in range(self.size)])

    def barrier(self):
        pass


class TorchDistComm:
    """One subdomain per rank.  `device` is where halo buffers live."""

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

    def handshake(self, get_lists):
        """Counts and ids travel as int64 tensors (tags 1 and 2 of the
        reference become one all_gather of counts + point-to-point ids)."""
        torch, dist = self._torch, self._dist
        mine = get_lists[self.rank]
        counts = torch.zeros(self.size, dtype=torch.int64)
        for p, ids in mine:
            counts[p] = len(ids)
        counts = counts.to(self.device)
        all_counts = [torch.zeros(self.size, dtype=torch.int64, device=self.device)
                      for _ in range(self.size)]
        dist.all_gather(all_counts, counts, group=self.group)  # setup only
        all_counts = [t.cpu() for t in all_counts]
        ops, keep = [], []
        for p, ids in mine:
            t = torch.from_numpy(np.ascontiguousarray(ids, dtype=np.int64)).to(self.device)
            keep.append(t)
            ops.append(dist.P2POp(dist.isend, t, p, group=self.group))
        incoming = []
        for q in range(self.size):
            c = int(all_counts[q][self.rank])
            if q != self.rank and c > 0:
                buf = torch.empty(c, dtype=torch.int64, device=self.device)
                incoming.append((q, buf))
                ops.append(dist.P2POp(dist.irecv, buf, q, group=self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return {self.rank: [(q, buf.cpu().numpy()) for q, buf in incoming]}

    def start_exchange(self, sends, recvs):
        """Grouped send/recv (ncclGroupStart..End under the nccl backend)."""
        dist = self._dist
        ops = []
        for (src, dst), buf in sorted(sends.items()):
            ops.append(dist.P2POp(dist.isend, buf, dst, group=self.group))
        for (src, dst), buf in sorted(recvs.items()):
            ops.append(dist.P2POp(dist.irecv, buf, src, group=self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def finish_exchange(self, handle):
        for w in handle or []:
            w.wait()

    def exchange(self, sends, recvs):
        self.finish_exchange(self.start_exchange(sends, recvs))

    def allgather_scalars(self, values):
        torch, dist = self._torch, self._dist
        mine = torch.tensor([float(values[self.rank])], dtype=torch.float64, device=self.device)
        out = [torch.zeros(1, dtype=torch.float64, device=self.device) for _ in range(self.size)]
        dist.all_gather(out, mine, group=self.group)
        return [float(t.item()) for t in out]

    def gather_vectors(self, pieces):
        dist = self._dist
        objs = [None] * self.size if self.is_root else None
        dist.gather_object(pieces[self.rank], objs, dst=0, group=self.group)
        if self.is_root:
            return np.concatenate(objs)
        return None

    def barrier(self):
        self._dist.barrier(group=self.group)
