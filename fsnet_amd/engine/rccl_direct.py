"""Direct RCCL calls: every collective of a data-parallel step on the engine's own HIP streams.

A data-parallel step issues 100 all-reduces of a few hundred bytes for SyncBatchNorm plus the gradient buckets
(DESIGN.md 6).  Through torch.distributed each small exchange costs ~30 us of host time — tensor checks, the event
hand-shake with the process group's internal stream, the work object and its watchdog bookkeeping — which makes the
eager N > 1 step host-bound, and the work objects keep the step from being captured into a hipGraph.  The exchange
itself is one ncclAllReduce on the stream the producing and consuming kernels already run on, so this module opens
an RCCL communicator over the same ranks (unique id from rank 0, distributed with torch.distributed) and calls
ncclAllReduce through ctypes on the current HIP stream: stream order does the rest, nothing else is launched, and
under stream capture the launch becomes a graph node.  Once it exists, the engine sends ALL of a step's collectives
through this one communicator (dataparallel.py); torch.distributed's own communicator is idle during a step.

Everything here fails soft: any error while loading the library, creating the communicator or in the start-up
self-tests leaves `DirectComm.create` returning None and the torch.distributed path in use.  Self-tests: (1) a
known integer-valued f64 vector reduced over the ranks must equal its closed-form sum on every rank; (2) `capture_ok`:
all-reduces on two streams captured into a hipGraph and replayed twice must give the exact sums — only then are
data-parallel steps captured.

Control plane (round 4).  The engine issues NO torch.distributed NCCL collective any more: the unique id, every
yes/no agreement between the ranks (StoreAgreement) and the start-up parameter broadcast (ncclBroadcast on the direct
communicator) go through the process group's key-value store and this module.  What that buys is determinism: the
c10d NCCL watchdog thread polls the events of collectives it has not reaped yet, and such a poll landing inside an open
hipGraph capture std::terminate()s the process — with no c10d work object ever created by the engine
there is nothing for it to poll.  quiesce_watchdog() remains for collectives the CALLER issues (an epoch barrier
right before a re-capture).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

NCCL_SUM = 0
NCCL_DTYPE = {torch.float64: 8, torch.float32: 7, torch.int32: 2, torch.int64: 4}


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = C.CDLL(path)          # the copy torch already mapped: same RCCL for both communicators
    lib.ncclGetUniqueId.restype = C.c_int
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.restype = C.c_int
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
    lib.ncclAllReduce.restype = C.c_int
    lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclBroadcast.restype = C.c_int
    lib.ncclBroadcast.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.ncclCommDestroy.restype = C.c_int
    lib.ncclCommDestroy.argtypes = [C.c_void_p]
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    return lib


def quiesce_watchdog(device):
    """Call before opening a hipGraph capture while a torch.distributed NCCL process group is alive.  Its watchdog
    thread polls the events of every collective it has not reaped yet (100 ms cadence); such a poll landing inside
    an open capture raises in that thread and std::terminate()s the process (backtrace: ProcessGroupNCCL::Watchdog::
    run -> rethrow_exception; seen in ~50 % of captures that started within 100 ms of a broadcast / all_reduce).  With
    the device idle, three watchdog periods are enough for its work list to drain — the engine itself issues no
    torch.distributed collective during a step."""
    import time
    torch.cuda.synchronize(device)
    time.sleep(0.35)


class StoreAgreement(object):
    """Rank agreement and small blobs through the process group's key-value store (TCPStore / FileStore): no
    collective, no communicator, no watchdog work item.  Every rank must make the same calls in the same order."""
    _generations = {}

    def __init__(self, group=None):
        self.store = dist.distributed_c10d._get_default_store()
        self.ranks = list(range(dist.get_world_size())) if group is None else dist.get_process_group_ranks(group)
        self.me = dist.get_rank()
        # the group is named by ALL its ranks (a digest: groups that share their first ranks do not collide) and by how many
        # contexts THIS group has had in this process — every member creates a group's contexts in the same order, whatever
        # other (overlapping) groups it creates in between
        import hashlib
        gid = hashlib.sha1(",".join(str(r) for r in self.ranks).encode()).hexdigest()[:16]
        gen = StoreAgreement._generations[gid] = StoreAgreement._generations.get(gid, 0) + 1
        self.prefix = "fsnet_amd/agree/%s/%d" % (gid, gen)
        self.seq = 0

    def _prefix(self):
        return self.prefix

    def all_agree(self, ok):
        """True on every rank iff ok on every rank"""
        self.seq += 1
        pre = self._prefix()
        self.store.set("%s/%d/%d" % (pre, self.seq, self.me), b"1" if ok else b"0")
        good = True
        for r in self.ranks:
            good = good and bytes(self.store.get("%s/%d/%d" % (pre, self.seq, r))) == b"1"
        if self.seq > 1:
            # (every rank has read round seq - 1 before it posted round seq: its own old key can go)
            try:
                self.store.delete_key("%s/%d/%d" % (pre, self.seq - 1, self.me))
            except Exception:
                pass
        return good

    def gather(self, blob):
        """every rank's bytes, in rank order, on every rank"""
        self.seq += 1
        pre = self._prefix()
        self.store.set("%s/%d/g%d" % (pre, self.seq, self.me), bytes(blob))
        return [bytes(self.store.get("%s/%d/g%d" % (pre, self.seq, r))) for r in self.ranks]

    def share(self, blob):
        """bytes from the group's first rank to everybody (blob is ignored elsewhere)"""
        self.seq += 1
        key = "%s/%d/blob" % (self._prefix(), self.seq)
        if self.me == self.ranks[0]:
            self.store.set(key, bytes(blob))
        return bytes(self.store.get(key))


class DirectComm(object):
    def __init__(self, lib, comm, world, rank, device):
        self.lib, self.comm, self.world, self.rank, self.device = lib, comm, world, rank, device
        self.capture_ok = False
        self.capture_test = "not run"      # what the start-up capture self-test exercised (bench.py reports it)
        self.agreement = None
        self._selftest_graph = None        # kept until close(): see BaseTrainingHook on destroying graph execs

    @classmethod
    def create(cls, group=None, device=None):
        """Collective over `group`; returns None when the direct path is unavailable or fails its self-test."""
        if os.environ.get("FSNET_AMD_RCCL_DIRECT", "1") == "0" or not torch.cuda.is_available():
            return None
        try:
            if dist.get_backend(group) != "nccl":
                return None
            world, rank = dist.get_world_size(group), dist.get_rank(group)
            device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
            # stage 1 (local): library + unique id.  The ranks agree on its outcome BEFORE anyone enters the blocking
            # ncclCommInitRank, so a rank that cannot load the library does not strand the others there.
            lib, uid, err = None, _UniqueId(), None
            agreement = StoreAgreement(group)
            try:
                lib = _load()
                if rank == 0:
                    cls._check(lib, lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
            except Exception as e:      # noqa: BLE001
                err = e
            if not agreement.all_agree(err is None):
                raise RuntimeError("a rank could not prepare the RCCL communicator (%s)" % (err,))
            C.memmove(C.byref(uid), agreement.share(bytes(uid)), 128)
            comm = C.c_void_p()
            with torch.cuda.device(device):
                cls._check(lib, lib.ncclCommInitRank(C.byref(comm), world, uid, rank), "ncclCommInitRank")
            self = cls(lib, comm, world, rank, device)
            self.agreement = agreement
            if not self._self_test(group):
                self.close()
                return None
            self.capture_ok = self._capture_test(group)
            return self
        except Exception as e:      # noqa: BLE001 — any failure means "use torch.distributed"
            import warnings
            warnings.warn("fsnet_amd: direct RCCL path unavailable (%s: %s); SyncBN exchanges use torch.distributed" % (
                type(e).__name__, e))
            return None

    def close(self):
        if self.comm is not None and self.comm.value:
            torch.cuda.synchronize(self.device)
            if self._selftest_graph is not None:
                self._selftest_graph.reset()           # RCCL nodes of this communicator: gone before it is
                self._selftest_graph = None
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None

    @staticmethod
    def _check(lib, status, what):
        if status != 0:
            raise RuntimeError("%s failed: %s" % (what, lib.ncclGetErrorString(status).decode()))

    def all_reduce_sum(self, t, out=None):
        """SUM over the ranks into `out` (default: in place), enqueued on the current HIP stream"""
        from ..hip.binding import raw_stream
        assert t.is_cuda and t.is_contiguous()
        dst = t if out is None else out
        assert dst.is_contiguous() and dst.numel() == t.numel() and dst.dtype == t.dtype
        self._check(self.lib, self.lib.ncclAllReduce(t.data_ptr(), dst.data_ptr(), t.numel(), NCCL_DTYPE[t.dtype], NCCL_SUM,
                                                     self.comm, C.c_void_p(raw_stream(t.device.index))), "ncclAllReduce")

    def broadcast(self, t, root=0):
        """rank `root`'s values into `t` on every rank, enqueued on the current HIP stream (any dtype: moved as bytes
        or as 32/64-bit words)"""
        from ..hip.binding import raw_stream
        assert t.is_cuda and t.is_contiguous()
        nbytes = t.numel() * t.element_size()
        if nbytes == 0:
            return
        # ncclInt8 = 0 (bytes), ncclInt32 = 2: word-sized counts keep the element count small for the 100 MB arena
        count, code = (nbytes // 4, 2) if nbytes % 4 == 0 and t.data_ptr() % 4 == 0 else (nbytes, 0)
        self._check(self.lib, self.lib.ncclBroadcast(t.data_ptr(), t.data_ptr(), count, code, root, self.comm,
                                                     C.c_void_p(raw_stream(t.device.index))), "ncclBroadcast")

    def _self_test(self, group):
        # integer-valued f64: the sum over the ranks is exact and known in closed form — no second communicator needed
        base = torch.arange(37, dtype=torch.float64, device=self.device) + 1.0
        a = base * (self.rank + 1)
        self.all_reduce_sum(a)
        b = base.clone()
        if self.rank != 0:
            b.zero_()
        self.broadcast(b, 0)
        torch.cuda.synchronize(self.device)
        good = torch.equal(a, base * (self.world * (self.world + 1) / 2.0)) and torch.equal(b, base)
        return self.agreement.all_agree(good)            # every rank takes the same decision

    def _capture_test(self, group):
        """all-reduces on a capture stream and a forked stream inside one hipGraph, replayed twice: exact sums on
        every rank, or data-parallel steps stay eager"""
        if os.environ.get("FSNET_AMD_GRAPH_DP", "1") == "0":
            return False
        dev, W = self.device, self.world

        def agree(flag):
            return self.agreement.all_agree(flag)

        graph, a, b = None, None, None
        try:
            a = torch.full((257,), float(self.rank + 1), dtype=torch.float64, device=dev)
            b = torch.full((70001,), float(self.rank + 1), dtype=torch.float32, device=dev)
            main, side = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            torch.cuda.synchronize(dev)       # (no torch.distributed collective has been issued: nothing to quiesce)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=main, capture_error_mode="thread_local"):
                self.all_reduce_sum(a)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    self.all_reduce_sum(b)
                torch.cuda.current_stream(dev).wait_stream(side)
            captured = True
        except Exception as e:      # noqa: BLE001
            import warnings
            warnings.warn("fsnet_amd: RCCL collectives could not be captured into a hipGraph (%s: %s); data-parallel "
                          "steps run eagerly" % (type(e).__name__, e))
            captured = False
        self._selftest_graph = graph
        if not agree(captured):          # nobody replays unless everybody captured
            return False
        graph.replay()
        graph.replay()
        torch.cuda.synchronize(dev)
        s1 = W * (W + 1) / 2.0
        good = bool((a == s1 * W).all()) and bool((b == s1 * W).all())
        # a one-rank all-reduce in place enqueues nothing: the captured graph is empty and the test proves nothing about
        # RCCL kernel nodes (torch warns "The CUDA Graph is empty") — say so instead of claiming a pass
        self.capture_test = "vacuous (world size 1: RCCL enqueues nothing)" if W == 1 else "passed at world size %d" % W
        return agree(good)
