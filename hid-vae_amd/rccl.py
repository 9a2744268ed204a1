"""A RCCL communicator of the package's own: the gradient exchange as plain stream operations.

Why not torch.distributed's process group for the step's collectives (reference: accelerate -> DistributedDataParallel's bucketed
all-reduce, train_hidvae.py:186-189, 630-632):  ProcessGroupNCCL keeps every eagerly issued collective on a list that its watchdog
thread polls with hipEventQuery until the collective is seen complete (period 100 ms), and it runs all of a group's collectives on ONE
internal stream.  When a later collective is CAPTURED into a HIP graph that internal stream joins the capture -- and from then on a query
of any event that was recorded on it, even eagerly and long before, fails with hipErrorCapturedEvent ("operation not permitted on an
event last recorded in a capturing stream").  The watchdog turns that into an uncaught C++ exception: the process aborts.  So a captured
data-parallel step dies whenever one of the warm-up steps' collectives is still on the watchdog's list when the capture reaches its
first collective -- a race against a 100 ms poll (round 3's and round 4's intermittent SIGABRT of the first captured tagged step;
scratch probe: profiles/r04_capture_event_query_probe.log; the capture mode of step.py removes the OTHER way the same watchdog breaks a
capture, not this one).

Here a collective is ncclAllReduce(ptr, ptr, n, ncclFloat32, ncclSum, comm, stream) on a stream the caller names, nothing else: no work
list, no watchdog, no thread.  Eagerly it is an ordinary launch; under capture it is a node of the graph.  The communicator is
bootstrapped over the process group that already exists (its 128-byte unique id travels as a broadcast object; any backend), once, at
start-up -- that group's own stream never joins a capture.

Fails loudly: no communicator -> RuntimeError; callers that can live without it (parallel.DataParallel) catch it and keep the
collectives on torch.distributed BETWEEN graphs, where nothing of the process group's is ever captured."""
import ctypes
import os

import torch

NCCL_UNIQUE_ID_BYTES = 128
ncclFloat32, ncclSum = 7, 0  # rccl.h: ncclDataType_t / ncclRedOp_t


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * NCCL_UNIQUE_ID_BYTES)]


_LIB = None


def _library():
    """librccl as torch loaded it (one RCCL per process: the same code object torch.distributed's "nccl" backend runs)"""
    global _LIB
    if _LIB is not None:
        return _LIB
    tried = []
    for path in (os.environ.get("HIDVAE_RCCL_LIB"), os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"),
                 "/opt/rocm/lib/librccl.so"):
        if not path:
            continue
        try:
            lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError as e:
            tried.append(f"{path}: {e}")
            continue
        vp, i, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        lib.ncclGetErrorString.restype, lib.ncclGetErrorString.argtypes = ctypes.c_char_p, [i]
        lib.ncclGetUniqueId.restype, lib.ncclGetUniqueId.argtypes = i, [ctypes.POINTER(_UniqueId)]
        lib.ncclCommInitRank.restype, lib.ncclCommInitRank.argtypes = i, [ctypes.POINTER(vp), i, _UniqueId, i]
        lib.ncclCommDestroy.restype, lib.ncclCommDestroy.argtypes = i, [vp]
        lib.ncclAllReduce.restype, lib.ncclAllReduce.argtypes = i, [vp, vp, sz, i, i, vp, vp]
        lib.ncclBroadcast.restype, lib.ncclBroadcast.argtypes = i, [vp, vp, sz, i, i, vp, vp]
        _LIB = lib
        return lib
    raise RuntimeError("hidvae_amd.rccl: no usable librccl (" + "; ".join(tried or ["no candidate path"]) + ")")


def _check(lib, rc, what):
    if rc != 0:
        raise RuntimeError(f"hidvae_amd.rccl: {what} failed ({rc}): {lib.ncclGetErrorString(rc).decode(errors='replace')}")


class _Done:
    """what an asynchronous collective hands back: wait() orders the CURRENT stream behind it (no host blocking), as a
    torch.distributed work handle's does"""
    __slots__ = ("event",)

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)
        return True


class Communicator:
    """One RCCL communicator over the ranks of `group` (default: the world), one rank per device."""

    def __init__(self, group=None, device=None):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("hidvae_amd.rccl.Communicator needs an initialised torch.distributed group to bootstrap over")
        if not torch.cuda.is_available():
            raise RuntimeError("hidvae_amd.rccl.Communicator needs a GPU")
        self.lib = _library()
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        uid = _UniqueId()
        if self.rank == 0:
            _check(self.lib, self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        # (string_at: the raw 128 bytes; reading `internal` as a c_char array would stop at the first NUL)
        box = [ctypes.string_at(ctypes.addressof(uid), NCCL_UNIQUE_ID_BYTES) if self.rank == 0 else None]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        ctypes.memmove(ctypes.addressof(uid), box[0], NCCL_UNIQUE_ID_BYTES)
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib, self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
            self.stream = torch.cuda.Stream(device=self.device)  # where asynchronous collectives run, beside the caller's stream
        self._warm = False

    # ---- collectives: in place, float32, on the stream named (default: the caller's current stream) ------------------------------
    def _args(self, t):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("hidvae_amd.rccl: collectives take contiguous float32 device tensors")
        return ctypes.c_void_p(t.data_ptr()), t.numel()

    def all_reduce_sum_(self, t, stream=None):
        p, n = self._args(t)
        st = torch.cuda.current_stream(t.device) if stream is None else stream
        _check(self.lib, self.lib.ncclAllReduce(p, p, n, ncclFloat32, ncclSum, self.comm, ctypes.c_void_p(st.cuda_stream)), "ncclAllReduce")

    def broadcast_(self, t, root=0, stream=None):
        p, n = self._args(t)
        st = torch.cuda.current_stream(t.device) if stream is None else stream
        _check(self.lib, self.lib.ncclBroadcast(p, p, n, ncclFloat32, root, self.comm, ctypes.c_void_p(st.cuda_stream)), "ncclBroadcast")

    def all_reduce_sum_async_(self, t):
        """ordered behind everything queued on the current stream so far, running on the communicator's own stream beside what the
        caller queues next; -> handle whose wait() makes the current stream wait for the result.  Under capture: two graph edges."""
        cur = torch.cuda.current_stream(t.device)
        self.stream.wait_stream(cur)
        self.all_reduce_sum_(t, stream=self.stream)
        t.record_stream(self.stream)
        ev = torch.cuda.Event()
        ev.record(self.stream)
        return _Done(ev)

    def warm_up(self):
        """the first collective of a communicator connects its rings (allocations: not capturable): run one, eagerly, and -- while at
        it -- check this communicator against torch.distributed's on the same numbers.  Collective call: every rank."""
        if self._warm:
            return
        import torch.distributed as dist
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("hidvae_amd.rccl: the communicator's first collective must run before any graph capture")
        probe = torch.arange(1, 65, device=self.device, dtype=torch.float32) * float(self.rank + 1)
        mine = probe.clone()
        self.all_reduce_sum_(mine)
        if dist.get_backend(self.group) == "nccl":
            theirs = probe.clone()
            dist.all_reduce(theirs, group=self.group)
        else:
            theirs = torch.arange(1, 65, device=self.device, dtype=torch.float32) * float(self.world * (self.world + 1) // 2)
        torch.cuda.synchronize(self.device)
        if not torch.equal(mine, theirs):
            raise RuntimeError("hidvae_amd.rccl: the communicator's all-reduce disagrees with torch.distributed's on a probe vector")
        self._warm = True

    def destroy(self):
        if getattr(self, "comm", None):
            torch.cuda.synchronize(self.device)
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None
