"""Python face of the C-ABI communicator (csrc/comm.hip, include/x2vlm_hip.h `x2_comm_*`): RCCL over xGMI without
torch.distributed in the data path.  The 128-byte RCCL id is created by rank 0 and shared through any key-value store the
host already has (here: a torch.distributed Store, e.g. the TCPStore of the rendezvous, or a file path).

    comm = X2Comm.from_store(store, rank, world)            # collective
    ev = comm.allreduce_bucket(flat_fp32, average=True, stream=side)   # enqueued on `side`, returns a HIP event
    comm.allgather(feat, out); comm.broadcast(flat, root=0); comm.destroy()

accelerator.GradientBuckets uses it instead of dist.all_reduce when X2_COMM=rccl (default: torch.distributed "nccl",
which is the same RCCL underneath)."""
import ctypes as C
import os
import time

import torch

from ._lib import X2HipError, lib

_DTYPE = {torch.float32: 0, torch.bfloat16: 1}


def _check(rc, what):
    if rc != 0:
        raise X2HipError("%s failed (%d): %s" % (what, rc, lib().x2_last_error().decode()))


def _stream_handle(stream):
    return (stream if stream is not None else torch.cuda.current_stream()).cuda_stream


class X2Comm:
    def __init__(self, id_bytes, rank, world):
        assert len(id_bytes) == 128
        h = C.c_void_p()
        _check(lib().x2_comm_init(id_bytes, rank, world, C.byref(h)), "x2_comm_init")
        self.h, self.rank, self.world = h, rank, world

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().x2_comm_unique_id(buf), "x2_comm_unique_id")
        return buf.raw

    _SEQ = 0          # communicators this process has created through from_store (every rank counts the same way)

    @classmethod
    def from_store(cls, store, rank, world, key="x2_comm_id"):
        """store: torch.distributed Store (set/get), or a file path shared by all ranks.  Every communicator gets a key of
        its own (`key` + a per-process sequence number: all ranks create their communicators in the same order), so a second
        communicator - or an id left behind by an earlier run in a file store - can never be picked up for this one; with a
        path the launch id (MASTER_PORT / TORCHELASTIC_RUN_ID) is part of the file name, and rank 0 removes the file once every
        rank has read it (ranks acknowledge through marker files)."""
        cls._SEQ += 1
        key = "%s.%d" % (key, cls._SEQ)
        if isinstance(store, str):
            launch = os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("MASTER_PORT") or "0"
            path = "%s.%s.%s" % (store, launch, key)
            if rank == 0:
                with open(path + ".tmp", "wb") as f:
                    f.write(cls.unique_id())
                os.replace(path + ".tmp", path)
            t0 = time.time()
            while not os.path.exists(path):
                if time.time() - t0 > 120:
                    raise X2HipError("x2_comm: rank 0 never published the RCCL id at %s" % path)
                time.sleep(0.05)
            with open(path, "rb") as f:
                ident = f.read()
            comm = cls(ident, rank, world)        # x2_comm_init returns once every rank has joined: all have read the id
            open("%s.ack%d" % (path, rank), "w").close()
            if rank == 0:
                t0 = time.time()
                while not all(os.path.exists("%s.ack%d" % (path, r)) for r in range(world)) and time.time() - t0 < 120:
                    time.sleep(0.05)
                for r in range(world):
                    try:
                        os.remove("%s.ack%d" % (path, r))
                    except OSError:
                        pass
                try:
                    os.remove(path)
                except OSError:
                    pass
            return comm
        if rank == 0:
            store.set(key, cls.unique_id())
        ident = bytes(store.get(key))         # blocks until rank 0 has set THIS communicator's key
        return cls(ident, rank, world)

    def info(self):
        r, w = C.c_int(), C.c_int()
        _check(lib().x2_comm_info(self.h, C.byref(r), C.byref(w)), "x2_comm_info")
        return r.value, w.value

    def allreduce_bucket(self, flat, average=True, stream=None, want_event=False):
        assert flat.is_cuda and flat.is_contiguous() and flat.dtype in _DTYPE
        ev = None
        if want_event:
            ev = torch.cuda.Event()
            ev.record(stream if stream is not None else torch.cuda.current_stream())    # materialises the hipEvent_t
        _check(lib().x2_comm_allreduce_bucket(self.h, flat.data_ptr(), flat.numel(), _DTYPE[flat.dtype], 1 if average else 0,
                                              ev.cuda_event if ev is not None else None, _stream_handle(stream)), "x2_comm_allreduce_bucket")
        return ev

    def allgather(self, send, recv, stream=None):
        assert send.is_cuda and send.is_contiguous() and recv.is_contiguous() and recv.numel() == send.numel() * self.world
        _check(lib().x2_comm_allgather(self.h, send.data_ptr(), recv.data_ptr(), send.numel(), _DTYPE[send.dtype], None,
                                       _stream_handle(stream)), "x2_comm_allgather")
        return recv

    def broadcast(self, flat, root=0, stream=None):
        assert flat.is_cuda and flat.is_contiguous()
        _check(lib().x2_comm_broadcast(self.h, flat.data_ptr(), flat.numel(), _DTYPE[flat.dtype], root, None, _stream_handle(stream)),
               "x2_comm_broadcast")
        return flat

    def destroy(self):
        if self.h is not None:
            _check(lib().x2_comm_destroy(self.h), "x2_comm_destroy")
            self.h = None
