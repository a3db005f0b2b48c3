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
        path the launch id (MASTER_PORT / TORCHELASTIC_RUN_ID) is part of the file name and the exchange is `_from_file`'s
        announce / publish / acknowledge protocol."""
        cls._SEQ += 1
        key = "%s.%d" % (key, cls._SEQ)
        if isinstance(store, str):
            launch = os.environ.get("TORCHELASTIC_RUN_ID") or os.environ.get("MASTER_PORT") or "0"
            path = "%s.%s.%s" % (store, launch, key)
            return cls._from_file(path, rank, world)
        if rank == 0:
            store.set(key, cls.unique_id())
        ident = bytes(store.get(key))         # blocks until rank 0 has set THIS communicator's key
        return cls(ident, rank, world)

    @classmethod
    def _from_file(cls, path, rank, world, timeout=120.0):
        """File rendezvous that survives leftovers of a crashed launch with the same name (default port, same sequence number)
        WITHOUT trusting file timestamps (coarse mtime granularity / clock skew on shared file systems): every rank r > 0 draws
        a random 16-byte token and announces it (`path.hello<r>`); rank 0 first removes whatever it finds under this name
        (its own leftovers included), waits for all announcements, and publishes the RCCL id FOLLOWED BY the tokens it
        collected.  A rank accepts an id file only if it carries the rank's own token at the rank's position - an id left behind
        by an earlier launch cannot (fresh random token), however new its timestamp looks.  A rank whose announcement rank 0's
        clean-up removed writes it again (same token).  Time-outs raise; rank 0 removes the files once every rank has joined
        and acknowledged."""
        hello, ack = "%s.hello%d" % (path, rank), "%s.ack%d" % (path, rank)
        TOK = 16

        def put(p, data=b""):
            with open(p + ".tmp%d" % rank, "wb") as f:
                f.write(data)
            os.replace(p + ".tmp%d" % rank, p)      # readers never see a half-written file

        def get(p, n):
            try:
                with open(p, "rb") as f:
                    d = f.read()
                return d if len(d) == n else None
            except OSError:
                return None

        def wait(cond, what):
            t0 = time.time()
            while not cond():
                if time.time() - t0 > timeout:
                    raise X2HipError("x2_comm: timed out after %.0f s waiting for %s (%s)" % (timeout, what, path))
                time.sleep(0.05)

        if rank == 0:
            for r in range(world):
                for f in ("%s.hello%d" % (path, r), "%s.ack%d" % (path, r)):
                    if os.path.exists(f):
                        os.remove(f)
            for f in (path, path + ".tmp0"):
                if os.path.exists(f):
                    os.remove(f)
            tokens = {}

            def all_announced():
                for r in range(1, world):
                    if r not in tokens:
                        t = get("%s.hello%d" % (path, r), TOK)
                        if t is not None:
                            tokens[r] = t
                return len(tokens) == world - 1
            wait(all_announced, "every rank's announcement")
            ident = cls.unique_id()
            put(path, ident + b"".join(tokens[r] for r in range(1, world)))
        else:
            token = os.urandom(TOK)
            put(hello, token)
            found = {}

            def mine():
                if not os.path.exists(hello):
                    put(hello, token)             # removed by rank 0's clean-up: announce again
                d = get(path, 128 + TOK * (world - 1))
                if d is not None and d[128 + TOK * (rank - 1): 128 + TOK * rank] == token:
                    found["id"] = d[:128]
                    return True
                return False
            wait(mine, "rank 0's RCCL id carrying this rank's token")
            ident = found["id"]
        comm = cls(ident, rank, world)            # x2_comm_init returns once every rank has joined: all have read the id
        put(ack)
        if rank == 0:
            wait(lambda: all(os.path.exists("%s.ack%d" % (path, r)) for r in range(world)), "every rank's acknowledgement")
            for r in range(world):
                for f in ("%s.hello%d" % (path, r), "%s.ack%d" % (path, r)):
                    try:
                        os.remove(f)
                    except OSError:
                        pass
            try:
                os.remove(path)
            except OSError:
                pass
        return comm

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
        if send.dtype not in _DTYPE:              # a byte copy: any 4- / 8-byte element type travels as fp32 words
            assert send.dtype == recv.dtype and send.element_size() % 4 == 0
            send, recv = send.view(torch.float32), recv.view(torch.float32)
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
