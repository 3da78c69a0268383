"""One CPU process of the mock-RCCL self-test (tests/test_mock_rccl.py): binds tests/mock_rccl/libmockrccl.so with ctypes in
host-buffer mode ($MOCK_RCCL_HOST_BUFFERS=1) and plays one rank of a scenario.

  python worker.py <scenario> <unique id, hex> <rank> <world>

Prints `RC <code>` and `ERR <text>` for the call the scenario is about; exit status 0 when the scenario went as that rank expects."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class UniqueId(C.Structure):
    # (sixteen words, not `c_char * 128`: ctypes passes a struct that holds one large array by value incorrectly)
    _fields_ = [(f"w{i}", C.c_uint64) for i in range(16)]


def bind():
    lib = C.CDLL(os.path.join(HERE, "libmockrccl.so"))
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    lib.ncclCommInitRank.argtypes = [C.POINTER(vp), i32, UniqueId, i32]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclSend.argtypes = [vp, sz, i32, i32, vp, vp]
    lib.ncclRecv.argtypes = [vp, sz, i32, i32, vp, vp]
    lib.ncclAllReduce.argtypes = [vp, vp, sz, i32, i32, vp, vp]
    lib.ncclGetLastError.argtypes = [vp]
    lib.ncclGetLastError.restype = C.c_char_p
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.mockrccl_stats.argtypes = [C.POINTER(C.c_uint64), i32]
    return lib


U8, I64, U64 = 1, 4, 5
SUM, MAX = 0, 2


def main():
    scen, uid, rank, world = sys.argv[1], bytes.fromhex(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    lib = bind()
    uid_s = UniqueId()
    C.memmove(C.byref(uid_s), uid, 128)
    comm = C.c_void_p()
    rc = lib.ncclCommInitRank(C.byref(comm), world, uid_s, rank)
    if scen == "missing_peer":
        print("RC", rc); print("ERR", lib.ncclGetLastError(None).decode())
        sys.exit(0 if rc != 0 else 1)
    assert rc == 0, lib.ncclGetLastError(None)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    left, right = rank - 1, rank + 1
    rc = 0

    def stats():
        v = (C.c_uint64 * 14)()
        lib.mockrccl_stats(v, 14)
        return list(v)

    if scen == "ok":
        # neighbour messages longer than the 256 KiB ring in both directions, inside one group; then reductions
        n = 700_001
        out_l, out_r = np.full(n, 10 * rank + 1, np.uint8), np.full(n + 3, 10 * rank + 2, np.uint8)
        in_l, in_r = np.zeros(n + 3, np.uint8), np.zeros(n, np.uint8)
        for _ in range(2):
            assert lib.ncclGroupStart() == 0
            if left >= 0:
                assert lib.ncclSend(ptr(out_l), n, U8, left, comm, None) == 0
                assert lib.ncclRecv(ptr(in_l), n + 3, U8, left, comm, None) == 0
            if right < world:
                assert lib.ncclSend(ptr(out_r), n + 3, U8, right, comm, None) == 0
                assert lib.ncclRecv(ptr(in_r), n, U8, right, comm, None) == 0
            rc = lib.ncclGroupEnd()
            assert rc == 0, lib.ncclGetLastError(comm)
            if left >= 0:
                assert (in_l == 10 * left + 2).all()
            if right < world:
                assert (in_r == 10 * right + 1).all()
        a = np.array([rank, 100 - rank, 7, 1 << 40], dtype=np.uint64)
        assert lib.ncclAllReduce(ptr(a), ptr(a), 4, U64, MAX, comm, None) == 0
        assert a.tolist() == [world - 1, 100, 7, 1 << 40], a
        b = np.arange(1000, dtype=np.int64) * (rank + 1)
        c = np.zeros_like(b)
        assert lib.ncclAllReduce(ptr(b), ptr(c), 1000, I64, SUM, comm, None) == 0
        assert (c == np.arange(1000) * (world * (world + 1) // 2)).all()
        # a zero-byte message is a message
        z = np.zeros(1, np.uint8)
        lib.ncclGroupStart()
        if left >= 0:
            lib.ncclSend(ptr(z), 0, U8, left, comm, None)
        if right < world:
            lib.ncclRecv(ptr(z), 0, U8, right, comm, None)
        assert lib.ncclGroupEnd() == 0
        s = stats()
        n_nb = (left >= 0) + (right < world)
        assert s[2] == 2 * n_nb + (left >= 0) and s[3] == 2 * n_nb + (right < world) and s[6] == 2 and s[7] == 0, s
        assert lib.ncclCommDestroy(comm) == 0
        assert stats()[11] == 0
        sys.exit(0)
    if scen == "size_mismatch":
        buf = np.zeros(128, np.uint8)
        rc = lib.ncclSend(ptr(buf), 100, U8, 1, comm, None) if rank == 0 else lib.ncclRecv(ptr(buf), 96, U8, 0, comm, None)
    elif scen == "missing_recv":
        buf = np.zeros(128, np.uint8)
        if rank == 0:
            rc = lib.ncclSend(ptr(buf), 100, U8, 1, comm, None)
        else:
            import time
            time.sleep(4.0)
            rc = 1          # this rank has nothing to report: the sender's deadline is the subject
    elif scen == "group_mismatch":
        # rank 0 closes a group after each send; rank 1 receives both in one group: legal bytes, different call sequences
        buf = np.zeros(64, np.uint8)
        if rank == 0:
            rc = lib.ncclSend(ptr(buf), 64, U8, 1, comm, None) or lib.ncclSend(ptr(buf), 64, U8, 1, comm, None)
        else:
            lib.ncclGroupStart()
            lib.ncclRecv(ptr(buf), 64, U8, 0, comm, None)
            lib.ncclRecv(ptr(buf), 64, U8, 0, comm, None)
            rc = lib.ncclGroupEnd()
    elif scen == "coll_mismatch":
        a = np.zeros(8, np.uint64)
        rc = lib.ncclAllReduce(ptr(a), ptr(a), 4 if rank == 0 else 8, U64, MAX, comm, None)
    elif scen == "interleave_mismatch":
        # rank 0: message then collective; rank 1: collective then message — a real RCCL deadlocks on one communicator
        a, buf = np.zeros(4, np.uint64), np.zeros(64, np.uint8)
        if rank == 0:
            lib.ncclGroupStart(); lib.ncclSend(ptr(buf), 64, U8, 1, comm, None); lib.ncclAllReduce(ptr(a), ptr(a), 4, U64, MAX, comm, None)
            rc = lib.ncclGroupEnd()     # (one group so that this rank's calls cannot block each other: the check is the epoch comparison)
            rc = rc or lib.ncclAllReduce(ptr(a), ptr(a), 4, U64, MAX, comm, None)
        else:
            rc = lib.ncclAllReduce(ptr(a), ptr(a), 4, U64, MAX, comm, None)
            lib.ncclGroupStart(); lib.ncclRecv(ptr(buf), 64, U8, 0, comm, None)
            rc = rc or lib.ncclGroupEnd()
            rc = rc or lib.ncclAllReduce(ptr(a), ptr(a), 4, U64, MAX, comm, None)
    elif scen == "open_group_at_destroy":
        lib.ncclGroupStart()
        rc = lib.ncclCommDestroy(comm)
        print("RC", rc); print("ERR", lib.ncclGetLastError(None).decode())
        sys.exit(0 if rc != 0 and stats()[7] >= 1 else 1)
    elif scen == "group_end_without_start":
        rc = lib.ncclGroupEnd()
    else:
        raise SystemExit("unknown scenario " + scen)
    print("RC", rc)
    print("ERR", (lib.ncclGetLastError(comm) or b"").decode())
    sys.exit(0 if rc != 0 else 1)


if __name__ == "__main__":
    main()
