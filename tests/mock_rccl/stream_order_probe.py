"""What the two modes of the RCCL double can and cannot see (tests/test_mock_rccl_gpu.py::test_double_sees_stream_order_mistakes).

One process, two communicators on one GPU (ncclCommInitAll), rank 0 sends a buffer to rank 1 on stream A.  Three callers:

  ordered      the consumer stream waits for stream A before it reads the receive buffer          — right under any RCCL
  early_read   the consumer stream reads the receive buffer WITHOUT waiting for stream A           — a race under the real RCCL
  early_pack   another stream overwrites the send buffer right after the call, without waiting     — a race under the real RCCL

Prints one JSON line: for every caller, in how many of the rounds the consumer saw the bytes that were sent.  The default mode of the double
completes every operation inside the call, so all three look right; with $MOCK_RCCL_ASYNC=1 (and a delay in the proxy) the two races lose.

  python stream_order_probe.py <rounds>"""
import ctypes as C
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
U8 = 1


def main():
    rounds = int(sys.argv[1])
    lib = C.CDLL(os.path.join(HERE, "libmockrccl.so"))
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    lib.ncclCommInitAll.argtypes = [C.POINTER(vp), i32, C.POINTER(i32)]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclSend.argtypes = [vp, sz, i32, i32, vp, vp]
    lib.ncclRecv.argtypes = [vp, sz, i32, i32, vp, vp]
    lib.ncclGetLastError.argtypes = [vp]
    lib.ncclGetLastError.restype = C.c_char_p
    lib.mockrccl_stats.argtypes = [C.POINTER(C.c_uint64), i32]
    torch.cuda.set_device(0)
    comms = (vp * 2)()
    devs = (i32 * 2)(0, 0)
    assert lib.ncclCommInitAll(comms, 2, devs) == 0, lib.ncclGetLastError(None)
    n = 1 << 16
    send = torch.zeros(n, dtype=torch.uint8, device="cuda")
    recv = torch.zeros(n, dtype=torch.uint8, device="cuda")
    out = torch.zeros(n, dtype=torch.uint8, device="cuda")
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    right = {}
    for caller in ("ordered", "early_read", "early_pack"):
        good = 0
        for k in range(rounds):
            value = 1 + (k % 200)
            torch.cuda.synchronize()
            recv.zero_(); out.zero_()
            torch.cuda.synchronize()
            with torch.cuda.stream(a):
                send.fill_(value)
            assert lib.ncclGroupStart() == 0
            assert lib.ncclSend(send.data_ptr(), n, U8, 1, comms[0], a.cuda_stream) == 0
            assert lib.ncclRecv(recv.data_ptr(), n, U8, 0, comms[1], a.cuda_stream) == 0
            assert lib.ncclGroupEnd() == 0, lib.ncclGetLastError(comms[0])
            if caller == "early_pack":
                with torch.cuda.stream(b):
                    send.fill_(255)                       # the next step's packing, on a stream that never waited for the send
            if caller != "early_read":
                b.wait_stream(a)
            with torch.cuda.stream(b):
                out.copy_(recv)
            torch.cuda.synchronize()
            good += int(bool((out == value).all().item()))
        right[caller] = good
    torch.cuda.synchronize()
    for c in comms:
        assert lib.ncclCommDestroy(c) == 0, lib.ncclGetLastError(None)
    v = (C.c_uint64 * 15)()
    lib.mockrccl_stats(v, 15)
    print(json.dumps({"rounds": rounds, "right": right, "violations": int(v[7]), "by_proxy": int(v[14]), "groups": int(v[1])}))


if __name__ == "__main__":
    main()
