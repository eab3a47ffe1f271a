#!/usr/bin/env python3
"""hipIpcOpenMemHandle between two PyTorch-hosted processes on one GPU (the HIP / ROCr copies PyTorch bundles, not /opt/rocm's), for
allocations of the sizes given: is the handle opened, and how fast (tools/native/ipc_open_probe.hip asks the same of /opt/rocm's copies).
    python tools/ipc_open_probe.py [--prefill GB] size [size ...]"""
import ctypes as C
import os
import subprocess
import sys
import time


def hip():
    import torch
    lib = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    return torch, lib


def exporter(n, prefill_gb, path):
    torch, L = hip()
    torch.cuda.set_device(0)
    junk = []
    if prefill_gb:                                   # an address space that has seen allocations come and go, like a rank's
        a = torch.empty(int(prefill_gb * 2**30), dtype=torch.uint8, device="cuda")
        junk.append(torch.empty(3 << 30, dtype=torch.uint8, device="cuda"))
        del a
        torch.cuda.empty_cache()
        p0 = C.c_void_p()
        assert L.hipMalloc(C.byref(p0), C.c_size_t(int(prefill_gb * 2**29))) == 0
    p = C.c_void_p()
    assert L.hipMalloc(C.byref(p), C.c_size_t(n)) == 0, "hipMalloc"
    h = (C.c_uint8 * 64)()
    rc = L.hipIpcGetMemHandle(h, p)
    assert rc == 0, "hipIpcGetMemHandle %d" % rc
    with open(path + ".tmp", "wb") as f:
        f.write(bytes(h))
    os.rename(path + ".tmp", path)
    t0 = time.time()
    while not os.path.exists(path + ".done") and time.time() - t0 < 40:
        time.sleep(0.01)


def importer(n, prefill_gb, path):
    torch, L = hip()
    torch.cuda.set_device(0)
    if prefill_gb:
        a = torch.empty(int(prefill_gb * 2**30), dtype=torch.uint8, device="cuda")
        b = torch.empty(3 << 30, dtype=torch.uint8, device="cuda")
        del a
        torch.cuda.empty_cache()
        p0 = C.c_void_p()
        assert L.hipMalloc(C.byref(p0), C.c_size_t(int(prefill_gb * 2**29))) == 0
        own = C.c_void_p()
        assert L.hipMalloc(C.byref(own), C.c_size_t(n)) == 0
    t0 = time.time()
    while not os.path.exists(path) and time.time() - t0 < 40:
        time.sleep(0.01)
    class Handle(C.Structure):
        _fields_ = [("reserved", C.c_uint8 * 64)]
    h = Handle.from_buffer_copy(open(path, "rb").read())
    q = C.c_void_p()
    t0 = time.time()
    L.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
    rc = L.hipIpcOpenMemHandle(C.byref(q), h, 1)
    print("  size %d (%.1f MB): hipIpcOpenMemHandle -> %d in %.3f s" % (n, n / 2**20, rc, time.time() - t0), flush=True)
    open(path + ".done", "w").close()


def main():
    if sys.argv[1] in ("--exporter", "--importer"):
        return (exporter if sys.argv[1] == "--exporter" else importer)(int(sys.argv[2]), float(sys.argv[3]), sys.argv[4])
    args = sys.argv[1:]
    prefill = 0.0
    if args[0] == "--prefill":
        prefill, args = float(args[1]), args[2:]
    for k, n in enumerate(int(a, 0) for a in args):
        path = "/tmp/ipc_probe_%d_%d" % (os.getpid(), k)
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), role, str(n), str(prefill), path]) for role in ("--exporter", "--importer")]
        try:
            ps[1].wait(timeout=25)
        except subprocess.TimeoutExpired:
            print("  size %d (%.1f MB): no answer within 25 s" % (n, n / 2**20), flush=True)
            ps[1].kill()
        open(path + ".done", "w").close()
        ps[0].wait(timeout=60)
        for f in (path, path + ".done"):
            if os.path.exists(f):
                os.remove(f)


if __name__ == "__main__":
    main()
