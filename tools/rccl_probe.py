"""GPU box: how long a communicator of one rank takes to come up through ss_comm_* (RCCL by dlopen), with and without PyTorch's
RCCL already in the process"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch  # noqa: F401
import ctypes as C
from sandstorm_amd import _lib, backend as be
ctx = be.Context(0)
t0 = time.time()
buf = C.create_string_buffer(128)
be.check(_lib.load().ss_comm_unique_id(buf))
t1 = time.time()
comm = C.c_void_p()
be.check(_lib.load().ss_comm_create(ctx.handle, buf.raw, 0, 1, C.byref(comm)))
t2 = time.time()
print("%s: unique id %.2f s, communicator of one rank %.2f s" % ("with torch" if "torch" in sys.modules else "without torch", t1 - t0, t2 - t1), flush=True)
_lib.load().ss_comm_destroy(comm)
ctx.close()
