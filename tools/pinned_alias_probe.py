#!/usr/bin/env python3
"""hipPointerGetAttributes of a bpgpu_host_alloc buffer (and of an interior address): is it seen as device-mapped host memory?"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpc_bulletproof_amd as mb


class Attr(C.Structure):
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p),
                ("isManaged", C.c_int), ("allocationFlags", C.c_uint)]


hip = C.CDLL("libamdhip64.so")
gpu = mb.BpGpu(0)
p = mb.lib.host_alloc(1 << 20)
for off in (0, 4096, 12345):
    a = Attr()
    rc = hip.hipPointerGetAttributes(C.byref(a), C.c_void_p(p.value + off))
    print(off, "rc", rc, "type", a.type, "dev", a.device, hex(a.devicePointer or 0), hex(a.hostPointer or 0), hex(p.value + off))
