import ctypes as C, time, numpy as np
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes=[C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes=[C.c_void_p]
hip.hipMalloc.argtypes=[C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes=[C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemcpyAsync.argtypes=[C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hip.hipHostMalloc.argtypes=[C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
d=C.c_void_p(); assert hip.hipMalloc(C.byref(d), 128<<20)==0
hip.hipDeviceSynchronize()
for mb in (2, 16, 64, 96):
    a=np.random.randint(0,255,size=mb<<20,dtype=np.uint8)
    p=a.ctypes.data
    for rep in range(3):
        t0=time.perf_counter(); rc=hip.hipHostRegister(p, a.nbytes, 0); t1=time.perf_counter()
        assert rc==0, rc
        hip.hipMemcpy(d, p, a.nbytes, 1); t2=time.perf_counter()
        hip.hipHostUnregister(p); t3=time.perf_counter()
        hip.hipMemcpy(d, p, a.nbytes, 1); t4=time.perf_counter()
        print(f"{mb} MB: register {1e3*(t1-t0):.2f} ms, copy(pinned) {1e3*(t2-t1):.2f} ms = {a.nbytes/(t2-t1)/1e9:.1f} GB/s, unregister {1e3*(t3-t2):.2f} ms, copy(pageable) {1e3*(t4-t3):.2f} ms = {a.nbytes/(t4-t3)/1e9:.1f} GB/s")
    h=C.c_void_p(); assert hip.hipHostMalloc(C.byref(h), a.nbytes, 0)==0
    t0=time.perf_counter(); C.memmove(h, p, a.nbytes); t1=time.perf_counter()
    print(f"   host memcpy into pinned: {1e3*(t1-t0):.2f} ms = {a.nbytes/(t1-t0)/1e9:.1f} GB/s")
