"""Time of one call of a leaf slot (include/welship_leaf.h) on the GPU box: a python ctypes loop on one thread.  Measured: 18 us per call
(profiles/r04_leaf_slot_call_time.txt) -- the reason the leaf level is for bring-up and parity, not for throughput."""
import ctypes as C, time, numpy as np, sys
sys.path.insert(0,'.')
lib=C.CDLL('openh264_amd/libwelship.so')
a=np.random.default_rng(1).integers(0,256,(64,96),dtype=np.uint8); b=np.random.default_rng(2).integers(0,256,(64,160),dtype=np.uint8)
pa=C.cast(a.ctypes.data+8*96+8,C.POINTER(C.c_uint8)); pb=C.cast(b.ctypes.data+8*160+8,C.POINTER(C.c_uint8))
f=lib.WelsHipSampleSad16x16; f.restype=C.c_int32
for _ in range(200): f(pa,96,pb,160)
n=5000; t=time.time()
for _ in range(n): f(pa,96,pb,160)
d1=(time.time()-t)/n*1e6
g=lib.WelsHipMcLuma
dst=np.zeros((16,16),np.uint8); pd=C.cast(dst.ctypes.data,C.POINTER(C.c_uint8))
for _ in range(100): g(pb,160,pd,16,C.c_int16(1),C.c_int16(3),16,16)
t=time.time()
for _ in range(n): g(pb,160,pd,16,C.c_int16(1),C.c_int16(3),16,16)
d2=(time.time()-t)/n*1e6
lib.WelsHipLeafCalls.restype=C.c_uint64
print("leaf call latency on the MI355X box (python ctypes loop, one thread): WelsHipSampleSad16x16 %.1f us per call, WelsHipMcLuma 16x16 quarter-sample %.1f us per call; calls served %d"%(d1,d2,lib.WelsHipLeafCalls()))
