import os, sys, importlib, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'oracle'))
import numpy as np
import oracle_py as O
O._SO = os.environ['CRSDR_SAN_DIR'] + '/liboracle.so'; O._lib = None
synth = importlib.import_module("coherent-rtlsdr_amd.synth")
for mode in (0,1):
    for (nsig,L) in ((3,8192),(7,512),(2,8),(21,2048)):
        e = O.Engine(nsig+1, 2*L, mode); p = synth.RowParams(nsig,L,3,dmax=max(1,L//4))
        for t in range(2):
            rows,_ = synth.make_block(nsig,L,3,t,params=p)
            e.block(rows, seq=t, nthreads=1 if t==0 else 4, lag_mask=None if t==0 else np.ones(nsig+1,dtype=np.uint8))
        e.close()
x = (np.random.randn(3,4096)+1j*np.random.randn(3,4096)).astype(np.complex64)
O.fft(x,-1); O.fft(x,1); O.convto8bit(x[0]); O.magsquared(x[0]); O.conj_dotproduct(x[0],x[1]); O.indexofmax(np.abs(x[0]).astype(np.float32))
lib = C.CDLL(os.environ['CRSDR_SAN_DIR'] + '/libcsynth.so')
lib.csynth_params_create.restype = C.c_void_p; lib.csynth_params_create.argtypes=[C.c_int,C.c_int,C.c_uint64,C.c_int,C.c_int]
lib.csynth_make_block.argtypes=[C.c_void_p,C.c_int,C.c_double,C.c_void_p]; lib.csynth_make_row.argtypes=[C.c_void_p,C.c_int,C.c_int,C.c_double,C.c_void_p]
lib.csynth_params_destroy.argtypes=[C.c_void_p]
pp = lib.csynth_params_create(5,1024,77,-1,0); rows = np.zeros((6,2048),dtype=np.int8); lib.csynth_make_block(pp,2,-1.0,rows.ctypes.data)
r1 = np.zeros(2048,dtype=np.int8)
for r in range(6):
    lib.csynth_make_row(pp,2,r,-1.0,r1.ctypes.data); assert np.array_equal(r1, rows[r])
lib.csynth_params_destroy(pp)
# beamformer restatement: covariance -> noise subspace -> MUSIC scan
rng = np.random.default_rng(5)
mat = rng.integers(-100, 100, size=(22, 2048), dtype=np.int8)
rxx = O.covariance(mat); vec, sv = O.noisesubspace(rxx); pm = O.pmusic2d(vec, 1, 0.5063, 7, 3, 20, 20)
assert np.all(np.isfinite(pm)) and np.all(np.diff(sv) <= 1e-6)
print("sanitizer run ok")
