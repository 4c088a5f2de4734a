import ctypes as C, sys, time
import numpy as np, torch
lib = C.CDLL("jpezy_amd/libjpezy_hip.so")
lib.jpezy_ctx_create.restype = C.c_void_p
lib.jpezy_fdct_quant.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]
lib.jpezy_ctx_set_host_chunk_bytes.argtypes = [C.c_void_p, C.c_size_t]
ctx = lib.jpezy_ctx_create(0)
p8 = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(0)
for (W, H, F) in ((4096, 4096, 1), (1920, 1080, 32)):
    r, g, b = (rng.integers(0, 256, W * H * F, dtype=np.uint8) for _ in range(3))
    ncoef = ((W + 15) // 16) * ((H + 15) // 16) * 6 * 64 * F
    out = np.empty(ncoef, dtype=np.int16)
    for chunk in (2 << 20, 4 << 20, 8 << 20, 16 << 20, 32 << 20):
        lib.jpezy_ctx_set_host_chunk_bytes(ctx, chunk)
        for _ in range(2):
            assert lib.jpezy_fdct_quant(ctx, p8(r), p8(g), p8(b), W, H, 0, F, p8(out)) == 0
        ts = []
        for _ in range(5):
            t = time.perf_counter(); assert lib.jpezy_fdct_quant(ctx, p8(r), p8(g), p8(b), W, H, 0, F, p8(out)) == 0; ts.append(time.perf_counter() - t)
        tf = []
        for _ in range(4):
            rr, gg, bb, oo = r.copy(), g.copy(), b.copy(), np.empty(ncoef, dtype=np.int16)
            t = time.perf_counter(); assert lib.jpezy_fdct_quant(ctx, p8(rr), p8(gg), p8(bb), W, H, 0, F, p8(oo)) == 0; tf.append(time.perf_counter() - t)
        print(f"{F} x {W}x{H} chunk {chunk>>20} MB: reused {min(ts)*1e3:.2f} ms, fresh {min(tf)*1e3:.2f} ms")
