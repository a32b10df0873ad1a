"""Does what ran earlier in the process change the speed of the single-precision headline fit?  (bench `extra`
C2_single_precision went 4.54 -> 5.90 s once C3 / C5 ran their slices on concurrent host threads; the fit kernels are
unchanged: tools/r3_bisect.sh.)     python tools/r3_single_ctx.py <pre> [<pre> ...]
pre: none | c1 | c3serial | c3conc | c4 | c5conc | kron | gc   (executed in order before the timed single-precision fit + predict)"""
import gc, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("SHOW_QUEUES"):
    print("GPU_MAX_HW_QUEUES before importing gpim_amd:", os.environ.get("GPU_MAX_HW_QUEUES"), "| torch already imported:", "torch" in sys.modules,
          "| cuda initialised:", torch.cuda.is_initialized(), flush=True)
import gpim_amd as gpim
if os.environ.get("SHOW_QUEUES"):
    print("GPU_MAX_HW_QUEUES after importing gpim_amd: ", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
from gpim_amd import dist as gdist
import bench
from problems import ckpfm_cube, hyperspectral_cube, lattice_image

sync = torch.cuda.synchronize
_mode = os.environ.get("MAINSTREAM")
if _mode == "cumask":
    # everything on a stream created with a CU mask that enables EVERY compute unit: such a stream gets a hardware
    # queue of its own (like the engine's bulk stream), which no other stream of the process can be mapped onto
    import ctypes
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    torch.cuda.init(); torch.zeros(1, device="cuda")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ncu + 31) // 32
    mask = (ctypes.c_uint32 * words)(*([0xFFFFFFFF] * words))
    sp = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), ctypes.c_uint32(words), mask)
    assert rc == 0, rc
    _ms = torch.cuda.ExternalStream(sp.value); torch.cuda.set_stream(_ms)
    print("main stream: CU-masked (all %d CUs) external stream" % ncu, flush=True)
elif _mode == "prio":
    _ms = torch.cuda.Stream(priority=-1); torch.cuda.set_stream(_ms)
    print("main stream: high-priority torch stream", flush=True)
elif _mode:
    # everything (pre steps and the timed fits) on ONE non-default torch stream instead of the legacy default stream
    _ms = torch.cuda.Stream(); torch.cuda.set_stream(_ms)
    print("main stream: torch pool stream", flush=True)
for pre in sys.argv[1:]:
    t0 = time.perf_counter()
    if pre == "c3serial":
        cube, _ = hyperspectral_cube()
        gdist.reconstruct_slices(cube, axis=-1, batch=64, batch_concurrency=1, **dict(bench.C3, iterations=20))
    elif pre == "c3conc":
        cube, _ = hyperspectral_cube()
        gdist.reconstruct_slices(cube, axis=-1, batch=16, batch_concurrency=4, **dict(bench.C3, iterations=20))
    elif pre == "c5conc":
        gdist.reconstruct_slices(ckpfm_cube(), axis=-1, sparse=True, indpoints=512, kernel="RBF", learning_rate=0.05, iterations=20)
    elif pre == "c1":
        from problems import spiral_pfm_image
        R = spiral_pfm_image()
        gpim.reconstructor(gpim.utils.get_sparse_grid(R), R, gpim.utils.get_full_grid(R), **dict(bench.C1, iterations=30, verbose=0)).run()
    elif pre == "c4":
        import tempfile
        from problems import notebook_problem
        trial_func, Z = notebook_problem(4)
        bo = gpim.boptimizer(gpim.utils.get_sparse_grid(Z), Z, gpim.utils.get_full_grid(Z), trial_func, acquisition_function="ei",
                             exploration_steps=5, verbose=0, filename=os.path.join(tempfile.mkdtemp(), "bo"))
        bo.run(); del bo
    elif pre == "kron":
        R5 = ckpfm_cube()[..., 0]
        Xf5 = gpim.utils.get_full_grid(R5)
        gpim.reconstructor(Xf5, R5, Xf5, structured=True, verbose=0, kernel="RBF", learning_rate=0.05, iterations=20).run()
    elif pre.startswith("streams") and pre[7:].isdigit():
        # k extra torch streams (streams4: four), each touched once from the main thread: no library call involved
        for _ in range(int(pre[7:])):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                torch.zeros(1024, device="cuda").add_(1.0)
            st.synchronize()
    elif pre in ("streams4d", "streams4k"):
        # four raw HIP streams (hipStreamCreateWithFlags, non-blocking), each touched once through torch; "d": destroyed
        # again afterwards, "k": kept alive -- does the slow-down follow the LIVE streams of the process?
        import ctypes
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        raw = []
        for _ in range(4):
            sp = ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(sp), ctypes.c_uint(1)) == 0
            raw.append(sp)
            ext = torch.cuda.ExternalStream(sp.value)
            with torch.cuda.stream(ext):
                torch.zeros(1024, device="cuda").add_(1.0)
            ext.synchronize(); del ext
        if pre == "streams4d":
            for sp in raw:
                assert hip.hipStreamDestroy(sp) == 0
        else:
            _keep = globals().setdefault("_kept_streams", []); _keep.extend(raw)
    elif pre == "c3groups":
        # the four batches of 16 one after another on the main thread, a fresh handle each (no threads, no extra streams)
        cube, _ = hyperspectral_cube()
        gdist.reconstruct_slices(cube, axis=-1, batch=16, batch_concurrency=1, **dict(bench.C3, iterations=20))
    elif pre == "gc":
        gc.collect()
    sync(); print("pre %-8s %.2f s" % (pre, time.perf_counter() - t0), flush=True)

if os.environ.get("MIDN"):
    # the mid-size regime (one captured iteration replayed): C1 on the reference's spiral scan, and one lock-step batch of C3
    from problems import spiral_pfm_image
    R = spiral_pfm_image()
    Xm, Xfm = gpim.utils.get_sparse_grid(R), gpim.utils.get_full_grid(R)
    gpim.reconstructor(Xm, R, Xfm, **dict(bench.C1, iterations=3, verbose=0)).run()
    rec = gpim.reconstructor(Xm, R, Xfm, **dict(bench.C1, iterations=100, verbose=0))
    sync(); t0 = time.perf_counter(); rec.train(); sync()
    print("C1 size (N=4212, graph replay): %.3f ms/iter" % ((time.perf_counter() - t0) / 100 * 1e3), flush=True)
    del rec
    cube, _ = hyperspectral_cube()
    Hm = gpim._lib.Handle() if hasattr(gpim, "_lib") else None
    from gpim_amd import _lib as _l
    Hm = _l.Handle()
    gdist.reconstruct_slices(cube, axis=-1, batch=64, handle=Hm, **dict(bench.C3, iterations=3))
    sync(); t0 = time.perf_counter()
    gdist.reconstruct_slices(cube, axis=-1, batch=64, handle=Hm, **dict(bench.C3, iterations=100))
    sync(); print("C3 one batch of 64 (N=1207): %.3f ms/iter incl. predict" % ((time.perf_counter() - t0) / 100 * 1e3), flush=True)
W = bench.WORKLOAD
R2, _ = lattice_image(size=W["size"], frac=W["frac"], seed=1)
X2, Xf2 = gpim.utils.get_sparse_grid(R2), gpim.utils.get_full_grid(R2)
T = int(os.environ.get("ITERS", "20"))
for prec in os.environ.get("PRECS", "single,double").split(","):
    kw = dict(kernel=W["kernel"], lengthscale=W["lengthscale"], learning_rate=W["learning_rate"], verbose=0, seed=0, precision=prec)
    dt_ = np.float32 if prec == "single" else np.float64
    rec = gpim.reconstructor(X2.astype(dt_), R2.astype(dt_), Xf2.astype(dt_), iterations=2, **kw)
    rec.run(); rec.iterations = T
    sync(); t0 = time.perf_counter(); rec.train(); sync(); t1 = time.perf_counter(); rec.predict(); sync(); t2 = time.perf_counter()
    print("%s: train %.2f ms/iter, predict %.3f s" % (prec, (t1 - t0) / T * 1e3, t2 - t1), flush=True)
    if os.environ.get("STAGES"):
        import ctypes
        lib, h = rec._handle.lib, rec._handle.h
        ms, cnt = ctypes.c_double(), ctypes.c_int64()
        lib.gpimhip_timing_enable(h, 1)
        for st_ in range(4):
            lib.gpimhip_timing_read(h, st_, ctypes.byref(ms), ctypes.byref(cnt))
        rec.iterations = 6; rec.train(); sync()
        lib.gpimhip_timing_enable(h, 0)
        parts = []
        for st_, nm in enumerate(["potrf", "trtri", "lauum"]):
            lib.gpimhip_timing_read(h, st_, ctypes.byref(ms), ctypes.byref(cnt))
            parts.append("%s %.2f" % (nm, ms.value / max(cnt.value, 1)))
        print("   stages (ms per call): " + ", ".join(parts), flush=True)
    del rec
