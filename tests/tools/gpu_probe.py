"""Quick on-GPU sanity sweep (op-level errors vs the oracle / torch CPU). Prints, never asserts."""
import ctypes, sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import gpim_amd
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec
from oracle import gpim_oracle as O

torch.manual_seed(0)
dev = torch.device("cuda:0")
H = _lib.Handle()
lib = H.lib
print("device", torch.cuda.get_device_name(0))

def potrf_test(n):
    g = torch.Generator().manual_seed(n)
    B = torch.randn(n, n, generator=g, dtype=torch.float64)
    A = B @ B.T / n + torch.eye(n, dtype=torch.float64) * 0.5
    Ad = A.to(dev).contiguous()
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    t = time.time()
    _lib.check(lib.gpimhip_potrf(H.h, _lib.ptr(Ad), n, n, _lib.ptr(info)))
    torch.cuda.synchronize(); dt = time.time() - t
    L = torch.linalg.cholesky(A)
    err = (torch.tril(Ad.cpu()) - L).abs().max().item()
    print(f"potrf n={n}: max|L-Lref|={err:.3e} info={info.item()} t={dt*1e3:.2f} ms")

for n in (5, 16, 100, 128, 129, 300, 700, 1500):
    potrf_test(n)

def make_problem(kind, N, d, seed=0, iso=False, grid=24):
    rng = np.random.default_rng(seed)
    X = rng.integers(0, grid, size=(N * 3, d)).astype(np.float64)
    X = np.unique(X, axis=0)[:N]
    rng.shuffle(X)
    y = np.sin(X.sum(1) / 5.0) + 0.1 * rng.standard_normal(len(X))
    ls = [0.5, grid / 2.0] if iso else [[0.5] * d, [grid / 2.0] * d]
    return torch.from_numpy(X), torch.from_numpy(y), ls

for kind in ("RBF", "Matern52", "RationalQuadratic"):
    for (N, d, iso) in ((7, 2, False), (40, 2, False), (130, 3, False), (300, 2, True), (600, 4, False)):
        X, y, ls = make_problem(kind, N, d, seed=N, iso=iso)
        N = len(X)
        torch.manual_seed(1)
        kp = O.KernelParams(kind, d, ls)
        torch.manual_seed(1)
        spec = KernelSpec(kind, d, ls, jitter=1e-5)
        u = spec.draw_initial_u()
        # make noise smaller so K is not trivially well conditioned
        with torch.no_grad():
            kp.u_noise.fill_(-3.0)
        u[1 + spec.n_ls] = -3.0
        m = spec.struct()
        Xd, yd, ud = X.to(dev).contiguous(), y.to(dev).contiguous(), u.to(dev)
        # kmat
        theta = torch.cat([kp.variance.detach().reshape(1), kp.lengthscale.detach().reshape(-1),
                           (kp.scale_mixture.detach().reshape(1) if kind == "RationalQuadratic" else torch.ones(1, dtype=torch.float64))]).to(dev)
        Kd = torch.empty(N, N, dtype=torch.float64, device=dev)
        _lib.check(lib.gpimhip_kmat(H.h, ctypes.byref(m), _lib.ptr(Xd), N, None, 0, _lib.ptr(theta), 0.25, _lib.ptr(Kd), N))
        Kref = kp.K(X).detach() + 0.25 * torch.eye(N, dtype=torch.float64)
        ek = (Kd.cpu() - Kref).abs().max().item()
        # nll + grad
        gp = O.ExactGP(X, y, kp, 1e-5)
        loss_ref, g_ref = gp.loss_and_grad()
        out = torch.empty(1 + spec.n_params, dtype=torch.float64, device=dev)
        _lib.check(lib.gpimhip_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                        ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
        o = out.cpu()
        el = abs(o[0].item() - loss_ref.item()) / abs(loss_ref.item())
        eg = ((o[1:] - g_ref).abs() / (g_ref.abs() + 1e-12)).max().item()
        # predict
        M = 333
        Xs = torch.from_numpy(np.random.default_rng(5).uniform(0, 24, size=(M, d)))
        Xs[7] = float("nan")
        mean = torch.empty(M, dtype=torch.float64, device=dev); var = torch.empty_like(mean)
        _lib.check(lib.gpimhip_predict_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(ud),
                                             _lib.ptr(Xs.to(dev).contiguous()), M, _lib.ptr(mean), _lib.ptr(var)))
        mref, vref = gp.predict(Xs)
        ok = ~torch.isnan(mref)
        em = (mean.cpu()[ok] - mref[ok]).abs().max().item()
        ev = (var.cpu()[ok] - vref[ok]).abs().max().item()
        nan_ok = bool(torch.isnan(mean.cpu()[7]) and torch.isnan(var.cpu()[7]))
        print(f"{kind:18s} N={N:4d} d={d} iso={iso}: K {ek:.2e} | loss rel {el:.2e} grad rel {eg:.2e} | mean {em:.2e} var {ev:.2e} nanrow {nan_ok}")
        print("     grad dev", o[1:].numpy(), "\n     grad ref", g_ref.numpy())

# fit trajectory
X, y, ls = make_problem("RBF", 60, 2, seed=3)
torch.manual_seed(0); kp = O.KernelParams("RBF", 2, ls); gp = O.ExactGP(X, y, kp, 1e-6)
opt = torch.optim.Adam(kp.parameters(), lr=0.05)
ref_hist = []
for i in range(200):
    opt.zero_grad(); l = gp.loss(); l.backward(); opt.step()
    ref_hist.append([kp.variance.item(), *kp.lengthscale.tolist(), kp.noise.item()])
ref_hist = np.array(ref_hist)
torch.manual_seed(0); spec = KernelSpec("RBF", 2, ls, jitter=1e-6); u = spec.draw_initial_u().to(dev); m = spec.struct()
hist = torch.empty(200, 4, dtype=torch.float64, device=dev); loss = torch.empty(200, dtype=torch.float64, device=dev)
t = time.time()
Xd, yd = X.to(dev).contiguous(), y.to(dev).contiguous()
_lib.check(lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), len(X), _lib.ptr(u), 0.05, 200, _lib.ptr(hist), _lib.ptr(loss)))
dt = time.time() - t
hh = hist.cpu().numpy()
print("fit 200 its N=60: %.1f us/iter; max rel hist err %.3e; final dev %s ref %s" % (dt / 200 * 1e6, (np.abs(hh - ref_hist) / np.abs(ref_hist)).max(), hh[-1], ref_hist[-1]))

# acquisition / topk
mean = torch.randn(1000, dtype=torch.float64); sd = torch.rand(1000, dtype=torch.float64) + 0.01
from scipy.stats import norm
imp = mean.numpy() - 0.3 - 0.01; z = imp / sd.numpy()
ei_ref = imp * norm.cdf(z) + sd.numpy() * norm.pdf(z)
out = torch.empty(1000, dtype=torch.float64, device=dev)
md, sdd = mean.to(dev), sd.to(dev)
_lib.check(lib.gpimhip_acq(H.h, 1, _lib.ptr(md), _lib.ptr(sdd), 1000, 0.3, 0.01, None, _lib.ptr(out)))
print("EI max abs err", np.abs(out.cpu().numpy() - ei_ref).max())
vals = torch.empty(100, dtype=torch.float64, device=dev); idx = torch.empty(100, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
_lib.check(lib.gpimhip_topk(H.h, _lib.ptr(out), 1000, 100, 1, _lib.ptr(vals), _lib.ptr(idx), _lib.ptr(cnt)))
ref_idx = np.argsort(ei_ref)[::-1][:100]
print("topk idx equal:", np.array_equal(idx.cpu().numpy(), ref_idx), "count", cnt.item())

# timing at medium size
for N in (1024, 4096, 16384):
    X, y, ls = make_problem("RBF", N, 2, seed=9, grid=128)
    N = len(X)
    torch.manual_seed(0); spec = KernelSpec("RBF", 2, [[1., 1.], [4., 4.]], jitter=1e-5); u = spec.draw_initial_u().to(dev); m = spec.struct()
    Xd, yd = X.to(dev).contiguous(), y.to(dev).contiguous()
    T = 5
    hist = torch.empty(T, 4, dtype=torch.float64, device=dev)
    for rep in range(2):
        t = time.time()
        _lib.check(lib.gpimhip_fit_exact(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, _lib.ptr(u), 0.1, T, _lib.ptr(hist), None))
        dt = time.time() - t
    print(f"fit N={N}: {dt/T*1e3:.2f} ms/iter  ({N**3/ (dt/T) /1e12:.2f} TFLOP/s on N^3)")
