import ctypes, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from gpim_amd import _lib
from gpim_amd.kernels import KernelSpec
from oracle import gpim_oracle as O
dev = torch.device("cuda:0"); H = _lib.Handle(); lib = H.lib
for kind, N, Mu, d in (("RBF", 200, 20, 2), ("Matern52", 700, 150, 2), ("RBF", 1500, 300, 3), ("RationalQuadratic", 300, 40, 2)):
    rng = np.random.default_rng(N)
    X = torch.from_numpy(np.unique(rng.integers(0, 40, size=(4 * N, d)), axis=0)[:N].astype(np.float64))
    N = len(X)
    y = torch.from_numpy(np.sin(X.numpy().sum(1) / 6.0) + 0.1 * rng.standard_normal(N))
    ls = [[1.0] * d, [15.0] * d]
    torch.manual_seed(1); kp = O.KernelParams(kind, d, ls)
    spec = KernelSpec(kind, d, ls, jitter=1e-5); u_t = spec.draw_initial_u(torch.Generator().manual_seed(1))
    with torch.no_grad(): kp.u_noise.fill_(-2.0)
    u_t[1 + spec.n_ls] = -2.0
    Xu0 = X[::N // Mu].clone(); Mu = len(Xu0)
    gp = O.SparseGP(X, y, kp, Xu0, 1e-5)
    loss_ref, g_ref = gp.loss_and_grad()
    u = torch.cat([u_t, Xu0.reshape(-1)]).to(dev)
    m = spec.struct(); Xd, yd = X.to(dev).contiguous(), y.to(dev).contiguous()
    P = spec.n_params
    out = torch.empty(1 + P + Mu * d, dtype=torch.float64, device=dev)
    _lib.check(lib.gpimhip_vfe_nll_grad(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, Mu, _lib.ptr(u), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(out.data_ptr() + 8)))
    o = out.cpu()
    gt, gx = o[1:1 + P], o[1 + P:]
    print(f"{kind} N={N} Mu={Mu} d={d}: loss {o[0].item():.10f} ref {loss_ref.item():.10f} rel {abs(o[0].item()-loss_ref.item())/abs(loss_ref.item()):.2e}")
    print("   theta grad dev", gt.numpy(), "\n   theta grad ref", g_ref[:P].numpy())
    print("   xu grad max abs err %.3e (max |ref| %.3e)" % ((gx - g_ref[P:]).abs().max().item(), g_ref[P:].abs().max().item()))
    # predict
    Xs = torch.from_numpy(rng.uniform(0, 40, size=(500, d))); Xsd = Xs.to(dev).contiguous()
    mean = torch.empty(500, dtype=torch.float64, device=dev); var = torch.empty_like(mean)
    _lib.check(lib.gpimhip_predict_vfe(H.h, ctypes.byref(m), _lib.ptr(Xd), _lib.ptr(yd), N, Mu, _lib.ptr(u), _lib.ptr(Xsd), 500, _lib.ptr(mean), _lib.ptr(var)))
    mr, vr = gp.predict(Xs)
    print("   predict: mean err %.3e var err %.3e" % ((mean.cpu() - mr).abs().max().item(), (var.cpu() - vr).abs().max().item()))
