import ctypes, numpy as np, torch, sys
sys.path.insert(0,'/root/repo')
import lgu_slam_amd as lgu
lib=lgu._lib.load()
np.set_printoptions(linewidth=250, precision=2)
for P in (6,):
    rng=np.random.default_rng(P); n=6*P
    M=rng.standard_normal((n,n)); A=M@M.T+0.5*np.eye(n); b=rng.standard_normal(n)
    lm,ep=1e-4,0.1
    Ld=A.copy(); Ld[np.diag_indices(n)]+=ep+lm*np.diag(A)
    Ad=torch.from_numpy(A).cuda(); bd=torch.from_numpy(b).cuda(); x=torch.full((P,6),7.0,device='cuda')
    work=torch.zeros(int(lib.lgu_ba_solve_blocked_work_doubles(P)),dtype=torch.float64,device='cuda')
    rc=lib.lgu_ba_solve_blocked_f64(ctypes.c_void_p(Ad.data_ptr()),ctypes.c_void_p(bd.data_ptr()),ctypes.c_void_p(x.data_ptr()),ctypes.c_void_p(work.data_ptr()),P,lm,ep,None)
    torch.cuda.synchronize()
    Lw=np.linalg.cholesky(Ld)
    fac=np.tril(Ad.cpu().numpy())
    E=np.abs(fac-Lw)>1e-8
    print("bad entries map (rows x cols), 1 = wrong")
    for i in range(n): print("".join("1" if E[i,j] else "." for j in range(n)))
