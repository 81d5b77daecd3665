"""Golden vectors for the derivative orders of the fermionic Green's function used by the leaf kernels
(example/benchmark.jl:93-111: (-1)^n/n! d^n/dw^n kernelFermiT(tau, w, beta), n = 1..5).  Lehmann.jl, which
implements kernelFermiT_dw*, is not part of the reference checkout; these vectors come from mpmath at 60 digits
(symbolic-free numerical differentiation of the closed form), so they pin the DEFINITION, not Lehmann.jl's
floating-point output.  Run here (mpmath is in this container); the .npz travels."""
import os
import numpy as np
import mpmath as mp

mp.mp.dps = 60
HERE = os.path.dirname(os.path.abspath(__file__))


def kernel(tau, w, beta):
    tau, w, beta = mp.mpf(tau), mp.mpf(w), mp.mpf(beta)
    if tau == 0:
        tau = mp.mpf("-1e-10")
    if tau > 0:
        return mp.e ** (-w * tau) / (1 + mp.e ** (-w * beta))
    return -mp.e ** (-w * (tau + beta)) / (1 + mp.e ** (-w * beta))


def main():
    rng = np.random.default_rng(2024)
    rows = []
    betas = [1.0, 3.0, 25.0]
    for beta in betas:
        taus = list(rng.uniform(-beta, beta, 14)) + [0.0, beta, -beta * 0.999999, 1e-9, beta * 0.5]
        ws = list(rng.uniform(-6, 6, 10)) + [0.0, 1e-8, -1e-8, 40.0 / beta, -40.0 / beta, 700.0 / beta, -700.0 / beta]
        for tau in taus:
            for w in ws:
                for n in range(1, 6):
                    f = lambda x: kernel(tau, x, beta)
                    d = mp.diff(f, mp.mpf(w), n)
                    val = (-1) ** n * d / mp.factorial(n)
                    rows.append((tau, w, beta, n, float(val)))
    # the parameter set of example/benchmark.jl:10-13,46-51 itself: kF = 1.919, beta = 3.0, w = k^2 - kF^2 with |k| drawn by
    # FermiK(dim, kF, 0.2 kF, 10 kF) (so k from 0 to ~10 kF, most of the weight within 0.2 kF of the Fermi surface), tau in (-beta, beta)
    kF, beta = 1.919, 3.0
    ks = [0.0, 0.5 * kF, 0.8 * kF, 0.9 * kF, 0.95 * kF, kF, 1.05 * kF, 1.1 * kF, 1.2 * kF, 1.5 * kF, 2.0 * kF, 5.0 * kF, 10.0 * kF]
    taus = list(rng.uniform(-beta, beta, 10)) + [0.0, 1e-10, -1e-10, beta * (1 - 1e-12), 0.5 * beta, -0.5 * beta]
    for tau in taus:
        for k in ks:
            w = k * k - kF * kF
            for n in range(1, 6):
                d = mp.diff(lambda x: kernel(tau, x, beta), mp.mpf(w), n)
                rows.append((tau, w, beta, n, float((-1) ** n * d / mp.factorial(n))))
    a = np.array(rows)
    np.savez_compressed(os.path.join(HERE, "green_derive.npz"), tau=a[:, 0], w=a[:, 1], beta=a[:, 2], order=a[:, 3].astype(np.int32), value=a[:, 4])
    print("green_derive.npz:", a.shape[0], "points")


if __name__ == "__main__":
    main()
