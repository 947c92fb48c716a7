#!/usr/bin/env python3
"""F(4x4, 3x3) with other interpolation points: Lavin's (0, +-1, +-2, inf) amplify fp32 rounding through constants up to 5 (B^T) and 8 (A^T);
the same Cook-Toom construction on (0, +-a, +-b, inf) keeps the transforms' STRUCTURE (rows +-a are p +- a q with p = x4 - b^2 x2,
q = x3 - b^2 x1; rows +-b likewise with a^2; rows 0 / inf are x4 - (a^2 + b^2) x2 + a^2 b^2 x0: twelve operations per six values whatever
a and b) and changes only constants.  This tool (1) scans (a, b) for the per-layer error against float64 and (2) repeats
tools/wino_f43_policy_study.py's end-to-end table with the chosen points: which layer groups can then take the 4 x 4 tile?

Usage: python tools/wino_f43_points_study.py [--scan] [--a 3/4 --b 3/2] [--T 5,10] [--threads 8]"""
import argparse
import os
import sys
import time
from fractions import Fraction as Fr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tools', 'experiments', 'probes'))
import numpy as np
import torch
import torch.nn.functional as F

M_, R_, N_ = 4, 3, 6


def matrices(a, b):
    """(A^T [4, 6], G [6, 3], B^T [6, 6]) of Cook-Toom F(4, 3) for correlation on the points (0, a, -a, b, -b, inf), as float64 arrays.
    Position order = point order (0, +a, -a, +b, -b, inf): the order of Lavin's matrices with a = 1, b = 2."""
    P = [Fr(0), Fr(a), -Fr(a), Fr(b), -Fr(b)]
    G, AT = [], [[None] * N_ for _ in range(M_)]
    for j, p in enumerate(P):
        n = Fr(1)
        for l, q in enumerate(P):
            if l != j:
                n *= (p - q)
        G.append([p ** k / n for k in range(R_)])
        for i in range(M_):
            AT[i][j] = p ** i
    G.append([Fr(0), Fr(0), Fr(1)])
    for i in range(M_):
        AT[i][N_ - 1] = Fr(1) if i == M_ - 1 else Fr(0)
    A = np.array([[float(AT[i][j] * G[j][k]) for j in range(N_)] for i in range(M_) for k in range(R_)])
    BT = np.zeros((N_, N_))
    for l in range(N_):
        rhs = np.array([1.0 if l == i + k else 0.0 for i in range(M_) for k in range(R_)])
        BT[:, l] = np.linalg.lstsq(A, rhs, rcond=None)[0]
    BT = np.array([[float(Fr(v).limit_denominator(1 << 16)) for v in row] for row in BT])
    f = lambda m: np.array([[float(v) for v in row] for row in m])
    return f(AT), f(G), BT


def layer_error(AT, G, BT, C, K, H, W, N=2, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    Gt, bt, at = torch.tensor(G), torch.tensor(BT).float(), torch.tensor(AT).float()
    U = torch.einsum('ai,kcij,bj->abkc', Gt, w.double(), Gt).float().reshape(36, K, C)
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 6, 4).unfold(3, 6, 4)
    TH, TW = d.shape[2], d.shape[3]
    t = torch.einsum('ai,nctuij->nctuaj', bt, d)
    V = torch.einsum('nctuaj,bj->abcntu', t, bt).reshape(36, C, N * TH * TW)
    Mm = torch.bmm(U, V).reshape(6, 6, K, N, TH, TW)
    Y = torch.einsum('ibkntu,jb->nktiuj', torch.einsum('ia,abkntu->ibkntu', at, Mm), at).reshape(N, K, H, W)
    return float((Y.double() - ref).abs().max() / ref.abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--scan', action='store_true')
    ap.add_argument('--a', default='3/4')
    ap.add_argument('--b', default='3/2')
    ap.add_argument('--T', default='5,10')
    ap.add_argument('--threads', type=int, default=8)
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    if args.scan:
        print('per-layer error of F(4x4, 3x3) on (0, +-a, +-b, inf), max |y - ref| / max |ref| against float64, N(0, 1) inputs, two seeds each')
        print('%-6s %-6s %-22s %-22s' % ('a', 'b', '64 -> 64 @ 32 x 32', '256 -> 64 @ 16 x 16'))
        for a in (Fr(1, 2), Fr(5, 8), Fr(2, 3), Fr(3, 4), Fr(7, 8), Fr(1)):
            for b in (Fr(1), Fr(5, 4), Fr(4, 3), Fr(3, 2), Fr(7, 4), Fr(2)):
                if b <= a:
                    continue
                m = matrices(a, b)
                e1 = np.mean([layer_error(*m, 64, 64, 32, 32, seed=s) for s in range(2)])
                e2 = np.mean([layer_error(*m, 256, 64, 16, 16, seed=s) for s in range(2)])
                print('%-6s %-6s %-22.2e %-22.2e%s' % (a, b, e1, e2, '   <- Lavin' if (a, b) == (1, 2) else ''), flush=True)
        return
    # ---- the end-to-end table of tools/wino_f43_policy_study.py with these points
    import wino_f43_study as w43
    import wino_f43_policy_study as pol
    AT, G, BT = matrices(Fr(args.a), Fr(args.b))
    w43.AT4, w43.G4, w43.BT4 = torch.tensor(AT).float(), torch.tensor(G), torch.tensor(BT).float()
    xi = torch.randint(-3, 4, (1, 8, 8, 8), generator=torch.Generator().manual_seed(1)).double()
    wi = torch.randint(-2, 3, (4, 8, 3, 3), generator=torch.Generator().manual_seed(2)).double()
    assert float((w43.wino3x3_f43(xi, wi) - F.conv2d(xi, wi, padding=1)).abs().max()) < 1e-9
    print('F(4x4, 3x3) on the points (0, +-%s, +-%s, inf)' % (args.a, args.b), flush=True)
    sys.argv = [sys.argv[0], '--T', args.T, '--threads', str(args.threads)]
    pol.main()


if __name__ == '__main__':
    main()
