"""CPU: the algebra of the folded 1x1 conv -> training-mode batch norm -> (+ residual) -> ReLU (csrc/foldbn.hip, layers.conv_bn_fold)
restated in float64 torch and held against torch's own autograd of the three-op chain (models/resnet.py:281-296): statistics from
the column sums and the Gram matrix of the conv's input, every backward quantity from G = g^T x -- the conv output and its gradient
never appear on the folded side."""
import pytest
import torch


@pytest.mark.parametrize("n,K,N,relu,res", [(500, 16, 48, True, True), (300, 32, 32, False, False), (257, 8, 64, True, False)])
def test_fold_algebra_equals_autograd(n, K, N, relu, res):
    torch.manual_seed(n + K)
    dt = torch.float64
    X = torch.relu(torch.randn(n, K, dtype=dt) + 0.3).requires_grad_(True)
    W = (torch.randn(N, K, dtype=dt) * 0.2).requires_grad_(True)           # [Cout][Cin]
    gamma = (torch.rand(N, dtype=dt) + 0.5).requires_grad_(True)
    beta = torch.randn(N, dtype=dt).requires_grad_(True)
    R = torch.randn(n, N, dtype=dt).requires_grad_(True)
    eps = 1e-5
    # the three-op chain, by autograd
    Z = X @ W.t()
    mu, var = Z.mean(0), Z.var(0, unbiased=False)
    invstd = (var + eps).rsqrt()
    U = (Z - mu) * invstd * gamma + beta + (R if res else 0)
    Y = torch.relu(U) if relu else U
    dy = torch.randn(n, N, dtype=dt)
    Y.backward(dy)
    with torch.no_grad():
        # forward from the moments of X
        S, A1 = X.t() @ X, X.sum(0)
        P = W @ S
        sz, szz = W @ A1, (W * P).sum(1)
        mu2 = sz / n
        inv2 = (szz / n - mu2 ** 2 + eps).rsqrt()
        a, b = gamma * inv2, beta - mu2 * gamma * inv2
        U2 = (X @ W.t()) * a + b + (R if res else 0)           # what the conv tile's epilogue computes
        Y2 = torch.relu(U2) if relu else U2
        assert torch.allclose(mu2, mu, atol=1e-12) and torch.allclose(inv2, invstd, rtol=1e-10) and torch.allclose(Y2, Y, atol=1e-10)
        # backward from g and G = g^T X
        g = dy * (Y2 > 0) if relu else dy
        sg = g.sum(0)
        G = g.t() @ X
        sgz = inv2 * ((W * G).sum(1) - mu2 * sg)
        m, q = sg / n, sgz / n
        dW = a[:, None] * (G - m[:, None] * A1[None] - (q * inv2)[:, None] * (P - mu2[:, None] * A1[None]))
        d = a * inv2 * q
        T = W.t() @ (d[:, None] * W)
        c0 = (d * mu2 - a * m) @ W
        dX = g @ (a[:, None] * W) - X @ T + c0
        assert torch.allclose(sgz, gamma.grad, atol=1e-9) and torch.allclose(sg, beta.grad, atol=1e-10)
        assert torch.allclose(dW, W.grad, atol=1e-9) and torch.allclose(dX, X.grad, atol=1e-9)
        if res:
            assert torch.allclose(g, R.grad, atol=1e-12)
        # the CENTRED form the kernels use since round 6 (rcf_fold_fwd_f32): P_c = W (S - A1 A1^T / n) -- the variance without a
        # subtraction, P - mean A1 without one either; and its two-rank recombination (SyncBN): each rank centres on ITS means, the
        # all-reduced sums are the uncentred ones put back together, the backward pass re-centres P_c on the global mean
        Pc = W @ (S - torch.outer(A1, A1) / n)
        assert torch.allclose((W * Pc).sum(1) / n, var, atol=1e-12) and torch.allclose(Pc, P - mu2[:, None] * A1[None], atol=1e-9)
        h = n // 2
        parts = []
        for Xr in (X[:h], X[h:]):
            Sr, Ar, nr = Xr.t() @ Xr, Xr.sum(0), Xr.shape[0]
            Pcr = W @ (Sr - torch.outer(Ar, Ar) / nr)
            szr = W @ Ar
            parts.append((Pcr, Ar, szr / nr, szr, (W * Pcr).sum(1) + szr ** 2 / nr))
        sz_all, szz_all = parts[0][3] + parts[1][3], parts[0][4] + parts[1][4]
        assert torch.allclose(sz_all / n, mu, atol=1e-12) and torch.allclose(szz_all / n - (sz_all / n) ** 2, var, atol=1e-11)
        dW2 = sum(a[:, None] * ((gr.t() @ Xr) - m[:, None] * Ar[None] - (q * inv2)[:, None] * (Pcr - (mu2 - mloc)[:, None] * Ar[None]))
                  for (Pcr, Ar, mloc, _, _), gr, Xr in zip(parts, (g[:h], g[h:]), (X[:h], X[h:])))
        assert torch.allclose(dW2, W.grad, atol=1e-9)
