"""Numerical check of the 3D Winograd F(2x2x2, 3x3x3) formulation used by conv_wino_kernel (fp64 and fp32 error)."""
import numpy as np
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
rng = np.random.RandomState(0)
Cin, Cout = 5, 3
d = rng.randn(4, 4, 4, Cin)
g = rng.randn(3, 3, 3, Cin, Cout)
# direct: 2x2x2 outputs, correlation
ref = np.zeros((2, 2, 2, Cout))
for a in range(2):
    for b in range(2):
        for c in range(2):
            ref[a, b, c] = np.einsum('xyzi,xyzio->o', d[a:a + 3, b:b + 3, c:c + 3], g)
for dt in (np.float64, np.float32):
    V = np.einsum('ax,by,cz,xyzi->abci', BT, BT, BT, d).astype(dt)
    U = np.einsum('ax,by,cz,xyzio->abcio', G, G, G, g).astype(dt)
    M = np.einsum('abci,abcio->abco', V, U).astype(dt)
    Y = np.einsum('pa,qb,rc,abco->pqro', AT, AT, AT, M)
    print(dt.__name__, 'max abs err', np.abs(Y - ref).max(), 'rel', np.abs(Y - ref).max() / np.abs(ref).max())

# ---- F(3,2): weight-gradient form.  y_k = sum_j d[k+j] g[j], k = 0..2, j = 0..1 (d: 4 inputs, g: the 2 dY values of a tile)
A32 = np.array([[1, 0], [1, 1], [1, -1], [0, -1]], float)                 # "filter" transform (applied to dY pairs)
GT32 = np.array([[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]], float)  # output transform (4 -> 3 taps)
d = rng.randn(4); g = rng.randn(2)
ref = np.array([d[k] * g[0] + d[k + 1] * g[1] for k in range(3)])
y = GT32 @ ((A32 @ g) * (BT @ d))
print('F(3,2) 1D err', np.abs(y - ref).max())
# 2D (h, w) with a direct third dimension, accumulated over tiles: dW[kh][kw] = sum_tiles ...
X = rng.randn(6, 6); dY = rng.randn(4, 4)            # 2x2 tiles of 2x2 outputs; X has the +-1 halo
ref2 = np.zeros((3, 3))
for kh in range(3):
    for kw in range(3):
        ref2[kh, kw] = sum(X[oh + kh, ow + kw] * dY[oh, ow] for oh in range(4) for ow in range(4))
M = np.zeros((4, 4))
for th in range(2):
    for tw in range(2):
        V = BT @ X[2 * th:2 * th + 4, 2 * tw:2 * tw + 4] @ BT.T
        W = A32 @ dY[2 * th:2 * th + 2, 2 * tw:2 * tw + 2] @ A32.T
        M += V * W
print('F(3x3,2x2) accumulated err', np.abs(GT32 @ M @ GT32.T - ref2).max())
