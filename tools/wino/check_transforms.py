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
