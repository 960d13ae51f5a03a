"""CPU: the algebra of csrc/conv_wino4.hip restated in NumPy -- Winograd F(4x4,3x3) with the interpolation points 0, +-1, +-2, inf.

The kernel computes  Y = A^T [ (G g G^T) (.) (B^T d B) ] A  per 4x4 output tile and splits the 36 positions over two waves by
rows of the transformed patch: wave xh holds the rows 3xh .. 3xh+2, forms  P_xh = A^T[:, rows] (M[rows, :] A)  and the two
partial tiles are added after one exchange.  Checked here: the matrices the packer / the kernel's operation lists spell out,
the split, the exactness in float64, and the size of the rounding error in float32 (what the GPU tolerance in
tests/test_gpu_ops.py::test_conv3x3_winograd_f4_random_shapes is set against)."""
import numpy as np

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
               [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
              [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)       # pack_wino4_kernel
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)


def half_ops(x, lo):
    """the six packed operations of half a 1-D input transform as the kernel issues them (half_op<LO, K>), x = five inputs"""
    if lo:                                   # rows 0..2 of B^T from x0..x4
        ta = -5 * x[2] + x[4]
        o0 = 4 * x[0] + ta
        ta = -4 * x[2] + x[4]
        tb = -4 * x[1] + x[3]
        return o0, ta + tb, ta - tb
    ta, tb = x[3] - x[1], x[2] - x[0]        # rows 3..5 from x1..x5
    o0, o1 = 2 * tb + ta, -2 * tb + ta
    return o0, o1, 4 * x[0] + (-5 * x[2] + x[4])


def direct(d, g):
    return np.array([[np.sum(d[i:i + 3, j:j + 3] * g) for j in range(4)] for i in range(4)])


def test_matrices_and_operation_lists():
    rng = np.random.default_rng(0)
    for _ in range(20):
        d, g = rng.standard_normal((6, 6)), rng.standard_normal((3, 3))
        y = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        np.testing.assert_allclose(y, direct(d, g), rtol=0, atol=1e-12)
        x = rng.standard_normal(6)
        np.testing.assert_allclose(np.concatenate([half_ops(x[0:5], True), half_ops(x[1:6], False)]), BT @ x, rtol=0, atol=1e-13)


def test_split_over_two_waves_by_rows_of_the_transformed_patch():
    rng = np.random.default_rng(1)
    d, g = rng.standard_normal((6, 6)), rng.standard_normal((3, 3))
    M = (G @ g @ G.T) * (BT @ d @ BT.T)
    parts = []
    for xh in (0, 1):
        rows = slice(3 * xh, 3 * xh + 3)
        R = M[rows] @ AT.T                                     # own rows: 6 -> 4 along nu
        parts.append(AT[:, rows] @ R)                          # partial tile from the own rows of A^T
    np.testing.assert_allclose(parts[0] + parts[1], direct(d, g), rtol=0, atol=1e-12)
    # the shapes the epilogue relies on: rows 0..2 of A^T are (1,0,0,0) (1,1,1,1) (1,-1,1,-1), rows 3..5 (1,2,4,8) (1,-2,4,-8) (0,0,0,1)
    np.testing.assert_array_equal(AT[:, :3].T, [[1, 0, 0, 0], [1, 1, 1, 1], [1, -1, 1, -1]])
    np.testing.assert_array_equal(AT[:, 3:].T, [[1, 2, 4, 8], [1, -2, 4, -8], [0, 0, 0, 1]])


def test_float32_rounding_error_of_one_layer():
    """96 -> 96 channels, zero-mean random data: F(4x4,3x3) in float32 lands at 1e-6 relative L2 of the float64 convolution"""
    rng = np.random.default_rng(2)
    C, K, T = 96, 96, 40
    d = rng.standard_normal((T, C, 6, 6)).astype(np.float32)
    g = (rng.standard_normal((K, C, 3, 3)) * (2 / (9 * C)) ** 0.5).astype(np.float32)
    ref = np.einsum('tcij,kcij->tk', d[:, :, 1:4, 1:4].astype(np.float64), g.astype(np.float64))       # output pixel (1, 1) of every tile
    U = np.einsum('ia,kcab,jb->kcij', G, g.astype(np.float64), G).astype(np.float32)                   # packed in double, rounded once
    B32, A32 = BT.astype(np.float32), AT.astype(np.float32)
    V = np.einsum('ia,tcab,jb->tcij', B32, d, B32).astype(np.float32)
    Mm = np.einsum('kcij,tcij->tkij', U, V).astype(np.float32)
    Y = np.einsum('ia,tkab,jb->tkij', A32, Mm, A32).astype(np.float32)
    err = np.linalg.norm(Y[:, :, 1, 1] - ref) / np.linalg.norm(ref)
    assert 1e-8 < err < 4e-6, err
