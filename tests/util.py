"""Helpers shared by the parity tests."""
import numpy as np


def assert_close(got, want, rtol, atol, name):
    """np.testing.assert_allclose that PRINTS how much of the tolerance the worst element uses (no tolerance more
    than 2 x above its printed measurement): used = max |got - want| / (atol + rtol |want|)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    used = float((np.abs(got - want) / (atol + rtol * np.abs(want))).max()) if got.size else 0.0
    print(f'TOL {name}: worst element uses {used:.2f} of rtol={rtol:g} atol={atol:g}')
    assert used <= 1.0, f'{name}: worst element at {used:.2f} x the tolerance (rtol={rtol:g}, atol={atol:g})'
    return used


def d2_table(q, s, tab):
    """fp32 ((dx*dx + dy*dy) + dz*dz) for every table entry; shadow entries -> +inf."""
    q = np.asarray(q, np.float32)
    s_pad = np.concatenate([np.asarray(s, np.float32), np.full((1, 3), np.inf, np.float32)], 0)
    tab = np.asarray(tab)
    d = q[:, None, :] - s_pad[np.minimum(tab, len(s_pad) - 1)]
    sq = d * d
    out = (sq[..., 0] + sq[..., 1]).astype(np.float32) + sq[..., 2]
    out = out.astype(np.float32)
    out[tab >= len(s)] = np.inf
    return out


def assert_neighbors_equal_mod_ties(ours, ref, q, s):
    """Index-exact except inside groups of EXACTLY equal d2 (the reference's order there is whatever
    std::sort leaves from the KD-tree traversal; ours is ascending index).  Distance profiles must be
    identical, rows duplicate-free, and every disagreement must sit inside an exact tie."""
    ours, ref = np.asarray(ours), np.asarray(ref)
    assert ours.shape == ref.shape
    do, dr = d2_table(q, s, ours), d2_table(q, s, ref)
    assert np.array_equal(do, dr), "distance profiles differ"
    diff = ours != ref
    if diff.any():
        rows = np.where(diff.any(1))[0]
        for r in rows:
            cols = np.where(diff[r])[0]
            for c in cols:
                tie = (c > 0 and do[r, c - 1] == do[r, c]) or (c + 1 < do.shape[1] and do[r, c + 1] == do[r, c]) \
                    or c == do.shape[1] - 1
                assert tie, f"row {r} col {c}: {ours[r, c]} vs {ref[r, c]} without a d2 tie"
            real = ours[r][ours[r] < len(s)]
            assert len(np.unique(real)) == len(real)
    return int(diff.any(1).sum())


class DirectCylindricalNet:
    """TEST INFRASTRUCTURE: the direct-form Cylindrical_Net kernel (tests/native/convnet_direct.hip, its own library) as the
    k-ordered fp32 cross-check of the product's Winograd-domain kernel.  layers: as ops.CylindricalNet."""

    def __init__(self, layers, device):
        import ctypes as C
        import torch
        from buffer_amd import ops
        from tests.native import build as native_build
        self._C, self._torch = C, torch
        self.lib = C.CDLL(native_build.build())
        self.lib.buf_last_error.restype = C.c_char_p
        self.wt, self.bias, cin, cout, relu = [], [], [], [], []
        for w, b, r in layers:
            co, ci = w.shape[0], w.shape[1]
            wt = np.ascontiguousarray(np.transpose(w, (2, 3, 1, 0)).reshape(9 * ci, co), dtype=np.float32)   # [(ky*3+kx)*Cin + c][Cout]
            self.wt.append(torch.from_numpy(ops.mfma_tile_weights(wt)).to(device))
            self.bias.append(torch.from_numpy(np.ascontiguousarray(b, dtype=np.float32)).to(device))
            cin.append(ci); cout.append(co); relu.append(1 if r else 0)
        n = len(layers)
        self._wp = (C.c_void_p * n)(*[t.data_ptr() for t in self.wt])
        self._bp = (C.c_void_p * n)(*[t.data_ptr() for t in self.bias])
        self._ci, self._co, self._re = (C.c_int * n)(*cin), (C.c_int * n)(*cout), (C.c_int * n)(*relu)
        self.cout = cout

    def __call__(self, x):
        C, torch = self._C, self._torch
        x = x.contiguous()
        P = x.shape[0]
        y = torch.empty((P, self.cout[-1], 7, 20), dtype=torch.float32, device=x.device)
        rc = self.lib.buf_test_cylindrical_net_direct(C.c_void_p(x.data_ptr()), P, self._wp, self._bp, self._ci, self._co, self._re,
                                                      C.c_void_p(y.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, self.lib.buf_last_error()
        return y
