"""kornia.geometry.conversions.angle_axis_to_rotation_matrix, restated (kornia is an unpinned pip dependency of the
reference, README.md:33; absent from /root/reference and from this image).

Arithmetic of the published function: theta^2 = <aa, aa>; the axis is aa / (theta + 1e-6) -- NOT exactly unit length --
and R = cos*I + (1 - cos) * w w^T + sin * [w]_x written out element by element; rows with theta^2 <= 1e-6 take the
first-order form I + [aa]_x.  BUFFER only ever passes (0, 0, theta) with theta = ind * 2pi/20 + 1e-6
(models/BUFFER.py:295-299), for which the result is Rz(theta) up to the (1 - 1e-6/theta) factor on sin and a
2e-6/theta * (1 - cos) deficit on R[2,2]: the device kernel (csrc/registration.hip k_hypotheses) uses the exact Rz,
a difference below 4e-6 absolute for every reachable angle (tests/test_standins.py)."""
import torch


def angle_axis_to_rotation_matrix(angle_axis):
    """angle_axis f[N,3] -> rotation matrices f[N,3,3] (any device, any float dtype)."""
    if not torch.is_tensor(angle_axis):
        raise TypeError(f"Input type is not a torch.Tensor. Got {type(angle_axis)}")
    if angle_axis.dim() != 2 or angle_axis.shape[-1] != 3:
        raise ValueError(f"Input size must be a (*, 3) tensor. Got {tuple(angle_axis.shape)}")
    aa = angle_axis
    theta2 = (aa * aa).sum(dim=1, keepdim=True)
    theta = torch.sqrt(theta2)
    w = aa / (theta + 1e-6)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c, s = torch.cos(theta), torch.sin(theta)
    v = 1.0 - c
    full = torch.cat([c + wx * wx * v, wx * wy * v - wz * s, wy * s + wx * wz * v,
                      wz * s + wx * wy * v, c + wy * wy * v, -wx * s + wy * wz * v,
                      -wy * s + wx * wz * v, wx * s + wy * wz * v, c + wz * wz * v], dim=1).view(-1, 3, 3)
    rx, ry, rz = aa[:, 0:1], aa[:, 1:2], aa[:, 2:3]
    one = torch.ones_like(rx)
    small = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1).view(-1, 3, 3)
    use_full = (theta2 > 1e-6).view(-1, 1, 1)
    return torch.where(use_full, full, small)
