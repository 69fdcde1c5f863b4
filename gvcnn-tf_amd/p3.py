"""Host-side view of the three-plane storage of fp32 activations ("P3", GV_CONV_X_P3 / GV_CONV_Y_P3).

Every fp32 value a is kept as three bf16 planes a = a0 + a1 + a2 (a0 = bf16(a), a1 = bf16(a - a0),
a2 = bf16(a - a0 - a1): exact for every finite fp32 whose low planes do not underflow), laid out
[pixel][channel/16][plane][16] so that the 16 k-values of one MFMA step of one plane are 32 contiguous bytes — the
layout of the packed filters.  The device writes it in the producer's epilogue and reads it with the LDS-DMA loader;
these torch functions exist for tests, probes and tools."""
import torch


def to_p3(x):
    """fp32 [..., C] (C % 16 == 0) -> int16 [..., C/16, 3, 16] (bit patterns of the bf16 planes)."""
    assert x.dtype == torch.float32 and x.shape[-1] % 16 == 0
    a0 = x.to(torch.bfloat16)
    r1 = x - a0.float()
    a1 = r1.to(torch.bfloat16)
    a2 = (r1 - a1.float()).to(torch.bfloat16)
    g = x.shape[-1] // 16
    planes = torch.stack([p.reshape(x.shape[:-1] + (g, 16)) for p in (a0, a1, a2)], dim=-2)   # [..., g, 3, 16]
    return planes.contiguous().view(torch.int16)


def from_p3(p):
    """int16 [..., C/16, 3, 16] -> fp32 [..., C] (the exact sum of the planes)."""
    v = p.view(torch.bfloat16).float()
    s = (v[..., 2, :] + v[..., 1, :]) + v[..., 0, :]
    return s.reshape(s.shape[:-2] + (s.shape[-2] * 16,))
