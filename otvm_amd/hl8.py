"""Host-side view of the pre-split activation format OTVM_FMT_HL8 (include/otvm_hip.h, csrc/common.h).

Every 8 consecutive elements of a buffer (32 bytes, the footprint of 8 fp32 values) hold 8 fp16 "hi" halves then 8 fp16
"lo" halves with x ~= hi + lo, hi = fp16(x) rounded toward zero, lo = fp16(x - hi).  The kernels produce and consume the
format on the device; these two functions exist for tests, debugging and the occasional host-prepared input.
Buffers stay torch.float32 tensors of the same length (4 bytes per element either way)."""
import torch


def decode(buf_f32):
    """Flat float32-typed buffer holding HL8 data (numel % 8 == 0) -> flat float32 tensor of the values."""
    h = buf_f32.view(torch.float16).reshape(-1, 2, 8)
    return (h[:, 0].float() + h[:, 1].float()).reshape(-1)


def encode(values_f32):
    """Flat float32 values (numel % 8 == 0) -> float32-typed buffer holding their HL8 encoding (same device)."""
    x = values_f32.float().reshape(-1, 8)
    # hi: fp16 toward zero.  In fp16's normal range that is the fp32 value with its low 13 mantissa bits cleared (then
    # the cast is exact); below 2^-14 the fp16 grid is the multiples of 2^-24
    bits = x.view(torch.int32)
    trunc = (bits & ~0x1FFF).view(torch.float32)
    sub = torch.trunc(x * 16777216.0) / 16777216.0
    hi = torch.where(x.abs() < 6.103515625e-05, sub, trunc).half()
    lo = (x - hi.float()).half()
    return torch.stack([hi, lo], dim=1).reshape(-1).view(torch.float32)
