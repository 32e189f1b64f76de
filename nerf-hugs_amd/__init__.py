"""hugs-mi355x: MI355X-native Mip-NeRF 360 per-ray train/render path (drop-in for
cnhaox/NeRF-HuGS MipNeRF360/internal/{render,models,train_utils}.py).

All per-ray arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI declared in
include/hugs.h (csrc/libhugs_hip.so); PyTorch-ROCm only owns device memory, streams and
torch.distributed (RCCL).  There is no CPU fallback: importing `_lib` without the built
library, or calling an op without a GPU, raises."""
__version__ = '0.1.0'
