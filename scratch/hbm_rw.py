"""scratch: achievable HBM stream rates on this box: pure write (fill), pure read (sum), copy -- the ceilings the write-dominated fused
kernels (k_field_fwd: 97 % of its bytes are stores; k_cast_ipe; the PropMLP chain) should be read against."""
import torch
dev = 'cuda'
n = 1 << 30          # 1 GiB of bf16 = 2 GiB? no: elements
x = torch.empty(n // 2, dtype=torch.float32, device=dev)      # 2 GiB
y = torch.empty_like(x)
def t(fn, k=10):
  for _ in range(3): fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(k): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / k * 1e-3
B = x.numel() * 4
print(f'fill  (write {B/2**30:.1f} GiB): {B / t(lambda: x.fill_(1.5)) / 1e12:.2f} TB/s')
print(f'zero  (write): {B / t(lambda: x.zero_()) / 1e12:.2f} TB/s')
print(f'sum   (read) : {B / t(lambda: x.sum()) / 1e12:.2f} TB/s')
print(f'copy  (read + write, 2x bytes): {2 * B / t(lambda: y.copy_(x)) / 1e12:.2f} TB/s')
xb = x.view(torch.bfloat16)
print(f'bf16 relu out-of-place (r + w): {2 * B / t(lambda: torch.relu(xb)) / 1e12:.2f} TB/s')
