"""Host-side scalar schedules (reference MipNeRF360/internal/math.py:57-98) and PSNR."""
import math as _m


def log_lerp(t, v0, v1):
  if v0 <= 0 or v1 <= 0:
    raise ValueError(f'Interpolants {v0} and {v1} must be positive.')
  lv0, lv1 = _m.log(v0), _m.log(v1)
  return _m.exp(min(max(t, 0), 1) * (lv1 - lv0) + lv0)


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1):
  """Continuous learning rate decay (math.py:66-98)."""
  if lr_delay_steps > 0:
    delay_rate = lr_delay_mult + (1 - lr_delay_mult) * _m.sin(
        0.5 * _m.pi * min(max(step / lr_delay_steps, 0), 1))
  else:
    delay_rate = 1.
  return delay_rate * log_lerp(step / max_steps, lr_init, lr_final)
