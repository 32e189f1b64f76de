"""ctypes binding of the hugs C ABI (include/hugs.h).  Fails loudly if the HIP library is missing."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('HUGS_LIB_PATH', os.path.join(_HERE, 'csrc', 'libhugs_hip.so'))   # env override: A/B builds

# i = int, f = float, p = device/host pointer, q = long long, s = stream
_PROTOS = {
    'hugs_level_sample_fwd': 'ippiifffffppiiiippppppps',
    'hugs_test_explog': 'pipps',
    'hugs_test_arith': 'ppips',
    'hugs_cast_ipe_fwd': 'iipppppiiiiiips',
    'hugs_dir_enc_fwd': 'iipps',
    'hugs_gemm_nt': 'iiiii' 'pipipi' 'pp' 'iii' 'pi' 'pp' 'pi' 's',
    'hugs_gemm_nt_bits': 'iiiii' 'pipipi' 'p' 'i' 'pp' 'pi' 'pp' 's',
    'hugs_gemm_nt_chain': 'iiiipps',
    'hugs_gemm_tn': 'iiiiipipippps',
    'hugs_gemm_tn_batch': 'iipips',
    'hugs_gemm_tn_batch_nsplit': 'ip',
    'hugs_density_fwd': 'iiipippfpps',
    'hugs_mlp256_tail_fwd': 'iiipppppppfpps',
    'hugs_mlp256_tail_bwd': 'iiippppps',
    'hugs_density_bwd': 'iiipippfpppps',
    'hugs_rank1_mask': 'iiipppipis',
    'hugs_glo_gather': 'iippips',
    'hugs_raybias_fwd': 'iiiippppps',
    'hugs_raybias_bwd': 'iiiiii' 'p' 'i' 'ppppppp' 's',
    'hugs_rgb_fwd': 'iiipippfps',
    'hugs_rgb_bwd': 'iiipipppfpippps',
    'hugs_rgb_bwd_reduce': 'iippps',
    'hugs_composite_fwd': 'iippppifpppps',
    'hugs_composite_bwd': 'iippppifpppps',
    'hugs_composite_bwd_raw': 'iippppifpppppfps',
    'hugs_data_loss': 'iipppififppps',
    'hugs_robust_mask': 'iipppfififpppps',
    'hugs_nf_robust_mask': 'iipppfififpppps',
    'hugs_nf_prop_fwd': 'qiii' 'pipi' 'ppi' 'pppp' 'if' 's',
    'hugs_nf_prop_bwd': 'qiii' 'pipi' 'ppi' 'ppp' 'p' 'pppp' 'p' 'i' 'if' 's',
    'hugs_interlevel': 'iiippppfpps',
    'hugs_distortion': 'iippfpps',
    'hugs_sum': 'ipfps',
    'hugs_add_inplace': 'qpps',
    'hugs_axpy': 'qfpps',
    'hugs_noise_softplus': 'qqppffps',
    'hugs_axpy_op': 'iqfpps',
    'hugs_add_op': 'iqpps',
    'hugs_affine': 'qpffps',
    'hugs_bg_blend_fwd': 'iipppps',
    'hugs_bg_blend_bwd': 'iippppps',
    'hugs_opt_stats': 'iiippppfffppps',
    'hugs_opt_adam': 'iippppppppffffffffpps',
    'hugs_opt_adam_dyn': 'iippppppppffpfffpps',
    'hugs_set_floats': 'piffffs',
    'hugs_stage_step': 'ippppiffffs',
    'hugs_stage_step_pub': 'ippppipppps',
    'hugs_opt_adam_pub': 'iippppppppffpfffppps',
    'hugs_level_sample_fwd_dyn': 'ippiifffpfppiiiipppps',
    'hugs_cast_weights': 'iiippps',
    'hugs_cast_weights_batch': 'iipis',
    'hugs_pixels_to_rays': 'ipppipppipippppppps',
    'hugs_gather_pixels': 'iipppppiipps',
    'hugs_expand_patches': 'iiipppppps',
    'hugs_prng_bits': 'pqps',
    'hugs_prng_uniform': 'pqffps',
    'hugs_prng_normal': 'pqps',
    'hugs_prng_fold_in': 'pips',
    'hugs_prng_step_jitter': 'pipppps',
    'hugs_ssim': 'iiippffffpps',
    'hugs_mse': 'qpppps',
    'hugs_mask_input_fwd': 'iiippiips',
    'hugs_mask_head_fwd': 'iiipippps',
    'hugs_mask_head_bwd': 'iiiipippppps',
    'hugs_embed_scatter_add': 'iiipiipps',
    'hugs_hanerf_loss': 'iipppifpfppps',
    'hugs_hanerf_loss_dyn': 'iipppifppppps',
    'hugs_dual_composite_fwd': 'iipppppppiffpppps',
    'hugs_dual_composite_bwd': 'iipppppppifppfppppps',
    'hugs_rank1_add2_mask': 'iiipppppipis',
    'hugs_nerfw_loss': 'iipppifpfppps',
    'hugs_hashgrid_fwd': 'iiippppp' 'iips',
    'hugs_hashgrid_bwd': 'iiippppp' 'iips',
    'hugs_hashgrid_bwd_ws': 'iiippppp' 'iip' 'pq' 's',
    'hugs_hashgrid_fwd_t': 'iiippppp' 'iiips',
    'hugs_sh4_fwd': 'ipiiips',
    'hugs_hashgrid2d_fwd': 'iiippppp' 'piiips',
    'hugs_hashgrid2d_bwd': 'iiippppp' 'iips',
    'hugs_nf_sample': 'iiippffppiffipppps',
    'hugs_nf_positions': 'iipppifpps',
    'hugs_nf_weights_fwd': 'iipppipppppps',
    'hugs_nf_weights_bwd': 'iipppippppppps',
    'hugs_nf_interlevel': 'iiippppfpps',
    'hugs_nf_density_act': 'qipiippifs',
    'hugs_nf_base_grad': 'qipipppiiipiifs',
    'hugs_nf_head_input': 'qiippiipipis',
    'hugs_nf_field_fwd': 'iqipipippppppppppippppppppppifs',
    'hugs_nf_head_template': 'iippiips',
    'hugs_nf_field_bwd': 'iqippppppppppiipppppipiifs',
    'hugs_nf_app_bwd': 'iiipiiipps',
    'hugs_nf_rgb_act': 'qipifps',
    'hugs_nf_rgb_grad': 'qipppis',
    'hugs_nf_adam': 'qppppffffffs',
    'hugs_amp_check': 'qpps',
    'hugs_amp_prepare': 'ipffps',
    'hugs_nf_adam_amp': 'qppppffffpps',
    'hugs_amp_update': 'ppifffs',
    'hugs_debug_set_nt_cycles': 'p',
    'hugs_gemm_nt_queue_begin': 'pqs',
    'hugs_gemm_nt_queue_end': 'i',
    'hugs_gemm_nt_tiles': 'i' 'iiiii' 'pipipi' 'pp' 'iii' 'pi' 'pp' 'pi' 's',
    'hugs_gemm_tn_tiles': 'i' 'iiiiipipippps',
}
_CT = {'i': ctypes.c_int, 'f': ctypes.c_float, 'p': ctypes.c_void_p, 'q': ctypes.c_longlong,
       's': ctypes.c_void_p}


class HugsError(RuntimeError):
  pass


_GET_RAW = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _raw_stream():
  """hipStream_t of torch's current stream on the current device (the fast path avoids ~8 us of Python per launch)."""
  if _GET_RAW is not None:
    return _GET_RAW(torch.cuda.current_device())
  return torch.cuda.current_stream().cuda_stream


# In-step kernel timing (bench.py's roofline leg): when PROFILE is a list, every GEMM launch is bracketed by two
# HIP events recorded on the stream the kernel is launched on (torch's current stream at call time -- the side
# stream for the weight-gradient GEMMs) and (name, shape key, ev0, ev1) is appended.
PROFILE = None
_PROFILED = {'hugs_gemm_nt': lambda a: ('nt', a[1], a[2], a[3] + a[4], 'mask' if a[16] is not None else ('relu' if a[15] else 'plain')),
             'hugs_gemm_nt_bits': lambda a: ('nt', a[1], a[2], a[3] + a[4], 'mask' if a[18] is not None else ('relu' if a[12] else 'plain')),
             # all trunk layers of an MLP in ONE launch: (kind, rows, width, layers, 'relu')
             'hugs_gemm_nt_chain': lambda a: ('ntc', a[1], a[2], a[3], 'relu'),
             'hugs_gemm_tn': lambda a: ('tn', a[1], a[2], a[3], f'split{a[4]}'),
             # nerfacto (bench.py --config cfg5): (kind, samples, levels, features) / (kind, samples, in_dim, hidden)
             'hugs_hashgrid_fwd': lambda a: ('hg_fwd', a[0], a[1], a[2]),
             'hugs_hashgrid_fwd_t': lambda a: ('hg_fwd', a[0], a[1], a[2]),
             'hugs_hashgrid_bwd': lambda a: ('hg_bwd', a[0], a[1], a[2]),
             'hugs_hashgrid_bwd_ws': lambda a: ('hg_bwd', a[0], a[1], a[2]),
             'hugs_nf_prop_fwd': lambda a: ('prop_fwd', a[0], a[1], a[2]),
             'hugs_nf_prop_bwd': lambda a: ('prop_bwd', a[0], a[1], a[2]),
             # fused field networks: (kind, samples, geo features, appearance columns)
             'hugs_nf_field_fwd': lambda a: ('field_fwd', a[1], a[17], 128 - 16 - a[17]),
             'hugs_nf_field_bwd': lambda a: ('field_bwd', a[1], a[13], a[14])}


class _Lib:

  def __init__(self):
    if not os.path.exists(LIB_PATH):
      raise HugsError(
          f'{LIB_PATH} not found: build it with nerf-hugs_amd/csrc/build.sh (or __graft_entry__.build()). '
          'There is no CPU fallback.')
    self.cdll = ctypes.CDLL(LIB_PATH)
    self.cdll.hugs_last_error.restype = ctypes.c_char_p
    for n_ in ('hugs_raybias_bwd_ws_rows', 'hugs_gemm_tn_batch_ws_bytes', 'hugs_gemm_tn_ws_bytes', 'hugs_density_bwd_ws_bytes', 'hugs_rgb_bwd_ws_bytes', 'hugs_ssim_ws_bytes', 'hugs_gemm_nt_bits_bytes', 'hugs_nf_prop_ws_bytes', 'hugs_hashgrid_bwd_ws_bytes'):
      getattr(self.cdll, n_).restype = ctypes.c_longlong
    self.cdll.hugs_gemm_tn_batch_ws_bytes.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    for name, sig in _PROTOS.items():
      fn = getattr(self.cdll, name)
      fn.argtypes = [_CT[c] for c in sig]
      fn.restype = ctypes.c_int

  def declare(self, name, sig):
    _PROTOS[name] = sig
    fn = getattr(self.cdll, name)
    fn.argtypes = [_CT[c] for c in sig]
    fn.restype = ctypes.c_int

  def call(self, name, *args):
    sig = _PROTOS[name]
    if sig and sig[-1] != 's':
      return getattr(self.cdll, name)(*[int(a) for a in args])
    if len(args) != len(sig) - 1:
      raise TypeError(f'{name}: expected {len(sig) - 1} args (+stream), got {len(args)}')
    conv = []
    dev = -1
    for a, c in zip(args, sig):
      if c == 'p':
        if a is None:
          conv.append(None)
        elif isinstance(a, torch.Tensor):
          if not a.is_cuda:
            raise HugsError(f'{name}: tensor argument is not on the GPU (no CPU fallback)')
          if not a.is_contiguous():
            raise HugsError(f'{name}: tensor argument must be contiguous')
          if a.dtype == torch.float64:
            raise HugsError(f'{name}: float64 tensor passed to a float32 / bf16 kernel')
          if dev < 0:
            dev = a.get_device()
          elif a.get_device() != dev:
            raise HugsError(f'{name}: tensor arguments live on different GPUs')
          conv.append(a.data_ptr())
        else:
          conv.append(int(a))
      elif c == 'f':
        conv.append(float(a))
      else:
        conv.append(int(a))
    if dev >= 0 and dev != torch.cuda.current_device():
      # the kernels launch on torch's current stream of the CURRENT device: pointers of another GPU would be
      # dereferenced there.  Engine wraps its launches in torch.cuda.device(self.device).
      raise HugsError(f'{name}: tensors are on cuda:{dev} but the current device is cuda:{torch.cuda.current_device()} '
                      '(use `with torch.cuda.device(...)`)')
    conv.append(_raw_stream())
    if PROFILE is not None and name in _PROFILED:
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      rc = getattr(self.cdll, name)(*conv)
      e1.record()
      PROFILE.append((name, _PROFILED[name](args), e0, e1))
    else:
      rc = getattr(self.cdll, name)(*conv)
    if rc != 0:
      msg = self.cdll.hugs_last_error().decode()
      if rc == -2:
        raise ValueError(msg)       # the reference raises ValueError for these argument errors
      err = HugsError(f'{name} failed (rc={rc}): {msg}')
      err.rc = rc
      raise err


_LIB = None


def lib():
  global _LIB
  if _LIB is None:
    _LIB = _Lib()
  return _LIB


def call(name, *args):
  return lib().call(name, *args)
