"""ctypes binding of libmixstage_hip.so (the C-ABI declared in include/mixstage.h).

There is NO CPU fallback: if the shared library is missing or does not load, importing the ops
raises -- the product path must fail loudly (build it with `python -c "import __graft_entry__ as g;
g.build()"` or `make -C mix_stage_amd/csrc`).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmixstage_hip.so')
# A/B measurements only (tools/ab_libs.py): another build of the same library; debug setters it lacks are skipped
_ALT_LIB = os.environ.get('MS_LIB_PATH')
if _ALT_LIB:
  LIB_PATH = _ALT_LIB

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t


class BwdOptions(ctypes.Structure):
  """struct ms_bwd_options"""
  _fields_ = [('side_stream', ctypes.c_void_p), ('side_workspace', ctypes.c_void_p), ('side_workspace_bytes', ctypes.c_size_t),
              ('wt_prepared', ctypes.c_void_p), ('wgrad_partials', ctypes.c_void_p), ('defer_wgrad_launch', ctypes.c_int),
              ('prev_y', ctypes.c_void_p), ('prev_y_raw', ctypes.c_void_p), ('prev_save', ctypes.c_void_p), ('prev_gamma', ctypes.c_void_p),
              ('prev_dgamma', ctypes.c_void_p), ('prev_dbeta', ctypes.c_void_p), ('prev_dbias', ctypes.c_void_p), ('prev_slope', ctypes.c_float),
              ('bn_sync', ctypes.c_void_p), ('bn_sync_words', ctypes.c_int32), ('dy_is_dyr', ctypes.c_int),
              ('dx_accum', ctypes.c_void_p)]


class LossScale(ctypes.Structure):
  """struct ms_loss_scale"""
  _fields_ = [('scale', ctypes.c_float), ('scale_dev', ctypes.c_void_p)]


class FwdOptions(ctypes.Structure):
  """struct ms_fwd_options"""
  _fields_ = [('w_planes', ctypes.c_void_p), ('bn_sync', ctypes.c_void_p), ('bn_sync_words', ctypes.c_int32)]


class ConvDesc(ctypes.Structure):
  """struct ms_conv_desc"""
  _fields_ = [(n, ctypes.c_int32) for n in
              ('B', 'Cin', 'H', 'W', 'Cout', 'groups', 'KH', 'KW', 'SH', 'SW', 'PH', 'PW', 'OH', 'OW',
               'mode', 'in_mode')] + \
             [('slope', c_float), ('eps', c_float), ('momentum', c_float), ('dtype', ctypes.c_int32)]


class ChainDesc(ctypes.Structure):
  """struct ms_chain_desc"""
  _fields_ = [(n, ctypes.c_int32) for n in ('B', 'M', 'T', 'cin0', 'C', 'P', 'n_blocks', 'mode', 'dtype', 'sync_first_word', 'keep_all_raw')] + \
             [('slope', c_float), ('eps', c_float), ('momentum', c_float)]


class ChainTensors(ctypes.Structure):
  """struct ms_chain_tensors"""
  _fields_ = [('x', ctypes.c_void_p), ('score', ctypes.c_void_p)] + \
             [(n, ctypes.c_void_p * 4) for n in ('w', 'bias', 'gamma', 'beta', 'running_mean', 'running_var', 'y_raw', 'y', 'save')] + \
             [(n, ctypes.c_void_p) for n in ('w_logits', 'bias_logits', 'z', 'soft', 'out', 'prepared', 'sync')] + \
             [('sync_words', ctypes.c_int32)]


class Prep16Item(ctypes.Structure):
  """struct ms_prep16_item"""
  _fields_ = [('desc', ctypes.POINTER(ConvDesc))] + [(n, ctypes.c_void_p) for n in
             ('w', 'bias', 'gamma', 'beta', 'running_mean', 'running_var', 'fwd', 'dgrad')]


MS_BARE, MS_LRELU, MS_BN_TRAIN, MS_BN_EVAL = 0, 1, 2, 3
MS_IN_PLAIN, MS_IN_BCAST, MS_IN_UP2ADD = 0, 1, 2
MS_F32, MS_BF16, MS_F16, MS_DT_OUT_F32, MS_DT_BN_FOLDED = 0, 1, 2, 0x100, 0x200
MS_DT_STAT_PAIR = 0x400      # BN_TRAIN statistics per half of the batch (two passes of a module side by side: include/mixstage.h)

_P = c_void_p
_DESC = ctypes.POINTER(ConvDesc)

# name -> (restype, argtypes): every symbol include/mixstage.h declares
SIGNATURES = {
    'ms_last_error': (ctypes.c_char_p, []),
    'ms_abi_version': (c_int, []),
    'ms_set_counter_buffer': (c_int, [_P, c_int]),
    'ms_set_bn_sync_buffer': (c_int, [_P, c_int]),
    'ms_debug_set_bn_fused': (c_int, [c_int]),
    'ms_debug_set_bn_fused_min_workgroups': (c_int, [c_int]),
    'ms_conv_block_fwd_workspace': (c_size_t, [_DESC]),
    'ms_conv_block_bwd_workspace': (c_size_t, [_DESC]),
    'ms_conv_block_fwd': (c_int, [_DESC] + [_P] * 11 + [_P, c_size_t, _P]),
    'ms_conv_block_bwd': (c_int, [_DESC] + [_P] * 17 + [_P, c_size_t, _P]),
    'ms_conv_block_bwd_overlap': (c_int, [_DESC] + [_P] * 17 + [_P, c_size_t, _P, _P, _P, c_size_t]),
    'ms_conv_block_bwd_ex': (c_int, [_DESC] + [_P] * 17 + [_P, c_size_t, _P, _P]),
    'ms_dgrad_weights_elems': (c_size_t, [_DESC, _P]),
    'ms_dgrad_weights_prepare': (c_int, [c_int, _P, _P, _P, _P]),
    'ms_tuning_epoch': (c_int, []),
    'ms_conv_block_fwd_ex': (c_int, [_DESC] + [_P] * 11 + [_P, c_size_t, _P, _P]),
    'ms_fwd_weights_bytes': (c_size_t, [_DESC]),
    'ms_fwd_weights_prepare': (c_int, [c_int, _P, _P, _P, _P]),
    'ms_set_precision': (c_int, [c_int]),
    'ms_get_precision': (c_int, []),
    'ms_wgrad_partials_elems': (c_size_t, [_DESC, _P]),
    'ms_wgrad_reduce_multi': (c_int, [c_int, _P, _P, _P, _P, _P]),
    'ms_wgrad_flush': (c_int, [_P]),
    'ms_wgrad_discard': (c_int, []),
    'ms_weights16_bytes': (c_size_t, [_DESC, c_int]),
    'ms_weights16_prepare': (c_int, [c_int, _P, _P]),
    'ms_cb8_from_plain': (c_int, [c_int, _P, _P, c_int, c_int, c_int, _P]),
    'ms_cb8_to_plain': (c_int, [c_int, _P, _P, c_int, c_int, c_int, _P]),
    'ms_cb8_from_btc': (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'ms_cb8_to_btc': (c_int, [c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'ms_bn_stats': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'ms_bn_train_apply': (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_float, _P]),
    'ms_bn_bwd_workspace': (c_size_t, [c_int, c_int]),
    'ms_bn_bwd_sums': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, c_size_t, _P]),
    'ms_bn_bwd_apply': (c_int, [_P, _P, _P, _P, _P, ctypes.c_double, _P, c_int, c_int, c_int, c_float, _P]),
    'ms_lerp_time_fwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'ms_lerp_time_bwd': (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'ms_softmax_mix_fwd': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'ms_softmax_mix_bwd': (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    'ms_kmeans_labels': (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'ms_znorm_select': (c_int, [_P, _P, _P, _P, _P, c_size_t, c_int, c_int, _P]),
    'ms_step_metrics': (c_int, [_P] * 7 + [c_int, _P, c_int, c_int, c_int, c_int, _P]),
    'ms_eval_accumulate': (c_int, [_P] * 8 + [c_int, c_int, c_int, c_int, ctypes.c_double, c_int, _P]),
    'ms_concat_style_fwd': (c_int, [_P, _P, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    'ms_concat_style_bwd': (c_int, [_P, _P, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    'ms_cross_entropy_fwd': (c_int, [_P, _P, _P, _P] + [c_int] * 6 + [_P]),
    'ms_cross_entropy_bwd': (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [_P]),
    'ms_cross_entropy_fwd_ex': (c_int, [_P, _P, _P] + [c_int] * 6 + [_P, _P]),
    'ms_cross_entropy_bwd_ex': (c_int, [_P, _P, _P, _P] + [c_int] * 7 + [_P, _P]),
    'ms_lp_mean_fwd_ex': (c_int, [c_int, _P, _P, c_float, _P, _P, c_size_t, _P, _P]),
    'ms_lp_mean_bwd_ex': (c_int, [c_int, _P, _P, c_float, _P, _P, c_size_t, _P, _P]),
    'ms_lp_mean_pair_fwd': (c_int, [c_int, _P, _P, _P, c_size_t, _P, _P]),
    'ms_lp_mean_pair_bwd': (c_int, [c_int, _P, _P, _P, _P, _P, c_size_t, _P, _P]),
    'ms_velocity_fwd': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'ms_velocity_bwd': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'ms_transpose_btc': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'ms_transpose_bct': (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    'ms_l1_mean_fwd': (c_int, [_P, _P, c_float, _P, _P, c_size_t, _P]),
    'ms_l1_mean_bwd': (c_int, [_P, _P, c_float, _P, _P, c_size_t, _P]),
    'ms_l2_mean_fwd': (c_int, [_P, _P, c_float, _P, _P, c_size_t, _P]),
    'ms_l2_mean_bwd': (c_int, [_P, _P, c_float, _P, _P, c_size_t, _P]),
    'ms_copy_multi': (c_int, [c_int, _P, _P, _P, _P]),
    'ms_write_floats': (c_int, [_P, _P, c_int, _P]),
    'ms_decoder_chain_supported': (c_int, [_P]),
    'ms_decoder_chain_prepared_bytes': (c_size_t, [_P]),
    'ms_decoder_chain_workspace': (c_size_t, [_P]),
    'ms_decoder_chain_sync_words': (c_int, [_P]),
    'ms_decoder_chain_prepare': (c_int, [_P, _P, _P, _P, _P]),
    'ms_decoder_chain_fwd': (c_int, [_P, _P, _P, c_size_t, _P]),
    'ms_sqnorm': (c_int, [_P, c_size_t, _P, _P, _P]),
    'ms_adam_step': (c_int, [_P, _P, _P, _P, c_size_t, _P, c_float, c_float, c_float, c_float, c_float, _P, _P]),
    'ms_adam_step_segmented': (c_int, [_P, _P, _P, _P, c_size_t, _P, c_float, c_float, c_float, c_float, c_float, _P, _P, _P, _P,
                                       c_int, _P]),
    'ms_reduce_partials_count': (c_size_t, [c_size_t]),
    'ms_timing_enable': (c_int, [c_int]),
    'ms_timing_report': (c_size_t, [ctypes.c_char_p, c_size_t]),
    'ms_debug_set_patch_min_workgroups': (c_int, [c_int]),
    'ms_debug_set_patch_tuning': (c_int, [c_int, c_int]),
    'ms_debug_set_conv16_tile': (c_int, [c_int, c_int]),
    'ms_debug_set_conv16_ring': (c_int, [c_int, c_int]),
    'ms_debug_set_skip': (c_int, [ctypes.c_char_p]),
    'ms_debug_set_clip32': (c_int, [c_int]),
    'ms_debug_set_wgrad16_target': (c_int, [c_int]),
    'ms_debug_set_wgrad16_ring': (c_int, [c_int]),
    'ms_debug_set_wgrad_target': (c_int, [c_int]),
    'ms_debug_set_wgrad_wave': (c_int, [c_int]),
    'ms_debug_set_conv_tile': (c_int, [c_int]),
    'ms_set_wgrad_batched': (c_int, [c_int, c_int]),
    'ms_dgrad_fuses_prev_bn': (c_int, [_P]),
    'ms_dgrad_takes_accum': (c_int, [_P]),
    'ms_stat_pair_ok': (c_int, [_P]),
    'ms_selftest_mfma': (c_int, [_P, _P, _P, c_int, _P]),
    'ms_probe_peak': (c_int, [c_int, ctypes.c_long, _P, _P, ctypes.POINTER(ctypes.c_double), _P]),
}

_lib = None


class MixStageLibError(RuntimeError):
  pass


def lib():
  global _lib
  if _lib is None:
    if not os.path.isfile(LIB_PATH):
      raise MixStageLibError('%s not built: run `make -C mix_stage_amd/csrc` (or __graft_entry__.build()); '
                             'there is no CPU fallback' % LIB_PATH)
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      if _ALT_LIB and name.startswith('ms_debug_') and not hasattr(handle, name):
        continue
      fn = getattr(handle, name)          # AttributeError here = header/library drift
      fn.restype, fn.argtypes = res, args
    mode = os.environ.get('MS_PRECISION', '')
    if mode:                            # 'fp32' | 'bf16x6' (see ms_set_precision)
      if mode not in ('fp32', 'bf16x6'):
        raise MixStageLibError('MS_PRECISION=%s: expected fp32 or bf16x6' % mode)
      handle.ms_set_precision(1 if mode == 'bf16x6' else 0)
    if os.environ.get('MS_BN_FUSED'):           # ablations only: MS_BN_FUSED=0 keeps BatchNorm in its own launch
      handle.ms_debug_set_bn_fused(int(os.environ['MS_BN_FUSED']))
    if os.environ.get('MS_BN_FUSED_MIN_WGS'):
      handle.ms_debug_set_bn_fused_min_workgroups(int(os.environ['MS_BN_FUSED_MIN_WGS']))
    if os.environ.get('MS_CLIP32'):             # ablations only: MS_CLIP32=0 keeps the per-layer patch / gather kernels
      handle.ms_debug_set_clip32(int(os.environ['MS_CLIP32']))
    if os.environ.get('MS_CONV_TILE'):          # ablations only: MS_CONV_TILE=0 keeps the 2-D fp32 convs on conv_patch_kernel
      handle.ms_debug_set_conv_tile(int(os.environ['MS_CONV_TILE']))
    if os.environ.get('MS_WGRAD_WAVE'):         # ablations only: MS_WGRAD_WAVE=0 keeps the barrier-per-tile fp32 weight gradient
      handle.ms_debug_set_wgrad_wave(int(os.environ['MS_WGRAD_WAVE']))
    if os.environ.get('MS_PATCH_MIN_WGS'):      # tuning experiments only (ms_debug_set_patch_min_workgroups)
      handle.ms_debug_set_patch_min_workgroups(int(os.environ['MS_PATCH_MIN_WGS']))
    _lib = handle
  return _lib


def check(rc, what):
  if rc != 0:
    raise MixStageLibError('%s failed: %s' % (what, lib().ms_last_error().decode()))
