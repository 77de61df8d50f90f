"""ctypes binding of the C-ABI kernel library (include/rumpy_amd.h -> rumpy_amd/librumpy_amd.so).

The library is the product: there is NO fallback.  ``lib()`` raises ``RuntimeError`` when the shared object
is missing or an entry point fails, so a GPU box without the HIP extension fails loudly instead of silently
running something else.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RUMPY_AMD_LIB: load another build of the same C ABI (kernel experiments, tests/tools/); the default is the in-tree library
LIB_PATH = os.environ.get('RUMPY_AMD_LIB') or os.path.join(_HERE, 'librumpy_amd.so')

c_void_p, c_int32, c_int64, c_float = C.c_void_p, C.c_int32, C.c_int64, C.c_float

TILE_H, TILE_W = 8, 16
FMT_BF16, FMT_F16, FMT_F16_RESIDUAL, FMT_F32 = 0, 1, 2, 3          # include/rumpy_amd.h RUMPY_FMT_*


class _S(C.Structure):
    def __init__(self, **kw):
        super().__init__()
        names = {f[0] for f in self._fields_}
        for k, v in kw.items():
            if k not in names:
                raise AttributeError('%s has no field %s' % (type(self).__name__, k))
            setattr(self, k, v)


class ConvArgs(_S):
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('bias', c_void_p), ('out', c_void_p), ('mask', c_void_p),
                ('res1', c_void_p), ('res2', c_void_p), ('pool', c_void_p),
                ('N', c_int32), ('H', c_int32), ('W', c_int32), ('cin_chunks', c_int32), ('cout_tiles', c_int32),
                ('in_mode', c_int32), ('out_mode', c_int32), ('relu', c_int32), ('scale', c_float),
                ('grid_x', c_int32), ('fmt', c_int32), ('w_lo', c_void_p)]


class HeadFwdArgs(_S):
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('b', c_void_p), ('out', c_void_p),
                ('N', c_int32), ('C', c_int32), ('H', c_int32), ('W', c_int32), ('cout', c_int32), ('neg_slope_m1', c_float),
                ('fmt', c_int32), ('pad_', c_int32), ('x_ind', c_void_p)]


class EncConvArgs(_S):
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('bias', c_void_p), ('out', c_void_p),
                ('N', c_int32), ('H', c_int32), ('W', c_int32), ('cin', c_int32), ('cout', c_int32), ('stride', c_int32),
                ('neg_slope', c_float), ('fmt', c_int32), ('w_lo', c_void_p), ('out_fmt', c_int32), ('pad_', c_int32)]


class RcabArgs(_S):
    _fields_ = [('x', c_void_p), ('w1', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p),
                ('t', c_void_p), ('t2', c_void_p), ('t2_in', c_void_p), ('mask', c_void_p), ('res2', c_void_p), ('out', c_void_p),
                ('N', c_int32), ('H', c_int32), ('W', c_int32), ('cr', c_int32),
                ('ca_w1', c_void_p), ('ca_b1', c_void_p), ('ca_w2', c_void_p), ('ca_b2', c_void_p),
                ('mean', c_void_p), ('hidden', c_void_p), ('gate', c_void_p), ('qgate', c_void_p), ('dz', c_void_p), ('dzq', c_void_p),
                ('xchg', c_void_p), ('xchg_bytes', c_int64), ('epoch', c_void_p), ('status', c_void_p),
                ('seq', C.c_uint32), ('fmt', c_int32), ('maskbits', c_void_p),
                ('w1_f8', c_void_p), ('w2_f8', c_void_p), ('f8_sw1', c_void_p), ('f8_sw2', c_void_p), ('f8_site', c_void_p), ('f8_entries', c_int32)]


class Rcab2Args(_S):
    _fields_ = [('x', c_void_p), ('u_in', c_void_p), ('part_in', c_void_p), ('part_out', c_void_p), ('part_scratch', c_void_p),
                ('w1', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p),
                ('x_out', c_void_p), ('t', c_void_p), ('u_out', c_void_p), ('res2', c_void_p), ('maskbits', c_void_p),
                ('ca_w1', c_void_p), ('ca_b1', c_void_p), ('ca_w2', c_void_p), ('ca_b2', c_void_p),
                ('mean', c_void_p), ('hidden', c_void_p), ('gate', c_void_p), ('qgate', c_void_p), ('dz', c_void_p), ('dzq', c_void_p),
                ('N', c_int32), ('H', c_int32), ('W', c_int32), ('cr', c_int32), ('np_in', c_int32), ('fmt', c_int32)]


class ResChainBlock(_S):
    _fields_ = [('x', c_void_p), ('w1', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p), ('res2', c_void_p), ('t', c_void_p), ('out', c_void_p),
                ('maskbits', c_void_p), ('scale1', c_float), ('scale2', c_float)]


class ResChainArgs(_S):
    _fields_ = [('blocks', c_void_p), ('nblocks', c_int32), ('N', c_int32), ('H', c_int32), ('W', c_int32), ('backward', c_int32), ('fmt', c_int32),
                ('work', c_void_p), ('work_bytes', c_int64), ('status', c_void_p), ('fake_xcc', c_int32), ('force_sc1', c_int32),
                ('edge_w', c_void_p), ('edge_b', c_void_p), ('edge_x', c_void_p), ('edge_res', c_void_p), ('edge_out', c_void_p)]


class Op(_S):
    _fields_ = [('fn', c_void_p), ('args', c_void_p)]


class EncBnArgs(_S):
    _fields_ = [('x', c_void_p), ('gamma', c_void_p), ('beta', c_void_p), ('running_mean', c_void_p), ('running_var', c_void_p),
                ('num_batches_tracked', c_void_p), ('partial', c_void_p), ('scale_shift', c_void_p),
                ('P', c_int32), ('C', c_int32), ('eps', c_float), ('momentum', c_float), ('neg_slope', c_float), ('fmt', c_int32),
                ('x_fmt', c_int32)]


class EncBnBwdArgs(_S):
    _fields_ = [('z', c_void_p), ('da', c_void_p), ('dpool', c_void_p), ('scale_shift', c_void_p), ('saved', c_void_p), ('gamma', c_void_p),
                ('dgamma', c_void_p), ('dbeta', c_void_p), ('dz', c_void_p), ('partial', c_void_p), ('coef', c_void_p),
                ('N', c_int32), ('Ho', c_int32), ('Wo', c_int32), ('C', c_int32), ('up', c_int32), ('Hz', c_int32), ('Wz', c_int32),
                ('neg_slope', c_float), ('scale', c_float), ('fmt', c_int32)]


class HeadWgradArgs(_S):
    _fields_ = [('x', c_void_p), ('dy', c_void_p), ('slab', c_void_p), ('gw', c_void_p), ('gb', c_void_p),
                ('N', c_int32), ('C', c_int32), ('H', c_int32), ('W', c_int32), ('cout', c_int32),
                ('scale', c_float), ('x_ind', c_void_p)]


class TailFwdArgs(_S):
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('bias', c_void_p), ('out', c_void_p), ('target', c_void_p),
                ('dy4', c_void_p), ('loss_partial', c_void_p), ('loss', c_void_p),
                ('N', c_int32), ('C', c_int32), ('H', c_int32), ('W', c_int32), ('grid_x', c_int32), ('wslab', c_void_p),
                ('nonfinite', c_void_p), ('fmt', c_int32), ('pad_', c_int32), ('target_ind', c_void_p)]


class TailDgradArgs(_S):
    _fields_ = [('dy4', c_void_p), ('w', c_void_p), ('dx', c_void_p), ('N', c_int32), ('H', c_int32), ('W', c_int32)]


class Conv4dTailArgs(_S):       # = rumpy_conv4d_tail_args
    _fields_ = [('dy4', c_void_p), ('w_tail', c_void_p), ('dx', c_void_p), ('w', c_void_p), ('out', c_void_p),
                ('N', c_int32), ('H', c_int32), ('W', c_int32), ('grid_x', c_int32)]


class NchwToNhwc4Args(_S):
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('N', c_int32), ('C', c_int32), ('H', c_int32), ('W', c_int32)]


class PixelShuffleArgs(_S):
    _fields_ = [('src', c_void_p), ('dst', c_void_p), ('N', c_int32), ('H', c_int32), ('W', c_int32), ('F', c_int32), ('r', c_int32),
                ('inverse', c_int32)]


class TailWideArgs(_S):
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('bias', c_void_p), ('out', c_void_p), ('nonfinite', c_void_p),
                ('N', c_int32), ('H', c_int32), ('W', c_int32), ('F', c_int32), ('C', c_int32), ('fmt', c_int32)]


class WgradJob(_S):
    _fields_ = [('x', c_void_p), ('dy', c_void_p), ('slab', c_void_p), ('n0', c_int32), ('n1', c_int32), ('t0', c_int32), ('t1', c_int32),
                ('H', c_int32), ('W', c_int32), ('x_cstride', c_int32), ('x_coff', c_int32),
                ('dy_mode', c_int32), ('dy_cstride', c_int32), ('dy_coff', c_int32), ('mt', c_int32)]


class ReduceItem(_S):
    _fields_ = [('slab', c_void_p), ('slab_stride', c_int64), ('njobs', c_int32), ('mt', c_int32),
                ('co_count', c_int32), ('co_mode', c_int32), ('co_off', c_int32), ('ci_total', c_int32),
                ('ci_off', c_int32), ('write_bias', c_int32), ('scale', c_float), ('gw', c_void_p), ('gb', c_void_p)]


class PackItem(_S):
    _fields_ = [('w', c_void_p), ('b', c_void_p), ('w_fwd', c_void_p), ('w_dgrad', c_void_p), ('b_packed', c_void_p),
                ('cout', c_int32), ('cin', c_int32), ('kind', c_int32), ('shuffle', c_int32), ('fmt', c_int32), ('pad_', c_int32)]


class FinishReduceArgs(_S):
    _fields_ = [('items', c_void_p), ('nitems', c_int32), ('pad_', c_int32),
                ('tail_slabs', c_void_p), ('tail_nslabs', c_int32), ('tail_C', c_int32), ('tail_scale', c_float), ('pad2_', c_int32),
                ('tail_gw', c_void_p), ('tail_gb', c_void_p),
                ('head_slabs', c_void_p), ('head_nslabs', c_int32), ('head_C', c_int32), ('head_cout', c_int32), ('head_scale', c_float),
                ('head_gw', c_void_p), ('head_gb', c_void_p)]


class UpdateItem(_S):
    _fields_ = [('kind', c_int32), ('woff', c_int32), ('n', c_int32), ('cout', c_int32), ('cin', c_int32), ('shuffle', c_int32), ('ct', c_int32),
                ('ch', c_int32), ('q', c_int32), ('hf', c_int32), ('boff', c_int32), ('pad_', c_int32),
                ('w_fwd', c_void_p), ('w_dgrad', c_void_p), ('b_packed', c_void_p)]


class CaMlpFwdArgs(_S):
    _fields_ = [('pool', c_void_p), ('w1', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p),
                ('mean', c_void_p), ('hidden', c_void_p), ('gate', c_void_p),
                ('N', c_int32), ('C', c_int32), ('Cr', c_int32), ('ntiles', c_int32), ('inv_hw', c_float)]


class CaScaleArgs(_S):
    _fields_ = [('t', c_void_p), ('res', c_void_p), ('gate', c_void_p), ('out', c_void_p),
                ('N', c_int32), ('HW', c_int32), ('C', c_int32)]


class CaBwdReduceArgs(_S):
    _fields_ = [('dy', c_void_p), ('t', c_void_p), ('partial', c_void_p), ('N', c_int32), ('HW', c_int32), ('C', c_int32)]


class CaMlpBwdArgs(_S):
    _fields_ = [('partial', c_void_p), ('mean', c_void_p), ('hidden', c_void_p), ('gate', c_void_p), ('w1', c_void_p),
                ('w2', c_void_p), ('dpool', c_void_p), ('gw1', c_void_p), ('gb1', c_void_p), ('gw2', c_void_p),
                ('gb2', c_void_p), ('N', c_int32), ('C', c_int32), ('Cr', c_int32), ('nchunks', c_int32),
                ('inv_hw', c_float), ('scale', c_float)]


class CaBwdApplyArgs(_S):
    _fields_ = [('dy', c_void_p), ('gate', c_void_p), ('dpool', c_void_p), ('dt', c_void_p),
                ('N', c_int32), ('HW', c_int32), ('C', c_int32)]


class QcaLayer(_S):
    _fields_ = [('w', c_void_p), ('b', c_void_p), ('gw', c_void_p), ('gb', c_void_p), ('n_prev', c_int32), ('n_out', c_int32), ('cat', c_int32),
                ('relu_in', c_int32), ('act', c_int32), ('pad_', c_int32)]


class QcaArgs(_S):
    _fields_ = [('layers', QcaLayer * 4), ('nlayers', c_int32), ('N', c_int32), ('C', c_int32), ('M', c_int32), ('ntiles', c_int32),
                ('nchunks', c_int32), ('inv_hw', c_float), ('scale', c_float), ('pool', c_void_p), ('attr', c_void_p), ('acts', c_void_p),
                ('gate', c_void_p), ('partial', c_void_p), ('dpool', c_void_p), ('delta', c_void_p)]


QCA_ACT_STRIDE = 320


class CaFwdFusedArgs(_S):
    _fields_ = [('pool', c_void_p), ('w1', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p), ('mean', c_void_p),
                ('hidden', c_void_p), ('gate', c_void_p), ('t', c_void_p), ('res', c_void_p), ('out', c_void_p),
                ('N', c_int32), ('HW', c_int32), ('C', c_int32), ('Cr', c_int32), ('ntiles', c_int32), ('inv_hw', c_float),
                ('qgate', c_void_p), ('fmt', c_int32), ('pad_', c_int32)]


class CaBwdFusedArgs(_S):
    _fields_ = [('dy', c_void_p), ('partial', c_void_p), ('hidden', c_void_p), ('gate', c_void_p), ('w1', c_void_p), ('w2', c_void_p),
                ('dz', c_void_p), ('dt', c_void_p), ('N', c_int32), ('HW', c_int32), ('C', c_int32), ('Cr', c_int32),
                ('nchunks', c_int32), ('inv_hw', c_float), ('qgate', c_void_p), ('dzq', c_void_p)]


class QMlpItem(_S):
    _fields_ = [('w1', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p), ('hidden', c_void_p), ('gate', c_void_p),
                ('dzq', c_void_p), ('gw1', c_void_p), ('gb1', c_void_p), ('gw2', c_void_p), ('gb2', c_void_p), ('scale', c_float),
                ('pad_', c_int32)]


QN_MAX_LAYERS = 8            # include/rumpy_amd.h RUMPY_QN_MAX_LAYERS


class QMlpNItem(_S):         # = rumpy_q_mlpn_item
    _fields_ = [('w', c_void_p * QN_MAX_LAYERS), ('b', c_void_p * QN_MAX_LAYERS), ('gw', c_void_p * QN_MAX_LAYERS), ('gb', c_void_p * QN_MAX_LAYERS), ('acts', c_void_p),
                ('gate', c_void_p), ('dzq', c_void_p), ('n', c_int32 * (QN_MAX_LAYERS + 1)), ('nlayers', c_int32), ('scale', c_float), ('pad_', c_int32)]


class AdamHyper(_S):
    _fields_ = [('lr', c_float), ('beta1', c_float), ('beta2', c_float), ('eps', c_float), ('bias_c1', c_float),
                ('sqrt_bias_c2', c_float), ('grad_mult', c_float), ('max_norm', c_float)]


class AdamArgs(_S):
    _fields_ = [('p', c_void_p), ('g', c_void_p), ('m', c_void_p), ('v', c_void_p), ('n', c_int64),
                ('hyper', c_void_p), ('sumsq', c_void_p), ('hyper_value', AdamHyper), ('skip_if', c_void_p)]


class AdamPackArgs(_S):
    _fields_ = [('items', c_void_p), ('nitems', c_int32), ('pad_', c_int32), ('p', c_void_p), ('g', c_void_p), ('m', c_void_p), ('v', c_void_p),
                ('hyper', c_void_p), ('sumsq', c_void_p), ('hyper_value', AdamHyper), ('skip_if', c_void_p)]


class SumsqArgs(_S):
    _fields_ = [('g', c_void_p), ('n', c_int64), ('partial', c_void_p), ('out', c_void_p)]


class EvalPostArgs(_S):
    _fields_ = [('out', c_void_p), ('ref', c_void_p), ('rgb', c_void_p), ('ycbcr', c_void_p),
                ('sse_partial', c_void_p), ('sse', c_void_p), ('N', c_int32), ('H', c_int32), ('W', c_int32)]


class BlockArgs(_S):
    _fields_ = [('x', c_void_p), ('w1', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('b2', c_void_p), ('mask', c_void_p),
                ('res2', c_void_p), ('t', c_void_p), ('out', c_void_p), ('N', c_int32), ('H', c_int32), ('W', c_int32),
                ('relu1', c_int32), ('scale1', c_float), ('scale2', c_float), ('res_mode', c_int32), ('res1', c_void_p),
                ('pool', c_void_p), ('maskbits', c_void_p), ('fmt', c_int32), ('col_tile', c_int32),
                ('w1_f8', c_void_p), ('w2_f8', c_void_p), ('f8_sw1', c_void_p), ('f8_sw2', c_void_p), ('f8_site', c_void_p),
                ('f8_entries', c_int32)]


class Fp8PackItem(_S):
    _fields_ = [('w', c_void_p), ('img_fwd', c_void_p), ('img_dgrad', c_void_p), ('exponent', c_void_p)]


FP8_SITE_HEAD = 4
FP8_IMAGE_BYTES = 40960


class SsimArgs(_S):
    _fields_ = [('a', c_void_p), ('b', c_void_p), ('partial', c_void_p), ('out', c_void_p), ('P', c_int32), ('H', c_int32),
                ('W', c_int32), ('data_range', c_float)]


class PatchItem(_S):
    _fields_ = [('lr_off', c_int64), ('hr_off', c_int64), ('lr_h', c_int32), ('lr_w', c_int32), ('hflip', c_int32),
                ('vflip', c_int32), ('rot', c_int32), ('y', c_int32), ('x', c_int32), ('pad_', c_int32)]


class PatchArgs(_S):
    _fields_ = [('images', c_void_p), ('items', c_void_p), ('out_lr', c_void_p), ('out_hr', c_void_p),
                ('N', c_int32), ('C', c_int32), ('crop', c_int32), ('scale', c_int32)]


class DconvArgs(_S):
    _fields_ = [('x', c_void_p), ('w', c_void_p), ('bias', c_void_p), ('mask', c_void_p), ('res', c_void_p), ('y', c_void_p),
                ('N', c_int32), ('Cin', c_int32), ('Cout', c_int32), ('H', c_int32), ('W', c_int32), ('k', c_int32),
                ('relu', c_int32), ('transposed', c_int32)]


class DconvWgradArgs(_S):
    _fields_ = [('x', c_void_p), ('dy', c_void_p), ('partial', c_void_p), ('gw', c_void_p), ('gb', c_void_p),
                ('N', c_int32), ('Cin', c_int32), ('Cout', c_int32), ('H', c_int32), ('W', c_int32), ('k', c_int32),
                ('scale', c_float)]


class MseArgs(_S):
    _fields_ = [('out', c_void_p), ('target', c_void_p), ('grad', c_void_p), ('partial', c_void_p), ('loss', c_void_p), ('n', c_int64)]


class SgemmArgs(_S):
    _fields_ = [('A', c_void_p), ('B', c_void_p), ('C', c_void_p), ('bias', c_void_p), ('partial', c_void_p),
                ('M', c_int32), ('N', c_int32), ('K', c_int32), ('ldc', c_int32), ('sam', c_int64), ('sak', c_int64), ('sbk', c_int64), ('sbn', c_int64),
                ('alpha', c_float), ('leaky_slope', c_float), ('accumulate', c_int32), ('pad_', c_int32)]


# every symbol include/rumpy_amd.h declares: name -> (restype, argtypes)
_P = C.POINTER
SYMBOLS = {
    'rumpy_last_error': (C.c_char_p, []),
    'rumpy_abi_version': (C.c_int, []),
    'rumpy_device_cus': (C.c_int, []),
    'rumpy_conv3x3': (C.c_int, [_P(ConvArgs), c_void_p]),
    'rumpy_conv_pool_tiles': (C.c_int, [c_int32, c_int32, c_int32]),
    'rumpy_debug_conv_stamps': (C.c_int, [_P(ConvArgs), c_void_p]),
    'rumpy_head_fwd': (C.c_int, [_P(HeadFwdArgs), c_void_p]),
    'rumpy_head_wgrad': (C.c_int, [_P(HeadWgradArgs), c_void_p]),
    'rumpy_rcab_fwd': (C.c_int, [_P(RcabArgs), c_void_p]),
    'rumpy_rcab_bwd': (C.c_int, [_P(RcabArgs), c_void_p]),
    'rumpy_rcab_xchg_bytes': (c_int64, [c_int32, c_int32, c_int32]),
    'rumpy_rcab_strips': (C.c_int, [c_int32, c_int32]),
    'rumpy_block_pool_tiles': (C.c_int, [c_int32, c_int32]),
    'rumpy_fp8_pack': (C.c_int, [c_void_p, c_int32, c_void_p]),
    'rumpy_fp8_rotate': (C.c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    'rumpy_fp8_site_entries': (C.c_int, [c_int32, c_int32, c_int32]),
    'rumpy_fp8_convert': (C.c_int, [c_void_p, C.c_float, c_void_p, c_int32, c_int32, c_void_p]),
    'rumpy_rcab_epoch_advance': (C.c_int, [c_void_p, c_void_p]),
    'rumpy_res_chain': (C.c_int, [_P(ResChainArgs), c_void_p]),
    'rumpy_res_chain_work_bytes': (c_int64, [c_int32, c_int32]),
    'rumpy_res_chain_strips': (c_int32, [c_int32, c_int32, c_int32]),
    'rumpy_device_xcds': (C.c_int, []),
    'rumpy_rcab2_fwd': (C.c_int, [_P(Rcab2Args), c_void_p]),
    'rumpy_rcab2_bwd': (C.c_int, [_P(Rcab2Args), c_void_p]),
    'rumpy_rcab2_partials': (C.c_int, [c_int32, c_int32, c_int32]),
    'rumpy_sgemm': (C.c_int, [_P(SgemmArgs), c_void_p]),
    'rumpy_sgemm_partial_floats': (c_int64, [c_int32, c_int32, c_int32]),
    'rumpy_l2norm_rows': (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'rumpy_l2norm_rows_bwd': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'rumpy_rowdot': (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p]),
    'rumpy_label_match': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'rumpy_pos_vector': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p]),
    'rumpy_ce_rows': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'rumpy_ce_rows_bwd': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'rumpy_colsum': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p]),
    'rumpy_lrelu_bwd': (C.c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p]),
    'rumpy_row_axpy': (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    'rumpy_moco_enqueue': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'rumpy_enc_conv': (C.c_int, [_P(EncConvArgs), c_void_p]),
    'rumpy_enc_bn_train': (C.c_int, [_P(EncBnArgs), c_void_p]),
    'rumpy_enc_bn_partial_floats': (c_int64, [c_int32, c_int32]),
    'rumpy_enc_bn_train_keep': (C.c_int, [_P(EncBnArgs), c_void_p, c_void_p, c_void_p, c_void_p]),
    'rumpy_enc_bn_bwd': (C.c_int, [_P(EncBnBwdArgs), c_void_p]),
    'rumpy_ema': (C.c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_void_p]),
    'rumpy_enc_pool': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'rumpy_head_wgrad_slab_floats': (c_int64, [c_int32, c_int32]),
    'rumpy_tail_fwd': (C.c_int, [_P(TailFwdArgs), c_void_p]),
    'rumpy_set_pointers': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'rumpy_tail_fwd_grid': (C.c_int, [c_int32, c_int32, c_int32, c_int32]),
    'rumpy_tail_wgrad_reduce': (C.c_int, [c_void_p, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    'rumpy_tail_dgrad': (C.c_int, [_P(TailDgradArgs), c_void_p]),
    'rumpy_conv4d_tail': (C.c_int, [_P(Conv4dTailArgs), c_void_p]),
    'rumpy_nchw_to_nhwc4': (C.c_int, [_P(NchwToNhwc4Args), c_void_p]),
    'rumpy_pixel_shuffle': (C.c_int, [_P(PixelShuffleArgs), c_void_p]),
    'rumpy_tail_fwd_wide': (C.c_int, [_P(TailWideArgs), c_void_p]),
    'rumpy_tail_dgrad_wide': (C.c_int, [_P(TailWideArgs), c_void_p]),
    'rumpy_wgrad_grouped': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    'rumpy_wgrad_shares': (C.c_int, [c_void_p, c_void_p, c_int32, c_void_p]),
    'rumpy_wgrad_slab_floats': (c_int64, [c_int32]),
    'rumpy_wgrad_reduce': (C.c_int, [c_void_p, c_int32, c_void_p]),
    'rumpy_pack_weights': (C.c_int, [c_void_p, c_int32, c_void_p]),
    'rumpy_ca_mlp_fwd': (C.c_int, [_P(CaMlpFwdArgs), c_void_p]),
    'rumpy_ca_scale_res_fwd': (C.c_int, [_P(CaScaleArgs), c_void_p]),
    'rumpy_ca_bwd_reduce': (C.c_int, [_P(CaBwdReduceArgs), c_void_p]),
    'rumpy_ca_mlp_bwd': (C.c_int, [_P(CaMlpBwdArgs), c_void_p]),
    'rumpy_ca_fwd_fused': (C.c_int, [_P(CaFwdFusedArgs), c_void_p]),
    'rumpy_ca_bwd_fused': (C.c_int, [_P(CaBwdFusedArgs), c_void_p]),
    'rumpy_qca_gate_fwd': (C.c_int, [_P(QcaArgs), c_void_p]),
    'rumpy_qca_gate_bwd': (C.c_int, [_P(QcaArgs), c_void_p]),
    'rumpy_qca_bwd_params': (C.c_int, [c_void_p, c_int32, c_void_p]),
    'rumpy_q_mlp_fwd': (C.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'rumpy_q_mlp_bwd_params': (C.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'rumpy_q_mlp_bwd_meta': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'rumpy_q_mlpn_fwd': (C.c_int, [c_void_p, c_int32, c_void_p, c_int32, C.POINTER(c_int32), c_int32, c_void_p]),
    'rumpy_q_mlpn_bwd_params': (C.c_int, [c_void_p, c_int32, c_void_p, c_int32, C.POINTER(c_int32), c_int32, c_void_p]),
    'rumpy_q_mlpn_bwd_meta': (C.c_int, [c_void_p, c_int32, c_int32, C.POINTER(c_int32), c_int32, c_void_p, c_void_p]),
    'rumpy_ca_mlp_bwd_params': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'rumpy_ca_bwd_apply': (C.c_int, [_P(CaBwdApplyArgs), c_void_p]),
    'rumpy_adam_step': (C.c_int, [_P(AdamArgs), c_void_p]),
    'rumpy_sumsq': (C.c_int, [_P(SumsqArgs), c_void_p]),
    'rumpy_finish_reduce': (C.c_int, [_P(FinishReduceArgs), c_void_p]),
    'rumpy_head_wgrad_slabs': (C.c_int, [c_int32, c_int32, c_int32]),
    'rumpy_adam_pack': (C.c_int, [_P(AdamPackArgs), c_void_p]),
    'rumpy_eval_post': (C.c_int, [_P(EvalPostArgs), c_void_p]),
    'rumpy_to_uint8_hwc': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p]),
    'rumpy_run_list': (C.c_int, [c_void_p, c_int32, c_void_p]),
    'rumpy_conv_block': (C.c_int, [_P(BlockArgs), c_void_p]),
    'rumpy_ssim': (C.c_int, [_P(SsimArgs), c_void_p]),
    'rumpy_ssim_partial_floats': (c_int64, [c_int32, c_int32, c_int32]),
    'rumpy_patch_gather': (C.c_int, [_P(PatchArgs), c_void_p]),
    'rumpy_dconv': (C.c_int, [_P(DconvArgs), c_void_p]),
    'rumpy_dconv_wgrad': (C.c_int, [_P(DconvWgradArgs), c_void_p]),
    'rumpy_dconv_wgrad_partial_floats': (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    'rumpy_mse_loss': (C.c_int, [_P(MseArgs), c_void_p]),
    'rumpy_debug_occupy': (C.c_int, [c_int32, c_float, c_void_p]),
    'rumpy_probe_begin': (C.c_int, [C.c_int, C.c_int]),
    'rumpy_probe_end': (C.c_int, [_P(C.c_double)]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raise loudly if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError('rumpy_amd: %s is missing - build it with `python -c "import __graft_entry__ as g; '
                               'g.build()"` (or make -C rumpy_amd/csrc). There is no CPU fallback.' % LIB_PATH)
        # torch first: the process must run on ONE HIP runtime - the one torch ships and loads; this library (NEEDED libamdhip64.so.7) then
        # binds to that copy.  Loaded the other way round, the system's runtime comes in first and the two end up with different views of
        # the device ("no ROCm-capable device is detected" from every launch here)
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().rumpy_last_error()
        raise RuntimeError('rumpy_amd %s failed (%d): %s' % (what, rc, msg.decode() if msg else '?'))


def call(name, args, stream):
    """Invoke ``int name(const args*, stream)``."""
    check(getattr(lib(), name)(C.byref(args), stream), name)
