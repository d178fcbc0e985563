"""ctypes binding of libnf_hip.so (C ABI: include/nf_hip.h).

The product path has no CPU or PyTorch fallback: if the library is missing, or a tensor is not a
contiguous-row fp32 tensor on an MI355X, the call raises.  PyTorch is used for device memory,
streams and torch.distributed only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# NF_HIP_LIB: another build of the same library (kernel A/B experiments); never a different implementation
LIB_PATH = os.environ.get('NF_HIP_LIB') or os.path.join(_HERE, 'libnf_hip.so')

NF_DTYPE_F32 = 0
NF_DTYPE_BF16 = 1
NF_DTYPE_BF16X3 = 2
NF_DTYPE_F16X2 = 3
NF_AFFINE_INTERLEAVED = 0
NF_AFFINE_BLOCKED = 1
NF_ERR_BINS = -4

_lib = None

_c = ctypes
_SIGNATURES = {
    'nf_version': (_c.c_int, []),
    'nf_strerror': (_c.c_char_p, [_c.c_int]),
    'nf_made_packed_bytes': (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    'nf_made_ktab_len': (_c.c_int, [_c.c_int]),
    'nf_made_pack': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                _c.c_void_p, _c.c_void_p]),
    'nf_masked_linear_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                        _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                        _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                        _c.c_int, _c.c_void_p]),
    'nf_masked_linear_fwd_tiles': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                              _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                              _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                              _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_made_rqs_fwd_tiles': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                         _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p,
                                         _c.c_void_p, _c.c_size_t, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                         _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int, _c.c_int,
                                         _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_rqs_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_double,
                              _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64,
                              _c.c_int, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double,
                              _c.c_double, _c.c_int, _c.c_void_p]),
    'nf_linear_spline_bwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                        _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int,
                                        _c.c_int, _c.c_int, _c.c_double, _c.c_int, _c.c_void_p]),
    'nf_quadratic_spline_bwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                           _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int,
                                           _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_int,
                                           _c.c_void_p]),
    'nf_cubic_spline_bwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_double, _c.c_void_p,
                                       _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                       _c.c_int, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_int,
                                       _c.c_void_p]),
    'nf_linear_spline_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p,
                                        _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int,
                                        _c.c_int, _c.c_double, _c.c_int, _c.c_void_p]),
    'nf_quadratic_spline_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p,
                                           _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_int,
                                           _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_int,
                                           _c.c_void_p]),
    'nf_cubic_spline_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_double,
                                       _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_int,
                                       _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_int,
                                       _c.c_void_p]),
    'nf_affine_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int,
                                 _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_int64, _c.c_int,
                                 _c.c_int, _c.c_void_p]),
    'nf_std_normal_log_prob': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_double, _c.c_void_p, _c.c_int64,
                                          _c.c_int, _c.c_void_p]),
    'nf_affine_bwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int,
                                 _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p,
                                 _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_made_rqs_supported': (_c.c_int, [_c.c_int, _c.c_int]),
    'nf_made_rqs_tiles': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int]),
    'nf_made_rqs_packed_bytes': (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'nf_made_rqs_workspace_bytes': (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int]),
    'nf_made_rqs_workspace_bytes_dtype': (_c.c_size_t, [_c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'nf_made_rqs_pack': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                    _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'nf_made_rqs_pack_scaled': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_int, _c.c_int, _c.c_double, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                           _c.c_void_p]),
    'nf_coupling_rqs_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                       _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p,
                                       _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int64, _c.c_int,
                                       _c.c_int, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double,
                                       _c.c_double, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_made_rqs_fwd_hmax': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                        _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int,
                                        _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                        _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int,
                                        _c.c_int, _c.c_void_p]),
    'nf_coupling_layer_supported': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'nf_coupling_final_packed_bytes': (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'nf_coupling_final_pack': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_double,
                                          _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'nf_coupling_rqs_layer_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_void_p,
                                             _c.c_int, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int,
                                             _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int64,
                                             _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double,
                                             _c.c_int, _c.c_void_p]),
    'nf_made_hidden_supported': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int]),
    'nf_made_hidden_packed_bytes': (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_void_p]),
    'nf_made_hidden_pack': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int,
                                       _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    'nf_made_hidden_planes_bytes': (_c.c_size_t, [_c.c_int64, _c.c_int]),
    'nf_made_hidden_planes_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                             _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64,
                                             _c.c_void_p]),
    'nf_made_rqs_fwd_planes': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                          _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int,
                                          _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_int64, _c.c_int, _c.c_int,
                                          _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double,
                                          _c.c_int, _c.c_void_p]),
    'nf_linear_split_tiles': (_c.c_int, [_c.c_int]),
    'nf_linear_split_packed_bytes': (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    'nf_linear_split_pack': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                        _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'nf_linear_split_repack': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                          _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p]),
    'nf_linear_split_pack_t': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                          _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p]),
    'nf_masked_linear_split_bwd_weight': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                                      _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                                      _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_rqs_bwd_max_bf16': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_double, _c.c_void_p,
                                       _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p,
                                       _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double,
                                       _c.c_double, _c.c_int, _c.c_void_p]),
    'nf_masked_linear_split_bwd_input_bf16': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                                         _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p,
                                                         _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_masked_linear_half_bwd_weight_bf16': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                                         _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                                         _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                                         _c.c_size_t, _c.c_void_p]),
    'nf_masked_linear_split_bwd_weight_ws': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                                          _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                                          _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                                          _c.c_size_t, _c.c_void_p]),
    'nf_masked_linear_half_bwd_weight': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                                      _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                                      _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_size_t,
                                                      _c.c_void_p]),
    'nf_maxabs': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_void_p, _c.c_void_p]),
    'nf_masked_linear_split_bwd_input': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                                    _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                                    _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_masked_linear_split_bwd_input_add': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                                        _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                                        _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int,
                                                        _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_masked_linear_split_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                              _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                              _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                              _c.c_int, _c.c_void_p]),
    'nf_resnet_blob_floats': (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    'nf_resnet_hidden_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p,
                                        _c.c_int, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_void_p]),
    'nf_resnet_hidden_fwd_copy': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p,
                                             _c.c_int, _c.c_int, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                             _c.c_int64, _c.c_void_p]),
    'nf_made_rqs_inverse_rows_supported': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'nf_made_inverse_blob_floats': (_c.c_size_t, [_c.c_int, _c.c_int]),
    'nf_made_rqs_inverse_rows': (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                            _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p,
                                            _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_size_t,
                                            _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_double, _c.c_double,
                                            _c.c_double, _c.c_double, _c.c_void_p]),
    'nf_made_affine_inverse_rows_supported': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int]),
    'nf_made_affine_inverse_rows': (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                               _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p,
                                               _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_void_p]),
    'nf_made_family_inverse_rows_supported': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'nf_made_family_inverse_rows': (_c.c_int, [_c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p,
                                               _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p,
                                               _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                                               _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double,
                                               _c.c_void_p]),
    'nf_made_rqs_fwd_params': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                          _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p,
                                          _c.c_size_t, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_double,
                                          _c.c_double, _c.c_double, _c.c_double, _c.c_int, _c.c_double, _c.c_void_p,
                                          _c.c_int64, _c.c_void_p]),
    'nf_made_rqs_fwd_params_supported': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int]),
    'nf_made_rqs_fwd_params_bf16': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                               _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p,
                                               _c.c_size_t, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                               _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int, _c.c_double,
                                               _c.c_void_p, _c.c_int64, _c.c_void_p]),
    'nf_rqs_bwd_max_bf16p': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                        _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                        _c.c_int, _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double,
                                        _c.c_void_p]),
    'nf_made_rqs_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                   _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p,
                                   _c.c_void_p, _c.c_size_t, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                   _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int, _c.c_int,
                                   _c.c_void_p]),
    'nf_made_rqs_fwd_noalloc': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                           _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int, _c.c_void_p,
                                           _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                           _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int, _c.c_int,
                                           _c.c_void_p]),
    'nf_made_pack_t': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                  _c.c_void_p, _c.c_void_p]),
    'nf_masked_linear_bwd_input': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                              _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                                              _c.c_int, _c.c_void_p]),
    'nf_masked_linear_bwd_weight': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p,
                                               _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int,
                                               _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_rqs_bwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_double, _c.c_void_p, _c.c_int64,
                              _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int,
                              _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_void_p]),
    'nf_rqs_bwd_max': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_double, _c.c_void_p, _c.c_int64,
                                  _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_int,
                                  _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double, _c.c_int,
                                  _c.c_void_p]),
    'nf_rqs_inverse_bwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_double, _c.c_void_p,
                                      _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int,
                                      _c.c_int, _c.c_int, _c.c_double, _c.c_double, _c.c_double, _c.c_double,
                                      _c.c_void_p]),
    'nf_diag_mfma_loop': (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_diag_stream_copy': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    'nf_fold_f64': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p]),
    'nf_linear_f64_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int64,
                                     _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_maf_fwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p,
                              _c.c_int64, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_linear_f64_bwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p,
                                     _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_void_p,
                                     _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_maf_bwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p,
                              _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int,
                              _c.c_void_p]),
    'nf_maf_reverse_step_bwd': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                           _c.c_void_p, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p,
                                           _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'nf_maf_reverse_step': (_c.c_int, [_c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64, _c.c_void_p, _c.c_int64,
                                       _c.c_void_p, _c.c_int64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
}
EXPORTS = tuple(_SIGNATURES)


def load():
    """Load libnf_hip.so (after torch, so both share one HIP runtime).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                'libnf_hip.so not found at %s: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                '(or `make -C normalizing-flows_amd/csrc`).  There is no fallback path.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class NfError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = load().nf_strerror(code).decode()
        super().__init__('%s failed: %s (code %d)' % (where, msg, code))


def _check(code, where):
    if code != 0:
        if code == NF_ERR_BINS:
            # same exception type and wording as the reference (rational_quadratic.py:66-69)
            raise ValueError('Minimal bin width/height too large for the number of bins')
        raise NfError(code, where)


def _dev(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('%s must be a tensor on the GPU (MI355X); this package has no CPU path' % name)
    if t.dtype != dtype:
        raise TypeError('%s must be %s, got %s' % (name, dtype, t.dtype))
    return t


def _rows(t, name):
    """2-D tensor with unit inner stride -> (ptr, row stride in elements)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise RuntimeError('%s must be 2-D with contiguous rows' % name)
    return t.data_ptr(), (t.stride(0) if t.shape[0] > 1 else max(t.shape[1], t.stride(0)))


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


# ------------------------------------------------------------------------------------------------
class PackedLinear:
    """Device buffers produced by nf_made_pack for one [out,in] layer."""
    __slots__ = ('out', 'inp', 'dtype', 'packed', 'ktab', 'bias')

    def __init__(self, weight, mask, bias, dtype=NF_DTYPE_F32, reuse=None):
        """`reuse`: a previous pack of the same layer (same mask, dtype, device): its buffers and K-tile
        table are kept and only the weights are re-folded (the per-optimizer-step path)."""
        lib = load()
        w = _dev(weight.detach().contiguous(), 'weight')
        m = None if mask is None else _dev(mask.detach().contiguous(), 'mask')
        self.out, self.inp = int(w.shape[0]), int(w.shape[1])
        self.dtype = dtype
        fresh = reuse is None or reuse.dtype != dtype or reuse.packed.device != w.device
        if fresh:
            nbytes = lib.nf_made_packed_bytes(self.out, self.inp, dtype)
            self.packed = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
            self.ktab = torch.empty(lib.nf_made_ktab_len(self.out), dtype=torch.int32, device=w.device)
        else:
            self.packed, self.ktab = reuse.packed, reuse.ktab
        self.bias = None if bias is None else _dev(bias.detach().contiguous(), 'bias')
        with torch.cuda.device(w.device):
            _check(lib.nf_made_pack(w.data_ptr(), _ptr(m), self.out, self.inp, dtype, self.packed.data_ptr(),
                                    self.ktab.data_ptr() if fresh else None, _stream()), 'nf_made_pack')


class PackedLinearSplit:
    """Device buffers of nf_linear_split_pack: one [out,in] layer as fp16 / bf16 term planes in 256-row tiles."""
    __slots__ = ('out', 'inp', 'dtype', 'packed', 'ktab', 'bias', 'tiles')

    def __init__(self, weight, mask, bias, dtype, reuse=None, transposed=False):
        """reuse: the previous pack of the same layer with the same mask - its buffers and K-range table are kept
        (nf_linear_split_repack), only the weights, bias and operand scale are rewritten.
        transposed: pack the layer's transpose (the input-gradient GEMM's operand, out / inp swapped, no bias) straight
        from weight / mask as stored (nf_linear_split_pack_t)."""
        lib = load()
        w = _dev(weight.detach().contiguous(), 'weight')
        m = None if mask is None else _dev(mask.detach().contiguous(), 'mask')
        b = None if bias is None else _dev(bias.detach().contiguous(), 'bias')
        self.out, self.inp, self.dtype = int(w.shape[0]), int(w.shape[1]), dtype
        if transposed:
            if bias is not None:
                raise ValueError('the transposed pack takes no bias')
            self.out, self.inp = self.inp, self.out
        self.tiles = lib.nf_linear_split_tiles(self.out)
        fresh = reuse is None or (reuse.out, reuse.inp, reuse.dtype) != (self.out, self.inp, dtype) \
            or reuse.packed.device != w.device
        if fresh:
            self.packed = torch.empty(lib.nf_linear_split_packed_bytes(self.out, self.inp, dtype), dtype=torch.uint8,
                                      device=w.device)
            self.ktab = torch.zeros(2 * self.tiles + 2, dtype=torch.int32, device=w.device)
            self.bias = torch.empty(256 * self.tiles, dtype=torch.float32, device=w.device)
        else:
            self.packed, self.ktab, self.bias = reuse.packed, reuse.ktab, reuse.bias
        fn = lib.nf_linear_split_pack if fresh else lib.nf_linear_split_repack
        with torch.cuda.device(w.device):
            if transposed:
                _check(lib.nf_linear_split_pack_t(w.data_ptr(), _ptr(m), self.inp, self.out, dtype,
                                                  self.packed.data_ptr(), self.ktab.data_ptr(), self.bias.data_ptr(),
                                                  int(fresh), _stream()), 'nf_linear_split_pack_t')
            else:
                _check(fn(w.data_ptr(), _ptr(m), _ptr(b), self.out, self.inp, dtype, self.packed.data_ptr(),
                          self.ktab.data_ptr(), self.bias.data_ptr(), _stream()), 'nf_linear_split_pack')


# ---- weight-cache generation ------------------------------------------------------------------------------------------
# The packed copies of a layer's weights are cached on (tensor version, storage address, ...).  Not every writer bumps
# the version counter: torch's fused optimizers (Adam(fused=True) ...) update the parameters with one kernel that leaves
# `_version` alone - a cache keyed on versions only would keep serving the OLD weights after such a step, silently.  So
# every cache key also carries this generation, which any optimizer step in the process advances (a global
# post-step hook, registered on first import), and which callers that write parameters behind torch's back
# (custom kernels, raw storage writes) advance themselves with invalidate_weight_caches().
_WEIGHT_GENERATION = [0]


def invalidate_weight_caches():
    """Make every cached weight pack stale (they are rebuilt at their next use)."""
    _WEIGHT_GENERATION[0] += 1


def weight_version(t):
    """Cache-key element for a parameter / buffer tensor: (version counter, weight-cache generation)."""
    return (t._version, _WEIGHT_GENERATION[0])


def _register_optimizer_hook():
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
    except ImportError:             # (then only version-bumping writers are seen; invalidate_weight_caches() remains)
        return
    register_optimizer_step_post_hook(lambda optimizer, args, kwargs: invalidate_weight_caches())


_register_optimizer_hook()


def split_gemm_supported(x, in_features):
    """Shapes the split-operand hidden-layer kernel takes: 16-byte rows, in_features a multiple of 8."""
    return (in_features % 8 == 0 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0
            and x.data_ptr() % 16 == 0)


def tag_max(t, slot):
    """Attach `slot` (max |t| as the producing kernel left it) to t.  The tag names the storage and its version, so
    an in-place write to t afterwards (y.mul_(), normal_() into the same object ...) makes it stale instead of
    handing a wrong fp16 operand scale to the next split-operand launch."""
    t._nf_max = None if slot is None else (slot, t._version, t.data_ptr())


def tagged_max(t):
    """The slot attached by tag_max, or None when there is none or t was written since."""
    tag = getattr(t, '_nf_max', None)
    if tag is None:
        return None
    slot, version, ptr = tag
    return slot if (t._version == version and t.data_ptr() == ptr) else None


def maxabs(x):
    """int32[1] slot holding the bit pattern of the largest finite |x| (operand scale of the f16x2 kernels)."""
    lib = load()
    _dev(x, 'x')
    xp, xs = _rows(x, 'x')
    slot = torch.empty(1, dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.nf_maxabs(xp, xs, x.shape[0], x.shape[1], slot.data_ptr(), _stream()), 'nf_maxabs')
    return slot


def masked_linear_split(x, pls, residual=None, relu_in=False, relu_out=False, out=None, x_max=None, gate=None,
                        reverse_k=False, add=None):
    """Split-operand hidden-layer GEMM.  The result is tagged (tag_max) with the slot holding max |out| so that a
    following split-operand launch needs no pass over it; x's own slot (`x_max`, else its valid tag) is used when
    present, else computed here.  add (with reverse_k and a gate): a second gradient joining behind the gate
    (nf_masked_linear_split_bwd_input_add)."""
    lib = load()
    _dev(x, 'x')
    xp, xs = _rows(x, 'x')
    B = x.shape[0]
    if x.shape[1] != pls.inp:
        raise ValueError('Expected %d input features, got %d' % (pls.inp, x.shape[1]))
    if out is None:
        out = torch.empty((B, pls.out), dtype=torch.float32, device=x.device)
    yp, ys = _rows(out, 'out')
    rp, rs = (None, 0) if residual is None else _rows(_dev(residual, 'residual'), 'residual')
    if gate is not None:                      # y zeroed where gate <= 0 (ReLU derivative of an input-gradient GEMM)
        if residual is not None or relu_out:
            raise ValueError('gate excludes residual / relu_out')
        rp, rs = _rows(_dev(gate, 'gate'), 'gate')
        relu_out = 2
    if x_max is None:
        x_max = tagged_max(x)
    if x_max is None and pls.dtype == NF_DTYPE_F16X2:
        x_max = maxabs(x)
    y_max = torch.empty(1, dtype=torch.int32, device=x.device)

    if reverse_k and (relu_in or (residual is not None and gate is None) or relu_out == 1):
        raise ValueError('reverse_k is the input-gradient launch: x = grad_y, pls = the transposed pack, optional gate')
    if add is not None:
        if not reverse_k or gate is None:
            raise ValueError('add joins an input-gradient launch behind its gate (reverse_k, gate)')
        if add.shape != out.shape:
            raise ValueError('add: expected shape %s, got %s' % (tuple(out.shape), tuple(add.shape)))
        ap, as_ = _rows(_dev(add, 'add'), 'add')

    def launch():
        if add is not None:
            return lib.nf_masked_linear_split_bwd_input_add(xp, xs, _ptr(x_max), pls.packed.data_ptr(),
                                                            pls.ktab.data_ptr(), pls.bias.data_ptr(), rp, rs, ap, as_, yp,
                                                            ys, y_max.data_ptr(), B, pls.inp, pls.out, pls.dtype,
                                                            _stream())
        if reverse_k:       # nf_masked_linear_split_bwd_input: the live K ranges walked from their (common) end down
            return lib.nf_masked_linear_split_bwd_input(xp, xs, _ptr(x_max), pls.packed.data_ptr(), pls.ktab.data_ptr(),
                                                        pls.bias.data_ptr(), rp, rs, yp, ys, y_max.data_ptr(), B,
                                                        pls.inp, pls.out, pls.dtype, _stream())
        return lib.nf_masked_linear_split_fwd(xp, xs, _ptr(x_max), pls.packed.data_ptr(), pls.ktab.data_ptr(),
                                              pls.bias.data_ptr(), rp, rs, yp, ys, y_max.data_ptr(), B, pls.inp, pls.out,
                                              int(relu_in), int(relu_out), pls.dtype, _stream())
    with torch.cuda.device(x.device):
        code = launch() if _timer is None else _timer.record(('masked_linear', pls.inp, pls.out, B), launch)
        _check(code, 'nf_masked_linear_split_fwd')
    tag_max(out, y_max)
    return out


class PackedLinearT:
    """Transposed pack (nf_made_pack_t) for the input-gradient GEMM of one [out,in] layer."""
    __slots__ = ('out', 'inp', 'dtype', 'packed', 'ktab')

    def __init__(self, weight, mask, dtype=NF_DTYPE_F32, reuse=None):
        lib = load()
        w = _dev(weight.detach().contiguous(), 'weight')
        m = None if mask is None else _dev(mask.detach().contiguous(), 'mask')
        self.out, self.inp, self.dtype = int(w.shape[0]), int(w.shape[1]), dtype
        fresh = reuse is None or reuse.dtype != dtype or reuse.packed.device != w.device
        if fresh:
            self.packed = torch.empty(lib.nf_made_packed_bytes(self.inp, self.out, dtype), dtype=torch.uint8,
                                      device=w.device)
            self.ktab = torch.empty(lib.nf_made_ktab_len(self.inp), dtype=torch.int32, device=w.device)
        else:
            self.packed, self.ktab = reuse.packed, reuse.ktab
        with torch.cuda.device(w.device):
            _check(lib.nf_made_pack_t(w.data_ptr(), _ptr(m), self.out, self.inp, dtype, self.packed.data_ptr(),
                                      self.ktab.data_ptr() if fresh else None, _stream()), 'nf_made_pack_t')


def masked_linear_bwd_input(grad_y, plt, gate=None):
    lib = load()
    _dev(grad_y, 'grad_y')
    gp, gs = _rows(grad_y, 'grad_y')
    B = grad_y.shape[0]
    gx = torch.empty((B, plt.inp), dtype=torch.float32, device=grad_y.device)
    xp, xs = _rows(gx, 'grad_x')
    tp, ts = (None, 0) if gate is None else _rows(_dev(gate, 'gate'), 'gate')
    with torch.cuda.device(grad_y.device):
        _check(lib.nf_masked_linear_bwd_input(gp, gs, plt.packed.data_ptr(), plt.ktab.data_ptr(), tp, ts, xp, xs, B,
                                              plt.inp, plt.out, plt.dtype, _stream()), 'nf_masked_linear_bwd_input')
    return gx


def _wgrad_out(out, inp, want_bias, device, into):
    """(grad_w, grad_b) the weight-gradient kernels ACCUMULATE into: fresh zeros, or `into` = the caller's own
    (grad_w, grad_b or None) - e.g. the parameters' .grad views, already holding earlier contributions."""
    if into is None:
        gw = torch.zeros((out, inp), dtype=torch.float32, device=device)
        gb = torch.zeros(out, dtype=torch.float32, device=device) if want_bias else None
        return gw, gb
    gw, gb = into
    for t, shape, name in ((gw, (out, inp), 'grad_w'), (gb, (out,), 'grad_b')):
        if t is None and name == 'grad_b' and not want_bias:
            continue
        if t is None or t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous() or t.device != device:
            raise ValueError('%s to accumulate into: a contiguous float32 tensor of shape %s on %s' % (name, shape, device))
    return gw, (gb if want_bias else None)


def masked_linear_bwd_weight(x, grad_y, mask, ktab, relu_in, want_bias=True, into=None):
    """Returns (grad_w [out,in], grad_b [out] or None), freshly zeroed then accumulated by the kernel (`into`: the
    caller's own pair instead, see _wgrad_out)."""
    lib = load()
    _dev(x, 'x')
    _dev(grad_y, 'grad_y')
    xp, xs = _rows(x, 'x')
    gp, gs = _rows(grad_y, 'grad_y')
    B, inp = x.shape
    out = grad_y.shape[1]
    gw, gb = _wgrad_out(out, inp, want_bias, x.device, into)
    m = None if mask is None else _dev(mask.contiguous(), 'mask')
    with torch.cuda.device(x.device):
        _check(lib.nf_masked_linear_bwd_weight(xp, xs, gp, gs, _ptr(m), _ptr(ktab), gw.data_ptr(), _ptr(gb), B, inp, out,
                                               int(relu_in), NF_DTYPE_F32, _stream()), 'nf_masked_linear_bwd_weight')
    return gw, gb


def masked_linear_split_bwd_weight(x, grad_y, mask, ktab, relu_in, want_bias=True, x_max=None, grad_y_max=None,
                                   terms=2, into=None):
    """masked_linear_bwd_weight on two-term fp16 operands (the f16x2 training mode) or, terms=1, on one scaled fp16
    term (the bf16 training mode: nf_masked_linear_half_bwd_weight); the max slots are computed here when the caller
    has none."""
    lib = load()
    _dev(x, 'x')
    _dev(grad_y, 'grad_y')
    xp, xs = _rows(x, 'x')
    gp, gs = _rows(grad_y, 'grad_y')
    B, inp = x.shape
    out = grad_y.shape[1]
    if grad_y.shape[0] != B:
        raise ValueError('x and grad_y disagree on the batch size')
    x_max = maxabs(x) if x_max is None else x_max
    grad_y_max = maxabs(grad_y) if grad_y_max is None else grad_y_max
    gw, gb = _wgrad_out(out, inp, want_bias, x.device, into)
    m = None if mask is None else _dev(mask.contiguous(), 'mask')
    with torch.cuda.device(x.device):
        # wide layers: act(x) as fp16 plane(s) in a workspace (the operand every workgroup re-reads, fetched by LDS-DMA)
        # (two terms: every layer - the transposed-read kernel splits x only through its planes; measured 188 -> 110 us
        # on 512 x 512 at B = 32768, planes included)
        ws = torch.empty(terms * B * inp * 2, dtype=torch.uint8, device=x.device) \
            if (out >= 2048 or terms == 2) and inp % 8 == 0 else None
        if terms == 2:
            _check(lib.nf_masked_linear_split_bwd_weight_ws(xp, xs, x_max.data_ptr(), gp, gs, grad_y_max.data_ptr(), _ptr(m),
                                                            _ptr(ktab), gw.data_ptr(), _ptr(gb), B, inp, out,
                                                            int(relu_in), _ptr(ws), 0 if ws is None else ws.numel(),
                                                            _stream()), 'nf_masked_linear_split_bwd_weight')
        else:
            _check(lib.nf_masked_linear_half_bwd_weight(xp, xs, x_max.data_ptr(), gp, gs, grad_y_max.data_ptr(), _ptr(m),
                                                        _ptr(ktab), gw.data_ptr(), _ptr(gb), B, inp, out, int(relu_in),
                                                        _ptr(ws), 0 if ws is None else ws.numel(), _stream()),
                   'nf_masked_linear_half_bwd_weight')
    return gw, gb


# Test knob: False sends the bf16 mode's parameter gradients through the fp32 tensor again (same operand rounding in the
# GEMMs, so the two settings must agree to fp32 summation order: tests/test_gpu_parity.py compares them).
BF16_GRADIENT_PLANE = True


def bf16_gradient_path(inp, out):
    """Does the final layer [out, inp] take its output gradient as a bf16 matrix in both backward GEMMs
    (nf_masked_linear_split_bwd_input_bf16: whole 16-wide K steps; nf_masked_linear_half_bwd_weight_bf16: the wide-layer
    workspace path)?"""
    if not BF16_GRADIENT_PLANE:
        return False
    return out % 16 == 0 and inp % 8 == 0 and ((out + 127) // 128) * ((inp + 255) // 256) > 128


def masked_linear_split_bwd_input_bf16(grad_y16, pls, gate=None):
    """grad_x = grad_y @ (W * mask) with grad_y a torch.bfloat16 matrix, pls the transposed NF_DTYPE_BF16 pack."""
    lib = load()
    if grad_y16.dtype != torch.bfloat16 or not grad_y16.is_cuda or grad_y16.stride(1) != 1:
        raise TypeError('grad_y must be a CUDA bfloat16 matrix with contiguous rows')
    if pls.dtype != NF_DTYPE_BF16 or grad_y16.shape[1] != pls.inp:
        raise ValueError('expected the transposed bf16 pack of a layer with %d outputs' % grad_y16.shape[1])
    B = grad_y16.shape[0]
    gx = torch.empty((B, pls.out), dtype=torch.float32, device=grad_y16.device)
    xp, xs = _rows(gx, 'grad_x')
    tp, ts = (None, 0) if gate is None else _rows(_dev(gate, 'gate'), 'gate')
    y_max = torch.empty(1, dtype=torch.int32, device=grad_y16.device)
    with torch.cuda.device(grad_y16.device):
        _check(lib.nf_masked_linear_split_bwd_input_bf16(grad_y16.data_ptr(), grad_y16.stride(0), pls.packed.data_ptr(),
                                                         pls.ktab.data_ptr(), pls.bias.data_ptr(), tp, ts, xp, xs,
                                                         y_max.data_ptr(), B, pls.inp, pls.out, _stream()),
               'nf_masked_linear_split_bwd_input_bf16')
    tag_max(gx, y_max)
    return gx


def masked_linear_half_bwd_weight_bf16(x, grad_y16, mask, ktab, relu_in, want_bias=True, x_max=None, grad_y_max=None,
                                       into=None):
    """The one-term weight gradient with grad_y a torch.bfloat16 matrix (its max slot is required: the bf16 tensor's tag)."""
    lib = load()
    _dev(x, 'x')
    if grad_y16.dtype != torch.bfloat16 or not grad_y16.is_cuda or grad_y16.stride(1) != 1:
        raise TypeError('grad_y must be a CUDA bfloat16 matrix with contiguous rows')
    if grad_y_max is None:
        raise ValueError('grad_y_max: the slot nf_rqs_bwd_max_bf16 left')
    xp, xs = _rows(x, 'x')
    B, inp = x.shape
    out = grad_y16.shape[1]
    m = None if mask is None else _dev(mask.detach().contiguous(), 'mask')
    x_max = maxabs(x) if x_max is None else x_max
    gw, gb = _wgrad_out(out, inp, want_bias, x.device, into)
    ws = torch.empty(B * inp * 2, dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib.nf_masked_linear_half_bwd_weight_bf16(xp, xs, x_max.data_ptr(), grad_y16.data_ptr(), grad_y16.stride(0),
                                                         grad_y_max.data_ptr(), _ptr(m), _ptr(ktab), gw.data_ptr(),
                                                         _ptr(gb), B, inp, out, int(relu_in), ws.data_ptr(), ws.numel(),
                                                         _stream()), 'nf_masked_linear_half_bwd_weight_bf16')
    return gw, gb


def rqs_bwd(x, params, grad_y, grad_lad, num_bins, tails, tail_bound, pre_div=1.0, min_bin_width=1e-3,
            min_bin_height=1e-3, min_derivative=1e-3, inverse=False, want_max=False, bf16_out=False):
    """(grad wrt the spline's input, grad_params).  inverse: backward of the inverse direction - `x` is then that
    pass's OUTPUT and grad_y the gradient wrt it (nf_rqs_inverse_bwd).  bf16_out: grad_params come back as a
    torch.bfloat16 tensor tagged with max |gradient| (nf_rqs_bwd_max_bf16) - for a caller that feeds them to the bf16
    backward GEMMs and nowhere else."""
    lib = load()
    for t, n in ((x, 'inputs'), (grad_y, 'grad_y'), (grad_lad, 'grad_logabsdet')):
        _dev(t, n)
    _dev(params, 'params', params.dtype if params.dtype == torch.bfloat16 else torch.float32)
    B, D = x.shape
    xp, xs = _rows(x, 'inputs')
    pp, ps = _rows(params, 'params')
    gyp, gys = _rows(grad_y, 'grad_y')
    gx = torch.empty((B, D), dtype=torch.float32, device=x.device)
    gxp, gxs = _rows(gx, 'grad_x')
    gl = grad_lad.contiguous()
    if params.dtype == torch.bfloat16:
        # parameters kept as bf16 by made_rqs_params(keep_bf16=True): gradients come back as bf16 too
        if not bf16_out or inverse or pre_div != 1.0:
            raise ValueError('bf16 parameters: forward direction, bf16 gradients, no pre-divisor')
        gparams = torch.empty(params.shape, dtype=torch.bfloat16, device=x.device)
        slot = torch.empty(1, dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _check(lib.nf_rqs_bwd_max_bf16p(xp, xs, pp, ps, gyp, gys, gl.data_ptr(), gxp, gxs, gparams.data_ptr(),
                                            slot.data_ptr(), B, D, int(num_bins), 1 if tails == 'linear' else 0,
                                            float(tail_bound), float(min_bin_width), float(min_bin_height),
                                            float(min_derivative), _stream()), 'nf_rqs_bwd_max_bf16p')
        tag_max(gparams, slot)
        return gx, gparams
    if bf16_out:
        gparams = torch.empty(params.shape, dtype=torch.bfloat16, device=x.device)
        slot = torch.empty(1, dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _check(lib.nf_rqs_bwd_max_bf16(xp, xs, pp, ps, float(pre_div), gyp, gys, gl.data_ptr(), gxp, gxs,
                                           gparams.data_ptr(), slot.data_ptr(), B, D, int(num_bins),
                                           1 if tails == 'linear' else 0, float(tail_bound), float(min_bin_width),
                                           float(min_bin_height), float(min_derivative), int(inverse), _stream()),
                   'nf_rqs_bwd_max_bf16')
        tag_max(gparams, slot)
        return gx, gparams
    gparams = torch.empty_like(params)
    with torch.cuda.device(x.device):
        if want_max:
            # the kernel leaves max |grad_params| in a slot and the gradient carries it as a tag: the f16x2 GEMMs of
            # the conditioner's backward then skip their own pass over this (large) tensor
            slot = torch.empty(1, dtype=torch.int32, device=x.device)
            _check(lib.nf_rqs_bwd_max(xp, xs, pp, ps, float(pre_div), gyp, gys, gl.data_ptr(), gxp, gxs,
                                      gparams.data_ptr(), slot.data_ptr(), B, D, int(num_bins),
                                      1 if tails == 'linear' else 0, float(tail_bound), float(min_bin_width),
                                      float(min_bin_height), float(min_derivative), int(inverse), _stream()),
                   'nf_rqs_bwd_max')
            tag_max(gparams, slot)
            return gx, gparams
        fn = lib.nf_rqs_inverse_bwd if inverse else lib.nf_rqs_bwd
        _check(fn(xp, xs, pp, ps, float(pre_div), gyp, gys, gl.data_ptr(), gxp, gxs, gparams.data_ptr(), B,
                  D, int(num_bins), 1 if tails == 'linear' else 0, float(tail_bound),
                  float(min_bin_width), float(min_bin_height), float(min_derivative), _stream()),
               'nf_rqs_inverse_bwd' if inverse else 'nf_rqs_bwd')
    return gx, gparams


class PackedFusedFinal:
    """Device buffers of nf_made_rqs_pack: the final MADE layer regrouped to whole features per tile."""
    __slots__ = ('features', 'inp', 'K', 'tails', 'dtype', 'packed', 'ktab', 'bias', 'tiles')

    def __init__(self, weight, mask, bias, features, num_bins, tails, dtype=NF_DTYPE_F32, logit_div=1.0, reuse=None):
        """reuse: the previous pack of the same layer and spline shape with the same mask: its buffers are rewritten in
        place and (f32 / bf16) its K-range table is kept - the per-optimizer-step path of training."""
        lib = load()
        w = _dev(weight.detach().contiguous(), 'weight')
        m = None if mask is None else _dev(mask.detach().contiguous(), 'mask')
        b = None if bias is None else _dev(bias.detach().contiguous(), 'bias')
        t = 1 if tails == 'linear' else 0
        self.features, self.inp, self.K, self.tails, self.dtype = int(features), int(w.shape[1]), int(num_bins), t, dtype
        self.tiles = lib.nf_made_rqs_tiles(self.features, self.K, t)
        if self.tiles == 0:
            raise NfError(-2, 'nf_made_rqs_tiles')
        fresh = reuse is None or reuse.packed.device != w.device or \
            (reuse.features, reuse.inp, reuse.K, reuse.tails, reuse.dtype) != (self.features, self.inp, self.K, t, dtype)
        if fresh:
            self.packed = torch.empty(lib.nf_made_rqs_packed_bytes(self.features, self.inp, self.K, t, dtype),
                                      dtype=torch.uint8, device=w.device)
            self.ktab = torch.zeros(2 * self.tiles + 2, dtype=torch.int32, device=w.device)   # + the f16x2 scale pair
            self.bias = torch.empty(256 * self.tiles, dtype=torch.float32, device=w.device)
        else:
            self.packed, self.ktab, self.bias = reuse.packed, reuse.ktab, reuse.bias
        keep_table = not fresh and dtype in (NF_DTYPE_F32, NF_DTYPE_BF16)     # (f16x2 rewrites its scale pair there)
        with torch.cuda.device(w.device):
            _check(lib.nf_made_rqs_pack_scaled(w.data_ptr(), _ptr(m), _ptr(b), self.features, self.inp, self.K, t,
                                               dtype, float(logit_div), self.packed.data_ptr(),
                                               None if keep_table else self.ktab.data_ptr(),
                                               self.bias.data_ptr(), _stream()), 'nf_made_rqs_pack')


def fused_workspace(pf, rows, device, planes=True):
    """Scratch for made_rqs on `rows` samples: per-tile logdet partials, the split modes' scale slot and, with
    `planes`, their pre-split hidden state (whole-layer launches)."""
    lib = load()
    if planes:
        need = lib.nf_made_rqs_workspace_bytes_dtype(rows, pf.features, pf.inp, pf.K, pf.tails, pf.dtype)
    else:
        need = lib.nf_made_rqs_workspace_bytes(rows, pf.features, pf.K, pf.tails) + 512
    return torch.empty(max(1, (need + 3) // 4), dtype=torch.float32, device=device)


def made_rqs_supported(num_bins, tails, hidden):
    return (tails in (None, 'linear') and hidden % 8 == 0
            and load().nf_made_rqs_supported(int(num_bins), 1 if tails == 'linear' else 0) == 1)


def made_rqs(h, pf, x, tail_bound, inverse=False, out=None, logabsdet=None, accumulate=False, min_bin_width=1e-3,
             min_bin_height=1e-3, min_derivative=1e-3, want_bins=False, tiles=None, feat_limit=None, reduce=True,
             workspace=None, feat_index=None):
    """Fused final MADE layer + spline: h [B, hidden], x [B, features] -> (y, logabsdet[B], bins or None).
    `tiles=(begin, count)`, `feat_limit`, `reduce`, `workspace`: one pass of the incremental inverse.
    `feat_index` (int32 [features]): coupling layers - feature f lives in column feat_index[f] of x and y,
    whose other columns are left alone (nf_coupling_rqs_fwd)."""
    lib = load()
    _dev(h, 'hidden')
    _dev(x, 'inputs')
    B = x.shape[0]
    hp, hs = _rows(h, 'hidden')
    xp, xs = _rows(x, 'inputs')
    if h.shape[0] != B or h.shape[1] != pf.inp or (feat_index is None and x.shape[1] != pf.features):
        raise ValueError('Expected hidden [B,%d] and inputs [B,%d]' % (pf.inp, pf.features))
    if feat_index is not None:
        _dev(feat_index, 'feat_index', torch.int32)
        if feat_index.dtype != torch.int32 or feat_index.numel() != pf.features or out is None \
                or out.shape != x.shape or tiles is not None or feat_limit is not None:
            raise ValueError('coupling launch: int32 feat_index [%d], caller-provided outputs, no tile window'
                             % pf.features)
    if out is None:
        out = torch.empty_like(x)
    else:
        tag_max(out, None)              # (a max slot left by an earlier writer no longer describes this buffer)
    yp, ys = _rows(out, 'outputs')
    if logabsdet is None:
        logabsdet = torch.empty(B, dtype=torch.float32, device=x.device)
        accumulate = False
    bins = torch.empty((B, pf.features), dtype=torch.int32, device=x.device) if want_bins else None
    if workspace is None:
        workspace = fused_workspace(pf, B, x.device, planes=tiles is None)
    ws = workspace
    tb, tc = (0, pf.tiles) if tiles is None else tiles
    fl = pf.features if feat_limit is None else feat_limit
    # a split-operand producer left max |h| beside the tensor: whole-layer launches can skip their own pass
    whole_split = (tiles is None and feat_limit is None and feat_index is None
                   and pf.dtype in (NF_DTYPE_F16X2, NF_DTYPE_BF16X3))
    h_max = tagged_max(h) if whole_split else None
    y_max = torch.empty(1, dtype=torch.int32, device=x.device) if whole_split else None

    def launch():
        if feat_index is not None:
            return lib.nf_coupling_rqs_fwd(hp, hs, pf.packed.data_ptr(), pf.ktab.data_ptr(), pf.bias.data_ptr(), xp, xs,
                                           yp, ys, feat_index.data_ptr(), x.shape[1], logabsdet.data_ptr(),
                                           int(accumulate), _ptr(bins), ws.data_ptr(), ws.numel() * 4, B, pf.inp,
                                           pf.features, pf.K, pf.tails, float(tail_bound), float(min_bin_width),
                                           float(min_bin_height), float(min_derivative), int(inverse), pf.dtype,
                                           _stream())
        if whole_split:
            return lib.nf_made_rqs_fwd_hmax(hp, hs, _ptr(h_max), pf.packed.data_ptr(), pf.ktab.data_ptr(),
                                            pf.bias.data_ptr(), xp, xs, yp, ys, y_max.data_ptr(), logabsdet.data_ptr(),
                                            int(accumulate),
                                            _ptr(bins), ws.data_ptr(), ws.numel() * 4, B, pf.inp, pf.features, pf.K,
                                            pf.tails, float(tail_bound), float(min_bin_width), float(min_bin_height),
                                            float(min_derivative), int(inverse), pf.dtype, _stream())
        return lib.nf_made_rqs_fwd_tiles(hp, hs, pf.packed.data_ptr(), pf.ktab.data_ptr(), pf.bias.data_ptr(), xp, xs,
                                         yp, ys, logabsdet.data_ptr(), int(accumulate), _ptr(bins), ws.data_ptr(),
                                         ws.numel() * 4, B, pf.inp, pf.features, pf.K, pf.tails, float(tail_bound),
                                         float(min_bin_width), float(min_bin_height), float(min_derivative),
                                         int(inverse), pf.dtype, int(tb), int(tc), int(fl), int(bool(reduce)),
                                         _stream())
    with torch.cuda.device(x.device):
        code = launch() if _timer is None else _timer.record(('made_rqs', pf.inp, pf.features * (3 * pf.K - 1), B), launch)
        _check(code, 'nf_made_rqs_fwd')
    tag_max(out, y_max)                 # (None outside the split-operand whole-layer launches)
    return out, logabsdet, bins


class PackedHiddenStack:
    """Device blob of nf_made_hidden_pack: every hidden layer of a MADE in the operand order of the one-launch hidden
    stack kernel (f16x2)."""
    __slots__ = ('n_layers', 'hidden', 'in0', 'blob')

    def __init__(self, layers, types, k_live0):
        """layers: [(weight, mask or None, bias or None)], types: HS_* role of each (see include/nf_hip.h)."""
        lib = load()
        ws = [_dev(w.detach().contiguous(), 'weight') for w, _, _ in layers]
        ms = [None if m is None else _dev(m.detach().contiguous(), 'mask') for _, m, _ in layers]
        bs = [None if b is None else _dev(b.detach().contiguous(), 'bias') for _, _, b in layers]
        n = len(layers)
        self.n_layers, self.hidden, self.in0 = n, int(ws[0].shape[0]), int(ws[0].shape[1])
        ins = (_c.c_int * n)(*[int(w.shape[1]) for w in ws])
        tys = (_c.c_int * n)(*[int(t) for t in types])
        if any(int(w.shape[0]) != self.hidden for w in ws):
            raise ValueError('every hidden layer must have %d outputs' % self.hidden)
        nbytes = lib.nf_made_hidden_packed_bytes(n, self.hidden, ins)
        if nbytes == 0 or not lib.nf_made_hidden_supported(n, self.hidden, int(k_live0)):
            raise NfError(-2, 'nf_made_hidden_pack')
        self.blob = torch.empty(nbytes, dtype=torch.uint8, device=ws[0].device)
        wp = (_c.c_void_p * n)(*[w.data_ptr() for w in ws])
        mp = (_c.c_void_p * n)(*[_ptr(m) for m in ms])
        bp = (_c.c_void_p * n)(*[_ptr(b) for b in bs])
        with torch.cuda.device(ws[0].device):
            _check(lib.nf_made_hidden_pack(wp, mp, bp, ins, tys, n, self.hidden, int(k_live0), self.blob.data_ptr(),
                                           _stream()), 'nf_made_hidden_pack')


def made_hidden_supported(n_layers, hidden, k_live0):
    return load().nf_made_hidden_supported(int(n_layers), int(hidden), int(k_live0)) == 1


def made_hidden_planes(x, ph, want_hidden=False):
    """x [B, in0] -> (planes, row_scale, hidden or None): the whole hidden stack in one launch (nf_made_hidden_planes_fwd);
    planes / row_scale are what made_rqs_planes takes."""
    lib = load()
    _dev(x, 'inputs')
    xp, xs = _rows(x, 'inputs')
    B = x.shape[0]
    if x.shape[1] != ph.in0:
        raise ValueError('Expected %d input features, got %d' % (ph.in0, x.shape[1]))
    planes = torch.empty(max(16, lib.nf_made_hidden_planes_bytes(B, ph.hidden)), dtype=torch.uint8, device=x.device)
    row_scale = torch.zeros(2 * ((B + 127) // 128) + 2, dtype=torch.int32, device=x.device)
    hid = torch.empty((B, ph.hidden), dtype=torch.float32, device=x.device) if want_hidden else None

    def launch():
        return lib.nf_made_hidden_planes_fwd(xp, xs, ph.blob.data_ptr(), ph.n_layers, ph.hidden, ph.in0,
                                             planes.data_ptr(), row_scale.data_ptr(), _ptr(hid),
                                             0 if hid is None else hid.stride(0), B, _stream())
    with torch.cuda.device(x.device):
        code = launch() if _timer is None else _timer.record(('made_hidden', ph.in0, ph.hidden, B), launch)
        _check(code, 'nf_made_hidden_planes_fwd')
    return planes, row_scale, hid


def made_rqs_planes(planes, row_scale, pf, x, tail_bound, inverse=False, out=None, logabsdet=None, accumulate=False,
                    min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, want_bins=False, workspace=None):
    """made_rqs on the planes of made_hidden_planes (f16x2 packs only)."""
    lib = load()
    _dev(x, 'inputs')
    B = x.shape[0]
    xp, xs = _rows(x, 'inputs')
    if pf.dtype != NF_DTYPE_F16X2 or x.shape[1] != pf.features:
        raise ValueError('made_rqs_planes: f16x2 pack and inputs [B,%d]' % pf.features)
    if out is None:
        out = torch.empty_like(x)
    else:
        tag_max(out, None)              # (a max slot left by an earlier writer no longer describes this buffer)
    yp, ys = _rows(out, 'outputs')
    if logabsdet is None:
        logabsdet = torch.empty(B, dtype=torch.float32, device=x.device)
        accumulate = False
    bins = torch.empty((B, pf.features), dtype=torch.int32, device=x.device) if want_bins else None
    ws = fused_workspace(pf, B, x.device, planes=False) if workspace is None else workspace
    y_max = torch.empty(1, dtype=torch.int32, device=x.device)

    def launch():
        return lib.nf_made_rqs_fwd_planes(planes.data_ptr(), row_scale.data_ptr(), pf.packed.data_ptr(),
                                          pf.ktab.data_ptr(), pf.bias.data_ptr(), xp, xs, yp, ys, y_max.data_ptr(),
                                          logabsdet.data_ptr(), int(accumulate), _ptr(bins), ws.data_ptr(),
                                          ws.numel() * 4, B, pf.inp, pf.features, pf.K, pf.tails, float(tail_bound),
                                          float(min_bin_width), float(min_bin_height), float(min_derivative),
                                          int(inverse), _stream())
    with torch.cuda.device(x.device):
        code = launch() if _timer is None else _timer.record(('made_rqs', pf.inp, pf.features * (3 * pf.K - 1), B), launch)
        _check(code, 'nf_made_rqs_fwd_planes')
    tag_max(out, y_max)
    return out, logabsdet, bins


def made_rqs_keeps_params(pf):
    """Can the fused launch also write the conditioner output (nf_made_rqs_fwd_params)?  The f32 / bf16 kernel, and the
    f16x2 t4 kernel where its stages fit LDS."""
    return made_rqs_params_supported(pf.K, 'linear' if pf.tails else None, pf.dtype)


def made_rqs_params_supported(num_bins, tails, dtype):
    """Does the kept-parameters training forward (made_rqs_params) exist for this spline shape and pack dtype?"""
    return tails in (None, 'linear') and \
        load().nf_made_rqs_fwd_params_supported(int(num_bins), 1 if tails == 'linear' else 0, int(dtype)) == 1


def made_rqs_params(h, pf, x, tail_bound, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, logit_div=1.0,
                    keep_bf16=False):
    """The training forward of a whole autoregressive RQ layer: fused final MADE layer + spline that also KEEPS the
    conditioner output (what the spline's backward recomputes from).  -> (y, logabsdet[B], params[B, features * P]).
    `logit_div`: the value pf was packed with.  keep_bf16: params come back as torch.bfloat16
    (nf_made_rqs_fwd_params_bf16; y / logabsdet are computed from the fp32 accumulators either way)."""
    lib = load()
    _dev(h, 'hidden')
    _dev(x, 'inputs')
    B = x.shape[0]
    hp, hs = _rows(h, 'hidden')
    xp, xs = _rows(x, 'inputs')
    if h.shape[0] != B or h.shape[1] != pf.inp or x.shape[1] != pf.features:
        raise ValueError('Expected hidden [B,%d] and inputs [B,%d]' % (pf.inp, pf.features))
    P = 3 * pf.K - 1 if pf.tails else 3 * pf.K + 1
    out = torch.empty_like(x)
    yp, ys = _rows(out, 'outputs')
    logabsdet = torch.empty(B, dtype=torch.float32, device=x.device)
    params = torch.empty((B, pf.features * P), dtype=torch.bfloat16 if keep_bf16 else torch.float32, device=x.device)
    ws = fused_workspace(pf, B, x.device, planes=pf.dtype == NF_DTYPE_F16X2)     # (f16x2: the t4 kernel reads planes)
    y_max = torch.empty(1, dtype=torch.int32, device=x.device) if keep_bf16 else None

    def launch():
        if keep_bf16:       # (also leaves max |y|: the next layer's weight gradient scales its fp16 operand by it)
            return lib.nf_made_rqs_fwd_params_bf16(hp, hs, pf.packed.data_ptr(), pf.ktab.data_ptr(), pf.bias.data_ptr(),
                                                   xp, xs, yp, ys, y_max.data_ptr(), logabsdet.data_ptr(), 0,
                                                   ws.data_ptr(), ws.numel() * 4, B, pf.inp, pf.features, pf.K, pf.tails,
                                                   float(tail_bound), float(min_bin_width), float(min_bin_height),
                                                   float(min_derivative), pf.dtype, float(logit_div), params.data_ptr(),
                                                   params.stride(0), _stream())
        return lib.nf_made_rqs_fwd_params(hp, hs, pf.packed.data_ptr(), pf.ktab.data_ptr(), pf.bias.data_ptr(), xp, xs,
                                          yp, ys, logabsdet.data_ptr(), 0, ws.data_ptr(), ws.numel() * 4, B, pf.inp,
                                          pf.features, pf.K, pf.tails, float(tail_bound), float(min_bin_width),
                                          float(min_bin_height), float(min_derivative), pf.dtype, float(logit_div),
                                          params.data_ptr(), params.stride(0), _stream())
    with torch.cuda.device(x.device):
        code = launch() if _timer is None else _timer.record(('made_rqs', pf.inp, pf.features * P, B), launch)
        _check(code, 'nf_made_rqs_fwd_params')
    tag_max(out, y_max)
    return out, logabsdet, params


FAMILY_CODE = {'linear': 1, 'quadratic': 2, 'cubic': 3}


def inverse_params_per_feature(kind, num_bins, tails):
    t = tails == 'linear'
    return {'rq': 3 * num_bins - 1 if t else 3 * num_bins + 1, 'affine': 2, 'linear': num_bins,
            'quadratic': 2 * num_bins - 1 if t else 2 * num_bins + 1, 'cubic': 2 * num_bins + 2}[kind]


def inverse_rows_supported(kind, features, hidden, n_steps, num_bins, tails):
    """Does the row-resident inverse kernel take this network / elementwise transform?"""
    lib = load()
    t = 1 if tails == 'linear' else 0
    if tails not in (None, 'linear'):
        return False
    if kind == 'rq':
        return lib.nf_made_rqs_inverse_rows_supported(int(features), int(hidden), int(n_steps), int(num_bins), t) == 1
    if kind == 'affine':
        return lib.nf_made_affine_inverse_rows_supported(int(features), int(hidden), int(n_steps)) == 1
    return lib.nf_made_family_inverse_rows_supported(FAMILY_CODE[kind], int(features), int(hidden), int(n_steps),
                                                     int(num_bins), t) == 1


class PackedInverse:
    """Parameter blob of nf_made_rqs_inverse_rows (layout: include/nf_hip.h) for a MADE whose hidden units are in
    degree order.  `steps` = [(weight, mask, bias, flags, (bn_scale, bn_shift) or None)] of the hidden stack, then
    the final layer."""
    __slots__ = ('hidden', 'features', 'P', 'K', 'tails', 'n_steps', 'flags', 'blob')

    def __init__(self, steps, final_weight, final_mask, final_bias, features, num_bins, tails, kind='rq'):
        """kind: 'rq' (3K-1 / 3K+1 parameters per feature), 'affine' (2), 'linear' (K), 'quadratic' (2K-1 / 2K+1),
        'cubic' (2K+2)."""
        lib = load()
        H = int(final_weight.shape[1])
        t = 1 if tails == 'linear' else 0
        num_bins = 0 if num_bins is None else int(num_bins)
        P = inverse_params_per_feature(kind, num_bins, tails)
        HS = (H + 63) // 64 * 64
        dev = final_weight.device
        parts, flags = [], []
        for w, m, b, fl, _ in steps:
            w = _dev(w.detach(), 'weight')
            wm = w if m is None else w * m
            wt = torch.zeros((H + 1, HS), dtype=torch.float32, device=dev)
            wt[:H, :H] = wm[:, :H].t()
            if b is not None:
                wt[H, :H] = b.detach()
            parts.append(wt.reshape(-1))
            flags.append(int(fl))
        wf = _dev(final_weight.detach(), 'weight')
        if final_mask is not None:
            wf = wf * final_mask
        wf = wf.view(features, P, H)[:H + 1]
        bf = torch.zeros((H + 1, 32), dtype=torch.float32, device=dev)
        if final_bias is not None:
            bf[:, :P] = final_bias.detach().view(features, P)[:H + 1]
        parts.append(bf.reshape(-1))
        # the two newest units each feature reads join as rank-1 terms: wl[f][0][c] = W[f*P+c][f-1], [1]: [f-2]
        wl = torch.zeros((H + 1, 2, 32), dtype=torch.float32, device=dev)
        idx = torch.arange(1, H + 1, device=dev)
        wl[1:, 0, :P] = wf[idx, :, idx - 1]
        wl[2:, 1, :P] = wf[idx[1:], :, idx[1:] - 2]
        parts.append(wl.reshape(-1))
        # hidden layers, unit j: dg[j][s] = step s's weight on unit j of the step before (s = 0: on input j),
        # dg[j][8 + s] = its weight on unit j-1, dg[j][16 + s] / [24 + s] = the eval-BatchNorm scale / shift applied
        # to unit j of step s-1 on its way into step s (1 / 0 without BatchNorm)
        dg = torch.zeros((H, 32), dtype=torch.float32, device=dev)
        dg[:, 16:24] = 1.0
        ar = torch.arange(H, device=dev)
        for s, (w, m, b, fl, bn) in enumerate(steps):
            wm = w.detach() if m is None else w.detach() * m
            dg[:, s] = wm[ar, ar]
            dg[1:, 8 + s] = wm[ar[1:], ar[:-1]]
            if bn is not None:
                dg[:, 16 + s] = bn[0].detach()
                dg[:, 24 + s] = bn[1].detach()
        parts.append(dg.reshape(-1))
        nss = (torch.arange(H + 2, device=dev) + 7) // 8                  # super-steps of feature f
        ns_max = (H + 7) // 8
        dense = torch.zeros((H + 1, 32, ns_max * 8), dtype=torch.float32, device=dev)
        dense[:, :P, :H] = wf
        fi = torch.arange(H + 1, device=dev)[:, None, None]
        dense[(torch.arange(ns_max * 8, device=dev)[None, None, :] >= fi - 2).expand_as(dense)] = 0.0   # (rank-1 terms)
        dense = dense.view(H + 1, 32, ns_max, 2, 4).permute(0, 2, 3, 1, 4)         # [f][S][kh][col][4]
        live = torch.arange(ns_max, device=dev)[None, :] < nss[:H + 1, None]
        parts.append(dense[live].reshape(-1))
        self.blob = torch.cat(parts).contiguous()
        self.hidden, self.features, self.P, self.K, self.tails = H, int(features), P, int(num_bins), t
        self.n_steps, self.flags = len(steps), (ctypes.c_int * len(flags))(*flags)
        if self.blob.numel() != lib.nf_made_inverse_blob_floats(H, self.n_steps):
            raise RuntimeError('inverse blob has %d floats, the library expects %d'
                               % (self.blob.numel(), lib.nf_made_inverse_blob_floats(H, self.n_steps)))


def made_rqs_inverse_rows_supported(features, hidden, n_steps, num_bins, tails):
    return (tails in (None, 'linear')
            and load().nf_made_rqs_inverse_rows_supported(int(features), int(hidden), int(n_steps), int(num_bins),
                                                          1 if tails == 'linear' else 0) == 1)


def made_affine_inverse_rows_supported(features, hidden, n_steps):
    return load().nf_made_affine_inverse_rows_supported(int(features), int(hidden), int(n_steps)) == 1


def _context_args(ctx, B, hidden):
    """(context term, gate of block 0, gate of block 1) [B, hidden] tensors -> pointers + common row stride."""
    if ctx is None:
        return None, None, None, 0
    stride = None
    ptrs = []
    for t in ctx:
        if t is None:
            ptrs.append(None)
            continue
        _dev(t, 'context term')
        p, s = _rows(t, 'context term')
        if t.shape != (B, hidden) or (stride is not None and s != stride):
            raise ValueError('context terms must be [B,%d] tensors with one row stride' % hidden)
        stride = s
        ptrs.append(p)
    return ptrs[0], ptrs[1], ptrs[2], stride


def made_affine_inverse_rows(pi, z, out, hidden, ctx=None):
    """Features 0..H of the affine autoregressive inverse and the final hidden state (nf_made_affine_inverse_rows).
    ctx: None or (context term, gate0, gate1) as _context_args takes them."""
    lib = load()
    _dev(z, 'inputs')
    _dev(out, 'outputs')
    _dev(hidden, 'hidden')
    B = z.shape[0]
    zp, zs = _rows(z, 'inputs')
    xp, xs = _rows(out, 'outputs')
    hp, hs = _rows(hidden, 'hidden')
    if pi.P != 2 or z.shape[1] != pi.features or out.shape != z.shape or hidden.shape != (B, pi.hidden):
        raise ValueError('Expected an affine pack, inputs / outputs [B,%d] and hidden [B,%d]' % (pi.features, pi.hidden))

    c0, g0, g1, cs = _context_args(ctx, B, pi.hidden)

    def launch():
        return lib.nf_made_affine_inverse_rows(pi.blob.data_ptr(), pi.hidden, pi.n_steps, pi.flags, zp, zs, xp, xs, hp,
                                               hs, c0, g0, g1, cs, B, pi.features, _stream())
    with torch.cuda.device(z.device):
        code = launch() if _timer is None else _timer.record(('made_affine_inverse_rows', pi.hidden, pi.features, B), launch)
        _check(code, 'nf_made_affine_inverse_rows')


def made_family_inverse_rows(pi, kind, z, out, hidden, tail_bound=1.0, min_bin_width=1e-3, min_bin_height=1e-3,
                             pre_div=1.0, ctx=None):
    """Features 0..H of the autoregressive inverse of a linear / quadratic / cubic spline transform and the final
    hidden state (nf_made_family_inverse_rows)."""
    lib = load()
    _dev(z, 'inputs')
    _dev(out, 'outputs')
    _dev(hidden, 'hidden')
    B = z.shape[0]
    zp, zs = _rows(z, 'inputs')
    xp, xs = _rows(out, 'outputs')
    hp, hs = _rows(hidden, 'hidden')
    if z.shape[1] != pi.features or out.shape != z.shape or hidden.shape != (B, pi.hidden):
        raise ValueError('Expected inputs / outputs [B,%d] and hidden [B,%d]' % (pi.features, pi.hidden))
    c0, g0, g1, cs = _context_args(ctx, B, pi.hidden)

    def launch():
        return lib.nf_made_family_inverse_rows(FAMILY_CODE[kind], pi.blob.data_ptr(), pi.hidden, pi.n_steps, pi.flags,
                                               zp, zs, xp, xs, hp, hs, c0, g0, g1, cs, B, pi.features, pi.K, pi.tails,
                                               float(tail_bound), float(min_bin_width), float(min_bin_height),
                                               float(pre_div), _stream())
    with torch.cuda.device(z.device):
        code = launch() if _timer is None else _timer.record(('made_family_inverse_rows', pi.hidden, pi.features, B), launch)
        _check(code, 'nf_made_family_inverse_rows')


def made_rqs_inverse_rows(pi, z, out, hidden, workspace, tail_bound, min_bin_width=1e-3, min_bin_height=1e-3,
                          min_derivative=1e-3, ctx=None):
    """Features 0..H of the autoregressive inverse for every row of z (nf_made_rqs_inverse_rows); leaves the final
    hidden state in `hidden` and the logdet partials of the finished fused tiles in `workspace`."""
    lib = load()
    _dev(z, 'inputs')
    _dev(out, 'outputs')
    _dev(hidden, 'hidden')
    B = z.shape[0]
    zp, zs = _rows(z, 'inputs')
    xp, xs = _rows(out, 'outputs')
    hp, hs = _rows(hidden, 'hidden')
    if z.shape[1] != pi.features or out.shape != z.shape or hidden.shape != (B, pi.hidden):
        raise ValueError('Expected inputs / outputs [B,%d] and hidden [B,%d]' % (pi.features, pi.hidden))

    c0, g0, g1, cs = _context_args(ctx, B, pi.hidden)

    def launch():
        return lib.nf_made_rqs_inverse_rows(pi.blob.data_ptr(), pi.hidden, pi.n_steps, pi.flags, zp, zs, xp, xs, hp, hs,
                                            c0, g0, g1, cs, workspace.data_ptr(), workspace.numel() * 4, B, pi.features, pi.K,
                                            pi.tails, float(tail_bound), float(min_bin_width), float(min_bin_height),
                                            float(min_derivative), _stream())
    with torch.cuda.device(z.device):
        code = launch() if _timer is None else _timer.record(('made_rqs_inverse_rows', pi.hidden, pi.features, B), launch)
        _check(code, 'nf_made_rqs_inverse_rows')


def coupling_layer_supported(row_width, n_in, hidden, features, num_bins, tails):
    return tails in (None, 'linear') and load().nf_coupling_layer_supported(
        int(row_width), int(n_in), int(hidden), int(features), int(num_bins), 1 if tails == 'linear' else 0) == 1


class PackedCouplingFinal:
    """Device buffers of nf_coupling_final_pack: the conditioner's final Linear with the logit rows folded."""
    __slots__ = ('features', 'hidden', 'K', 'tails', 'w', 'b')

    def __init__(self, weight, bias, features, num_bins, tails, logit_div):
        lib = load()
        w = _dev(weight.detach().contiguous(), 'weight')
        b = None if bias is None else _dev(bias.detach().contiguous(), 'bias')
        t = 1 if tails == 'linear' else 0
        self.features, self.hidden, self.K, self.tails = int(features), int(w.shape[1]), int(num_bins), t
        P = 3 * self.K - 1 if t else 3 * self.K + 1
        if w.shape[0] != self.features * P:
            raise ValueError('final layer must have %d outputs' % (self.features * P))
        nbytes = lib.nf_coupling_final_packed_bytes(self.features, self.hidden, self.K, t)
        if nbytes == 0:
            raise NfError(-2, 'nf_coupling_final_pack')
        self.w = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        self.b = torch.empty(w.shape[0], dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            _check(lib.nf_coupling_final_pack(w.data_ptr(), _ptr(b), self.features, self.hidden, self.K, t,
                                              float(logit_div), self.w.data_ptr(), self.b.data_ptr(), _stream()),
                   'nf_coupling_final_pack')


def coupling_rqs_layer(x, in_index, feat_index, blob, hidden, num_blocks, pcf, tail_bound, inverse=False,
                       min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, want_bins=False):
    """One RQ-spline coupling layer in one launch (nf_coupling_rqs_layer_fwd): x [B, D] -> (y [B, D], logabsdet [B],
    bins [B, features] or None).  in_index / feat_index: int32 identity / transformed columns."""
    lib = load()
    _dev(x, 'inputs')
    _dev(in_index, 'in_index', torch.int32)
    _dev(feat_index, 'feat_index', torch.int32)
    xp, xs = _rows(x, 'inputs')
    B = x.shape[0]
    out = torch.empty_like(x)
    yp, ys = _rows(out, 'outputs')
    logabsdet = torch.empty(B, dtype=torch.float32, device=x.device)
    bins = torch.empty((B, pcf.features), dtype=torch.int32, device=x.device) if want_bins else None

    def launch():
        return lib.nf_coupling_rqs_layer_fwd(xp, xs, yp, ys, x.shape[1], in_index.data_ptr(), in_index.numel(),
                                             feat_index.data_ptr(), pcf.features, blob.data_ptr(), hidden, num_blocks,
                                             pcf.w.data_ptr(), pcf.b.data_ptr(), logabsdet.data_ptr(), 0, _ptr(bins), B,
                                             pcf.K, pcf.tails, float(tail_bound), float(min_bin_width),
                                             float(min_bin_height), float(min_derivative), int(inverse), _stream())
    with torch.cuda.device(x.device):
        code = launch() if _timer is None else _timer.record(('coupling_layer', in_index.numel(), pcf.features, B), launch)
        _check(code, 'nf_coupling_rqs_layer_fwd')
    return out, logabsdet, bins


def resnet_hidden_supported(n_in, hidden):
    return 0 < n_in <= 128 and 0 < hidden <= 128 and hidden % 32 == 0


def resnet_hidden(x, in_index, blob, n_in, hidden, num_blocks, copy_out=None):
    """Hidden state [B, hidden] of a ResidualNet (blob layout: nf_resnet_blob_floats) on columns `in_index`
    (int32, or None for the first n_in) of x.  copy_out: a [B, width] tensor that receives those columns of x at the
    same positions (nf_resnet_hidden_fwd_copy)."""
    lib = load()
    _dev(x, 'inputs')
    _dev(blob, 'blob')
    if in_index is not None:
        _dev(in_index, 'in_index', torch.int32)
        if in_index.numel() != n_in:
            raise ValueError('in_index must have %d entries' % n_in)
    if blob.numel() != lib.nf_resnet_blob_floats(n_in, hidden, num_blocks):
        raise ValueError('parameter blob has %d floats, expected %d'
                         % (blob.numel(), lib.nf_resnet_blob_floats(n_in, hidden, num_blocks)))
    xp, xs = _rows(x, 'inputs')
    B = x.shape[0]
    h = torch.empty((B, hidden), dtype=torch.float32, device=x.device)

    if copy_out is not None:
        _dev(copy_out, 'copy_out')
        if copy_out.shape != x.shape:
            raise ValueError('copy_out must have the shape of the inputs')
        cp, cs = _rows(copy_out, 'copy_out')

    def launch():
        if copy_out is not None:
            return lib.nf_resnet_hidden_fwd_copy(xp, xs, _ptr(in_index), n_in, x.shape[1], blob.data_ptr(), hidden,
                                                 num_blocks, h.data_ptr(), hidden, cp, cs, B, _stream())
        return lib.nf_resnet_hidden_fwd(xp, xs, _ptr(in_index), n_in, x.shape[1], blob.data_ptr(), hidden, num_blocks,
                                        h.data_ptr(), hidden, B, _stream())
    with torch.cuda.device(x.device):
        code = launch() if _timer is None else _timer.record(('resnet_hidden', n_in, hidden, B), launch)
        _check(code, 'nf_resnet_hidden_fwd')
    return h


def masked_linear(x, pl, residual=None, in_scale=None, in_shift=None, relu_in=False, relu_out=False, out=None,
                  tiles=None):
    """`tiles=(begin, count)` restricts the launch to those 128-column output tiles of `out`."""
    lib = load()
    _dev(x, 'x')
    xp, xs = _rows(x, 'x')
    B = x.shape[0]
    if x.shape[1] != pl.inp:
        raise ValueError('Expected %d input features, got %d' % (pl.inp, x.shape[1]))
    if out is None:
        out = torch.empty((B, pl.out), dtype=torch.float32, device=x.device)
    yp, ys = _rows(out, 'out')
    if getattr(out, '_nf_max', None) is not None:
        tag_max(out, None)              # a split-operand launch's max |out| no longer describes this buffer
    rp, rs = (None, 0) if residual is None else _rows(_dev(residual, 'residual'), 'residual')
    tb, tc = (0, (pl.out + 127) // 128) if tiles is None else tiles

    def launch():
        return lib.nf_masked_linear_fwd_tiles(xp, xs, pl.packed.data_ptr(), pl.ktab.data_ptr(), _ptr(pl.bias), rp, rs,
                                              _ptr(in_scale), _ptr(in_shift), yp, ys, B, pl.inp, pl.out,
                                              int(relu_in), int(relu_out), pl.dtype, int(tb), int(tc), _stream())
    with torch.cuda.device(x.device):
        code = launch() if _timer is None else _timer.record(('masked_linear', pl.inp, pl.out, B), launch)
        _check(code, 'nf_masked_linear_fwd')
    return out


def rqs(x, params, num_bins, tails, tail_bound, inverse=False, pre_div=1.0, feat_index=None, out=None,
        logabsdet=None, accumulate=False, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3,
        want_bins=False, num_features=None):
    """x [B, >=D] rows, params [B, D*P] -> (y, logabsdet[B], bins or None)."""
    lib = load()
    _dev(x, 'inputs')
    _dev(params, 'params')
    B = x.shape[0]
    D = int(num_features if num_features is not None else x.shape[1])
    xp, xs = _rows(x, 'inputs')
    pp, ps = _rows(params, 'params')
    if out is None:
        out = torch.empty_like(x)
    else:
        tag_max(out, None)              # (a max slot left by an earlier writer no longer describes this buffer)
    yp, ys = _rows(out, 'outputs')
    if logabsdet is None:
        logabsdet = torch.empty(B, dtype=torch.float32, device=x.device)
        accumulate = False
    bins = torch.empty((B, D), dtype=torch.int32, device=x.device) if want_bins else None
    def launch():
        return lib.nf_rqs_fwd(xp, xs, _ptr(feat_index), pp, ps, float(pre_div), yp, ys, logabsdet.data_ptr(),
                              int(accumulate), _ptr(bins), B, D, int(num_bins), 1 if tails == 'linear' else 0,
                              float(tail_bound), float(min_bin_width), float(min_bin_height),
                              float(min_derivative), int(inverse), _stream())
    with torch.cuda.device(x.device):
        code = launch() if _timer is None else _timer.record(('rqs', D, int(num_bins), B), launch)
        _check(code, 'nf_rqs_fwd')
    return out, logabsdet, bins


def _family_call(x, params, out, logabsdet, accumulate, want_bins, num_features):
    _dev(x, 'inputs')
    _dev(params, 'params')
    B = x.shape[0]
    D = int(num_features if num_features is not None else x.shape[1])
    xp, xs = _rows(x, 'inputs')
    pp, ps = _rows(params, 'params')
    if out is None:
        out = torch.empty_like(x)
    else:
        tag_max(out, None)              # (a max slot left by an earlier writer no longer describes this buffer)
    yp, ys = _rows(out, 'outputs')
    if logabsdet is None:
        logabsdet = torch.empty(B, dtype=torch.float32, device=x.device)
        accumulate = False
    bins = torch.empty((B, D), dtype=torch.int32, device=x.device) if want_bins else None
    return B, D, xp, xs, pp, ps, out, yp, ys, logabsdet, accumulate, bins


def linear_spline(x, params, num_bins, tails=None, tail_bound=1.0, inverse=False, feat_index=None, out=None,
                  logabsdet=None, accumulate=False, want_bins=False, num_features=None):
    """Piecewise-linear CDF spline: params [B, D*K] unnormalised pdf -> (y, logabsdet[B], bins or None)."""
    lib = load()
    B, D, xp, xs, pp, ps, out, yp, ys, logabsdet, accumulate, bins = _family_call(
        x, params, out, logabsdet, accumulate, want_bins, num_features)
    with torch.cuda.device(x.device):
        _check(lib.nf_linear_spline_fwd(xp, xs, _ptr(feat_index), pp, ps, yp, ys, logabsdet.data_ptr(), int(accumulate),
                                        _ptr(bins), B, D, int(num_bins), 1 if tails == 'linear' else 0,
                                        float(tail_bound), int(inverse), _stream()), 'nf_linear_spline_fwd')
    return out, logabsdet, bins


def quadratic_spline(x, params, num_bins, tails=None, tail_bound=1.0, inverse=False, feat_index=None, out=None,
                     logabsdet=None, accumulate=False, min_bin_width=1e-3, min_bin_height=1e-3, want_bins=False,
                     num_features=None):
    """Piecewise-quadratic CDF spline: params [B, D*(2K+1)] (or 2K-1 with linear tails)."""
    lib = load()
    B, D, xp, xs, pp, ps, out, yp, ys, logabsdet, accumulate, bins = _family_call(
        x, params, out, logabsdet, accumulate, want_bins, num_features)
    with torch.cuda.device(x.device):
        _check(lib.nf_quadratic_spline_fwd(xp, xs, _ptr(feat_index), pp, ps, yp, ys, logabsdet.data_ptr(),
                                           int(accumulate), _ptr(bins), B, D, int(num_bins),
                                           1 if tails == 'linear' else 0, float(tail_bound), float(min_bin_width),
                                           float(min_bin_height), int(inverse), _stream()), 'nf_quadratic_spline_fwd')
    return out, logabsdet, bins


def cubic_spline(x, params, num_bins, tails=None, bound=10.0, inverse=False, pre_div=1.0, feat_index=None, out=None,
                 logabsdet=None, accumulate=False, min_bin_width=1e-3, min_bin_height=1e-3, want_bins=False,
                 num_features=None):
    """Monotone piecewise-cubic spline on [-bound, bound]: params [B, D*(2K+2)]."""
    lib = load()
    B, D, xp, xs, pp, ps, out, yp, ys, logabsdet, accumulate, bins = _family_call(
        x, params, out, logabsdet, accumulate, want_bins, num_features)
    with torch.cuda.device(x.device):
        _check(lib.nf_cubic_spline_fwd(xp, xs, _ptr(feat_index), pp, ps, float(pre_div), yp, ys, logabsdet.data_ptr(),
                                       int(accumulate), _ptr(bins), B, D, int(num_bins), 1 if tails == 'linear' else 0,
                                       float(bound), float(min_bin_width), float(min_bin_height), int(inverse),
                                       _stream()), 'nf_cubic_spline_fwd')
    return out, logabsdet, bins


def family_spline(kind, x, params, num_bins, tails=None, tail_bound=1.0, inverse=False, pre_div=1.0,
                  min_bin_width=1e-3, min_bin_height=1e-3, **kw):
    """The linear / quadratic / cubic spline by name (same arguments for all three; `tail_bound` is the cubic's
    `bound`)."""
    if kind == 'linear':
        return linear_spline(x, params, num_bins, tails=tails, tail_bound=tail_bound, inverse=inverse, **kw)
    if kind == 'quadratic':
        return quadratic_spline(x, params, num_bins, tails=tails, tail_bound=tail_bound, inverse=inverse,
                                min_bin_width=min_bin_width, min_bin_height=min_bin_height, **kw)
    if kind == 'cubic':
        return cubic_spline(x, params, num_bins, tails=tails, bound=tail_bound, inverse=inverse, pre_div=pre_div,
                            min_bin_width=min_bin_width, min_bin_height=min_bin_height, **kw)
    raise ValueError(kind)


def family_spline_bwd(kind, x, params, grad_y, grad_lad, num_bins, tails=None, tail_bound=1.0, inverse=False,
                      pre_div=1.0, min_bin_width=1e-3, min_bin_height=1e-3):
    """(grad wrt the spline's input, grad_params) of nf_{linear,quadratic,cubic}_spline_fwd.  inverse: backward of the
    inverse direction - `x` is then that pass's OUTPUT and grad_y the gradient wrt it."""
    lib = load()
    for t, n in ((x, 'inputs'), (params, 'params'), (grad_y, 'grad_y'), (grad_lad, 'grad_logabsdet')):
        _dev(t, n)
    B, D = x.shape
    xp, xs = _rows(x, 'inputs')
    pp, ps = _rows(params, 'params')
    gyp, gys = _rows(grad_y, 'grad_y')
    gx = torch.empty((B, D), dtype=torch.float32, device=x.device)
    gxp, gxs = _rows(gx, 'grad_x')
    gparams = torch.empty_like(params)
    gl = grad_lad.contiguous()
    t = 1 if tails == 'linear' else 0
    with torch.cuda.device(x.device):
        if kind == 'linear':
            code = lib.nf_linear_spline_bwd(xp, xs, pp, ps, gyp, gys, gl.data_ptr(), gxp, gxs, gparams.data_ptr(), B, D,
                                            int(num_bins), t, float(tail_bound), int(inverse), _stream())
        elif kind == 'quadratic':
            code = lib.nf_quadratic_spline_bwd(xp, xs, pp, ps, gyp, gys, gl.data_ptr(), gxp, gxs, gparams.data_ptr(),
                                               B, D, int(num_bins), t, float(tail_bound), float(min_bin_width),
                                               float(min_bin_height), int(inverse), _stream())
        elif kind == 'cubic':
            code = lib.nf_cubic_spline_bwd(xp, xs, pp, ps, float(pre_div), gyp, gys, gl.data_ptr(), gxp, gxs,
                                           gparams.data_ptr(), B, D, int(num_bins), t, float(tail_bound),
                                           float(min_bin_width), float(min_bin_height), int(inverse), _stream())
        else:
            raise ValueError(kind)
        _check(code, 'nf_%s_spline_bwd' % kind)
    return gx, gparams


def affine(x, params, layout, inverse=False, feat_index=None, out=None, logabsdet=None, accumulate=False,
           num_features=None):
    lib = load()
    _dev(x, 'inputs')
    _dev(params, 'params')
    B = x.shape[0]
    D = int(num_features if num_features is not None else x.shape[1])
    xp, xs = _rows(x, 'inputs')
    pp, ps = _rows(params, 'params')
    if out is None:
        out = torch.empty_like(x)
    else:
        tag_max(out, None)              # (a max slot left by an earlier writer no longer describes this buffer)
    yp, ys = _rows(out, 'outputs')
    if logabsdet is None:
        logabsdet = torch.empty(B, dtype=torch.float32, device=x.device)
        accumulate = False
    with torch.cuda.device(x.device):
        _check(lib.nf_affine_fwd(xp, xs, _ptr(feat_index), pp, ps, int(layout), yp, ys, logabsdet.data_ptr(),
                                 int(accumulate), B, D, int(inverse), _stream()), 'nf_affine_fwd')
    return out, logabsdet


def std_normal_log_prob(z, logabsdet=None, lad_scale=1.0):
    """[B] = lad_scale * logabsdet + log N(z; 0, I), one launch (z: [B, D] with contiguous rows)."""
    lib = load()
    _dev(z, 'noise')
    zp, zs = _rows(z, 'noise')
    B, D = z.shape
    out = torch.empty(B, dtype=torch.float32, device=z.device)
    lad = None if logabsdet is None else _dev(logabsdet.contiguous(), 'logabsdet')
    with torch.cuda.device(z.device):
        _check(lib.nf_std_normal_log_prob(zp, zs, _ptr(lad), float(lad_scale), out.data_ptr(), B, D, _stream()),
               'nf_std_normal_log_prob')
    return out


def affine_bwd(x, params, layout, grad_y, grad_lad, inverse=False, feat_index=None, num_features=None):
    """(grad_x, grad_params) of `affine`.  With `feat_index` (coupling) grad_x starts as a copy of grad_y: the
    identity columns pass their gradient through and the kernel overwrites the transformed ones."""
    lib = load()
    for t, n in ((x, 'inputs'), (params, 'params'), (grad_y, 'grad_y')):
        _dev(t, n)
    B = x.shape[0]
    D = int(num_features if num_features is not None else x.shape[1])
    xp, xs = _rows(x, 'inputs')
    pp, ps = _rows(params, 'params')
    gyp, gys = _rows(grad_y, 'grad_y')
    gx = grad_y.clone() if feat_index is not None else torch.empty_like(x)
    gxp, gxs = _rows(gx, 'grad_x')
    gparams = torch.empty((B, 2 * D), dtype=torch.float32, device=x.device)
    gpp, gps = _rows(gparams, 'grad_params')
    gl = None if grad_lad is None else _dev(grad_lad.contiguous(), 'grad_logabsdet')
    with torch.cuda.device(x.device):
        _check(lib.nf_affine_bwd(xp, xs, _ptr(feat_index), pp, ps, int(layout), gyp, gys, _ptr(gl), gxp, gxs, gpp, gps,
                                 B, D, int(inverse), _stream()), 'nf_affine_bwd')
    return gx, gparams


# ------------------------------------------------------------------------------------------------
class KernelTimer:
    """Optional HIP-event timing of individual launches on the stream they are launched on
    (bench.py's roofline leg).  Off unless a bench installs one with `set_timer`."""

    def __init__(self):
        self.events = {}

    def record(self, key, launch):
        s = torch.cuda.current_stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s)
        r = launch()
        b.record(s)
        self.events.setdefault(key, []).append((a, b))
        return r

    def summary(self):
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.events.items()}


_timer = None


def set_timer(t):
    global _timer
    _timer = t


# ------------------------------------------------------------------------------------------------ fp64 MAF
def fold_f64(weight, mask):
    lib = load()
    w = _dev(weight.detach().contiguous(), 'weight', torch.float64)
    m = None if mask is None else _dev(mask.detach().to(torch.float64).contiguous(), 'mask', torch.float64)
    out = torch.empty_like(w)
    with torch.cuda.device(w.device):
        _check(lib.nf_fold_f64(w.data_ptr(), _ptr(m), w.shape[0], w.shape[1], out.data_ptr(), _stream()), 'nf_fold_f64')
    return out


def linear_f64(x, folded, bias, relu_out=False):
    lib = load()
    _dev(x, 'x', torch.float64)
    xp, xs = _rows(x, 'x')
    B = x.shape[0]
    if x.shape[1] != folded.shape[1]:
        raise ValueError('Expected %d input features, got %d' % (folded.shape[1], x.shape[1]))
    y = torch.empty((B, folded.shape[0]), dtype=torch.float64, device=x.device)
    yp, ys = _rows(y, 'y')
    b = None if bias is None else _dev(bias.detach().contiguous(), 'bias', torch.float64)
    with torch.cuda.device(x.device):
        _check(lib.nf_linear_f64_fwd(xp, xs, folded.data_ptr(), _ptr(b), yp, ys, B, folded.shape[1], folded.shape[0],
                                     int(relu_out), _stream()), 'nf_linear_f64_fwd')
    return y


def maf_fwd(z, made_out, reverse_input):
    lib = load()
    _dev(z, 'z', torch.float64)
    B, D = z.shape
    zp, zs = _rows(z, 'z')
    op, os_ = _rows(_dev(made_out, 'made_out', torch.float64), 'made_out')
    x = torch.empty_like(z)
    xp, xs = _rows(x, 'x')
    logJ = torch.empty(B, dtype=torch.float64, device=z.device)
    with torch.cuda.device(z.device):
        _check(lib.nf_maf_fwd(zp, zs, op, os_, xp, xs, logJ.data_ptr(), B, D, int(bool(reverse_input)), _stream()),
               'nf_maf_fwd')
    return x, logJ


def maf_reverse_step(x, made_out, z, dim, reverse_input, logJ=None):
    lib = load()
    B, D = x.shape
    xp, xs = _rows(_dev(x, 'x', torch.float64), 'x')
    op, os_ = _rows(_dev(made_out, 'made_out', torch.float64), 'made_out')
    zp, zs = _rows(_dev(z, 'z', torch.float64), 'z')
    with torch.cuda.device(x.device):
        _check(lib.nf_maf_reverse_step(xp, xs, op, os_, zp, zs, _ptr(logJ), B, D, int(dim), int(bool(reverse_input)),
                                       _stream()), 'nf_maf_reverse_step')


def linear_f64_bwd(x, y, folded, mask, grad_y, relu_out, want_input=True, want_weight=True, want_bias=True):
    """(grad_x, grad_w, grad_b) of linear_f64 (any of them None when not wanted); grad_w has the mask re-applied."""
    lib = load()
    _dev(x, 'x', torch.float64)
    _dev(grad_y, 'grad_y', torch.float64)
    xp, xs = _rows(x, 'x')
    gp, gs = _rows(grad_y, 'grad_y')
    yp, ys = _rows(_dev(y, 'y', torch.float64), 'y') if relu_out else (None, 0)
    B, inp = x.shape
    out = folded.shape[0]
    gx = torch.empty((B, inp), dtype=torch.float64, device=x.device) if want_input else None
    gxp, gxs = (None, 0) if gx is None else _rows(gx, 'grad_x')
    gw = torch.empty((out, inp), dtype=torch.float64, device=x.device) if want_weight else None
    gb = torch.empty(out, dtype=torch.float64, device=x.device) if want_bias else None
    m = None if mask is None else _dev(mask, 'mask', torch.float64)
    with torch.cuda.device(x.device):
        _check(lib.nf_linear_f64_bwd(xp, xs, yp, ys, folded.data_ptr(), _ptr(m), gp, gs, gxp, gxs, _ptr(gw), _ptr(gb),
                                     B, inp, out, int(relu_out), _stream()), 'nf_linear_f64_bwd')
    return gx, gw, gb


def maf_bwd(z, made_out, grad_x, grad_logJ, reverse_input):
    """(grad_z direct part, grad_made_out) of maf_fwd."""
    lib = load()
    B, D = z.shape
    zp, zs = _rows(_dev(z, 'z', torch.float64), 'z')
    op, os_ = _rows(_dev(made_out, 'made_out', torch.float64), 'made_out')
    gxp, gxs = (None, 0) if grad_x is None else _rows(_dev(grad_x, 'grad_x', torch.float64), 'grad_x')
    gl = None if grad_logJ is None else _dev(grad_logJ.contiguous(), 'grad_logJ', torch.float64)
    gz = torch.empty_like(z)
    gzp, gzs = _rows(gz, 'grad_z')
    go = torch.empty((B, 2 * D), dtype=torch.float64, device=z.device)
    gop, gos = _rows(go, 'grad_made_out')
    with torch.cuda.device(z.device):
        _check(lib.nf_maf_bwd(zp, zs, op, os_, gxp, gxs, _ptr(gl), gzp, gzs, gop, gos, B, D, int(bool(reverse_input)),
                              _stream()), 'nf_maf_bwd')
    return gz, go


def maf_reverse_step_bwd(x, made_out, grad_z_new, grad_logJ, dim, reverse_input):
    """(grad_x, grad_made_out, grad_z_old) of one out-of-place maf_reverse_step pass."""
    lib = load()
    B, D = x.shape
    xp, xs = _rows(_dev(x, 'x', torch.float64), 'x')
    op, os_ = _rows(_dev(made_out, 'made_out', torch.float64), 'made_out')
    gnp, gns = (None, 0) if grad_z_new is None else _rows(_dev(grad_z_new, 'grad_z_new', torch.float64), 'grad_z_new')
    gl = None if grad_logJ is None else _dev(grad_logJ.contiguous(), 'grad_logJ', torch.float64)
    gx = torch.empty_like(x)
    go = torch.empty((B, 2 * D), dtype=torch.float64, device=x.device)
    gzo = torch.empty_like(x)
    gxp, gxs = _rows(gx, 'grad_x')
    gop, gos = _rows(go, 'grad_made_out')
    gzp, gzs = _rows(gzo, 'grad_z_old')
    with torch.cuda.device(x.device):
        _check(lib.nf_maf_reverse_step_bwd(xp, xs, op, os_, gnp, gns, _ptr(gl), gxp, gxs, gop, gos, gzp, gzs, B, D,
                                           int(dim), int(bool(reverse_input)), _stream()), 'nf_maf_reverse_step_bwd')
    return gx, go, gzo


# ------------------------------------------------------------------------------------------------ diagnostics
def measure_device_peaks(seconds_hint=0.05):
    """(fp32-MFMA TFLOP/s, bf16-MFMA TFLOP/s on constant operands, copy GB/s counting read+write, fp16-MFMA TFLOP/s on
    pseudo-random operands) sustained by this device."""
    lib = load()
    scratch = torch.zeros(16, dtype=torch.float32, device='cuda')
    out = []
    for dtype, flop in ((NF_DTYPE_F32, 4096), (NF_DTYPE_BF16, 32768)):
        blocks, iters = 256 * 4, 20000
        best = 0.0
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _check(lib.nf_diag_mfma_loop(scratch.data_ptr(), blocks, iters, dtype, _stream()), 'nf_diag_mfma_loop')
            b.record()
            torch.cuda.synchronize()
            best = max(best, blocks * 4 * iters * 4 * flop / (a.elapsed_time(b) * 1e-3) / 1e12)
        out.append(best)
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device='cuda')
    dst = torch.empty(n, dtype=torch.uint8, device='cuda')
    best = 0.0
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _check(lib.nf_diag_stream_copy(src.data_ptr(), dst.data_ptr(), n, _stream()), 'nf_diag_stream_copy')
        b.record()
        torch.cuda.synchronize()
        best = max(best, 2 * n / (a.elapsed_time(b) * 1e-3) / 1e9)
    out.append(best)
    # fp16 MFMA on pseudo-random, rotating operands, sustained for about a second (the power controller needs that
    # long to settle): the rate the matrix pipe holds on real data under the board's power cap
    blocks, iters = 256 * 4, 20000
    flop = blocks * 4 * iters * 4 * 32768.0
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 160
    for _ in range(reps // 2):
        _check(lib.nf_diag_mfma_loop(scratch.data_ptr(), blocks, iters, NF_DTYPE_F16X2, _stream()), 'nf_diag_mfma_loop')
    a.record()
    for _ in range(reps // 2):
        _check(lib.nf_diag_mfma_loop(scratch.data_ptr(), blocks, iters, NF_DTYPE_F16X2, _stream()), 'nf_diag_mfma_loop')
    b.record()
    torch.cuda.synchronize()
    out.append((reps // 2) * flop / (a.elapsed_time(b) * 1e-3) / 1e12)
    return tuple(out)
