"""Channel coding on MI355X -- same public names as /root/reference/commpy/channelcoding/__init__.py:65-71
for the decoding hot path (Viterbi, BCJR/turbo, LDPC BP) and the host-side code descriptions around it."""
from commpy_amd.channelcoding.convcode import (Trellis, conv_encode, conv_encode_batch, viterbi_decode, puncturing,
                                               depuncturing)
from commpy_amd.channelcoding.interleavers import RandInterlv
from commpy_amd.channelcoding.turbo import turbo_encode, map_decode, turbo_decode
from commpy_amd.channelcoding.ldpc import (build_matrix, get_ldpc_code_params, ldpc_bp_decode, write_ldpc_params,
                                           triang_ldpc_systematic_encode)

__all__ = ['Trellis', 'conv_encode', 'conv_encode_batch', 'viterbi_decode', 'puncturing', 'depuncturing',
           'RandInterlv', 'turbo_encode', 'map_decode', 'turbo_decode', 'build_matrix', 'get_ldpc_code_params',
           'ldpc_bp_decode', 'write_ldpc_params', 'triang_ldpc_systematic_encode']
