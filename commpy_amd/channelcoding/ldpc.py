def build_matrix(*a, **k): raise NotImplementedError
def get_ldpc_code_params(*a, **k): raise NotImplementedError
def ldpc_bp_decode(*a, **k): raise NotImplementedError
def write_ldpc_params(*a, **k): raise NotImplementedError
def triang_ldpc_systematic_encode(*a, **k): raise NotImplementedError
