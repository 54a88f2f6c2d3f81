def turbo_encode(*a, **k): raise NotImplementedError
def map_decode(*a, **k): raise NotImplementedError
def turbo_decode(*a, **k): raise NotImplementedError
