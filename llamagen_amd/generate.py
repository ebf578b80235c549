"""placeholder; filled in with the engine"""
def generate(*a, **k):
    raise NotImplementedError
