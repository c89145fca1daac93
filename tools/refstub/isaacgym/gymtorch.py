def wrap_tensor(t):
    return t


def unwrap_tensor(t):
    return t
