class Box:
    def __init__(self, low, high, shape=None, dtype=float):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
