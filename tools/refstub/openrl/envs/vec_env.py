class BaseVecEnv:
    pass
